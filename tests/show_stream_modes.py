"""Diagnostic: one line per mode of a bench_stream.py JSON output (tick times, iterations, plans kept, distance to the converged loops, tube compliance).
Usage: python tests/show_stream_modes.py out.json"""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for r in d["results"]:
    t=r["tube_compliance_of_the_measured_states"]
    print("%-44s p50 %.2f p99 %.2f | its %.1f | applied %.3f plan %.3f | within1e-2 %.2f median dev %s | tube pos %.1e (max %.1e m) rot %.1e" % (r["mode"][:44], r["tick_ms_p50"], r["tick_ms_p99"], r["mean_iters"], r["applied_tick_fraction"], r["streams_with_a_plan_at_the_end"], r["streams_within_1e-2_rad_rms"], "%.3f phi %.2f" % (r.get("median_stream_rms_dev_rad") or 0, r.get("mean_phi_after_last_tick") or 0), t["fraction_outside_the_position_tube"], t["largest_position_excess_m"], t["fraction_outside_the_orientation_tube"]))
