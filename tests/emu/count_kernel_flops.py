"""TEST-ONLY generator of profiles/flops_current.json: fp64 operations the KERNEL TEXT (boundmpc_amd/csrc/bmpc_wave.inl) executes per
interior-point iteration, counted by the flop-counting build of the lane emulator (tests/emu/bmpc_emu_flops.cpp) on samples of the bench
batches, next to the count of the dense CPU oracle (oracle/flopcount.cpp) on the same samples.  bench.py prices roofline_fp64 with these
numbers (it reads the JSON; it never imports tests/).  Usage: python tests/emu/count_kernel_flops.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from boundmpc_amd import workload      # noqa: E402
from oracle import c_oracle            # noqa: E402
from tests.emu import emu              # noqa: E402

import bench                            # noqa: E402  (kernel_text_hash: the count is tied to the kernel text it was taken on)
out = {"kernel_hash": bench.kernel_text_hash(),
       "convention": "add / sub / mul = 1, a*b+c = 2, division / root / transcendental = 1 (tallied in `special`); kernel text: summed over the lanes "
                     "of every phase (predicated phases evaluate every role in every lane and count as executed); oracle: the dense scalar restatement",
       "configs": []}
for (N, tight, B, seed, label) in ((10, False, 64, 0, "configs[1]/[2]: N=10"), (30, True, 16, 2, "configs[3]: N=30, tight tubes")):
    P, X, _ = workload.make_batch(B, seed=seed, N=N, tight=tight)
    k = emu.count_flops(P, X, N, 4, 0.1)
    o = c_oracle.count_flops(P, X, N, 4, 0.1)
    out["configs"].append({"label": label, "N": N, "tight": tight, "sample": f"first {B} problems of the seed-{seed} batch",
                           "kernel_text_flops_per_iteration": k["flops_per_iteration"], "kernel_text_special_per_iteration": k["special"] / k["iterations"],
                           "kernel_text_iterations": k["iterations"],
                           "kernel_text_flops_per_iteration_by_phase_slot": {str(s): round(v / k["iterations"]) for s, v in sorted(k["per_phase"].items())},
                           "oracle_dense_flops_per_iteration": o["flops_per_iteration"], "oracle_special_per_iteration": o["special"] / o["iterations"],
                           "oracle_flops_per_iteration_by_phase": {n: round(f / o["iterations"]) for n, f, _ in o["per_region"]}})
    print(label, "kernel text %.0f, oracle %.0f flops per iteration" % (k["flops_per_iteration"], o["flops_per_iteration"]))
with open(os.path.join(ROOT, "profiles", "flops_current.json"), "w") as fh:
    json.dump(out, fh, indent=1)
