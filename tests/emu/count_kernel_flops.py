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
                     "of every phase (predicated phases evaluate every role in every lane and count as executed); mask_aware: the same operations split by data flow into those "
                     "whose result reaches a store (not the dummy word, not a second store to the same word within the phase) or a decision, and the rest (tests/emu/bmpc_emu_useful.cpp; "
                     "work that every lane repeats identically, like the 8x8 Cholesky of a stage, counts as useful); oracle: the dense scalar restatement",
       "configs": []}
for (N, tight, B, seed, label) in ((10, False, 64, 0, "configs[1]/[2]: N=10"), (30, True, 16, 2, "configs[3]: N=30, tight tubes")):
    P, X, _ = workload.make_batch(B, seed=seed, N=N, tight=tight)
    k = emu.count_flops(P, X, N, 4, 0.1)
    o = c_oracle.count_flops(P, X, N, 4, 0.1)
    # mask-aware tally (tests/emu/bmpc_emu_useful.cpp): of the executed operations, those whose result reaches a store (not the dummy word, not a clamped
    # duplicate of the same phase) or a decision; data-flow graph of whole solves of the first problems (N = 30: the first 12 iterations of one solve: memory)
    us = [emu.count_useful(P[i], X[i], N, 4, 0.1, opts=None if N <= 11 else emu.default_opts(mu_init=3.0, slack_push=0.1, stall_window=20, restoration=2, max_iter=12))
          for i in range(4 if N <= 11 else 1)]
    ue, uu, ui = sum(u["executed"] for u in us), sum(u["useful"] for u in us), sum(max(u["iterations"], 1) for u in us)
    slots = sorted(set(s_ for u in us for s_ in u["per_phase"]))
    useful = {"sample": f"first {len(us)} problem(s) of the same batch" + ("" if N <= 11 else ", first 12 iterations"),
              "executed_flops_per_iteration": ue / ui, "useful_flops_per_iteration": uu / ui, "useful_fraction": uu / ue,
              "stores": sum(u["stores"] for u in us), "stores_to_the_dummy_word": sum(u["dummy_stores"] for u in us), "duplicate_stores_within_a_phase": sum(u["duplicate_stores"] for u in us),
              "by_phase_slot": {str(s_): [round(sum(u["per_phase"].get(s_, (0, 0))[0] for u in us) / ui), round(sum(u["per_phase"].get(s_, (0, 0))[1] for u in us) / ui)] for s_ in slots}}
    print(label, "useful fraction %.3f (executed %.0f, useful %.0f per iteration)" % (useful["useful_fraction"], useful["executed_flops_per_iteration"], useful["useful_flops_per_iteration"]))
    out["configs"].append({"label": label, "N": N, "tight": tight, "sample": f"first {B} problems of the seed-{seed} batch", "mask_aware": useful,
                           "kernel_text_flops_per_iteration": k["flops_per_iteration"], "kernel_text_special_per_iteration": k["special"] / k["iterations"],
                           "kernel_text_iterations": k["iterations"],
                           "kernel_text_flops_per_iteration_by_phase_slot": {str(s): round(v / k["iterations"]) for s, v in sorted(k["per_phase"].items())},
                           "oracle_dense_flops_per_iteration": o["flops_per_iteration"], "oracle_special_per_iteration": o["special"] / o["iterations"],
                           "oracle_flops_per_iteration_by_phase": {n: round(f / o["iterations"]) for n, f, _ in o["per_region"]}})
    print(label, "kernel text %.0f, oracle %.0f flops per iteration" % (k["flops_per_iteration"], o["flops_per_iteration"]))
with open(os.path.join(ROOT, "profiles", "flops_current.json"), "w") as fh:
    json.dump(out, fh, indent=1)
