import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from boundmpc_amd import BatchedOCPSolver
D = np.load("build/c3_failures.npz")      # (copied from gpurun_out/ of tests/gpu_c3_failures.py run on the library BEFORE this change: build/ travels to the GPU box)
pb, xb = torch.tensor(D["p"], device="cuda"), torch.tensor(D["x0"], device="cuda")
for cap, mi in ((40, 500), (150, 1000)):
    s = BatchedOCPSolver(30, 4, 0.1, max_iter=mi); s.set_restoration(1, 6, cap)
    r = s.solve_batch(pb, xb, out={}); torch.cuda.synchronize()
    tag = f"m1_cap{cap}"
    print(f"cap {cap}: status equal {np.array_equal(r['status'].cpu().numpy(), D['st_' + tag])}, iterations equal {np.array_equal(r['iters'].cpu().numpy(), D['it_' + tag])}, x bit-equal {np.array_equal(r['x'].cpu().numpy(), D['x_' + tag])}, max |dx| {np.abs(r['x'].cpu().numpy() - D['x_' + tag]).max():.2e}", flush=True)
    s.close()
