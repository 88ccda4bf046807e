"""Diagnostic (GPU box): A/B timing of alternative builds of the HIP library on the bench batch.
Usage: python tests/gpu_ab.py lib1.so lib2.so ...   (each run in a fresh subprocess via BOUNDMPC_HIP_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import time, numpy as np, torch
    from boundmpc_amd import BatchedOCPSolver, workload
    for B in [int(b) for b in os.environ.get("BMPC_AB_BATCHES", "256,1024,8192").split(",")]:
        P, X, _ = workload.make_batch(B, seed=0)
        s = BatchedOCPSolver(10, 4, 0.1)
        p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
        s.set_timing(True)
        ms = []
        for r in range(4):
            o = s.solve_batch(p, x0); torch.cuda.synchronize(); ms.append(s.last_kernel_ms())
        it = o["iters"].float()
        ref = os.environ.get("BMPC_AB_REF")
        dev = ""
        if ref:
            f = f"{ref}_B{B}.npy"
            xs = o["x"].cpu().numpy()
            if os.path.exists(f):
                xr = np.load(f); q = slice(8, 15)
                d = (xs - xr).reshape(B, 10, 44)[:, :, 8:15]
                dev = f"; vs first lib: joint RMS {np.sqrt((d ** 2).mean()):.2e} max {np.abs(d).max():.2e} rad"
            else:
                np.save(f, xs)
        print(f"   B={B}: kernel ms {min(ms):.2f} (runs {['%.1f' % m for m in ms]})  -> {B/min(ms)*1e3:.0f} solves/s; iters mean {it.mean():.2f} max {int(it.max())}; ok {int((o['status']==0).sum())}{dev}", flush=True)
        s.close()
else:
    ref = os.path.join(ROOT, "gpurun_out", "ab_ref_%d" % os.getpid())
    os.makedirs(os.path.dirname(ref), exist_ok=True)
    for lib in sys.argv[1:]:
        print(lib, flush=True)
        env = dict(os.environ, BOUNDMPC_HIP_LIB=os.path.abspath(lib), BMPC_AB_REF=ref)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=env)
