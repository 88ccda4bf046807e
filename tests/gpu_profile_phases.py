"""Diagnostic (GPU box): per-phase cycle shares of the solver kernel from the -DBMPC_PROFILE build
(libboundmpc_hip_prof.so, built here on the fly; never used by the product)."""
import ctypes, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import workload

csrc = os.path.join(ROOT, "boundmpc_amd", "csrc")
# BMPC_PROF_LIB: a profile library built ahead of the GPU call (python tests/gpu_profile_phases.py --build-only, on the CPU box: build/ travels)
prof_lib = os.environ.get("BMPC_PROF_LIB") or os.path.join(ROOT, "build", "prof", "libboundmpc_hip_prof.so")
if "--build-only" in sys.argv or not os.path.exists(prof_lib):
    from boundmpc_amd import build as _b
    os.makedirs(os.path.dirname(prof_lib), exist_ok=True)
    objs = []
    procs = []
    for src in _b.UNITS:      # the translation units of the library, side by side, each with the product's flags (stamps only in the batch kernels)
        unit = os.path.splitext(os.path.basename(src))[0]
        objs.append(os.path.join(os.path.dirname(prof_lib), unit + "_prof.o"))
        procs.append(subprocess.Popen([_b.hipcc()] + _b.unit_flags(src) + ["-fPIC", "-DBMPC_PROFILE"] + os.environ.get("BMPC_PROF_DEFS", "").split() + ["-c", "-o", objs[-1], src]))
    assert all(pr.wait() == 0 for pr in procs)
    subprocess.check_call([_b.hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", prof_lib] + objs)
    if "--build-only" in sys.argv:
        print(prof_lib); sys.exit(0)
import torch
lib = ctypes.CDLL(prof_lib)
vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
lib.bmpc_create.argtypes = [ci, ci, cd, vp, ctypes.POINTER(vp)]
lib.bmpc_solve_batch.argtypes = [vp, ci] + [vp] * 11
lib.bmpc_get_profile.argtypes = [vp, vp]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
TIGHT = "tight" in sys.argv[3:]
TEAM = 4 if "team" in sys.argv[3:] else (2 if "pair" in sys.argv[3:] else (1 if "one" in sys.argv[3:] else 0))      # waves per problem: team kernels / pair kernel / one wave / the library's choice
P, X, _ = workload.make_batch(B, seed=0, N=N, tight=TIGHT)
h = vp()
assert lib.bmpc_create(N, 4, 0.1, None, ctypes.byref(h)) == 0
lib.bmpc_set_team_waves.argtypes = [vp, ci]
assert lib.bmpc_set_team_waves(h, TEAM) == 0
p, x0 = torch.tensor(P, device="cuda"), torch.tensor(X, device="cuda")
x = torch.empty_like(x0); it = torch.empty(B, dtype=torch.int32, device="cuda")
prof = np.zeros(32, dtype=np.uint64)
for rep in range(2):
    t = time.time()
    assert lib.bmpc_solve_batch(h, B, vp(p.data_ptr()), vp(x0.data_ptr()), vp(x.data_ptr()), None, None, None, None, vp(it.data_ptr()), None, None, None) == 0
    torch.cuda.synchronize(); dt = time.time() - t
    lib.bmpc_get_profile(h, vp(prof.ctypes.data))
names = ["eval", "adjoint", "kkt+mu", "qp-gradient", "prepare-rlv", "bwd:node(after p5..)", "bwd:stage-in", "forward", "step-dirs", "ls-trial", "nu-update", "(merged into nc:p1)", "st:S1 M-blocks", "st:S3d p1 of next stage", "st:S3a mfma + C", "load", "wide: rlv+iota loop", "st:S3b small roles", "nc:p2+p3 A1/A2/mu/gl", "st:S2a commit + prefetch", "st:S2b cholesky", "st:S0 q~/PR/U/PE", "st:S0b Mci/m", "st:S2 chol+gains", "bwd:staging burst", "eval:kinematics (2N lanes)", "eval:node refs+objective (N lanes)", "adjoint:node gradients (wide pass, x2)", "step-dirs: row loop", "step-dirs: grad.dz + theta loops", "st:S3c stores", "wide: rdy rows + AE loops"]
tot = float(prof.sum()); its = float(it.sum().item())
print(f"B={B} N={N}{' tight' if TIGHT else ''} waves/problem {TEAM or 'auto'} (team kernels: stamps of lane 0 of wave 0) wall {dt*1e3:.1f} ms, total iterations {its:.0f}, cycles/iteration (lane-0 stamps, profile build) {tot/its:.0f}")
for n, c in zip(names, prof):
    if c: print(f"  {n:28s} {100.0*float(c)/tot:5.1f} %   {float(c)/its:9.0f} cycles/iter")
