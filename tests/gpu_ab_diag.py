"""Diagnostic (GPU box): run tests/gpu_n30_diag.py against several builds of the library (BOUNDMPC_HIP_LIB), one subprocess each."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    print(lib, flush=True)
    subprocess.call([sys.executable, os.path.join(ROOT, "tests", "gpu_n30_diag.py")], env=dict(os.environ, BOUNDMPC_HIP_LIB=os.path.abspath(lib)))
