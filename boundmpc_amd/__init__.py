"""boundmpc_amd -- MI355X-native batched solver for the per-step optimal-control problem of
BoundMPC (Thieso/BoundMPC), behind the reference's own `solver(...)` / `BoundMPC.step()` API.

Only the hot path is here: HIP kernels + C ABI under csrc/, and the host
mirror of the reference interface.  No CPU fallback exists: without the HIP extension and a
GPU the solver constructors raise."""
from .solver import BatchedOCPSolver, NlpSolverShim  # noqa: F401
from ._lib import BoundMPCHipError, LIB_PATH  # noqa: F401

__all__ = ["BatchedOCPSolver", "NlpSolverShim", "BoundMPCHipError", "LIB_PATH"]
