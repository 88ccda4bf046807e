"""ctypes binding of the C ABI in include/boundmpc_hip.h (libboundmpc_hip.so).

There is deliberately NO CPU fallback: if the HIP extension is missing or no GPU is present
the import / create call raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOUNDMPC_HIP_LIB") or os.path.join(HERE, "csrc", "libboundmpc_hip.so")   # override: A/B builds of the same source

SYMBOLS = ["bmpc_default_options", "bmpc_default_options_for", "bmpc_error_string", "bmpc_create", "bmpc_destroy", "bmpc_num_vars", "bmpc_num_cons",
           "bmpc_num_params", "bmpc_get_bounds", "bmpc_solve_batch", "bmpc_solve_batch_host", "bmpc_set_timing",
           "bmpc_last_kernel_ms", "bmpc_kernel_ms", "bmpc_launch_info", "bmpc_state_len", "bmpc_solve_batch_warm", "bmpc_graph_create",
           "bmpc_graph_launch", "bmpc_graph_destroy", "bmpc_stream_lengths", "bmpc_stream_pack", "bmpc_stream_pack_rt", "bmpc_stream_post",
           "bmpc_stream_graph_create", "bmpc_set_latency_buffer", "bmpc_stream_set_rt_feasibility_tol", "bmpc_stream_tick", "bmpc_set_team_waves", "bmpc_team_info", "bmpc_stream_set_time_budget",
           "bmpc_set_restoration", "bmpc_get_restoration", "bmpc_options_size", "bmpc_build_hash", "bmpc_set_start_rollout", "bmpc_get_start_rollout", "bmpc_set_queue_order", "bmpc_get_queue_order", "bmpc_stream_set_rt_position_row_cap", "bmpc_set_barrier_hold", "bmpc_stream_set_level_rule", "bmpc_set_second_attempt", "bmpc_get_second_attempt"]


class Options(ctypes.Structure):
    _fields_ = [("tol", ctypes.c_double), ("max_iter", ctypes.c_int), ("mu_init", ctypes.c_double),
                ("mu_min_fac", ctypes.c_double), ("slack_push", ctypes.c_double),
                ("exact_hessian", ctypes.c_int), ("verbose", ctypes.c_int), ("mu_warm", ctypes.c_double), ("stall_window", ctypes.c_int),
                ("bound_margin", ctypes.c_double)]


class BoundMPCHipError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; import it FIRST so that this process has exactly one libamdhip64
    # (loading the system runtime before torch's leaves the second one without devices).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise BoundMPCHipError(
            f"HIP extension {LIB_PATH} is missing - build it with `python -m boundmpc_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.bmpc_default_options.argtypes = [ctypes.POINTER(Options)]
    lib.bmpc_default_options_for.argtypes = [ctypes.c_int, ctypes.POINTER(Options)]
    lib.bmpc_error_string.restype = ctypes.c_char_p
    lib.bmpc_error_string.argtypes = [ci]
    lib.bmpc_create.argtypes = [ci, ci, cd, ctypes.POINTER(Options), ctypes.POINTER(vp)]
    lib.bmpc_destroy.argtypes = [vp]
    for n in ("bmpc_num_vars", "bmpc_num_cons", "bmpc_num_params"):
        getattr(lib, n).argtypes = [vp]
    lib.bmpc_get_bounds.argtypes = [vp, vp, vp, vp, vp]
    lib.bmpc_solve_batch.argtypes = [vp, ci] + [vp] * 11
    lib.bmpc_solve_batch_host.argtypes = [vp, ci] + [vp] * 10
    lib.bmpc_state_len.argtypes = [vp]
    lib.bmpc_solve_batch_warm.argtypes = [vp, ci, vp, vp, vp, ci] + [vp] * 9
    lib.bmpc_graph_create.argtypes = [vp, ci, vp, vp, vp, ci] + [vp] * 8 + [ctypes.POINTER(vp)]
    lib.bmpc_graph_launch.argtypes = [vp, vp]
    lib.bmpc_graph_destroy.argtypes = [vp]
    lib.bmpc_stream_lengths.argtypes = [vp] + [ctypes.POINTER(ci)] * 4
    lib.bmpc_stream_pack.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp]
    lib.bmpc_stream_pack_rt.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.bmpc_stream_post.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp]
    lib.bmpc_stream_set_rt_feasibility_tol.argtypes = [vp, cd]
    lib.bmpc_stream_tick.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp]
    lib.bmpc_stream_graph_create.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, ctypes.POINTER(vp)]
    lib.bmpc_set_latency_buffer.argtypes = [vp, vp]
    lib.bmpc_stream_set_time_budget.argtypes = [vp, cd]
    lib.bmpc_set_team_waves.argtypes = [vp, ci]
    lib.bmpc_team_info.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.bmpc_set_timing.argtypes = [vp, ci]
    lib.bmpc_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.bmpc_kernel_ms.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_float)]
    lib.bmpc_launch_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_longlong)]
    if os.environ.get("BOUNDMPC_HIP_LIB") and not hasattr(lib, "bmpc_set_restoration"):
        pass      # diagnostic A/B run against a build of an older revision (tests/gpu_ab.py): no restoration entry points; the in-tree library is always checked
    else:
        lib.bmpc_set_restoration.argtypes = [vp, ci, ci, ci]
        lib.bmpc_get_restoration.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
        if hasattr(lib, "bmpc_set_start_rollout"):      # (absent from A/B builds of older revisions)
            lib.bmpc_set_start_rollout.argtypes = [vp, ci]
            lib.bmpc_get_start_rollout.argtypes = [vp]
        if hasattr(lib, "bmpc_set_queue_order"):
            lib.bmpc_set_queue_order.argtypes = [vp, ci]
            lib.bmpc_get_queue_order.argtypes = [vp]
        if hasattr(lib, "bmpc_set_barrier_hold"):
            lib.bmpc_set_barrier_hold.argtypes = [vp, ci]
            lib.bmpc_stream_set_level_rule.argtypes = [vp, cd, cd, cd]
        if hasattr(lib, "bmpc_set_second_attempt"):
            lib.bmpc_set_second_attempt.argtypes = [vp, ci]
            lib.bmpc_get_second_attempt.argtypes = [vp]
        if hasattr(lib, "bmpc_stream_set_rt_position_row_cap"):
            lib.bmpc_stream_set_rt_position_row_cap.argtypes = [vp, cd]
        lib.bmpc_build_hash.restype = ctypes.c_char_p
        from . import build as _build
        want, have = _build.source_hash(), lib.bmpc_build_hash().decode()
        if not os.environ.get("BOUNDMPC_HIP_LIB") and want != have:
            raise BoundMPCHipError(f"{LIB_PATH} was built from other sources or flags (library {have}, tree {want}): rebuild with `python -m boundmpc_amd.build`")
        if lib.bmpc_options_size() != ctypes.sizeof(Options):
            raise BoundMPCHipError(f"{LIB_PATH}: bmpc_options is {lib.bmpc_options_size()} bytes, this binding expects {ctypes.sizeof(Options)} (stale build?)")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise BoundMPCHipError(f"{what} failed: {load().bmpc_error_string(rc).decode()} (code {rc})")
