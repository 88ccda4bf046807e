"""The C-ABI shared library loads and exports every symbol include/boundmpc_hip.h declares (no compute calls).
Also: without a GPU the product refuses to construct a solver -- there is no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "boundmpc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bmpc_[a-z_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from boundmpc_amd import _lib, build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/boundmpc_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_options_and_error_strings():
    from boundmpc_amd import _lib
    lib = _lib.load()
    o = _lib.Options()
    assert lib.bmpc_default_options(ctypes.byref(o)) == 0
    assert o.tol == 1e-8 and o.max_iter == 500 and o.exact_hessian == 1
    assert lib.bmpc_error_string(0) == b"ok" and lib.bmpc_error_string(4) == b"no HIP device available"
    h = ctypes.c_void_p()
    assert lib.bmpc_create(0, 4, 0.1, None, ctypes.byref(h)) == 1      # invalid N
    assert lib.bmpc_create(10, 9, 0.1, None, ctypes.byref(h)) == 1     # invalid S


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from boundmpc_amd import BatchedOCPSolver, BoundMPCHipError
    with pytest.raises(BoundMPCHipError):
        BatchedOCPSolver(10, 4, 0.1)


def test_product_does_not_reference_oracle_or_emulator():
    """The product package must never import, link or execute anything under oracle/ or tests/."""
    pkg = os.path.join(ROOT, "boundmpc_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".inl", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                for pat in (r"^\s*(from|import)\s+oracle", r"^\s*(from|import)\s+tests", r"libbmpc_oracle", r"libbmpc_emu"):
                    assert not re.search(pat, src, flags=re.M), f"{f} refers to test infrastructure ({pat})"


def test_isa_lint_flags_copies_ahead_of_exec_restore(tmp_path):
    """build.lint_isa: the signature of the register-allocator defect described in DESIGN.md 4 (copies ahead of the exec
    restore of a join block) is flagged, ordinary join blocks are not; the ISA of the shipped library is clean."""
    from boundmpc_amd import build as b
    bad = """_Z6kernelv:
; %bb.0:
	s_and_saveexec_b64 s[12:13], s[0:1]
; %bb.1:
	v_add_f64 v[0:1], v[2:3], v[4:5]
; %bb.2:
	s_waitcnt vmcnt(4)
	v_accvgpr_write_b32 a149, v9
	v_accvgpr_write_b32 a148, v8
	s_mov_b32 s25, s63
	s_or_b64 exec, exec, s[12:13]
	v_mov_b64_e32 v[26:27], 0
.LBB0_3:
	v_readlane_b32 s4, v255, 34
	s_or_b64 exec, exec, s[4:5]
	v_accvgpr_write_b32 a1, v2
.LBB0_4:
	v_cmp_lt_f64_e64 vcc, v[0:1], v[2:3]
	v_accvgpr_read_b32 v52, a92
	s_or_b64 exec, exec, s[6:7]
	s_endpgm
"""
    f = tmp_path / "k.s"
    f.write_text(bad)
    hits = b.lint_isa(str(f))
    assert len(hits) == 1 and hits[0][1] == "bb.2" and len(hits[0][3]) == 2
    asm = os.path.join(ROOT, "build", "isa", "bmpc_hip_gfx950.s")
    if os.path.exists(asm):
        assert b.lint_isa(asm) == []


def test_isa_lint_flags_masked_load_read_after_its_join(tmp_path):
    """build.lint_isa_masked_loads: the second miscompile signature of DESIGN.md 4 (`cond ? p[i] : 0.0` lowered to a load under an exec
    mask whose join lost the other arm): a register defined only by a masked load and read after the exec restore is flagged; the
    correct lowering (default written ahead of the saveexec) and a load consumed inside its region are not."""
    from boundmpc_amd import build as b
    bad = """_Z6kernelv:
; %bb.0:
	v_cmp_gt_i32_e32 vcc, 7, v0
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	global_load_dwordx2 v[4:5], v[2:3], off offset:232
.LBB0_2:
	s_or_b64 exec, exec, s[2:3]
	s_waitcnt vmcnt(0)
	v_fma_f64 v[6:7], v[4:5], v[8:9], v[10:11]
	s_endpgm
_Z5good1v:
; %bb.0:
	v_mov_b64_e32 v[4:5], 0
	v_cmp_gt_i32_e32 vcc, 7, v0
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB1_2
; %bb.1:
	global_load_dwordx2 v[4:5], v[2:3], off offset:232
.LBB1_2:
	s_or_b64 exec, exec, s[2:3]
	s_waitcnt vmcnt(0)
	v_fma_f64 v[6:7], v[4:5], v[8:9], v[10:11]
	s_endpgm
_Z5good2v:
; %bb.0:
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB2_2
; %bb.1:
	ds_read_b64 v[4:5], v1 offset:64
	s_waitcnt lgkmcnt(0)
	global_store_dwordx2 v[2:3], v[4:5], off
.LBB2_2:
	s_or_b64 exec, exec, s[2:3]
	v_mov_b64_e32 v[4:5], 0
	v_add_f64 v[6:7], v[4:5], v[8:9]
	s_endpgm
"""
    f = tmp_path / "m.s"
    f.write_text(bad)
    hits = b.lint_isa_masked_loads(str(f))
    assert len(hits) == 1 and hits[0][0] == "_Z6kernelv" and hits[0][2] == [4, 5] and hits[0][3] is False
    asm = os.path.join(ROOT, "build", "isa", "bmpc_hip_gfx950.s")
    for unit in ("bmpc_hip_gfx950.s", "bmpc_team_gfx950.s"):      # the shipped kernels (solver, streams, teams) have no such region at all (round 4)
        asm = os.path.join(ROOT, "build", "isa", unit)
        if os.path.exists(asm):
            assert b.lint_isa_masked_loads(asm) == []


def test_library_carries_the_hash_of_its_sources(tmp_path, monkeypatch):
    """Round 5: the library is tied to the text it was built from.  build.source_hash() covers every source file and the compiler flags; the built
    library carries it (bmpc_build_hash, and as a marker in its bytes, which build() reads instead of file times); _lib.load() refuses an in-tree
    library whose hash differs; bench.kernel_text_hash -- the key of profiles/pmc_current.json and flops_current.json -- is the same hash."""
    import bench
    from boundmpc_amd import _lib, build
    build.build()
    want = build.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", want) and build.library_hash() == want and bench.kernel_text_hash() == want
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.bmpc_build_hash.restype = ctypes.c_char_p
    assert lib.bmpc_build_hash().decode() == want and lib.bmpc_options_size() == ctypes.sizeof(_lib.Options)
    # every translation unit and every header the kernels include is part of the hash
    names = {os.path.basename(p) for p in build.SOURCES}
    assert {"bmpc_hip.hip", "bmpc_team.hip", "bmpc_resto.hip", "bmpc_tick.hip", "bmpc_wave.inl", "bmpc_stream.inl", "bmpc_gpu_common.h", "boundmpc_hip.h"} <= names
    # a library from other sources is refused at load time
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 16)
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.BoundMPCHipError, match="built from other sources"):
        _lib.load()
    monkeypatch.undo()
    _lib._lib = None
    assert _lib.load() is not None
