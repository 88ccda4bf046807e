"""Build the HIP extension in-tree: boundmpc_amd/csrc/libboundmpc_hip.so (gfx950 only)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libboundmpc_hip.so")
SOURCES = [os.path.join(CSRC, "bmpc_hip.hip"), os.path.join(CSRC, "bmpc_wave.inl"), os.path.join(CSRC, "bmpc_stream.inl"),
           os.path.join(HERE, "..", "include", "boundmpc_hip.h"), os.path.join(CSRC, "bmpc_team.hip"), os.path.join(CSRC, "bmpc_gpu_common.h"),
           os.path.join(CSRC, "bmpc_resto.hip"), os.path.join(CSRC, "bmpc_tick.hip"), os.path.join(CSRC, "bmpc_pair.hip")]
# translation units of the library: the one-wave batch kernels + C ABI, the team kernels (NW cooperating waves per problem), the restoration
# kernels (the solver with the restoration phase, continuing what a batch kernel left jammed) and the fused closed-loop tick kernels
UNITS = [os.path.join(CSRC, "bmpc_hip.hip"), os.path.join(CSRC, "bmpc_team.hip"), os.path.join(CSRC, "bmpc_resto.hip"), os.path.join(CSRC, "bmpc_tick.hip"),
         os.path.join(CSRC, "bmpc_pair.hip")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
         "-Wno-unused-variable", "-Wno-unused-value", "-Wno-duplicate-decl-specifier"]
# per-unit flags (none at present; the 256-register experiment of bmpc_pair.hip wanted the default scheduler: 744 B of scratch against 1996 B)
UNIT_FLAGS = {}


def unit_flags(src):
    return UNIT_FLAGS.get(os.path.basename(src), FLAGS)


_COPY = re.compile(r"(v_accvgpr_(write|read)_b32|scratch_(store|load)_\w+|v_mov_b(32|64)(_e32|_e64)?) ")
_HARMLESS = re.compile(r"(s_\w+|v_readlane_b32|v_writelane_b32)( |$)")


def source_hash():
    """Hash of everything the library is built from: the text of every file in SOURCES and the compiler flags.  It is compiled into the
    library (bmpc_build_hash(), and as the marker BMPC_BUILD_HASH=... in its bytes): build() rebuilds when the in-tree library carries another
    hash (not by file times), _lib.load() refuses a library whose hash is not the tree's, and bench.py ties the counter / flop files in
    profiles/ to it."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(SOURCES, key=os.path.basename):
        h.update(os.path.basename(p).encode()); h.update(b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS).encode())
    for k in sorted(UNIT_FLAGS):
        h.update(("|" + k + ":" + " ".join(UNIT_FLAGS[k])).encode())
    return h.hexdigest()[:16]


def library_hash(path=None):
    """The hash a built library carries (None: no library, or one from before round 5)."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fh:
        m = re.search(rb"BMPC_BUILD_HASH=([0-9a-f]{16})", fh.read())
    return m.group(1).decode() if m else None


def lint_isa(asm_path):
    """Static check of the compiled ISA for one register-allocator defect of this toolchain (ROCm 7.2 LLVM) that silently
    corrupts results: a live-range split / spill copy placed at the top of a control-flow join block BEFORE the instruction
    that restores the exec mask (`s_or_b64 exec, exec, s[..]`), so the copy runs only for the lanes of the branch that just
    ended while every lane reads the copy later.  Seen once in this kernel (a prefetched stage input saved to an AGPR under
    the mask of lanes 21..31; DESIGN.md 4, lesson 10): N=30 solves converged to other local minima, nothing crashed.
    Signature: a basic block whose instructions ahead of its first exec restore are only scalar ops and register copies, with
    at least one vector copy among them.  Returns the list of offending (function, block, line, copies)."""
    hits, func, name, line0, block = [], None, None, 0, []

    def check():
        for j, s in enumerate(block):
            if s.startswith("s_or_b64 exec, exec, s["):
                head = block[:j]
                copies = [t for t in head if _COPY.match(t)]
                if copies and all(_COPY.match(t) or _HARMLESS.match(t) for t in head):
                    hits.append((func, name, line0, copies))
                return

    with open(asm_path) as f:
        for n, raw in enumerate(f, 1):
            s = raw.strip()
            m = re.match(r"^(_Z\w+):", s)
            if m:
                check(); func, name, line0, block = m.group(1), "entry", n, []
                continue
            m = re.match(r"^(\.LBB\w+):|^; %bb\.(\d+):", s)
            if m:
                check(); name, line0, block = (m.group(1) or "bb." + m.group(2)), n, []
                continue
            if s and s[0] not in ";.":
                block.append(s)
    check()
    return hits


_LOAD = re.compile(r"^(global_load|ds_read|scratch_load|flat_load|buffer_load)\w*\s+(v\[\d+:\d+\]|v\d+)")
_VREG = re.compile(r"v\[(\d+):(\d+)\]|(?<![a-z_\d])v(\d+)(?!\d)")
_REGION_END = ("s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec", "s_xor_b64 exec", "s_endpgm", "s_branch", "s_setpc")
_STORES = ("global_store", "ds_write", "scratch_store", "flat_store", "buffer_store", "global_atomic", "ds_add")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def _defs_uses(s):
    parts = s.split(None, 1)
    if len(parts) < 2:
        return set(), set()
    mn, ops = parts[0], [o.strip() for o in parts[1].split(",")]
    if mn.startswith(_STORES) or (mn.startswith(("s_", "v_cmp", "v_readlane", "v_readfirstlane")) and not mn.startswith("v_cmpx")):
        return set(), set().union(*[_vregs(o) for o in ops])
    d = _vregs(ops[0])
    u = set().union(*[_vregs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if mn.startswith(("v_fmac", "v_mac", "v_dot", "v_mfma", "v_writelane")):
        u |= d
    return d, u


def lint_isa_masked_loads(asm_path, window=400):
    """Second signature of the same toolchain's trouble with control flow (`cond ? state[i] : 0.0` on a possibly null
    pointer compiled into an exec-masked load whose join lost the other arm): a register whose only definition near a join is a LOAD
    executed under `s_and_saveexec` and which is READ after the matching `s_or_b64 exec, exec, ...` -- the lanes the mask switched
    off read whatever the register held.  For every such region (single block, no nested control flow) the loaded registers must
    have been written in the straight-line code ahead of the saveexec (the default arm: `v_mov`, `v_cndmask`, ...).
    Returns [(function, line of the saveexec, registers, defined_earlier)]: `defined_earlier` says whether the function writes the
    register anywhere before (in listing order) -- then the register may legitimately hold the default from further back (a long-lived
    value) and the hit is only a candidate; with no earlier write at all the inactive lanes certainly read garbage."""
    hits, func, lines = [], None, []

    def flush():
        n = len(lines)
        seen = set()       # registers written so far in listing order
        written_before = []
        for (_, s, _) in lines:
            written_before.append(set(seen))
            seen |= _defs_uses(s)[0] if s else set()
        for i, (ln, s, _) in enumerate(lines):
            # only `if` regions without an else arm: an else arm (s_andn2_saveexec / s_or_saveexec on the xor-ed mask) complements a then
            # arm that defined the register for the other lanes
            m = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\])", s)
            if not m:
                continue
            save, loads, j, closed = m.group(1), set(), i + 1, False
            while j < n and j < i + window:
                sj = lines[j][1]
                if sj.startswith("s_or_b64 exec, exec, " + save):
                    closed = True
                    break
                if sj.startswith(_REGION_END):
                    break
                d, _u = _defs_uses(sj)
                lm = _LOAD.match(sj)
                if lm:
                    loads |= _vregs(lm.group(2))
                else:
                    loads -= d          # recomputed inside the region: no longer a bare load result
                j += 1
            if not closed or not loads:
                continue
            pre, k = set(), i - 1
            while k >= 0 and k > i - window:      # straight-line code ahead of the saveexec (up to the previous label / branch / join)
                sk = lines[k][1]
                if lines[k][2] or sk.startswith(("s_or_b64 exec", "s_cbranch", "s_branch")):
                    break
                pre |= _defs_uses(sk)[0]
                k -= 1
            cand = loads - pre
            live, used, k = set(cand), set(), j + 1
            while k < n and k < j + window and live:
                sk = lines[k][1]
                if lines[k][2] or sk.startswith(("s_endpgm", "s_branch", "s_setpc", "s_cbranch")):
                    break
                d, u = _defs_uses(sk)
                used |= u & live
                live -= d
                k += 1
            if used:
                hits.append((func, ln, sorted(used), bool(used & written_before[i]) and used <= written_before[i]))

    with open(asm_path) as f:
        for n, raw in enumerate(f, 1):
            s = raw.strip()
            m = re.match(r"^(_Z\w+):", s)
            if m:
                flush(); func, lines = m.group(1), []
                continue
            if re.match(r"^(\.LBB\w+):", s):
                lines.append((n, "", True))
                continue
            if s and s[0] not in ";.":
                lines.append((n, s.split(";")[0].strip(), False))
    flush()
    return hits


def _lint_unit(src, asm, verbose):
    # same flags, device ISA only; a hit fails the build (the compiled code would compute wrong numbers for some lanes)
    subprocess.check_call([hipcc()] + unit_flags(src) + ["-S", "--cuda-device-only", "-o", asm, src], cwd=CSRC, stderr=subprocess.DEVNULL)
    bad = lint_isa(asm)
    if bad:
        raise RuntimeError("ISA lint: register copies ahead of an exec-mask restore (compiler defect, results would be wrong): %r" % (bad,))
    # masked loads read after their join: fatal in EVERY kernel of the library (round 4: the last source pattern that produced a candidate
    # -- `cond ? plan[...] : value` in the stream post-processing -- loads unconditionally behind an opaque barrier now, so the listing of
    # the shipped text has no hit at all and a new one is a change worth stopping for)
    ml = lint_isa_masked_loads(asm)
    if ml:
        raise RuntimeError("ISA lint: load under an exec mask whose result is read after the join (results may be wrong for the masked-off "
                           "lanes; `defined_earlier` = the register has an earlier definition in listing order): %r" % (ml,))


def _compile_unit(args):
    src, obj, asm, verbose, lint = args
    if lint:
        _lint_unit(src, asm, verbose)
    # -amdgpu-sched-strategy=iterative-ilp: the solver runs at one wave per SIMD, so the scheduler should chase instruction-level
    # parallelism (loads hoisted ahead of their uses), not occupancy; measured 12.7 -> 10.8 ms at B=1024 (profiles/)
    cmd = [hipcc()] + unit_flags(src) + ["-DBMPC_BUILD_HASH_STR=\"%s\"" % source_hash(), "-fPIC", "-c", "-o", obj, src]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return obj


def build(force=False, verbose=False, lint=True):
    if not force and library_hash() == source_hash():      # the library in the tree was built from exactly this text with these flags
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    out_dir = os.path.join(HERE, "..", "build", "isa")
    os.makedirs(out_dir, exist_ok=True)
    jobs = []
    for src in UNITS:
        stem = os.path.splitext(os.path.basename(src))[0]
        jobs.append((src, os.path.join(out_dir, stem + ".o"), os.path.join(out_dir, stem + "_gfx950.s"), verbose, lint))
    with ThreadPoolExecutor(len(jobs)) as ex:      # the two units compile side by side (each: ISA for the lints, then the object)
        objs = list(ex.map(_compile_unit, jobs))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
