"""Build the HIP extension in-tree: boundmpc_amd/csrc/libboundmpc_hip.so (gfx950 only)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libboundmpc_hip.so")
SOURCES = [os.path.join(CSRC, "bmpc_hip.hip"), os.path.join(CSRC, "bmpc_wave.inl"), os.path.join(CSRC, "bmpc_stream.inl"),
           os.path.join(HERE, "..", "include", "boundmpc_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
         "-Wno-unused-variable", "-Wno-unused-value", "-Wno-duplicate-decl-specifier"]
_COPY = re.compile(r"(v_accvgpr_(write|read)_b32|scratch_(store|load)_\w+|v_mov_b(32|64)(_e32|_e64)?) ")
_HARMLESS = re.compile(r"(s_\w+|v_readlane_b32|v_writelane_b32)( |$)")


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


def build(force=False, verbose=False, lint=True):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in SOURCES):
        return LIB
    if lint:
        # same flags, device ISA only; a hit fails the build (the compiled code would compute wrong numbers for some lanes)
        asm_dir = os.path.join(HERE, "..", "build", "isa")
        os.makedirs(asm_dir, exist_ok=True)
        asm = os.path.join(asm_dir, "bmpc_hip_gfx950.s")
        subprocess.check_call([hipcc()] + FLAGS + ["-S", "--cuda-device-only", "-o", asm, SOURCES[0]], cwd=CSRC, stderr=subprocess.DEVNULL)
        bad = lint_isa(asm)
        if bad:
            raise RuntimeError("ISA lint: register copies ahead of an exec-mask restore (compiler defect, results would be wrong): %r" % (bad,))
    # -amdgpu-sched-strategy=iterative-ilp: the solver runs at one wave per SIMD, so the scheduler should chase instruction-level
    # parallelism (loads hoisted ahead of their uses), not occupancy; measured 12.7 -> 10.8 ms at B=1024 (profiles/)
    cmd = [hipcc()] + FLAGS + ["-fPIC", "-shared", "-o", LIB, SOURCES[0]]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
