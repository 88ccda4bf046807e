"""Diagnostic (CPU, needs hipcc): static statistics of the solver kernel's ISA per phase.

The kernel text is compiled with -DBMPC_MARKS (bmpc_hip.hip: every BMPC_PROF stamp becomes an `s_nop ; BMPCMARK id` comment in the
listing) and the listing of one kernel is cut at the marks; the instructions ahead of mark `id` belong to phase slot `id`
(names: tests/gpu_profile_phases.py).  Three views ("Reading the ISA per phase"):
  mix    instruction mix per phase: fp64 VALU, other VALU, AGPR moves, MFMA, LDS, global loads / stores, scratch, waits, branches, SALU
  trips  `s_waitcnt vmcnt` that follow at least one load since the previous one = dependent round trips to the workspace, per phase
  ops    opcode histogram of the phases given with --slots (e.g. the Riccati stage: 6,24,5,11,21,22,12,19,20,23,14,17,30,13)
Usage: python tests/isa_phase_stats.py {mix|trips|ops} [--kernel true|false|tick] [--slots a,b,...] [--asm listing.s]"""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = {"true": "_Z17bmpc_solve_kernelILb1EE", "false": "_Z17bmpc_solve_kernelILb0EE", "tick": "_Z23bmpc_stream_tick_kernelILb1EE"}      # prefixes of the mangled names


def listing(path):
    if path:
        return open(path).read().split("\n")
    from boundmpc_amd import build
    out = os.path.join(tempfile.mkdtemp(), "marks.s")
    subprocess.check_call([build.hipcc()] + build.FLAGS + ["-DBMPC_MARKS", "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "boundmpc_amd", "csrc", "bmpc_hip.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernel_lines(src, name):
    start = next(i for i, l in enumerate(src) if l.startswith(name))
    end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
    return src[start:end]


def phases(lines):
    """[(slot id, [instruction text, ...])] in listing order"""
    out, cur = [], []
    for l in lines:
        m = re.search(r"BMPCMARK (\d+)", l)
        if m:
            out.append((int(m.group(1)), cur)); cur = []
            continue
        t = l.strip()
        if t and t[0] not in ";." and not t.endswith(":"):
            cur.append(t.split(";")[0].strip())
    out.append((-1, cur))
    return out


def kind(t):
    op = t.split(" ")[0]
    if op.startswith("v_accvgpr"): return "agpr"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_") and "f64" in op: return "f64"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_load", "flat_load")): return "gld"
    if op.startswith(("global_store", "flat_store")): return "gst"
    if op.startswith("scratch_"): return "scr"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("view", choices=["mix", "trips", "ops"])
    ap.add_argument("--kernel", default="true", choices=list(KERNELS))
    ap.add_argument("--slots", default="")
    ap.add_argument("--asm", default=None)
    a = ap.parse_args()
    ph = phases(kernel_lines(listing(a.asm), KERNELS[a.kernel]))
    if a.view == "mix":
        keys = ["f64", "valu", "agpr", "mfma", "lds", "gld", "gst", "scr", "wait", "br", "salu"]
        print("slot " + " ".join("%6s" % k for k in keys) + "   VALU total")
        for sid, ins in ph:
            c = collections.Counter(kind(t) for t in ins)
            print("%4d " % sid + " ".join("%6d" % c[k] for k in keys) + "   %6d" % (c["f64"] + c["valu"] + c["agpr"] + c["mfma"]))
    elif a.view == "trips":
        print("slot  dependent-load-waits  loads")
        for sid, ins in ph:
            trips = loads = since = 0
            for t in ins:
                if t.startswith(("global_load", "scratch_load", "flat_load")):
                    since += 1; loads += 1
                elif t.startswith("s_waitcnt") and "vmcnt" in t:
                    trips += 1 if since else 0; since = 0
            print("%4d %10d %12d" % (sid, trips, loads))
    else:
        want = set(int(x) for x in a.slots.split(",") if x)
        seen, tot = set(), collections.Counter()
        for sid, ins in ph:
            if sid in want and sid not in seen:      # first occurrence of each slot
                seen.add(sid)
                tot.update(t.split(" ")[0] for t in ins)
        for k, v in tot.most_common(40):
            print("%6d %s" % (v, k))


if __name__ == "__main__":
    main()
