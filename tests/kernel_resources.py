"""Diagnostic (CPU, needs hipcc): the resource table of every kernel of libboundmpc_hip.so, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage) -- registers, AGPRs, scratch per lane, SGPR / VGPR spills, LDS, occupancy.  DESIGN.md 4 quotes THIS output
(profiles/rNN_kernel_resources.txt) instead of numbers copied by hand.
Usage: python tests/kernel_resources.py [build.log]      (no argument: runs `python -m boundmpc_amd.build --force` and parses its output)"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boundmpc_amd import build      # noqa: E402

if len(sys.argv) > 1:
    text = open(sys.argv[1]).read()
else:
    text = subprocess.run([sys.executable, "-m", "boundmpc_amd.build", "--force"], cwd=ROOT, capture_output=True, text=True).stderr
rows, cur = [], None
for line in text.split("\n"):
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = {"_Z17bmpc_solve_kernelILb1EE": "bmpc_solve_kernel<ZLDS=true> (batch, N<=11, S<=4: the headline's)", "_Z17bmpc_solve_kernelILb0EE": "bmpc_solve_kernel<false> (batch, long horizons / S>4)",
         "_Z17bmpc_resto_kernelILb1EE": "bmpc_resto_kernel<true> (restoration, continues jammed problems)", "_Z17bmpc_resto_kernelILb0EE": "bmpc_resto_kernel<false>",
         "_Z22bmpc_team_solve_kernel": "bmpc_team_solve_kernel (batch, 4 waves per problem)", "_Z21bmpc_team_tick_kernelILb1EE": "bmpc_team_tick_kernel<RESTO=true> (fused tick, teams)",
         "_Z21bmpc_team_tick_kernelILb0EE": "bmpc_team_tick_kernel<false> (time-budgeted ticks)", "_Z23bmpc_stream_tick_kernelILb1ELb1EE": "bmpc_stream_tick_kernel<ZLDS=true, RESTO=true>",
         "_Z23bmpc_stream_tick_kernelILb1ELb0EE": "bmpc_stream_tick_kernel<true, false>", "_Z23bmpc_stream_tick_kernelILb0ELb1EE": "bmpc_stream_tick_kernel<false, true>",
         "_Z23bmpc_stream_tick_kernelILb0ELb0EE": "bmpc_stream_tick_kernel<false, false>", "_Z22bmpc_pair_solve_kernel": "bmpc_pair_solve_kernel (batch, 2 waves per problem, 256 < B <= 512)", "_Z18queue_order_kernel": "queue_order_kernel (ranking of a long-horizon batch)",
         "_Z23bmpc_stream_pack_kernel": "bmpc_stream_pack_kernel", "_Z23bmpc_stream_post_kernel": "bmpc_stream_post_kernel"}
print(f"kernel resources of libboundmpc_hip.so, source hash {build.source_hash()} (hipcc -Rpass-analysis=kernel-resource-usage; flags: {' '.join(build.FLAGS)})")
print("%-78s %5s %5s %8s %6s %6s %8s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "sgprS", "vgprS", "LDS B", "occ"))
seen = set()
for r in rows:
    label = next((v for k, v in names.items() if r["name"].startswith(k)), r["name"][:70])
    if label in seen:
        continue
    seen.add(label)
    print("%-78s %5d %5d %6d B %6d %6d %8d %4d" % (label, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1),
                                                   r.get("LDS Size", -1), r.get("Occupancy", -1)))
