"""Collects what scripts/prof.sh, scripts/traffic.sh and scripts/pmc.sh left under gpurun_out/ into profiles/:
usage: assemble_profiles.py <tag>   (e.g. r02_b: reads gpurun_out/prof_<tag>, gpurun_out/traffic_*.json, gpurun_out/pmc_r02_*.txt)"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1]

shutil.copy(os.path.join(G, "prof_" + tag, "kernel_stats.csv"), os.path.join(P, tag + "_kernel_stats.csv"))
bench_line = open(os.path.join(G, "prof_" + tag, "bench.json")).read().strip().splitlines()[-1]
open(os.path.join(P, tag + "_bench.json"), "w").write(bench_line + "\n")
bench = json.loads(bench_line)

entries = []
for f in sorted(glob.glob(os.path.join(G, "traffic_*.json"))):
    entries += json.load(open(f))["entries"]
shas = {e["kernels_sha256"] for e in entries}
newest = max(entries, key=lambda e: os.path.getmtime(os.path.join(G, "traffic_%s_%d.json" % (e["config"], e["traversal"]))))["kernels_sha256"]
entries = [e for e in entries if e["kernels_sha256"] == newest]       # one kernel source per file
json.dump({"entries": entries}, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)


def rows(name):
    out = {}
    for line in open(os.path.join(G, name)).read().splitlines():
        m = re.match(r"(.*?)\s+(\S+)\s+sum (\S+)\s+per-dispatch (\S+) \(n=(\d+)\)", line)      # kernel names may hold spaces
        if m:
            out[m.group(2)] = float(m.group(4))
    return out


nf, ro = {}, {}
for f in ("pmc_r02_nf_a.txt", "pmc_r02_nf_b.txt", "pmc_r02_nf_c.txt"):
    nf.update(rows(f))
for f in ("pmc_r02_ro_a.txt", "pmc_r02_ro_b.txt"):
    ro.update(rows(f))

stats = {}
for row in csv.DictReader(open(os.path.join(P, tag + "_kernel_stats.csv"))):
    stats[row["Name"].split("(")[0].replace("void ", "")] = row
kt, kf = stats["rt_pairkernel_strict<true>"], stats["rt_pairkernel_strict<false>"]
tr = {(e["config"], e["traversal"]): e for e in entries}
t11, t10 = tr[("1", 1)], tr[("1", 0)]
ms_t, ms_f = float(kt["AverageNs"]) / 1e6, float(kf["AverageNs"]) / 1e6
busy = nf["SQ_BUSY_CYCLES"] / 32

L = []
L.append("rocprofv3 --pmc, rt_pairkernel_strict (two samples of a pixel per lane, 8 consecutive frames per launch), default scene")
L.append("2560x1369, 1 ray/pixel per frame, per LAUNCH of 8 frames (mean of 2), MI355X. <true>: near-first traversal of the rebuilt")
L.append("trees (the benchmark's headline flavour); in brackets <false>: reference order (bit-exact with the oracle).")
L.append("One pass per group (scripts/pmc.sh <tag> \"<counters>\" 16 2560 1369 0 3 full <traversal>):")
L.append("  a: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES")
L.append("  b: SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA")
L.append("  c: TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum   (near-first only)")
L.append("  FETCH_SIZE / WRITE_SIZE: scripts/traffic.sh (profiles/hbm_traffic.json)")
L.append("")
for k in ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
          "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY",
          "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]:
    extra = "   %+.1f %%" % (100.0 * (nf[k] / ro[k] - 1.0)) if k in ro and k != "SQ_WAVES" else ""
    L.append("%-30s %.5e   [%s]%s" % (k, nf[k], ("%.5e" % ro[k]) if k in ro else "-", extra))
L.append("FETCH_SIZE                     %.5e KB (x2 on gfx950 -> %.2f GB)   [%.5e KB -> %.2f GB]" % (
    t11["fetch_size_kb"], 2 * t11["fetch_size_kb"] * 1024 / 1e9, t10["fetch_size_kb"], 2 * t10["fetch_size_kb"] * 1024 / 1e9))
L.append("WRITE_SIZE                     %.5e KB (%.2f GB)                   [%.5e KB -> %.2f GB]" % (
    t11["write_size_kb"], t11["write_size_kb"] * 1024 / 1e9, t10["write_size_kb"], t10["write_size_kb"] * 1024 / 1e9))
L.append("")
L.append("derived (near-first):")
L.append("  kernel time %.2f ms per launch of 8 frames = %.2f ms per frame (rocprofv3 --kernel-trace --stats, profiles/%s_kernel_stats.csv:" % (ms_t, ms_t / 8, tag))
L.append("  %s launches of rt_pairkernel_strict<true>, min %.2f max %.2f; bench.py's HIP events: %.2f); reference order %.2f ms" % (
    kt["Calls"], float(kt["MinNs"]) / 1e6, float(kt["MaxNs"]) / 1e6, bench["roofline"]["launch_ms"], ms_f))
L.append("  (%s launches of <false>, min %.2f max %.2f; HIP events %.2f)" % (kf["Calls"], float(kf["MinNs"]) / 1e6, float(kf["MaxNs"]) / 1e6,
                                                                           bench["other_flavour"]["launch_ms"]))
L.append("  SQ_BUSY_CYCLES / 32 shader engines = %.3e cycles = %.1f ms at 2.4 GHz" % (busy, busy / 2.4e6))
L.append("  vector-ALU issue: SQ_INSTS_VALU x 2 cycles (a wave64 instruction occupies its SIMD-32 for two) / (cycles x 1024 SIMDs) = %.1f %%" % (
    100 * nf["SQ_INSTS_VALU"] * 2 / (busy * 1024)))
L.append("  lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.1f %%   [%.1f %%]" % (
    100 * nf["SQ_THREAD_CYCLES_VALU"] / (64 * nf["SQ_ACTIVE_INST_VALU"]), 100 * ro["SQ_THREAD_CYCLES_VALU"] / (64 * ro["SQ_ACTIVE_INST_VALU"])))
L.append("  a wave's time: issuing %.1f %%, stalled at issue %.1f %%, parked on a wait %.1f %%   [%.1f / %.1f / %.1f]" % (
    100 * nf["SQ_ACTIVE_INST_ANY"] / nf["SQ_WAVE_CYCLES"], 100 * nf["SQ_WAIT_INST_ANY"] / nf["SQ_WAVE_CYCLES"], 100 * nf["SQ_WAIT_ANY"] / nf["SQ_WAVE_CYCLES"],
    100 * ro["SQ_ACTIVE_INST_ANY"] / ro["SQ_WAVE_CYCLES"], 100 * ro["SQ_WAIT_INST_ANY"] / ro["SQ_WAVE_CYCLES"], 100 * ro["SQ_WAIT_ANY"] / ro["SQ_WAVE_CYCLES"]))
L.append("  vector L1 hit rate = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES = %.1f %%; texture addressers busy %.0f %% of the launch" % (
    100 * (1 - nf["TCP_TCC_READ_REQ_sum"] / nf["TCP_TOTAL_CACHE_ACCESSES_sum"]), 100 * nf["TA_TA_BUSY_sum"] / 256 / (nf["GRBM_GUI_ACTIVE"] / 8)))
L.append("  HBM traffic %.2f GB per launch = %.0f GB/s, %.1f %% of the HBM peak; the write half is the spilled path state: %.1e wave-level" % (
    t11["bytes_per_launch"] / 1e9, t11["bytes_per_launch"] / 1e9 / (ms_t * 1e-3), 100 * t11["bytes_per_launch"] / 1e9 / (ms_t * 1e-3) / 8000, nf["SQ_INSTS_VMEM_WR"]))
L.append("  stores per launch = about %.0f registers stored (and loaded back) around each of the ~1.8e6 wave-level GetHit calls" % (nf["SQ_INSTS_VMEM_WR"] / 1.8e6))
L.append("  (56 320 waves x 4 pairs of samples x up to 4 bounces x 2 calls), outside the walk loops.")
L.append("  Near-first against reference order: %.0f %% fewer vector instructions, %.0f %% fewer vector loads, the same lane utilisation:" % (
    100 * (1 - nf["SQ_INSTS_VALU"] / ro["SQ_INSTS_VALU"]), 100 * (1 - nf["SQ_INSTS_VMEM_RD"] / ro["SQ_INSTS_VMEM_RD"])))
L.append("  the rebuilt trees and the ordering shorten every walk, they do not make the lanes of a wave agree.")
open(os.path.join(P, tag + "_pmc_pairs.txt"), "w").write("\n".join(L) + "\n")
print("\n".join(L[9:]))
print("traffic entries:", [(e["config"], e["traversal"], round(e["bytes_per_launch"] / 1e9, 2)) for e in entries])
print("bench:", bench["value"], bench["ms_per_frame"], bench["roofline"]["frac"], bench["roofline"]["traffic"], bench["other_flavour"]["value"])
