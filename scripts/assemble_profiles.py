"""Collects what scripts/profile_round.sh <tag> left under gpurun_out/ into profiles/:
  <tag>_kernel_stats.csv, <tag>_bench.json      kernel trace of the default bench.py command and the line it printed
  hbm_traffic.json                              FETCH_SIZE x 2 + WRITE_SIZE per launch (one entry per kernel, configuration and
                                                flavour, keyed on the hash of the kernel sources), with the issue statistics of
                                                the SQ passes attached where they were taken - what bench.py reports as
                                                roofline.traffic / valu_busy / lane_util
  <tag>_pmc.txt                                 the counter passes with the derived figures
  <tag>_wf_<config>_p<pipeline>_kernel_stats.csv   kernel traces of the wavefront pipeline (2), the path / walker waves (4), kernel 0
  <tag>_bench_cfg<config>.json                  the bench line of every configuration
usage: assemble_profiles.py <tag>"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def rows(name):
    out = {}
    path = os.path.join(G, name)
    if not os.path.exists(path):
        return out
    for line in open(path).read().splitlines():
        m = re.match(r"(.*?)\s+(\S+)\s+sum (\S+)\s+per-dispatch (\S+) \(n=(\d+)\)", line)      # kernel names may hold spaces
        if m and "rt_pairkernel" in m.group(1):
            out[m.group(2)] = float(m.group(4))
    return out


def derived(c):
    """Issue statistics of one launch from the SQ / TCP / TCC passes (dict counter -> value of the largest launch)."""
    d = {}
    if "SQ_BUSY_CYCLES" in c and "SQ_INSTS_VALU" in c:
        busy = c["SQ_BUSY_CYCLES"] / 32                      # summed over 32 shader engines
        d["busy_cycles"] = busy
        d["valu_busy"] = c["SQ_INSTS_VALU"] * 2 / (busy * 1024)      # a wave64 instruction occupies its SIMD for two cycles; 1024 SIMDs
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_ANY" in c:
        d["wave_issuing"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wave_stalled_at_issue"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wave_waiting"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        d["l1_hit"] = 1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "TCC_HIT_sum" in c:
        d["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE" in c:
        d["ta_busy"] = c["TA_TA_BUSY_sum"] / 256 / (c["GRBM_GUI_ACTIVE"] / 8)
    for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
        if k in c:
            d[k.lower()] = c[k]
    return d


# kernel trace + bench line of the default command
src = os.path.join(G, "prof_" + tag)
bench = None
if os.path.exists(os.path.join(src, "kernel_stats.csv")):
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(P, tag + "_kernel_stats.csv"))
    line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
    open(os.path.join(P, tag + "_bench.json"), "w").write(line + "\n")
    bench = json.loads(line)

# counter passes
pmc = {}
for cfg, trav in (("1", 1), ("1", 0), ("4s", 1)):
    c = {}
    for part in "abc":
        c.update(rows("pmc_%s_%s_%d_%s.txt" % (tag, cfg, trav, part)))
    if c:
        pmc[(cfg, trav)] = (c, derived(c))

# traffic entries (newest sources only), PMC summary attached
entries = []
for f in sorted(glob.glob(os.path.join(G, "traffic_*.json"))):
    entries += json.load(open(f))["entries"]
entries = [e for e in entries if "sources_sha256" in e]
if entries:
    newest = max(entries, key=lambda e: os.path.getmtime(os.path.join(G, "traffic_%s_%d.json" % (e["config"], e["traversal"]))))["sources_sha256"]
    entries = [e for e in entries if e["sources_sha256"] == newest]
    for e in entries:
        key = (str(e["config"]), e["traversal"])
        if key in pmc and "rt_pairkernel" in e["kernel"]:
            e["pmc"] = pmc[key][1]
    json.dump({"entries": entries}, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)

# the counter passes as text
names = {1: "near-first", 0: "reference order"}
L = ["rocprofv3 --pmc passes of rt_pairkernel_strict (two samples of a pixel per lane), per LAUNCH (the largest of the run: 32 frames of",
     "configs[1], 16 of the 1M-triangle scene, rendered in pieces of two frames), one pass per counter group (scripts/pmc_cfg.sh), MI355X.", ""]
for (cfg, trav), (c, d) in pmc.items():
    L.append("config %s, %s" % (cfg, names[trav]))
    for k in sorted(c):
        L.append("  %-30s %.5e" % (k, c[k]))
    tr = [e for e in entries if str(e["config"]) == cfg and e["traversal"] == trav and "rt_pairkernel" in e["kernel"]]
    if tr:
        t = tr[0]
        L.append("  FETCH_SIZE x 2 %.3f GB, WRITE_SIZE %.3f GB, with the pieces' sum kernel %.3f GB per launch" % (
            2 * t["fetch_size_kb"] * 1024 / 1e9, t["write_size_kb"] * 1024 / 1e9, t["bytes_per_launch"] / 1e9))
    L.append("  derived: " + ", ".join("%s %.3g" % (k, v) for k, v in d.items() if not k.startswith("sq_")))
    L.append("")
if pmc:
    open(os.path.join(P, tag + "_pmc.txt"), "w").write("\n".join(L) + "\n")

# other pipelines
for d in sorted(glob.glob(os.path.join(G, "wf_%s_*" % tag))):
    name = os.path.basename(d)[len("wf_%s_" % tag):]
    if os.path.exists(os.path.join(d, "kernel_stats.csv")):
        shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(P, "%s_wf_%s_kernel_stats.csv" % (tag, name)))
        said = [l for l in open(os.path.join(d, "run.log")).read().splitlines() if l.startswith("config ")]
        open(os.path.join(P, "%s_wf_%s_run.txt" % (tag, name)), "w").write((said[-1] if said else "") + "\n")

# bench lines
for f in sorted(glob.glob(os.path.join(G, "bench_%s_cfg*.json" % tag))):
    line = open(f).read().strip().splitlines()[-1]
    open(os.path.join(P, os.path.basename(f).replace("bench_%s_" % tag, "%s_bench_" % tag)), "w").write(line + "\n")

print("\n".join(L))
print("traffic entries:", [(e["kernel"], e["config"], e["traversal"], round(e["bytes_per_launch"] / 1e9, 2)) for e in entries])
if bench:
    print("bench:", bench["value"], bench["ms_per_frame"], bench["roofline"]["frac"], bench["roofline"]["traffic"], bench["other_flavour"]["value"])
