#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters per kernel) into a small text table."""
import csv, glob, os, sys, collections
root = sys.argv[1]


def short(n):
    n = n.replace("stg::", "").replace("sta::", "").replace("stm::", "")
    for a, b in (("gemm_kernel<4, FramedNT, AnalysisW, PolarStore>", "gemm:analysis_fwd"),
                 ("gemm_kernel<3, PlainTN, FramedTN, StoreC>", "gemm:wgrad(TN)"),
                 ("gemm_kernel<2, PlainNT, PlainTN, StoreC>", "gemm:synthesis_frames"),
                 ("gemm_kernel<2, FramedNT, PlainNT, StoreC>", "gemm:synthesis_dgrad")):
        if a in n:
            return b
    return n.split("(")[0][:60]


for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:24]:
        print(f"  {short(r['Name']):44s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):>10.0f} total%={r['Percentage']}")
for sub in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5"):
    for f in sorted(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)):
        print("== counters:", os.path.relpath(f, root))
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k in acc:
            print("  " + k[:44].ljust(44) + "  " + "  ".join(f"{c}={v / max(cnt[(k, c)], 1):.4g}" for c, v in sorted(acc[k].items())))

# per-launch memory-side traffic of every kernel (FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch):
# profiles/<tag>_pmc_traffic.json is what bench.py's roofline.traffic quotes.
import json
traffic = collections.defaultdict(dict)
for sub, cname in (("pmc3", "FETCH_SIZE"), ("pmc4", "WRITE_SIZE")):
    for f in sorted(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                k = r["Kernel_Name"].split("(")[0]; acc[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in acc:
            traffic[k][cname + "_KB"] = acc[k] / cnt[k]
# round 5: the matrix-pipe / busy counters of the first PMC pass ride along (per-dispatch averages), so that bench.py can quote MFMA-busy for its roofline kernel
for f in sorted(glob.glob(os.path.join(root, "pmc1", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU"):
            k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in acc:
        for c, v in acc[k].items():
            traffic[k][c] = v / cnt[(k, c)]
# rocprofv3's own average duration per kernel (kernel-trace stats of the same command): bench.py's roofline prefers it over its in-process HIP-event
# timing (which reads 2-3 us high per launch) when the file matches the sources it runs
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        traffic[k]["avg_ns"] = float(r["AverageNs"]); traffic[k]["calls"] = int(r["Calls"])
if traffic:
    # meta: which sources and which bench arguments these counters were measured on (bench.py refuses anything else)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from src_sha import src_sha
    extra = os.environ.get("BENCH_EXTRA", "").split()
    def opt(name, default):
        return extra[extra.index(name) + 1] if name in extra else default
    meta = {"src_sha": src_sha(), "dtype": opt("--dtype", "f32"), "scale": int(opt("--scale", "1")), "batch": int(opt("--batch", "256")),
            "scheme": opt("--scheme", "lean"), "bench_extra": " ".join(extra)}
    json.dump({"meta": meta, "kernels": traffic}, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
