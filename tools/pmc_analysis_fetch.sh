cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for dt in f32 bf16_all; do
  CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32x3 --no-graph --dtype $dt"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/an_$dt/trace -o st -- $CMD > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/an_$dt/pmc3 -o st -- $CMD > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
root="$R/gpurun_out/an_$dt"
for f in glob.glob(root+"/trace/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "PolarStore" in r["Name"] or "AnalysisW" in r["Name"]: print("$dt", r["Name"][:60], "avg_ns", r["AverageNs"])
for f in glob.glob(root+"/pmc3/**/*counter_collection.csv",recursive=True):
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and ("PolarStore" in r["Kernel_Name"] or "AnalysisW" in r["Kernel_Name"]): acc[r["Kernel_Name"][:60]]+=float(r["Counter_Value"]); cnt[r["Kernel_Name"][:60]]+=1
    for k in acc: print("$dt", k, "FETCH_KB", acc[k]/cnt[k])
PY
  rm -rf $R/gpurun_out/an_$dt
done
