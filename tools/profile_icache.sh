#!/bin/bash
TAG=${1:-ic}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*" | sort -u | head -40 > $OUT/counters.txt
cat $OUT/counters.txt
CMD="python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc6 -o st -- $CMD > $OUT/pmc6_stdout.txt 2>&1
tail -3 $OUT/pmc6_stdout.txt
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("$OUT/pmc6/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:50]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in acc:
    if "ae_" in k or "gemm" in k: print(k, {c: round(v/max(cnt[(k,c)],1)) for c,v in acc[k].items()})
PY
