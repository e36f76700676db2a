#!/bin/bash
# GPU box: the round-6 evidence set -- rocprofv3 kernel stats + PMC passes (tools/profile_gpu.sh) for the headline configurations, one bench.py JSON line per
# configuration of the matrix, the randomized parity sweep.  Only summaries are kept (the raw traces exceed gpurun's 64 MiB return).  The six lines that carry
# roofline.traffic / step_traffic / mfma_busy are run by tools/r06_bench_lines.sh once profiles/r06_pmc_traffic_*.json of the SAME sources are in place.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r06; mkdir -p $OUT
prof() {  # tag, bench args
    BENCH_EXTRA="$2" bash $REPO/tools/profile_gpu.sh $1 > $OUT/prof_$1.log 2>&1
    local P=$REPO/gpurun_out/prof_$1
    cp $P/summary.txt $OUT/rocprofv3_summary_$1.txt; cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$1.csv; cp $P/pmc_traffic.json $OUT/pmc_traffic_$1.json
    rm -rf $P
}
prof f32 ""
prof bf16_all "--dtype bf16_all"
prof f16_all "--dtype f16_all"
prof bf16_all_b1024 "--dtype bf16_all --batch 1024"
prof scale8_b64_f32 "--scale 8 --batch 64"
prof scale8_b64_f16_all "--scale 8 --batch 64 --dtype f16_all"
[ -n "$ONLY_PROF" ] && { ls $OUT | wc -l; exit 0; }      # ONLY_PROF=1: the profiles / PMC passes alone (after a source change that does not move any number, e.g. a header comment)
cd $REPO
b() { timeout 600 python bench.py $2 > $OUT/bench_$1.json 2> $OUT/bench_$1.err; tail -c 600 $OUT/bench_$1.json | head -c 300; echo; }
b bf16 "--dtype bf16 --no-cpu-baseline"
b f32x3 "--dtype f32x3 --no-cpu-baseline"
b f32_b512 "--batch 512 --no-cpu-baseline"
b f32_b1024 "--batch 1024 --no-cpu-baseline"
b bf16_all_b512 "--dtype bf16_all --batch 512 --no-cpu-baseline"
b scale8_b64_bf16_all "--scale 8 --batch 64 --dtype bf16_all --no-cpu-baseline"
b scale8_b63_f16_all "--scale 8 --batch 63 --dtype f16_all --no-cpu-baseline"
b legacy8_b64_f16_all "--scale 8 --scheme legacy --batch 64 --dtype f16_all --no-cpu-baseline"
b forcedp_lib "--force-dp --no-cpu-baseline"
b forcedp_lib_staged "--force-dp --dp-schedule staged --no-cpu-baseline"
b forcedp_lib_bf16_all "--force-dp --dtype bf16_all --no-cpu-baseline"
b forcedp_lib_bf16_all_staged_pack16 "--force-dp --dtype bf16_all --dp-schedule staged --dp-pack16 --no-cpu-baseline"
for dt in f32 bf16_all; do python tools/train_throughput.py $dt 2>&1 | grep "train windows\|==>" >> $OUT/train_loop_throughput.txt; done
timeout 560 python tools/fuzz_parity.py 500 2>&1 | grep -v "RuntimeWarning\|_prep(" > $OUT/fuzz_parity.txt
tail -2 $OUT/fuzz_parity.txt
ls $OUT | wc -l
