#!/bin/bash
# GPU box: one train step as a kernel timeline (rocprofv3 --kernel-trace), plain fused step and the exchange step with one rank -> gpurun_out/tl/*.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/tl; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # tag, extra bench args
    local T=/tmp/st_tl_$1
    rocprofv3 --kernel-trace --output-format csv -d "$T" -o st -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32x3 --no-graph $2 > "$OUT/$1.out" 2>&1
    python "$REPO/tools/step_timeline.py" "$T" > "$OUT/$1.txt" 2>&1
}
run plain ""
run forcedp "--force-dp"
run bf16_all "--dtype bf16_all"
run forcedp_bf16_all "--force-dp --dtype bf16_all"
cat "$OUT/plain.txt" "$OUT/forcedp.txt" "$OUT/bf16_all.txt" "$OUT/forcedp_bf16_all.txt"
