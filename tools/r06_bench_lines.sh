#!/bin/bash
# GPU box: the bench.py JSON lines that carry roofline.traffic / step_traffic / mfma_busy, run AFTER profiles/r06_pmc_traffic_*.json of the same sources are in
# place (those fields are quoted only from a source-hash-matched file).  Output: gpurun_out/r06/bench_*.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r06; mkdir -p $OUT; cd $REPO
b() { timeout 600 python bench.py $2 > $OUT/bench_$1.json 2> $OUT/bench_$1.err; python - "$OUT/bench_$1.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d.get("roofline", {})
print(sys.argv[1].split("bench_")[-1], d["ms_per_step"], "graph", d.get("ms_per_step_graph"), "roofline", r.get("kernel"), r.get("frac"), r.get("frac_rocprof"), "traffic", r.get("traffic"),
      "step_traffic", r.get("step_traffic"), "ratio", r.get("step_traffic_ratio"), "mfma_busy", r.get("mfma_busy"))
PY
}
b f32 ""
b bf16_all "--dtype bf16_all --no-cpu-baseline"
b f16_all "--dtype f16_all --no-cpu-baseline"
b bf16_all_b1024 "--dtype bf16_all --batch 1024 --no-cpu-baseline"
b scale8_b64_f32 "--scale 8 --batch 64 --no-cpu-baseline"
b scale8_b64_f16_all "--scale 8 --batch 64 --dtype f16_all --no-cpu-baseline"
