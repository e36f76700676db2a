#!/bin/bash
# GPU box: the bench line's step / autoencoder kernel times for every variant build under variants/ (ST_LIB_PATH), plus the in-tree library.
# Usage: tools/var_bench.sh [extra bench args]   -> gpurun_out/var/<name>.json + one summary line each
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/var; mkdir -p "$OUT"
for lib in "" $REPO/variants/lib_*.so; do
    n=$(basename "${lib:-intree}" .so)
    ST_LIB_PATH=$lib python "$REPO/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --no-f32x3 --no-graph "$@" > "$OUT/$n.json" 2> "$OUT/$n.err"
    python - "$OUT/$n.json" "$n" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d.get("kernels", {})
    print(f"{sys.argv[2]:14s} step {d['ms_per_step']*1e3:7.1f} us  " + "  ".join(f"{n} {k[n]['avg_us']:.1f}" for n in k if n.startswith("ae_") or n in ("post_ae", "synthesis_frames", "analysis_fwd")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done
