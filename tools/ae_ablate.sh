#!/bin/bash
# Timing-only ablation of ae_bwd_kernel: builds variant libraries with -DST_AE_ABLATE=<bits> (here, hipcc cross-compiles) and,
# on the GPU box, times the kernel with each (bench.py's per-kernel HIP-event leg).  Results of ablated builds are INVALID.
#   tools/ae_ablate.sh build   (in the build container)      tools/ae_ablate.sh run   (on the GPU box, via gpurun)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
VARIANTS="${VARIANTS:-0 1 2 4 8 16 32 64 65 127}"
if [ "$1" = "build" ]; then
  mkdir -p $REPO/gpurun_in
  for v in $VARIANTS; do
    ( make -s -C $REPO/signaltrain_amd/csrc OUT=$REPO/gpurun_in/libst_ablate_$v.so EXTRA=-DST_AE_ABLATE=$v 2>&1 | grep -E "error" ) &
  done
  wait; ls -la $REPO/gpurun_in/
else
  mkdir -p $REPO/gpurun_out/ablate
  for v in $VARIANTS; do
    ST_LIB_PATH=$REPO/gpurun_in/libst_ablate_$v.so python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph $BENCH_EXTRA 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernels']; print('ablate %4d  %s   step %.3f ms' % ($v, '  '.join('%s %.1f us' % (n, k[n]['avg_us']) for n in k if n.startswith('ae_')), d['ms_per_step']))" | tee -a $REPO/gpurun_out/ablate/ae_ablate.txt
  done
fi
