#!/bin/bash
# diagnostics: sweep the split-K settings of the weight-gradient GEMMs (st_set_tuning 200+cap, 1000+rows-per-slice) on the GPU box
R=${GRAFT_REPO_ROOT:-.}
for t in ${SWEEP:-"210" "211" "212" "213" "214" "212 1150" "212 1250" "212 1300" "212 1400"}; do
python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-graph --tune $t 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);k=d['kernels'];print('tune $t', round(d['ms_per_step'],4), 'an', round(k['analysis_wgrad']['avg_us']+k['analysis_wgrad_reduce']['avg_us'],1), 'syn', round(k['synthesis_wgrad']['avg_us']+k['synthesis_wgrad_reduce']['avg_us'],1), {n:round(v['avg_us'],1) for n,v in k.items() if 'wgrad' in n})"
done
