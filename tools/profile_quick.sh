#!/bin/bash
# quick PMC passes (no full trace stats) -> gpurun_out/prof_<tag>/summary.txt
TAG=${1:-q}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32x3 --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o st -- $CMD > $OUT/trace_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o st -- $CMD > $OUT/pmc1_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -o st -- $CMD > $OUT/pmc2_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc5 -o st -- $CMD > $OUT/pmc5_stdout.txt 2>&1
python $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -i "ae_bwd\|ae_fwd" $OUT/summary.txt
