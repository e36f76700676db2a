cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r2i
CMD="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r2i/pmc1 -o st -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r2i/pmc2 -o st -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/r2i/pmc3 -o st -- $CMD > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for d in ("pmc1","pmc2","pmc3"):
    f=glob.glob("$R/gpurun_out/r2i/%s/**/*counter_collection.csv"%d,recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:48]
        if "ae_" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    first=None
    for k,v in acc.items():
        n=None
        print(k, {c:"%.3g"%(x/6.0) for c,x in v.items()})
PY
