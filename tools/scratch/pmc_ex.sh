run() { n=$1; shift; rm -rf /tmp/pm_$n
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pm_$n -o pm -- python3 bench.py --workload flac_pipeline --steps 1 --warmup 0 --prewarm 0 --cpu-streams 0 --extra-windows 0 > /dev/null 2>&1
  python3 - /tmp/pm_$n <<PY
import csv,glob,collections,sys
acc=collections.OrderedDict()
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_flac_extract" in r["Kernel_Name"]:
            acc.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in list(acc.items())[-1:]: print({a:"%.4g"%b for a,b in v.items()})
PY
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run b GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run c SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT
