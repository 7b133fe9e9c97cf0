# PMC passes over the PointPillar encoder (tests/tools/encoder_bench.py): what bounds the split-precision convolutions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02/conv_pmc
mkdir -p $OUT
B="python3 tests/tools/encoder_bench.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD -f csv -d $OUT/sq1 -o p -- $B > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS -f csv -d $OUT/sq2 -o p -- $B > $OUT/sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("sq1", "sq2"):
    f = glob.glob("$OUT/%s/*counter_collection.csv" % d)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in agg.items():
        if "k_conv" in k: print(d, k, {a: "%.3g" % b for a, b in v.items()})
PY
