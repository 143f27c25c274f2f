#!/bin/bash
# SQ counters of ONE product (two passes; FETCH_SIZE / TCC_*_sum together in one pass abort rocprofv3 on this image and hang the call): bash tools/dbg/pmc_one.sh <tag> wgrad 65536 128 128 bf16
TAG=$1; shift
export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc1_${TAG}_$n -o c -- python tools/dbg/one_gemm.py "$@" > gpurun_out/pmc1_${TAG}_$n.log 2>&1
done
python - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
for d in sorted(glob.glob(f"gpurun_out/pmc1_{tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "gemm" in k or "reduce" in k:
                print(k, {c: round(x / cnt[(k, c)], 1) for c, x in v.items()}, "launches", max(cnt[(k, c)] for c in v))
PY
