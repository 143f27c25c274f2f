#!/bin/bash
# End-of-round evidence run (on the GPU box, from the repo root):   bash profiles/collect.sh r02
# 1. the three counter passes (profiles/pmc_passes.sh)  2. a kernel trace of 30 steps  3. the bench lines quoted in
# profiles/<tag>_bench.json.  Everything lands under gpurun_out/; profiles/assemble.py turns it into the committed files.
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/bench_$TAG
bash profiles/pmc_passes.sh $TAG > gpurun_out/pmc_${TAG}.log 2>&1
BENCH_ARGS="--workload peract" bash profiles/pmc_passes.sh ${TAG}p > gpurun_out/pmc_${TAG}p.log 2>&1
rm -f gpurun_out/prof/${TAG}z_*
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}z -- python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches > gpurun_out/prof/${TAG}z.log 2>&1
O=gpurun_out/bench_$TAG
python bench.py --steps 30 --warmup 10 2>$O/headline.err | tail -1 > $O/headline.json
Q="--no-cpu-baseline --no-roofline --no-fresh-batches --no-side-workloads"
python bench.py --with-optimizer $Q --no-other-modes 2>/dev/null | tail -1 > $O/with_optimizer.json
python bench.py --workload mp $Q --no-other-modes 2>/dev/null | tail -1 > $O/mp.json
python bench.py --workload peract $Q --no-other-modes 2>/dev/null | tail -1 > $O/peract.json
python bench.py --workload peract --act-storage fp32 $Q --no-other-modes 2>/dev/null | tail -1 > $O/peract_fp32_storage.json
for b in 64 128; do
  python bench.py --workload peract --batch $b --steps 20 --warmup 8 $Q --no-other-modes 2>/dev/null | tail -1 > $O/peract_batch_$b.json
  python bench.py --workload peract --act-storage fp32 --batch $b --steps 20 --warmup 8 $Q --no-other-modes 2>/dev/null | tail -1 > $O/peract_fp32_storage_batch_$b.json
done
python bench.py --ragged $Q --no-other-modes 2>/dev/null | tail -1 > $O/ragged.json
python bench.py --batch 38 $Q --no-other-modes 2>/dev/null | tail -1 > $O/batch_38.json
for b in 32 64 128; do python bench.py --batch $b --steps 12 --warmup 4 $Q 2>/dev/null | tail -1 > $O/batch_$b.json; done
LOTUS_FORCE_COLLECTIVES=1 python bench.py $Q --no-other-modes 2>/dev/null | tail -1 > $O/one_rank_rccl.json
LOTUS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 $Q --no-other-modes 2>/dev/null | tail -1 > $O/two_ranks_one_gpu_gloo.json
sha256sum robot-3dlotus_amd/csrc/liblotus_hip.so | cut -c1-16 > $O/so.sha
ls -la $O
