#!/bin/bash
# Round evidence in ONE gpurun call: full GPU test suite (-> parity ledger), counter passes, kernel traces, bench lines; the
# summaries are assembled on the box and copied to gpurun_out/profiles_<tag>/ (the raw traces are too large to travel back).
TAG=${1:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_ledger.json
python -m pytest tests -m gpu -q --timeout 1200 2>&1 | tail -8 > gpurun_out/gpu_tests_$TAG.log
bash profiles/collect.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}pz -- python bench.py --workload peract --steps 20 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches > gpurun_out/prof/${TAG}pz.log 2>&1
python profiles/assemble.py $TAG > gpurun_out/assemble_$TAG.log 2>&1
python profiles/pmc_report.py ${TAG}p gpurun_out/pmc_${TAG}p_fetch gpurun_out/pmc_${TAG}p_write gpurun_out/pmc_${TAG}p_sq gpurun_out/pmc_${TAG}p_so.sha >> gpurun_out/assemble_$TAG.log 2>&1
python profiles/summarize.py gpurun_out/prof/${TAG}pz_results.db 30 > profiles/${TAG}p_kernels.md 2>> gpurun_out/assemble_$TAG.log
python tools/host_floor.py > gpurun_out/host_floor_$TAG.txt 2>&1
LOTUS_PAIR=0 python tools/host_floor.py > gpurun_out/host_floor_${TAG}_nopair.txt 2>&1
# (round 5's kernel-lab tables, the vector-ALU census and the MFMA / VALU co-issue micro-benchmark are unchanged by round 6:
#  profiles/r05_gemm_lab*.txt, r05_valu_census.md, r05_mfma_coissue.txt; EVIDENCE_LAB=1 regenerates them)
if [ -n "${EVIDENCE_LAB:-}" ]; then
python tools/valu_census.py run > gpurun_out/census_$TAG.log 2>&1 && python tools/valu_census.py report gpurun_out/census > profiles/${TAG}_valu_census.md 2>> gpurun_out/census_$TAG.log
rm -f gpurun_out/census/*.csv
[ -x tools/ubench/mfma_coissue ] && tools/ubench/mfma_coissue > profiles/${TAG}_mfma_coissue.txt 2>&1
if [ -x tools/lab/gemm_lab ]; then
  for k in fwd dgrad wgrad; do timeout 600 tools/lab/gemm_lab tools/gemm_shapes.json $k 16000 > gpurun_out/lab_$k.txt 2>&1; done
  cat gpurun_out/lab_fwd.txt gpurun_out/lab_dgrad.txt gpurun_out/lab_wgrad.txt | cut -c1-400 > profiles/${TAG}_gemm_lab.txt
  timeout 300 tools/lab/gemm_lab tools/gemm_shapes.json abl | cut -c1-400 > profiles/${TAG}_gemm_lab_ablations.txt 2>&1
fi
python tools/dbg/dma_onoff.py > profiles/${TAG}_dma_onoff.txt 2>&1
python tools/dbg/ln_fused_ab.py > profiles/${TAG}_ln_fused_ab.txt 2>&1
fi
# round 6: the data-parallel step on one GPU — what a statistics message costs the stream it is issued on per way of issuing
# it, the one-rank RCCL rehearsal in 20 fresh processes (VERDICT r5 item 1 acceptance), what its parts cost, its per-queue sequence
python tools/dbg/msg_cost.py big 2>/dev/null | grep -E "per message|per iteration|32 MB" > profiles/${TAG}_msg_cost.txt
REPS=20 bash tools/dbg/dp_rehearsal_dist.sh > /dev/null 2>&1; cp gpurun_out/dp_rehearsal_dist.txt profiles/${TAG}_dp_rehearsal_dist.txt
bash tools/dbg/dp_parts.sh > /dev/null 2>&1; cp gpurun_out/dp_parts.txt profiles/${TAG}_dp_parts.txt
LOTUS_FORCE_COLLECTIVES=1 bash tools/dbg/timeline_cfg.sh rehearsal > /dev/null 2>&1; head -60 gpurun_out/seq_rehearsal.txt | cut -c1-160 > profiles/${TAG}_rehearsal_sequence.txt
LOTUS_FORCE_COLLECTIVES=1 CENSUS_ARGS="" bash tools/dbg/hip_api_census.sh 2>&1 | cut -c1-160 > profiles/${TAG}_rehearsal_hip_api_census.txt
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-other-modes --no-fresh-batches --no-side-workloads --gemm-report profiles/${TAG}_gemm_report.txt > /dev/null 2>&1
LOTUS_GEMM_DMA=0 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-other-modes --no-fresh-batches --no-side-workloads --gemm-report profiles/${TAG}_gemm_report_dma_off.txt > /dev/null 2>&1
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}* gpurun_out/profiles_$TAG/ 2>/dev/null
cp gpurun_out/parity_ledger.json gpurun_out/profiles_$TAG/${TAG}_parity.json 2>/dev/null
rm -rf gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}p_fetch gpurun_out/pmc_${TAG}p_write gpurun_out/pmc_${TAG}p_sq gpurun_out/prof
cat gpurun_out/gpu_tests_$TAG.log; tail -5 gpurun_out/assemble_$TAG.log; cat gpurun_out/host_floor_$TAG.txt | head -3; du -sh gpurun_out
