#!/bin/bash
# in-step timeline of the one-rank RCCL rehearsal (LOTUS_FORCE_COLLECTIVES=1): gpurun_out/rccl_timeline.txt, rccl_sequence.txt
export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -f gpurun_out/prof/rc_*
LOTUS_FORCE_COLLECTIVES=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o rc -- python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/rc.log 2>&1
python profiles/timeline.py gpurun_out/prof/rc_results.db > gpurun_out/rccl_timeline.txt
python profiles/summarize.py gpurun_out/prof/rc_results.db 20 > gpurun_out/rccl_kernels.md
python tools/step_sequence.py gpurun_out/prof/rc_results.db > gpurun_out/rccl_sequence.txt 2>&1
rm -rf gpurun_out/prof
head -30 gpurun_out/rccl_timeline.txt
