#!/bin/bash
# in-step sequence of the one-rank RCCL rehearsal (LOTUS_FORCE_COLLECTIVES=1), default hardware queues and GPU_MAX_HW_QUEUES=8:
# gpurun_out/rccl_sequence_q{4,8}.txt (tools/step_sequence.py), rccl_kernels_q{4,8}.md
export TMPDIR=/tmp; mkdir -p gpurun_out/prof
for q in 4 8; do
  rm -f gpurun_out/prof/rc_*
  GPU_MAX_HW_QUEUES=$q LOTUS_FORCE_COLLECTIVES=1 rocprofv3 --kernel-trace -d gpurun_out/prof -o rc -- python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/rc.log 2>&1
  python profiles/summarize.py gpurun_out/prof/rc_results.db 12 > gpurun_out/rccl_kernels_q$q.md
  python tools/step_sequence.py gpurun_out/prof/rc_results.db > gpurun_out/rccl_sequence_q$q.txt 2>&1
  head -8 gpurun_out/rccl_sequence_q$q.txt
done
rm -rf gpurun_out/prof
