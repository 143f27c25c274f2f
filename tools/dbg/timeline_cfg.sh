#!/bin/bash
# kernel-trace sequence of one steady-state step for the configuration given in the environment: gpurun_out/seq_$1.txt
export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -f gpurun_out/prof/rc_*
rocprofv3 --kernel-trace -d gpurun_out/prof -o rc -- python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/rc.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/prof/rc.log | head -1
python tools/step_sequence.py gpurun_out/prof/rc_results.db > gpurun_out/seq_$1.txt 2>&1
head -6 gpurun_out/seq_$1.txt
rm -rf gpurun_out/prof
