export TMPDIR=/tmp; mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace -d gpurun_out/prof -o seqp -- python bench.py --workload peract --batch 64 --steps 12 --warmup 5 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/seqp.log 2>&1
DB=$(ls gpurun_out/prof/*seqp*results.db gpurun_out/prof/*/*seqp*results.db 2>/dev/null | head -1)
tail -1 gpurun_out/prof/seqp.log | cut -c1-200
python tools/step_sequence.py $DB 8 > gpurun_out/step_sequence_peract64.txt 2>&1
python profiles/summarize.py $DB 17 > gpurun_out/peract64_kernels.md 2>&1
rm -rf gpurun_out/prof
head -8 gpurun_out/step_sequence_peract64.txt; head -30 gpurun_out/peract64_kernels.md | cut -c1-210
