export TMPDIR=/tmp; mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace -d gpurun_out/prof -o seq -- python bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/seq.log 2>&1
DB=$(ls gpurun_out/prof/*seq*results.db gpurun_out/prof/*/*seq*results.db 2>/dev/null | head -1)
echo "db: $DB"; tail -2 gpurun_out/prof/seq.log | cut -c1-200
python tools/step_sequence.py $DB 12 > gpurun_out/step_sequence.txt 2>&1
rm -rf gpurun_out/prof
head -8 gpurun_out/step_sequence.txt
