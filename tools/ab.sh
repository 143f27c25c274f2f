export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -f gpurun_out/prof/pa*
B="python bench.py --workload peract --steps 20 --warmup 10 --no-cpu-baseline --no-roofline --no-fresh-batches --no-other-modes"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["dtype"])'
$B 2>/dev/null | python -c "$P"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o pa -- $B > gpurun_out/prof/pa.log 2>&1
python profiles/summarize.py gpurun_out/prof/pa_results.db 30 2>/dev/null | head -45
