python -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks.py -x -q -k "conv or cpe" 2>&1 | tail -2
export TMPDIR=/tmp; mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r2f -- python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches > gpurun_out/prof/r2f.log 2>&1
python - <<'PY'
import sqlite3
cur=sqlite3.connect('gpurun_out/prof/r2f_results.db').cursor()
for r in cur.execute("select name, count(*)/30.0, sum(end-start)/30.0/1e6 from kernels where name like '%conv_%' group by name"): print(r)
PY
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-fresh-batches --no-other-modes 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
