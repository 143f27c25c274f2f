python -m pytest tests/test_gpu_blocks.py tests/test_gpu_model.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-fresh-batches --no-other-modes"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for f in 0 1 2; do $B 2>/dev/null | python -c "$P"; done
