#!/bin/bash
# A/B of ops.UnpoolSkipFn (skip halves of the unpoolings ahead of the deepest stage, on the weight-gradient stream): LOTUS_SKIP_AHEAD=1|0
mkdir -p gpurun_out
F="--steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-other-modes --no-fresh-batches --no-side-workloads"
one() { python bench.py $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_ms_per_step', {}).get('forward'), d.get('host_ms_per_step', {}).get('backward'))"; }
{
python -m pytest tests/test_gpu_model.py -x -q -k "skip_halves or stream_modes" --timeout 900 2>&1 | tail -3
for rep in 1 2 3 4; do
for a in 0 1; do echo -n "policy ahead=$a: "; LOTUS_SKIP_AHEAD=$a one; done
done
for rep in 1 2; do
for a in 0 1; do echo -n "peract ahead=$a: "; LOTUS_SKIP_AHEAD=$a one --workload peract; done
for a in 0 1; do echo -n "peract64 ahead=$a: "; LOTUS_SKIP_AHEAD=$a one --workload peract --batch 64; done
for a in 0 1; do echo -n "mp ahead=$a: "; LOTUS_SKIP_AHEAD=$a one --workload mp; done
done
} > gpurun_out/ab_skip_ahead.txt 2>&1
cat gpurun_out/ab_skip_ahead.txt
