#!/bin/bash
# (Block, CABlock) pair node by level size (LOTUS_PAIR unset = "auto", round 6) against never (LOTUS_PAIR=0): throughput, host time
mkdir -p gpurun_out
F="--steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-other-modes --no-side-workloads --no-fresh-batches"
one() { python bench.py $F "$@" 2>gpurun_out/ab_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get('host_ms_per_step',{}); print(d['value'], 'host', h.get('forward'), h.get('backward'), 'wait', h.get('of_forward_waiting_for_the_prefetched_front_end'))"; }
{
python -m pytest tests/test_gpu_parallel.py tests/test_gpu_round4.py -x -q --timeout 900 2>&1 | tail -2
for rep in 1 2 3; do
  echo -n "rehearsal pair=0: "; LOTUS_PAIR=0 LOTUS_FORCE_COLLECTIVES=1 one
  echo -n "rehearsal pair=auto: "; LOTUS_FORCE_COLLECTIVES=1 one
done
for rep in 1 2; do
  echo -n "policy pair=0: "; LOTUS_PAIR=0 one
  echo -n "policy pair=auto: "; one
done
for v in 0 auto; do if [ $v = 0 ]; then export LOTUS_PAIR=0; else unset LOTUS_PAIR; fi; echo "host floor pair=$v: $(python tools/host_floor.py 2>/dev/null | grep 'host ' | head -1)"; done
} > gpurun_out/ab_pair_levels2.txt 2>&1
cat gpurun_out/ab_pair_levels2.txt
mkdir -p gpurun_out/hl; for i in 1 2 3; do python bench.py --steps 30 --warmup 10 2>gpurun_out/hl/h$i.err | tail -1 > gpurun_out/hl/h$i.json; python -c "
import json; d=json.load(open('gpurun_out/hl/h$i.json')); print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"; done
