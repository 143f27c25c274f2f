#!/bin/bash
# PerAct workload (BASELINE configs[4]) with bf16 and fp32 activation storage at several batch sizes: bash tools/peract_ab.sh "16 64"
Q="--no-cpu-baseline --no-roofline --no-fresh-batches --no-other-modes"
for b in ${1:-16 64}; do
  for s in bf16 fp32; do
    r=$(python bench.py --workload peract --act-storage $s --batch $b --steps 16 --warmup 6 $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "peract storage=$s batch=$b: $r"
  done
done
