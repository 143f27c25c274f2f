#!/bin/bash
# A/B of environment knobs on the PerAct workload, alternating on one box: bash tools/peract_env_ab.sh "<env A>" "<env B>" ...; BATCH=64 selects the batch
Q="--workload peract --batch ${BATCH:-16} --steps ${STEPS:-20} --warmup 8 --no-cpu-baseline --no-roofline --no-fresh-batches --no-other-modes"
for round in 1 2 3; do
  for v in "$@"; do
    r=$(env $v python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$v] $r"
  done
done
