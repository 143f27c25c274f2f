B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
run() { v=$(env $1 timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>&1 | tail -1); echo "$v  $1"; }
for i in 1 2; do for cfg in "X=base" "LOTUS_SIDE_CUMASK=FFFFFFFE" "LOTUS_SIDE_CUMASK=FEFEFEFE" "LOTUS_SIDE_CUMASK=EEEEEEEE" "LOTUS_SIDE_CUMASK=FFFFFFFF" "LOTUS_SIDE_CUMASK=FFFFFF00"; do run "$cfg"; done; done
