B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
run() { v=$(env $1 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null); echo "$v  $1"; }
for i in 1 2 3; do for cfg in "X=base" "LOTUS_CPE_WG_LATE=1" "LOTUS_CPE_WG_LATE=1 LOTUS_SIDE_LOWPRIO=1"; do run "$cfg"; done; done
