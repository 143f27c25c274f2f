B="python bench.py --no-cpu-baseline --no-other-modes --no-roofline --no-side-workloads --no-fresh-batches"
val() { python -c "import json,sys; L=[l for l in sys.stdin.readlines() if l.startswith('{')]; d=json.loads(L[-1]) if L else {}; h=d.get('host_ms_per_step', {}); print(d.get('value'), d.get('ms_per_step'), [h.get(k) for k in ('forward','backward','finish','of_forward_waiting_for_the_prefetched_front_end')])"; }
for rep in 1 2; do
echo "peract16 early"; $B --workload peract --steps 30 --warmup 10 2>/dev/null | val
echo "peract16 late"; LOTUS_BENCH_PREFETCH_LATE=1 $B --workload peract --steps 30 --warmup 10 2>/dev/null | val
echo "peract64 early"; $B --workload peract --batch 64 --steps 12 --warmup 5 2>/dev/null | val
echo "peract64 late"; LOTUS_BENCH_PREFETCH_LATE=1 $B --workload peract --batch 64 --steps 12 --warmup 5 2>/dev/null | val
echo "mp early"; $B --workload mp --steps 30 --warmup 10 2>/dev/null | val
echo "mp late"; LOTUS_BENCH_PREFETCH_LATE=1 $B --workload mp --steps 30 --warmup 10 2>/dev/null | val
echo "plain pair=1 early"; LOTUS_PAIR=1 $B --steps 30 --warmup 10 2>/dev/null | val
echo "rehearsal pair=1"; LOTUS_PAIR=1 LOTUS_FORCE_COLLECTIVES=1 $B --steps 30 --warmup 10 2>/dev/null | val
done
