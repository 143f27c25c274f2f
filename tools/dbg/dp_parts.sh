#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=gpurun_out/dp_parts.txt; : > $OUT
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-side-workloads --no-fresh-batches"
val() { python -c "import json,sys; L=[l for l in sys.stdin.readlines() if l.startswith('{')]; d=json.loads(L[-1]) if L else {}; h=d.get('host_ms_per_step', {}); print(d.get('value'), d.get('ms_per_step'), [h.get(k) for k in ('forward','backward','finish','of_forward_waiting_for_the_prefetched_front_end')])"; }
run() { echo "== $1" >> $OUT; shift; env "$@" $B 2>gpurun_out/err_parts.log | val >> $OUT; }
for rep in 1 2 3; do
run "plain" X=1
run "rehearsal" LOTUS_FORCE_COLLECTIVES=1
run "rehearsal, no pack copies" LOTUS_FORCE_COLLECTIVES=1 LOTUS_DIAG_NO_PACK=1
run "rehearsal, no messages" LOTUS_FORCE_COLLECTIVES=1 LOTUS_DIAG_NO_MESSAGES=1
run "rehearsal, no pack, no messages" LOTUS_FORCE_COLLECTIVES=1 LOTUS_DIAG_NO_MESSAGES=1 LOTUS_DIAG_NO_PACK=1
run "rehearsal, no syncbn" LOTUS_FORCE_COLLECTIVES=1 LOTUS_BENCH_NO_SYNCBN=1
done
cat $OUT
