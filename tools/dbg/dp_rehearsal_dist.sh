#!/bin/bash
# VERDICT r5 item 1 acceptance: the one-rank RCCL rehearsal of the data-parallel step (LOTUS_FORCE_COLLECTIVES=1) in N fresh
# processes at the default GPU_MAX_HW_QUEUES, the plain step interleaved: gpurun_out/dp_rehearsal_dist.txt
export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=gpurun_out/dp_rehearsal_dist.txt; : > $OUT
B="python bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-side-workloads --no-fresh-batches"
val() { python -c "import json,sys; L=[l for l in sys.stdin.readlines() if l.startswith('{')]; d=json.loads(L[-1]) if L else {}; h=d.get('host_ms_per_step', {}); print(d.get('value'), d.get('ms_per_step'), [h.get(k) for k in ('forward','backward','finish','of_forward_waiting_for_the_prefetched_front_end')], d.get('library_sha256_16'))"; }
run() { echo "== $1" >> $OUT; shift; env "$@" $B 2>gpurun_out/err_dist.log | val >> $OUT; }
for rep in $(seq 1 ${REPS:-20}); do
  run "rehearsal" LOTUS_FORCE_COLLECTIVES=1
  if [ $((rep % 4)) -eq 1 ]; then run "plain" X=1; fi
done
python - <<PY
import statistics as st
d={}
L=open("$OUT").read().strip().split("\n")
for a,b in zip(L[::2],L[1::2]):
    if a.startswith("=="): d.setdefault(a[3:],[]).append(float(b.split()[0]))
open("$OUT","a").write("\n")
for k,v in d.items():
    m=st.median(v); line="%s: n %d min %.1f median %.1f max %.1f; within +-1.5%% of the median: %d of %d; values %s" % (k, len(v), min(v), m, max(v), sum(abs(x-m)<=0.015*m for x in v), len(v), sorted(v))
    print(line); open("$OUT","a").write(line+"\n")
if "plain" in d and "rehearsal" in d:
    line="rehearsal median / plain median = %.4f" % (st.median(d["rehearsal"])/st.median(d["plain"])); print(line); open("$OUT","a").write(line+"\n")
PY
