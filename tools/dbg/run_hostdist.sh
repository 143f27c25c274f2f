Q="--batch 1 --npoints 256 --steps 40 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads"
for cfg in "X=0" "LOTUS_FORCE_COLLECTIVES=1"; do env $cfg python bench.py $Q 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d.get('comm'))"; done
LOTUS_FORCE_COLLECTIVES=1 python -m cProfile -o /tmp/prof.out bench.py $Q > /dev/null 2>&1
python - <<'PY'
import pstats
p=pstats.Stats('/tmp/prof.out'); p.sort_stats('tottime').print_stats(28)
PY
