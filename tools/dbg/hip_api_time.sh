#!/bin/bash
# HIP runtime API time per training step by call name and thread (rocprofv3 --hip-runtime-trace): where the host side of a step goes
export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -f gpurun_out/prof/api_*
rocprofv3 --hip-runtime-trace -d gpurun_out/prof -o api -- python bench.py $CENSUS_ARGS --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/api.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/prof/api.log | head -1
python - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/prof/api_results.db")[0]
c = sqlite3.connect(db).cursor()
rows = list(c.execute("select name, tid, count(*), sum(end - start) / 1e3, max(end - start) / 1e3 from regions group by name, tid order by sum(end - start) desc"))
print("%-38s %8s %10s %12s %10s" % ("call", "tid", "n/step", "us/step", "max us"))
for n, tid, k, tot, mx in rows[:28]:
    print("%-38s %8d %10.1f %12.1f %10.1f" % (n, tid, k / 40.0, tot / 40.0, mx))
PY
rm -rf gpurun_out/prof
