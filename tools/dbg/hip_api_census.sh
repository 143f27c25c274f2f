#!/bin/bash
# HIP runtime calls per training step (rocprofv3 --hip-runtime-trace --stats; counts over a 30-step bench run / 30):
# event records and stream waits are marker / barrier packets in the queues, memsets and copies are extra launches
export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -f gpurun_out/prof/api_*
rocprofv3 --hip-runtime-trace --stats -d gpurun_out/prof -o api -- python bench.py $CENSUS_ARGS --steps 30 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/api.log 2>&1
python - <<EOF
import sqlite3, glob
db = glob.glob("gpurun_out/prof/api_results.db")[0]
c = sqlite3.connect(db).cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if "region" in x.lower() or "api" in x.lower()]
print(t[:10])
for name in ("regions", "rocpd_region"):
    if name in tabs or any(name in x for x in tabs):
        break
try:
    rows = list(c.execute("select name, count(*) from regions group by name order by count(*) desc"))
except Exception as e:
    rows = []
    print("query failed", e)
for n, k in rows[:25]:
    print("%-40s %8.1f per step" % (n, k / 40.0))
# event records / stream waits by stream argument (which queue gets the marker / barrier packet)
try:
    cols = [r[1] for r in c.execute("pragma table_info(region_args)")]
    print(cols)
    q = "select r.name, a.value, count(*) from regions r join region_args a on a.id = r.id where r.name in ('hipEventRecord', 'hipStreamWaitEvent') and a.name = 'stream' group by r.name, a.value order by count(*) desc"
    for n, v, k in list(c.execute(q))[:16]:
        print("%-22s stream %-20s %7.1f per step" % (n, v, k / 40.0))
except Exception as e:
    print("by-stream query failed:", e)
EOF
[ -n "$CENSUS_EXTRA" ] && python $CENSUS_EXTRA
rm -rf gpurun_out/prof
