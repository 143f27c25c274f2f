import sqlite3, glob
db = glob.glob("gpurun_out/prof/api_results.db")[0]
c = sqlite3.connect(db).cursor()
cols = [r[1] for r in c.execute("pragma table_info(regions)")]
print(cols)
rows = list(c.execute("select id, name, start, end, tid from regions order by start"))
idx = [i for i, r in enumerate(rows) if r[1] == "hipExtMallocWithFlags"]
print(len(idx), "mallocs")
for i in idx[len(idx)//2: len(idx)//2 + 3]:
    for r in rows[max(0, i - 6): i + 8]:
        args = list(c.execute("select name, value from region_args where id = ?", (r[0],)))
        print("   ", r[1], "tid", r[4], "dur_us %.1f" % ((r[3] - r[2]) / 1e3), [a for a in args if a[0] in ("size", "sizeBytes", "flags", "stream", "ptr")][:4])
    print("----")
