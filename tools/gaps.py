"""Diagnostic: gaps between consecutive kernels of the busiest queue in a rocprofv3 kernel trace (results.db)."""
import collections, sqlite3, sys
import numpy as np
db, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select queue_id,start,end,name from kernels order by start").fetchall()
byq = collections.defaultdict(list)
for q, s, e, n in rows:
    byq[q].append((s, e, n))
q = max(byq, key=lambda k: sum(e - s for s, e, _ in byq[k]))
l = byq[q]
l = l[len(l) // 3:]
g = np.array([l[i + 1][0] - l[i][1] for i in range(len(l) - 1)]) / 1e3
print("launches per queue", {k: len(v) for k, v in byq.items()}, "critical-queue launches/step %.0f" % (len(l) / steps))
for lo, hi in [(-1e9, 0.5), (0.5, 2), (2, 5), (5, 10), (10, 20), (20, 100)]:
    m = (g >= lo) & (g < hi)
    print("   gap %g-%g us: %.0f/step, %.3f ms/step" % (lo, hi, m.sum() / steps, g[m].sum() / steps / 1e3))
c, k = collections.Counter(), collections.Counter()
for i in range(len(l) - 1):
    gg = (l[i + 1][0] - l[i][1]) / 1e3
    if 2 <= gg < 100:
        c[(l[i][2][:34], l[i + 1][2][:34])] += gg
        k[(l[i][2][:34], l[i + 1][2][:34])] += 1
for key, v in c.most_common(14):
    print("      %.3f ms/step in %.1f gaps/step" % (v / steps / 1e3, k[key] / steps), key)
