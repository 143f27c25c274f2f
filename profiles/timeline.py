"""Per-queue busy time, idle gaps and the largest gaps of a rocprofv3 --kernel-trace run (rocpd sqlite .db).

    python profiles/timeline.py gpurun_out/prof/<name>_results.db [skip_fraction [stop_fraction]]

Reports, for the steady-state part of the run (the first `skip_fraction` of the wall time is dropped, default 0.5):
wall time, union-busy time of all queues, and per queue: launches, busy time, idle time split by gap size, and the
kernels either side of the widest gaps — i.e. where the critical stream waits for the host or for a dependency.
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, skip=0.5, stop=1.0):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select queue_id, start, end, name from kernels order by start"))
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[1] >= cut and r[2] <= t0 + (t1 - t0) * stop]
    wall = (max(r[2] for r in rows) - rows[0][1]) / 1e6
    # union busy
    busy, cur_end = 0, 0
    for _, s, e, _ in rows:
        if s > cur_end:
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    print(f"steady-state window {wall:.2f} ms, {len(rows)} launches, any-queue busy {busy / 1e6:.2f} ms "
          f"({100 * busy / 1e6 / wall:.1f} %)")
    byq = defaultdict(list)
    for q, s, e, n in rows:
        byq[q].append((s, e, n))
    for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        b = sum(e - s for s, e, _ in ks) / 1e6
        gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
        pos = [g for g in gaps if g[0] > 0]
        bins = [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)]
        hist = ", ".join(f"{lo}-{hi if hi < 1e9 else 'inf'}us: {sum(1 for g in pos if lo * 1e3 <= g[0] < hi * 1e3)} "
                         f"({sum(g[0] for g in pos if lo * 1e3 <= g[0] < hi * 1e3) / 1e6:.2f} ms)" for lo, hi in bins)
        print(f"\nqueue {q}: {len(ks)} launches, busy {b:.2f} ms, idle between launches {sum(g[0] for g in pos) / 1e6:.2f} ms")
        print("  gaps " + hist)
        agg = defaultdict(lambda: [0, 0])
        for g, a, bn in pos:
            k = (a.split("(")[0][:40], bn.split("(")[0][:40])
            agg[k][0] += g
            agg[k][1] += 1
        for (a, bn), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
            print(f"  {g / 1e6:7.3f} ms in {c:4d} gaps (avg {g / c / 1e3:6.1f} us)  {a}  ->  {bn}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
