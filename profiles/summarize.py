"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a per-kernel table.

    python profiles/summarize.py gpurun_out/prof/<name>_results.db <steps_in_run> > profiles/<round>_kernels.md
"""
import sqlite3
import sys


def main(path, nsteps):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                            "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    nlaunch = sum(r[1] for r in rows)
    print(f"total kernel time {tot / 1e3:.2f} ms over {nsteps} steps = {tot / nsteps / 1e3:.3f} ms/step, "
          f"{nlaunch / nsteps:.0f} kernel launches/step (all queues, incl. memset / copy kernels)\n")
    try:  # per queue: the busiest one is the critical (forward / input-gradient) stream
        q = list(cur.execute("select queue_id, count(*), sum(end-start)/1e3 from kernels group by queue_id order by 3 desc"))
        print("per queue: " + "; ".join(f"queue {a}: {c / 1e3 / nsteps:.2f} ms/step in {b / nsteps:.0f} launches/step" for a, b, c in q) + "\n")
    except sqlite3.Error:
        pass
    print("| ms/step | % | launches/step | avg us | min us | max us | kernel |")
    print("|---:|---:|---:|---:|---:|---:|---|")
    for r in rows:
        if r[2] / tot < 0.0005:
            continue
        print(f"| {r[2] / nsteps / 1e3:.3f} | {100 * r[2] / tot:.1f} | {r[1] / nsteps:.1f} | {r[3]:.1f} | {r[4]:.1f} | "
              f"{r[5]:.1f} | `{r[0][:140]}` |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
