"""One steady-state training step as a kernel SEQUENCE (rocprofv3 --kernel-trace, rocpd sqlite .db): every launch of the
step on every queue with its start offset, duration and grid, plus per-phase sums of the critical queue — where the step
time goes in order, not only by kernel name.

    python tools/step_sequence.py gpurun_out/prof/<name>_results.db [step_index] > gpurun_out/step_sequence.txt
"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("lotus_f32::", "")
    n = re.sub(r"\(.*$", "", n)
    n = n.replace(", float, float, float", "")
    return n[:70]


def main(path, which):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
    gy = gx.replace("_x", "_y") if gx != "0" else "0"
    gz = gx.replace("_x", "_z") if gx != "0" else "0"
    rows = list(cur.execute(f"select name, start, end, queue_id, {gx}, {gy}, {gz}, {wx} from kernels order by start"))
    marks = [r[1] for r in rows if "conv_smallcin_kernel" in r[0]]  # the stem convolution: once per forward pass
    if len(marks) < which + 2:
        which = len(marks) - 2
    t0, t1 = marks[which], marks[which + 1]
    step = [r for r in rows if t0 <= r[1] < t1]
    qbusy = {}
    for r in step:
        qbusy[r[3]] = qbusy.get(r[3], 0) + (r[2] - r[1])
    crit = max(qbusy, key=qbusy.get)
    print(f"step {which}: {(t1 - t0) / 1e6:.3f} ms between two stem convolutions; {len(step)} launches; busy per queue (ms): "
          + ", ".join(f"q{q}: {b / 1e6:.2f}" for q, b in sorted(qbusy.items())) + f"; critical queue q{crit}")
    # phases of the critical queue: forward until the loss kernel, backward after
    loss_t = next((r[1] for r in step if "small_loss_kernel" in r[0]), t1)
    for nm, a, b in (("forward", t0, loss_t), ("backward", loss_t, t1)):
        ks = [r for r in step if r[3] == crit and a <= r[1] < b]
        busy = sum(r[2] - r[1] for r in ks)
        print(f"{nm}: {(b - a) / 1e6:.3f} ms wall, critical queue busy {busy / 1e6:.3f} ms in {len(ks)} launches")
        other = [r for r in step if r[3] != crit and a <= r[1] < b]
        print(f"   other queues: {sum(r[2] - r[1] for r in other) / 1e6:.3f} ms in {len(other)} launches")
    print("\nq  start_us   dur_us  gap_us  grid(blocks)  kernel")
    last_end = {}
    for r in step:
        g = (r[4] // max(r[7], 1)) * max(r[5], 1) * max(r[6], 1) if r[4] else 0
        gap = (r[1] - last_end[r[3]]) / 1e3 if r[3] in last_end else 0.0
        last_end[r[3]] = r[2]
        print(f"{r[3]} {(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} {gap:7.1f} {g:8d}  {short(r[0])}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
