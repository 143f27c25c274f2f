"""Summary of `gemm_lab shapes.json trace`: phase durations per block and how the resident blocks of a CU overlap."""
import sys, collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l[0].isdigit()]
import statistics as st
def us(c): return c / 2400.0  # shader clock ticks -> us at 2.4 GHz (upper bound of the clock)
ph = {"wait_slab0": [r[3] - r[2] for r in rows], "main_loop": [r[4] - r[3] for r in rows], "store_issue": [r[5] - r[4] for r in rows],
      "store_drain": [r[6] - r[5] for r in rows], "total": [r[6] - r[2] for r in rows]}
for k, v in ph.items():
    v = sorted(v); n = len(v)
    print(f"{k:12s} median {us(v[n//2]):7.2f} us   p10 {us(v[n//10]):7.2f}   p90 {us(v[9*n//10]):7.2f}   (ticks {v[n//2]})")
t0 = min(r[1] for r in rows)
print("kernel span (100 MHz wall clock): %.1f us" % ((max(r[1] for r in rows) - t0) / 100.0))
# per CU: blocks in start order
cu = collections.defaultdict(list)
for r in rows: cu[(r[8], r[7] & 0xfffff0)].append(r)   # xcc, hw id without the wave slot bits
print("distinct (xcc, hwid>>4) groups:", len(cu))
k = sorted(cu)[0]
print("one CU, blocks in start order: start_us(wall)  wait / loop / store / drain (us at 2.4 GHz)")
for r in sorted(cu[k], key=lambda r: r[1]):
    print("  %8.2f   %5.2f %6.2f %5.2f %5.2f" % ((r[1] - t0) / 100.0, us(r[3]-r[2]), us(r[4]-r[3]), us(r[5]-r[4]), us(r[6]-r[5])))
