"""HBM-side bytes per GEMM launch grouped by kernel variant and grid (rocprofv3 PMC passes, see pmc_summary.py):

    python profiles/pmc_gemm_groups.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv > profiles/r01_pmc_gemm_groups.md

FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as reported.  Template arguments: <BM, BN, BK, A k-contiguous, B k-contiguous,
SUM_A (weight gradient with bias sums), FAST>: (true,true) = forward, (true,false) = input gradient, (false,false) = weight gradient."""
import collections
import csv
import sys


def agg(path):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if "gemm_kernel" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0].replace("void gemm_kernel", ""), int(r["Grid_Size"]) // 256)
            d[k][0] += float(r["Counter_Value"]) * 1024
            d[k][1] += 1
    return d


def main(fp, wp):
    f, w = agg(fp), agg(wp)
    rows = sorted(((2 * f[k][0] + w[k][0], k, f[k][1], 2 * f[k][0] / f[k][1], w[k][0] / max(w[k][1], 1)) for k in f), reverse=True)
    tot = sum(r[0] for r in rows)
    kinds = collections.defaultdict(float)
    for t, k, n, fb, wb in rows:
        kinds["weight gradient" if k[0].split(",")[3].strip() == "false" else
              ("forward" if k[0].split(",")[4].strip() == "true" else "input gradient")] += t
    print(f"GEMM HBM-side traffic over the run: {tot / 1e9:.1f} GB; by kind: "
          + ", ".join(f"{k} {100 * v / tot:.0f} %" for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])) + "\n")
    print("| share | launches | fetch MB / launch | write MB / launch | blocks | variant |")
    print("|---:|---:|---:|---:|---:|---|")
    for t, k, n, fb, wb in rows[:30]:
        print(f"| {100 * t / tot:.1f} % | {n} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | {k[1]} | `{k[0]}` |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
