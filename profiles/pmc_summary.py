"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace --output-format csv) into
profiles/r01_pmc_hbm.json: HBM-side bytes per launch and kernel family.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    python profiles/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KB and reports half of the bytes
of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE (KB) is used as reported.  Calibration on ln_fwd_kernel
(reads M*C*4 B, writes M*C*4 B): 2 x 6.0 MB fetched / 10.2 MB written per launch against 10.5 / 10.5 MB algorithmic."""
import collections
import csv
import json
import os
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][0] += float(r["Counter_Value"])
        agg[r["Kernel_Name"]][1] += 1
    return agg


def main(fetch_csv, write_csv):
    f, w = load(fetch_csv), load(write_csv)
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for n, (v, c) in f.items():
        k = "gemm_kernel" if "gemm_kernel" in n else n.split("(")[0].replace("void ", "")[:40]
        fam[k][0] += 2 * v * 1024
        fam[k][1] += w.get(n, [0, 0])[0] * 1024
        fam[k][2] += c
    out = {"source": __doc__.split("\n\n")[1].strip(), "kernels": {}}
    for k, (fb, wb, c) in sorted(fam.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:20]:
        out["kernels"][k] = {"launches": c, "fetch_bytes_per_launch": round(fb / c), "write_bytes_per_launch": round(wb / c),
                             "hbm_bytes_per_launch": round((fb + wb) / c)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r01_pmc_hbm.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["kernels"].get("gemm_kernel")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
