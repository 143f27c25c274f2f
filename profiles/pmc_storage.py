"""profiles/rNN_pmc_storage.json: HBM bytes per training step of the PerAct workload with bf16 vs fp32 activation storage, from
the four counter passes of profiles/pmc_storage.sh (FETCH_SIZE doubled, WRITE_SIZE as reported: MI355X_MICROARCH.md, HBM
section; the same corrections as profiles/pmc_summary.py).  Steps counted from the launches of the once-per-step loss kernel.

    python profiles/pmc_storage.py r03
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(pattern):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", pattern, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter csv under gpurun_out/{pattern}")
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        agg[r["Kernel_Name"]][0] += float(r["Counter_Value"])
        agg[r["Kernel_Name"]][1] += 1
    return agg


def family(name):
    n = name.replace("void ", "")
    for ns in ("lotus_b16::", "lotus_f32::"):
        n = n.replace(ns, "")
    if "gemm_kernel" in n:
        return "gemm_kernel"
    if n.startswith("_ZN9lotus_"):
        n = n[len("_ZN9lotus_b16"):].lstrip("0123456789")
    return n.split("(")[0].split("<")[0][:36]


def main(tag):
    out = {"source": __doc__.split("\n\n")[0].strip(), "workload": "bench.py --workload peract (16 clouds x 4096 points, bf16 operands)"}
    per = {}
    for s in ("bf16", "fp32"):
        f, w = load(f"pmc_{tag}_peract_{s}_fetch"), load(f"pmc_{tag}_peract_{s}_write")
        steps = max(c for n, (v, c) in f.items() if "small_loss_kernel" in n)
        fam = collections.defaultdict(lambda: [0.0, 0.0])
        for n, (v, c) in f.items():
            fam[family(n)][0] += 2 * v * 1024 / steps
        for n, (v, c) in w.items():
            fam[family(n)][1] += v * 1024 / steps
        tot_f, tot_w = sum(a for a, _ in fam.values()), sum(b for _, b in fam.values())
        per[s] = fam
        out[s + "_storage"] = {"steps_profiled": steps, "fetch_MB_per_step": round(tot_f / 1e6, 1), "write_MB_per_step": round(tot_w / 1e6, 1),
                               "hbm_MB_per_step": round((tot_f + tot_w) / 1e6, 1)}
    out["ratio_bf16_over_fp32"] = round(out["bf16_storage"]["hbm_MB_per_step"] / out["fp32_storage"]["hbm_MB_per_step"], 3)
    rows = {}
    for k in sorted(per["fp32"], key=lambda k: -sum(per["fp32"][k]))[:14]:
        a, b = per["fp32"][k], per["bf16"].get(k, [0.0, 0.0])
        rows[k] = {"fp32_MB": round(sum(a) / 1e6, 1), "bf16_MB": round(sum(b) / 1e6, 1)}
    out["families_MB_per_step"] = rows
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_storage.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "families_MB_per_step"}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
