"""Turn the outputs of profiles/collect.sh into the committed evidence files of a round:

    python profiles/assemble.py r02

profiles/<tag>_pmc.json + <tag>_sq.md (pmc_report.py), <tag>_kernels.md (summarize.py), <tag>_timeline.txt (timeline.py)
and <tag>_bench.json (the bench lines, trimmed to the fields quoted in DESIGN.md / README.md)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
g = os.path.join(ROOT, "gpurun_out")
py = sys.executable
subprocess.run([py, os.path.join(ROOT, "profiles", "pmc_report.py"), tag, f"{g}/pmc_{tag}_fetch", f"{g}/pmc_{tag}_write", f"{g}/pmc_{tag}_sq",
                f"{g}/pmc_{tag}_so.sha"], check=True, cwd=ROOT)
db = f"{g}/prof/{tag}z_results.db"
open(os.path.join(ROOT, "profiles", f"{tag}_kernels.md"), "w").write(
    subprocess.run([py, os.path.join(ROOT, "profiles", "summarize.py"), db, "30"], check=True, capture_output=True, text=True).stdout)
open(os.path.join(ROOT, "profiles", f"{tag}_timeline.txt"), "w").write(
    subprocess.run([py, os.path.join(ROOT, "profiles", "timeline.py"), db], check=True, capture_output=True, text=True).stdout)
bdir = f"{g}/bench_{tag}"
flags = {"headline": "--steps 30 --warmup 10", "with_optimizer": "--with-optimizer", "mp": "--workload mp", "peract": "--workload peract",
         "peract_fp32_storage": "--workload peract --act-storage fp32", "peract_batch_64": "--workload peract --batch 64",
         "peract_fp32_storage_batch_64": "--workload peract --act-storage fp32 --batch 64", "peract_batch_128": "--workload peract --batch 128",
         "peract_fp32_storage_batch_128": "--workload peract --act-storage fp32 --batch 128",
         "ragged": "--ragged (clouds of U(2048, 4096) points)", "batch_38": "--batch 38 (the batch of the published A100 run, BASELINE.md 2)",
         "batch_32": "--batch 32", "batch_64": "--batch 64", "batch_128": "--batch 128",
         "one_rank_rccl": "LOTUS_FORCE_COLLECTIVES=1 (one-rank RCCL rehearsal of the DP step)",
         "two_ranks_one_gpu_gloo": "--gpus 2 on ONE device (gloo; functional check, both ranks share the GPU)"}
out = {"where": "1x MI355X (gpurun box), library sha256[:16] " + open(f"{bdir}/so.sha").read().strip() +
       ", v1 config unless noted; python bench.py flags in the key"}
for key, fl in flags.items():
    path = f"{bdir}/{key}.json"
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        out[key] = {"flags": fl, "error": f"no line ({e})"}
        continue
    if key == "headline":
        rec = {"flags": fl}
        for k in ("value", "unit", "ms_per_step", "n_gpus", "rccl_ranks", "host_ms_per_step", "opt_in_modes", "with_optimizer", "fresh_batches", "cpu_baseline", "roofline", "counters"):
            if k in d:
                rec[k] = d[k]
    else:
        rec = {"flags": fl, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "n_gpus": d["n_gpus"],
               "rccl_ranks": d.get("rccl_ranks")}
        for k in ("opt_in_modes", "host_ms_per_step", "comm", "reducer"):
            if k in d:
                rec[k] = d[k]
    out[key] = rec
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w"), indent=1)
print(json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
