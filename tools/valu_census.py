"""Where the SIMD time of a training step goes, per kernel (round 5).

On gfx950 an fp32 MFMA and any vector-ALU instruction of the same SIMD never overlap (tools/ubench/mfma_coissue.hip): the
time a kernel NEEDS on the SIMDs is   64 x (32x32x2 MFMAs) + ~5 x (other VALU instructions)   cycles, whatever it hides
behind memory.  This tool runs ONE rocprofv3 counter pass over `bench.py --steps 3` and prints that demand per kernel next to
its duration:
    python tools/valu_census.py run  [bench args]     (on the GPU box; writes gpurun_out/census/)
    python tools/valu_census.py report gpurun_out/census [> profiles/rNN_valu_census.md]
"""
import collections, csv, glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES"
STEPS, WARM = 3, 2


def run(extra):
    out = os.path.join(ROOT, "gpurun_out", "census")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = (["rocprofv3", "--pmc"] + COUNTERS.split() + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "c", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", str(WARM), "--no-cpu-baseline", "--no-roofline",
           "--no-other-modes", "--no-fresh-batches"] + extra)
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True)
    open(os.path.join(out, "run.log"), "w").write(r.stdout[-4000:] + r.stderr[-4000:])
    print("rc", r.returncode)


def short(name):
    n = name.replace("lotus_f32::", "").replace("lotus_b16::", "b16::").replace("void ", "")
    if "(" in n and not n.startswith("("):
        n = n[:n.rfind("(")] if n.rfind("(") > 0 else n
    return n[:110]


def report(d):
    path = glob.glob(os.path.join(d, "**", "c_counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        e = per[r["Dispatch_Id"]]
        e["name"] = r["Kernel_Name"]
        e["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for e in per.values():
        f = fam[short(e["name"])]
        for k, v in e.items():
            if k != "name":
                f[k] += v
        f["n"] += 1
    steps = STEPS + WARM + 1  # (+ the extra steps bench.py runs around the timed ones: normalise by launches of a once-per-step kernel)
    once = [f["n"] for k, f in fam.items() if "small_loss_kernel" in k]
    steps = once[0] if once else steps
    rows = []
    for k, f in fam.items():
        mfma, valu = f["SQ_INSTS_MFMA"], f["SQ_INSTS_VALU"]
        other = max(0.0, valu - mfma)               # SQ_INSTS_VALU counts the MFMAs too
        busy = f["SQ_VALU_MFMA_BUSY_CYCLES"]        # cycles (64 per fp32 32x32x2, 32 per bf16 32x32x16)
        demand = busy + 5.0 * other                 # SIMD cycles
        rows.append((demand / steps, k, f["n"] / steps, f["dur"] / steps / 1e3, mfma / steps, other / steps, busy / steps,
                     f["SQ_INSTS_SALU"] / steps, f["SQ_INSTS_LDS"] / steps, f["SQ_INSTS_VMEM"] / steps))
    rows.sort(reverse=True)
    simd_hz = 1024 * 2.4e9
    tot_d = sum(r[0] for r in rows); tot_t = sum(r[3] for r in rows)
    print("# SIMD demand per kernel: 64-cycle fp32 MFMAs (busy cycles) + 5 cycles per other VALU wave-instruction, per step; `demand us` = that over")
    print("# 1024 SIMDs x 2.4 GHz = the time the kernel would take if nothing but its vector-ALU / MFMA issue bounded it; serialised stand-alone durations.")
    print(f"\ntotal: demand {tot_d / simd_hz * 1e6:.0f} us/step (MFMA {sum(r[6] for r in rows) / simd_hz * 1e6:.0f} + other VALU {sum(5 * r[5] for r in rows) / simd_hz * 1e6:.0f}), stand-alone kernel time {tot_t:.0f} us/step\n")
    print("| demand us | MFMA us | VALU us | stand-alone us | launches | VALU instr / MFMA | SALU M | LDS M | VMEM M | kernel |")
    print("|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|")
    for dem, k, n, dur, mfma, other, busy, salu, lds, vmem in rows[:60]:
        print(f"| {dem / simd_hz * 1e6:.0f} | {busy / simd_hz * 1e6:.0f} | {5 * other / simd_hz * 1e6:.0f} | {dur:.0f} | {n:.0f} | {other / mfma if mfma else float('nan'):.2f} | {salu / 1e6:.2f} | {lds / 1e6:.2f} | {vmem / 1e6:.2f} | `{k}` |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2:])
    else:
        report(sys.argv[2])
