"""Turn the three rocprofv3 counter passes of profiles/pmc_passes.sh into profiles/<tag>_pmc.json and <tag>_sq.md.

    python profiles/pmc_report.py r02 gpurun_out/pmc_r02_fetch gpurun_out/pmc_r02_write gpurun_out/pmc_r02_sq gpurun_out/pmc_r02_so.sha

Per kernel family (counter passes serialise the kernels, so durations are stand-alone durations):
  * HBM-side bytes per launch: FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM
    section) -> doubled; WRITE_SIZE (KB) as reported; calibration rows for ln_fwd_kernel (reads and writes M*C*4 B)
    are printed so the correction can be checked;
  * achieved GB/s = (fetch + write) / duration;
  * MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMD-cycles of the launch) with SIMD-cycles = GRBM_GUI_ACTIVE / 8 XCDs
    x 1024 SIMDs.  Calibration: the 65536 x 128 x 512 fp32 GEMM issues 2 M N K / 4096 = 2 097 152 v_mfma_f32_32x32x2_f32
    of 64 cycles each = 134 217 728 busy cycles — exactly the counter value;
    NORMALISATION: GRBM_GUI_ACTIVE spans more than the kernel's own execution (`effective_clock_GHz` =
    GRBM_GUI_ACTIVE / 8 / duration reads 2.6-2.9 for short kernels, above the 2.4 GHz maximum), so this quotient UNDERSTATES the
    utilisation by up to ~1.2x; `mfma_util_time` = busy cycles / (stand-alone duration x 2.4 GHz x 1024 SIMDs) is the
    time-normalised figure at the peak clock (an upper bound of the error the other way: the clock may sit below 2.4 GHz);
  * where the waves' time goes: SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY (issue stalls: MFMA RAW / pipe) / SQ_WAIT_ANY
    (parked on s_waitcnt or a barrier) as fractions of SQ_WAVE_CYCLES; SQ_WAIT_INST_LDS share.
The .so hash the passes ran against is stored; bench.py only reports `traffic` when it matches the library it loaded."""
import collections
import csv
import json
import os
import sys

FAMILIES = [("gemm_dma_tap_kernel", "gemm_dma_tap_kernel (tap-grouped conv on the LDS-DMA tiles, every level)"), ("conv_tap_reduce_kernel", "conv_tap_reduce_kernel"),
            ("attn_bwd2_kernel", "attn_bwd2_kernel"), ("gemm_dma_kernel", "gemm_dma_kernel (LDS-DMA dense kernels, levels 0-1)"), ("gemm_kernel", "gemm_kernel"), ("conv_pairs_kernel", "conv_pairs_kernel"), ("conv_os_kernel", "conv_os_kernel (bf16 modes)"),
            ("conv_wgrad_bf16_kernel", "conv_wgrad_bf16_kernel"), ("conv_wgrad_kernel", "conv_wgrad_kernel"),
            ("xq_fwd_kernel", "xq_fwd_kernel (cross attention)"), ("xq_bwd_kernel", "xq_bwd_kernel (cross attention)"),
            ("conv_smallcin", "conv_smallcin (stem)"), ("attn_fwd_kernel", "attn_fwd_kernel"), ("attn_bwd_kernel", "attn_bwd_kernel"),
            ("fe_neighbour_kernel", "fe_neighbour_kernel"), ("fe_hash_build_kernel", "fe_hash_build_kernel"), ("ln_bwd_kernel", "ln_bwd_kernel"),
            ("ln_fwd_kernel", "ln_fwd_kernel"), ("bn_stat_kernel", "bn_stat_kernel"), ("bn_apply_kernel", "bn_apply_kernel"),
            ("reduce_parts_kernel", "reduce_parts_kernel"), ("rs_scatter_kernel", "rs_scatter_kernel (radix sort)"),
            ("pool_max", "pool_max_*"), ("unpool", "unpool_*"), ("pos_ce", "pos_ce_*"), ("cloud_max", "cloud_max_*")]


def family(name):
    # the tap-grouped sparse convolution of the deep levels runs on the dense kernel (template flag TAP = last argument true):
    # its launches are convolution work and get a row of their own
    if "gemm_kernel" in name and name.split("(")[0].rstrip().endswith("true>"):
        return "gemm_kernel<..., TAP> (tap-grouped conv, levels 2-4)"
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None


def load(dirname, prefix):
    path = os.path.join(dirname, prefix + "_counter_collection.csv")
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = per[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"]
        d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(per.values())


def main(tag, fetch_dir, write_dir, sq_dir, sha_file):
    here = os.path.dirname(os.path.abspath(__file__))
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in load(fetch_dir, "f"):
        f = family(d["name"])
        if f:
            fam[f]["fetch"] += 2.0 * d.get("FETCH_SIZE", 0.0) * 1024
            fam[f]["n_f"] += 1
            fam[f]["dur_f"] += d["dur"]
    for d in load(write_dir, "w"):
        f = family(d["name"])
        if f:
            fam[f]["write"] += d.get("WRITE_SIZE", 0.0) * 1024
            fam[f]["n_w"] += 1
    sq_keys = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
               "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")
    for d in load(sq_dir, "s"):
        f = family(d["name"])
        if f:
            for k in sq_keys:
                fam[f][k] += d.get(k, 0.0)
            fam[f]["n_s"] += 1
            fam[f]["dur_s"] += d["dur"]
    out = {"tag": tag, "library_sha256_16": open(sha_file).read().strip() if os.path.exists(sha_file) else None,
           "source": "rocprofv3 --pmc passes of profiles/pmc_passes.sh over `bench.py --steps 3 --warmup 2` (v1, 16 x 4096, fp32); "
                     "FETCH_SIZE doubled (gfx950), WRITE_SIZE as reported; kernels run serialised under counter collection",
           "kernels": {}}
    lines = [f"# {tag}: SQ / HBM counters per kernel family (rocprofv3 --pmc, kernels serialised)\n",
             "MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); wave-time split = fractions of SQ_WAVE_CYCLES; "
             "GB/s = (2 x FETCH_SIZE + WRITE_SIZE) / stand-alone duration.\n",
             "MFMA util (time) = the same busy cycles / (stand-alone duration x 2.4 GHz x 1024 SIMDs): GRBM_GUI_ACTIVE covers more than the "
             "kernel (effective clock > 2.4 GHz on short kernels), so the first column is a lower bound.\n",
             "| kernel family | launches/step | avg us | HBM MB/launch (fetch + write) | GB/s | MFMA util | MFMA util (time) | active | issue-stall | parked (waitcnt/barrier) | LDS-issue share |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    steps = 5.0  # 2 warm-up + 3 timed steps in every pass
    for f, v in sorted(fam.items(), key=lambda kv: -kv[1]["dur_s"]):
        nf, nw, ns = max(v["n_f"], 1), max(v["n_w"], 1), max(v["n_s"], 1)
        fetch, write = v["fetch"] / nf, v["write"] / nw
        dur_us = v["dur_f"] / nf / 1e3 if v["n_f"] else v["dur_s"] / ns / 1e3
        gbps = (fetch + write) / (dur_us * 1e-6) / 1e9 if dur_us else 0.0
        simd_cycles = v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles if simd_cycles else 0.0
        util_t = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["dur_s"] * 2.4 * 1024.0) if v["dur_s"] else 0.0  # dur in ns x 2.4 cycles/ns
        wc = max(v["SQ_WAVE_CYCLES"], 1.0)
        rec = {"launches_per_step": round(v["n_s"] / steps, 1), "avg_us": round(dur_us, 2),
               "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
               "hbm_bytes_per_launch": round(fetch + write), "achieved_GBps": round(gbps, 1),
               "mfma_util": round(util, 4), "mfma_util_time": round(util_t, 4), "wave_active": round(v["SQ_ACTIVE_INST_ANY"] / wc, 4),
               "wave_issue_stall": round(v["SQ_WAIT_INST_ANY"] / wc, 4), "wave_parked": round(v["SQ_WAIT_ANY"] / wc, 4),
               "lds_issue_stall": round(v["SQ_WAIT_INST_LDS"] / wc, 4),
               "effective_clock_GHz": round(v["GRBM_GUI_ACTIVE"] / 8.0 / max(v["dur_s"], 1.0), 3)}
        out["kernels"][f] = rec
        lines.append(f"| `{f}` | {rec['launches_per_step']} | {rec['avg_us']} | {fetch / 1e6:.2f} + {write / 1e6:.2f} | {gbps:.0f} | "
                     f"{util:.3f} | {util_t:.3f} | {rec['wave_active']:.2f} | {rec['wave_issue_stall']:.2f} | {rec['wave_parked']:.2f} | {rec['lds_issue_stall']:.3f} |")
    json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
    open(os.path.join(here, f"{tag}_sq.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:6])
