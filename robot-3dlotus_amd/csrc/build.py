"""Build csrc/liblotus_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python robot-3dlotus_amd/csrc/build.py [--force]

One object per translation unit, compiled in parallel, linked into a C-ABI shared library with
no torch dependency.  Objects are rebuilt only when their sources changed.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["lotus_capi.cpp", "comm.cpp", "blocks.cpp", "gemm.hip", "gemm_dma.hip", "conv.hip", "conv_pairs.hip", "norm.hip", "attention.hip", "front_end.hip", "pool_head.hip", "optim.hip", "stream_probe.hip"]
HEADERS = ["common.h", "mma.h", "gemm_common.h", "gemm_dma.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# measurement builds (never the shipped library): LOTUS_BUILD_DEFINES="LOTUS_EXP_SKIP_PROBE" python build.py --force
FLAGS += ["-D" + d for d in os.environ.get("LOTUS_BUILD_DEFINES", "").split()]
# per-source extras: the LDS-DMA GEMM keeps its MFMA accumulators in VGPRs (its epilogue stores straight from them)
SOURCE_FLAGS = {"gemm_dma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
LIB = os.path.join(HERE, "liblotus_hip.so")
# The translation units that touch activation tensors are compiled twice: act_t = float (lotus_*) and, with
# -DLOTUS_ACT_BF16 and the generated rename header, act_t = bf16 (lotus_b16_*, include/lotus_hip_b16.h) — gen_twin.py.
ACT_SOURCES = ["blocks.cpp", "gemm.hip", "gemm_dma.hip", "conv.hip", "conv_pairs.hip", "norm.hip", "attention.hip", "pool_head.hip"]
RENAME = os.path.join(HERE, "lotus_rename_b16.h")
PUBLIC = [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "lotus_hip.h")]


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(job):
    src, b16 = job
    obj = os.path.join(HERE, os.path.splitext(src)[0] + ("_b16.o" if b16 else ".o"))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS] + PUBLIC + ([RENAME] if b16 else [])
    if not _stale(obj, deps):
        return obj, ""
    # Reproducible objects: hipcc derives the compilation-unit id (-cuid, baked into symbol names of
    # the device code) from the ABSOLUTE input path and embeds the output name, so the same sources built in another
    # directory gave a library with a different hash than the one the profiles were taken on.  Compile with relative
    # names from the source directory and a fixed cuid per object: the bytes then depend on the sources and the
    # toolchain only.
    extra = ["-DLOTUS_ACT_BF16", "-include", os.path.basename(RENAME)] if b16 else []
    oname = os.path.basename(obj)
    cmd = ([HIPCC] + FLAGS + SOURCE_FLAGS.get(src, []) + ["-cuid=lotus-" + os.path.splitext(oname)[0], "-ffile-prefix-map=" + HERE + "=."] + extra +
           (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", oname])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=HERE)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    return obj, r.stderr


def build(force=False):
    sys.path.insert(0, HERE)
    import gen_twin

    gen_twin.generate()  # rename header + include/lotus_hip_b16.h (rewritten only when they change)
    jobs = [(s, False) for s in SOURCES] + [(s, True) for s in ACT_SOURCES]
    if force:
        for s, b16 in jobs:
            o = os.path.join(HERE, os.path.splitext(s)[0] + ("_b16.o" if b16 else ".o"))
            if os.path.exists(o):
                os.remove(o)
    with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 8)) as ex:
        objs = [o for o, _ in ex.map(_compile, jobs)]
    if _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.basename(LIB)] +
                           [os.path.basename(o) for o in objs] + ["-ldl"], capture_output=True, text=True, cwd=HERE)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    build_fastcall()
    return LIB


def build_fastcall():
    """CPython trampolines (generated from include/lotus_hip.h) -> csrc/_lotus_fastcall.so, linked against the C-ABI
    library next to it.  Host-side only: plain gcc."""
    import sysconfig

    sys.path.insert(0, HERE)
    import gen_fastcall

    src, _ = gen_fastcall.generate()
    out = os.path.join(HERE, "_lotus_fastcall.so")
    inc = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
    if _stale(out, [src, os.path.join(inc, "lotus_hip.h"), os.path.join(inc, "lotus_hip_b16.h"), LIB]):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], src, "-o", out,
               "-L" + HERE, "-llotus_hip", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("fastcall module failed to build:\n" + r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
