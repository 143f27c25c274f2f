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
SOURCES = ["lotus_capi.cpp", "blocks.cpp", "gemm.hip", "conv.hip", "conv_pairs.hip", "norm.hip", "attention.hip", "front_end.hip", "pool_head.hip", "optim.hip"]
HEADERS = ["common.h", "mma.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
LIB = os.path.join(HERE, "liblotus_hip.so")


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(HERE, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS]
    if not _stale(obj, deps):
        return obj, ""
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    return obj, r.stderr


def build(force=False):
    if force:
        for s in SOURCES:
            o = os.path.join(HERE, os.path.splitext(s)[0] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = [o for o, _ in ex.map(_compile, SOURCES)]
    if _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
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
    hdr = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "lotus_hip.h")
    if _stale(out, [src, hdr, LIB]):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], src, "-o", out,
               "-L" + HERE, "-llotus_hip", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("fastcall module failed to build:\n" + r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
