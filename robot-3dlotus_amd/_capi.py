"""ctypes binding of csrc/liblotus_hip.so.

The prototypes are read from include/lotus_hip.h (single source of truth), so a signature change
in the header is picked up here and checked by tests/test_capi.py.  There is no CPU fallback: if the
library is missing, `lib()` raises with the build instruction.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblotus_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lotus_hip.h")
TWIN_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lotus_hip_b16.h")  # bf16-storage twins (generated)

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "unsigned long long": ctypes.c_ulonglong, "unsigned": ctypes.c_uint,
}


def parse_header(path=None):
    """-> {name: (restype, [argtypes], [argnames])} for every `lotus_*` prototype (both headers when path is None)."""
    if path is None:
        src = open(HEADER_PATH).read() + (open(TWIN_HEADER_PATH).read() if os.path.exists(TWIN_HEADER_PATH) else "")
    else:
        src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\s*\b(lotus_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret == "const char*":
            restype = ctypes.c_char_p
        else:
            restype = _CTYPES[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                else:
                    ty, nm = a.rsplit(" ", 1)
                    ty = ty.replace("const ", "").strip()
                    argtypes.append(_CTYPES[ty])
                    argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


ABI_VERSION = 3  # lotus_abi_version() of the library this binding matches (include/lotus_hip.h)


class LotusError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise LotusError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python robot-3dlotus_amd/csrc/build.py` or `__graft_entry__.build()`.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fn = {}
        for name, (restype, argtypes, _) in self.protos.items():
            f = getattr(self.cdll, name)
            f.restype = restype
            f.argtypes = argtypes
            self.fn[name] = f
        got = self.fn["lotus_abi_version"]()
        if got != ABI_VERSION:  # (an entry point changed its arguments: a stale library would take them misaligned, silently)
            raise LotusError(f"{LIB_PATH} has ABI version {got}, this binding was written for {ABI_VERSION}: rebuild the library")

    def last_error(self):
        return self.fn["lotus_last_error"]().decode()


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def _conv(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = torch.cuda.current_device


def stream_ptr():
    """hipStream_t of torch's current stream (the raw-handle query is ~20x cheaper than building a
    torch.cuda.Stream object, and this runs once per kernel launch)."""
    if _raw_stream is not None:
        return _raw_stream(_get_dev() if _get_dev is not None else _cur_dev())
    return torch.cuda.current_stream().cuda_stream


def _load_fastcall():
    """csrc/_lotus_fastcall.so: generated METH_FASTCALL trampolines to the same entry points (csrc/gen_fastcall.py).
    ~1 us per call instead of ~10 us through ctypes; optional — ctypes is the fallback binding."""
    path = os.path.join(_HERE, "csrc", "_lotus_fastcall.so")
    if os.environ.get("LOTUS_NO_FASTCALL") == "1" or not os.path.exists(path):
        return None
    lib()  # liblotus_hip.so first (the module links it by $ORIGIN rpath)
    import importlib.machinery
    import importlib.util
    loader = importlib.machinery.ExtensionFileLoader("_lotus_fastcall", path)
    spec = importlib.util.spec_from_loader("_lotus_fastcall", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    missing = [n for n in lib().protos if not hasattr(mod, n)]
    if missing:  # header changed after the module was built
        raise LotusError(f"_lotus_fastcall.so is stale (missing {missing[:3]}...): rerun robot-3dlotus_amd/csrc/build.py")
    return mod


_FAST = None
_FAST_TRIED = False
_get_dev = getattr(torch._C, "_cuda_getDevice", None)


def fastcall():
    global _FAST, _FAST_TRIED
    if not _FAST_TRIED:
        _FAST = _load_fastcall()
        _FAST_TRIED = True
    return _FAST


# Priority of the streams the package creates (weight-gradient, front-end, communication): see DESIGN.md "streams"
STREAM_PRIORITY = int(os.environ.get("LOTUS_STREAM_PRIO", "0"))

_STEP_STREAMS = {}


def step_stream(name):
    """The package's stream `name` ("train", "side" = weight gradients, "comm" = gradient buckets, "fe" = front-end prefetch),
    created on first request with STREAM_PRIORITY ("train": always high)."""
    st = _STEP_STREAMS.get(name)
    if st is None:
        st = _STEP_STREAMS[name] = torch.cuda.Stream(priority=-1 if name == "train" else STREAM_PRIORITY)
    return st


def prime_step_streams(names=("train", "side", "comm", "fe")):
    """Create the step's streams NOW, back to back, and put one packet into each so that their hardware queues exist.
    Hardware queues are dealt onto the compute pipes of the GPU in creation order, and two queues of one pipe take turns
    instead of running side by side: when the training stream and the weight-gradient stream end up on one pipe the backward
    pass is ~6 % slower — about one process in four, when the queues are created lazily in between those of RCCL's threads
    (measured: 12 fresh processes per setting).  Created consecutively, before any communicator, the four
    queues sit on four different pipes in every process.  -> the training stream (make it current: torch.cuda.set_stream)."""
    for n in names:
        with torch.cuda.stream(step_stream(n)):
            torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    return _STEP_STREAMS["train"]


# When non-zero, every call() enqueues on this hipStream_t instead of torch's current stream (ops._OnSide: the
# weight-gradient stream).  Cheaper than switching torch's current stream, which nothing inside those blocks needs.
STREAM_OVERRIDE = 0

# Activation storage of the calls being issued: False = fp32 entry points (lotus_*), True = their bf16-storage twins
# (lotus_b16_*, include/lotus_hip_b16.h).  Set per forward pass / per autograd node by ops (like the operand precision).
BF16 = False
_TWIN = {}


def _twin(name):
    t = _TWIN.get(name)
    if t is None:
        cand = "lotus_b16_" + name[6:]
        t = _TWIN[name] = cand if cand in lib().protos else name  # entry points without activations have no twin
    return t


def call(name, *args):
    """Call an int-returning entry point; tensors -> device pointers; appends the current stream."""
    if BF16:
        name = _TWIN.get(name) or _twin(name)
    F = _FAST if _FAST_TRIED else fastcall()
    if F is not None:
        rc = getattr(F, name)(*args, STREAM_OVERRIDE or _raw_stream(_get_dev()))
        if rc != 0:
            raise LotusError(f"{name} failed ({rc}): {F.lotus_last_error()}")
        return
    L = _LIB or lib()
    rc = L.fn[name](*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], STREAM_OVERRIDE or stream_ptr())
    if rc != 0:
        raise LotusError(f"{name} failed ({rc}): {L.last_error()}")


def call_raw(name, *args):
    """Entry points without a trailing stream parameter (stream link)."""
    if BF16:
        name = _TWIN.get(name) or _twin(name)
    F = _FAST if _FAST_TRIED else fastcall()
    if F is not None:
        rc = getattr(F, name)(*args)
    else:
        rc = lib().fn[name](*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args])
    if rc != 0:
        raise LotusError(f"{name} failed ({rc}): {lib().last_error()}")


def query(name, *args):
    """Call a size_t-returning *_workspace() function."""
    if BF16:
        name = _TWIN.get(name) or _twin(name)
    F = _FAST if _FAST_TRIED else fastcall()
    if F is not None:
        return getattr(F, name)(*args)
    return lib().fn[name](*args)


def wait_event(ev):
    """Block the host until `ev` (a torch.cuda.Event recorded earlier) has completed — by polling hipEventQuery.

    NOT ev.synchronize(): on this ROCm runtime hipEventSynchronize of an event that is not yet complete enqueues a NEW
    marker at the current TAIL of the event's hardware queue and waits for that one, i.e. for everything submitted since —
    on the event's own stream and on every stream that shares its hardware queue (HIP maps its streams onto
    GPU_MAX_HW_QUEUES = 4 of them per priority).  Measured in the data-parallel rehearsal: the host waited 11-16 ms per
    step for an event the GPU had passed long before — the count copy of the prefetched front-end (its stream shared a
    queue with the weight-gradient stream) and the one-step-old usage flags (recorded on the training stream itself) —
    which serialised host and GPU (28 instead of 15.5 ms per step) and changed with every re-mapping of streams to queues.
    A query never enqueues anything."""
    if ev.query():
        return
    import time
    t0 = time.perf_counter()
    while not ev.query():
        if time.perf_counter() - t0 > 2e-4:
            time.sleep(2e-5)   # the GPU is far behind: stop spinning on the core the launch threads need


class Workspace:
    """Grow-only per-device scratch buffer.  All ops run in stream order on the current stream, so
    one buffer can be reused by consecutive calls."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes, device, slot=0):
        key = (device, slot)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


WS = Workspace()
