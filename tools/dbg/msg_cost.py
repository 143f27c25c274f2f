"""Diagnostic (GPU box, one device): what ONE latency-bound statistics message costs the stream it is issued on, per way of
issuing it.  A chain of small dependent kernels on a high-priority stream T, with one fp64 all-reduce of 257 values between
every two of them:

  base      no message
  pg_sync   dist.all_reduce(sums, group)  -- torch >= 2.8 runs a blocking collective on the CURRENT stream (kernel + end event)
  pg_async  async_op=True + wait()        -- ProcessGroupNCCL's own stream: event, wait, kernel, event, wait
  pg_comm   sync collective under a communication stream C, T waits for C afterwards (what a hidden message costs T)
  direct    ncclAllReduce(..., stream = T) on a communicator of our own (ctypes on the librccl torch loaded): the kernel alone

prints microseconds per message = (t_variant - t_base) / n.  One rank: transfer time is nil, what is measured is the packets each
variant puts into T's queue."""
import ctypes
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=0, world_size=1)
g2 = dist.new_group(backend="nccl")
dev = torch.device("cuda", 0)
T = torch.cuda.Stream(priority=-1)
C = torch.cuda.Stream()
torch.cuda.set_stream(T)
x = torch.ones(1 << 16, device=dev)
sums = torch.zeros(257, dtype=torch.float64, device=dev)

lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))


class UID(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UID)]
lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
uid = UID()
assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
comm = ctypes.c_void_p()
assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0


def k():
    x.mul_(1.0000001)


def v_base():
    k()


def v_pg_sync():
    k()
    dist.all_reduce(sums, group=g2)


def v_pg_async():
    k()
    dist.all_reduce(sums, group=g2, async_op=True).wait()


def v_pg_comm():
    k()
    C.wait_stream(T)
    with torch.cuda.stream(C):
        dist.all_reduce(sums, group=g2)
    T.wait_stream(C)


def v_direct():
    k()
    rc = lib.ncclAllReduce(sums.data_ptr(), sums.data_ptr(), sums.numel(), 8, 0, comm, T.cuda_stream)
    assert rc == 0


def run(fn, n=400):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


res = {}
for name, fn in (("base", v_base), ("pg_sync", v_pg_sync), ("pg_async", v_pg_async), ("pg_comm", v_pg_comm), ("direct", v_direct),
                 ("base2", v_base)):
    res[name] = run(fn)
    print("%-9s %7.2f us per iteration" % (name, res[name]), flush=True)
for name in ("pg_sync", "pg_async", "pg_comm", "direct"):
    print("%-9s %7.2f us per message" % (name, res[name] - res["base"]))
if len(sys.argv) > 1 and sys.argv[1] == "big":  # a bucket-sized all-reduce (32 MB) the three ways, alone on the GPU
    big = torch.zeros(8 << 20, device=dev)
    for name, fn in (("pg_sync", lambda: dist.all_reduce(big, op=dist.ReduceOp.AVG)),
                     ("pg_async", lambda: dist.all_reduce(big, op=dist.ReduceOp.AVG, async_op=True).wait()),
                     ("direct", lambda: lib.ncclAllReduce(big.data_ptr(), big.data_ptr(), big.numel(), 7, 4, comm, T.cuda_stream))):
        print("32 MB %-9s %7.2f us" % (name, run(fn, 50)))
dist.destroy_process_group()
