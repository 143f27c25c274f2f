"""Diagnostic: what creating RCCL communicators does to the HOST side of the process (CPU affinity, threads that spin)."""
import os, sys, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import parallel

def pyspeed():
    t = time.perf_counter(); x = 0
    for i in range(2000000): x += i
    return time.perf_counter() - t

def threads():
    out = {}
    for p in glob.glob("/proc/self/task/*/stat"):
        try:
            f = open(p).read().split(")")[1].split()
            out[p.split("/")[4]] = (int(f[11]) + int(f[12]))  # utime + stime (ticks)
        except OSError:
            pass
    return out

def report(tag):
    a = os.sched_getaffinity(0)
    t0 = threads(); time.sleep(1.0); t1 = threads()
    busy = {k: t1[k] - t0.get(k, 0) for k in t1 if t1[k] - t0.get(k, 0) > 5}
    print(f"{tag}: affinity {len(a)} cpus {sorted(a)[:4]}..{sorted(a)[-2:]}, threads {len(t1)}, busy threads (ticks/s) {busy}, python loop {pyspeed()*1e3:.0f} ms", flush=True)

report("start")
torch.zeros(1, device="cuda")
report("after cuda init")
os.environ["LOTUS_FORCE_COLLECTIVES"] = "1"
parallel.init_distributed()
report("after init_process_group")
t = torch.zeros(4, device="cuda"); torch.distributed.all_reduce(t); torch.cuda.synchronize()
report("after first PG collective")
c = parallel.native_comm(None, "main")
report("after native lane main")
c2 = parallel.native_comm(None, "comm")
report("after native lane comm")
