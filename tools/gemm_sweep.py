"""Diagnostic: for every dense-layer shape of one training step, time the GEMM under forced tile /
split-K configurations (LOTUS_GEMM_TILE / LOTUS_GEMM_NZ are read once per process, so each config is a
child process).  Usage: python tools/gemm_sweep.py  -> prints the best config per (kind, M, N, K)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import torch
    import robot_3dlotus_amd
    from robot_3dlotus_amd import ops
    shapes = json.loads(os.environ["SHAPES"])
    out = {}
    for kind, M, N, K in shapes:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; dy = torch.randn(M, N, device="cuda")
        fn = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(dy, w), "wgrad": lambda: ops.linear_wgrad(dy, x)}[kind]
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); e1.synchronize()
        out[f"{kind} {M} {N} {K}"] = e0.elapsed_time(e1) / 5 * 1e3
    print("RESULT " + json.dumps(out))

if os.environ.get("SHAPES"):
    child(); sys.exit(0)

sys.path.insert(0, os.path.join(ROOT, "tests"))
shapes = json.load(open(os.path.join(ROOT, "tools", "gemm_shapes.json")))
res = {}
TILES = (0,) if os.environ.get('AUTO_ONLY') else (0, 1, 2, 3)
NZS = (0,) if os.environ.get('AUTO_ONLY') else (0, 1, 2, 4, 8, 16, 32, 64)
for tile in TILES:
    for nz in NZS:
        env = dict(os.environ, SHAPES=json.dumps(shapes), LOTUS_GEMM_TILE=str(tile), LOTUS_GEMM_NZ=str(nz))
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("config failed", tile, nz, r.stderr[-300:]); continue
        for k, v in json.loads(line[0][7:]).items():
            res.setdefault(k, {})[(tile, nz)] = v
tot_auto = tot_best = 0
for k, d in sorted(res.items(), key=lambda kv: -kv[1][(0, 0)]):
    best = min(d, key=d.get)
    cnt = next(c for (kk, *_), c in [((f"{s[0]} {s[1]} {s[2]} {s[3]}",), 1) for s in shapes] if kk == k)
    tot_auto += d[(0, 0)]; tot_best += d[best]
    print(f"{k:28s} auto {d[(0,0)]:8.1f} us   best {d[best]:8.1f} us  tile={best[0]} nz={best[1]}   " +
          " ".join(f"t{t}z{z}:{d[(t,z)]:.0f}" for (t, z) in sorted(d) if d[(t, z)] < 1.15 * d[best] and (t, z) != best))
print("sum auto", tot_auto, "sum best", tot_best)
