"""Diagnostic: for every dense-layer shape of one training step (tools/gemm_shapes.json, from `bench.py --gemm-report`), time the GEMM under forced tile / slab-depth / split-K configurations
(LOTUS_GEMM_TILE / LOTUS_GEMM_BK / LOTUS_GEMM_NZ are read once per process, so each config is a child
process).  Writes gpurun_out/gemm_sweep.json: {"kind M N K": {"count": c, "tile,bk,nz": us, ...}}."""
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
        for _ in range(8): fn()
        e1.record(); e1.synchronize()
        out[f"{kind} {M} {N} {K}"] = e0.elapsed_time(e1) / 8 * 1e3
    print("RESULT " + json.dumps(out))

if os.environ.get("SHAPES"):
    child(); sys.exit(0)

rows = json.load(open(os.path.join(ROOT, "tools", "gemm_shapes.json")))  # [kind, M, N, K, count] of one v1 step
shapes = [tuple(r[:4]) for r in rows]
res = {f"{r[0]} {r[1]} {r[2]} {r[3]}": {"count": r[4]} for r in rows}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
configs = [(0, 0, 0)] + [(1, 16, nz) for nz in (1, 2, 4)] + [(3, bk, nz) for bk in (16, 32, 64) for nz in (1, 2, 4, 8, 16, 32, 64, 128)]
for tile, bk, nz in configs:
    env = dict(os.environ, SHAPES=json.dumps(shapes), LOTUS_GEMM_TILE=str(tile), LOTUS_GEMM_NZ=str(nz), LOTUS_GEMM_BK=str(bk))
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print("config failed", tile, bk, nz, r.stderr[-300:]); continue
    for k, v in json.loads(line[0][7:]).items():
        res[k][f"{tile},{bk},{nz}"] = v
    print("done", tile, bk, nz, flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_sweep.json"), "w"))
