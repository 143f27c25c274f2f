"""Diagnostic: A/B of one GEMM tuning switch over the forward / input-gradient shapes of a training step
(tools/gemm_shapes.json), each setting in its own process.   python tools/gemm_ab.py LOTUS_GEMM_DMA 0 1"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import torch
    import robot_3dlotus_amd
    from robot_3dlotus_amd import ops
    out = {}
    for kind, M, N, K in json.loads(os.environ["SHAPES"]):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; dy = torch.randn(M, N, device="cuda")
        fn = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(dy, w), "wgrad": lambda: ops.linear_wgrad(dy, x)}[kind]
        for _ in range(3): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        out[f"{kind} {M} {N} {K}"] = best
    print("RESULT " + json.dumps(out))

if os.environ.get("SHAPES"):
    child(); sys.exit(0)
var, vals = sys.argv[1], sys.argv[2:]
rows = [r for r in json.load(open(os.path.join(ROOT, "tools", "gemm_shapes.json"))) if r[0] in ("fwd", "dgrad")]
res = {}
for v in vals:
    env = dict(os.environ, SHAPES=json.dumps([r[:4] for r in rows]))
    env[var] = v
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    res[v] = json.loads(line[0][7:]) if line else print(r.stderr[-500:])
tot = {v: 0.0 for v in vals}
print("shape count " + " ".join(f"{var}={v}" for v in vals))
for kind, M, N, K, c in sorted(rows, key=lambda r: -r[4] * res[vals[0]][f"{r[0]} {r[1]} {r[2]} {r[3]}"]):
    k = f"{kind} {M} {N} {K}"
    for v in vals: tot[v] += c * res[v][k]
    print(k, c, " ".join("%.1f" % res[v][k] for v in vals))
print("weighted total ms/step", {v: round(t / 1e3, 3) for v, t in tot.items()})
