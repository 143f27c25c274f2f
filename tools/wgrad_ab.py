"""Diagnostic: the dense weight-gradient shapes of one training step (tools/gemm_shapes.json) under settings of one
environment switch, each setting in its own process: correctness against float64 (dw, db; accumulate mode; odd row
counts) and stand-alone time per shape.   python tools/wgrad_ab.py LOTUS_WGRAD_STREAM 0 1 12 16"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import torch
    import robot_3dlotus_amd
    from robot_3dlotus_amd import ops
    out, err = {}, {}
    torch.manual_seed(0)
    for kind, M, N, K in json.loads(os.environ["SHAPES"]):
        x = torch.randn(M, K, device="cuda"); dy = torch.randn(M, N, device="cuda")
        dw, db = ops.linear_wgrad(dy, x)
        torch.cuda.synchronize()
        ref_w = (dy.double().t() @ x.double()); ref_b = dy.double().sum(0)
        e_w = float((dw.double() - ref_w).abs().max() / ref_w.abs().max()); e_b = float((db.double() - ref_b).abs().max() / ref_b.abs().max())
        dw2, db2 = ops.linear_wgrad(dy, x)           # determinism
        same = bool(torch.equal(dw, dw2) and torch.equal(db, db2))
        ops.linear_wgrad(dy, x, into=(dw2, db2))     # accumulate: 2 x
        torch.cuda.synchronize()
        e_acc = float((dw2.double() - 2 * ref_w).abs().max() / ref_w.abs().max())
        err[f"{M} {N} {K}"] = [e_w, e_b, e_acc, same]
        fn = lambda: ops.linear_wgrad(dy, x)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        out[f"{M} {N} {K}"] = best
    print("RESULT " + json.dumps({"us": out, "err": err}))

if os.environ.get("SHAPES"):
    child(); sys.exit(0)
var, vals = sys.argv[1], sys.argv[2:]
rows = [r for r in json.load(open(os.path.join(ROOT, "tools", "gemm_shapes.json"))) if r[0] == "wgrad" and r[2] % 4 == 0 and r[3] % 4 == 0]
rows += [["wgrad", 4099, 128, 64, 0], ["wgrad", 1001, 64, 64, 0], ["wgrad", 257, 192, 64, 0]]  # odd row counts (correctness only: count 0)
res = {}
for v in vals:
    env = dict(os.environ, SHAPES=json.dumps([r[:4] for r in rows]))
    env[var] = v
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print("setting", v, "failed:", r.stderr[-800:]); sys.exit(1)
    res[v] = json.loads(line[0][7:])
tot = {v: 0.0 for v in vals}
print("M N K count roof_us | " + " ".join(f"{var}={v}" for v in vals) + " | worst rel err (dw, db, accumulate) deterministic")
bad = 0
for kind, M, N, K, c in sorted(rows, key=lambda r: -r[4] * res[vals[0]]["us"][f"{r[1]} {r[2]} {r[3]}"]):
    k = f"{M} {N} {K}"
    for v in vals: tot[v] += c * res[v]["us"][k]
    roof = max(2.0 * M * N * K / 157.3e12, 4.0 * (M * N + M * K + N * K) / 6.3e12) * 1e6
    e = [max(res[v]["err"][k][i] for v in vals) for i in range(3)]
    det = all(res[v]["err"][k][3] for v in vals)
    if max(e) > 2e-5 or not det: bad += 1
    print(k, c, "%.1f |" % roof, " ".join("%.1f" % res[v]["us"][k] for v in vals), "| %.1e %.1e %.1e %s" % (e[0], e[1], e[2], det))
print("weighted total ms/step", {v: round(t / 1e3, 3) for v, t in tot.items()}, "shapes out of tolerance:", bad)
