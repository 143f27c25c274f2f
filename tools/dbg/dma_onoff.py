"""A/B of the LDS-DMA dense kernels (LOTUS_GEMM_DMA=0/1) on a few tall products through ops.linear_*; each setting in its own process."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SH = [("fwd", 65536, 512, 128), ("fwd", 65536, 128, 512), ("fwd", 65536, 128, 128), ("dgrad", 65536, 512, 128), ("dgrad", 65536, 128, 512),
      ("fwd", 23894, 512, 128), ("fwd", 65536, 64, 64), ("wgrad", 65536, 128, 128), ("wgrad", 65536, 512, 128), ("wgrad", 65536, 128, 512), ("wgrad", 23894, 128, 128),
      ("wgrad", 23894, 512, 128), ("wgrad", 65536, 64, 64), ("wgrad", 65536, 256, 64), ("wgrad", 65536, 384, 128), ("wgrad", 65536, 128, 64)]
if os.environ.get("CHILD"):
    import torch
    import robot_3dlotus_amd
    from robot_3dlotus_amd import ops
    out = {}
    for kind, M, N, K in SH:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; dy = torch.randn(M, N, device="cuda")
        b = torch.randn(N, device="cuda")
        fns = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(dy, w), "wgrad": lambda: ops.linear_wgrad(dy, x),
               "fwd+gelu": lambda: ops.linear_fwd(x, w, b, act=1, save_pre=True)}
        for name in ([kind] + (["fwd+gelu"] if kind == "fwd" else [])):
            fn = fns[name]
            for _ in range(3): fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn()
                e1.record(); e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            out[f"{name} {M} {N} {K}"] = round(best, 1)
    print("RESULT " + json.dumps(out)); sys.exit(0)
res = {}
for v in ("0", "1"):
    r = subprocess.run([sys.executable, __file__], env=dict(os.environ, CHILD="1", LOTUS_GEMM_DMA=v), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    res[v] = json.loads(line[0][7:]) if line else print(r.stderr[-800:])
for k in res["0"]: print(f"{k:28s} off {res['0'][k]:7.1f}  on {res['1'][k]:7.1f}")
