import sys, numpy as np, torch
sys.path[:0]=["/root/repo","/root/repo/tests"]
import robot_3dlotus_amd
from robot_3dlotus_amd import optim as lo
from oracle import optim as oo
g = torch.Generator().manual_seed(5)
shapes = [(1,), (3,), (4097,), (70001,), (128, 128), (64, 5, 5, 5, 7), (5,)]
wds = [0.0, 0.05, 0.05, 0.0, 0.05, 0.05, 0.05]
ps = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
opt = lo.AdamW([{"params": [p], "weight_decay": w} for p, w in zip(ps, wds)], lr=3e-4, betas=(0.9, 0.98))
ref_p = [p.detach().cpu().numpy().ravel().copy() for p in ps]
ref_m = [np.zeros_like(x) for x in ref_p]; ref_v = [np.zeros_like(x) for x in ref_p]
steps=[0]*len(ps)
for it in range(3):
    grads = [torch.randn(s, generator=g) * (3.0 if it == 1 else 0.1) for s in shapes]
    for i, p in enumerate(ps):
        p.grad = None if (i == 6 and it == 0) else grads[i].cuda()
    live = [i for i, p in enumerate(ps) if p.grad is not None]
    gl = [grads[i].numpy().ravel() for i in live]
    before=[p.detach().cpu().numpy().ravel().copy() for p in ps]
    opt.step()
    torch.cuda.synchronize()
    for k, i in enumerate(live):
        steps[i] += 1
        ref_p[i], ref_m[i], ref_v[i] = oo.adamw_step(ref_p[i], gl[k], ref_m[i], ref_v[i], steps[i], 3e-4, 0.9, 0.98, 1e-6, wds[i])
    for i, p in enumerate(ps):
        got = p.detach().cpu().numpy().ravel()
        print(it, i, shapes[i], "maxdiff", np.abs(got-ref_p[i]).max(), "moved", np.abs(got-before[i]).max(), "ptr%16", p.data_ptr()%16, p.grad is not None and p.grad.data_ptr()%16)
        vv = opt.state[p]["exp_avg_sq"].cpu().numpy().ravel()
        print("      v maxdiff", np.abs(vv-ref_v[i]).max(), "v max", np.abs(ref_v[i]).max())
