"""Debug probe: run the same training pass twice (second time on NaN-poisoned recycled memory) and compare gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from protoasnet_amd import synth
from util import CFG_VIDEO_X3D, synth_model

DEV = "cuda"
m = synth_model(CFG_VIDEO_X3D).to(DEV).train()
x = synth.echo_clips((2, 3, 4, 64, 64)).to(DEV)
g = torch.Generator().manual_seed(5)
wl, ws, wo = torch.randn(2, 3, generator=g).to(DEV), torch.randn(2, 30, generator=g).to(DEV), (torch.randn(2, 30, 1, 4, 2, 2, generator=g) * 0.1).to(DEV)

def step(poison):
    if poison:
        junk = torch.full((1 << 28,), float("nan"), device=DEV)  # 1 GiB of NaNs, then handed back to the caching allocator
        del junk
    for p in m.parameters():
        p.grad = None
    logits, sim, occ = m(x)
    ((logits * wl).sum() + (sim * ws).sum() + (occ * wo).sum()).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, logits.detach().clone()

ga, la = step(False)
gb, lb = step(True)
print("logits diff", float((la - lb).abs().max()))
rows = []
for n in ga:
    a, b = ga[n], gb[n]
    nan = bool(torch.isnan(b).any())
    rel = float((a - b).abs().max() / (a.abs().max() + 1e-12)) if not nan else float("nan")
    rows.append((rel if rel == rel else 9e9, n, nan))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"{r[1]:55s} rel {r[0]:.3e} nan={r[2]}")
