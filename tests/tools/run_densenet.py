"""Run the DenseNet engine alone (profiling aid): python tests/tools/run_densenet.py [n_img] [bf16|fp32] [fwd|fwdbwd] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd"), os.path.join(ROOT, "tests")]
import torch
from oracle import tcvn_oracle as O
from transformercvn.hip.engine import DenseNetEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = 1 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else 0
bwd = len(sys.argv) > 3 and sys.argv[3] == "fwdbwd"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = O.tutorial_config()
PFX = "network.prong_embedding.prong_pixel_embedding"
sd = O.fill_state(cfg, 1)
eng = DenseNetEngine(3, 256, 64, 32, 4, [3, 6, 12, 6, 3], 400, 280, 0.1, mode)
data = {k[len(PFX) + 1:]: v.cuda().contiguous() for k, v in sd.items() if k.startswith(PFX + ".") and v.is_floating_point()}
grads = {k: torch.zeros_like(v) for k, v in data.items()}
eng.bind(data, grads)
batch = O.synthetic_batch([n], 3, cfg)
coords, values = batch[5].cuda(), batch[6].cuda()
out = torch.empty(n, 256, device="cuda")
d_out = torch.randn(n, 256, device="cuda")
for _ in range(iters):
    eng.forward(coords, values, n, out, train=True, seed=1)
    if bwd:
        eng.backward(d_out)
torch.cuda.synchronize()
print("ok", out.float().abs().mean().item())
if os.environ.get("TCVN_PROFILE"):
    from transformercvn.hip import _lib
    _lib.lib.tcvn_profile_reset(); _lib.lib.tcvn_profile_enable(1)
    eng.forward(coords, values, n, out, train=True, seed=1)
    if bwd:
        eng.backward(d_out)
    torch.cuda.synchronize(); _lib.lib.tcvn_profile_enable(0)
    agg = {}
    for name, ms, fl, by in _lib.profile_records():
        a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
    for k, a in sorted(agg.items(), key=lambda x: -x[1][1]):
        print(f"{k:34s} {a[0]:4d} {a[1]:8.3f} ms {a[2]/a[1]/1e9:8.1f} TF/s")
