"""round 5 debugging aid: the fused-1x1-backward variant test's step under several validation-build switch sets; prints pairwise gradient differences"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd"), os.path.join(ROOT, "tests")]
import torch
from variant_utils import run_on_debug_build
over = dict(densenet_structure=[3, 3], num_encoder_layers=2, dropout=0.0, pixel_noise_std=0.0)
body = f"""
import test_densenet_gpu as T
from oracle import tcvn_oracle as O
cfg = O.tutorial_config(**{over!r})
batch = O.synthetic_batch([2, 1], 23, cfg)
sd = O.fill_state(cfg, 9)
n_img = int(batch[7].sum())
d_out = torch.randn(n_img, O.embed_dims(cfg)[0], generator=torch.Generator().manual_seed(6))
out, taps, grads = T._run_bf16(cfg, sd, batch, True, d_out)
result = dict(out=out, grads=grads)
"""
sets = {}
for spec in sys.argv[1:]:
    knobs = dict(kv.split("=") for kv in spec.split(",") if kv and kv != "none")
    sets[spec] = run_on_debug_build(body, knobs)
names = list(sets)
isb = lambda k: k.endswith(("conv0.bias", "conv1.bias", "conv2.bias", "conv.bias"))
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = sets[names[i]], sets[names[j]]
        errs = sorted((((a["grads"][k] - b["grads"][k]).norm() / b["grads"][k].norm().clamp_min(1e-30)).item(), k) for k in a["grads"] if not isb(k))
        print(names[i], "vs", names[j], "out equal", torch.equal(a["out"], b["out"]), "worst", [(f"{e:.1e}", k[-40:]) for e, k in errs[-4:]])
