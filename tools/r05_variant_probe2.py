"""round 5 debugging aid: the fused-1x1-forward variant test's step (dropout 0.1) under several validation-build switch sets"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd"), os.path.join(ROOT, "tests")]
import torch
from variant_utils import run_on_debug_build
over = dict(densenet_structure=[3, 3], num_encoder_layers=2, dropout=float(os.environ.get("PROBE_DROPOUT", "0.1")), pixel_noise_std=0.0)
body = f"""
import test_densenet_gpu as T
from oracle import tcvn_oracle as O
cfg = O.tutorial_config(**{over!r})
batch = O.synthetic_batch([2, 1], 29, cfg)
sd = O.fill_state(cfg, 11)
n_img = int(batch[7].sum())
d_out = torch.randn(n_img, O.embed_dims(cfg)[0], generator=torch.Generator().manual_seed(8))
eng, data, grads = T._engine(cfg, sd, mode=MODE, with_grad=True)
out = torch.empty(n_img, eng.out_dim, device="cuda")
eng.forward(batch[5].cuda(), batch[6].cuda(), n_img, out, train=True, seed=1)
torch.cuda.synchronize()
raw = dict((k, eng.tap(k).double().cpu().flatten()) for k in ("raw:tabs", "raw:bstat1", "raw:bstat2", "raw:ystat1.0", "raw:ystat1.1", "raw:ystat1.2", "raw:ystat2.0", "raw:ystat2.2"))
eng.backward(d_out.cuda())
torch.cuda.synchronize()
taps = dict(("dense" + str(i + 1), eng.tap("dense" + str(i + 1)).float().cpu()) for i in range(2))
out, grads = out.cpu(), dict((k, v.cpu()) for k, v in grads.items())
result = dict(out=out, grads=grads, taps=taps, raw=raw)
"""
sets = {}
ref32 = run_on_debug_build("MODE = 0\n" + body, {})
for spec in sys.argv[1:]:
    knobs = dict(kv.split("=") for kv in spec.split(",") if kv and kv != "none")
    sets[spec] = run_on_debug_build("MODE = 1\n" + body, knobs)
isb0 = lambda k: k.endswith(("conv0.bias", "conv1.bias", "conv2.bias", "conv.bias"))
for n, a in sets.items():
    errs = sorted((((a["grads"][k] - ref32["grads"][k]).norm() / ref32["grads"][k].norm().clamp_min(1e-30)).item(), k) for k in a["grads"] if not isb0(k))
    import statistics
    print(n, "vs fp32 engine: median grad err", f"{statistics.median(e for e, _ in errs):.2e}", "worst", [(f"{e:.1e}", k[-40:]) for e, k in errs[-5:]])
names = list(sets)
isb = lambda k: k.endswith(("conv0.bias", "conv1.bias", "conv2.bias", "conv.bias"))
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = sets[names[i]], sets[names[j]]
        errs = sorted((((a["grads"][k] - b["grads"][k]).norm() / b["grads"][k].norm().clamp_min(1e-30)).item(), k) for k in a["grads"] if not isb(k))
        te = sorted(((a["taps"][k].float() - b["taps"][k].float()).abs().max().item() / b["taps"][k].float().abs().max().item(), k) for k in a["taps"])
        for rk in a["raw"]:
            d = (a["raw"][rk] - b["raw"][rk]).abs()
            bad = (d > 1e-5 * b["raw"][rk].abs().clamp_min(1e-3)).nonzero().flatten()
            print("   ", rk, "n", d.numel(), "max abs diff", f"{d.max().item():.2e}", "entries off", bad.numel(), bad[:12].tolist())
        if os.environ.get("PROBE_ALL"):
            for k in a["grads"]:
                e = ((a["grads"][k] - b["grads"][k]).norm() / b["grads"][k].norm().clamp_min(1e-30)).item()
                print(f"      {k[-60:]:60s} {e:.2e}  |b| {b['grads'][k].norm().item():.3e}")
        print(names[i], "vs", names[j], "out equal", torch.equal(a["out"], b["out"]), "taps", [(f"{e:.1e}", k) for e, k in te[-2:]], "worst grads", [(f"{e:.1e}", k[-44:]) for e, k in errs[-6:]])
