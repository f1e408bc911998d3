import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dune-transformercvn_amd")]
import test_densenet_gpu as T
from oracle import tcvn_oracle as O
from golden_utils import load_case, train_cfg
for name, training in [("small_b3", True), ("tutorial_b2p4", False), ("tutorial_b2p4", True), ("tutorial_b2p4", True)]:
    cfg, over, batch, g = load_case(name)
    if training:
        cfg = train_cfg(over)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    with torch.no_grad():
        ref, ctx = T._oracle_densenet(cfg, sd, batch, training)
    out, taps, _ = T._run_bf16(cfg, sd, batch, training)
    print(name, training, "out err", ((out - ref).norm() / ref.norm()).item())
    for b in range(1, len(cfg.densenet_structure) + 1):
        mine = taps[f"dense{b}"].permute(0, 3, 1, 2).double()
        r = ctx.taps[T.PFX + f":dense{b}"].double()
        d = (mine - r).abs()
        per_c = d.amax(dim=(0, 2, 3)) / r.abs().max()
        bad = (per_c > 0.05).nonzero().flatten().tolist()
        print("  dense", b, "shape", tuple(r.shape), "max rel", per_c.max().item(), "bad channels", bad[:40], flush=True)
        if bad:
            c = bad[0]
            idx = (d[:, c] > 0.05 * r.abs().max()).nonzero()
            print("    first bad positions (n,h,w) of channel", c, idx[:10].tolist(), "count", len(idx))
            n, h, w = idx[0].tolist()
            print("    mine", mine[n, c, h, w].item(), "ref", r[n, c, h, w].item())
