"""Run the same train-mode bf16 DenseNet forward(+backward) repeatedly; every repetition must be bit-identical."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "dune-transformercvn_amd")]
import test_densenet_gpu as T
from oracle import tcvn_oracle as O
from golden_utils import load_case, train_cfg
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg, over, batch, g = load_case("tutorial_b2p4")
cfg = train_cfg(over)
sd = O.fill_state(cfg, int(g["weight_seed"]))
n_img = int(batch[7].sum())
d_out = torch.randn(n_img, O.embed_dims(cfg)[0], generator=torch.Generator().manual_seed(5)).cuda()
eng, data, grads = T._engine(cfg, sd, mode=1, with_grad=True)
coords, values = batch[5].cuda(), batch[6].cuda()
out = torch.empty(n_img, eng.out_dim, device="cuda")
nb = len(cfg.densenet_structure)
def analyse(first, cur):
    """Fit (wrong - right) of layer 0's 3x3 output at the bad positions with per-(tap, k-step) partial sums."""
    import torch.nn.functional as F
    good = first["dense1"][..., 64:96].float()
    bad = cur["dense1"][..., 64:96].float()
    px = ((good - bad).abs().amax(dim=-1) > 0).nonzero()
    if len(px) == 0 or len(px) > 600:
        print("   analyse: bad positions", len(px)); return
    ya = first["ya1.0"].float()                                  # [n,h,w,128] (identical in both reps)
    same_in = torch.equal(first["ya1.0"].view(torch.int16), cur["ya1.0"].view(torch.int16))
    w = sd[T.PFX + ".features.dense1.layers.0.output_block.conv2.weight"].cuda().to(torch.bfloat16).float()   # [32,128,3,3]
    n, H, W, C = ya.shape
    yap = F.pad(ya, (0, 0, 1, 1, 1, 1))                          # pad h and w
    cols = []
    for (b, y, x) in px.tolist():
        cols.append(yap[b, y:y + 3, x:x + 3, :])                  # [3,3,128]
    patch = torch.stack(cols)                                    # [P,3,3,128]
    # contributions [P, 9 taps, 8 ksteps, 32 out]
    contrib = torch.einsum("pyxkc,nkcyx->pyxkn", patch.view(-1, 3, 3, 8, 16), w.view(32, 8, 16, 3, 3)).reshape(len(px), 72, 32)
    delta = torch.stack([bad[b, y, x] - good[b, y, x] for (b, y, x) in px.tolist()])     # [P,32]
    total = contrib.sum(1)
    gpos = ((px[:, 0] * (H + 2) + px[:, 1] + 1) * (W + 2) + px[:, 2] + 1)
    print("   tiles", sorted(set((gpos // 128).tolist())), "waves", sorted(set(((gpos % 128) // 32).tolist())))
    print("   analyse: positions", len(px), "input identical", same_in, "|delta|/|out|", (delta.norm() / total.norm()).item())
    # does a wrong vector equal the right output of some OTHER pixel?
    allgood = good.reshape(-1, 32)
    flat = (px[:, 0] * H + px[:, 1]) * W + px[:, 2]
    for i in range(0, len(px), max(1, len(px) // 6)):
        dist = (allgood - bad[tuple(px[i].tolist())]).abs().amax(dim=1)
        j = int(dist.argmin())
        print("   pos", int(flat[i]), "nearest right vector at", j, "max-abs distance", float(dist[j]), "own distance", float(dist[int(flat[i])]),
              "bad", [round(float(x), 2) for x in bad[tuple(px[i].tolist())][:6]], "good", [round(float(x), 2) for x in good[tuple(px[i].tolist())][:6]])
    A = contrib.permute(0, 2, 1).reshape(-1, 72)                 # [(P*32), 72]
    sol = torch.linalg.lstsq(A, delta.reshape(-1, 1)).solution.flatten()
    res = (A @ sol - delta.reshape(-1)).norm() / delta.norm()
    print("   lstsq residual", res.item(), "coefficients (tap-major, 8 k-steps each):")
    # is the wrong vector the right weights applied to some OTHER position's input (image/table mismatch)?
    wt = w.permute(2, 3, 1, 0).reshape(9 * 128, 32)
    b = sd[T.PFX + ".features.dense1.layers.0.output_block.conv2.bias"].cuda().float()
    full = F.conv2d(ya.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, 32)     # fp32 recompute, all pixels
    print("   recompute vs right", float((full - allgood).abs().max()))
    for i in range(0, len(px), max(1, len(px) // 6)):
        dist = (full - bad[tuple(px[i].tolist())]).abs().amax(dim=1)
        j = int(dist.argmin())
        print("   pos", int(flat[i]), "nearest recomputed vector at", j, "distance", float(dist[j]))


first = None
nbad = 0
worst_grad = [0.0, ""]
for it in range(reps):
    for gr in grads.values():
        gr.zero_()
    if os.environ.get("DBG_TRASH") and eng._ws is not None:        # stale-read detector: nothing in the workspace may survive from the previous step
        eng._ws.random_(0, 256)
    eng.forward(coords, values, n_img, out, train=True, seed=1)
    taps = {"conv0": eng.tap("conv0").clone(), "raw:wk": eng.tap("raw:wk").clone(), "raw:tabs": eng.tap("raw:tabs").clone(),
            "raw:bstat1": eng.tap("raw:bstat1").clone()}
    for l in range(cfg.densenet_structure[0]):
        for nm in (("xa", "bottleneck", "ya") if not os.environ.get("TCVN_XA_ONTHEFLY") else ("bottleneck", "ya")):
            taps[f"{nm}1.{l}"] = eng.tap(f"{nm}1.{l}").clone()
    taps.update({f"dense{i + 1}": eng.tap(f"dense{i + 1}").clone() for i in range(nb)})
    eng.backward(d_out)
    torch.cuda.synchronize()
    import ctypes, numpy as np
    from transformercvn.hip import _lib as L_
    rec = None
    if hasattr(L_.lib, "tcvn_debug_fwd_forensic"):
        buf = np.zeros(512 * 4 * 16, dtype=np.uint32)
        L_.lib.tcvn_debug_fwd_forensic(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
        rec = buf.reshape(512, 4, 16)[:256]
        if it == 0:
            rec0 = rec.copy()
        else:
            for name, col in (("weights", 6), ("lds", 7), ("acc", 8), ("wfrag", 9), ("n_off", 10), ("out", 11), ("tile", 12)):
                bad = np.argwhere(rec[:, :, col] != rec0[:, :, col])
                if len(bad):
                    print("rep", it, "forensic:", name, "hash differs for (block, wave):", bad[:8].tolist(), flush=True)
                    for b_, w_ in bad[:4]:
                        r_ = rec[b_, w_]
                        print("    hw_id %06x xcc %d cycles %d tile %d lb %d" % (r_[0], r_[1], (int(r_[5]) << 32 | int(r_[4])) - (int(r_[3]) << 32 | int(r_[2])), r_[12], r_[13]),
                              "same-block waves cycles", [((int(x[5]) << 32 | int(x[4])) - (int(x[3]) << 32 | int(x[2]))) for x in rec[b_]],
                              "start offsets", [int(x[2]) - int(rec[b_, 0, 2]) for x in rec[b_]])
    cur = dict(taps, out=out.clone(), **{"g:" + k: v.clone() for k, v in grads.items()})
    if first is None:
        first = cur
        continue
    for k, v in cur.items():
        if not torch.equal(v.view(torch.int16 if v.dtype == torch.bfloat16 else v.dtype), first[k].view(torch.int16 if v.dtype == torch.bfloat16 else v.dtype)):
            d = (v.float() - first[k].float()).abs()
            idx = (d > 0).nonzero()
            if k.startswith("g:") or k.startswith("raw:") or not (k == "dense1" or k == "xa1.1"):
                nbad += 1
                if k.startswith("g:"):              # atomics reorder fp32 sums (~1e-6); anything larger would be a race
                    rel = (d.max() / first[k].float().abs().max().clamp_min(1e-20)).item()
                    if rel > worst_grad[0]:
                        worst_grad[0], worst_grad[1] = rel, f"{k} rep {it}"
                continue
            print("rep", it, "tensor", k, "shape", tuple(v.shape), "differs in", len(idx), "elements; first", idx[:6].tolist(),
                  "max diff", d.max().item(), flush=True)
            nbad += 1
            if k == "dense1":
                analyse(first, cur)
            if False:
                for c0 in range(0, v.shape[-1], 32):
                    dd = d[..., c0:c0 + 32]
                    px = (dd.amax(dim=-1) > 0).nonzero()
                    print("   channels", c0, "count", int((dd > 0).sum()), "pixels", len(px), "first", px[:4].tolist(), "last", px[-2:].tolist(),
                          "max", dd.max().item())
                    if 0 < len(px) < 4096:
                        flat = (px[:, 0] * v.shape[1] + px[:, 1]) * v.shape[2] + px[:, 2]
                        print("   flat pixel ids", flat[:64].tolist())
            if k.startswith("dense2") or k == "out":
                break
print("repetitions", reps, "mismatching tensors", nbad, "worst gradient run-to-run difference (max-norm, relative)", worst_grad)
