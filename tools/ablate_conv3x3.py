"""Where do the bf16 3x3 tile kernels spend a launch?  Runs the prong-sized DenseNet forward + backward (256 maps, train mode) on the
-DTCVN_DEBUG_KNOBS build under every TCVN_DBG ablation (1 no DMA, 2 no MFMA, 4 no epilogue, 8 no eff build) in child processes and
prints the per-launch times of block 1's three 3x3 launches (forward / data gradient / weight gradient).
    python tools/ablate_conv3x3.py [n_maps]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os, json, torch
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd"), os.path.join(ROOT, "tests")]
from transformercvn.hip import _libselect
_libselect.use("libtcvn_hip_dbg.so")
from transformercvn.hip import _lib
import test_densenet_gpu as T
from oracle import tcvn_oracle as O
cfg = O.tutorial_config()
sd = O.fill_state(cfg, 1)
batch = O.synthetic_batch([NMAPS], 3, cfg)
eng, data, grads = T._engine(cfg, sd, mode=1, with_grad=True)
coords, values = batch[5].cuda(), batch[6].cuda()
out = torch.empty(NMAPS, eng.out_dim, device="cuda")
_lib.lib.tcvn_backward_overlap(0)
for it in range(2):
    if it == 1:
        _lib.lib.tcvn_profile_filter(None); _lib.lib.tcvn_profile_reset(); _lib.lib.tcvn_profile_enable(1)
    eng.forward(coords, values, NMAPS, out, train=True, seed=1)
    eng.backward(torch.ones_like(out))
    torch.cuda.synchronize()
_lib.lib.tcvn_profile_enable(0)
rec = _lib.profile_records()
res = {}
for name, ms, fl, by in rec:
    res.setdefault(name, []).append(round(ms * 1000, 1))
print("RESULT " + json.dumps(res))
"""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = {}
for dbg in (0, 4096):
    code = f"ROOT = {ROOT!r}\nNMAPS = {n}\n" + CHILD
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TCVN_DBG=str(dbg)), capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print("dbg", dbg, "failed", p.stderr[-800:]); continue
    res = json.loads(line[0][7:])
    f = res.get("k_conv3x3_fwd_bf16", [])
    d = res.get("k_conv3x3_dgrad_bf16", [])
    w = res.get("k_conv3x3_wgrad_bf16", [])
    print(f"TCVN_DBG={dbg:2d}  fwd b1 {f[:3]} b2 {f[3:5]} b3 {f[9:11]} | dgrad b1 {d[-3:]} b2 {d[-5:-3]} | wgrad b1 {w[-3:]} b2 {w[-5:-3]}", flush=True)
