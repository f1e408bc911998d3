"""Wave-cycles per phase of k_conv3x3_wgrad_bf16 (validation build: clock64 markers summed over the waves of every 16th workgroup).
One dense layer on 256 maps of 99x69 (dense block 1's 3x3 weight gradient): python tools/wgrad_phases.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd"), os.path.join(ROOT, "tests")]
from transformercvn.hip import _libselect
_libselect.use("libtcvn_hip_dbg.so")
import torch
from transformercvn.hip import _lib
import test_densenet_gpu as T
from oracle import tcvn_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = O.tutorial_config(densenet_structure=[1])
sd = O.fill_state(cfg, 1)
batch = O.synthetic_batch([n], 3, cfg)
eng, data, grads = T._engine(cfg, sd, mode=1, with_grad=True)
coords, values = batch[5].cuda(), batch[6].cuda()
out = torch.empty(n, eng.out_dim, device="cuda")
_lib.lib.tcvn_backward_overlap(0)
fn = _lib.lib.tcvn_debug_wgrad_phases
fn.restype = None
buf = (ctypes.c_ulonglong * 16)()
for it in range(2):
    if it == 1:
        _lib.lib.tcvn_profile_filter(None); _lib.lib.tcvn_profile_reset(); _lib.lib.tcvn_profile_enable(1)
    eng.forward(coords, values, n, out, train=True, seed=1)
    eng.backward(torch.ones_like(out))
    torch.cuda.synchronize()
    fn(buf, 1)
_lib.lib.tcvn_profile_enable(0)
for name, ms, fl, by in _lib.profile_records():
    if "wgrad" in name or "dgrad" in name:
        print(f"{name:32s} {ms * 1000:8.1f} us")
v = list(buf)
tiles = (n * 101 * 71 + 127) // 128
waves = 4 * tiles / 16       # wave-tiles that report (every 16th workgroup)
names = {0: "M k loop (tr reads + 72 MFMA)", 1: "M barrier", 8: "H dma issue (17 DMA)", 9: "H eff loads + store (hash)",
         10: "H table fill", 11: "H wait vmcnt(0) (act_fused)", 13: "H in-LDS activation (act_fused)", 12: "H barrier"}
for k in sorted(names):
    print(f"{names[k]:36s} {v[k] / 1e6:10.1f} Mcycles   {v[k] / waves:8.0f} cycles per wave and tile")
print("tiles", tiles, "sum M", sum(v[:8]) / waves, "sum H", sum(v[8:]) / waves)
