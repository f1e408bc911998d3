"""Wave-cycles per phase of k_conv3x3_fwd_pair_bf16 (validation build: clock64 markers summed over the waves of each role).
One dense layer on 256 maps of 99x69 (dense block 1's 3x3 launch): python tools/pair_phases.py"""
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
eng, data, grads = T._engine(cfg, sd, mode=1, with_grad=False)
coords, values = batch[5].cuda(), batch[6].cuda()
out = torch.empty(n, eng.out_dim, device="cuda")
fn = _lib.lib.tcvn_debug_pair_phases
fn.restype = None
buf = (ctypes.c_ulonglong * 16)()
for it in range(2):
    eng.forward(coords, values, n, out, train=True, seed=1)
    torch.cuda.synchronize()
    fn(buf, 1)
v = list(buf)
tiles = (n * 101 * 71 + 127) // 128
waves = 4 * tiles / 16       # wave-tiles per role that report (every 16th workgroup)
names = {0: "A wait vmcnt", 1: "A barrier", 2: "A dma issue", 3: "A mfma + xchg store", 4: "A table fill",
         8: "B wait vmcnt", 9: "B barrier", 10: "B dma issue", 11: "B epilogue", 12: "B mfma"}
for k in sorted(names):
    print(f"{names[k]:22s} {v[k] / 1e6:10.1f} Mcycles   {v[k] / waves:8.0f} cycles per wave and tile")
print("tiles", tiles, "sum A", sum(v[:8]) / waves, "sum B", sum(v[8:]) / waves)
