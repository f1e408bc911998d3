"""A/B timing on the VALIDATION build (libtcvn_hip_dbg.so honours the TCVN_* switches in the environment): the config-2 bf16 step of
bench.py, N timed steps + one serialised survey step with per-kernel event timings.  Not a benchmark of the product library.
    TCVN_XA_ONTHEFLY=1 python tools/time_dbg.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd")]
from transformercvn.hip import _libselect
_libselect.use(os.environ.get("TIME_LIB", "libtcvn_hip_dbg.so"))      # TIME_LIB: e.g. a kept copy of an earlier validation build (same-box A/B)
import torch
import bench
from transformercvn.options import Options
from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
from transformercvn.hip import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
batch_n = int(os.environ.get("TIME_BATCH", "32"))
dev = torch.device("cuda", 0)
opt = Options.load(os.path.join(ROOT, "dune-transformercvn_amd", "option_files", "tutorial_densenet_synthetic.json"))
opt.batch_size, opt.num_gpu, opt.hip_precision, opt.seed = batch_n, 1, os.environ.get("TIME_PRECISION", "bf16"), 1234
opt.training_file = "synthetic:64:8"
torch.manual_seed(0)
model = NeutrinoFullDenseTrainer(opt).to(dev)
model.train()
rt = model.network.hip_runtime()
rt.ensure_bound()
batch = bench.make_batch(batch_n, 8, 1234, dev)


def step():
    rt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
rt.overlap_embedders = False
_lib.lib.tcvn_backward_overlap(0)
_lib.lib.tcvn_profile_filter(None); _lib.lib.tcvn_profile_reset(); _lib.lib.tcvn_profile_enable(1)
step(); torch.cuda.synchronize()
_lib.lib.tcvn_profile_enable(0)
agg = {}
for name, m, fl, by in _lib.profile_records():
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += m
print(json.dumps({"knobs": {k: v for k, v in os.environ.items() if k.startswith("TCVN_")}, "ms_per_step": round(ms, 3), "loss": float(loss),
                  "survey": {k: [a[0], round(a[1], 3)] for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])}}))
