"""How long does the host need to enqueue one training step (all launches), compared with the GPU time of the step?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd")]
import bench
from transformercvn.options import Options
from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
dev = torch.device("cuda:0")
opt = Options.load(os.path.join(bench.PKG, "option_files", "tutorial_densenet_synthetic.json"))
BN = int(os.environ.get("TIME_BATCH", "32"))
opt.batch_size, opt.num_gpu, opt.hip_precision = BN, 1, "bf16"
opt.training_file = "synthetic:64:8"
model = NeutrinoFullDenseTrainer(opt).to(dev); model.train()
rt = model.network.hip_runtime(); rt.ensure_bound()
batch = bench.make_batch(BN, 8, 1234, dev)
def step():
    rt.zero_grad(); loss = model.training_step(batch, 0); loss.backward(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, until GPU done {1e3*(t2-t0):.1f} ms")
