"""Where does one training step spend its time?  Host enqueue time vs GPU time, and GPU time per phase (events recorded on the
stream each engine call is issued on).  python tools/phase_timeline.py [bf16|fp32]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dune-transformercvn_amd")]
import bench
from transformercvn.options import Options
from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
dev = torch.device("cuda:0")
opt = Options.load(os.path.join(bench.PKG, "option_files", "tutorial_densenet_synthetic.json"))
BN = int(os.environ.get("TIME_BATCH", "32"))
opt.batch_size, opt.num_gpu, opt.hip_precision = BN, 1, (sys.argv[1] if len(sys.argv) > 1 else "bf16")
opt.training_file = "synthetic:64:8"
model = NeutrinoFullDenseTrainer(opt).to(dev); model.train()
rt = model.network.hip_runtime(); rt.ensure_bound()
batch = bench.make_batch(BN, 8, 1234, dev)
marks = []
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def inner(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter(); e0.record(); r = fn(*a, **k); e1.record(); h1 = time.perf_counter()
        marks.append((label, e0, e1, h1 - h0)); return r
    setattr(obj, name, inner)
for eng, nm in ((rt.ev_engine, "event"), (rt.pr_engine, "prong"), (rt.head, "head")):
    wrap(eng, "forward", nm + ".fwd"); wrap(eng, "backward", nm + ".bwd")
wrap(rt.head, "loss", "head.loss")
def step():
    rt.zero_grad(); loss = model.training_step(batch, 0); loss.backward(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
for it in range(3):
    marks.clear()
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s0.record(); step(); s1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"step {it}: host enqueue {1e3*(t1-t0):.2f} ms, wall until GPU done {1e3*(t2-t0):.2f} ms, GPU (main stream) {s0.elapsed_time(s1):.2f} ms")
    for label, e0, e1, host in marks:
        print(f"   {label:11s} gpu {e0.elapsed_time(e1):7.3f} ms  (starts {s0.elapsed_time(e0):7.3f} ms into the step)  host {1e3*host:6.2f} ms")
