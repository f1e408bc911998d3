#!/usr/bin/env python
"""Headline benchmark: events/s of one forward+loss+backward of the TransformerCVN DenseNet model (BASELINE.json
config 2: batch 32, 8 prongs/event, 3x400x280 maps) on N MI355X GPUs, one process per GPU, RCCL gradient all-reduce
overlapped with backward.  Prints ONE JSON line (rank 0).

    python bench.py [--gpus 1 --steps 5 --warmup 2 --precision bf16]
    python bench.py --gpus N ...            # starts its N ranks itself (one child process per GPU, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus 8 --global-batch 64      # strong scaling: the global batch is split over the ranks
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "dune-transformercvn_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_IMAGE_FWD_BWD = 14.385e9      # conv+linear MACs x2 of one DenseNet pass fwd+bwd
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA TFLOP/s, MI355X_MICROARCH.md


T0 = time.perf_counter()


def note(msg):
    """progress line on stderr (rank 0 only prints the JSON on stdout)"""
    print(f"[bench +{time.perf_counter() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def launch_ranks(n):
    """`--gpus N` without a launcher around us (reference: train.py:123-127 starts its ranks from num_gpu): this process -- which has
    made NO GPU call yet -- starts N children of this same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, lets
    rank 0 print the JSON line on the shared stdout, and exits with the worst child's code.  No process that touched the GPU is
    ever re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    pending = list(procs)
    while pending:
        for pr in list(pending):
            rc = pr.poll()
            if rc is None:
                continue
            pending.remove(pr)
            if rc != 0:
                worst = worst or rc
                for other in pending:              # one rank failed: the others would wait in a collective forever
                    other.terminate()
        time.sleep(0.2)
    return worst


def host_threads():
    """CPU threads this process may actually use (all cores it has): the affinity mask, limited by the cgroup
    CPU quota when there is one (a GPU box hands each lease a share of the host, e.g. 16 of 256 cores: running 256 threads on
    a 16-core quota is several times slower than 16 threads).  Without a readable quota the share the harness documents (16)
    caps it."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    quota = int(parts[0]) / int(parts[1])
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        quota = q / int(f2.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    else:
        n = min(n, 16)
    return max(1, n)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def sdxl_flops_per_image(init_ch=64, out_dim=256, H=400, W=280, in_ch=3):
    """conv MACs x2 of one SDXL-embedder pass (layers/sdxl_net.py:19-34 schedule), forward and forward+backward (3x forward minus
    the data gradient of conv_in, which has no input gradient)."""
    chans = [init_ch, init_ch, 2 * init_ch, 2 * init_ch, 4 * init_ch, 4 * init_ch, 8 * init_ch, 8 * init_ch, out_dim]
    fwd = 2.0 * H * W * 9 * in_ch * chans[0]
    conv_in = fwd
    cin, h, w = chans[0], H, W
    for i, c in enumerate(chans):
        fwd += 2.0 * h * w * 9 * (cin * c + c * c) + 2.0 * h * w * 9 * 2 * c * c            # resnet 0 + resnet 1
        if cin != c:
            fwd += 2.0 * h * w * cin * c
        if i + 1 != len(chans):
            h, w = (h - 2) // 2 + 1, (w - 2) // 2 + 1
            fwd += 2.0 * h * w * 9 * c * c
        cin = c
    c = chans[-1]
    fwd += 2.0 * h * w * (4 * 9 * c * c + 2 * c * c) + 2.0 * h * w * 9 * c * out_dim + 2.0 * out_dim * out_dim
    return fwd, 3.0 * fwd - conv_in


def make_batch(batch, prongs, seed, device):
    """`prongs`: int (fixed) or (lo, hi) for ragged events"""
    from transformercvn.dataset.minkowski_dataset import SyntheticDataset, MinkowskiCollection
    ds = SyntheticDataset(batch, prongs, seed=seed)
    b = MinkowskiCollection()([ds[i] for i in range(batch)])
    n_prongs = int(b[7].sum())
    width = int(b[7].sum(1).max())
    dev = tuple(t.to(device) for t in b)
    return dev + ((width, n_prongs),)


def cpu_baseline(threads, prongs):
    """The oracle (CPU restatement of the reference, oracle/tcvn_oracle.py; parity-pinned to the reference's goldens) timed on
    the host cores in this same run: the GPU workload's model (6-layer encoder, dropout 0.1, `prongs` prongs/event) on a
    bounded sample of 4 events (36 maps), fp32, forward+loss+backward.  A baseline for context, not the target."""
    from oracle import tcvn_oracle as O
    torch.set_num_threads(threads)
    cfg = O.tutorial_config()
    sd = O.fill_state(cfg, 1)
    ev = 4
    batch = O.synthetic_batch([prongs] * ev, 11, cfg)
    t0 = time.perf_counter()
    O.train_step(sd, cfg, batch, apply_dropout=True)               # warm-up (oneDNN primitive creation)
    warm = time.perf_counter() - t0
    ts = []
    for _ in range(3 if warm < 12 else 1):                         # bounded: about 10-30 s of CPU work in all
        t0 = time.perf_counter()
        O.train_step(sd, cfg, batch, apply_dropout=True)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    note(f"cpu baseline: {cpu_model()}, os.cpu_count()={os.cpu_count()}, threads used {threads}, torch threads "
         f"{torch.get_num_threads()}, warm-up {warm:.2f} s, steps {[round(t, 2) for t in ts]} s")
    # SURVEY.md 8(d)'s config-1 shape beside it (the reference's own CPU-runnable case): B = 2, 4 prongs, 2-layer encoder
    cfg1 = O.tutorial_config(num_encoder_layers=2)
    sd1 = O.fill_state(cfg1, 1)
    b1 = O.synthetic_batch([4, 4], 11, cfg1)
    O.train_step(sd1, cfg1, b1, apply_dropout=True)
    t0 = time.perf_counter()
    O.train_step(sd1, cfg1, b1, apply_dropout=True)
    c1 = 2 / (time.perf_counter() - t0)
    note(f"cpu baseline, config-1 shape (2 events x 4 prongs, 2-layer encoder): {c1:.2f} events/s")
    return {"value": round(ev / ts[len(ts) // 2], 3), "unit": "events/s", "cores": threads, "kind": "port",
            "config1_shape_value": round(c1, 3),
            "config1_shape_sample": "oracle fp32 fwd+loss+bwd, 2 events x 4 prongs (10 maps), 2-layer encoder, 1 step after 1 warm-up",
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "sample": f"oracle fp32 fwd+loss+bwd, {ev} events x {prongs} prongs ({ev * (1 + prongs)} maps), 6-layer encoder, dropout 0.1, "
                      f"median of {len(ts)} steps after 1 warm-up"}


def fp32_parity_mode_ms(opt_path, args, dev, batch):
    """ms/step of the fp32 parity mode (the mode that meets the 1e-3 logit gate) on the same batch: 1 warm-up + 2 timed steps."""
    from transformercvn.options import Options
    from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
    opt = Options.load(opt_path)
    opt.batch_size, opt.num_gpu, opt.hip_precision, opt.seed = args.batch, 1, "fp32", 1234
    opt.training_file = f"synthetic:64:{args.prongs}"
    torch.manual_seed(0)
    model = NeutrinoFullDenseTrainer(opt).to(dev)
    model.train()
    rt = model.network.hip_runtime()

    def step():
        rt.zero_grad()
        model.training_step(batch, 0).backward()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / 2


def small_batch_ms(opt_path, args, dev, batch_n, steps=5):
    """ms/step of the same bf16 fwd+loss+bwd step at `batch_n` events per GPU (8 = one rank's share of the north star's batch-64
    strong-scaling case at 8 GPUs; 64 = that whole batch on one GPU): 2 warm-up + `steps` timed steps."""
    from transformercvn.options import Options
    from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
    opt = Options.load(opt_path)
    opt.batch_size, opt.num_gpu, opt.hip_precision, opt.seed = batch_n, 1, "bf16", 1234
    opt.training_file = f"synthetic:64:{args.prongs}"
    torch.manual_seed(0)
    model = NeutrinoFullDenseTrainer(opt).to(dev)
    model.train()
    rt = model.network.hip_runtime()
    batch = make_batch(batch_n, args.prongs, 1234, dev)

    def step():
        rt.zero_grad()
        model.training_step(batch, 0).backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / steps


def sdxl_mode_ms(opt_path, args, dev):
    """ms/step of BASELINE config 4 (--sdxl embedder, batch 16 x 8 prongs, bf16; parity unpinned: diffusers is not available) so that the
    default run puts it on the driver's record: 1 warm-up + 2 timed steps."""
    from transformercvn.options import Options
    from transformercvn.network.trainers.neutrino_full_sdxl_trainer import NeutrinoFullSDXLTrainer
    opt = Options.load(opt_path)
    opt.batch_size, opt.num_gpu, opt.hip_precision, opt.seed = 16, 1, "bf16", 1234
    opt.training_file = "synthetic:64:8"
    torch.manual_seed(0)
    model = NeutrinoFullSDXLTrainer(opt).to(dev)
    model.train()
    rt = model.network.hip_runtime()
    batch = make_batch(16, 8, 1234, dev)

    def step():
        rt.zero_grad()
        model.training_step(batch, 0).backward()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / 2


def pmc_traffic(label):
    """HBM bytes per launch of `label` from the committed rocprofv3 PMC passes of this command (tools/pmc_traffic.py): the
    newest round's file that has the kernel, or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f).get(label)
            if rec:
                return round(rec["traffic_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--sdxl", action="store_true", help="BASELINE config 4: the SDXL-style embedder (train.py --sdxl), batch 16")
    ap.add_argument("--ragged-inference", action="store_true",
                    help="BASELINE config 5: eval-mode forward only, 1..16 prongs per event (packed attention mask)")
    ap.add_argument("--batch", type=int, default=None, help="events per GPU (weak scaling; default 32, 16 with --sdxl)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling: total events per step, split evenly over the ranks (north star: 64 at 8 GPUs)")
    ap.add_argument("--no-sdxl", action="store_true", help="skip the --sdxl (config 4) timing of the default run (sdxl_ms_per_step)")
    ap.add_argument("--no-batch8", action="store_true", help="skip the small-batch timings of the default run (batch8_ms_per_step, batch64_ms_per_step)")
    ap.add_argument("--prongs", type=int, default=8)
    ap.add_argument("--dropout", type=float, default=None, help="override options.dropout (experiments only; the metric uses the file's 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--side-priority", type=int, default=None, help="A/B: priority of the event-embedder side stream (-1 = high)")
    ap.add_argument("--no-bwd-overlap", action="store_true", help="keep the weight-gradient kernels on the main stream (tcvn_backward_overlap(0): the library's default since round 4)")
    ap.add_argument("--bwd-overlap", action="store_true", help="A/B: 3x3 weight gradients on the plan's side stream (tcvn_backward_overlap(1), the default of rounds 2-3)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 parity-mode timing (fp32_ms_per_step)")
    ap.add_argument("--no-optimizer-leg", action="store_true", help="skip the with_optimizer_ms_per_step leg (5 extra steps): counter passes that are divided by the steps in the run")
    ap.add_argument("--dump-records", default="", help="write every profiled launch (name, ms, flops) to this JSON file")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # no launcher around us: start the ranks (no GPU call so far)
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line's n_gpus would not be what was asked for")
    strong = args.global_batch is not None
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of {world} ranks")
        args.batch = args.global_batch // world
    if args.batch is None:
        args.batch = 16 if args.sdxl else 32
    # rehearsal on a one-GPU box (control flow of the multi-rank path only): TCVN_BENCH_REHEARSAL=1 puts every rank on
    # cuda:0 and exchanges over gloo; the real run is one rank per GPU over RCCL ("nccl")
    rehearsal = os.environ.get("TCVN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from transformercvn.options import Options
    from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
    if args.sdxl:
        from transformercvn.network.trainers.neutrino_full_sdxl_trainer import NeutrinoFullSDXLTrainer as NeutrinoFullDenseTrainer
    from transformercvn.hip import _lib
    from transformercvn.hip.distributed import broadcast_buffers
    if os.path.basename(_lib.LIB_PATH) != "libtcvn_hip.so":
        raise SystemExit(f"bench.py measures the product library only, not {_lib.LIB_PATH}")

    opt_path = os.path.join(PKG, "option_files", "tutorial_densenet_synthetic.json")
    opt = Options.load(opt_path)
    opt.batch_size, opt.num_gpu, opt.hip_precision, opt.seed = args.batch, world, args.precision, 1234 + rank
    opt.training_file = f"synthetic:64:{args.prongs}"
    if args.dropout is not None:
        opt.dropout = args.dropout
    torch.manual_seed(rank)              # every rank draws its own initial weights; enable_data_parallel() broadcasts rank 0's (as DDP does
    model = NeutrinoFullDenseTrainer(opt).to(dev)      # for the reference, whose train.py sets no seed)
    model.train()
    rt = model.network.hip_runtime()
    if args.side_priority is not None:
        rt.side_priority = args.side_priority
    rt.ensure_bound()
    _lib.lib.tcvn_backward_overlap(1 if args.bwd_overlap and not args.no_bwd_overlap else 0)
    batch = make_batch(args.batch, (1, 16) if args.ragged_inference else args.prongs, 1234 + rank, dev)
    if args.ragged_inference:
        model.eval()
    reducer = model.enable_data_parallel() if world > 1 else None    # state broadcast from rank 0 + overlapped arena all-reduce hooks

    def step():
        if args.ragged_inference:                 # inference only: no loss, no backward, no exchange (embarrassingly parallel)
            with torch.no_grad():
                return model.shared_step(batch)[2].sum()
        if reducer:
            broadcast_buffers(rt.flat_buf)
        rt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        if reducer:
            reducer.finish()
        return loss

    note(f"model + batch ready (rank {rank}/{world}, {args.precision}); warm-up ...")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # Untimed survey step: every convolution-class launch bracketed by HIP events on its own stream, embedders serialised on
    # one stream so that the per-kernel times are not smeared by the overlap.  It names the dominant kernel.  Every rank runs
    # the same two extra steps (they contain collectives); only rank 0 records.
    agg, kernels, top, biggest = {}, None, None, {}
    profiled = rank == 0 and not args.no_profile
    if not args.no_profile:
        rt.overlap_embedders = False
        _lib.lib.tcvn_backward_overlap(0)          # serial per-kernel timings for the survey
        if profiled:
            _lib.lib.tcvn_profile_filter(None)
            _lib.lib.tcvn_profile_reset()
            _lib.lib.tcvn_profile_enable(1)
        step()
        torch.cuda.synchronize()
        rt.overlap_embedders = True
        _lib.lib.tcvn_backward_overlap(1 if args.bwd_overlap and not args.no_bwd_overlap else 0)
        if profiled:
            _lib.lib.tcvn_profile_enable(0)
            records = _lib.profile_records()
            if args.dump_records:
                with open(args.dump_records, "w") as f:
                    json.dump(records, f)
            biggest = {}                    # per label: the launch with the most algorithmic bytes (dense block 1 of the prong embedder)
            for name, ms, fl, by in records:
                a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += ms; a[2] += fl; a[3] += by
                if by > biggest.get(name, (0.0, 0.0, 0.0))[0]:
                    biggest[name] = (by, ms, fl)
            _lib.lib.tcvn_profile_reset()
            kernels = sorted(({"kernel": k, "launches": a[0], "ms": round(a[1], 3), "avg_ms": round(a[1] / a[0], 4),
                               "tflops": round(a[2] / a[1] / 1e9, 2) if a[1] > 0 else 0.0,
                               "gbps": round(a[3] / a[1] / 1e6, 1) if a[1] > 0 else 0.0} for k, a in agg.items()),
                             key=lambda r: -r["ms"])
            top = kernels[0]["kernel"]
            # the timed region below records the dominant kernel only (two events per launch of that one kernel)
            _lib.lib.tcvn_profile_filter(top.encode())
            _lib.lib.tcvn_profile_enable(1)
        step()                                       # settle after the serialised survey step
        torch.cuda.synchronize()
        if profiled:
            _lib.lib.tcvn_profile_reset()

    note("timed steps ...")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    lossv = float(loss.detach())

    note(f"timed {args.steps} steps in {elapsed:.3f}s")
    roof = None
    if profiled:
        _lib.lib.tcvn_profile_enable(0)
        recs = [r for r in _lib.profile_records() if r[0] == top]
        _lib.lib.tcvn_profile_filter(None)
        _lib.lib.tcvn_profile_reset()
        n = len(recs)
        ms = sum(r[1] for r in recs)
        fl = sum(r[2] for r in recs)
        by = sum(r[3] for r in recs)
        # roofline bound of the dominant kernel from its arithmetic intensity against the machine balance
        intensity = fl / by if by > 0 else float("inf")
        hbm_bound = intensity < PEAK[args.precision] * 1e12 / (HBM_PEAK_GBPS * 1e9)
        # `achieved` uses the SERIALISED average launch duration (survey step: the two embedders on one stream, which is also what
        # rocprofv3's kernel trace shows and what profiles/ holds); inside the timed region the event embedder shares the CUs with
        # the prong embedder, so the same kernel's launches read longer there (reported as overlapped_avg_launch_ms).
        sn, sms, sfl, sby = agg[top][0], agg[top][1], agg[top][2], agg[top][3]
        if hbm_bound:
            ach, peak, unit = sby / sms / 1e6, HBM_PEAK_GBPS, "GB/s"
        else:
            ach, peak, unit = sfl / sms / 1e9, PEAK[args.precision], "TFLOP/s"
        roof = {"bound": "hbm" if hbm_bound else "mfma", "kernel": top, "launches_per_step": sn,
                "avg_launch_ms": round(sms / sn, 4), "flop_per_launch": sfl / sn, "bytes_per_launch": sby / sn,
                "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": pmc_traffic(top),
                "measured": f"HIP events on the launch stream around all {sn} launches of one step, embedders serialised",
                "bytes_definition": "SURVEY 8(d) strict: every operand read once, every result written once (the read of an "
                                    "accumulated-into gradient buffer is not counted)",
                # the kernel's largest launch of the survey step (most algorithmic bytes: dense block 1 of the prong embedder), where launch
                # overheads and the small maps of the deep blocks do not dilute the figure; `frac` above stays the all-launch average
                "largest_launch": ({"bytes": biggest[top][0], "ms": round(biggest[top][1], 4),
                                    "achieved": round(biggest[top][0] / biggest[top][1] / 1e6, 1), "unit": "GB/s",
                                    "frac": round(biggest[top][0] / biggest[top][1] / 1e6 / HBM_PEAK_GBPS, 4)}
                                   if hbm_bound and top in biggest and biggest[top][1] > 0 else None),
                "overlapped_avg_launch_ms": round(ms / n, 4) if n else None,    # same kernel inside the timed (two-stream) region
                "survey_ms_per_step": round(sum(a[1] for a in agg.values()), 2)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        events = world * args.batch * args.steps
        value = events / elapsed
        flop_img = FLOP_PER_IMAGE_FWD_BWD
        if args.sdxl:        # per-image FLOPs of the two embedders differ only in the last stage (out 256 / 288): use the prong value
            flop_img = (args.prongs * sdxl_flops_per_image(out_dim=256)[1] + sdxl_flops_per_image(out_dim=288)[1]) / (1 + args.prongs)
        per_gpu_tflops = (args.batch * (1 + args.prongs) * flop_img) / (elapsed / args.steps) / 1e12
        n_maps = args.batch * (1 + args.prongs)
        if args.ragged_inference:                 # forward only over the actual number of maps of this batch
            n_maps = args.batch + batch[10][1]
            flop_img = 4.9706e9
            per_gpu_tflops = n_maps * flop_img / (elapsed / args.steps) / 1e12
        out = {
            "metric": "events/sec (inference only) on ragged batches, 1-16 prongs/event, packed attention mask" if args.ragged_inference else
                      "events/sec (fwd+bwd) at batch=32, 8 prongs/event; fraction of MFMA roofline" if not args.sdxl else
                      "events/sec (fwd+bwd) at batch=16, 8 prongs/event, --sdxl embedder; fraction of MFMA roofline",
            "value": round(value, 2), "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"TransformerCVN DenseNet [3,6,12,6,3] g32, eval-mode forward only, batch {args.batch}/GPU, ragged "
                                    f"1-16 prongs/event ({n_maps - args.batch} prong maps in this batch), 3x400x280 maps, 6-layer encoder")
                                   if args.ragged_inference else
                                   (f"TransformerCVN DenseNet [3,6,12,6,3] g32, fwd+loss+bwd, batch {args.batch}/GPU, "
                                    f"{args.prongs} prongs/event, 3x400x280 maps, 6-layer encoder, dropout 0.1") if not args.sdxl else
                                   (f"TransformerCVN --sdxl embedder (VAE-encoder blocks [64,64,128,128,256,256,512,512,out], GroupNorm(1), "
                                    f"parity unpinned), fwd+loss+bwd, batch {args.batch}/GPU, {args.prongs} prongs/event, 3x400x280 maps"),
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "precision": args.precision},
            "model_tflops_per_gpu": round(per_gpu_tflops, 2),
            "frac_of_mfma_peak_whole_step": round(per_gpu_tflops / PEAK[args.precision], 4),
            "loss": round(lossv, 5),
            "library": os.path.basename(_lib.LIB_PATH),
        }
        if roof:
            out["roofline"] = roof
            out["kernels"] = kernels[:8]
        if not args.ragged_inference and world == 1 and not args.no_optimizer_leg:
            # beside the fwd+bwd figure BASELINE.json's metric names: the same step followed by the optimizer update the reference's
            # train.py performs (AdamW over two groups + gradient clipping; here one fused launch over the flat arenas)
            try:
                optimizer = model.configure_optimizers()[0][0]
                for _ in range(2):
                    step(); optimizer.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    step(); optimizer.step()
                torch.cuda.synchronize()
                out["with_optimizer_ms_per_step"] = round(1000 * (time.perf_counter() - t0) / 3, 3)
                out["with_optimizer_note"] = f"fwd+loss+bwd + {type(optimizer).__name__}.step(), 3 steps"
            except Exception as ex:                # a reported extra, never a reason to lose the line
                note(f"optimizer-step timing skipped: {ex}")
        if args.precision == "bf16" and world == 1 and not args.no_fp32 and not args.sdxl and not args.ragged_inference:
            note("fp32 parity mode (1 warm-up + 2 steps) ...")
            del model, rt
            torch.cuda.empty_cache()
            out["fp32_ms_per_step"] = round(fp32_parity_mode_ms(opt_path, args, dev, batch), 2)
            out["fp32_events_per_s"] = round(args.batch / out["fp32_ms_per_step"] * 1000, 1)
        if (args.precision == "bf16" and world == 1 and not args.no_batch8 and not args.sdxl and not args.ragged_inference and not strong
                and args.batch == 32):
            # the strong-scaling regime of the north star (batch 64 over 8 GPUs = 8 events per rank) as this one GPU sees it: the per-rank
            # share and the whole batch, so that the 8-way ceiling (batch-64 time / batch-8 time, before any exchange cost) is on the record
            note("batch 8 and batch 64 per GPU (2 warm-up + 5 / 3 steps) ...")
            model = rt = None
            torch.cuda.empty_cache()
            out["batch8_ms_per_step"] = round(small_batch_ms(opt_path, args, dev, 8), 3)
            out["batch8_events_per_s"] = round(8 / out["batch8_ms_per_step"] * 1000, 1)
            torch.cuda.empty_cache()
            out["batch64_ms_per_step"] = round(small_batch_ms(opt_path, args, dev, 64, steps=3), 3)
            out["strong_scaling_ceiling_8way"] = round(out["batch64_ms_per_step"] / out["batch8_ms_per_step"], 2)
            out["batch8_note"] = ("same bf16 fwd+loss+bwd step at 8 events/GPU (one rank's share of batch 64 over 8 GPUs) and at 64 events on "
                                  "this one GPU; ceiling = their ratio, no exchange cost in it")
            torch.cuda.empty_cache()
        if args.precision == "bf16" and world == 1 and not args.no_sdxl and not args.sdxl and not args.ragged_inference and not strong:
            note("--sdxl embedder, config 4 (1 warm-up + 2 steps) ...")
            model = rt = None
            torch.cuda.empty_cache()
            out["sdxl_ms_per_step"] = round(sdxl_mode_ms(opt_path, args, dev), 2)
            out["sdxl_events_per_s"] = round(16 / out["sdxl_ms_per_step"] * 1000, 1)
            out["sdxl_note"] = "BASELINE config 4: batch 16 x 8 prongs, bf16, fwd+loss+bwd; parity of this embedder is unpinned (diffusers absent)"
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1 and not args.sdxl and not args.ragged_inference:       # rank 0 at N=1 only (DenseNet workload)
            note("cpu baseline (oracle on host cores) ...")
            out["cpu_baseline"] = cpu_baseline(host_threads(), args.prongs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
