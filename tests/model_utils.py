"""Build the drop-in Lightning module of this repo from an oracle config (tests only)."""
import torch


def build_trainer(cfg, state=None, precision="fp32", device="cuda", train_file="synthetic:16:4"):
    from transformercvn.options import Options
    from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
    if getattr(cfg, "embedder", "dense") == "sdxl":
        from transformercvn.network.trainers.neutrino_full_sdxl_trainer import NeutrinoFullSDXLTrainer as NeutrinoFullDenseTrainer
    o = Options()
    o.update_options({k: v for k, v in vars(cfg).items() if k in vars(o)})
    o.training_file = train_file
    o.hip_precision = precision
    o.batch_size = 2
    o.num_dataloader_workers = 0
    m = NeutrinoFullDenseTrainer(o)
    if state is not None:
        res = m.load_state_dict(state, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    return m.to(device) if device else m


def to_device(batch, device="cuda"):
    return tuple(t.to(device) if torch.is_tensor(t) else t for t in batch)
