"""Helpers to read the committed golden vectors (tests/golden/*.npz; made by oracle/make_golden.py)."""
import ast
import os

import numpy as np
import torch

from oracle import tcvn_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    g = dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))
    over = {str(k): ast.literal_eval(str(v)) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    cfg = O.tutorial_config(**over)
    batch = (
        torch.from_numpy(g["features"]), torch.from_numpy(g["extra"]),
        torch.from_numpy(g["event_coords"].astype(np.int32)), torch.from_numpy(g["event_values"].astype(np.float32)),
        torch.from_numpy(g["event_mask"]), torch.from_numpy(g["prong_coords"].astype(np.int32)),
        torch.from_numpy(g["prong_values"].astype(np.float32)), torch.from_numpy(g["prong_mask"]),
        torch.from_numpy(g["event_targets"]), torch.from_numpy(g["prong_targets"]),
    )
    return cfg, over, batch, g


def train_cfg(over):
    return O.tutorial_config(**dict(over, dropout=0.0, pixel_noise_std=0.0))


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def tap_sample(t):
    f = t.detach().double().reshape(-1)
    step = max(1, f.numel() // 256)
    return np.array([f.mean().item(), f.std().item(), f.abs().max().item()]), f[::step][:256].float().numpy()
