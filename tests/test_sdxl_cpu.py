"""SDXL embedder on the CPU: the restated encoder (oracle/sdxl_oracle.py, parity unpinned) meets the shape facts the reference
fixes, the drop-in modules reproduce its state_dict layout key for key, and the C-ABI plan lists the same slots."""
import torch

from oracle import tcvn_oracle as O
from oracle import sdxl_oracle as S


def test_block_channels_and_shape_facts():
    assert S.block_channels(64, 256) == [64, 64, 128, 128, 256, 256, 512, 512, 256]        # sdxl_net.py:19-25
    cfg = O.tutorial_config(embedder="sdxl", initial_pixel_dim=8, pixel_embedding_dim=64)
    sd = O.fill_state(cfg, 2)
    pfx = "network.prong_embedding.prong_pixel_embedding"
    x = torch.zeros(2, 3, 400, 280)
    x[0, :, 10, 20] = 0.5
    x[1, :, 399, 279] = 1.0
    taps = {}
    with torch.no_grad():
        out = S.sdxl_forward(sd, pfx, x, taps)
    assert out.shape == (2, 64)                                               # Flatten + Linear(out, out) needs ...
    assert tuple(taps[pfx + ":mid"].shape) == (2, 64, 1, 1)                   # ... the 1x1 final map (8 stride-2 stages)
    assert tuple(taps[pfx + ":block1"].shape[2:]) == (200, 140) and tuple(taps[pfx + ":block4"].shape[2:]) == (25, 17)
    assert tuple(taps[pfx + ":block5"].shape[2:]) == (12, 8)                  # (0,1,0,1) pad + stride 2: floor halving
    assert not torch.allclose(out[0], out[1])


def test_sdxl_modules_reproduce_the_state_dict_layout_and_plan_slots():
    from model_utils import build_trainer
    cfg = O.tutorial_config(embedder="sdxl", initial_pixel_dim=8)
    m = build_trainer(cfg, None, device=None)
    assert type(m).__name__ == "NeutrinoFullSDXLTrainer"
    lay = O.state_layout(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(lay.keys())
    assert all(tuple(sd[k].shape) == tuple(lay[k]) for k in lay)
    m.load_state_dict(O.fill_state(cfg, 5), strict=True)
    pfx = "network.prong_embedding.event_pixel_embedding."
    eng = m.network.prong_embedding.event_pixel_embedding.hip_engine(0, 400, 280)
    assert [s[0] for s in eng.slots()] == [k[len(pfx):] for k in lay if k.startswith(pfx)]
    # BASELINE config 4 (batch 16 x 8 prongs = 128 prong maps + 16 event maps, d = 64, bf16) fits one GPU's 288 GB
    from transformercvn.hip.engine import SdxlEngine
    big = SdxlEngine(3, 256, 64, 2, 4, 400, 280, 1)
    assert big.workspace_bytes(128, True) < 200e9
    print("sdxl bf16 workspace, 128 maps with backward:", big.workspace_bytes(128, True) / 1e9, "GB")
