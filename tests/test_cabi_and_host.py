"""CPU-side checks: the C-ABI library loads and exports every symbol of include/tcvn_hip.h; plan/slot tables match the
reference state_dict layout; host logic (options, collate, token rows, LR schedules) behaves like the reference."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from transformercvn.hip import _lib
    header = open(os.path.join(ROOT, "include", "tcvn_hip.h")).read()
    declared = set(re.findall(r"\b(tcvn_[a-z0-9_]+)\s*\(", header))
    declared -= {"tcvn_densenet_cfg", "tcvn_head_cfg"}
    assert len(declared) >= 20
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(dll, name), name
    assert _lib.lib.tcvn_version() == 1
    assert set(_lib.EXPORTS) <= declared


def test_product_library_has_no_environment_switches():
    """The shipped library must not change behaviour with the environment: the validation / ablation knobs (TCVN_DBG,
    TCVN_DISABLE_TILE, ...) exist only in the -DTCVN_DEBUG_KNOBS build (libtcvn_hip_dbg.so)."""
    import subprocess
    from transformercvn.hip import _lib
    assert os.path.basename(_lib.LIB_PATH) == "libtcvn_hip.so"
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    blob = open(_lib.LIB_PATH, "rb").read()
    for knob in (b"TCVN_DBG", b"TCVN_DISABLE_TILE", b"TCVN_XA_ONTHEFLY", b"TCVN_BWD_SERIAL", b"TCVN_POOL0_BWD_FLAT", b"TCVN_DENSE_STEM", b"TCVN_SPARSE_STEM_TRAIN"):
        assert knob not in blob, knob
    # ... and the Python loader binds the product library whatever the environment says (the debug build is selected only by an
    # explicit _libselect.use() call in a test child process)
    import sys
    code = ("import sys; sys.path[:0] = %r\nfrom transformercvn.hip import _lib\nimport os\nprint(os.path.basename(_lib.LIB_PATH))" % (sys.path,))
    env = dict(os.environ, TCVN_HIP_LIBRARY="libtcvn_hip_dbg.so", TCVN_DISABLE_TILE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, check=True).stdout.strip()
    assert out == "libtcvn_hip.so", out
    src = open(_lib.__file__).read() + open(_lib._libselect.__file__).read()
    assert "os.environ" not in src and "getenv" not in src


def test_densenet_plan_slots_match_reference_layout():
    from transformercvn.hip.engine import DenseNetEngine
    cfg = O.tutorial_config()
    eng = DenseNetEngine(3, 256, 64, 32, 4, [3, 6, 12, 6, 3], 400, 280, 0.1, 0)
    lay = {}
    O.densenet_layout("x", cfg, 3, 256, lay)
    slots = eng.slots()
    assert [s[0] for s in slots] == [k[2:] for k in lay]
    for (name, numel, kind), (k, shp) in zip(slots, lay.items()):
        assert numel == (int(np.prod(shp)) if len(shp) else 1), name
    # fp32 and bf16 workspaces: 256 prong maps with backward must fit comfortably in 288 GB
    assert eng.workspace_bytes(256, True) < 40e9
    assert DenseNetEngine(3, 256, 64, 32, 4, [3, 6, 12, 6, 3], 400, 280, 0.1, 1).workspace_bytes(256, True) < 25e9


def test_module_state_dict_is_the_reference_layout():
    from model_utils import build_trainer
    for over in (dict(), dict(dropout=0.0), dict(densenet_structure=[2, 2], num_prong_decoder_layers=3, hidden_dim=64)):
        cfg = O.tutorial_config(**over)
        m = build_trainer(cfg, None, device=None)
        lay = O.state_layout(cfg)
        sd = m.state_dict()
        assert list(sd.keys()) == list(lay.keys())
        assert all(tuple(sd[k].shape) == tuple(lay[k]) for k in lay)
        m.load_state_dict(O.fill_state(cfg, 5), strict=True)


def test_head_plan_slots_are_state_dict_keys():
    from model_utils import build_trainer
    from transformercvn.hip import _lib
    cfg = O.tutorial_config()
    m = build_trainer(cfg, None, device=None)
    rt = m.network.hip_runtime()
    names = {n for n, _, kind in rt.head.slots()}
    keys = {k[len("network."):] for k in m.state_dict() if k.startswith("network.")}
    assert names <= keys
    assert "encoder.encoder.layers.5.self_attn.in_proj_weight" in names and "prong_decoder.hidden_layers.12.weight" in names


def test_options_json_overlay_and_coercion(tmp_path):
    from transformercvn.options import Options
    p = tmp_path / "o.json"
    p.write_text(json.dumps({"hidden_dim": "64", "dropout": 0.25, "verbose_output": 1, "brand_new_key": [1, 2]}))
    o = Options.load(str(p))
    assert o.hidden_dim == 64 and isinstance(o.hidden_dim, int)
    assert o.dropout == 0.25 and o.brand_new_key == [1, 2]
    assert o.verbose_output == 1                      # bool defaults are ints first -> int() coercion, like the reference
    assert Options().densenet_structure == [6, 12, 24, 16]


def test_collate_rebases_prong_indices_to_packed_order():
    from transformercvn.dataset.minkowski_dataset import SyntheticDataset, MinkowskiCollection
    ds = SyntheticDataset(8, (1, 5), seed=3, event_hits=(5, 10), prong_hits=(2, 4))
    items = [ds[i] for i in range(4)]
    b = MinkowskiCollection()(items)
    counts = [int(it[7].sum()) for it in items]
    assert b[0].shape == (4, 20, 4) and b[7].shape == (4, 20)
    assert int(b[5][:, 0].max()) + 1 == sum(counts)
    assert int(b[2][:, 0].max()) + 1 == 4
    # images of event e occupy packed indices [sum(counts[:e]), sum(counts[:e+1]))
    lo = 0
    off = 0
    for it, c in zip(items, counts):
        n = it[5].shape[0]
        seg = b[5][off:off + n, 0]
        assert int(seg.min()) == lo and int(seg.max()) == lo + c - 1
        lo += c
        off += n


def test_token_rows_and_pack_indices():
    from transformercvn.network.layers.packed_data import token_rows, pack_indices, masked_pad_1d_precomputed
    mask = torch.tensor([[1, 1, 0], [1, 0, 0], [1, 1, 1]], dtype=torch.bool)
    tr = token_rows(mask, 3)
    assert tr.tolist() == [[0, 3, 4, -1], [1, 5, -1, -1], [2, 6, 7, 8]]
    i1, i2 = pack_indices(mask)
    o1, o2 = O.pack_indices(mask)
    assert i1.tolist() == o1.tolist() and i2.tolist() == o2.tolist()
    packed = torch.arange(6.).view(6, 1)
    assert masked_pad_1d_precomputed(packed, i1, i2, 3, 3)[2, :, 0].tolist() == [3., 4., 5.]


def test_lr_schedules_match_closed_form():
    import math
    from transformercvn.network.networks.learning_rate_schedules import (get_linear_schedule_with_warmup,
                                                                        get_cosine_with_hard_restarts_schedule_with_warmup)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    s = get_cosine_with_hard_restarts_schedule_with_warmup(opt, 10, 110, 4)
    lrs = []
    for _ in range(112):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); s.step()
    assert lrs[0] == 0 and abs(lrs[5] - 0.5) < 1e-12 and abs(lrs[10] - 1.0) < 1e-12
    assert abs(lrs[35] - 1.0) < 1e-9                       # hard restart at progress 0.25
    assert abs(lrs[22] - 0.5 * (1 + math.cos(math.pi * ((4 * 0.12) % 1.0)))) < 1e-9 and lrs[111] == 0.0
    opt = torch.optim.SGD([p], lr=1.0)
    s = get_linear_schedule_with_warmup(opt, 10, 110)
    for _ in range(60):
        opt.step(); s.step()
    assert abs(opt.param_groups[0]["lr"] - 0.5) < 1e-12


def test_optimizer_groups_reproduce_reference_decay_mask():
    from model_utils import build_trainer
    m = build_trainer(O.tutorial_config(), None, device=None)
    (opt,), (sch,) = m.configure_optimizers()
    n_decay, n_nodecay = len(opt.param_groups[0]["params"]), len(opt.param_groups[1]["params"])
    assert (n_decay, n_nodecay) == (468, 314)            # SURVEY.md Appendix C, measured on the reference
    assert sch["interval"] == "step"


def test_device_feeder_cpu_passthrough_attaches_host_counts():
    """the feeder stages batches ahead and adds the host-side (max_prongs, n_prongs) pair."""
    import torch
    from oracle import tcvn_oracle as O
    from transformercvn.hip.feeder import DeviceFeeder, host_counts
    cfg = O.tutorial_config()
    batches = [O.synthetic_batch([2, 4, 1], 5 + i, cfg) for i in range(4)]
    out = list(DeviceFeeder(batches, "cpu", depth=2))
    assert len(out) == 4
    for src, got in zip(batches, out):
        assert len(got) == 11 and got[10] == host_counts(src[7])
        assert got[10] == (int(src[7].sum(1).max()), int(src[7].sum()))
        for a, b in zip(src[:10], got[:10]):
            assert torch.equal(a, b)


def test_lightning_precision_16_selects_the_bf16_engines():
    """train.py -fp16 -> pl.Trainer(precision=16) (reference train.py:141,172): the drop-in module maps the trainer's precision to its
    engines unless the option file names hip_precision itself."""
    from types import SimpleNamespace
    from model_utils import build_trainer
    from transformercvn.hip import _lib
    cfg = O.tutorial_config(densenet_structure=[1, 1], densenet_growth_rate=8, initial_pixel_dim=16, num_encoder_layers=1)
    for prec, mode in ((16, _lib.MODE_BF16), ("16-mixed", _lib.MODE_BF16), ("bf16", _lib.MODE_BF16), (32, _lib.MODE_F32), ("32-true", _lib.MODE_F32)):
        m = build_trainer(cfg, None, device=None)
        del m.options.hip_precision                         # option file silent about it (the reference's files are)
        m.trainer = SimpleNamespace(precision=prec)
        m.setup("fit")
        assert m.network.hip_runtime().mode == mode, prec
    m = build_trainer(cfg, None, precision="fp32", device=None)      # an explicit option wins over the trainer
    m.trainer = SimpleNamespace(precision=16)
    m.setup("fit")
    assert m.network.hip_runtime().mode == _lib.MODE_F32
