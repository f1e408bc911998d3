"""SURVEY.md 8(f) rows 2 and 4 on the CPU: the HDF5 dataset interface on an in-memory stand-in for the h5py file (keys per the
reference's dataset/minkowski_dataset.py:113-181; h5py itself is not installed here) and the torchmetrics-free validation
metrics against scikit-learn."""
import os
import sys
import types

import numpy as np
import pytest
import torch


class _FakeId:
    def __init__(self, offset):
        self._o = offset

    def get_offset(self):
        return self._o


class _FakeDataset:
    """h5py.Dataset look-alike over a numpy array that also lives at `offset` of a flat binary file (for np.memmap)."""

    def __init__(self, arr, offset):
        self.arr, self.id = arr, _FakeId(offset)
        self.shape, self.dtype = arr.shape, arr.dtype

    def __getitem__(self, idx):
        return self.arr[idx]


def _make_file(tmp_path, n=12, max_p=20, F=4, E=2, seed=0):
    rng = np.random.default_rng(seed)
    arrays = {
        "features": rng.normal(size=(n, max_p, F)).astype(np.float32),
        "extra": rng.normal(size=(n, E)).astype(np.float32),
        "prong_mask": (rng.random((n, max_p)) < 0.3),
        "event_target": rng.integers(0, 10, size=n).astype(np.int64),
        "prong_target": rng.integers(0, 8, size=(n, max_p)).astype(np.int8),
        "full_pixels_shape": np.array([3, 400, 280]),
        "event_pixels_shape": np.array([n, 400, 280]), "prong_pixels_shape": np.array([n * max_p, 400, 280]),
    }
    arrays["prong_mask"][:, 0] = rng.random(n) < 0.5           # the dataset must force column 0 to True
    for kind in ("event", "prong"):
        counts = rng.integers(3, 9, size=n)
        hi = np.cumsum(counts)
        arrays[f"{kind}_compressed_index"] = np.stack([hi - counts, hi], 1).astype(np.int64)
        nnz = int(hi[-1])
        arrays[f"{kind}_pixels_coordinates"] = np.stack([rng.integers(0, 3, nnz), rng.integers(0, 400, nnz), rng.integers(0, 280, nnz)],
                                                         1).astype(np.int32)
        arrays[f"{kind}_pixels_values"] = rng.integers(1, 256, size=(nnz, 3)).astype(np.float32)
    path = str(tmp_path / "fake.h5")
    offsets, blob = {}, b""
    for k, a in arrays.items():
        offsets[k] = len(blob)
        blob += np.ascontiguousarray(a).tobytes()
    with open(path, "wb") as f:
        f.write(blob)
    store = {k: _FakeDataset(a, offsets[k]) for k, a in arrays.items()}
    mod = types.ModuleType("h5py")
    mod.File = lambda p, mode="r": store
    return path, arrays, mod


@pytest.mark.parametrize("full", [False, True])
def test_minkowski_dataset_reads_the_reference_layout(tmp_path, monkeypatch, full):
    from transformercvn.dataset.minkowski_dataset import MinkowskiDataset, MinkowskiCollection
    path, a, h5 = _make_file(tmp_path)
    monkeypatch.setitem(sys.modules, "h5py", h5)
    ds = MinkowskiDataset(path, 1.0, load_full_dataset=full)
    # reference quirk (:112-119): max_limit = limit_index.max() is used as an EXCLUSIVE slice bound, so the last event of every
    # range is dropped -- reproduced, because steps_per_epoch and the train/validation split depend on it
    assert len(ds) == 11 and ds.pixel_shape == (400, 280) and ds.pixel_features == 3
    assert ds.num_features == 4 and ds.num_extra == 2 and ds.max_particles == 20
    assert bool(ds.prong_mask[:, 0].all())                                         # reference :181
    for i in (0, 5, 10):
        f, x, ec, ev, em, pc, pv, pm, et, pt = ds[i]
        for kind, c, v in (("event", ec, ev), ("prong", pc, pv)):
            lo, hi = a[f"{kind}_compressed_index"][i]
            assert np.array_equal(c.numpy(), a[f"{kind}_pixels_coordinates"][lo:hi])
            assert np.array_equal(v.numpy(), a[f"{kind}_pixels_values"][lo:hi])
        assert torch.equal(f, torch.from_numpy(a["features"][i])) and int(et) == int(a["event_target"][i])
        assert em.shape == (1,) and bool(em[0])
    # sub-ranges: the second half re-bases the compressed indices (reference :138-146, :185-186)
    tail = MinkowskiDataset(path, -0.5, load_full_dataset=full)
    assert len(tail) == 5
    f, x, ec, ev, em, pc, pv, pm, et, pt = tail[0]
    lo, hi = a["prong_compressed_index"][6]
    assert np.array_equal(pc.numpy(), a["prong_pixels_coordinates"][lo:hi])
    mid = MinkowskiDataset(path, (0.25, 0.75), load_full_dataset=full)
    assert len(mid) == 5 and torch.equal(mid[0][0], torch.from_numpy(a["features"][3]))
    # event_current_targets: 4-7 -> 1, 8 -> 2, 9 -> 3, else 0 (reference :127-133)
    cur = MinkowskiDataset(path, 1.0, event_current_targets=True, load_full_dataset=full)
    t = a["event_target"]
    expect = np.where((t > 3) & (t <= 7), 1, np.where(t == 8, 2, np.where(t == 9, 3, 0)))
    assert np.array_equal(cur.event_targets.numpy(), expect[:11])
    # statistics over real prongs only
    mean, std, em_, es_, _, _ = ds.compute_statistics()
    masked = torch.from_numpy(a["features"][:11])[ds.prong_mask]
    assert torch.allclose(mean, masked.mean(0)) and torch.allclose(std, masked.std(0))
    # collate: prong image index re-based by the number of real prongs of the preceding events
    batch = MinkowskiCollection()([ds[i] for i in range(3)])
    assert batch[0].shape == (3, 20, 4) and batch[7].shape == (3, 20)
    n0 = int(ds.prong_mask[0].sum())
    first_of_second = a["prong_compressed_index"][0, 1] - a["prong_compressed_index"][0, 0]
    assert int(batch[5][first_of_second, 0]) == int(a["prong_pixels_coordinates"][a["prong_compressed_index"][1, 0], 0]) + n0


def test_trainer_opens_hdf5_files_like_the_reference(tmp_path, monkeypatch):
    """NeutrinoBase.create_datasets: one file split by train_validation_split, dimensions derived from the file (:55-62, :68-86)."""
    from transformercvn.options import Options
    from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer
    path, a, h5 = _make_file(tmp_path, n=20)
    monkeypatch.setitem(sys.modules, "h5py", h5)
    o = Options()
    o.update_options(dict(densenet_structure=[1, 1], densenet_growth_rate=8, initial_pixel_dim=16, num_encoder_layers=1,
                          training_file=path, train_validation_split=0.8, batch_size=2, num_dataloader_workers=0))
    m = NeutrinoFullDenseTrainer(o)
    assert len(m.training_dataset) == 15 and len(m.validation_dataset) == 3 and m.testing_dataset is None      # quirk above
    assert m.network.event_decoder.hidden_layer.out_features == int(a["event_target"][:15].max()) + 1
    assert m.training_dataset.pixel_shape == (400, 280)
    assert tuple(m.mean.shape) == (4,) and not m.mean.requires_grad
    with pytest.raises(ValueError):
        m.test_dataloader()
    b = next(iter(m.train_dataloader()))
    assert len(b) == 10 and b[0].shape[0] == 2 and b[5].dtype == torch.int32


def test_fallback_metrics_match_sklearn():
    from sklearn.metrics import roc_auc_score, accuracy_score
    from transformercvn.network.trainers.metrics import _Accuracy, _Auroc
    rng = np.random.default_rng(3)
    acc, auc = _Accuracy(), _Auroc(4)
    ps, ts = [], []
    for _ in range(3):                                   # three validation batches
        p = torch.softmax(torch.from_numpy(rng.normal(size=(17, 4)).astype(np.float32)), 1)
        t = torch.from_numpy(rng.integers(0, 4, size=17))
        acc.update(p, t); auc.update(p, t)
        ps.append(p.numpy()); ts.append(t.numpy())
    P, T = np.concatenate(ps), np.concatenate(ts)
    assert abs(acc.compute().item() - accuracy_score(T, P.argmax(1))) < 1e-7
    assert abs(auc.compute().item() - roc_auc_score(T, P, multi_class="ovr", average="macro")) < 1e-6
    acc.reset(); auc.reset()
    assert acc.total == 0 and auc.p == []
