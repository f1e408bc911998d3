"""Feeder side of the hot path: the sparse-HDF5 dataset interface and its collate function.

* ``MinkowskiCollection`` -- the collate_fn (reference: transformercvn/dataset/minkowski_dataset.py:29-86): stacks the
  per-event tensors and re-bases the prong image index of every event's COO list to the *packed* batch index
  (= number of real prongs of the preceding events).
* ``SyntheticDataset`` -- seeded in-memory stand-in with the attributes the trainer reads (no data ships with the
  reference and there is no network here): events with 3x400x280 maps, unique hit coordinates, every image >= 1 hit.
* ``MinkowskiDataset`` -- the HDF5-backed dataset; needs h5py at construction time.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import Dataset


class MinkowskiCollection:
    @staticmethod
    def collate_sparse(coordinates: Sequence[Tensor], values: Sequence[Tensor], masks: Sequence[Tensor]) -> Tuple[Tensor, Tensor]:
        shifted, base = [], 0
        for coord, mask in zip(coordinates, masks):
            if base:
                coord = coord.clone()
                coord[:, 0] += base
            shifted.append(coord)
            base += int(mask.sum())
        return torch.cat(shifted), torch.cat(list(values))

    def __call__(self, samples):
        (features, extra, ev_coords, ev_values, ev_masks, pr_coords, pr_values, pr_masks, ev_targets, pr_targets) = zip(*samples)
        event_coordinates, event_values = self.collate_sparse(ev_coords, ev_values, ev_masks)
        prong_coordinates, prong_values = self.collate_sparse(pr_coords, pr_values, pr_masks)
        return (torch.stack(features), torch.stack(extra), event_coordinates, event_values, torch.stack(ev_masks),
                prong_coordinates, prong_values, torch.stack(pr_masks), torch.stack(ev_targets), torch.stack(pr_targets))


class SyntheticDataset(Dataset):
    """Seeded synthetic events: event maps with U{500..4000} hits, prong maps with U{20..800} hits,
    integer pixel values U{1..255} in all channels, ``prongs`` real prongs per event (fixed int or (lo, hi) range)."""

    def __init__(self, num_events: int = 1024, prongs=8, max_particles: int = 20, num_features: int = 4, num_extra: int = 2,
                 pixel_shape: Tuple[int, int] = (400, 280), pixel_features: int = 3, num_event_classes: int = 4,
                 num_prong_classes: int = 8, seed: int = 1234, event_hits=(500, 4000), prong_hits=(20, 800)):
        self.num_events, self.max_particles = num_events, max_particles
        self.num_features, self.num_extra = num_features, num_extra
        self.pixel_shape, self.pixel_features = tuple(pixel_shape), pixel_features
        self.num_event_classes, self.num_prong_classes = num_event_classes, num_prong_classes
        self.prongs, self.seed, self.event_hits, self.prong_hits = prongs, seed, event_hits, prong_hits
        self.pixels = None
        self.mean, self.std, self.extra_mean, self.extra_std = 0, 1, 0, 1

    def __len__(self) -> int:
        return self.num_events

    def compute_statistics(self):
        return (torch.zeros(self.num_features), torch.ones(self.num_features), torch.tensor(0.0), torch.tensor(1.0), None, None)

    def _image(self, rng, index: int, hits) -> Tuple[np.ndarray, np.ndarray]:
        H, W = self.pixel_shape
        nnz = int(rng.integers(hits[0], hits[1] + 1))
        flat = np.sort(rng.choice(H * W, size=nnz, replace=False))
        coords = np.stack([np.full(nnz, index), flat // W, flat % W], 1).astype(np.int32)
        values = rng.integers(1, 256, size=(nnz, self.pixel_features)).astype(np.float32)
        return coords, values

    def __getitem__(self, item: int):
        rng = np.random.Generator(np.random.Philox(key=self.seed * 1000003 + int(item)))
        n = self.prongs if isinstance(self.prongs, int) else int(rng.integers(self.prongs[0], self.prongs[1] + 1))
        n = max(1, min(n, self.max_particles))
        ec, ev = self._image(rng, 0, self.event_hits)
        pcs, pvs = zip(*(self._image(rng, i, self.prong_hits) for i in range(n)))
        mask = torch.zeros(self.max_particles, dtype=torch.bool)
        mask[:n] = True
        targets = torch.full((self.max_particles,), -1, dtype=torch.int8)
        targets[:n] = torch.from_numpy(rng.integers(0, self.num_prong_classes, size=n).astype(np.int8))
        return (torch.zeros(self.max_particles, self.num_features), torch.zeros(self.num_extra),
                torch.from_numpy(ec), torch.from_numpy(ev), torch.ones(1, dtype=torch.bool),
                torch.from_numpy(np.concatenate(pcs)), torch.from_numpy(np.concatenate(pvs)), mask,
                torch.tensor(int(rng.integers(0, self.num_event_classes)), dtype=torch.int64), targets)


class MinkowskiDataset(Dataset):
    """HDF5 sparse pixel store (reference: dataset/minkowski_dataset.py:89-281).  Keys: features, extra, prong_mask,
    event_target, prong_target, {event,prong}_compressed_index, {event,prong}_pixels_{coordinates,values,shape},
    full_pixels_shape.  Per-event COO slices are read through np.memmap unless ``load_full_dataset``."""

    def __init__(self, data_file: str, limit_index=1.0, event_current_targets: bool = False, load_full_dataset: bool = False):
        super().__init__()
        try:
            import h5py
        except ImportError as e:                                        # pragma: no cover
            raise ImportError("MinkowskiDataset needs h5py to open " + data_file) from e
        self.load_full_dataset = load_full_dataset
        self.mean, self.std, self.extra_mean, self.extra_std = 0, 1, 0, 1
        self.pixel_mean, self.pixel_std, self.pixels = 0, 1, None
        f = h5py.File(data_file, "r")
        self.num_events = f["features"].shape[0]
        index = self.compute_limit_index(limit_index)
        lo, hi = int(index.min()), int(index.max())
        self.min_limit, self.max_limit = lo, hi
        self.features = torch.from_numpy(f["features"][lo:hi])
        self.extra = torch.from_numpy(f["extra"][lo:hi])
        self.prong_mask = torch.from_numpy(f["prong_mask"][lo:hi]).bool()
        self.prong_mask[:, 0] = True                                    # every event keeps >= 1 prong (:181)
        self.event_targets = torch.from_numpy(f["event_target"][lo:hi])
        self.prong_targets = torch.from_numpy(f["prong_target"][lo:hi])
        if event_current_targets:                                       # 4-7 -> 1, 8 -> 2, 9 -> 3, else 0 (:127-133)
            t = self.event_targets.numpy()
            cur = np.zeros_like(t)
            cur[(t > 3) & (t <= 7)] = 1
            cur[t == 8] = 2
            cur[t == 9] = 3
            self.event_targets = torch.from_numpy(cur)
        self._stores = {}
        for kind in ("event", "prong"):
            comp = f[f"{kind}_compressed_index"][lo:hi]
            base = int(comp[0, 0])
            top = int(comp[-1, -1])
            if load_full_dataset:
                coords = torch.from_numpy(f[f"{kind}_pixels_coordinates"][base:top])
                values = torch.from_numpy(f[f"{kind}_pixels_values"][base:top])
                shift = base
            else:
                coords, values = (np.memmap(data_file, mode="r", shape=f[k].shape, offset=f[k].id.get_offset(), dtype=f[k].dtype)
                                  for k in (f"{kind}_pixels_coordinates", f"{kind}_pixels_values"))
                shift = 0
            self._stores[kind] = (comp, coords, values, shift)
        full = f["full_pixels_shape"][:].tolist()
        self.pixel_features, self.pixel_shape = full[0], tuple(full[1:])
        self.num_events, self.max_particles, self.num_features = self.features.shape
        self.num_extra = self.extra.shape[1]
        self.num_event_classes = int(self.event_targets.max()) + 1
        self.num_prong_classes = int(self.prong_targets.max()) + 1
        self.event_mask = torch.ones(self.num_events, 1, dtype=torch.bool)

    def compute_limit_index(self, limit_index) -> np.ndarray:
        if isinstance(limit_index, float):
            limit_index = (0.0, limit_index) if limit_index > 0 else (1.0 + limit_index, 1.0)
        if isinstance(limit_index, (list, tuple)):
            limit_index = np.arange(int(round(limit_index[0] * self.num_events)), int(round(limit_index[1] * self.num_events)))
        if isinstance(limit_index, Tensor):
            limit_index = limit_index.numpy()
        return np.sort(limit_index)

    def compute_statistics(self, mean=None, std=None, extra_mean=None, extra_std=None):
        if mean is None:
            masked = self.features[self.prong_mask]
            mean, std = masked.mean(0), masked.std(0)
            std[std < 1e-5] = 1
        if extra_mean is None:
            extra_mean, extra_std = self.extra.mean(), self.extra.std()
        self.mean, self.std, self.extra_mean, self.extra_std = mean, std, extra_mean, extra_std
        return mean, std, extra_mean, extra_std, None, None

    def __len__(self) -> int:
        return self.num_events

    def _slice(self, kind: str, item: int):
        comp, coords, values, shift = self._stores[kind]
        lo, hi = int(comp[item][0]) - shift, int(comp[item][-1]) - shift
        c, v = coords[lo:hi], values[lo:hi]
        if not torch.is_tensor(c):
            c, v = torch.from_numpy(np.array(c)), torch.from_numpy(np.array(v))
        return c, v

    def __getitem__(self, item):
        ec, ev = self._slice("event", item)
        pc, pv = self._slice("prong", item)
        return (self.features[item], self.extra[item], ec, ev, self.event_mask[item], pc, pv, self.prong_mask[item],
                self.event_targets[item], self.prong_targets[item])
