"""Asynchronous host -> device batch feeder.

Reference input side: ``MinkowskiDataset.__getitem__`` + ``MinkowskiCollection`` (transformercvn/dataset/minkowski_dataset.py:
29-86, 247-281) hand a collated 10-tuple of CPU tensors to Lightning, which copies it synchronously; ``shared_step`` then calls
``.item()``-style reductions on the prong mask (neutrino_full_base_trainer.py:118-146: a device sync per step).

Here the loader's batches are staged ``depth`` ahead: pinned, copied on a dedicated copy stream while the current step
computes, and extended by the host-side ``(max_prongs, n_prongs)`` pair that ``shared_step`` accepts as an optional 11th element
(no device sync on the hot path).  Batches are handed out ordered after their copy on the consumer's current stream."""
from __future__ import annotations

from collections import deque
from typing import Iterable, Iterator

import torch


def host_counts(prong_mask: torch.Tensor):
    """(max prongs per event, total prongs) from the CPU mask [B, P] -- what shared_step would otherwise reduce on the device."""
    per_event = prong_mask.sum(1)
    return int(per_event.max()), int(per_event.sum())


class DeviceFeeder:
    def __init__(self, loader: Iterable, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self._copy_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        counts = host_counts(batch[7])
        if self._copy_stream is None:
            return tuple(batch[:10]) + (counts,), None
        with torch.cuda.stream(self._copy_stream):
            dev = []
            for i, t in enumerate(batch[:10]):
                if i in (2, 5) and t.dtype != torch.int32:
                    t = t.to(torch.int32)                           # COO coordinates travel as int32 (image, y, x)
                dev.append(t.pin_memory().to(self.device, non_blocking=True))
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        return tuple(dev) + (counts,), ready

    def __iter__(self) -> Iterator:
        it = iter(self.loader)
        queue = deque()
        for batch in it:
            queue.append(self._stage(batch))
            if len(queue) > self.depth:
                yield self._release(queue.popleft())
        while queue:
            yield self._release(queue.popleft())

    def _release(self, staged):
        batch, ready = staged
        if ready is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            for t in batch[:10]:
                t.record_stream(cur)                                # the caching allocator must not recycle it under the step
        return batch
