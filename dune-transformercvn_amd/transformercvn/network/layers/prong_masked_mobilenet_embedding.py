"""Only the channel-rounding helper of the reference file is on the hot path
(reference: transformercvn/network/layers/prong_masked_mobilenet_embedding.py:10-23)."""
from typing import Optional


def make_divisible_channel_count(v: float, divisor: int, min_value: Optional[int] = None) -> int:
    """Round ``v`` to the nearest multiple of ``divisor`` (at least ``min_value``), never dropping more than 10 %."""
    floor = divisor if min_value is None else min_value
    rounded = max(floor, (int(v + divisor / 2) // divisor) * divisor)
    return rounded + divisor if rounded < 0.9 * v else rounded
