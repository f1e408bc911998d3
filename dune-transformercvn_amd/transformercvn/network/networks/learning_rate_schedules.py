"""LambdaLR schedules used by configure_optimizers (reference: networks/learning_rate_schedules.py:49-75, :113-145)."""
import math

from torch.optim.lr_scheduler import LambdaLR


def _warmup(step: int, warmup: int) -> float:
    return step / max(1, warmup)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """Linear ramp 0 -> lr over the warm-up, then linear decay to 0 at ``num_training_steps``."""
    span = max(1, num_training_steps - num_warmup_steps)

    def factor(step: int) -> float:
        if step < num_warmup_steps:
            return _warmup(step, num_warmup_steps)
        return max(0.0, (num_training_steps - step) / span)

    return LambdaLR(optimizer, factor, last_epoch)


def get_cosine_with_hard_restarts_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, num_cycles=1,
                                                       last_epoch=-1):
    """Warm-up, then ``num_cycles`` half-cosine decays with hard restarts; 0 once the training length is reached."""
    span = max(1, num_training_steps - num_warmup_steps)

    def factor(step: int) -> float:
        if step < num_warmup_steps:
            return _warmup(step, num_warmup_steps)
        progress = (step - num_warmup_steps) / span
        if progress >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((num_cycles * progress) % 1.0))))

    return LambdaLR(optimizer, factor, last_epoch)
