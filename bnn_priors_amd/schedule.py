"""Learning-rate factor of the cyclical schedule (reference: bnn_priors/utils.py:5-10):
within a cycle of S optimizer steps the step size decays as 0.5 (1 + cos(pi k / S))."""
import math


class CosineSchedule:
    """Callable ``k -> factor`` usable as ``lr_lambda``; ``k`` counts scheduler steps since the
    start of the run, cycles restart every ``period`` steps."""
    __slots__ = ("period",)

    def __init__(self, period):
        self.period = int(period)

    def __call__(self, k):
        phase = (k % self.period) / self.period
        return 0.5 * (math.cos(math.pi * phase) + 1.)


def get_cosine_schedule(samples_per_cycle):
    return CosineSchedule(samples_per_cycle)
