import math
from typing import Callable


def get_cosine_schedule(samples_per_cycle: int) -> Callable[[int], float]:
    "lr factor 0.5 (cos(pi * (i mod S) / S) + 1)   (reference: bnn_priors/utils.py:5-10)"
    def schedule(i: int) -> float:
        progress = (i % samples_per_cycle) / samples_per_cycle
        return 0.5 * (math.cos(math.pi * progress) + 1.)
    return schedule
