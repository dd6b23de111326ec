from .base import *          # noqa: F401,F403
from .loc_scale import *     # noqa: F401,F403
