from .base import *          # noqa: F401,F403
from .loc_scale import *     # noqa: F401,F403
from .transformed import *   # noqa: F401,F403
from .hierarchical import *  # noqa: F401,F403
from .mixture import *       # noqa: F401,F403
from .distributions import GeneralizedNormal  # noqa: F401
