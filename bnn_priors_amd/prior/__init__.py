from .base import *          # noqa: F401,F403
from .loc_scale import *     # noqa: F401,F403
from .transformed import *   # noqa: F401,F403
from .hierarchical import *  # noqa: F401,F403
from .correlated import *   # noqa: F401,F403
from .empirical_bayes import *  # noqa: F401,F403
from .mixture import *       # noqa: F401,F403
from .distributions import DoubleGamma as DoubleGammaDistribution, GeneralizedNormal  # noqa: F401
