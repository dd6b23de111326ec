from .base import *          # noqa: F401,F403
from .layers import *        # noqa: F401,F403
from .nets import *          # noqa: F401,F403
from .prior_only import *    # noqa: F401,F403
from .factory import *       # noqa: F401,F403
