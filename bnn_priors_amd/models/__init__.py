from .base import *          # noqa: F401,F403
from .nets import *          # noqa: F401,F403
from .targets import *       # noqa: F401,F403
from .factory import *       # noqa: F401,F403
