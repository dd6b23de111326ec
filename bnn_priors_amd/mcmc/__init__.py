from .sgld import SGLD, dot                # noqa: F401
from .verlet_sgld import VerletSGLD        # noqa: F401
from .hmc import HMC                       # noqa: F401
