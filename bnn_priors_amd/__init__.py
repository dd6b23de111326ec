"""bnn_priors_amd -- MI355X-native SG-MCMC leapfrog engine behind the
bnn_priors sampler API (SGLD / VerletSGLD / HMC + runners).  See DESIGN.md."""
from . import prior, models  # noqa: F401
