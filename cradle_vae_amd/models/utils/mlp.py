from ...mlp import *  # noqa: F401,F403
from ...mlp import get_likelihood_mlp  # noqa: F401
