"""dgp_amd -- MI355X-native stochastic-imputation engine behind dgpsi's
dgp()/emulator()/train()/predict() API (hand-written HIP for gfx950 through a
ctypes C-ABI; see include/dgp_amd.h)."""
from .kernel_class import kernel, combine  # noqa: F401
from .imputation import imputer  # noqa: F401
from .dgp import dgp  # noqa: F401
from .emulation import emulator  # noqa: F401
from .gp import gp  # noqa: F401
from .linkgp import container, lgp  # noqa: F401
from .likelihood_class import Hetero, Poisson, NegBin, ZIP, ZINB, Categorical  # noqa: F401
from .synthetic import path  # noqa: F401
from .utils import nb_seed, set_thread, get_thread, write, read, summary, save_structure, load_structure  # noqa: F401

__version__ = '0.1.0'
