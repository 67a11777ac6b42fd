"""MI355X-native many-chain NUTS hot path behind DynamicHMC.jl's calling surface.

Import name: the directory is `dynamichmc.jl_amd` (not a valid dotted module name), so it is
loaded under the alias `dynamichmc_jl_amd` by `__graft_entry__.load_package()`.
"""
from . import _abi as abi  # noqa: F401
from . import diagnostics, sharding  # noqa: F401
from .api import *  # noqa: F401,F403
from .context import DeviceContext, DynamicHMCError  # noqa: F401
