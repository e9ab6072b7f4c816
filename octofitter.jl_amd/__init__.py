"""
octofitter.jl_amd — MI355X-native epoch-loop likelihood path for Octofitter.jl.

The directory name carries a dot (it is the reference's name), so it is loaded by path:
`from __graft_entry__ import load_package; pkg = load_package()` registers it as the module
`octofitter_jl_amd`. Contents: csrc/ (HIP kernels + C ABI), host/ (Python mirror of the
reference's observation / system / ln_like surface over ctypes), julia/ (the ccall shim a
maintainer would drop into Octofitter.jl).
"""
from .host import capi  # noqa: F401
from .host.observations import (  # noqa: F401
    PlanetRelAstromObs, PlanetRelAstromLikelihood, StarAbsoluteRVObs, StarAbsoluteRVLikelihood,
    MarginalizedStarAbsoluteRVObs, PlanetRelativeRVObs, PlanetRelativeRVLikelihood, ObsPriorAstromONeil2019,
    HGCAInstantaneousObs, HGCAInstantaneousLikelihood,
)
from .host.system import Planet, System, make_ln_like, BatchedLnLike, accelerate, not_on_device  # noqa: F401
from .host.sharding import shard_range, ShardedLnLike  # noqa: F401,E402
from .host.tempering import TemperedSwap  # noqa: F401,E402
from .host.ofti import OftiLinearSolver, ofti_linear_solve  # noqa: F401,E402
from .host.priors import (Uniform, LogUniform, Normal, TruncatedNormal, truncated, Sine, UniformCircular,  # noqa: F401,E402
                          θ_at_epoch_to_tperi, variables)
from .host.model import LogDensityModel  # noqa: F401,E402
from .host.callers import guess_starting_position, octofit_rejection, rejection_evaluate_likelihoods, pointwise_like  # noqa: F401,E402
