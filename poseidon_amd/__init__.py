"""poseidon_amd: MI355X-native min-cost max-flow scheduling solver for the
Poseidon/Firmament scheduling round (host-side Python mirror over libpsn.so)."""
from ._lib import load_library, PsnError  # noqa: F401
from .network import Cluster, Network, synth_cluster, build_network  # noqa: F401
from .solver import Solver, SolveResult  # noqa: F401
from .service import SchedulerService, SchedulingDelta  # noqa: F401
