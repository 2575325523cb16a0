"""emdee.jl_amd -- MI355X-native nonbonded hot path behind EmDee.jl's operator API.

Host-side mirror (Python over ctypes) of the reference's src/lennard_jones.jl, src/nonbonded.jl and
src/cells.jl; the Julia binding of the same C ABI is under julia/ (see INTEGRATION.md).  All compute
is hand-written HIP in libemdee_hip.so (csrc/).  There is no CPU path: importing this package
without the built library raises.

The directory name contains a dot, so import it through `__graft_entry__.load_package()` (alias
`emdee_jl_amd`).
"""
from . import _lib

_lib.load()   # fail loudly if the HIP library is missing

from ._lib import CUTOFF, LITERAL, EmDeeError                                   # noqa: E402
from .cells import Cells, update_cells_                                          # noqa: E402
from .device import Context, context_for, cu, gpu_available, to_host           # noqa: E402
from .lennard_jones import (LJAtom, LennardJonesAtom, LennardJonesModel,        # noqa: E402
                            interaction, lennard_jones_atoms)
from .nonbonded import (ENERGIES, FORCES, VIRIALS, WAVESIZE, AllPairsTiles,     # noqa: E402
                        NeighborTiles, Val, compute_nonbonded_, naively_compute_nonbonded_,
                        nonbonded_computation_tiles)
from .verlet import VelocityVerlet                                              # noqa: E402
from . import synthetic                                                         # noqa: E402
from . import domain                                                            # noqa: E402
from . import ingest                                                            # noqa: E402
from . import dd                                                                # noqa: E402
from .dd import DomainDecomposition                                             # noqa: E402

__all__ = ["LennardJonesModel", "LennardJonesAtom", "LJAtom", "lennard_jones_atoms", "interaction",
           "FORCES", "ENERGIES", "VIRIALS", "Val", "nonbonded_computation_tiles", "compute_nonbonded_",
           "naively_compute_nonbonded_", "NeighborTiles", "AllPairsTiles", "Cells", "update_cells_",
           "VelocityVerlet", "cu", "to_host", "context_for", "Context", "gpu_available", "synthetic",
           "EmDeeError", "LITERAL", "CUTOFF", "WAVESIZE", "domain", "ingest", "dd", "DomainDecomposition"]
