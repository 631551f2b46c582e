"""platypus_amd -- MI355X-native (gfx950 HIP) drop-in for Platypus' read->haplotype likelihood and
local-assembly hot path.  See DESIGN.md; the C ABI is include/platypus_mi355x.h."""
from . import _lib                                    # noqa: F401
from .batch import HostBatch                          # noqa: F401

__all__ = ["_lib", "HostBatch"]
