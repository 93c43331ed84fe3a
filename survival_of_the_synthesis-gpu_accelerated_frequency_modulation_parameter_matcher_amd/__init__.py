"""MI355X-native evolutionary FM sound matcher: the HIP backend of the reference's
Evolutionary_Strategy API.  The compute path is libsots_hip.so (csrc/, gfx950 kernels
behind the C-ABI of include/sots_hip.h); this package only binds it."""
from . import capi, island
from .capi import HipES, HipGroup, SotsError

__all__ = ["capi", "island", "HipES", "HipGroup", "SotsError"]
