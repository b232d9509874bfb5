"""toycluster_amd -- MI355X-native SPH density / WVT relaxation path of Toycluster.

The compute lives in csrc/ (hand-written HIP for gfx950) behind the C ABI of
include/tcgpu.h; this package only holds the ctypes binding, the model/synthetic
input helpers and the build recipe.  There is no CPU fallback: importing
`toycluster_amd.binding` without a built libtcgpu.so raises.
"""
__all__ = ["model"]
