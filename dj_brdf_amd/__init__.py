"""dj_brdf_amd -- MI355X-native (gfx950) batch engine behind the dj_brdf operator surface.

``from dj_brdf_amd import djb`` gives the reference's namespace (djb.ggx, djb.beckmann, djb.merl,
djb.utia, djb.tabular, djb.fresnel.*, djb.microfacet.params, djb.exc) with batched operators
running as hand-written HIP kernels through the C ABI of ``lib/libdjb_hip.so``
(``include/djb_hip.h``).  ``synth`` holds the bit-reproducible synthetic workloads.
"""
from . import _lib, synth  # noqa: F401
from . import djb  # noqa: F401

__all__ = ["djb", "synth", "_lib"]
__version__ = "0.1.0"
