"""harry_amd -- MI355X-native .hry hot path (attribute quantisation, prediction residuals, arithmetic coding).

`harry_amd.codec` is the host-side mirror of the reference's interface; all work happens in libharry_amd.so
(C ABI in include/harry_amd.h).  `harry_amd.meshgen` generates the synthetic benchmark meshes.
"""
__all__ = ["codec", "meshgen"]
