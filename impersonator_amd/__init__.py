"""impersonator_amd -- MI355X-native implementation of the Liquid Warping GAN inference hot path
(`Imitator.forward()` of svip-lab/impersonator) behind the reference's own Python API.

Device work goes through liblwg.so (hand-written HIP for gfx950, C ABI in include/lwg.h).  There is no
CPU fallback: importing `impersonator_amd._lib` without the built library raises.
"""
__version__ = "0.1.0"
