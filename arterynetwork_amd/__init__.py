"""arterynetwork_amd - MI355X-native variational region growing (the hot path of zjx1805/ArteryNetwork).

Only what that path needs: csrc/ (HIP kernels + C-ABI, include/vrg.h), the ctypes binding, the Python
mirror of the reference function, synthetic phantoms, and the Z-slab multi-GPU driver.
"""
from .variationalRegionGrowing import variationalRegionGrowing  # noqa: F401

__all__ = ['variationalRegionGrowing']
