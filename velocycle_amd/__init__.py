"""velocycle_amd: MI355X-native engine for VeloCycle's SVI hot path (phase / velocity inference).

The per-step ELBO + reparameterised gradient runs in hand-written HIP kernels for gfx950
(velocycle_amd/csrc -> libvelocycle_hip.so, C ABI in include/velocycle_hip.h); this package is the
Python host side mirroring the reference's `fit()` entry points.
"""
__version__ = "0.1.0"
