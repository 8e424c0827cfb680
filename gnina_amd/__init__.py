"""gnina_amd: MI355X-native implementation of gnina's CNN scoring hot path.

Host-side mirror of the reference's TorchModel / CNNScorer surface over a C-ABI HIP library
(include/mi_gnina.h).  There is no CPU fallback: importing `gnina_amd.capi` raises if the HIP
library has not been built (python __graft_entry__.py build).
"""
__version__ = "0.1.0"
