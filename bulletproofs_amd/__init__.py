"""bulletproofs_amd: MI355X (gfx950) implementation of the l-adic/bulletproofs
hot path -- multi-scalar multiplication and the inner-product-argument vector
kernels -- behind the C ABI in include/bp_gpu.h.  Importing the package loads
libbp_gpu.so and fails if it has not been built."""
from .api import (  # noqa: F401
    IPA_BULLETPROOF,
    IPA_POLYCOMMIT,
    SECP256K1,
    VESTA,
    Bases,
    BpError,
    Context,
    IpaSession,
)
