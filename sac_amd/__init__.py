"""sac_amd -- MI355X-native sampling engine for Stochastic Analytic Continuation.

Host-side mirror of the reference's sampler interface (gabes135/sac, sac/free/sac_free.jl)
on top of the C ABI of libsac_hip.so (include/sac_hip.h).  There is no CPU fallback: every
compute call goes through the HIP library and raises if it is missing.
"""
from .lib import SacError, SacHandle, load_library  # noqa: F401
