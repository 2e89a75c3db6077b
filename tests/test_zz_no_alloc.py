"""sac_build_table_noalloc: the entry point that allocates nothing (tau and U staged in the library's file-scope
device array instead of two hipMalloc'ed buffers) against sac_build_table, through the C ABI, bit for bit, on two
calls in a row -- the second with a smaller input, so that whatever the first call left in the shared scratch array
must not leak into it.  Collected last (test_zz_*): a failure here must not hide the other GPU tests."""
import os

import numpy as np
import pytest

from sac_amd.lib import SacError, SacHandle
from sac_amd.synth import synthetic_timedata
from sac_amd.tin import read_tin

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _both(td, kind, n_grid, i_first, d_omega, which=0, peak=False, **spec):
    kw = dict(kind=1, n_peak=1, bosonic=int(kind == "bosonic")) if peak else {}
    wi_0 = 0 if peak else i_first
    out = []
    for noalloc in (False, True):
        h = SacHandle(td.n_tau, 40, n_grid, wi_0, wi_0 + n_grid - 1, **kw)
        h.build_table(which, kind, td.tau, td.beta, td.cov, i_first, d_omega, noalloc=noalloc, **spec)
        out.append(h.get_table(which))
        h.close()
    return out


def test_noalloc_table_build_equals_the_allocating_one():
    big = synthetic_timedata(n_tau=128, seed=5, n_boot=500)                  # N_tau = 128: the scratch array's maximum
    a, b = _both(big, "bosonic", 5001, 0, 2e-3)
    assert np.array_equal(a, b) and np.isfinite(a).all() and np.abs(a).max() > 0.1
    small = read_tin(f"{GOLD}/cv/t_001a.in")                                 # N_tau = 22, right after the large call
    a, b = _both(small, "finiteT", 3003, -1000, 1e-3, mirror=1)
    assert np.array_equal(a, b) and np.abs(a).max() > 0.1
    # second table of a two-axis peak handle (Kn: negated, half-offset grid), N_tau = 64
    mid = synthetic_timedata(n_tau=64, seed=6, n_boot=500, kernel_type="finiteT")
    a, b = _both(mid, "finiteT", 2049, 0, 5e-3, which=1, peak=True, offset=0.5, negate=1)
    assert np.array_equal(a, b) and np.abs(a).max() > 1e-3
    # and the table is the one the host builds (row scale 1e-13), i.e. the stale tail of the scratch array was not read
    from sac_amd.synth import kernel_values
    ref = kernel_values("finiteT", -(np.arange(2049) + 0.5) * 5e-3, mid.tau, mid.beta) @ mid.cov
    assert (np.abs(b - ref) / (np.abs(ref).max(axis=1, keepdims=True) + 1e-300)).max() < 1e-13     # rows far out underflow to 0


def test_noalloc_argument_errors():
    td = read_tin(f"{GOLD}/cv/t_001a.in")
    h = SacHandle(td.n_tau, 40, 1001, 0, 1000)
    with pytest.raises(SacError, match="does not exist"):
        h.build_table(1, "bosonic", td.tau, td.beta, td.cov, 0, 1e-3, noalloc=True)
    bad = td.tau.copy()
    bad[3] = np.nan
    with pytest.raises(SacError, match="non-finite"):
        h.build_table(0, "bosonic", bad, td.beta, td.cov, 0, 1e-3, noalloc=True)
    h.build_table(0, "bosonic", td.tau, td.beta, td.cov, 0, 1e-3, noalloc=True)
    h.upload_data(td.G, td.sigma_inv)            # a table built this way completes a problem like any other
    h.close()


def _hip_free_bytes():
    """hipMemGetInfo through the HIP runtime libsac_hip.so itself is linked against (found in this process's maps: a second
    runtime -- torch's bundled one -- cannot open the GPU beside it)."""
    import ctypes
    path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)
    hip = ctypes.CDLL(path)
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_noalloc_call_leaves_free_device_memory_unchanged():
    """The point of the entry: no device allocation.  hipMemGetInfo before and after calls on a warmed-up handle (the first
    call may load code objects) must read the same number of free bytes, for the largest supported N_tau; the allocating
    entry point is not required to (and is not checked)."""
    big = synthetic_timedata(n_tau=128, seed=5, n_boot=500)
    h = SacHandle(big.n_tau, 40, 4001, 0, 4000)
    h.build_table(0, "bosonic", big.tau, big.beta, big.cov, 0, 2e-3, noalloc=True)      # warm-up
    free0 = _hip_free_bytes()
    for _ in range(3):
        h.build_table(0, "bosonic", big.tau, big.beta, big.cov, 0, 2e-3, noalloc=True)
    free1 = _hip_free_bytes()
    assert free1 == free0, (free0, free1)
    h.close()
