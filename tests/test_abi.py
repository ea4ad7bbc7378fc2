"""CPU: the C-ABI library loads, exports every symbol include/mfx.h declares, fails loudly
without a GPU, and its host-side scheduler matches the golden level order."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mfrec_amd import _shim as S
from mfrec_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    B.build(verbose=False)
    return S.lib()


def header_functions():
    text = open(os.path.join(ROOT, "include", "mfx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mfx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libmfx.so does not export %s" % n
    # and the binding table covers exactly the header
    assert sorted(S.SIGNATURES) == names


def test_fails_loudly_without_gpu(lib):
    if S.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(S.MfxError, match="no CPU fallback"):
        S.init(0)
    from mfrec_amd.lib import kmf_train
    u, v = np.zeros((4, 3)), np.zeros((4, 5))
    idx = np.array([[0, 0], [1, 2]], dtype=np.int32)
    with pytest.raises(S.MfxError):
        kmf_train.train_linear_kernel(1, 4, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0, u, v, idx,
                                      np.array([3.0, 4.0]), np.zeros(3), np.zeros(5))
    # every family: the Funk pass, ALS, the resident plan, the batched predict / top-N, the device queries
    from mfrec_amd.lib import gd_estimator, als_implicit
    from mfrec_amd.plan import SgdPlan
    from mfrec_amd.recommendation import KMFRecommender, metrics
    r = np.array([3.0, 4.0])
    with pytest.raises(S.MfxError):
        gd_estimator.estimator_loop_without_bias(1, 1, 1e9, 4, 0.1, 0.001, 0.05, u, v, idx, r, 5, 3)
    with pytest.raises(S.MfxError):
        als_implicit.als_wrmf(1, 4, u, v, None, None, np.array([0, 1, 1, 0, 0, 0], dtype=np.int32),
                              np.array([0, 2], dtype=np.int32), np.array([0, 1, 0, 1], dtype=np.int32),
                              np.array([0, 1], dtype=np.int32), 5, 3)
    with pytest.raises(S.MfxError):
        SgdPlan(idx, r, 5, 3, 4)
    R = KMFRecommender(5, 3, {"nbr_features": 4})
    R.set_ratings(idx, r)
    R.svd_u, R.svd_v, R.items_bias, R.users_bias, R.overall_bias = u, v, np.zeros(3), np.zeros(5), 3.5
    with pytest.raises(S.MfxError):
        metrics.predict_batch(R, idx, "predict_linear", r)
    with pytest.raises(S.MfxError):
        R.find_recommended_items_batch([0, 1], 2, "predict_linear")
    with pytest.raises(S.MfxError):
        S.check(lib.mfx_device_mem_info(None, None))


def test_buffer_validation_mimics_cython(lib):
    from mfrec_amd.lib import kmf_train
    u, v = np.zeros((4, 3)), np.zeros((4, 5))
    idx = np.array([[0, 0], [1, 2]], dtype=np.int32)
    r, ib, ub = np.array([3.0, 4.0]), np.zeros(3), np.zeros(5)
    a = (1, 4, 0.1, 0.01, 0, 0, 0.1, 0.1, 0.007, 0.0)
    with pytest.raises(ValueError, match="dtype mismatch"):
        kmf_train.train_linear_kernel(*a, u.astype(np.float32), v, idx, r, ib, ub)
    with pytest.raises(ValueError, match="dtype mismatch"):
        kmf_train.train_linear_kernel(*a, u, v, idx.astype(np.int64), r, ib, ub)
    with pytest.raises(ValueError, match="dimensions"):
        kmf_train.train_linear_kernel(*a, u, v, idx, r.reshape(2, 1), ib, ub)
    with pytest.raises(ValueError, match="contiguous"):
        kmf_train.train_linear_kernel(*a, np.zeros((3, 4)).T, v, idx, r, ib, ub)
    with pytest.raises(TypeError):
        kmf_train.train_linear_kernel(*a, [[0.0]], v, idx, r, ib, ub)


def _levels(lib, idx, nu, ni):
    lev = np.empty(idx.shape[0], dtype=np.int32)
    n = C.c_int64(0)
    S.check(lib.mfx_schedule_levels(S.i32(idx, "idx", 2), idx.shape[0], nu, ni,
                                    lev.ctypes.data_as(S._ip), C.byref(n)))
    perm = np.empty(idx.shape[0], dtype=np.int64)
    off = np.empty(n.value + 1, dtype=np.int64)
    S.check(lib.mfx_schedule_permutation(lev.ctypes.data_as(S._ip), idx.shape[0], n.value,
                                         perm.ctypes.data_as(S._lp), off.ctypes.data_as(S._lp)))
    return lev, perm, off


def test_scheduler_matches_golden_level_order(lib, golden):
    g = golden("g8_levels")
    for tag in ("a", "b"):
        idx = g["in_idx_" + tag]
        lev, perm, off = _levels(lib, idx, 60, 40)
        assert np.array_equal(lev, g["out_level_" + tag])
        assert np.array_equal(perm, g["out_perm_" + tag])
        assert off[0] == 0 and off[-1] == idx.shape[0]
        # inside a level no user and no item repeats
        for l in range(len(off) - 1):
            s = idx[perm[off[l]:off[l + 1]]]
            assert len(set(s[:, 0])) == len(s) and len(set(s[:, 1])) == len(s)


def test_scheduler_properties_random(lib):
    rs = np.random.RandomState(0)
    idx = np.stack([rs.randint(0, 300, 20000), rs.randint(0, 200, 20000)], 1).astype(np.int32)
    lev, perm, off = _levels(lib, idx, 300, 200)
    assert sorted(perm.tolist()) == list(range(20000))
    # order is preserved inside every user chain and every item chain
    pos = np.empty(20000, dtype=np.int64); pos[perm] = np.arange(20000)
    for col in (0, 1):
        order = np.lexsort((np.arange(20000), idx[:, col]))
        same = idx[order[1:], col] == idx[order[:-1], col]
        assert np.all(lev[order[1:]][same] > lev[order[:-1]][same])
    # empty input and range errors
    e = np.zeros((0, 2), dtype=np.int32)
    lev, perm, off = _levels(lib, e, 3, 3)
    assert off.tolist() == [0]
    bad = np.array([[0, 5]], dtype=np.int32)
    with pytest.raises(ValueError, match="outside"):
        _levels(lib, bad, 3, 3)


def test_threaded_pack_equals_permutation_gather(lib):
    """mfx_schedule_pack (what a plan uploads) == gathering with mfx_schedule_permutation, on a
    problem big enough (> 2^20 ratings) for the multi-threaded path."""
    rs = np.random.RandomState(5)
    n = (1 << 20) + 12345
    idx = np.stack([rs.randint(0, 20000, n), rs.randint(0, 9000, n)], 1).astype(np.int32)
    r = rs.randint(1, 6, n).astype(np.float64)
    lev, perm, off = _levels(lib, idx, 20000, 9000)
    rec = np.zeros(n, dtype=np.dtype([("user", np.int32), ("item", np.int32), ("r", np.float64)]))
    off2 = np.zeros(len(off), dtype=np.int64)
    S.check(lib.mfx_schedule_pack(S.i32(idx, "idx", 2), S.f64(r, "r", 1), lev.ctypes.data_as(S._ip), n,
                                  len(off) - 1, rec.ctypes.data_as(C.c_void_p), off2.ctypes.data_as(S._lp)))
    assert np.array_equal(off2, off)
    assert np.array_equal(rec["user"], idx[perm, 0]) and np.array_equal(rec["item"], idx[perm, 1])
    assert np.array_equal(rec["r"], r[perm])


def test_balanced_order_is_a_permutation_with_max_degree_levels(lib):
    from mfrec_amd.lib.datasets import balanced_order
    rs = np.random.RandomState(1)
    n, nu, ni = 200000, 3000, 1500
    idx = np.stack([rs.randint(0, nu, n), rs.randint(0, ni, n)], 1).astype(np.int32)
    perm, rounds = balanced_order(idx, nu, ni, seed=3)
    assert sorted(perm.tolist()) == list(range(n))
    maxdeg = max(np.bincount(idx[:, 0]).max(), np.bincount(idx[:, 1]).max())
    assert maxdeg <= rounds <= maxdeg + 8                       # first fit lands on (almost) Delta
    lev, _, off = _levels(lib, np.ascontiguousarray(idx[perm]), nu, ni)
    assert len(off) - 1 == rounds                                # the level schedule keeps the rounds
    plain, _, off2 = _levels(lib, idx, nu, ni)
    assert len(off2) - 1 > 1.5 * rounds                          # an i.i.d. order needs far more levels
    perm2, _ = balanced_order(idx, nu, ni, seed=3)
    assert np.array_equal(perm, perm2)                           # seeded, deterministic
    assert not np.array_equal(perm, balanced_order(idx, nu, ni, seed=4)[0])


def test_rccl_abi_slice_matches_the_real_header():
    """mfrec_amd/csrc/mfx_rccl_abi.h declares by hand the twelve RCCL entry points mfx_comm.hip binds with dlsym (libmfx.so
    neither links librccl nor needs its headers to build).  tests/rccl_abi_check.cpp static_asserts every pointer type, the
    enum values and the 128-byte id against <rccl/rccl.h> of this ROCm: if RCCL's ABI moves, this fails instead of a run."""
    import shutil
    import subprocess
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr) or not shutil.which("g++"):
        pytest.skip("no RCCL header / g++ here")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_abi_check.cpp")
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-3000:]
