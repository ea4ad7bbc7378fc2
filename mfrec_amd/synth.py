"""Counter-based synthetic ratings (SURVEY.md section 8(d) "Synthetic inputs").

Every rating t in [0, nnz) is a pure function of (seed, t): two successive
splitmix64 outputs give the (user, item) pair -- uniform degrees, duplicates
kept -- and the value comes from a planted rank-8 model plus noise, rounded
and clipped to the reference's rating scale [1, 5]
(mfrec/recommendation/base.py:93-94 min_rating/max_rating).

The natural order t = 0..nnz-1 is an i.i.d. ("shuffled") order, i.e. one
realisation of get_ratings(randomize_order=True) (base.py:1127-1129);
`user_major()` gives the randomize_order=False order (COO of a lil_matrix:
users ascending, items ascending inside a user).
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

SEED_DATA = 20261003
SEED_INIT = 7
PLANTED_RANK = 8


def _mix64(z):
    """splitmix64 output function on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def splitmix64_pair(seed, t):
    """(h0, h1) = the two outputs of a splitmix64 stream whose state is seed + 2*t*GAMMA."""
    with np.errstate(over="ignore"):
        t = np.asarray(t, dtype=np.uint64)
        s0 = np.uint64(seed) + (np.uint64(2) * t + np.uint64(1)) * _GAMMA
        return _mix64(s0), _mix64(s0 + _GAMMA)


def _u01(h):
    """uint64 -> double in (0, 1): top 53 bits, never exactly 0."""
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _normal(seed, counter):
    """Standard normal per counter (Box-Muller on two hashed uniforms)."""
    h0, h1 = splitmix64_pair(seed, counter)
    return np.sqrt(-2.0 * np.log(_u01(h0))) * np.cos(2.0 * np.pi * _u01(h1))


def planted_factors(n, seed, rank=PLANTED_RANK, std=0.6):
    """f64 [n, rank] ~ N(0, std^2), row r / column j drawn from counter r*rank + j."""
    c = np.arange(n * rank, dtype=np.uint64)
    return (std * _normal(seed, c)).reshape(n, rank)


def make_ratings(nu, ni, nnz, seed=SEED_DATA, t0=0, user_base=0, nu_total=None, chunk=1 << 18,
                 workers=None):
    """Return (ratings_index int32[nnz,2] (user,item), ratings f64[nnz]).

    t0 / user_base let a rank generate only its own shard: rating counters
    t0..t0+nnz-1, users drawn uniformly from [user_base, user_base + nu).
    nu_total (default nu) is the size of the planted user-factor table.
    Chunks are independent (counter-based), so they are filled by a thread pool (numpy
    releases the GIL inside its loops); the result does not depend on `workers`.
    """
    import os
    nu_total = nu if nu_total is None else nu_total
    P = planted_factors(nu_total, seed + 1)
    Q = planted_factors(ni, seed + 2)
    idx = np.empty((nnz, 2), dtype=np.int32)
    r = np.empty(nnz, dtype=np.float64)

    def fill(a):
        b = min(nnz, a + chunk)
        t = np.arange(t0 + a, t0 + b, dtype=np.uint64)
        h0, h1 = splitmix64_pair(seed, t)
        user = (h0 % np.uint64(nu)).astype(np.int64) + user_base
        item = (h1 % np.uint64(ni)).astype(np.int64)
        noise = 0.5 * _normal(seed + 3, t)
        val = 3.5 + np.einsum("ij,ij->i", P[user], Q[item]) + noise
        idx[a:b, 0] = user
        idx[a:b, 1] = item
        r[a:b] = np.clip(np.rint(val), 1.0, 5.0)

    starts = list(range(0, nnz, chunk))
    if workers is None:
        workers = min(16, os.cpu_count() or 1)
    if workers <= 1 or len(starts) <= 1:
        for a in starts:
            fill(a)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(fill, starts))
    return idx, r


def user_major(idx, r):
    """Reorder to the reference's un-shuffled order (user asc, item asc; stable)."""
    order = np.lexsort((idx[:, 1], idx[:, 0]))
    return np.ascontiguousarray(idx[order]), np.ascontiguousarray(r[order])


def init_normal(k, n, seed, std=0.1, mean=0.0):
    """Feature-major f64 [k, n] ~ N(mean, std^2): the seeded stand-in for
    MFRecommender.init_feature_normal (mfrec/recommendation/mf.py:124-133)."""
    c = np.arange(k * n, dtype=np.uint64)
    return (mean + std * _normal(seed, c)).reshape(k, n)


def init_const(k, n, value=0.1):
    """zeros + feature_init (gradient_descent.py:524-525, wrmf.py:92-93)."""
    return np.zeros((k, n), dtype=np.float64) + value
