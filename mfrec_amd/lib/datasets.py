"""Run-length row/column views of a boolean sparse matrix: the ALS kernel's input format
(counterpart of mfrec/lib/datasets.py:13-32)."""
import numpy as np
import scipy.sparse as sp


def _run_length(csr):
    csr = sp.csr_matrix(csr, copy=True)
    csr.eliminate_zeros()
    csr.sort_indices()
    lengths = np.diff(csr.indptr).astype(np.int32)
    # the reference counts with np.bincount(rows): the array stops at the last non-empty row
    last = int(np.flatnonzero(lengths)[-1]) + 1 if csr.nnz else 0
    head = np.zeros(last + 1, dtype=np.int32)
    head[1:] = lengths[:last]
    return head, csr.indices.astype(np.int32)


def create_bool_sparse_row(sparse_matrix):
    """-> (int32 [0, n_0, n_1, ...] up to the last rated row, int32 column ids row by row)."""
    return _run_length(sp.csr_matrix(sparse_matrix))


def create_bool_sparse_col(sparse_matrix):
    """Same on the transpose: per item, the users that rated it."""
    return _run_length(sp.csr_matrix(sparse_matrix).T)


def balanced_order_gpu(ratings_index, nbr_users, nbr_items):
    """The GPU builder of the same kind of order (mfx_schedule_balanced_order_gpu): the dependency
    levels of the GIVEN order, first-fit recoloured level by level; milliseconds instead of
    seconds.  Deterministic in the input order: shuffle first for a random one."""
    import ctypes as C
    from .. import _shim as S
    idx = np.ascontiguousarray(ratings_index, dtype=np.int32)
    perm = np.empty(idx.shape[0], dtype=np.int64)
    n = C.c_int64(0)
    S.check(S.lib().mfx_schedule_balanced_order_gpu(S.i32(idx, "ratings_index", 2), idx.shape[0],
                                                    int(nbr_users), int(nbr_items),
                                                    perm.ctypes.data_as(S._lp), C.byref(n)))
    return perm, n.value


def balanced_order(ratings_index, nbr_users, nbr_items, seed=0):
    """Permutation of the ratings with (close to) the minimum number of dependency levels for
    the exact GPU schedule: a seeded first-fit edge colouring, round-major (libmfx host code,
    mfx_schedule_balanced_order).  A legitimate stand-in for the reference's random shuffle
    (base.py:1127-1129).  Raises MfxError if the degree distribution is too skewed."""
    import ctypes as C
    from .. import _shim as S
    idx = np.ascontiguousarray(ratings_index, dtype=np.int32)
    perm = np.empty(idx.shape[0], dtype=np.int64)
    n = C.c_int64(0)
    S.check(S.lib().mfx_schedule_balanced_order(S.i32(idx, "ratings_index", 2), idx.shape[0],
                                                int(nbr_users), int(nbr_items), int(seed) & (2 ** 63 - 1),
                                                perm.ctypes.data_as(S._lp), C.byref(n)))
    return perm, n.value
