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
