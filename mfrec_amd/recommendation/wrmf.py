"""WRMFRecommender -- implicit-feedback ALS on the GPU.

Counterpart of mfrec/recommendation/wrmf.py (Hu, Koren, Volinsky: "Collaborative filtering for
implicit feedback datasets", ICDM 2008): same constructor, parameter names, defaults,
train()/predict() surface; the solver is libmfx's MFMA kernel instead of the Cython loop."""
import numpy as np

from ..lib.als_implicit import als_wrmf, als_wrmf_sharded
from ..lib.datasets import create_bool_sparse_col, create_bool_sparse_row
from .mf import MFRecommender

# confidence weight and ridge term the reference hard-codes at its call site (wrmf.py:107-110);
# the class attribute K ("regularization_model") is accepted and, as there, never reaches the solver
C_POS = 1
LAMBDA = 0.015


class WRMFRecommender(MFRecommender):
    PARAMETERS_INDEX = dict(nbr_features="dimensionality", nbr_epochs="nbr_epochs",
                            feature_init="feature_init", regularization_model="K",
                            neighborhood="neighborhood")                      # wrmf.py:29-33

    def __init__(self, nbr_users=4, nbr_items=6, parameters=None):
        MFRecommender.__init__(self, nbr_users, nbr_items)
        self.dimensionality, self.nbr_epochs = 20, 20                         # wrmf.py:40-44
        self.feature_init, self.K = 0.1, 0.025
        if parameters:
            self.set_parameters(parameters)

    def predict(self, item_index, user_index):
        """Preference score = plain dot product of the two factor columns (wrmf.py:67-69)."""
        return np.dot(self.svd_u[:, item_index], self.svd_v[:, user_index])

    def predict_rating_by_label(self, user_label, item_label):
        """wrmf.py:72-80: the dot product by labels; an unknown label scores 0.0."""
        try:
            return np.dot(self.svd_u[:, self.items_index[item_label]], self.svd_v[:, self.users_index[user_label]])
        except KeyError:
            return 0.0

    def _run_length_views(self):
        rows = create_bool_sparse_row(self.relationship_matrix)
        cols = create_bool_sparse_col(self.relationship_matrix)
        return rows + cols

    def train(self, initialize_model=True, handle_bias=False, verbose=False, sharded=False):
        """One als_wrmf call over `nbr_epochs` user/item half-passes (wrmf.py:83-110).
        sharded=True: this process is one rank of an initialised RCCL communicator."""
        self._mark_csc_due()                  # wrmf.py:88 (relationship_matrix_csc: formed on first use)
        if initialize_model:
            self.init_feature_const()
        # wrmf.py:98-99 also rebuilds the rated-feedback tables here although the ALS solver never reads them (two
        # Python loops over every rating); they are built on demand by initialize_rated_feedback()
        self.items_feedback = np.zeros([self.dimensionality, self.nbr_items])
        k = self.dimensionality
        scratch = np.zeros([k, k])            # the reference's m / m_inv arguments (unused here)
        users_row, users_col, items_row, items_col = self._run_length_views()
        self.compute_overall_avg()
        solver = als_wrmf_sharded if sharded else als_wrmf
        solver(self.nbr_epochs, k, self.svd_u, self.svd_v, scratch, scratch.copy(), users_row,
               users_col, items_row, items_col, self.nbr_users, self.nbr_items, c_pos=C_POS, k=LAMBDA,
               verbose=verbose)

    fit = train
