"""WRMFRecommender: implicit-feedback ALS (reference: mfrec/recommendation/wrmf.py)."""
import numpy as np

from ..lib import als_implicit as _kernels
from ..lib.datasets import create_bool_sparse_col, create_bool_sparse_row
from .mf import MFRecommender


class WRMFRecommender(MFRecommender):
    PARAMETERS_INDEX = {"nbr_epochs": "nbr_epochs", "feature_init": "feature_init",     # :29-33
                        "regularization_model": "K", "neighborhood": "neighborhood",
                        "nbr_features": "dimensionality"}

    def __init__(self, nbr_users=4, nbr_items=6, parameters=None):
        MFRecommender.__init__(self, nbr_users, nbr_items)
        self.nbr_epochs = 20                                                  # :40-44
        self.feature_init = 0.1
        self.K = 0.025
        self.dimensionality = 20
        if parameters:
            self.set_parameters(parameters)

    def predict(self, item_index, user_index):                                # :67-69
        return np.dot(self.svd_u[:, item_index], self.svd_v[:, user_index])

    def train(self, initialize_model=True, handle_bias=False, verbose=False):
        """wrmf.py:83-110.  c_pos = 1 and lambda = 0.015 are hard-coded there (self.K is ignored)."""
        if initialize_model:
            self.init_feature_const()
        m = np.zeros([self.dimensionality, self.dimensionality])
        m_inv = np.zeros([self.dimensionality, self.dimensionality])
        users_row, users_col = create_bool_sparse_row(self.relationship_matrix)
        items_row, items_col = create_bool_sparse_col(self.relationship_matrix)
        self.compute_overall_avg()
        _kernels.als_wrmf(self.nbr_epochs, self.dimensionality, self.svd_u, self.svd_v, m, m_inv,
                          users_row, users_col, items_row, items_col, self.nbr_users,
                          self.nbr_items, c_pos=1, k=0.015, verbose=verbose)

    fit = train
