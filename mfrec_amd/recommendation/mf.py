"""Matrix-factorisation base: factor init and the BellKor bias pre-compute
(reference: mfrec/recommendation/mf.py:78-133), vectorised instead of per-row loops."""
import numpy as np
import scipy.sparse as sp

from .base import BaseRecommender


class MFRecommender(BaseRecommender):
    def __init__(self, nbr_users=4, nbr_items=6, parameters=None):
        BaseRecommender.__init__(self, nbr_users, nbr_items, parameters)
        self.neighborhood = 500
        self.K2 = self.K3 = 0.01

    def _coo(self):
        coo = self.relationship_matrix.tocoo()
        keep = coo.data != 0
        return coo.row[keep], coo.col[keep], coo.data[keep].astype(np.float64)

    def compute_items_bias_bk(self):
        """b_i = sum_{u in R(i)} (r_ui - mu) / (K3 + |R(i)|)   (mf.py:78-98)."""
        if not self.overall_bias:
            self.compute_overall_avg()
        _, cols, vals = self._coo()
        n = np.bincount(cols, minlength=self.nbr_items)
        s = np.bincount(cols, weights=vals - self.overall_bias, minlength=self.nbr_items)
        self.items_bias = np.where(n > 0, s / (self.K3 + np.maximum(n, 1)), 0.0)
        self.items_bias[np.isnan(self.items_bias)] = 0.0

    def compute_users_bias_bk(self):
        """b_u = sum_{i in R(u)} (r_ui - mu - b_i) / (K2 + |R(u)|)   (mf.py:101-121)."""
        if not self.overall_bias:
            self.compute_overall_avg()
        if self.items_bias is None:
            self.compute_items_bias_bk()
        rows, cols, vals = self._coo()
        n = np.bincount(rows, minlength=self.nbr_users)
        s = np.bincount(rows, weights=vals - self.overall_bias - self.items_bias[cols],
                        minlength=self.nbr_users)
        self.users_bias = np.where(n > 0, s / (self.K2 + np.maximum(n, 1)), 0.0)
        self.users_bias[np.isnan(self.users_bias)] = 0.0

    def init_feature_normal(self, mean=0.0, std=0.1):
        """Global-RNG normal init, items first then users (mf.py:124-133)."""
        self.svd_u = np.random.normal(mean, std, [self.dimensionality, self.nbr_items])
        self.svd_v = np.random.normal(mean, std, [self.dimensionality, self.nbr_users])

    def init_feature_const(self):
        """zeros + feature_init (gradient_descent.py:524-525, wrmf.py:92-93)."""
        self.svd_v = np.zeros([self.dimensionality, self.nbr_users]) + self.feature_init
        self.svd_u = np.zeros([self.dimensionality, self.nbr_items]) + self.feature_init

    def train_rmse(self, predictor="predict"):
        """sqrt(mean((r - predict(item, user))^2)) over the training ratings
        (formula of metrics.py:69-73), evaluated on the GPU."""
        from .metrics import rmse_on
        idx, r = self.get_ratings(False)
        return rmse_on(self, idx, r, predictor)
