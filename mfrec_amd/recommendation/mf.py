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

    def clamping(self, value, min=1.0, max=5.0):
        """mf.py:44-53: hard [1, 5] clamp (the min / max arguments are ignored there too)."""
        if value > 5.0:
            value = 5.0
        if value < 1.0:
            value = 1.0
        return value

    def predict_rating_by_label(self, user_label, item_label, predictor="predict_logistic"):
        """mf.py:67-75: predictor(item, user) by labels; an unknown label gives overall_avg."""
        try:
            return getattr(self, predictor)(self.items_index[item_label], self.users_index[user_label])
        except KeyError:
            return self.overall_avg

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

    def init_user_features(self, user_index, mean=0.0, std=0.1):           # mf.py:136-137
        self.svd_v[:, user_index] = np.random.normal(mean, std, self.dimensionality)

    def init_item_features(self, item_index, mean=0.0, std=0.1):           # mf.py:140-141
        self.svd_u[:, item_index] = np.random.normal(mean, std, self.dimensionality)

    def _grow(self, users=0, items=0):
        """One more user / item column in the model (folding-in: the ratings matrix is not
        touched, as in gradient_descent.py:908-929 / kmf.py:149-169)."""
        k = self.dimensionality
        if users:
            self.svd_v = np.ascontiguousarray(np.hstack([self.svd_v, np.zeros((k, users))]))
            if self.users_bias is not None:
                self.users_bias = np.concatenate([self.users_bias, np.zeros(users)])
            self._nbr_users += users
        if items:
            self.svd_u = np.ascontiguousarray(np.hstack([self.svd_u, np.zeros((k, items))]))
            if self.items_bias is not None:
                self.items_bias = np.concatenate([self.items_bias, np.zeros(items)])
            self._nbr_items += items

    def add_user(self, user_label, users_ratings_index, users_ratings):
        """Fold a new user in: returns its index.  users_ratings_index = item ids."""
        items = np.asarray(users_ratings_index)
        vals = np.asarray(users_ratings, dtype=np.float64)
        if items.shape[0] != vals.shape[0]:
            from .base import Error
            raise Error("The index and the ratings array must be the same size")
        self._grow(users=1)
        new_id = self._get_new_user_id()               # kmf.py:156-158: default label, then the caller's
        self.users_index[user_label] = new_id
        self.users_label[new_id] = user_label
        idx = np.zeros([vals.shape[0], 2], dtype=np.int32)
        idx[:, 0] = new_id
        idx[:, 1] = items
        self.retrain_user(new_id, idx, vals)
        return new_id

    def add_item(self, item_label, items_ratings_index, items_ratings):
        """Fold a new item in: returns its index.  items_ratings_index = user ids."""
        users = np.asarray(items_ratings_index)
        vals = np.asarray(items_ratings, dtype=np.float64)
        if users.shape[0] != vals.shape[0]:
            from .base import Error
            raise Error("The index and the ratings array must be the same size")
        self._grow(items=1)
        new_id = self._get_new_item_id()
        self.items_index[item_label] = new_id
        self.items_label[new_id] = item_label
        idx = np.zeros([vals.shape[0], 2], dtype=np.int32)
        idx[:, 0] = users
        idx[:, 1] = new_id
        self.retrain_item(new_id, idx, vals)
        return new_id

    def find_recommended_items(self, user_index=None, user_label=None, nbr_recommendations=5,
                               output_label=False, predictor="predict"):
        """Top-N items for a user (mf.py:144-193) -> (items, scores), two lists.  As in the reference: the
        candidates are items 0 .. neighborhood-1 (it draws a random subset with get_items_subset and then
        indexes by position, mf.py:159-163 -- the draw is kept so that a seeded RNG stream stays aligned), the
        user's rated items AND the item whose id equals user_index (mf.py:162) score 0, NaN scores 0, zero
        scores are dropped, the rest goes by descending score (ties: ascending item).  The scoring itself is
        one batched GPU pass instead of a Python loop over predict()."""
        if user_index is None:
            user_index = self.users_index[user_label]
        self.neighborhood = min([self.neighborhood, self.nbr_items])
        n = int(self.neighborhood)
        self.get_items_subset(count=n)
        top, scores = self._top_items(user_index, n, predictor, nbr_recommendations)
        return ([self.items_label[i] for i in top] if output_label else top), scores

    def find_recommended_items_batch(self, users, nbr_recommendations=5, predictor="predict"):
        """Extension: find_recommended_items for MANY users in one device call -> (items int32[n, N], -1 where
        fewer than N items qualify; scores f64[n, N]).  Row j equals find_recommended_items(users[j]) -- same
        candidates (items 0 .. neighborhood-1), exclusions, order and score bits; the per-call RNG draw of the
        single-user method (which does not influence its result) is not made."""
        from .metrics import topn_batch
        self.neighborhood = min([self.neighborhood, self.nbr_items])
        return topn_batch(self, users, int(self.neighborhood), nbr_recommendations, predictor)

    def _top_items(self, user_index, n, predictor, nbr_recommendations):
        """Score items 0 .. n-1 for the user in one batched GPU pass and rank them the reference's way
        (mf.py:161-191, gradient_descent.py:774-800)."""
        from .metrics import predict_batch
        idx = np.zeros([n, 2], dtype=np.int32)
        idx[:, 0] = user_index
        idx[:, 1] = np.arange(n)
        user_ratings, _ = predict_batch(self, idx, predictor, one_user=int(user_index))
        rel = self.relationship_matrix
        if user_index < rel.shape[0]:              # a folded-in user has no row in the ratings matrix
            row = rel[user_index, :].tocoo()
            rated = row.col[row.data != 0]
            user_ratings[rated[rated < n]] = 0.0
        if 0 <= user_index < n:
            user_ratings[user_index] = 0.0
        user_ratings[np.isnan(user_ratings)] = 0.0
        nz = user_ratings.nonzero()[0]
        top = nz[np.argsort(-user_ratings[nz], kind="stable")][:nbr_recommendations]
        return [int(i) for i in top], [float(user_ratings[i]) for i in top]

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
