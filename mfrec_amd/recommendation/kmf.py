"""KMFRecommender: all-k SGD with learned biases (reference: mfrec/recommendation/kmf.py)."""
import numpy as np

from ..lib import kmf_train as _kernels
from .mf import MFRecommender


class KMFRecommender(MFRecommender):
    # kmf.py:32-41 maps the three regularisation names to K/K2/K3 while train() reads
    # K_users/K_items/K_bias (kmf.py:54-56,219): in the reference those three public
    # parameters are therefore inert.  Reproduced as is; the *_direct names are ours.
    PARAMETERS_INDEX = {"nbr_epochs": "nbr_epochs", "min_improvement": "min_improvement",
                        "feature_init": "feature_init", "learning_rate": "learning_rate",
                        "learning_rate_users": "learning_rate_users",
                        "learning_rate_items": "learning_rate_items",
                        "regularization_users": "K", "regularization_items": "K2",
                        "regularization_bias": "K3", "nbr_features": "dimensionality",
                        "K_users_direct": "K_users", "K_items_direct": "K_items",
                        "K_bias_direct": "K_bias"}

    def __init__(self, nbr_users=4, nbr_items=6, parameters=False, filename=False):
        MFRecommender.__init__(self, nbr_users, nbr_items)
        self.nbr_epochs = 200                       # kmf.py:48-56
        self.feature_init = 0.1
        self.learning_rate = self.learning_rate_users = self.learning_rate_items = 0.01
        self.K_users, self.K_items, self.K_bias = 0.1, 0.1, 0.007
        self.dimensionality = 40
        if parameters:
            self.set_parameters(parameters)

    def _dot_bias(self, item_index, user_index):
        s = np.dot(self.svd_u[:, item_index], self.svd_v[:, user_index])
        return s + (self.items_bias[item_index] + self.users_bias[user_index])

    def predict_logistic(self, item_index, user_index):                      # kmf.py:79-85
        s = self._dot_bias(item_index, user_index)
        return self.min_rating + (1.0 / (1.0 + np.exp(-s))) * (self.max_rating - self.min_rating)

    def predict_linear(self, item_index, user_index):                        # kmf.py:88-94
        return self._dot_bias(item_index, user_index)

    def predict_linear_neg(self, item_index, user_index):                    # kmf.py:97-103
        return self.min_rating + self._dot_bias(item_index, user_index) * (self.max_rating - self.min_rating)

    predict = predict_logistic                                               # kmf.py:106

    def train(self, initialize_model=True, verbose=False, kernel="train_logistic_kernel", order=True):
        """kmf.py:197-220: N(0, 0.1) init, shuffled ratings, zero biases, one kernel call."""
        if kernel not in ("train_logistic_kernel", "train_linear_kernel"):
            raise NameError("name '%s' is not defined" % kernel)     # the reference eval()s the name
        self._mark_csc_due()                         # kmf.py:207 (relationship_matrix_csc: formed on first use)
        if initialize_model:
            self.init_feature_normal(0.0, 0.1)
        # order=True: the reference's shuffle; order="balanced": fewest dependency levels (extension); order="blocks": the
        # reference's shuffle, then grouped by (user block, item block) -- for problems whose smaller side does not fit the register
        # file of one GPU (sharding.kmf_train_block_order: every block on the static resident-side epoch; exact on that order)
        blocks = order == "blocks"
        ratings_index, ratings = self.get_ratings(randomize_order=True if blocks else order)
        self.compute_overall_avg()
        self.items_bias = np.zeros(self.nbr_items)
        self.users_bias = np.zeros(self.nbr_users)
        if blocks:
            from ..sharding import kmf_train_block_order
            self.block_order = kmf_train_block_order(1 if kernel == "train_logistic_kernel" else 0, self.nbr_epochs, self.dimensionality,
                                                     self.learning_rate, self.K_users, self.K_items, self.K_bias, self.svd_u, self.svd_v,
                                                     ratings_index, ratings, self.items_bias, self.users_bias)
            return
        getattr(_kernels, kernel)(self.nbr_epochs, self.dimensionality, self.feature_init,
                                  self.learning_rate, self.learning_rate_users,
                                  self.learning_rate_items, self.K_users, self.K_items, self.K_bias,
                                  self.overall_bias, self.svd_u, self.svd_v, ratings_index, ratings,
                                  self.items_bias, self.users_bias, 1, 1, int(verbose))

    fit = train

    def retrain_user(self, user_index, ratings_index, ratings, verbose=False,
                     kernel="train_logistic_kernel"):
        """Folding-in (kmf.py:120-131): only this user's row is updated (gates 1, 0)."""
        sel = np.where(ratings_index[:, 0] == user_index)[0]
        self.init_user_features(user_index)
        getattr(_kernels, kernel)(self.nbr_epochs, self.dimensionality, self.feature_init,
                                  self.learning_rate, self.learning_rate_users,
                                  self.learning_rate_items, self.K_users, self.K_items, self.K_bias,
                                  self.overall_bias, self.svd_u, self.svd_v,
                                  np.ascontiguousarray(ratings_index[sel, :]),
                                  np.ascontiguousarray(ratings[sel]), self.items_bias,
                                  self.users_bias, 1, 0, int(verbose))

    def retrain_item(self, item_index, ratings_index, ratings, verbose=False,
                     kernel="train_logistic_kernel"):
        """Folding-in of an item (kmf.py:134-146): only this item's row is updated (gates 0, 1)."""
        sel = np.where(ratings_index[:, 1] == item_index)[0]
        self.init_item_features(item_index)
        getattr(_kernels, kernel)(self.nbr_epochs, self.dimensionality, self.feature_init,
                                  self.learning_rate, self.learning_rate_users,
                                  self.learning_rate_items, self.K_users, self.K_items, self.K_bias,
                                  self.overall_bias, self.svd_u, self.svd_v,
                                  np.ascontiguousarray(ratings_index[sel, :]),
                                  np.ascontiguousarray(ratings[sel]), self.items_bias,
                                  self.users_bias, 0, 1, int(verbose))
