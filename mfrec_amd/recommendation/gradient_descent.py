"""GDRecommender: Funk-style per-feature SGD (reference: mfrec/recommendation/gradient_descent.py)."""
import numpy as np

from ..lib import gd_estimator as _kernels
from .mf import MFRecommender


class GDRecommender(MFRecommender):
    PARAMETERS_INDEX = {"min_epochs": "min_epochs", "max_epochs": "max_epochs",       # :60-70
                        "min_improvement": "min_improvement", "feature_init": "feature_init",
                        "learning_rate": "learning_rate",
                        "learning_rate_users": "learning_rate_users",
                        "learning_rate_items": "learning_rate_items",
                        "regularization_model": "K", "regularization_users_bias": "K2",
                        "regularization_items_bias": "K3", "nbr_features": "dimensionality"}

    def __init__(self, nbr_users=4, nbr_items=6, parameters=False, filename=False):
        MFRecommender.__init__(self, nbr_users, nbr_items)
        self.min_epochs = self.max_epochs = 275                               # :77-87
        self.min_improvement = 0.0001
        self.feature_init = 0.1
        self.learning_rate = self.learning_rate_users = self.learning_rate_items = 0.001
        self.K, self.K2, self.K3 = 0.05, 0.01, 0.01
        self.dimensionality = 40
        if parameters:
            self.set_parameters(parameters)

    def feature_training(self, initialize_model=True, handle_bias=False, verbose=False, order=True):
        """gradient_descent.py:506-542.  order=True: the reference's shuffle; order="balanced"
        (extension): a random order with the fewest dependency levels -- the Funk pass costs one
        short launch per level, so this is ~2x faster on the GPU."""
        if initialize_model:
            self.init_feature_const()
        ratings_index, ratings = self.get_ratings(randomize_order=order)
        if handle_bias:
            self.compute_overall_avg()
            self.compute_items_bias_bk()
            self.compute_users_bias_bk()
            _kernels.estimator_loop_with_bias(
                self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality,
                self.feature_init, self.learning_rate, self.learning_rate_users,
                self.learning_rate_items, self.K, self.overall_bias, self.svd_u, self.svd_v,
                ratings_index, ratings, self.items_bias, self.users_bias, self.nbr_users,
                self.nbr_items, int(verbose))
        else:
            _kernels.estimator_loop_without_bias(
                self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality,
                self.feature_init, self.learning_rate, self.K, self.svd_u, self.svd_v,
                ratings_index, ratings, self.nbr_users, self.nbr_items, int(verbose))

    train = feature_training                                                  # :545
    fit = feature_training

    def feature_training_bias(self, initialize_model=True, handle_bias=False, verbose=False):
        """gradient_descent.py:472-503: learned biases, full k-dot per rating."""
        if initialize_model:
            self.init_feature_const()
        ratings_index, ratings = self.get_ratings(randomize_order=True)
        self.compute_overall_avg()
        self.compute_items_bias_bk()
        self.compute_users_bias_bk()
        _kernels.estimator_loop_with_learned_bias(
            self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality,
            self.feature_init, self.learning_rate, self.learning_rate_users,
            self.learning_rate_items, self.K, self.K2, self.overall_bias, self.svd_u, self.svd_v,
            ratings_index, ratings, self.items_bias, self.users_bias, self.nbr_users,
            self.nbr_items, int(verbose))

    def feature_training2(self, initialize_model=True, verbose=False):
        """gradient_descent.py:299-328: the per-epoch variant -- estimator_subloop once per epoch on a dense
        nu x ni cache owned by the caller, predictor_subloop after every feature; ratings in matrix order.
        (`rmse` carries over from feature to feature, `rmse_last` is first assigned inside the loop: with
        min_epochs = 0 the reference raises UnboundLocalError, and so does this.)"""
        rmse = 2.0
        if initialize_model:
            self.init_feature_const()
        ratings_cache = np.zeros(self.nbr_users * self.nbr_items, dtype=np.float64)
        ratings_index, ratings = self.get_ratings()
        for f in range(self.dimensionality):
            epoch = 0
            while epoch < self.min_epochs or rmse <= rmse_last - self.min_improvement:
                rmse_last = rmse
                rmse = _kernels.estimator_subloop(f, epoch, self.min_improvement, self.dimensionality,
                                                  self.feature_init, self.learning_rate, self.K, self.svd_u,
                                                  self.svd_v, ratings_index, ratings, ratings_cache,
                                                  self.nbr_users, self.nbr_items, int(verbose))
                epoch += 1
            _kernels.predictor_subloop(f, epoch, self.dimensionality, self.feature_init, self.svd_u, self.svd_v,
                                       ratings_index, ratings, ratings_cache, self.nbr_users, self.nbr_items)

    def feature_training_folding(self, initialize_model=False, handle_bias=False, verbose=False):
        """gradient_descent.py:450-470: feature_training on the model as it stands (no re-initialisation by
        default, biases as they are), then the per-component means."""
        if initialize_model:
            self.init_feature_const()
        ratings_index, ratings = self.get_ratings(randomize_order=True)
        if handle_bias:
            # the reference passes this argument list (:461-463) -- two learning rates short of the kernel's
            # signature (gd_estimator.pyx:489-507) -- so handle_bias=True raises TypeError there, and here
            _kernels.estimator_loop_with_bias(
                self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality,
                self.feature_init, self.learning_rate, self.K, self.overall_bias, self.svd_u, self.svd_v,
                ratings_index, ratings, self.items_bias, self.users_bias, self.nbr_users,
                self.nbr_items, int(verbose))
        else:
            _kernels.estimator_loop_without_bias(
                self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality,
                self.feature_init, self.learning_rate, self.K, self.svd_u, self.svd_v,
                ratings_index, ratings, self.nbr_users, self.nbr_items, int(verbose))
        self.compute_components_mean()

    def feature_training_dev(self, initialize_model=True, probe=None, verbose=False):
        """gradient_descent.py:577-599: estimator_loop (dense cache, honours max_epochs) -> the rmse history
        f64[max_epochs * k].  The reference draws the shuffle and then overwrites the arrays with the matrix
        order (:589-593); the draw is kept for the RNG stream, the order is the matrix order."""
        rmse = np.zeros(self.max_epochs * self.dimensionality)
        if initialize_model:
            self.init_feature_const()
        self.get_ratings(randomize_order=True)
        ratings_index, ratings = self.get_ratings(randomize_order=False)
        _kernels.estimator_loop(self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality,
                                self.feature_init, self.learning_rate, self.K, self.svd_u, self.svd_v,
                                ratings_index, ratings, 0, rmse, self.nbr_users, self.nbr_items, int(verbose))
        return rmse

    def compute_components_mean(self):                                        # :803-809
        self.components_mean = self.svd_u.mean(axis=1)

    def estimate_rating(self, feature_index, user_index, f, cache=False, trailing=False):
        """gradient_descent.py:602-619: the Funk estimate of one pair at feature f (host scalar helper)."""
        total = cache if (cache and cache > 0) else 1.0
        total += self.svd_u[f, feature_index] * self.svd_v[f, user_index]
        total = self.clamping(total)
        if trailing:
            total += (self.dimensionality - f - 1) * self.feature_init * self.feature_init
            total = self.clamping(total)
        return total

    def feature_training_implicit(self, initialize_model=True, verbose=False):
        """gradient_descent.py:547-574: the SVD++-like trainer (sequential by nature; see lib.gd_estimator)."""
        if initialize_model:
            self.init_feature_const()
        ratings_index, ratings = self.get_ratings(randomize_order=True)
        self.compute_overall_avg()
        self.compute_items_bias_bk()
        self.compute_users_bias_bk()
        self.initialize_rated_feedback()
        self.items_feedback = np.zeros([self.dimensionality, self.nbr_items])
        _kernels.estimator_loop_with_implicit_feedback(
            self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality, self.feature_init,
            self.learning_rate, self.learning_rate_users, self.learning_rate_items, self.K, self.overall_bias,
            self.svd_u, self.svd_v, self.items_feedback, ratings_index, ratings, self.feedback_rated,
            self.feedback_hash, self.items_bias, self.users_bias, self.nbr_users, self.nbr_items, int(verbose))

    def predict_rating(self, item_index, user_index):                         # :621-631
        return np.dot(self.svd_u[:, item_index], self.svd_v[:, user_index]) + 1.0

    predict = predict_rating                                                  # :634

    def predict_rating_with_bias(self, item_index, user_index):               # :637-648
        s = np.dot(self.svd_u[:, item_index], self.svd_v[:, user_index])
        return s + self.overall_bias + (self.items_bias[item_index] + self.users_bias[user_index])

    def find_user_top_match(self, user_index, nbr_recommendations=5):
        """gradient_descent.py:768-800: every item scored with predict_rating, the user's rated items and the item
        whose id equals user_index excluded, zero scores dropped -> (items, scores)."""
        self._mark_csc_due()                         # :777
        return self._top_items(user_index, self.nbr_items, "predict_rating", nbr_recommendations)

    def predict_rating_implicit(self, item_index, user_index):
        """gradient_descent.py:651-667: the predictor of the implicit-feedback model -- the user vector plus the
        normalised sum of the feedback vectors of the items the user rated, dotted with the item vector, + biases."""
        seek, span = self.feedback_hash[user_index]
        feedback_norm = 1.0 / np.sqrt(span)
        items_id = self.feedback_rated[seek:seek + span, 1]
        p_u = self.svd_v[:, user_index] + feedback_norm * self.items_feedback[:, items_id].sum(axis=1)
        s = np.dot(self.svd_u[:, item_index], p_u)
        return s + self.overall_bias + (self.items_bias[item_index] + self.users_bias[user_index])

    def _retrain(self, sel, ratings_index, ratings, update_users, update_items, verbose):
        if self.items_bias is None or self.users_bias is None:
            self.compute_overall_avg(); self.compute_items_bias_bk(); self.compute_users_bias_bk()
        _kernels.estimator_loop_with_bias_dev(
            self.min_epochs, self.max_epochs, self.min_improvement, self.dimensionality,
            self.feature_init, self.learning_rate, self.learning_rate_users,
            self.learning_rate_items, self.K, self.overall_bias, self.svd_u, self.svd_v,
            np.ascontiguousarray(ratings_index[sel, :]), np.ascontiguousarray(ratings[sel]),
            self.items_bias, self.users_bias, self.nbr_users, self.nbr_items, update_users,
            update_items, int(verbose))

    def retrain_user(self, user_index, ratings_index, ratings, verbose=False):
        """Folding-in (gradient_descent.py:879-891): Funk passes with gates (1, 0) on this user's
        ratings only."""
        self.init_user_features(user_index)
        self._retrain(np.where(ratings_index[:, 0] == user_index)[0], ratings_index, ratings, 1, 0, verbose)

    def retrain_item(self, item_index, ratings_index, ratings, verbose=False):
        """gradient_descent.py:894-905: gates (0, 1)."""
        self.init_item_features(item_index)
        self._retrain(np.where(ratings_index[:, 1] == item_index)[0], ratings_index, ratings, 0, 1, verbose)
