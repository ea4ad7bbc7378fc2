#!/usr/bin/env python3
"""mfrec/examples/example1b_movielens_100k_wrmf.py on this library -- the same calls in the same order (constructor with
the same parameter dictionary, set_item_by_id per rating, set_item_label, train(verbose=True, handle_bias=True),
precision_recall(...)), on synthetic data of MovieLens-100K's shape (the dataset itself is not redistributable and
there is no network here).  Needs an MI355X.  The similarity-graph part of the original is outside this library's scope."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mfrec_amd import synth
from mfrec_amd.recommendation.metrics import precision_recall
from mfrec_amd.recommendation.wrmf import WRMFRecommender

if __name__ == "__main__":
    nbr_user, nbr_feature = 943, 1682
    idx, r = synth.make_ratings(nbr_user, nbr_feature, 100000, seed=100)
    cut = 80000
    u = np.column_stack([idx[:cut] + 1, r[:cut]])                 # MovieLens ids start at 1
    u_test = np.column_stack([idx[cut:] + 1, r[cut:]])
    parameters = {"nbr_epochs": 30, "feature_init": 0.1, "regularization_model": 0.015, "nbr_features": 20,
                  "neighborhood": 1500}
    MovieLensRecommender = WRMFRecommender(nbr_user + 1, nbr_feature + 1, parameters)
    t0 = time.time()
    for rating in u:
        MovieLensRecommender.set_item_by_id(int(rating[0]), int(rating[1]), 1.0)
    for i in range(1, nbr_feature + 1):
        MovieLensRecommender.set_item_label(i, "movie %d (19%02d)" % (i, i % 100))
    t1 = time.time()
    MovieLensRecommender.train(verbose=True, handle_bias=True)
    t2 = time.time()
    print("loading %.2f s, training %.3f s" % (t1 - t0, t2 - t1))
    print(precision_recall(MovieLensRecommender, u_test, nbr_recommendations=5, verbose=True))
    print(precision_recall(MovieLensRecommender, u_test, nbr_recommendations=5, batched=True))
    items, scores = MovieLensRecommender.find_recommended_items(user_index=1, nbr_recommendations=5, output_label=True)
    print(items, scores)
