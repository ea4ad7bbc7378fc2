"""Drop-in equivalents of mfrec.lib.{kmf_train, gd_estimator, als_implicit, datasets}."""
