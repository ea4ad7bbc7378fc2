"""Recommender classes that own the training hot path (reference: mfrec/recommendation)."""
from .base import BaseRecommender, Error  # noqa: F401
from .gradient_descent import GDRecommender  # noqa: F401
from .kmf import KMFRecommender  # noqa: F401
from .wrmf import WRMFRecommender  # noqa: F401
