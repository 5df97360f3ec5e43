"""``implicit.als`` as the reference uses it: the one class it constructs."""

from movie_recommender_system_amd.als import AlternatingLeastSquares  # noqa: F401

__all__ = ["AlternatingLeastSquares"]
