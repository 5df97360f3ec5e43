"""Stand-in for the ``implicit`` package, for a serving or training environment that keeps the reference's own
plugin (``shrekommender_system.recommenders.als_recommender``, which does ``import implicit`` and builds
``implicit.als.AlternatingLeastSquares(factors=, iterations=, regularization=, alpha=, random_state=, use_gpu=)``
at ``recommenders/als_recommender.py:128-135``; ``als/als.py:167-173`` likewise) but wants the fit and the scoring
to run on the MI355X core.  Put THIS directory's parent (``movie_recommender_system_amd/compat``) on ``PYTHONPATH``
ahead of site-packages and the reference code trains and serves through ``movie_recommender_system_amd`` unchanged.
It is not imported by anything in this package and never shadows a real ``implicit`` unless put on the path.
"""

from . import als  # noqa: F401

__version__ = "0.7.2+mirec"
