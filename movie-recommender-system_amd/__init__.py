"""MI355X-native numeric core of the Shrekommender movie recommender:
implicit-feedback ALS half-sweeps, fused score + top-K, LightFM hybrid scoring.

Host code is PyTorch-ROCm plumbing over ``libmirec.so`` (hand-written HIP for
gfx950 behind a C ABI, ``include/mirec.h``).  Importing this package needs no
GPU; running anything numeric does, and fails loudly without one.
"""

from .als import AlternatingLeastSquares  # noqa: F401
from .lightfm import LightFM, LightFMScorer  # noqa: F401
from .recommender import ALSRecommender  # noqa: F401

__all__ = ["AlternatingLeastSquares", "LightFM", "LightFMScorer", "ALSRecommender"]
__version__ = "0.1.0"
