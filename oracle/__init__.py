"""CPU oracle for the recommender hot path -- test infrastructure only.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the shipped package never does.

PARITY UNPINNED: the arithmetic lives in implicit==0.7.2 / lightfm==1.17
(reference ``requirements.txt:2,4``), which are neither vendored nor installed
here, and the reference's tests mock them
(``src/shrekommender_system/tests/test_als_recommender.py:30-32``).  What is
restated, and from which call sites, is written at the top of
``mirec_oracle.c``.
"""

from .oracle import (  # noqa: F401
    als_fit,
    als_half_sweep,
    als_half_sweep_exact,
    als_loss,
    build,
    gram,
    lightfm_auc,
    lightfm_init,
    lightfm_warp_epoch,
    lightfm_predict,
    lightfm_ranks,
    lightfm_repr,
    lightfm_topk,
    num_threads,
    scores,
    set_num_threads,
    use_widest_build,
    topk,
)
