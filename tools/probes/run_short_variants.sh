#!/bin/bash
# Times the regular build and every variant under build/variants on the matrix-free user half-sweep.
set -e
cd "$(dirname "$0")/../.."
V=movie-recommender-system_amd/build/variants
timeout -k 10 120 python tools/probes/als_short_variants.py movie-recommender-system_amd/libmirec.so
for lib in $V/libmirec_short_*.so; do
  timeout -k 10 120 python tools/probes/als_short_variants.py "$lib"
done
