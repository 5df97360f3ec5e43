#!/bin/bash
# Times the regular build and every item-sweep variant under build/variants on the explicit-Gram item half-sweep.
set -e
cd "$(dirname "$0")/../.."
V=movie-recommender-system_amd/build/variants
timeout -k 10 120 python tools/probes/als_item_variants.py movie-recommender-system_amd/libmirec.so
for lib in $V/libmirec_item_*.so; do
  timeout -k 10 120 python tools/probes/als_item_variants.py "$lib"
done
