#!/bin/bash
# Where to hand rows from the matrix-free kernel to the explicit-Gram kernel: times the user half-sweep over a range of
# limits, with the gathered table on chip (27 k items) and beyond it (200 k items).
cd "$(dirname "$0")/../.."
for sm in 32 64 128 256 512 1024; do
  timeout -k 10 120 python tools/microbench.py sweep --side user --short-max $sm 2>&1 | grep sweep
done
for sm in 64 128 256 512 1024; do
  timeout -k 10 160 python tools/microbench.py sweep --side user --users 2000000 --items 200000 --mean-degree 200 --short-max $sm 2>&1 | grep sweep
done
