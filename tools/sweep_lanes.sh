#!/bin/bash
# K1 throughput over group size x lanes at the 7.1 / 6.15 s size (run on the GPU box from the repo root)
for B in 8 12 16 24 32; do for lanes in 1 2 3 4; do
  timeout -k 10 120 python tools/k1_rate.py 391270 295270 $B $lanes $((640 / B)) 2>&1 | tail -1
done; done
