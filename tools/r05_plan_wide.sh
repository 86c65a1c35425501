#!/bin/bash
# round 5: the wide split (plan 4: 6 144 < B <= 49 152) re-swept around its round-4 values (fixed / variable terms per lane, bases per table
# lane, points per inversion, window lanes)
for p in "4 4 64 16 32 16" "4 2 64 16 32 16" "4 8 64 16 32 16" "4 4 64 8 32 16" "4 4 64 32 32 16" "4 4 64 16 16 16" "4 4 64 16 64 16" "4 4 64 16 32 8" "4 4 32 16 32 16" "4 4 64 16 32 12" "4 2 64 8 16 16" "4 8 64 32 64 8"; do
  python tools/r05_small.py --quiet --plan $p 16384 32768 2>&1 | grep -v amdgpu | grep "plan\|^B="
done
