#!/bin/bash
# round 5: the latency split (plan 1: 768 < B <= 2 560) and the small split (plan 5) re-swept around their round-4 values
for p in "1 2 16 4 8 8" "1 2 16 4 8 16" "1 2 8 4 8 16" "1 1 16 4 8 16" "1 2 32 4 8 16" "1 2 16 2 4 16" "1 2 16 8 8 16" "1 4 16 4 8 16" "1 2 16 4 8 12"; do
  python tools/r05_small.py --quiet --plan $p 1024 2048 2>&1 | grep -v amdgpu | grep "plan\|^B="
done
for p in "5 1 8 2 4 8" "5 1 8 2 4 16" "5 1 4 2 4 16" "5 2 8 2 4 16"; do
  python tools/r05_small.py --quiet --plan $p 256 512 2>&1 | grep -v amdgpu | grep "plan\|^B="
done
