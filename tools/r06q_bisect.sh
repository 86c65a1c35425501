#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
cp mental-poker_amd/libmpshuffle.so /tmp/lib_keep.so
for v in nolist noreduce; do
  cp tools/ab/lib_$v.so mental-poker_amd/libmpshuffle.so
  timeout 200 python tools/r06_group_sweep.py --batches 8192 --verify-only --steps 1 --configs "243712:13:0" > /tmp/o.json 2> /tmp/e.txt
  echo "$v rc=$? $(grep -a -c aborting /tmp/e.txt) $(tail -c 300 /tmp/o.json | head -c 200)"
done
cp /tmp/lib_keep.so mental-poker_amd/libmpshuffle.so
