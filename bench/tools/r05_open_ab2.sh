#!/bin/bash
# Round 5: the opening argument at k = 20 from the resident Python mirror (twice) and from host Vecs through the C++ mirror (h2_open), with the
# parity tests of the whole-argument entry points.  (The run kept in profiles/r05_open_entry.txt also alternated H2_TABLE_WIDE_MAX=16384 -- the table's
# doubling chain on one lane per point at 2^15 points -- which lost to the quad-lane chain and was removed with its switch.)
mkdir -p gpurun_out
{
  for rep in 1 2; do
    TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  done
  echo "== host vectors, C++ mirror, one call (h2_open)"
  build/host_mirror_check opening-time 20 5
  python -m pytest tests/test_gpu_opening.py -q -x -k "whole_argument or native_host_mirror or at_size" 2>&1 | tail -3
} > gpurun_out/r05_open_ab2.txt 2>&1
tail -30 gpurun_out/r05_open_ab2.txt
