#!/bin/bash
# Round 5: opening argument k = 20 after the host divstep inversion; A/B of the table chain's quad-lane threshold at the switch (2^15 + 4 points)
mkdir -p gpurun_out
{
  for rep in 1 2; do
    echo "== default (chain on quads of lanes up to 65536 points)"
    TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
    echo "== H2_TABLE_WIDE_MAX=16384 (one lane per point at 2^15)"
    H2_TABLE_WIDE_MAX=16384 TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  done
  echo "== host vectors, C++ mirror, one call (h2_open)"
  build/host_mirror_check opening-time 20 5
  python -m pytest tests/test_gpu_opening.py -q -x -k "whole_argument or native_host_mirror or at_size" 2>&1 | tail -3
} > gpurun_out/r05_open_ab2.txt 2>&1
tail -30 gpurun_out/r05_open_ab2.txt
