#!/bin/bash
# Round 5: the whole-argument entry points (h2_open / h2_open_device) -- parity tests, then the k = 20 opening argument in both forms from the
# resident Python mirror and from host vectors through the C++ mirror.  Output: gpurun_out/r05_open_entry.txt
mkdir -p gpurun_out
{
  python -m pytest tests/test_gpu_opening.py -q -x 2>&1 | tail -5
  echo "== resident (bench/tools/opening_probe.py), one call"
  TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  echo "== resident, step by step (NATIVE=0)"
  TABLES=0 NATIVE=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  echo "== resident, one call, again"
  TABLES=0 python bench/tools/opening_probe.py 2>&1 | tail -1
  echo "== host vectors, C++ mirror, one call (h2_open)"
  build/host_mirror_check opening-time 20 5
  echo "== host vectors, C++ mirror, step by step"
  build/host_mirror_check opening-time 20 4 stepwise
} > gpurun_out/r05_open_entry.txt 2>&1
tail -40 gpurun_out/r05_open_entry.txt
