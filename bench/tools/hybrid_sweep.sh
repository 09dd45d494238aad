#!/bin/bash
# the opening argument's switch point: total ms at 2^K for hybrid_rounds = J  (K, J lists from the environment)
for K in ${KS:-20}; do for J in ${JS:-3 4 5 6 7}; do
  echo -n "J=$J "; K=$K HYBRID=$J python bench/tools/opening_probe.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"k\"], d[\"total_ms\"], d[\"before_round0_ms\"])"
done; done
