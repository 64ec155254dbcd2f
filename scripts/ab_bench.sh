#!/bin/bash
# A/B helper: full-model bench with alternative builds of libbts_hip.so (bts_amd/libbts_hip_<tag>.so)
for tag in "$@"; do
  cp bts_amd/libbts_hip_$tag.so bts_amd/libbts_hip.so
  echo "=== $tag"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep timed
done
