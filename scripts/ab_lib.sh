#!/bin/bash
# A/B helper: run conv_bench with alternative builds of libbts_hip.so (bts_amd/libbts_hip_<tag>.so)
for tag in "$@"; do
  cp bts_amd/libbts_hip_$tag.so bts_amd/libbts_hip.so
  echo "=== $tag"
  python scripts/conv_bench.py --reps 7 2>&1 | grep -v amdgpu.ids
done
