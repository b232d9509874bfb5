#!/bin/bash
# profiling helper: k_curl time of several builds of libtcgpu.so (toycluster_amd/lib/libtcgpu_<tag>.so)
cp toycluster_amd/lib/libtcgpu.so /tmp/libtcgpu_orig.so
for tag in "$@"; do
  cp toycluster_amd/lib/libtcgpu_$tag.so toycluster_amd/lib/libtcgpu.so
  echo "== $tag"; python3 tools/curl_time.py 2e6 2>&1 | tail -1
done
cp /tmp/libtcgpu_orig.so toycluster_amd/lib/libtcgpu.so
