#!/bin/bash
# profiling helper: bench several builds of libtcgpu.so (toycluster_amd/lib/libtcgpu_<tag>.so)
for tag in "$@"; do
  cp toycluster_amd/lib/libtcgpu.so /tmp/libtcgpu_orig.so
  cp toycluster_amd/lib/libtcgpu_$tag.so toycluster_amd/lib/libtcgpu.so
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-relax 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['err_mean_last'])"
  cp /tmp/libtcgpu_orig.so toycluster_amd/lib/libtcgpu.so
done
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-relax 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['err_mean_last'])"
