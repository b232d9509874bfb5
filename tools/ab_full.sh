for tag in base default; do
  if [ $tag = base ]; then cp toycluster_amd/lib/libtcgpu.so /tmp/libtcgpu_orig.so; cp toycluster_amd/lib/libtcgpu_base.so toycluster_amd/lib/libtcgpu.so; fi
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['config']['ranks'][0]['phase_ms_per_step']; r=d['roofline']
print('$tag', round(d['ms_per_step'],3), {k: round(v,3) for k,v in p.items() if v>0.1}, 'cand', r['candidates_per_particle'], 'q', r['queries_per_particle'], 'pairs', r['solver_pair_evals_per_particle'], 'relax', d['relaxation']['relax_wall_s'])"
  if [ $tag = base ]; then cp /tmp/libtcgpu_orig.so toycluster_amd/lib/libtcgpu.so; fi
done
