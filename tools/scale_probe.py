"""Measurement helper: whole relaxation to the reference's stop rule at config-2 size, curl timing, and a
single-GPU run at config-3 size (1.6e7 particles)."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, model as M

out = {}
n = 2_000_000
m = M.preset("merger", n)
pos, ids = M.sample_gas(m, n, seed=14041981)
g = binding.TcGpu(0)
g.set_model(m)
t0 = time.perf_counter(); g.upload(pos, ids); t_up = time.perf_counter() - t0
t0 = time.perf_counter(); log = g.Regularise_sph_particles(); t_relax = time.perf_counter() - t0
t0 = time.perf_counter(); g.Find_sph_quantities(); t_final = time.perf_counter() - t0
p = g.particles()
rm = g.Global_density_model()
a = (rm / np.float32(m.halos[0].rho0)) ** np.float32(0.5)
apot = np.stack([a, a, a], axis=1).astype(np.float32)
g.phase_times(reset=True)
t0 = time.perf_counter(); b = g.Bfld_from_rotA_SPH(apot); t_curl = time.perf_counter() - t0
ph = g.phase_times()
err = np.abs(p["rho"] - rm) / rm
out["config2_full_relaxation"] = dict(n=n, iterations=len(log), seconds=t_relax, upload_seconds=t_up,
                                      err_mean_at_stop=log[-1]["err_mean"], err_max_at_stop=log[-1]["err_max"],
                                      err_median_final=float(np.median(err)), final_density_pass_seconds=t_final,
                                      particles_per_second_to_convergence=n / t_relax,
                                      curl_kernel_ms=1e3 * ph["curl"][0], curl_call_seconds_incl_pcie=t_curl,
                                      log=[binding.format_log_line(l) for l in log])
g.close()
print(json.dumps(out["config2_full_relaxation"])[:600], flush=True)

n = 16_000_000
m = M.preset("merger", n)
t0 = time.perf_counter(); pos, ids = M.sample_gas(m, n, seed=14041981); t_s = time.perf_counter() - t0
g = binding.TcGpu(0)
g.set_model(m)
g.upload(pos, ids)
steps = []
for k in range(4):
    t0 = time.perf_counter()
    e = g.density_error(); g.wvt_step(0.0085, fetch=False)
    steps.append((time.perf_counter() - t0, e[0]))
    print("1.6e7 step", k, steps[-1], flush=True)
ph = g.phase_times()
out["config3_size_on_one_gpu"] = dict(n=n, sample_seconds=t_s, step_seconds=[s[0] for s in steps],
                                      err_mean=[s[1] for s in steps],
                                      phase_ms_per_step={k: 1e3 * v[0] / 4 for k, v in ph.items() if v[1]})
g.close()
json.dump(out, open("gpurun_out/scale_probe.json", "w"), indent=1)
print("written")
