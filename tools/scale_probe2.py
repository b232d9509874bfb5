"""Measurement helper (round 2): single-GPU phase times of a warm WVT iteration at 2e6 / 1.6e7 / 1e8 particles,
whole relaxation at config-2 size, k_curl at each size, peak device memory.  Native C sampler for the inputs."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from toycluster_amd import binding, hostio

out = {}
for n in (2_000_000, 16_000_000, 100_000_000):
    s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
    pos, ids = hostio.sample_gas(s, nthreads=16)
    m = hostio.setup_to_model(s)
    g = binding.TcGpu(0)
    g.set_model(m)
    g.upload(pos, ids)
    del pos
    r = {}
    if n == 2_000_000:
        t0 = time.perf_counter(); log = g.Regularise_sph_particles(); r["relax_seconds"] = time.perf_counter() - t0
        r["relax_iterations"] = len(log); r["err_mean_at_stop"] = log[-1]["err_mean"]
        g.upload(*hostio.sample_gas(s, nthreads=16))
    for _ in range(3):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    g.phase_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(3):
        g.density_error(); g.wvt_step(0.0085, fetch=False)
    r["wall_ms_per_iteration"] = 1e3 * (time.perf_counter() - t0) / 3
    ph = g.phase_times()
    r["phase_ms"] = {k: round(1e3 * v[0] / 3, 3) for k, v in ph.items() if v[1]}
    g.Find_sph_quantities()
    rm = g.Global_density_model().astype(np.float64)
    a = ((rm / max(h.rho0 for h in m.halos)) ** 0.5).astype(np.float32)
    apot = np.repeat(a[:, None], 3, axis=1)
    del rm, a
    g.Bfld_from_rotA_SPH(apot)
    g.phase_times(reset=True)
    g.Bfld_from_rotA_SPH(apot)
    r["k_curl_ms"] = 1e3 * g.phase_times()["curl"][0]
    free, total = torch.cuda.mem_get_info(0)
    r["device_memory_in_use_GB"] = (total - free) / 1e9
    out[str(n)] = r
    print(n, json.dumps(r), flush=True)
    g.close()
    del apot
json.dump(out, open("gpurun_out/scale_probe2.json", "w"), indent=1)
