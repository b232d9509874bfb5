import sys, threading
sys.path.insert(0, '.')
import numpy as np
from toycluster_amd import binding, hostio
R, per = int(sys.argv[1]), int(float(sys.argv[2]))
n = R * per
s = hostio.setup_system("tests/golden/cluster.par", {"ntotal": 2 * n, "mass_ratio": 0.3125})
pos, ids = hostio.sample_gas(s, nthreads=8)
m = hostio.setup_to_model(s)
ctxs = [binding.TcGpu(0) for _ in range(R)]
binding.loopback_group(ctxs)
out = [None] * R
def work(r):
    g = ctxs[r]; g.set_model(m); g.upload(pos, ids)
    infos = []
    for it in range(4):
        g.density_error(); infos.append(g.local_set_info()); g.wvt_step(0.0085, fetch=False)
    p = g.particles()                      # collective (presentation): every rank calls it
    out[r] = (infos, p['hsml'] if r == 0 else None)
th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
[t.start() for t in th]; [t.join() for t in th]
for r in range(R): print(r, [(i['nloc'], i['retries']) for i in out[r][0]], flush=True)
h = out[0][1] / m.boxsize
print('hsml/box pct', np.percentile(h, [0, 50, 99, 99.9, 99.99, 100]))
