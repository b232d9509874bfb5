"""Experiment: will RCCL build a 2-rank communicator with BOTH ranks on device 0?  (The pool's boxes have one GPU; if it
does, the real ncclSend / ncclRecv / all-gather path can be exercised with two ranks here.)  Two fresh processes, the
worker of tests/test_gpu_multiproc.py with the device forced to 0."""
import os, subprocess, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from tests.test_gpu_multiproc import WORKER
tmp = tempfile.mkdtemp()
src = WORKER.replace("g = binding.TcGpu(rank, rank=rank", "g = binding.TcGpu(0, rank=rank")
script = os.path.join(tmp, "worker.py")
open(script, "w").write(src)
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
n = 40009
one = subprocess.run([sys.executable, script, "0", "1", tmp, str(n)], env=env, capture_output=True, text=True, timeout=300)
print("single rank rc", one.returncode, one.stderr[-300:])
procs = [subprocess.Popen([sys.executable, script, str(r), "2", tmp, str(n)], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True) for r in range(2)]
ok = True
for r, p in enumerate(procs):
    try:
        so, se = p.communicate(timeout=100)
    except subprocess.TimeoutExpired:
        p.kill(); so, se = p.communicate(); ok = False
        print("rank", r, "timed out")
    print("rank", r, "rc", p.returncode, (so + se)[-600:])
    ok = ok and p.returncode == 0
if ok:
    ref = np.load(os.path.join(tmp, "out_0_of_1.npz"))
    for r in range(2):
        z = np.load(os.path.join(tmp, "out_%d_of_2.npz" % r))
        print("rank", r, "log equal", np.array_equal(z["log"], ref["log"]),
              {k: bool(np.array_equal(z[k], ref[k])) for k in ("id", "pos", "hsml", "rho", "varhsmlfac", "rho_model")},
              "nloc", int(z["nloc"]), "recv", float(z["recv"]))
