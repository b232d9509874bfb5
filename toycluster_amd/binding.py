"""ctypes binding of libtcgpu.so (include/tcgpu.h).  No CPU fallback: a missing library or a
failing device call raises.

Python mirror of the reference's operator interface for this path -- the method names are the
reference's function names (proto.h:17,23,25,46; peano.h:5-6; tree.h:2,5):
    Find_sph_quantities, Regularise_sph_particles, Bfld_from_rotA_SPH, Global_density_model,
    Sort_Particles_By_Peano_Key, Peano_Key, Find_ngb_tree, Guess_hsml
acting on a `TcGpu` context instead of the reference's process globals.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtcgpu.so")

NGBMAX = 2360
DESNNGB = 295
MAXLOG = 66

_ERR = {-1: "HIP", -2: "ARG", -3: "NOMEM", -4: "NONFINITE", -5: "COORD_RANGE", -6: "NO_CONVERGENCE",
        -7: "COMM", -8: "OVERFLOW"}


class TcParams(C.Structure):
    _fields_ = [("boxsize", C.c_double), ("mpart_gas", C.c_double), ("mtotal", C.c_double),
                ("bfld_eta", C.c_double), ("nhalos", C.c_int32), ("reserved", C.c_int32)]


class TcHalo(C.Structure):
    _fields_ = [("mass_gas", C.c_double), ("d_com", C.c_double * 3), ("rho0", C.c_double), ("beta", C.c_double),
                ("rcore", C.c_double), ("rcut", C.c_double), ("have_cuspy", C.c_int32), ("reserved", C.c_int32),
                ("rho0_cc", C.c_double), ("rc_cc", C.c_double)]


def cool_core_component(model, h):
    """(rho0_cc, rc_cc) of the reference's -DDOUBLE_BETA_COOL_CORES build (src/setup.c:604-612): set for halos with
    Have_Cuspy when the model carries Param.Rho0_Fac / Param.Rc_Fac; (0, 0) -- the default build -- otherwise."""
    f_rho, f_rc = getattr(model, "rho0_fac", 0.0), getattr(model, "rc_fac", 0.0)
    if f_rho and f_rc and getattr(h, "have_cuspy", 0):
        return h.rho0 * f_rho, h.rcore / f_rc
    return 0.0, 0.0


class TcIterLog(C.Structure):
    _fields_ = [("it", C.c_int32), ("reserved", C.c_int32), ("err_max", C.c_double), ("err_mean", C.c_double),
                ("err_diff", C.c_double), ("step", C.c_double)]


class TcDensityStats(C.Structure):
    _fields_ = [("queries_per_particle", C.c_double), ("solver_iters_per_particle", C.c_double),
                ("pair_evals_per_particle", C.c_double), ("candidates_per_particle", C.c_double)]


# every symbol include/tcgpu.h declares
EXPORTS = [
    "tcgpu_create", "tcgpu_destroy", "tcgpu_last_error", "tcgpu_version", "tcgpu_set_model",
    "tcgpu_upload_particles", "tcgpu_download_particles", "tcgpu_num_particles",
    "tcgpu_sort_particles_by_peano_key", "tcgpu_download_keys", "tcgpu_peano_keys",
    "tcgpu_find_sph_quantities", "tcgpu_last_density_stats", "tcgpu_global_density_model", "tcgpu_find_ngb",
    "tcgpu_build_neighbour_index", "tcgpu_guess_hsml", "tcgpu_wvt_step", "tcgpu_density_error", "tcgpu_regularise_sph_particles",
    "tcgpu_bfld_from_rotA_sph", "tcgpu_comm_unique_id", "tcgpu_comm_init", "tcgpu_comm_init_loopback",
    "tcgpu_set_option",
    "tcgpu_phase_times", "tcgpu_stream", "tcgpu_comm_bytes", "tcgpu_local_set_info",
]

_libs = {}


def lib(variant=""):
    """Load libtcgpu.so (variant "m4": libtcgpu_m4.so, the build restating the reference's -DSPH_CUBIC_SPLINE build);
    raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    _lib = _libs.get(variant)
    if _lib is None:
        path = LIB_PATH if not variant else LIB_PATH.replace(".so", "_%s.so" % variant)
        if not os.path.exists(path):
            raise RuntimeError("%s not built: %s missing (run __graft_entry__.build())" % (os.path.basename(path), path))
        L = C.CDLL(path)
        vp, i32, i64, dbl, flt = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float
        L.tcgpu_create.argtypes = [C.POINTER(vp), i32]
        L.tcgpu_destroy.argtypes = [vp]
        L.tcgpu_destroy.restype = None
        L.tcgpu_last_error.argtypes = [vp]
        L.tcgpu_last_error.restype = C.c_char_p
        L.tcgpu_version.restype = C.c_char_p
        L.tcgpu_set_model.argtypes = [vp, C.POINTER(TcParams), C.POINTER(TcHalo)]
        L.tcgpu_upload_particles.argtypes = [vp, i64, vp, vp, vp]
        L.tcgpu_download_particles.argtypes = [vp] + [vp] * 6
        L.tcgpu_num_particles.argtypes = [vp]
        L.tcgpu_num_particles.restype = i64
        L.tcgpu_sort_particles_by_peano_key.argtypes = [vp]
        L.tcgpu_download_keys.argtypes = [vp, vp, vp]
        L.tcgpu_peano_keys.argtypes = [vp, i64, vp, vp, vp]
        L.tcgpu_find_sph_quantities.argtypes = [vp]
        L.tcgpu_last_density_stats.argtypes = [vp, C.POINTER(TcDensityStats)]
        L.tcgpu_global_density_model.argtypes = [vp, vp]
        L.tcgpu_find_ngb.argtypes = [vp, i64, flt, vp, C.POINTER(C.c_int32)]
        L.tcgpu_build_neighbour_index.argtypes = [vp]
        L.tcgpu_guess_hsml.argtypes = [vp, vp]
        L.tcgpu_wvt_step.argtypes = [vp, dbl, vp, vp, i32]
        L.tcgpu_density_error.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl)]
        L.tcgpu_regularise_sph_particles.argtypes = [vp, i32, C.POINTER(TcIterLog), C.POINTER(C.c_int32)]
        L.tcgpu_bfld_from_rotA_sph.argtypes = [vp, vp, vp]
        L.tcgpu_comm_unique_id.argtypes = [vp]
        L.tcgpu_comm_init.argtypes = [vp, i32, i32, vp]
        L.tcgpu_comm_init_loopback.argtypes = [C.POINTER(vp), i32]
        L.tcgpu_set_option.argtypes = [vp, C.c_char_p, dbl]
        L.tcgpu_phase_times.argtypes = [vp, vp, vp, vp, C.POINTER(i32), i32]
        L.tcgpu_stream.argtypes = [vp]
        L.tcgpu_stream.restype = vp
        L.tcgpu_comm_bytes.argtypes = [vp, i32]
        if hasattr(L, "tcgpu_debug_comm_selftest"):
            L.tcgpu_debug_comm_selftest.argtypes = [vp]
            L.tcgpu_debug_comm_selftest.restype = i32
        L.tcgpu_comm_bytes.restype = dbl
        L.tcgpu_local_set_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_int32)]
        _libs[variant] = _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class TcGpuError(RuntimeError):
    pass


def comm_unique_id():
    buf = np.zeros(128, np.uint8)
    rc = lib().tcgpu_comm_unique_id(_p(buf))
    if rc:
        raise TcGpuError("tcgpu_comm_unique_id failed (%s)" % _ERR.get(rc, rc))
    return buf


def loopback_group(contexts):
    """Testing: make `contexts` (one per host thread, possibly all on one GPU) ranks of an in-process
    loopback communicator (tcgpu_comm_init_loopback)."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = lib().tcgpu_comm_init_loopback(arr, len(contexts))
    if rc:
        raise TcGpuError("tcgpu_comm_init_loopback failed (%s)" % _ERR.get(rc, rc))
    for r, c in enumerate(contexts):
        c.rank, c.nranks = r, len(contexts)


def device_memory_used():
    """Bytes in use on the current device (hipMemGetInfo: total - free) -- for tests and probes of what the contexts hold."""
    hip = C.CDLL("libamdhip64.so")
    fr, tot = C.c_size_t(), C.c_size_t()
    if hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) != 0:
        raise TcGpuError("hipMemGetInfo failed")
    return tot.value - fr.value


class TcGpu:
    """One context per GPU (the reference's globals, made explicit)."""

    def __init__(self, device=0, rank=0, nranks=1, unique_id=None, options=None, variant=""):
        self._L = lib(variant)
        h = C.c_void_p()
        rc = self._L.tcgpu_create(C.byref(h), int(device))
        if rc:
            raise TcGpuError("tcgpu_create(device=%d) failed: %s -- is a gfx950 GPU visible?" % (device, _ERR.get(rc, rc)))
        self._h = h
        self.n = 0
        for k, v in (options or {}).items():
            self.set_option(k, v)
        if nranks > 1 or (options or {}).get("force_comm"):
            uid = np.ascontiguousarray(unique_id if unique_id is not None else comm_unique_id(), dtype=np.uint8)
            self._ck(self._L.tcgpu_comm_init(self._h, int(rank), int(nranks), _p(uid)))
        self.rank, self.nranks = rank, nranks

    def close(self):
        if getattr(self, "_h", None):
            self._L.tcgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            msg = self._L.tcgpu_last_error(self._h)
            raise TcGpuError("%s: %s" % (_ERR.get(rc, rc), msg.decode() if msg else ""))

    def set_option(self, name, value):
        self._ck(self._L.tcgpu_set_option(self._h, name.encode(), float(value)))
        if name == "timing":
            self._timing_on = bool(value)

    # ---- model / particles -------------------------------------------------------------
    def set_model(self, model):
        par = TcParams(model.boxsize, model.mpart_gas, model.mtotal, getattr(model, "bfld_eta", 0.5),
                       len(model.halos), 0)
        halos = (TcHalo * max(1, len(model.halos)))()
        for k, h in enumerate(model.halos):
            halos[k].mass_gas = h.mass_gas
            for c in range(3):
                halos[k].d_com[c] = h.d_com[c]
            halos[k].rho0, halos[k].beta, halos[k].rcore, halos[k].rcut = h.rho0, h.beta, h.rcore, h.rcut
            halos[k].have_cuspy = int(h.have_cuspy)
            halos[k].rho0_cc, halos[k].rc_cc = cool_core_component(model, h)
        self._ck(self._L.tcgpu_set_model(self._h, C.byref(par), halos))
        self.model = model

    def upload(self, pos, ids=None, hsml=None):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        hsml = None if hsml is None else np.ascontiguousarray(hsml, dtype=np.float32)
        self.n = pos.shape[0]
        self._ck(self._L.tcgpu_upload_particles(self._h, self.n, _p(pos), _p(ids), _p(hsml)))

    def particles(self):
        n = self.n
        out = dict(pos=np.empty((n, 3), np.float32), id=np.empty(n, np.int32), hsml=np.empty(n, np.float32),
                   rho=np.empty(n, np.float32), varhsmlfac=np.empty(n, np.float32),
                   rho_model=np.empty(n, np.float32))
        self._ck(self._L.tcgpu_download_particles(self._h, _p(out["pos"]), _p(out["id"]), _p(out["hsml"]),
                                                  _p(out["rho"]), _p(out["varhsmlfac"]), _p(out["rho_model"])))
        return out

    # ---- the reference's operator names ---------------------------------------------------
    def Sort_Particles_By_Peano_Key(self):
        self._ck(self._L.tcgpu_sort_particles_by_peano_key(self._h))
        hi, lo = np.empty(self.n, np.uint64), np.empty(self.n, np.uint64)
        self._ck(self._L.tcgpu_download_keys(self._h, _p(hi), _p(lo)))
        return hi, lo

    def Peano_Key(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        hi, lo = np.empty(len(xyz), np.uint64), np.empty(len(xyz), np.uint64)
        self._ck(self._L.tcgpu_peano_keys(self._h, len(xyz), _p(xyz), _p(hi), _p(lo)))
        return [(int(h) << 64) | int(l) for h, l in zip(hi, lo)]

    def Find_sph_quantities(self):
        self._ck(self._L.tcgpu_find_sph_quantities(self._h))

    def density_stats(self):
        s = TcDensityStats()
        self._ck(self._L.tcgpu_last_density_stats(self._h, C.byref(s)))
        return dict(queries=s.queries_per_particle, solver_iters=s.solver_iters_per_particle,
                    pair_evals=s.pair_evals_per_particle, candidates=s.candidates_per_particle)

    def Global_density_model(self):
        out = np.empty(self.n, np.float32)
        self._ck(self._L.tcgpu_global_density_model(self._h, _p(out)))
        return out

    def build_neighbour_index(self):
        self._ck(self._L.tcgpu_build_neighbour_index(self._h))

    def Find_ngb_tree(self, ipart, hsml):
        buf = np.empty(NGBMAX, np.int32)
        cnt = C.c_int32()
        self._ck(self._L.tcgpu_find_ngb(self._h, int(ipart), float(np.float32(hsml)), _p(buf), C.byref(cnt)))
        return buf[:cnt.value].copy()

    def Guess_hsml(self):
        out = np.empty(self.n, np.float32)
        self._ck(self._L.tcgpu_guess_hsml(self._h, _p(out)))
        return out

    def wvt_step(self, step, move=True, fetch=True):
        hs = np.empty(self.n, np.float32) if fetch else None
        de = np.empty((self.n, 3), np.float32) if fetch else None
        self._ck(self._L.tcgpu_wvt_step(self._h, float(step), _p(hs), _p(de), int(bool(move))))
        return hs, de

    def density_error(self):
        a, b = C.c_double(), C.c_double()
        self._ck(self._L.tcgpu_density_error(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def Regularise_sph_particles(self, max_iter=-1):
        log = (TcIterLog * MAXLOG)()
        nlog = C.c_int32()
        self._ck(self._L.tcgpu_regularise_sph_particles(self._h, int(max_iter), log, C.byref(nlog)))
        return [dict(it=l.it, err_max=l.err_max, err_mean=l.err_mean, err_diff=l.err_diff, step=l.step)
                for l in log[:min(nlog.value, MAXLOG)]]

    def Bfld_from_rotA_SPH(self, apot):
        apot = np.ascontiguousarray(apot, dtype=np.float32).reshape(-1, 3)
        b = np.empty((self.n, 3), np.float32)
        self._ck(self._L.tcgpu_bfld_from_rotA_sph(self._h, _p(apot), _p(b)))
        return b

    def phase_times(self, reset=False):
        """Device seconds and launch counts per phase.  The library records HIP events per phase only with
        option "timing" = 1 (off by default in the product path); the first call switches it on, so measure
        between a phase_times(reset=True) and a phase_times()."""
        if not getattr(self, "_timing_on", False):
            self.set_option("timing", 1)
            self._timing_on = True
        cap = 32
        names = (C.c_char_p * cap)()
        secs = (C.c_double * cap)()
        launches = (C.c_int64 * cap)()
        n = C.c_int(cap)
        self._ck(self._L.tcgpu_phase_times(self._h, names, secs, launches, C.byref(n), int(reset)))
        return {names[i].decode(): (secs[i], launches[i]) for i in range(n.value)}

    def local_set_info(self):
        a, b, r = C.c_int64(), C.c_int64(), C.c_int32()
        self._ck(self._L.tcgpu_local_set_info(self._h, C.byref(a), C.byref(b), C.byref(r)))
        return dict(nloc=a.value, nown=b.value, retries=r.value)

    def comm_bytes(self, reset=False):
        return self._L.tcgpu_comm_bytes(self._h, int(reset))

    def stream(self):
        return self._L.tcgpu_stream(self._h)


def format_log_line(l):
    """The reference's per-iteration line, wvt_relax.c:91-92."""
    return "   #%02d: Err max=%3g mean=%03g diff=%03g step=%g" % (
        l["it"], l["err_max"], l["err_mean"], l["err_diff"], l["step"])
