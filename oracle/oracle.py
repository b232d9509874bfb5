"""ctypes wrapper around oracle/libtcoracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libtcoracle.so")

NGBMAX = 2360
DESNNGB = 295
MAXLOG = 66


class OrcHalo(C.Structure):
    _fields_ = [("mass_gas", C.c_double), ("d_com", C.c_double * 3), ("rho0", C.c_double),
                ("beta", C.c_double), ("rcore", C.c_double), ("rcut", C.c_double),
                ("have_cuspy", C.c_int), ("pad_", C.c_int)]


class OrcIterLog(C.Structure):
    _fields_ = [("it", C.c_int), ("err_max", C.c_double), ("err_mean", C.c_double),
                ("err_diff", C.c_double), ("step", C.c_double)]


def build(force=False, variant=""):
    """variant "m4": the reference's -DSPH_CUBIC_SPLINE build (libtcoracle_m4.so)."""
    src = os.path.join(_HERE, "tc_oracle.c")
    path = _LIB if not variant else _LIB.replace(".so", "_%s.so" % variant)
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(path)], stdout=subprocess.DEVNULL)
    return path


_libs = {}


def lib(variant=""):
    _lib = _libs.get(variant)
    if _lib is None:
        L = C.CDLL(build(variant=variant))
        vp, ci, cd, cf = C.c_void_p, C.c_int, C.c_double, C.c_float
        L.orc_create.restype = vp
        L.orc_create.argtypes = [ci, cd, cd, cd, ci, C.POINTER(OrcHalo), ci]
        L.orc_destroy.argtypes = [vp]
        L.orc_set_particles.argtypes = [vp, vp, vp, vp]
        L.orc_get_particles.argtypes = [vp] + [vp] * 6
        L.orc_set_apot.argtypes = [vp, vp]
        L.orc_get_apot.argtypes = [vp, vp]
        L.orc_get_bfld.argtypes = [vp, vp]
        L.orc_peano_key.argtypes = [cd, cd, cd, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_reversed_peano_key.argtypes = [cd, cd, cd, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_sort_by_peano_key.argtypes = [vp, vp, vp, vp]
        L.orc_build_tree.argtypes = [vp]
        L.orc_build_tree.restype = ci
        L.orc_tree_nodes.argtypes = [vp] + [vp] * 6
        L.orc_tree_nodes.restype = ci
        L.orc_find_ngb_tree.argtypes = [vp, ci, cf, vp]
        L.orc_find_ngb_tree.restype = ci
        L.orc_find_ngb_simple.argtypes = [vp, ci, cf, vp]
        L.orc_find_ngb_simple.restype = ci
        L.orc_guess_hsml.argtypes = [vp, ci]
        L.orc_guess_hsml.restype = cf
        L.orc_find_sph_quantities.argtypes = [vp]
        L.orc_find_sph_quantities.restype = ci
        L.orc_global_density_model.argtypes = [vp, vp]
        L.orc_regularise.argtypes = [vp, C.POINTER(OrcIterLog), ci]
        L.orc_regularise.restype = ci
        L.orc_wvt_step.argtypes = [vp, cd, vp, vp, ci]
        L.orc_bfld_from_rotA.argtypes = [vp]
        L.orc_last_stats.argtypes = [vp, C.POINTER(cd), C.POINTER(cd), C.POINTER(cd)]
        L.orc_reassign_to_halos.argtypes = [ci, vp, cd, ci, C.POINTER(OrcHalo), vp, vp, vp, vp]
        L.orc_reassign_to_halos.restype = ci
        _libs[variant] = _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def peano_key(x, y, z):
    hi, lo = C.c_uint64(), C.c_uint64()
    lib().orc_peano_key(x, y, z, C.byref(hi), C.byref(lo))
    return (hi.value << 64) | lo.value


def reversed_peano_key(x, y, z):
    hi, lo = C.c_uint64(), C.c_uint64()
    lib().orc_reversed_peano_key(x, y, z, C.byref(hi), C.byref(lo))
    return (hi.value << 64) | lo.value


DEV_SWEEP_ROUND_ONCE, DEV_TREE_SUMS, DEV_POW_ULP, DEV_KERNEL_ULP, DEV_EXACT_BALL = 1, 2, 4, 8, 16


def set_deviation(mask=0):
    """Attribution experiments (tc_oracle.h ORC_DEV_*): 0 = the faithful restatement.  Process-wide; reset after use.
    DEV_EXACT_BALL answers every ball query as the reference's brute-force Find_ngb_simple would (its tree search misses
    the particles of the occasional mis-placed node)."""
    L = lib()
    L.orc_set_deviation.argtypes = [C.c_int]
    L.orc_set_deviation.restype = None
    L.orc_set_deviation(int(mask))


def set_double_beta(rho0_fac=0.0, rc_fac=0.0):
    """The reference's -DDOUBLE_BETA_COOL_CORES build as a process-wide switch of the oracle (Param.Rho0_Fac,
    Param.Rc_Fac; both 0 = the default build).  Reset it after use."""
    L = lib()
    L.orc_set_double_beta.argtypes = [C.c_double, C.c_double]
    L.orc_set_double_beta.restype = None
    L.orc_set_double_beta(float(rho0_fac), float(rc_fac))


class Oracle:
    """State handle mirroring the reference's globals P / SphP / Param / Halo for the gas particles."""

    def __init__(self, model, pos, ids=None, hsml=None, nthreads=0, variant=""):
        """model: any object with boxsize, mpart_gas, mtotal and halos (list of dicts/objs with
        mass_gas, d_com, rho0, beta, rcore, rcut, have_cuspy)."""
        self._L = lib(variant)                    # "m4": the -DSPH_CUBIC_SPLINE build of the oracle
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        self.n = pos.shape[0]
        halos = (OrcHalo * len(model.halos))()
        for k, h in enumerate(model.halos):
            g = (lambda name: h[name]) if isinstance(h, dict) else (lambda name: getattr(h, name))
            halos[k].mass_gas = g("mass_gas")
            for c in range(3):
                halos[k].d_com[c] = g("d_com")[c]
            halos[k].rho0, halos[k].beta = g("rho0"), g("beta")
            halos[k].rcore, halos[k].rcut = g("rcore"), g("rcut")
            halos[k].have_cuspy = int(g("have_cuspy"))
        self._h = self._L.orc_create(self.n, model.boxsize, model.mpart_gas, model.mtotal,
                                   len(model.halos), halos, nthreads)
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        hsml = None if hsml is None else np.ascontiguousarray(hsml, dtype=np.float32)
        self._L.orc_set_particles(self._h, _p(pos), _p(ids), _p(hsml))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    def particles(self):
        n = self.n
        out = dict(pos=np.empty((n, 3), np.float32), id=np.empty(n, np.int32), hsml=np.empty(n, np.float32),
                   rho=np.empty(n, np.float32), varhsmlfac=np.empty(n, np.float32),
                   rho_model=np.empty(n, np.float32))
        self._L.orc_get_particles(self._h, _p(out["pos"]), _p(out["id"]), _p(out["hsml"]), _p(out["rho"]),
                                _p(out["varhsmlfac"]), _p(out["rho_model"]))
        return out

    def sort_by_peano_key(self):
        n = self.n
        hi, lo, perm = np.empty(n, np.uint64), np.empty(n, np.uint64), np.empty(n, np.int64)
        self._L.orc_sort_by_peano_key(self._h, _p(hi), _p(lo), _p(perm))
        return hi, lo, perm

    def build_tree(self):
        return self._L.orc_build_tree(self._h)

    def tree_nodes(self):
        nn = self._L.orc_tree_nodes(self._h, None, None, None, None, None, None)
        out = dict(bitfield=np.empty(nn, np.uint32), dnext=np.empty(nn, np.int32), pos=np.empty((nn, 3), np.float32),
                   npart=np.empty(nn, np.int32), size=np.empty(nn, np.float32),
                   tree_parent=np.empty(self.n, np.int32))
        self._L.orc_tree_nodes(self._h, _p(out["bitfield"]), _p(out["dnext"]), _p(out["pos"]), _p(out["npart"]),
                             _p(out["size"]), _p(out["tree_parent"]))
        return out

    def find_ngb_tree(self, ipart, hsml):
        buf = np.empty(NGBMAX, np.int32)
        c = self._L.orc_find_ngb_tree(self._h, int(ipart), float(np.float32(hsml)), _p(buf))
        return buf[:c].copy()

    def find_ngb_simple(self, ipart, hsml):
        buf = np.empty(NGBMAX, np.int32)
        c = self._L.orc_find_ngb_simple(self._h, int(ipart), float(np.float32(hsml)), _p(buf))
        return buf[:c].copy()

    def guess_hsml(self, ipart):
        return np.float32(self._L.orc_guess_hsml(self._h, int(ipart)))

    def find_sph_quantities(self):
        rc = self._L.orc_find_sph_quantities(self._h)
        if rc < 0:
            raise RuntimeError("oracle find_sph_quantities failed rc=%d" % rc)

    def last_stats(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._L.orc_last_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(queries=a.value, solver_iters=b.value, pair_evals=c.value)

    def global_density_model(self):
        out = np.empty(self.n, np.float32)
        self._L.orc_global_density_model(self._h, _p(out))
        return out

    def regularise(self, max_iter=-1):
        log = (OrcIterLog * MAXLOG)()
        nlog = self._L.orc_regularise(self._h, log, max_iter)
        if nlog < 0:
            raise RuntimeError("oracle regularise failed")
        return [dict(it=l.it, err_max=l.err_max, err_mean=l.err_mean, err_diff=l.err_diff, step=l.step)
                for l in log[:min(nlog, MAXLOG)]]

    def wvt_step(self, step, move=True):
        hs = np.empty(self.n, np.float32)
        de = np.empty((self.n, 3), np.float32)
        self._L.orc_wvt_step(self._h, float(step), _p(hs), _p(de), int(bool(move)))
        return hs, de

    def set_apot(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)
        self._L.orc_set_apot(self._h, _p(a))

    def bfld_from_rotA(self):
        self._L.orc_bfld_from_rotA(self._h)
        b = np.empty((self.n, 3), np.float32)
        self._L.orc_get_bfld(self._h, _p(b))
        return b


def _orc_halos(model):
    halos = (OrcHalo * len(model.halos))()
    for k, h in enumerate(model.halos):
        halos[k].mass_gas = h.mass_gas
        for c in range(3):
            halos[k].d_com[c] = h.d_com[c]
        halos[k].rho0, halos[k].beta, halos[k].rcore, halos[k].rcut = h.rho0, h.beta, h.rcore, h.rcut
        halos[k].have_cuspy = int(h.have_cuspy)
    return halos


def reassign_to_halos(model, pos):
    """positions.c:264-445 for the gas block: (halo_id[n], perm[n], npart[nhalos]); new[i] = old[perm[i]]."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    n, nh = pos.shape[0], len(model.halos)
    rs = np.array([h.r_sample for h in model.halos], np.float64)
    hid, perm, npart = np.empty(n, np.int32), np.empty(n, np.int64), np.zeros(nh, np.int64)
    rc = lib().orc_reassign_to_halos(n, _p(pos), model.boxsize, nh, _orc_halos(model), _p(rs), _p(hid), _p(perm),
                                     _p(npart))
    if rc:
        raise RuntimeError("orc_reassign_to_halos: %d" % rc)
    return hid, perm, npart


def set_vector_potential(model, pos, eta):
    """magnetic_field.c:33-69."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    a = np.empty_like(pos)
    L = lib()
    L.orc_set_vector_potential.restype = None
    L.orc_set_vector_potential.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_double, C.c_void_p]
    L.orc_set_vector_potential(len(pos), _p(pos), model.boxsize, len(model.halos), C.cast(_orc_halos(model), C.c_void_p),
                               float(eta), _p(a))
    return a


def normalise_magnetic_field(model, pos, bfld, bfld_norm, r_sample_dm, sub_first=2):
    """magnetic_field.c:71-131 on a copy of bfld: (bfld, norm, n_limited)."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    b = np.array(bfld, dtype=np.float32, order="C", copy=True)
    rg = np.array([h.r_sample for h in model.halos], np.float64)
    rd = np.ascontiguousarray(r_sample_dm, np.float64)
    norm, cnt = C.c_double(), C.c_int()
    L = lib()
    L.orc_normalise_magnetic_field.restype = None
    L.orc_normalise_magnetic_field.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.orc_normalise_magnetic_field(len(pos), _p(pos), _p(b), model.boxsize, len(model.halos),
                                   C.cast(_orc_halos(model), C.c_void_p), _p(rg), _p(rd), int(sub_first), float(bfld_norm),
                                   C.byref(norm), C.byref(cnt))
    return b, norm.value, cnt.value


class OrcSubHalo(C.Structure):
    _fields_ = [("Mtotal200", C.c_double), ("Mass200", C.c_double * 2), ("C_nfw", C.c_double), ("R200", C.c_double),
                ("Rs", C.c_double), ("A_hernq", C.c_double), ("Rho0", C.c_double), ("Beta", C.c_double),
                ("Rcore", C.c_double), ("Rcut", C.c_double), ("R_Sample", C.c_double * 2), ("Mass", C.c_double * 2),
                ("Mtotal", C.c_double), ("MassCorrFac", C.c_double), ("D_CoM", C.c_double * 3),
                ("Npart", C.c_longlong * 2), ("Have_Cuspy", C.c_int), ("Is_Stripped", C.c_int)]


class OrcSubState(C.Structure):
    _fields_ = [("Mpart", C.c_double * 2), ("Redshift", C.c_double), ("Mass_Ratio", C.c_double),
                ("GravSofteningLength", C.c_double), ("Baryon_Fraction", C.c_double), ("UnitMass", C.c_double),
                ("UnitDensity", C.c_double), ("Rho_crit0", C.c_double), ("OverdensityParameter", C.c_double),
                ("Nhalos", C.c_int), ("Cuspy", C.c_int), ("SUBHOST", C.c_int), ("Seed", C.c_ushort * 3),
                ("First", C.c_int), ("SubNhalos", C.c_int), ("Mtotal", C.c_double), ("MassFraction", C.c_double),
                ("SubNpart", C.c_longlong * 2)]


def setup_substructure(setup, seed, subhost=0):
    """Setup_Substructure (src/substructure.c:31-109) restated in oracle/tc_oracle_sub.c, run on the state a native
    set-up (toycluster_amd.hostio.Setup BEFORE its own substructure step) left behind.  `seed` = thread 0's erand48
    state.  Returns (state, [halo dicts])."""
    L = lib()
    halos = (OrcSubHalo * 128)()
    for i in range(setup.nhalos):
        h, o = setup.halo[i], halos[i]
        o.Mtotal200, o.C_nfw, o.R200, o.Rs, o.A_hernq = h.mtotal200, h.c_nfw, h.r200, h.rs, h.a_hernq
        o.Rho0, o.Beta, o.Rcore, o.Rcut, o.Mtotal, o.MassCorrFac = h.rho0, h.beta, h.rcore, h.rcut, h.mtotal, h.mass_corr_fac
        for k in range(2):
            o.Mass200[k], o.R_Sample[k], o.Mass[k], o.Npart[k] = h.mass200[k], h.r_sample[k], h.mass[k], h.npart[k]
        for k in range(3):
            o.D_CoM[k] = h.d_com[k]
        o.Have_Cuspy, o.Is_Stripped = h.have_cuspy, h.is_stripped
    st = OrcSubState()
    st.Mpart[0], st.Mpart[1] = setup.mpart[0], setup.mpart[1]
    st.Redshift, st.Mass_Ratio = setup.par.redshift, setup.par.mass_ratio
    st.GravSofteningLength, st.Baryon_Fraction = setup.grav_softening, setup.par.baryon_fraction
    st.UnitMass, st.UnitDensity = setup.unit_mass, setup.unit_mass / setup.unit_length ** 3
    st.Rho_crit0 = 3.0 / 8.0 / np.pi / 6.673e-8 * setup.h0_cgs ** 2       # src/cosmo.c:20, GSL's cgs G
    st.OverdensityParameter = setup.delta
    st.Nhalos, st.Cuspy, st.SUBHOST = setup.nhalos, setup.par.cuspy, subhost
    for k in range(3):
        st.Seed[k] = seed[k]
    L.orc_setup_substructure.argtypes = [C.POINTER(OrcSubState), C.POINTER(OrcSubHalo)]
    rc = L.orc_setup_substructure(C.byref(st), halos)
    if rc:
        raise RuntimeError("orc_setup_substructure: %d" % rc)
    fields = [f[0] for f in OrcSubHalo._fields_]
    out = []
    for i in range(st.Nhalos):
        d = {}
        for f in fields:
            v = getattr(halos[i], f)
            d[f] = list(v) if hasattr(v, "__len__") else v
        out.append(d)
    return st, out


def format_log_line(l):
    """The reference's per-iteration line, wvt_relax.c:91-92."""
    return "   #%02d: Err max=%3g mean=%03g diff=%03g step=%g" % (
        l["it"], l["err_max"], l["err_mean"], l["err_diff"], l["step"])
