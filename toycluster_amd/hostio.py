"""Python access to the C host routines (toycluster_amd/host -> lib/libtchost.so) and readers for
the files they produce.  Used by the tests and by tools that prepare inputs for `toycluster_hip`."""
import ctypes as C
import os
import struct

import numpy as np

from .binding import TcHalo, TcParams, cool_core_component

_HERE = os.path.dirname(os.path.abspath(__file__))
HOSTLIB = os.path.join(_HERE, "lib", "libtchost.so")
EXE = os.path.join(_HERE, "host", "toycluster_hip")
SHIMLIB = os.path.join(_HERE, "lib", "libtcshim.so")


class ParFile(C.Structure):
    _fields_ = [("output_file", C.c_char * 512), ("ntotal", C.c_longlong), ("mtot200", C.c_double),
                ("redshift", C.c_double), ("mass_ratio", C.c_double), ("impact_param", C.c_double),
                ("zero_e_orbit_frac", C.c_double), ("cuspy", C.c_int), ("bfld_norm", C.c_double),
                ("bfld_eta", C.c_double), ("baryon_fraction", C.c_double), ("unit_length", C.c_double),
                ("unit_mass", C.c_double), ("unit_vel", C.c_double), ("double_beta", C.c_int), ("pad_", C.c_int),
                ("rho0_fac", C.c_double), ("rc_fac", C.c_double)]


class Snapshot(C.Structure):
    _fields_ = [("npart", C.c_longlong * 6), ("mpart", C.c_double * 6), ("boxsize", C.c_double),
                ("hubble_param", C.c_double), ("pos", C.c_void_p), ("vel", C.c_void_p), ("id", C.c_void_p),
                ("u", C.c_void_p), ("rho", C.c_void_p), ("hsml", C.c_void_p), ("bfld", C.c_void_p),
                ("rho_model", C.c_void_p)]


class HaloSetup(C.Structure):
    _fields_ = [("mtotal200", C.c_double), ("mass200", C.c_double * 2), ("c_nfw", C.c_double), ("r200", C.c_double),
                ("rs", C.c_double), ("a_hernq", C.c_double), ("rho0", C.c_double), ("beta", C.c_double),
                ("rcore", C.c_double), ("rcut", C.c_double), ("r_sample", C.c_double * 2), ("mass", C.c_double * 2),
                ("mtotal", C.c_double), ("mass_corr_fac", C.c_double), ("d_com", C.c_double * 3),
                ("npart", C.c_longlong * 2), ("have_cuspy", C.c_int), ("is_stripped", C.c_int),
                ("rho0_cc", C.c_double), ("rc_cc", C.c_double)]


class Setup(C.Structure):
    _fields_ = [("par", ParFile), ("unit_length", C.c_double), ("unit_mass", C.c_double), ("unit_vel", C.c_double),
                ("unit_time", C.c_double), ("h_100", C.c_double), ("omega_m", C.c_double), ("omega_l", C.c_double),
                ("h0_cgs", C.c_double), ("rho_crit", C.c_double), ("delta", C.c_double), ("nhalos", C.c_int),
                ("pad_", C.c_int), ("halo", HaloSetup * 72), ("boxsize", C.c_double), ("mtotal", C.c_double),
                ("mpart", C.c_double * 2), ("npart", C.c_longlong * 2),
                ("sub_first", C.c_int), ("sub_nhalos", C.c_int), ("subhost", C.c_int), ("pad2_", C.c_int),
                ("sub_mtotal", C.c_double), ("sub_mass_fraction", C.c_double), ("grav_softening", C.c_double),
                ("sub_npart", C.c_longlong * 2)]


def setup_system(parfile_path, overrides=None):
    """Set_units + Set_cosmology + Setup of the reference, natively (host/tc_setup.c). `overrides` patches
    parameter-file fields (e.g. {"ntotal": 200000, "mass_ratio": 0.3125}) before the set-up runs."""
    L = _lib()
    p = ParFile()
    err = C.create_string_buffer(1024)
    rc = L.tc_read_param_file(parfile_path.encode(), C.byref(p), err, 1024)
    if rc:
        raise RuntimeError(err.value.decode())
    for k, v in (overrides or {}).items():
        setattr(p, k, v)
    s = Setup()
    L.tc_setup_system.argtypes = [C.POINTER(ParFile), C.POINTER(Setup)]
    L.tc_setup_system(C.byref(p), C.byref(s))
    return s


def sample_gas(setup, nthreads=1, seed0=None):
    """Make_positions (gas) + Make_IDs + Shift_Origin; `seed0` = thread 0's erand48 state as left by
    setup_substructure (None: fresh stream)."""
    L = _lib()
    n = int(setup.npart[0])
    pos = np.empty((n, 3), np.float32)
    ids = np.empty(n, np.int32)
    L.tc_sample_gas_seeded.argtypes = [C.POINTER(Setup), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    s0 = None if seed0 is None else (C.c_ushort * 3)(*seed0)
    L.tc_sample_gas_seeded(C.byref(setup), int(nthreads), s0, pos.ctypes.data, ids.ctypes.data)
    return pos, ids


def setup_substructure(setup, subhost=0):
    """Setup_Substructure (src/substructure.c; the reference's -DSUBSTRUCTURE -DSUBHOST=n build) on a native
    set-up, in place.  Returns thread 0's erand48 state afterwards (pass it to sample_gas)."""
    L = _lib()
    seed = (C.c_ushort * 3)()
    L.tc_thread_seed.argtypes = [C.c_int, C.c_void_p]
    L.tc_thread_seed(0, seed)
    L.tc_setup_substructure.argtypes = [C.POINTER(Setup), C.c_int, C.c_void_p]
    rc = L.tc_setup_substructure(C.byref(setup), int(subhost), seed)
    if rc:
        raise RuntimeError("tc_setup_substructure: %d" % rc)
    return tuple(seed)


def setup_to_model(setup):
    """ClusterModel (the scalars the hot path reads) from a native set-up."""
    from .model import ClusterModel, Halo
    halos = []
    for i in range(setup.nhalos):
        h = setup.halo[i]
        halos.append(Halo(rho0=h.rho0, beta=h.beta, rcore=h.rcore, rcut=h.rcut, d_com=tuple(h.d_com),
                          r_sample=h.r_sample[0], mass_gas=h.mass[0], have_cuspy=h.have_cuspy))
    db = bool(setup.par.double_beta)
    return ClusterModel(boxsize=setup.boxsize, halos=halos, mpart_gas=setup.mpart[0], mtotal=setup.mtotal,
                        bfld_eta=setup.par.bfld_eta, name="native-setup",
                        rho0_fac=setup.par.rho0_fac if db else 0.0, rc_fac=setup.par.rc_fac if db else 0.0)


def _lib():
    if not os.path.exists(HOSTLIB):
        raise RuntimeError("%s missing: run __graft_entry__.build()" % HOSTLIB)
    L = C.CDLL(HOSTLIB)
    L.tc_read_param_file.argtypes = [C.c_char_p, C.POINTER(ParFile), C.c_char_p, C.c_size_t]
    L.tc_write_snapshot.argtypes = [C.c_char_p, C.POINTER(Snapshot)]
    return L


def read_param_file(path):
    """Returns (rc, dict-or-None, message) with the reference's semantics (io.c:298-507)."""
    p = ParFile()
    err = C.create_string_buffer(1024)
    rc = _lib().tc_read_param_file(path.encode(), C.byref(p), err, 1024)
    if rc:
        return rc, None, err.value.decode()
    d = {k: getattr(p, k) for k, _ in ParFile._fields_}
    d["output_file"] = d["output_file"].decode()
    return 0, d, ""


def write_snapshot(path, npart, mpart, boxsize, pos, vel, ids, u, rho, hsml, bfld, rho_model, hubble=0.7):
    keep = [np.ascontiguousarray(a, dtype=t) for a, t in ((pos, np.float32), (vel, np.float32), (ids, np.int32),
            (u, np.float32), (rho, np.float32), (hsml, np.float32), (bfld, np.float32), (rho_model, np.float32))]
    s = Snapshot()
    for i in range(6):
        s.npart[i] = int(npart[i])
        s.mpart[i] = float(mpart[i])
    s.boxsize, s.hubble_param = boxsize, hubble
    (s.pos, s.vel, s.id, s.u, s.rho, s.hsml, s.bfld, s.rho_model) = [a.ctypes.data for a in keep]
    return _lib().tc_write_snapshot(path.encode(), C.byref(s))


def read_snapshot(path):
    """Minimal Gadget-2 format-2 reader: {label: raw bytes}, plus the parsed header."""
    blocks = {}
    order = []
    with open(path, "rb") as f:
        data = f.read()
    off = 0
    while off < len(data):
        m0, = struct.unpack_from("<i", data, off)
        assert m0 == 8, "label record marker"
        label = data[off + 4:off + 8].decode()
        nxt, m1 = struct.unpack_from("<ii", data, off + 8)
        assert m1 == 8
        off += 16
        n0, = struct.unpack_from("<i", data, off)
        assert n0 + 8 == nxt, "next-block length = payload + 2 markers"
        payload = data[off + 4:off + 4 + n0]
        n1, = struct.unpack_from("<i", data, off + 4 + n0)
        assert n1 == n0
        off += 8 + n0
        blocks[label] = payload
        order.append(label)
    h = blocks["HEAD"]
    assert len(h) == 256
    header = dict(npart=struct.unpack_from("<6i", h, 0), mass=struct.unpack_from("<6d", h, 24),
                  time=struct.unpack_from("<d", h, 72)[0], redshift=struct.unpack_from("<d", h, 80)[0],
                  npartTotal=struct.unpack_from("<6I", h, 96), num_files=struct.unpack_from("<i", h, 124)[0],
                  BoxSize=struct.unpack_from("<d", h, 128)[0], Omega0=struct.unpack_from("<d", h, 136)[0],
                  OmegaLambda=struct.unpack_from("<d", h, 144)[0], HubbleParam=struct.unpack_from("<d", h, 152)[0])
    return header, blocks, order


def write_state(path, model, pos, ids):
    """The hot-path state file consumed by `toycluster_hip` (host/tc_state.c)."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    par = TcParams(model.boxsize, model.mpart_gas, model.mtotal, getattr(model, "bfld_eta", 0.5), len(model.halos), 0)
    with open(path, "wb") as f:
        f.write(b"TCSTATE2")
        f.write(struct.pack("<q", len(ids)))
        f.write(bytes(par))
        for h in model.halos:
            th = TcHalo()
            th.mass_gas = h.mass_gas
            for c in range(3):
                th.d_com[c] = h.d_com[c]
            th.rho0, th.beta, th.rcore, th.rcut, th.have_cuspy = h.rho0, h.beta, h.rcore, h.rcut, int(h.have_cuspy)
            th.rho0_cc, th.rc_cc = cool_core_component(model, h)
            f.write(bytes(th))
        f.write(pos.tobytes())
        f.write(ids.tobytes())
        if all(getattr(h, "r_sample", 0) > 0 for h in model.halos):      # optional trailer: Halo[i].R_Sample[0]
            f.write(np.array([h.r_sample for h in model.halos], "<f8").tobytes())


def _model_structs(model):
    par = TcParams(model.boxsize, model.mpart_gas, model.mtotal, getattr(model, "bfld_eta", 0.5), len(model.halos), 0)
    halos = (TcHalo * max(1, len(model.halos)))()
    for k, h in enumerate(model.halos):
        halos[k].mass_gas = h.mass_gas
        for c in range(3):
            halos[k].d_com[c] = h.d_com[c]
        halos[k].rho0, halos[k].beta, halos[k].rcore, halos[k].rcut = h.rho0, h.beta, h.rcore, h.rcut
        halos[k].have_cuspy = int(h.have_cuspy)
        halos[k].rho0_cc, halos[k].rc_cc = cool_core_component(model, h)
    return par, halos


def reassign_particles_to_halos(model, pos):
    """Reassign_particles_to_halos for the gas block (host/tc_reassign.c; src/positions.c:264-445):
    returns (halo_id[n], perm[n], npart[nhalos]) with new[i] = old[perm[i]]."""
    L = _lib()
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    n, nh = pos.shape[0], len(model.halos)
    par, halos = _model_structs(model)
    rs = np.array([h.r_sample for h in model.halos], np.float64)
    hid, perm, npart = np.empty(n, np.int32), np.empty(n, np.uint64), np.zeros(max(nh, 1), np.int64)
    L.tc_reassign_particles_to_halos.argtypes = [C.POINTER(TcParams), C.POINTER(TcHalo), C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.tc_reassign_particles_to_halos(C.byref(par), halos, rs.ctypes.data, n, pos.ctypes.data, hid.ctypes.data,
                                          perm.ctypes.data, npart.ctypes.data)
    if rc:
        raise RuntimeError("tc_reassign_particles_to_halos: %d" % rc)
    return hid, perm.astype(np.int64), npart[:nh]


def normalise_magnetic_field(model, pos, bfld, bfld_norm, r_sample_dm=None, sub_first=2):
    """normalise_magnetic_field (src/magnetic_field.c:71-131) of the C host library, on copies.
    Returns (bfld, norm, n_limited)."""
    L = _lib()
    par, halos = _model_structs(model)
    pos = np.ascontiguousarray(pos, np.float32)
    b = np.array(bfld, dtype=np.float32, order="C", copy=True)
    rg = np.array([h.r_sample for h in model.halos], np.float64)
    rd = None if r_sample_dm is None else np.ascontiguousarray(r_sample_dm, np.float64)
    norm, cnt = C.c_double(), C.c_longlong()
    L.tc_normalise_magnetic_field.argtypes = [C.POINTER(TcParams), C.POINTER(TcHalo), C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_double, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                                              C.POINTER(C.c_longlong)]
    L.tc_normalise_magnetic_field(C.byref(par), halos, rg.ctypes.data, None if rd is None else rd.ctypes.data,
                                  int(sub_first), float(bfld_norm), len(pos), pos.ctypes.data, b.ctypes.data,
                                  C.byref(norm), C.byref(cnt))
    return b, norm.value, cnt.value


def set_magnetic_vector_potential(model, pos, bfld_eta):
    """set_magnetic_vector_potential (src/magnetic_field.c:33-69) of the C host library."""
    L = _lib()
    par, halos = _model_structs(model)
    pos = np.ascontiguousarray(pos, np.float32)
    a = np.empty((len(pos), 3), np.float32)
    L.tc_set_magnetic_vector_potential.argtypes = [C.POINTER(TcParams), C.POINTER(TcHalo), C.c_double, C.c_size_t,
                                                   C.c_void_p, C.c_void_p]
    L.tc_set_magnetic_vector_potential.restype = None
    L.tc_set_magnetic_vector_potential(C.byref(par), halos, float(bfld_eta), len(pos), pos.ctypes.data, a.ctypes.data)
    return a


def heapsort_index_i32(keys):
    L = _lib()
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    p = np.empty(len(keys), np.uint64)
    L.tc_heapsort_index_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.tc_heapsort_index_i32(p.ctypes.data, keys.ctypes.data, len(keys))
    return p.astype(np.int64)


class State(C.Structure):
    _fields_ = [("par", TcParams), ("halos", C.POINTER(TcHalo)), ("ngas", C.c_int64), ("pos", C.POINTER(C.c_float)),
                ("id", C.POINTER(C.c_int32)), ("r_sample", C.POINTER(C.c_double)),
                ("r_sample_dm", C.POINTER(C.c_double)), ("sub_first", C.c_int)]


def read_state_header(path):
    """Load a state file through the C reader (host/tc_state.c) and report what it holds."""
    L = _lib()
    st = State()
    err = C.create_string_buffer(1024)
    L.tc_read_state.argtypes = [C.c_char_p, C.POINTER(State), C.c_char_p, C.c_size_t]
    rc = L.tc_read_state(path.encode(), C.byref(st), err, 1024)
    if rc:
        raise RuntimeError(err.value.decode())
    nh = st.par.nhalos
    out = {"ngas": int(st.ngas), "nhalos": nh, "boxsize": st.par.boxsize, "has_r_sample": bool(st.r_sample),
           "r_sample": [st.r_sample[i] for i in range(nh)] if st.r_sample else None}
    L.tc_free_state.argtypes = [C.POINTER(State)]
    L.tc_free_state(C.byref(st))
    return out
