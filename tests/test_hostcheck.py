"""CPU checks of toycluster_amd/csrc/tc_math.h -- the scalar arithmetic the HIP kernels compile --
built for the host (tests/hostcheck.c) and compared with the oracle bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from toycluster_amd import model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hc(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hc") / "libhostcheck.so")
    # same floating-point contract as the device build: no contraction; -mfma so fmaf() is one rounding
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "hostcheck.c"), "-lm"])
    L = C.CDLL(so)
    f, d, u64 = C.c_float, C.c_double, C.c_uint64
    L.hc_peano_key.argtypes = [f, f, f, d, C.POINTER(u64), C.POINTER(u64)]
    L.hc_peano_key_lut.argtypes = [f, f, f, d, C.POINTER(u64), C.POINTER(u64)]
    L.hc_common_levels.argtypes = [u64, u64, u64, u64]
    L.hc_fdiv.argtypes = [f, f]; L.hc_fdiv.restype = f
    L.hc_wc6.argtypes = [f, f]; L.hc_wc6.restype = f
    L.hc_dwc6.argtypes = [f, f]; L.hc_dwc6.restype = f
    L.hc_wvt_wc6.argtypes = [f, f]; L.hc_wvt_wc6.restype = d
    return L


def test_device_key_arithmetic_matches_oracle(hc):
    rng = np.random.default_rng(0)
    box = 13923.0
    pos = (rng.random((4000, 3)) * box).astype(np.float32)
    pos[0] = 0; pos[1] = np.float32(box); pos[2] = (np.float32(box), 0, np.float32(box / 2))
    keys = []
    for p in pos:
        hi, lo = C.c_uint64(), C.c_uint64()
        hc.hc_peano_key(p[0], p[1], p[2], box, C.byref(hi), C.byref(lo))
        k = (hi.value << 64) | lo.value
        x, y, z = (float(np.float64(c) / box) for c in p)
        assert k == O.peano_key(x, y, z)
        # the table-driven form the kernels run (24 orientations, two levels per look-up): the same 128 bits,
        # coordinates on the box faces included
        hc.hc_peano_key_lut(p[0], p[1], p[2], box, C.byref(hi), C.byref(lo))
        assert ((hi.value << 64) | lo.value) == k
        keys.append(k)
    # shared Hilbert levels = floor(common leading key bits / 3)
    for a, b in zip(keys[:500], keys[1:501]):
        lv = hc.hc_common_levels(a >> 64, a & (2**64 - 1), b >> 64, b & (2**64 - 1))
        x = a ^ b
        want = 42 if x == 0 else (128 - x.bit_length()) // 3
        assert lv == want


def test_correctly_rounded_quotient_without_divide(hc):
    rng = np.random.default_rng(1)
    b = np.float32(rng.random(300) * 3000 + 1e-3)
    for bb in b:
        a = np.float32(rng.random(300) * bb * 1.3)
        for aa in a:
            assert np.float32(hc.hc_fdiv(aa, bb)) == np.float32(aa) / np.float32(bb)
    for bits in (0x3FFFFFFF, 0x40FFFFFF):            # significand all ones: falls back to the divide
        bb = np.array([bits], np.uint32).view(np.float32)[0]
        assert np.float32(hc.hc_fdiv(np.float32(1.7), bb)) == np.float32(1.7) / bb


def test_kernels_bit_identical_to_oracle(hc):
    L = O.lib()
    L.orc_wc6.argtypes = [C.c_float, C.c_float]; L.orc_wc6.restype = C.c_float
    L.orc_dwc6.argtypes = [C.c_float, C.c_float]; L.orc_dwc6.restype = C.c_float
    L.orc_wvt_wc6.argtypes = [C.c_float, C.c_float]; L.orc_wvt_wc6.restype = C.c_double
    rng = np.random.default_rng(2)
    h = np.float32(rng.random(2000) * 4000 + 10)
    r = np.float32(rng.random(2000) * h)
    r[:20] = 0
    r[20:40] = h[20:40]
    for rr, hh in zip(r, h):
        assert hc.hc_wc6(rr, hh) == L.orc_wc6(rr, hh)
        assert hc.hc_dwc6(rr, hh) == L.orc_dwc6(rr, hh)
        assert hc.hc_wvt_wc6(rr, hh) == L.orc_wvt_wc6(rr, hh)
    assert hc.hc_wc6(np.float32(5), np.float32(5)) == 0 and hc.hc_dwc6(np.float32(5), np.float32(5)) == 0
