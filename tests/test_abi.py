"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/tcgpu.h
declares.  No compute calls here (there is no GPU)."""
import ctypes
import os
import re

import pytest

from toycluster_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tcgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tcgpu_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(binding.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = ctypes.CDLL(binding.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "libtcgpu.so does not export %s" % s
    assert sorted(binding.EXPORTS) == syms, set(binding.EXPORTS) ^ set(syms)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(binding.TcParams) == 40
    assert ctypes.sizeof(binding.TcHalo) == 88      # 72 + rho0_cc, rc_cc (the -DDOUBLE_BETA_COOL_CORES component)
    assert ctypes.sizeof(binding.TcIterLog) == 40
    assert ctypes.sizeof(binding.TcDensityStats) == 32


def test_no_gpu_fails_loudly():
    """Without a device the product raises; it never falls back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.TcGpuError):
        binding.TcGpu(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "toycluster_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "tc_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_code_object_guard_accepts_the_libraries_and_refuses_a_tight_limit():
    """tools/check_codeobj.py (run by the link step of csrc/Makefile): every kernel of both libraries stays inside the
    short-branch range and carries no long-branch expansion; with an artificially small limit the
    same tool must fail -- i.e. the guard really looks at the kernels."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_codeobj.py")
    for name in ("libtcgpu.so", "libtcgpu_m4.so"):
        lib = os.path.join(os.path.dirname(binding.LIB_PATH), name)
        if not os.path.exists(lib):
            import __graft_entry__ as g
            g.build()
        ok = subprocess.run([sys.executable, tool, lib], capture_output=True, text=True)
        assert ok.returncode == 0, ok.stderr[-500:]
        assert "k_iter" in ok.stdout
        tight = subprocess.run([sys.executable, tool, lib, "--limit", "50000"], capture_output=True, text=True)
        assert tight.returncode == 1 and "FAIL" in tight.stderr


def test_bench_contract_surface_and_profile_guard():
    """bench.py keeps the driver's flags; the counter-derived figures it quotes from profiles/dominant_kernel.json are
    refused when the kernel sources have changed since the profile was taken (tools/roofline_valu.py hashes them) --
    the committed profile must belong to the committed sources."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
    sys.path.insert(0, ROOT)
    from tools import roofline_valu
    pj = json.load(open(os.path.join(ROOT, "profiles", "dominant_kernel.json")))
    assert pj["kernel"] == "k_iter" and pj["n_particles"] == 2_000_000
    assert pj["kernel_sources_sha256"] == roofline_valu.kernel_sources_sha256(ROOT), \
        "kernel sources changed since profiles/dominant_kernel.json was taken: rerun tools/profile_round2.sh + tools/roofline_valu.py"
