#!/usr/bin/env python3
"""Build-time guard on the gfx950 code objects inside libtcgpu.so.

    python3 tools/check_codeobj.py toycluster_amd/lib/libtcgpu.so [--limit BYTES] [--report FILE]

Lists every kernel of every embedded code object with its code size and fails (exit 1) when

  * a kernel's code is larger than the short-branch range of s_branch / s_cbranch_* (simm16 dwords =
    +-131 068 bytes): beyond it the compiler has to expand branches into s_getpc_b64 / s_add_u32 /
    s_addc_u32 / s_setpc_b64 sequences on a reserved SGPR pair.  No kernel of this library has been
    validated on hardware in that regime, and the one variant that was ever run there (a 270 KB two-call-site
    build of k_iter, round 1) hung for a reason that was never established (DESIGN.md section 4.1,
    "The round-1 hang").  Until a long-branch kernel has passed the GPU suite the build refuses to make one.
  * a kernel contains such an expansion anyway (s_setpc_b64 outside a function return).

No GPU needed: works on the ELF symbol table and the disassembly (llvm-readelf / llvm-objdump of ROCm's LLVM).
"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
SHORT_BRANCH_RANGE = 32767 * 4          # bytes reachable by a 16-bit signed dword offset


def fatbin_section(path):
    with tempfile.NamedTemporaryFile(suffix=".bin") as t:
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, t.name],
                       check=True)
        return open(t.name, "rb").read()


def device_code_objects(blob):
    """Every amdgcn entry of every offload bundle in the .hip_fatbin section (one bundle per translation unit)."""
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "amdgcn" in ident and size:
                out.append((ident, blob[pos + off:pos + off + size]))
        pos = blob.find(MAGIC, pos + 1)
    return out


def kernels(co_bytes):
    """(name, code bytes, n_long_branches) of every kernel (symbol with a .kd descriptor) of a code object."""
    with tempfile.NamedTemporaryFile(suffix=".co") as t:
        t.write(co_bytes)
        t.flush()
        sym = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", t.name], check=True, capture_output=True,
                             text=True).stdout
        funcs, kds = {}, set()
        for line in sym.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC":
                funcs[f[7]] = int(f[2])
            if len(f) >= 8 and f[7].endswith(".kd"):
                kds.add(f[7][:-3])
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", t.name], check=True,
                             capture_output=True, text=True).stdout
    longbr, cur = {}, None
    for line in dis.splitlines():
        if line.endswith(">:") and "<" in line:
            cur = line[line.index("<") + 1:-2]
        elif cur and "s_setpc_b64" in line and "s[30:31]" not in line:     # s[30:31] = the return address
            longbr[cur] = longbr.get(cur, 0) + 1
    return [(k, funcs.get(k, 0), longbr.get(k, 0)) for k in sorted(kds)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--limit", type=int, default=SHORT_BRANCH_RANGE)
    ap.add_argument("--report")
    a = ap.parse_args()
    rows, bad = [], []
    for ident, co in device_code_objects(fatbin_section(a.lib)):
        for name, size, nlong in kernels(co):
            rows.append((size, nlong, name))
            if size > a.limit or nlong:
                bad.append((size, nlong, name))
    rows.sort(reverse=True)
    text = ["%8d B  long-branches=%d  %s" % r for r in rows]
    text.append("limit %d B (short-branch range of s_branch); %d kernels" % (a.limit, len(rows)))
    if a.report:
        open(a.report, "w").write("\n".join(text) + "\n")
    print("\n".join(text[:6] + text[-1:]))
    if not rows:
        print("check_codeobj: no gfx950 kernels found in %s" % a.lib, file=sys.stderr)
        return 1
    if bad:
        for r in bad:
            print("check_codeobj: FAIL %d B, %d long branches: %s" % r, file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
