"""Guards on the machine code inside libmijpeg.so (no GPU needed: the library is disassembled, not run).

v_ashr_pk_u8_i32 (gfx950) writes 16 bits of its destination and keeps the rest (tools/microbench/ashr_pk.hip prints what the
hardware does); the compiler of ROCm 7.2 assumes zeros in the other half when it matches the instruction from C code, which
gave wrong pixels in the first build of the rebuilt fused_tile_kernel.  csrc/kernels.hip therefore only uses it through
ashr_sat_pack4: a low-half instruction directly followed by the op_sel:[0,0,0,1] instruction that fills the upper half of the
same register.  This test fails when the compiler matched a lone one somewhere.
"""
import os
import re
import shutil
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "libjpeg_amd", "libmijpeg.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def gfx950_code_objects(path):
    """The device ELFs of the clang offload bundles embedded in a host shared object."""
    blob = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    at = blob.find(magic)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", blob, at + len(magic))
        p = at + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(magic, at + 1)
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_every_ashr_pk_instruction_is_half_of_a_pair():
    objs = gfx950_code_objects(LIB)
    assert objs, "no gfx950 code object found in libmijpeg.so"
    pat = re.compile(r"v_ashr_pk_u8_i32\s+(v\d+),")
    pairs = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i, co in enumerate(objs):
            f = os.path.join(tmp, f"k{i}.co")
            open(f, "wb").write(co)
            asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f], capture_output=True, text=True, check=True).stdout
            lines = [ln for ln in asm.splitlines() if ln.strip() and not ln.lstrip().startswith(("//", ";"))]
            for n, ln in enumerate(lines):
                m = pat.search(ln)
                if not m:
                    continue
                if "op_sel" in ln:
                    prev = pat.search(lines[n - 1])
                    assert prev and "op_sel" not in lines[n - 1] and prev.group(1) == m.group(1), f"upper half without its lower half: {ln.strip()}"
                    pairs += 1
                else:
                    nxt = pat.search(lines[n + 1])
                    assert nxt and "op_sel" in lines[n + 1] and nxt.group(1) == m.group(1), f"lone v_ashr_pk_u8_i32 (the compiler matched it?): {ln.strip()}"
    assert pairs > 100  # the 8-bit kernels pack through it
