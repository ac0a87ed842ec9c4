#!/usr/bin/env python3
"""Which 64-bit instructions of the engine's hot loops are NOT 8-byte aligned?

Measured on gfx950 (profiles/r04/README.md, "instruction alignment"): a wave issues a 64-bit VALU instruction (VOP3P packed f32,
VOP3, DPP) that does not start on an 8-byte boundary more slowly -- the symmetric pass's 295-dword loop ran 12 % longer at one
wave per SIMD, 5 % at two, when its head sat at 4 mod 8 (every packed instruction misaligned) than with one `s_nop` in front of it.
hipcc aligns nothing inside a function, so the parity of a loop is luck: a 32-bit instruction anywhere before it flips it.

This tool disassembles the device code of libnbody3d_hip.so (or of a code object given on the command line), finds every loop
(a backward branch) of at least MIN_DWORDS dwords and reports the share of its 64-bit instructions that are misaligned.

    python tools/loop_parity.py [--min 64] [--kernel SUBSTR] [lib.so | code.elf]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def device_disassembly(path):
    """llvm-objdump -d of the gfx950 code object inside a HIP shared library / bundled object (or of a plain code object)."""
    with tempfile.TemporaryDirectory() as tmp:
        elf = os.path.join(tmp, "dev.elf")
        data = open(path, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        pos = data.find(magic)
        if data[:4] == b"\x7fELF" and pos < 0:
            elf = path                                    # already a device code object
        elif data[:4] == b"\x7fELF":
            # a host shared library: the fat binary sits in the .hip_fatbin section; cut the bundle out and unbundle it
            bundle = os.path.join(tmp, "bundle.bin")
            open(bundle, "wb").write(data[pos:])
            _unbundle(bundle, elf)
        else:
            _unbundle(path, elf)
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", elf], text=True)


def _unbundle(bundle, out):
    """The code object of TARGET out of a clang offload bundle (parsed by hand: the bundler wants a file type it can recognise)."""
    data = open(bundle, "rb").read()
    assert data[:24] == b"__CLANG_OFFLOAD_BUNDLE__", "not an offload bundle"
    n = int.from_bytes(data[24:32], "little")
    off = 32
    for _ in range(n):
        o = int.from_bytes(data[off:off + 8], "little")
        s = int.from_bytes(data[off + 8:off + 16], "little")
        t = int.from_bytes(data[off + 16:off + 24], "little")
        triple = data[off + 24:off + 24 + t].decode()
        off += 24 + t
        if triple.startswith("hipv4-amdgcn") and "gfx950" in triple:
            open(out, "wb").write(data[o:o + s])
            return
    raise RuntimeError("no gfx950 code object in the bundle")


LINE = re.compile(r"^\s*(\S+)\s.*//\s*([0-9A-Fa-f]+):\s+((?:[0-9A-Fa-f]{8}\s*)+)")


def loops(disassembly, min_dwords=64):
    """[(kernel, head address, dwords, 64-bit instructions, misaligned ones, 32-bit instructions)] for every backward branch."""
    out = []
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^\s*$)", disassembly, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ins = []                                          # (address, size in bytes, mnemonic, operand text)
        for ln in body.splitlines():
            mm = LINE.match(ln)
            if not mm:
                continue
            size = 4 * len(mm.group(3).split())
            ins.append((int(mm.group(2), 16), size, mm.group(1), ln))
        addr_index = {a: i for i, (a, _, _, _) in enumerate(ins)}
        for i, (a, size, op, ln) in enumerate(ins):
            if not (op.startswith("s_cbranch") or op == "s_branch"):
                continue
            off = int(ln.split()[1])
            if off < 32768:
                continue                                  # forward branch
            head = a + 4 - (65536 - off) * 4
            if head not in addr_index or (a + 4 - head) // 4 < min_dwords:
                continue
            seg = ins[addr_index[head]:i + 1]
            wide = [x for x in seg if x[1] == 8]
            bad = [x for x in wide if x[0] % 8]
            out.append((name, head, (a + 4 - head) // 4, len(wide), len(bad), len(seg) - len(wide)))
    return out


def main():
    args = sys.argv[1:]
    min_dw, kern, path = 64, None, os.path.join(ROOT, "nbody3d-webgpu_amd", "csrc", "libnbody3d_hip.so")
    while args:
        a = args.pop(0)
        if a == "--min":
            min_dw = int(args.pop(0))
        elif a == "--kernel":
            kern = args.pop(0)
        else:
            path = a
    for name, head, dwords, wide, bad, narrow in loops(device_disassembly(path), min_dw):
        if kern and kern not in name:
            continue
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void nb::", "")
        print("%-58s loop @%x (%d mod 8) %4d dwords: %3d of %3d 64-bit instructions misaligned (%3.0f %%), %2d 32-bit" % (
            short[-58:], head, head % 8, dwords, bad, wide, 100.0 * bad / max(wide, 1), narrow))


if __name__ == "__main__":
    main()
