#!/usr/bin/env python3
"""Build step: keep every 64-bit instruction inside a loop of the gfx950 device code on an 8-byte boundary.

    python3 align_loops.py in.s out.s [--report]

Why.  Measured on MI355X (profiles/r04/README.md, "instruction alignment"): a wave issues a 64-bit encoded instruction (VOP3P
packed f32, VOP3, DPP, SMEM ...) that straddles an 8-byte boundary more slowly -- the symmetric pass's 295-dword loop ran 12 %
longer at one wave per SIMD and 5 % at two when its head sat at 4 mod 8, config 2's LDS-tile kernel lost 1.9 % between two
builds whose loops execute the same 476 instructions.  hipcc aligns loop HEADS on request (-falign-loops=32) but nothing inside a
loop, where every 32-bit instruction (s_waitcnt, s_add_u32, v_mov_b32_e32 ...) flips the parity of all that follows.

How.  The device assembly hipcc writes (-S --cuda-device-only) is assembled once to learn every instruction's size, the loops
are read off the backward branches, and an `s_nop 0` (32 bits, one issue slot of the scalar unit) is put in front of every
64-bit instruction of a loop that would start at 4 mod 8.  The result is assembled, linked and bundled by the Makefile exactly as
hipcc would have done with its own output (clang -cc1as / lld / clang-offload-bundler / -fcuda-include-gpubinary).  Inserting a
no-op only lengthens the distance between instructions: no hazard can appear that was not there, and branch offsets are
labels; the one place where a distance is hard-coded -- the +4 / +12 addends of an `s_getpc_b64` ... `sym@rel32@lo+4` /
`sym@rel32@hi+12` sequence -- is never split (nothing is inserted between the s_getpc_b64 and the @rel32@hi instruction).  Functions with fp64 arithmetic in their loops are left as hipcc wrote them (they alternate 32- and 64-bit encodings:
there the no-ops cost more than they save, measured), and so is any function that would need a no-op per 12 loop instructions.  tests/test_isa_guard.py checks the built library:
no misaligned 64-bit instruction in any loop of the packed-f32 force kernels.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("NB_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
DENSITY_LIMIT = 12          # at most one inserted no-op per this many loop instructions, else the function is left alone
DIS_LINE = re.compile(r"^\s*(\S+)\s.*//\s*([0-9A-Fa-f]+):\s+((?:[0-9A-Fa-f]{8}\s*)+)")


def base(mnemonic):
    return re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mnemonic)


def disassemble(asm_path):
    """{symbol: [(address, size, mnemonic, branch offset or None)]} of the assembled text."""
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, "a.o")
        subprocess.check_call([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm_path, "-o", obj])
        text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", obj], text=True)
    out = {}
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^\s*$)", text, re.S | re.M):
        ins = []
        for ln in m.group(2).splitlines():
            mm = DIS_LINE.match(ln)
            if not mm:
                continue
            op = mm.group(1)
            off = None
            if op.startswith("s_cbranch") or op == "s_branch":
                off = int(ln.split()[1])
            ins.append((int(mm.group(2), 16), 4 * len(mm.group(3).split()), op, off))
        out[m.group(1)] = ins
    return out


def is_instruction(line):
    t = line.strip()
    return bool(t) and not t.startswith((".", ";", "#")) and not t.split(";")[0].strip().endswith(":") and bool(t.split(";")[0].strip())


def process(src_lines, dis):
    """Returns (new lines, report rows)."""
    out, report = [], []
    i, n = 0, len(src_lines)
    func_label = re.compile(r"^([A-Za-z_][\w$.]*):")
    while i < n:
        m = func_label.match(src_lines[i])
        if not (m and m.group(1) in dis):
            out.append(src_lines[i]); i += 1
            continue
        name = m.group(1)
        j = i + 1
        while j < n and not src_lines[j].startswith(".Lfunc_end"):
            j += 1
        body = src_lines[i + 1:j]
        d = dis[name]
        # map the body's instruction lines onto the disassembly (which also holds the assembler's alignment padding)
        size_of, dis_index = {}, {}
        k = 0
        for li, ln in enumerate(body):
            if not is_instruction(ln):
                continue
            op = ln.split(";")[0].split()[0]
            while k < len(d) and base(d[k][2]) != base(op) and d[k][2] in ("s_nop", "s_code_end"):
                k += 1                                   # padding
            if k >= len(d) or not (base(d[k][2]) == base(op) or d[k][2].startswith(base(op)) or op.startswith(base(d[k][2]))):
                raise SystemExit("align_loops: %s: cannot match '%s' (line %d of the function) with '%s'" % (name, ln.strip(), li, d[k][2] if k < len(d) else "<end>"))
            size_of[li] = d[k][1]
            dis_index[k] = li
            k += 1
        # loops: backward branches of the disassembly, as ranges of body lines
        addr_to_k = {a: kk for kk, (a, _, _, _) in enumerate(d)}
        in_loop = set()
        loops = []                                       # (first body line, last body line) of every loop
        for kk, (a, size, op, off) in enumerate(d):
            if off is None or off < 32768 or kk not in dis_index:
                continue
            head = a + 4 - (65536 - off) * 4
            hk = addr_to_k.get(head)
            while hk is not None and hk not in dis_index and hk < kk:
                hk += 1                                  # the head may be a padding no-op
            if hk is None or hk not in dis_index:
                continue
            in_loop.update(range(dis_index[hk], dis_index[kk] + 1))
            loops.append((dis_index[hk], dis_index[kk]))
        # the innermost loops (those that contain no other): where a no-op is paid for on every trip
        innermost = set()
        for lo, hi in loops:
            if not any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in loops):
                innermost.update(range(lo, hi + 1))
        out.append(src_lines[i])
        # dry run first: a no-op costs an issue slot.  Left as hipcc wrote them: functions with fp64 arithmetic in their loops (they
        # alternate 32-bit v_fmac_f64_e32 / v_rsq_f64_e32 with 64-bit VOP3 instructions: aligned by no-ops nb_force_symw64 ran 3.4 %
        # SLOWER, profiles/r04/README.md) and functions that would need a no-op per DENSITY_LIMIT loop instructions or more
        def walk(emit, inner_only=False):
            parity, inserted, wide = 0, 0, 0
            pc_rel = False          # between an s_getpc_b64 and the last instruction of its sym@rel32 sequence: the +4 / +12 addends are
                                    # distances from the s_getpc_b64 -- a no-op put in between would silently move the computed address
            for li, ln in enumerate(body):
                pa = re.match(r"\.p2align\s+(\d+)", ln.strip())
                if pa:
                    if int(pa.group(1)) >= 3:
                        parity = 0
                elif li in size_of:
                    sz = size_of[li]
                    code = ln.split(";")[0]
                    if pc_rel and "@rel32@hi" in code:
                        closes = True
                    else:
                        closes = False
                    if li in in_loop and sz == 8 and not pc_rel:
                        wide += 1
                        if parity == 4:
                            if emit is not None:
                                emit.append("\ts_nop 0                                  ; align_loops.py: the next 64-bit instruction on an 8-byte boundary\n")
                            parity = 0
                            inserted += 1 if (not inner_only or li in innermost) else 0
                    parity = (parity + sz) % 8
                    if code.split()[0] == "s_getpc_b64":
                        pc_rel = True
                    elif closes:
                        pc_rel = False
                if emit is not None:
                    emit.append(ln)
            return inserted, wide
        need, wide_in_loops = walk(None)
        loop_instr = sum(1 for li in in_loop if li in size_of)
        fp64 = not os.environ.get("NB_ALIGN_F64") and any("_f64" in body[li].split(";")[0].split()[0] for li in in_loop if li in size_of)      # (NB_ALIGN_F64=1: the A/B arm that aligns them too)
        # The density test counts a function's INNERMOST loops when the whole function fails it: a cold block of an outer loop (the
        # LDS flush of nb_force_symw's resident sums: 96 ds_read / ds_write among 32-bit adds, run once per 4,096 rotation steps) must
        # not leave the rotation loops next to it unaligned.
        inner_need, _ = walk(None, inner_only=True)
        inner_instr = sum(1 for li in innermost if li in size_of)
        if need and (fp64 or (need * DENSITY_LIMIT > loop_instr and inner_need * DENSITY_LIMIT > inner_instr)):
            out.extend(body)
            report.append((name, len(in_loop), wide_in_loops, 0, need))
            i = j
            continue
        inserted, _ = walk(out)
        report.append((name, len(in_loop), wide_in_loops, inserted, 0))
        i = j
    return out, report


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).readlines()
    new, report = process(lines, disassemble(src))
    open(dst, "w").writelines(new)
    if "--report" in sys.argv:
        for name, nloop, wide, ins, skipped in report:
            if ins or skipped:
                print("%-110s %5d loop lines, %4d 64-bit instructions, %3d no-ops inserted%s" % (
                    name[:110], nloop, wide, ins, ("  (LEFT ALONE: would need %d)" % skipped) if skipped else ""))
    print("align_loops: %d functions, %d no-ops inserted, %d functions left alone (too dense)" % (
        len(report), sum(r[3] for r in report), sum(1 for r in report if r[4])))


if __name__ == "__main__":
    main()
