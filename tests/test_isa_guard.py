"""Static guard for the hand-placed scalar prefetch of nb_force_pk_sgpr.

The kernel requests 4 bodies with `s_load_dwordx4` inside an asm statement and waits for
them in a LATER asm statement (`s_waitcnt lgkmcnt(0)`); hipcc does not model asm loads, so
nothing but the statement order stops it from reading, copying or spilling the destination
SGPRs before the data has landed (cdna_hip_programming.md §5.7 item 1).  The parity tests
would catch a miscompile on the GPU; this test catches it at build time, on the CPU box:
it disassembles the gfx950 code and checks that no instruction touches a requested SGPR
range between the request and the next lgkmcnt(0) wait.
"""
import os
import re
import shutil
import subprocess

import pytest

from conftest import PKG

CSRC = os.path.join(PKG, "csrc")
ASM = os.path.join(CSRC, "nb_engine.gfx950.s")


def kernel_bodies():
    if shutil.which("/opt/rocm/bin/hipcc") is None and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    subprocess.check_call(["make", "-C", CSRC, "-s", "asm"])
    text = open(ASM).read()
    out = {}
    # nb_force_pk_sgpr<NG, WS, PAIRS>: NG in {2, 4} packed groups, WS in {1, 4} j-splitting waves,
    # PAIRS: bodies fetched as 64-bit pairs (s_load_dwordx2) instead of quads (s_load_dwordx4)
    for m in re.finditer(r"^(_ZN2nb16nb_force_pk_sgprILi(\d)ELi(\d)ELb([01])EEE\w+):\s*;.*?$(.*?)s_endpgm", text, re.S | re.M):
        out[(int(m.group(2)), int(m.group(3)), int(m.group(4)))] = m.group(5).splitlines()
    return out


def sregs(operand_text):
    """All SGPR indices mentioned in an instruction's operand text."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", operand_text):
        regs.add(int(a))
    return regs


def test_no_instruction_touches_a_requested_sgpr_before_its_wait():
    bodies = kernel_bodies()
    assert set(bodies) == {(2, 1, 0), (2, 4, 0), (4, 1, 0), (4, 4, 0), (4, 4, 1)}, "expected all five nb_force_pk_sgpr instantiations"
    for ng, lines in bodies.items():
        pending = set()
        requests = waits = 0
        in_asm = False
        for ln in lines:
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            code = ln.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            op, _, rest = code.partition(" ")
            # only the hand-placed requests (inside asm statements) are untracked by hipcc;
            # its own scalar loads (kernel arguments) get compiler-inserted waits
            if op in ("s_load_dwordx4", "s_load_dwordx2") and in_asm:
                dst = rest.split(",")[0]
                src = ",".join(rest.split(",")[1:])
                # the address registers of a request must not be pending either
                assert not (sregs(src) & pending), (ng, code)
                pending |= sregs(dst)
                requests += 1
                continue
            if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
                pending.clear()
                waits += 1
                continue
            if op == "s_branch":
                # the textual successor is not the control-flow successor: this linear scan
                # stops here (the loop header the branch returns to opens with the wait)
                pending.clear()
                continue
            touched = sregs(rest) & pending
            assert not touched, "NG,WS=%s: `%s` touches s%s before the wait" % (ng, code, sorted(touched))
        assert requests >= 8 and waits >= 2, (ng, requests, waits)
