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
    assert set(bodies) == {(2, 1, 0), (2, 4, 0), (2, 4, 1), (4, 1, 0), (4, 4, 0), (4, 4, 1)}, "expected all six nb_force_pk_sgpr instantiations"
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


def jpk_bodies():
    if shutil.which("/opt/rocm/bin/hipcc") is None and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    if not os.path.exists(ASM):
        subprocess.check_call(["make", "-C", CSRC, "-s", "asm"])
    text = open(ASM).read()
    return {int(m.group(1)): m.group(2).splitlines()
            for m in re.finditer(r"^_ZN2nb11nb_step_jpkILi(\d+)EEE\w+:\s*;.*?$(.*?)s_endpgm", text, re.S | re.M)}


def vregs(operand_text):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", operand_text):
        regs.add(int(a))
    return regs


def test_jpk_scalar_requests_and_the_asynchronous_sink_register():
    """nb_step_jpk: (1) the s_load_dwordx8 requests of a 4-pair unit obey the same rule as above;
    (2) the L2 warm-up loads (`global_load_dword` inside an asm statement) complete asynchronously into
    one VGPR: nothing may read or write that register until the explicit `s_waitcnt vmcnt(0)` asm --
    a register the allocator had recycled was overwritten by a late return on the GPU (a memory
    fault, fixed by keeping the register live across the statements); (3) the inner loop carries
    exactly 96 packed + 16 v_rsq_f32 (VOP3 encoding) VALU instructions per two units and no v_mov."""
    bodies = jpk_bodies()
    assert set(bodies) == {4, 8, 16}
    for ws, lines in bodies.items():
        pending, sink, in_asm = set(), set(), False
        requests = waits = sink_loads = 0
        loop, in_loop = [], False
        for ln in lines:
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            code = ln.split(";")[0].strip()
            if not code or code.startswith("."):
                continue
            if code.endswith(":"):
                in_loop = False
                continue
            op, _, rest = code.partition(" ")
            if op == "s_load_dwordx8" and in_asm:
                dst, src = rest.split(",")[0], ",".join(rest.split(",")[1:])
                assert not (sregs(src) & pending), (ws, code)
                pending |= sregs(dst)
                requests += 1
                in_loop = True
                continue
            if op == "global_load_dword" and in_asm:
                sink |= vregs(rest.split(",")[0])
                sink_loads += 1
                continue
            if op == "s_waitcnt" and "vmcnt(0)" in rest and in_asm:
                sink.clear()
                continue
            if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
                pending.clear()
                waits += 1
                continue
            if op in ("s_branch",):
                pending.clear()
                continue
            assert not (sregs(rest) & pending), "WS=%d: `%s` touches a requested SGPR before the wait" % (ws, code)
            if sink and not op.startswith("s_") and op != "global_load_dword":
                assert not (vregs(rest) & sink), "WS=%d: `%s` touches the sink VGPR before vmcnt(0)" % (ws, code)
        assert requests >= 12 and waits >= 3 and sink_loads == 1, (ws, requests, waits, sink_loads)
        # the steady-state loop: between the loop label that precedes the first in-loop wait and its back edge
        text = "\n".join(lines)
        m = re.search(r"^(\.LBB\d+_\d+):[^\n]*\n(?:(?!^\.LBB).*\n)*?\s*s_cbranch_scc\d \1\s*$", text, re.M)
        assert m, ws
        ops = [l.split(";")[0].strip().split(" ")[0] for l in m.group(0).splitlines()]
        ops = [o for o in ops if o.startswith("v_")]
        # (v_rsq_f32 in its 64-bit VOP3 form, nb_rsq: a 32-bit one would flip the 8-byte parity of everything behind it)
        assert ops.count("v_rsq_f32_e64") == 16 and sum(o.startswith("v_pk_") for o in ops) == 96 and len(ops) == 112, (ws, len(ops))


def test_no_packed_multiply_reads_a_reciprocal_square_root_issued_right_before_it():
    """The one-pair-per-lane LDS kernels multiply the mass in with an asm v_pk_mul_f32 (high-half broadcast) whose
    operand comes out of v_rsq_f32.  gfx950 needs a wait state between a transcendental and the VALU instruction
    that reads its result, and hipcc does not insert one in front of an asm statement: in every packed kernel, no
    v_pk_mul_f32 ... op_sel:[1,0] may directly follow a v_rsq_f32 that wrote one of its source registers."""
    if shutil.which("/opt/rocm/bin/hipcc") is None and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    if not os.path.exists(ASM):
        subprocess.check_call(["make", "-C", CSRC, "-s", "asm"])
    text = open(ASM).read()
    kernels = asm_muls = 0
    for m in re.finditer(r"^(_ZN2nb(?:13nb_step_fused|11nb_force_pk)ILi1ELi\d+ELi\d+EEE\w+):\s*;.*?$(.*?)s_endpgm", text, re.S | re.M):
        kernels += 1
        prev_op, prev_dst = None, set()
        for ln in m.group(2).splitlines():
            code = ln.split(";")[0].strip()
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            op, _, rest = code.partition(" ")
            if op == "v_pk_mul_f32" and "op_sel:[1,0] op_sel_hi:[1,1]" in rest:
                asm_muls += 1
                srcs = vregs(",".join(rest.split("op_sel")[0].split(",")[1:]))
                assert not (prev_op is not None and prev_op.startswith("v_rsq_f32") and (prev_dst & srcs)), (m.group(1), code)
            prev_op, prev_dst = op, vregs(rest.split(",")[0])
    assert kernels >= 20 and asm_muls >= 4 * kernels, (kernels, asm_muls)


def test_symmetric_pass_rotation_loop_is_the_pair_arithmetic_and_the_rotation_only():
    """The rotation loops of the symmetric force pass must carry exactly: per packed group of residents 16 packed
    instructions + 2 v_rsq_f32 (f64: 19 double-precision instructions + v_rsq_f64 per resident), 10 (f64: 14) v_mov_b32_dpp
    wave_ror:1 per traveler and step -- and nothing else: no scratch access (a second loop in the same kernel once made the
    register allocator spill INTO this loop), no v_mov_b32 copies, no LDS traffic.  The measured issue rate (4.41 cycles per VALU
    instruction, VALU instructions = 1.0015 x the pair arithmetic: profiles/r03/rocprof_f32_default) is this count."""
    if shutil.which("/opt/rocm/bin/hipcc") is None and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    if not os.path.exists(ASM):
        subprocess.check_call(["make", "-C", CSRC, "-s", "asm"])
    text = open(ASM).read()

    def innermost_loops(body):
        lines = [l.split(";")[0].strip() for l in body.splitlines()]
        lines = [l for l in lines if l and (not l.startswith(".") or l.startswith(".LBB"))]
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
        loops = []
        for i, l in enumerate(lines):
            m = re.match(r"s_cbranch_\w+\s+(\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append(lines[labels[m.group(1)]:i + 1])
        rot = [lp for lp in loops if any(o.startswith("v_mov_b32_dpp") for o in lp)]
        # the rotation loops themselves: the loops that rotate and hold no other loop that does
        return sorted((lp for lp in rot if not any(o is not lp and len(o) < len(lp) and o[0] in lp for o in rot)), key=len)

    seen = 0
    # (mangled prefix, residents' packed groups NG or residents IPL, travelers J, f64)
    for pat, ng, j, f64 in ((r"_ZN2nb13nb_force_symwILi2ELi1EEE", 2, 1, False), (r"_ZN2nb13nb_force_symwILi4ELi1EEE", 4, 1, False), (r"_ZN2nb13nb_force_symwILi4ELi2EEE", 4, 2, False),
                            (r"_ZN2nb13nb_force_symwILi8ELi1EEE", 8, 1, False), (r"_ZN2nb12nb_force_symILi4ELi4ELi2EEE", 4, 2, False),
                            (r"_ZN2nb15nb_force_symw64ILi8EEE", 8, 1, True)):
        m = re.search(r"^(%s\w*):.*?$(.*?)^\.Lfunc_end" % pat, text, re.S | re.M)      # to the end of the function: an s_endpgm may sit mid-body
        assert m, pat
        loops = innermost_loops(m.group(2))
        # the wave-granular kernels have two forms of the loop: a sweep over an OWN chunk keeps no traveler sums (12 packed + 2 v_rsq
        # per group, 4 rotations; f64: 16 + v_rsq_f64 per resident, 8 rotations); the workgroup form has the one
        # ... and each of them twice: once for the wave's own range, once for the pieces it draws from the queue afterwards
        two_forms = "symw" in pat
        assert len(loops) == (4 if two_forms else 1), (pat, len(loops))
        for both, lp in zip((False, False, True, True) if two_forms else (True,), loops):
            ops = [l.split()[0] for l in lp if not l.endswith(":")]
            valu = [o for o in ops if o.startswith("v_")]
            assert not any(o.startswith("scratch_") or o.startswith("ds_") or o.startswith("global_") or o.startswith("buffer_") for o in ops), pat
            per_step = ((14 if both else 8) if f64 else (10 if both else 4)) * j
            u = ops.count("v_mov_b32_dpp") // per_step            # rotation steps per loop iteration (hipcc unrolls the short 4-resident body by 2)
            assert u >= 1 and ops.count("v_mov_b32_dpp") == per_step * u, (pat, both, ops.count("v_mov_b32_dpp"))
            if f64:
                assert ops.count("v_rsq_f64_e32") == ng * u and len(valu) == (ng * (20 if both else 16) + per_step) * u, (pat, both, len(valu))
            else:
                pk = 16 if both else 12
                assert ops.count("v_rsq_f32_e64") == 2 * ng * j * u and sum(o.startswith("v_pk_") for o in valu) == pk * ng * j * u, (pat, both)
                assert len(valu) == ((pk + 2) * ng + per_step // j) * j * u, (pat, both, len(valu))
        seen += 1
    assert seen == 6


def test_no_64_bit_instruction_of_a_loop_straddles_an_8_byte_boundary():
    """gfx950 issues a 64-bit encoded instruction (packed f32, VOP3, DPP) that does not start on an 8-byte boundary more slowly:
    the symmetric pass's loop ran 12 % longer at one wave per SIMD when its head sat at 4 mod 8, config 2's LDS-tile kernel lost
    1.9 % between two builds of the same 476-instruction loop (profiles/r04/README.md).  The build aligns them (csrc/align_loops.py
    between hipcc's code generation and the assembler; the fp64 kernels, and any function that would need a no-op per 12 loop instructions, are left alone); this test disassembles the BUILT library -- what the GPU box loads -- and wants not one misaligned
    64-bit instruction in any loop of the packed-f32 force kernels."""
    import sys
    lib = os.path.join(CSRC, "libnbody3d_hip.so")
    if not os.path.exists(lib) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built library and llvm-objdump")
    sys.path.insert(0, os.path.normpath(os.path.join(CSRC, "..", "..", "tools")))
    import loop_parity
    rows = loop_parity.loops(loop_parity.device_disassembly(lib), 16)
    aligned = ("nb_force_symwILi", "nb_force_symw_rankILi", "nb_force_symILi4", "nb_force_pk_sgprILi", "nb_force_pkILi4", "nb_step_jpkILi", "nb_step_fusedILi4")
    hot = [r for r in rows if any(k in r[0] for k in aligned)]
    assert len(hot) > 100, len(hot)
    assert sum(r[3] for r in hot) > 10000                                   # 64-bit instructions looked at
    bad = [(r[0][:60], hex(r[1]), r[4], r[3]) for r in hot if r[4]]
    assert not bad, bad[:5]


def test_align_loops_never_splits_a_pc_relative_address_sequence(tmp_path):
    """csrc/align_loops.py puts an s_nop in front of a misaligned 64-bit instruction of a loop.  The addends of an
    `s_getpc_b64` / `s_add_u32 ..., sym@rel32@lo+4` / `s_addc_u32 ..., sym@rel32@hi+12` sequence are distances from the
    s_getpc_b64: a no-op inside the sequence would move the computed address with no diagnostic.  Today's device code has no such
    sequence inside a loop; this feeds the tool a loop that does and wants the three instructions left adjacent -- and the
    misaligned 64-bit instruction AFTER the sequence still aligned."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        pytest.skip("needs the ROCm assembler")
    src = tmp_path / "in.s"
    src.write_text("""\t.text
\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.globl\tf
\t.p2align\t8
\t.type\tf,@function
f:
\ts_mov_b32 s0, 0
.LBB0_1:
\ts_add_u32 s0, s0, 1
\ts_getpc_b64 s[4:5]
\ts_add_u32 s4, s4, tbl@rel32@lo+4
\ts_addc_u32 s5, s5, tbl@rel32@hi+12
\tv_pk_add_f32 v[0:1], v[0:1], v[2:3]
""" + "\ts_add_u32 s1, s1, 1\n" * 12 + """\ts_cmp_lt_u32 s0, 16
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
\t.size\tf, .Lfunc_end0-f
\t.type\ttbl,@object
\t.data
tbl:
\t.long 0
""")
    dst = tmp_path / "out.s"
    subprocess.check_call(["python3", os.path.join(CSRC, "align_loops.py"), str(src), str(dst)])
    ops = [l.split(";")[0].split()[0] for l in dst.read_text().splitlines() if l.startswith("\t") and not l.strip().startswith(".")]
    i = ops.index("s_getpc_b64")
    assert ops[i:i + 3] == ["s_getpc_b64", "s_add_u32", "s_addc_u32"], ops
    # s_mov (4) | head: s_add (4) s_getpc (4) s_add+literal (8) s_addc+literal (8) -> the packed add would start at 4 mod 8: one no-op, after the sequence
    assert ops[i + 3] == "s_nop" and ops[i + 4] == "v_pk_add_f32" and ops.count("s_nop") == 1, ops
