"""CPU-side audit of the compiled gfx950 code (hipcc cross-compiles without a GPU).

The SGPR j-source issues its scalar loads from asm statements, which hipcc neither counts nor waits
for (cdna_hip_programming.md section 5.7).  A compiler-inserted copy or a reuse of a batch's SGPRs
between the s_load and its s_waitcnt is silent corruption -- or a memory fault when the late data
lands in a register that meanwhile holds an address (this happened once: DESIGN.md section 3.1).
This test disassembles every force_kernel instance and proves that cannot happen.
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "nbody-demo-2023_amd", "csrc", "nbx_api.hip")


def _shipped_hipflags():
    """The flags libnbx.so is built with (top-level Makefile, HIPFLAGS): the audited ISA must be the executed ISA."""
    mk = open(os.path.join(ROOT, "Makefile")).read()
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)
    flags = re.search(r"^HIPFLAGS\s*=\s*(.+)$", mk, re.M).group(1).replace("$(ARCH)", arch).split()
    assert "--offload-arch=gfx950" in flags and "-O3" in flags, flags
    return flags


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "nbx_api.s"
    subprocess.check_call(["hipcc"] + _shipped_hipflags() + ["-S", "--cuda-device-only", SRC, "-o", str(out)])
    txt = open(out).read()
    ks = {}
    for m in re.finditer(r"\n(_ZN3nbx\w+):(.*?)\n\s+s_endpgm.*?\.amdhsa_kernel \1(.*?)\.end_amdhsa_kernel", txt, re.S):
        ks[m.group(1)] = (m.group(2), m.group(3))
    # epilogue blocks after the first s_endpgm belong to the same function: take up to .Lfunc_end instead
    for m in re.finditer(r"\n(_ZN3nbx\w+):(.*?)\.Lfunc_end", txt, re.S):
        if m.group(1) in ks:
            ks[m.group(1)] = (m.group(2), ks[m.group(1)][1])
    return ks


def _sregs(operand_text):
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", operand_text):
        regs.add(int(a))
    return regs


def test_every_force_kernel_instance_is_present(kernels):
    names = [k for k in kernels if "force_kernel" in k]
    assert len(names) >= 36, len(names)
    assert any("integrate_kernel" in k for k in kernels) and any("ke_reduce_kernel" in k for k in kernels)


def test_no_scratch_no_spills(kernels):
    for name, (body, desc) in kernels.items():
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc)
        assert m and int(m.group(1)) == 0, name
        # SGPR spills go to VGPR lanes (v_writelane / v_readlane): forbidden in the force kernels, where a spill placed
        # between an asm s_load and its wait would copy registers whose data has not landed yet
        if "force_kernel" in name:
            assert "v_writelane_b32" not in body and "v_readlane_b32" not in body, name


def test_asm_scalar_loads_are_never_touched_before_their_wait(kernels):
    checked = 0
    for name, (body, _) in kernels.items():
        if "force_kernel" not in name:
            continue
        pending = {}  # sgpr -> line of the asm load that will write it
        in_asm = False
        for ln, line in enumerate(body.split("\n")):
            t = line.strip()
            if t.startswith(";;#ASMSTART") or t.startswith("; ;#ASMSTART") or "#ASMSTART" in t:
                in_asm = True
                continue
            if "#ASMEND" in t:
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            t = t.split(";")[0].strip()
            if not t:
                continue
            if re.match(r"s_waitcnt\b.*lgkmcnt\(0\)", t) or t == "s_waitcnt lgkmcnt(0)":
                pending.clear()
                continue
            ops = t.split(None, 1)[1] if " " in t or "\t" in t else ""
            if in_asm and t.startswith("s_load_dwordx16"):
                dst = ops.split(",")[0]
                rest = ",".join(ops.split(",")[1:])
                assert not (_sregs(rest) & set(pending)), (name, ln, t)
                for r in _sregs(dst):
                    assert r not in pending, (name, ln, t)
                    pending[r] = ln
                checked += 1
                continue
            if re.match(r"s_(cbranch|branch|endpgm|barrier|nop|waitcnt|setprio|sleep)", t) or t.endswith(":"):
                continue
            touched = _sregs(ops)
            bad = touched & set(pending)
            assert not bad, "%s: line %d `%s` touches s%s while the scalar load issued at line %d is in flight" % (
                name, ln, t, sorted(bad), min(pending[r] for r in bad))
    assert checked >= 40, checked


def test_sgpr_source_really_uses_scalar_loads_and_lds_source_broadcast_reads(kernels):
    for name, (body, _) in kernels.items():
        m = re.search(r"force_kernelI[fd]Li\dELi([12])E", name)
        if not m:
            continue
        if m.group(1) == "2":
            assert body.count("s_load_dwordx16") >= 3, name      # prologue + two pipelined requests per trip
        else:
            assert body.count("ds_read_b128") + body.count("ds_read2_b64") >= 8, name  # unrolled broadcast reads of the tile
            assert "s_load_dwordx16" not in body, name


def test_hot_loop_uses_raw_rsq_and_packed_math(kernels):
    for name, (body, _) in kernels.items():
        if "force_kernelIf" not in name:
            continue
        assert "v_rsq_f32" in body and "v_div_scale" not in body and "v_sqrt_f32" not in body, name
        if re.search(r"force_kernelIfLi[248]ELi\dELb[01]ELi1ELi1E", name):  # MATH_PACKED instances
            assert "v_pk_fma_f32" in body and "v_pk_mul_f32" in body, name


def test_euler_update_is_not_fma_contracted(kernels):
    """The reference's build has no FMA; hipcc fuses `a*b + c` by default.  integrate_kernel (fp32 and fp64) must keep
    the multiply and the add separately rounded -- the NBX_KERNEL_EXACT bit-equality tests depend on it."""
    seen = 0
    for name, (body, _) in kernels.items():
        if "integrate_kernel" not in name:
            continue
        seen += 1
        assert not re.search(r"\bv_(pk_)?fma(c|ak|mk)?_f(32|64)\b", body), name
    assert seen == 2


def test_generated_asm_loop_is_in_sync_with_its_generator(tmp_path):
    """csrc/nbx_sgpr_loop.inc is committed (the build needs no python); it must be what tools/gen_sgpr_loop.py writes.
    The generator itself asserts the two structural rules of the stream: 8-byte instructions on 8-byte offsets, and no
    VALU instruction reading a register written by the instruction just before it."""
    out = tmp_path / "loop.inc"
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_sgpr_loop.py"), str(out)])
    assert open(out).read() == open(os.path.join(ROOT, "nbody-demo-2023_amd", "csrc", "nbx_sgpr_loop.inc")).read()


def test_generated_jlane_loop_is_in_sync_with_its_generator(tmp_path):
    out = tmp_path / "jlane.inc"
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_jlane_loop.py"), str(out)])
    assert open(out).read() == open(os.path.join(ROOT, "nbody-demo-2023_amd", "csrc", "nbx_jlane_loop.inc")).read()


def test_hand_scheduled_instances_have_no_scalar_moves_in_the_loop(kernels):
    seen = 0
    for name, (body, _) in kernels.items():
        if not re.search(r"force_kernelIfLi[24]ELi2ELi[01]ELi1ELi1ELb[01]ELi1E", name):
            continue
        seen += 1
        asm = body[body.index("#ASMSTART"):body.index("#ASMEND")]
        loop = asm[asm.index("1:"):]
        assert "s_mov_b32" not in loop and "s_mov_b64" not in loop, name
        assert loop.count("s_load_dwordx16") in (8, 16) and loop.count("v_rsq_f32") == 128, name  # 32 / 64 records per trip
    assert seen == 6, seen  # B in {2, 4} x {row, slab} without wave split, x {slab} with


def test_two_records_per_operation_instances(kernels):
    """One body per lane + LOOP_ASM = sgpr_loop_asm_jpair (row and slab epilogue): per 256 records of a trip 128 pair-interleaved blocks of
    9 packed instructions + 2 v_rsq_f32 + 6 accumulator updates (plain fused multiply-adds, j ascending), no scalar moves, no s_nop spent
    on alignment except none at all: the run of 4-byte v_fmac encodings in front of a lone s_waitcnt starts with the 8-byte v_fma_f32."""
    seen = 0
    for name, (body, _) in kernels.items():
        if not re.search(r"force_kernelIfLi1ELi2ELi[01]ELi1ELi0ELb0ELi1E", name):
            continue
        seen += 1
        asm = body[body.index("#ASMSTART"):body.index("#ASMEND")]
        loop = asm[asm.index("1:"):]
        assert "s_mov_b32" not in loop and "s_mov_b64" not in loop and "s_nop" not in loop, name
        assert loop.count("s_load_dwordx16") == 64 and loop.count("v_rsq_f32") == 256 and loop.count("global_load_dword") == 1, name
        assert len(re.findall(r"\bv_pk_(add|mul|fma)_f32\b", loop)) == 128 * 9, name
        acc = re.findall(r"\bv_fma(?:c_f32_e32|_f32) (v\d+),", loop)
        assert len(acc) == 128 * 6, name
        # three accumulators, updated x, y, z, x, y, z, ... : each is one sequential chain over ascending j
        assert len(set(acc)) == 3 and all(acc[k] == acc[k % 3] for k in range(len(acc))), name
        assert "pair_transpose_kernel" not in name
    assert seen == 2, seen
    assert any("pair_transpose_kernel" in k for k in kernels)


def test_l2_prefetch_instances_differ_from_the_plain_loop_by_one_vector_load_per_trip(kernels):
    """LOOP_ASM_PF (last template argument 3; row epilogue, no wave split, B = 2 and 4): the plain hand-scheduled loop plus ONE
    global_load_dword per trip whose destination no instruction reads, and a vmcnt(0) behind the loop before that register is reused.
    Four bodies per lane: the very same trip.  Two bodies per lane (what AUTO takes for one wave per SIMD): a trip four times as
    long -- 256 records -- whose VALU stream is the plain trip's four times over, instruction for instruction."""
    seen = 0
    for name, (body, _) in kernels.items():
        m = re.search(r"force_kernelIfLi([24])ELi2ELi1ELi1ELi1ELb0ELi3E", name)
        if not m:
            continue
        seen += 1
        plain = [b for k, (b, _) in kernels.items() if re.search(r"force_kernelIfLi%sELi2ELi1ELi1ELi1ELb0ELi1E" % m.group(1), k)]
        assert len(plain) == 1
        asm = body[body.index("#ASMSTART"):body.index("#ASMEND")]
        loop, ref = asm[asm.index("1:"):], plain[0][plain[0].index("#ASMSTART"):plain[0].index("#ASMEND")]
        ref = ref[ref.index("1:"):]
        pf = [l for l in loop.split("\n") if l.strip().startswith("global_load_dword")]
        assert len(pf) == 1 and re.match(r"\s*global_load_dword v64, v\d+, s\[30:31\]", pf[0]), (name, pf)
        assert not re.search(r"\bv64\b", loop.replace(pf[0], "")), name                     # prefetched data is never read
        strip = lambda t: [re.sub(r"\bv\d+\b|v\[\d+:\d+\]", "V", l.strip()) for l in t.split("\n") if l.strip() and not l.strip().startswith(";")]
        a, b = strip(loop.replace(pf[0], "")), strip(ref)
        assert a[-1].startswith("s_waitcnt vmcnt(0)"), name
        valu = lambda t: [l for l in t if l.startswith("v_")]
        if m.group(1) == "4":
            assert a[:-1] == b, name
        else:
            assert valu(a) == valu(b) * 4 and loop.count("s_load_dwordx16") == 64 and loop.count("s_cbranch_scc1 1b") == 1, name
            assert "s_mov_b32" not in loop and "s_mov_b64" not in loop, name
    assert seen == 2, seen


def test_time_sliced_instances_set_priority_once_per_trip(kernels):
    """LOOP_ASM_TS (the last template argument = 2): row epilogue, no wave split, B = 2 and 4.  The loop is the plain one plus one
    clock read, the two-instruction decision, the s_setprio pair around a forward branch -- and nothing else that is scalar."""
    seen = 0
    for name, (body, _) in kernels.items():
        if not re.search(r"force_kernelIfLi[24]ELi2ELi1ELi1ELi1ELb0ELi2E", name):
            continue
        seen += 1
        assert body.count("s_getreg_b32") == 1, name
        asms = [m.group(0) for m in re.finditer(r"#ASMSTART.*?#ASMEND", body, re.S)]
        loop = [a for a in asms if "s_memrealtime" in a]
        assert len(loop) == 1, name
        loop = loop[0][loop[0].index("1:"):]
        assert loop.count("s_memrealtime") == 1 and loop.count("s_setprio 0") == 1 and loop.count("s_setprio 3") == 1, name
        assert loop.count("s_cbranch_scc0 2f") == 1 and loop.count("s_cbranch_scc1 1b") == 1, name
        assert "s_mov_b32" not in loop and "s_mov_b64" not in loop, name
        assert loop.count("s_load_dwordx16") in (8, 16) and loop.count("v_rsq_f32") == 128, name
        # the wave returns to priority 0 behind the loop (the epilogue and the next kernel start level)
        assert "s_setprio 0" in body[body.rindex("#ASMEND"):], name
    assert seen == 2, seen
