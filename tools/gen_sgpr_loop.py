#!/usr/bin/env python3
"""gen_sgpr_loop.py -- writes nbody-demo-2023_amd/csrc/nbx_sgpr_loop.inc: the hand-scheduled gfx950 inner loop of
the packed-fp32 SGPR force kernel (nbx_kernels.hpp, JSRC_SGPR + MATH_PACKED, B = 2 or 4 bodies per lane).

Why asm (measured with tools/ubench2.x on MI355X, profiles/r02_ubench2_*.txt):
  * ONE wave per SIMD is what the reference summation order leaves a rank that owns 131072 of 1M bodies.  A lone
    wave issues one instruction per 4 cycles whatever it is: v_pk_*_f32 4, v_rsq_f32 8, and every SALU instruction
    (s_mov, s_add, s_waitcnt, ...) a full 4-cycle slot too; dependent VALU instructions do not stall.
  * hipcc's loop spends ~1.7 s_mov per j record on odd-element splats plus 9 address instructions per 16 records:
    1192 cycles per 16 records against 1024 of VALU work.
  * An 8-byte VOP3P instruction at an address = 4 mod 8 costs ~0.6 cycle extra on average (70 per 112): 4-byte
    instructions must come in pairs.
The loop below keeps the arithmetic of pair2() bit for bit (same operations, same order of summation), reads the odd
record elements through op_sel (no s_mov), addresses the scalar loads with immediate offsets from one pointer that
advances once per trip, pairs every 4-byte instruction with another one, and keeps every producer at least two
instructions away from its consumer (gfx940-family VALU hazards need one wait state after a transcendental or packed
result; nothing is inserted inside an asm statement, so the order below IS the guarantee).

Register plan (explicit, all clobbered):  temporaries v[T:T+23] = two slots x {dx dy dz r2 q s};  j ring
s[36:67] (group A: 8 records) and s[68:99] (group B);  s[30:31] = pointer (biased by -TRIP bytes so every immediate
offset is positive), s[28:29] = end value of that pointer, s[34] = softening.

Time-sliced variants (sgpr_loop_asm_b*_ts, LOOP_ASM_TS): waves that share a SIMD issue in strict age order on this chip
(tools/wave_fair.hip: of two resident waves the older one leaves the loop at 0.50 of the kernel time, the younger runs
alone afterwards with nobody to fill its issue bubbles).  Once per trip the wave reads the 100 MHz clock
(s_memrealtime, requested with the trip's first loads, landed by the wait that is there anyway) and takes priority 3
when (clock & slice_bit) equals the parity of its wave slot, priority 0 otherwise: each of two waves is the favoured one
half of the time, both stay resident to the end and fill each other's bubbles.  Six more scalar instructions per trip;
the arithmetic is untouched (same bits).  s[26:27] = clock.

Two-j-records-per-operation variant (sgpr_loop_asm_jpair, round 4): ONE body per lane, for slices that leave less than
one wave per SIMD with two bodies per lane (<= 65536 owned bodies: the reference summation order gives one chain per
owned body and nothing else to parallelise over).  The plain B = 1 kernel issues 12 unpacked VALU + 1 v_rsq_f32 per
pair = 56 cycles; here the packed lanes hold two CONSECUTIVE j records of the same body instead of two bodies: 9 packed
instructions + 2 v_rsq_f32 produce (dx, dy, dz, s) of records j and j + 1, and six plain v_fmac_f32 add the two terms to
the single accumulator in ascending j -- 76 cycles per two pairs, the operations and the order of pair<float>() exactly
(same bits as every other reference-order shape).  A packed operand is one aligned 64-bit register pair, so x_j and
x_{j+1} must be neighbours: the loop reads a pair-interleaved copy of the record array, {x0 x1 y0 y1 | z0 z1 w0 w1} per
two records (pair_transpose_kernel rebuilds it once per step: 32 B per record of L2 traffic against 76 n cycles of
arithmetic).  Four records (two such blocks) are interleaved instruction by instruction.  4-byte VALU encodings
(v_rsq_f32_e32, v_fmac_f32_e32) come in even runs; where a lone s_waitcnt follows, the first v_fmac of the run is
written in its 8-byte VOP3 form (v_fma_f32) instead of spending an s_nop on the alignment.
"""
import os
import sys

TBASE = 40            # first temporary VGPR (even)
RING_A, RING_B = 36, 68
SP, SE, SEPS = 30, 28, 34
NEG = "neg_lo:[0,1] neg_hi:[0,1]"
SGPR_PREFETCH = int(os.environ.get("NBX_SGPR_PREFETCH", "1"))  # B = 2 / 4 loops, _pf variants: trips ahead of the L2 prefetch
# ring groups (8 records each) per trip of the jpair loop.  It only ever runs with one wave per SIMD at most, where the taken branch, the
# pointer update and the prefetch are paid at full price: 64 / 128 / 256 records per trip measured 46.97 / 48.25 / 48.93 % at 65536 of 1M bodies
# (profiles/r04_jpair_trip_ab.txt); 256 records = the j tile every array is padded to, 14 KB of loop.
JPAIR_GROUPS = int(os.environ.get("NBX_JPAIR_GROUPS", "32"))


def tmp(slot, k):
    b = TBASE + 12 * slot + 2 * k
    return "v[%d:%d]" % (b, b + 1), "v%d" % b, "v%d" % (b + 1)


def record_ops(slot, rbase, xi, yi, zi, ax, ay, az):
    """The 14 instructions of pair2() for one packed pair of i-bodies and the record in s[rbase:rbase+3]."""
    xy, zw = "s[%d:%d]" % (rbase, rbase + 1), "s[%d:%d]" % (rbase + 2, rbase + 3)
    (dx, _, _), (dy, _, _), (dz, _, _), (r2, r2l, r2h), (q, _, _), (s, _, _) = (tmp(slot, k) for k in range(6))
    head = [
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[0,1] %s" % (dx, xy, xi, NEG),
        "v_pk_add_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[1,1] %s" % (dy, xy, yi, NEG),
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[0,1] %s" % (dz, zw, zi, NEG),
        "v_pk_fma_f32 %s, %s, %s, s[%d:%d] op_sel_hi:[1,1,0]" % (r2, dz, dz, SEPS, SEPS + 1),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dy, dy, r2),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dx, dx, r2),
        "v_rsq_f32_e32 %s, %s" % (r2l, r2l),
        "v_rsq_f32_e32 %s, %s" % (r2h, r2h),
        "v_pk_mul_f32 %s, %s, %s" % (q, r2, r2),
        "v_pk_mul_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[1,1]" % (s, zw, r2),
        "v_pk_mul_f32 %s, %s, %s" % (s, s, q),
    ]
    tail = [
        "v_pk_fma_f32 %s, %s, %s, %s" % (ax, dx, s, ax),
        "v_pk_fma_f32 %s, %s, %s, %s" % (ay, dy, s, ay),
        "v_pk_fma_f32 %s, %s, %s, %s" % (az, dz, s, az),
    ]
    return head, tail


def zip2(a, b):
    out = []
    for x, y in zip(a, b):
        out += [x, y]
    return out


def group_ops(B, ring):
    """VALU instructions that apply the 8 records of one ring group, in ascending record order per accumulator."""
    out = []
    if B == 2:
        I = ("%3", "%4", "%5", "%0", "%1", "%2")
        for u in range(0, 8, 2):  # two records in flight; the accumulator updates of record u precede those of u+1
            ha, ta = record_ops(0, ring + 4 * u, *I)
            hb, tb = record_ops(1, ring + 4 * (u + 1), *I)
            out += zip2(ha, hb) + ta + tb
    else:
        I0 = ("%6", "%7", "%8", "%0", "%1", "%2")
        I1 = ("%9", "%10", "%11", "%3", "%4", "%5")
        for u in range(8):        # one record, the two packed pairs in flight
            ha, ta = record_ops(0, ring + 4 * u, *I0)
            hb, tb = record_ops(1, ring + 4 * u, *I1)
            out += zip2(ha, hb) + zip2(ta, tb)
    return out


def loads(ring, off):
    return ["s_load_dwordx16 s[%d:%d], s[%d:%d], 0x%x" % (ring, ring + 15, SP, SP + 1, off),
            "s_load_dwordx16 s[%d:%d], s[%d:%d], 0x%x" % (ring + 16, ring + 31, SP, SP + 1, off + 64)]


WAIT = "s_waitcnt lgkmcnt(0)"


STIME = 26


def loop_text(B, groups_per_trip, ts=False, pf=False):
    """groups_per_trip even; a trip covers 8*groups_per_trip records = trip_bytes of the record array.
    ts: once per trip, wave priority from the clock (operands %MASK = slice bit, %PAR = that bit if the wave's slot is odd)."""
    assert groups_per_trip % 2 == 0 and groups_per_trip >= 2
    trip = 128 * groups_per_trip
    pro = ["s_mov_b64 s[%d:%d], %%%d" % (SP, SP + 1, 6 if B == 2 else 12),
           "s_mov_b64 s[%d:%d], %%%d" % (SE, SE + 1, 7 if B == 2 else 13),
           "s_mov_b32 s%d, 0x3a83126f" % SEPS,          # softeningSquared = 1e-3f (ver7/GSimulation.cpp:126)
           "s_mov_b32 s%d, 0x%x" % (SEPS + 1, trip),     # pointer increment, kept in a register: s_add stays 4 bytes
           "s_nop 4"]
    pro += loads(RING_A, trip) + [WAIT]                   # group A of the first trip (pointer is biased by -trip)
    body = []
    # group 0 (ring A) is resident at the loop head; group g+1 is requested before group g is consumed
    body += loads(RING_B, trip + 128)
    if pf:
        # L2 prefetch (see the jpair docstring paragraph): one vector load per trip, lane l touching byte 64 (l mod 16) of the trip
        # SGPR_PREFETCH trips ahead; never waited on inside the loop, destination never read
        body += ["global_load_dword v%d, %%%d, s[%d:%d]" % (TBASE + 24, (8 if B == 2 else 14) + (2 if ts else 0), SP, SP + 1)]
    if ts:
        body += ["s_memrealtime s[%d:%d]" % (STIME, STIME + 1)]
    body += group_ops(B, RING_A)
    # 4-byte instructions in even numbers keep the 8-byte alignment of what follows; no scalar load is in flight while the
    # pointer changes, and SCC (set by the last compare) is not written again before the branch
    cluster = [WAIT]
    if ts:
        # favoured (SCC = 1): priority 3; otherwise the branch skips the second s_setprio
        n0 = 8 if B == 2 else 14
        cluster += ["s_and_b32 s%d, s%d, %%%d" % (STIME, STIME, n0), "s_cmp_eq_u32 s%d, %%%d" % (STIME, n0 + 1),
                    "s_setprio 0", "s_cbranch_scc0 2f", "s_setprio 3", "2:", "s_nop 0"]
    cluster += ["s_add_u32 s%d, s%d, s%d" % (SP, SP, SEPS + 1), "s_addc_u32 s%d, s%d, 0" % (SP + 1, SP + 1),
                "s_cmp_lg_u64 s[%d:%d], s[%d:%d]" % (SP, SP + 1, SE, SE + 1)]
    body += cluster
    # from here on the pointer has advanced by one trip: offsets are relative to the NEW value
    rings = (RING_A, RING_B)
    for g in range(1, groups_per_trip):
        nxt = g + 1
        body += loads(rings[nxt % 2], trip if nxt == groups_per_trip else 128 * nxt)  # next trip's group 0 is over-read on the last trip
        body += group_ops(B, rings[g % 2])
        body += [WAIT, "s_cbranch_scc1 1b" if nxt == groups_per_trip else "s_nop 0"]
    return pro, body, trip


def check_alignment(body):
    """Every 8-byte instruction must start at a multiple of 8 bytes from the loop head."""
    off = 0
    for ins in body:
        op = ins.split()[0]
        if op.endswith(":"):
            continue
        size = 4 if (op in ("s_waitcnt", "s_add_u32", "s_addc_u32", "s_cmp_lg_u64", "s_cbranch_scc1", "s_nop", "s_and_b32", "s_cmp_eq_u32",
                            "s_setprio", "s_cbranch_scc0") or op.endswith("_e32")) else 8
        assert size == 4 or off % 8 == 0, (ins, off)
        off += size
    return off


def check_distance(body):
    """No VALU instruction reads a VGPR written by the instruction just before it (one wait state for trans / packed results)."""
    import re
    prev_defs = set()
    for ins in body:
        if not ins.startswith("v_"):
            prev_defs = set()
            continue
        ops = ins.split(None, 1)[1]
        regs = []
        for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)|%(\d+)", ops):
            if m.group(1):
                regs.append(frozenset("v%d" % r for r in range(int(m.group(1)), int(m.group(2)) + 1)))
            elif m.group(3):
                regs.append(frozenset(["v%s" % m.group(3)]))
            else:
                regs.append(frozenset(["%" + m.group(4)]))
        dst, srcs = regs[0], regs[1:]
        for s in srcs:
            assert not (s & prev_defs), "back-to-back dependency: " + ins
        prev_defs = set(dst)


def jpair_block_ops(slot, rbase, xy, zz, ax, ay, az, wide_first):
    """One pair-interleaved block s[rbase:rbase+7] = {x0 x1 y0 y1 z0 z1 w0 w1} applied to ONE body (xy = {xi, yi}, zz = {zi, -}):
    the operations of pair<float>() for record 2k in the low halves and record 2k+1 in the high halves, then the six
    accumulator updates in ascending record order.  wide_first: first v_fmac in its 8-byte form (alignment, see docstring)."""
    sx, sy, sz, sw = ("s[%d:%d]" % (rbase + 2 * k, rbase + 2 * k + 1) for k in range(4))
    (dx, dxl, dxh), (dy, dyl, dyh), (dz, dzl, dzh), (r2, r2l, r2h), (q, _, _), (s, sl, sh) = (tmp(slot, k) for k in range(6))
    head = [
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[1,0] %s" % (dx, sx, xy, NEG),                 # {x0 - xi, x1 - xi}
        "v_pk_add_f32 %s, %s, %s op_sel:[0,1] op_sel_hi:[1,1] %s" % (dy, sy, xy, NEG),    # {y0 - yi, y1 - yi}
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[1,0] %s" % (dz, sz, zz, NEG),
        "v_pk_fma_f32 %s, %s, %s, s[%d:%d] op_sel_hi:[1,1,0]" % (r2, dz, dz, SEPS, SEPS + 1),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dy, dy, r2),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dx, dx, r2),
        "v_rsq_f32_e32 %s, %s" % (r2l, r2l),
        "v_rsq_f32_e32 %s, %s" % (r2h, r2h),
        "v_pk_mul_f32 %s, %s, %s" % (q, r2, r2),
        "v_pk_mul_f32 %s, %s, %s" % (s, sw, r2),                                          # {Gm0 inv0, Gm1 inv1}
        "v_pk_mul_f32 %s, %s, %s" % (s, s, q),
    ]
    tail = [
        ("v_fma_f32 %s, %s, %s, %s" % (ax, dxl, sl, ax)) if wide_first else ("v_fmac_f32_e32 %s, %s, %s" % (ax, dxl, sl)),
        "v_fmac_f32_e32 %s, %s, %s" % (ay, dyl, sl),
        "v_fmac_f32_e32 %s, %s, %s" % (az, dzl, sl),
        "v_fmac_f32_e32 %s, %s, %s" % (ax, dxh, sh),
        "v_fmac_f32_e32 %s, %s, %s" % (ay, dyh, sh),
        "v_fmac_f32_e32 %s, %s, %s" % (az, dzh, sh),
    ]
    return head, tail


def jpair_group_ops(ring, odd_tail):
    """The 8 records (4 blocks) of one ring group on one body; odd_tail: make the group's 4-byte count odd (a lone s_waitcnt follows)."""
    out = []
    I = ("%3", "%4", "%0", "%1", "%2")
    for u in range(0, 4, 2):  # two blocks in flight; block u's six updates precede block u+1's
        ha, ta = jpair_block_ops(0, ring + 8 * u, *I, wide_first=(odd_tail and u == 2))
        hb, tb = jpair_block_ops(1, ring + 8 * (u + 1), *I, wide_first=False)
        out += zip2(ha, hb) + ta + tb
    return out


# Trips of the _pf variants relative to the plain loops.  They only run with ONE wave per SIMD, where the taken branch and the pointer update
# are paid at full price: 64 / 128 / 256 records per trip measured 54.3 / 55.0 / 55.2 % for a rank of eight at n = 1M (+1.7 %), +0.8 and +1.3 % at
# 131072 of 262144 / 524288 (profiles/r04_pf_trip_ab.txt; with two or four waves per SIMD longer trips LOSE 1-2 %, round 2 -- those shapes keep
# the plain and the time-sliced loops).  Four bodies per lane with the prefetch is never what AUTO takes (two per lane fill the same CUs
# with twice the workgroups): its instance keeps the short trip rather than another 11 000 lines of generated text.
PF_TRIP_FACTOR = int(os.environ.get("NBX_PF_TRIP_FACTOR", "4"))
TS_TRIP_FACTOR = int(os.environ.get("NBX_TS_TRIP_FACTOR", "1"))  # experiment knob: trips of the time-sliced variants (two waves per SIMD)
PF_TRIP_FACTOR_B4 = int(os.environ.get("NBX_PF_TRIP_FACTOR_B4", "1"))
JPAIR_PREFETCH = int(os.environ.get("NBX_JPAIR_PREFETCH", "1"))  # trips ahead of the L2 prefetch (0 = none); 2 KB ... 8 KB ahead measured alike
VPF = TBASE + 24                                                 # destination of the prefetch load (never read)


def jpair_loop_text(groups_per_trip):
    assert groups_per_trip % 2 == 0 and groups_per_trip >= 2
    trip = 128 * groups_per_trip
    pro = ["s_mov_b64 s[%d:%d], %%5" % (SP, SP + 1),
           "s_mov_b64 s[%d:%d], %%6" % (SE, SE + 1),
           "s_mov_b32 s%d, 0x3a83126f" % SEPS,          # softeningSquared = 1e-3f (ver7/GSimulation.cpp:126)
           "s_mov_b32 s%d, 0x%x" % (SEPS + 1, trip),
           "s_nop 4"]
    pro += loads(RING_A, trip) + [WAIT]
    body = []
    body += loads(RING_B, trip + 128)
    if JPAIR_PREFETCH:
        # L2 prefetch: a lone wave has only the 4 x 76 cycles of one ring group to cover a scalar load, and every wave of an XCD
        # asks for the same line at about the same time, so what it waits for is the first requester's Infinity-Cache round trip
        # (~545 cycles).  One vector load per trip, lane l touching byte 64 (l mod 16) of the trip JPAIR_PREFETCH trips ahead,
        # pulls those 16 lines into the XCD's L2 (vmcnt is never waited on inside the loop; the destination is never read).
        # (the distance travels in the VGPR operand -- pf_off = 64 (l mod 16) + kSgprJpairPrefetchBytes -- so that it is not bound by
        # the 13-bit immediate)
        assert (JPAIR_PREFETCH + 1) * (trip // 16) <= 512, "reads stay inside the spare records"
        body += ["global_load_dword v%d, %%7, s[%d:%d]" % (VPF, SP, SP + 1)]
    body += jpair_group_ops(RING_A, odd_tail=False)
    body += [WAIT, "s_add_u32 s%d, s%d, s%d" % (SP, SP, SEPS + 1), "s_addc_u32 s%d, s%d, 0" % (SP + 1, SP + 1),
             "s_cmp_lg_u64 s[%d:%d], s[%d:%d]" % (SP, SP + 1, SE, SE + 1)]
    rings = (RING_A, RING_B)
    for g in range(1, groups_per_trip):
        nxt = g + 1
        last = nxt == groups_per_trip
        body += loads(rings[nxt % 2], trip if last else 128 * nxt)
        body += jpair_group_ops(rings[g % 2], odd_tail=not last)
        body += [WAIT, "s_cbranch_scc1 1b"] if last else [WAIT]
    return pro, body, trip


def emit_jpair(groups_per_trip):
    pro, body, trip = jpair_loop_text(groups_per_trip)
    nbytes = check_alignment(body)
    check_distance(body)
    post = ["s_waitcnt vmcnt(0)"] if JPAIR_PREFETCH else []  # the prefetch destination is ours until the last one has landed
    lines = (['      "%s\\n"' % s for s in pro] + ['      ".p2align 3\\n"', '      "1:\\n"'] + ['      "%s\\n"' % s for s in body] +
             ['      "%s\\n"' % s for s in post])
    clob = (['"v%d"' % r for r in range(TBASE, TBASE + 24 + (1 if JPAIR_PREFETCH else 0))] + ['"s%d"' % r for r in range(28, 100) if r not in (32, 33)] +
            ['"scc"', '"memory"'])
    nv = sum(1 for s in body if s.startswith("v_"))
    ns = sum(1 for s in body if not s.startswith("v_") and not s.endswith(":"))
    txt = []
    txt.append("// ONE body per lane, two consecutive j records per packed operation: %d records per trip, %d VALU + %d scalar%s instructions, %d bytes of loop body." %
               (trip // 16, nv, ns - (1 if JPAIR_PREFETCH else 0), " + 1 vector-memory" if JPAIR_PREFETCH else "", nbytes))
    txt.append("// `first` / `last` delimit the j range in the PAIR-INTERLEAVED copy of the record array ({x0 x1 y0 y1 | z0 z1 w0 w1} per two records,")
    txt.append("// pair_transpose_kernel): same byte offsets as in the record array, a positive multiple of kSgprAsmTrip<1> records.  xy = {xi, yi}, zz = {zi, -}.")
    txt.append("template <> constexpr int kSgprAsmTrip<1> = %d;" % (trip // 16))
    txt.append("// pf_off = 64 * (lane mod kSgprJpairPrefetchLines) + kSgprJpairPrefetchBytes: byte offset (from the biased pointer) of the line this lane prefetches into L2, %d trips ahead" % JPAIR_PREFETCH)
    txt.append("// (reads up to %d records past `last`: spare)." % ((JPAIR_PREFETCH + 1) * (trip // 16) if JPAIR_PREFETCH else 8))
    txt.append("constexpr unsigned kSgprJpairPrefetchBytes = %d;" % (trip + JPAIR_PREFETCH * trip))
    assert trip // 64 in (8, 16, 32, 64)
    txt.append("constexpr unsigned kSgprJpairPrefetchLines = %d;  // 64-byte lines per trip: lane l touches line l mod this" % (trip // 64))
    txt.append("__device__ __forceinline__ void sgpr_loop_asm_jpair(const float4* first, const float4* last, f32x2 xy, f32x2 zz, unsigned pf_off, float& ax, float& ay, float& az) {")
    txt.append("  const char* q = reinterpret_cast<const char*>(first) - %d;     // biased: all immediate offsets positive" % trip)
    txt.append("  const char* qend = reinterpret_cast<const char*>(last) - %d;  // value of the pointer after the last trip's advance" % trip)
    txt.append("  asm volatile(")
    txt.append("\n".join(lines))
    txt.append('      : "+v"(ax), "+v"(ay), "+v"(az)')
    txt.append('      : "v"(xy), "v"(zz), "s"(q), "s"(qend), "v"(pf_off)')
    txt.append("      : %s);" % ", ".join(clob))
    txt.append("}")
    return "\n".join(txt)


def emit(B, groups_per_trip, ts=False, pf=False):
    assert not (ts and pf)
    pro, body, trip = loop_text(B, groups_per_trip, ts, pf)
    nbytes = check_alignment(body)
    check_distance(body)
    lines = ['      "%s\\n"' % s for s in pro] + ['      ".p2align 3\\n"', '      "1:\\n"'] + ['      "%s\\n"' % s for s in body]  # labels ("2:") included
    if pf:
        lines += ['      "s_waitcnt vmcnt(0)\\n"']  # the prefetch destination is ours until the last one has landed
    clob = (['"v%d"' % r for r in range(TBASE, TBASE + 24 + (1 if pf else 0))] + ['"s%d"' % r for r in range(STIME if ts else 28, 100) if r not in (32, 33)] +
            ['"scc"', '"memory"'])
    nv = sum(1 for s in body if s.startswith("v_"))
    ns = sum(1 for s in body if not s.startswith("v_") and not s.endswith(":"))
    if B == 2:
        sig = "f32x2 xi, f32x2 yi, f32x2 zi, f32x2& ax, f32x2& ay, f32x2& az"
        outs = '"+v"(ax), "+v"(ay), "+v"(az)'
        ins = '"v"(xi), "v"(yi), "v"(zi), "s"(q), "s"(qend)'
        extra = ', "s"(slice_bit), "s"(slot_bit)'
    else:
        sig = ("f32x2 xi0, f32x2 yi0, f32x2 zi0, f32x2 xi1, f32x2 yi1, f32x2 zi1, f32x2& ax0, f32x2& ay0, f32x2& az0, "
               "f32x2& ax1, f32x2& ay1, f32x2& az1")
        outs = '"+v"(ax0), "+v"(ay0), "+v"(az0), "+v"(ax1), "+v"(ay1), "+v"(az1)'
        ins = '"v"(xi0), "v"(yi0), "v"(zi0), "v"(xi1), "v"(yi1), "v"(zi1), "s"(q), "s"(qend)'
        extra = ', "s"(slice_bit), "s"(slot_bit)'
    txt = []
    txt.append("// B = %d bodies per lane%s: %d records per trip, %d VALU + %d scalar%s instructions, %d bytes of loop body." %
               (B, ", time-sliced wave priority" if ts else (", L2 prefetch %d trips ahead" % SGPR_PREFETCH if pf else ""), trip // 16, nv, ns - (1 if pf else 0),
                " + 1 vector-memory" if pf else "", nbytes))
    txt.append("// `first` points at the first record of the j range, `last` one past it; the range is a positive multiple of")
    txt.append("// kSgprAsmTrip<%d> records.  The final trip requests 8 records past `last` (never used; kSgprOverread spare)." % B)
    if ts:
        txt.append("// slice_bit = the clock bit (s_memrealtime, 10 ns units) that says whose turn it is; slot_bit = slice_bit when the wave's")
        txt.append("// slot on its SIMD is odd, else 0.  The wave leaves the loop at whatever priority it had last.")
        txt.append("__device__ __forceinline__ void sgpr_loop_asm_b%d_ts(const float4* first, const float4* last, unsigned slice_bit, unsigned slot_bit, %s) {" % (B, sig))
    elif pf:
        txt.append("// For launches that leave ONE wave per SIMD (e.g. a rank that owns 131072 of 1M bodies): the arithmetic of one ring group is all the")
        txt.append("// cover a scalar load gets, and what it waits for is an Infinity-Cache round trip; with the lines already in L2 the loop is +3.5 %.")
        txt.append("// With two or more waves per SIMD the other waves are the cover and the extra instruction costs 0.4-1.4 % (profiles/r04_b2_prefetch_ab.txt).")
        txt.append("__device__ __forceinline__ void sgpr_loop_asm_b%d_pf(const float4* first, const float4* last, %s) {" % (B, sig))
        assert trip // 64 in (8, 16, 32, 64) and (SGPR_PREFETCH + 1) * (trip // 16) <= 512
        txt.append("  // the line this lane touches, %d trips ahead (distance in the register: not bound by the 13-bit immediate); reads up to %d records past `last` (spare)" %
                   (SGPR_PREFETCH, (SGPR_PREFETCH + 1) * (trip // 16)))
        txt.append("  const unsigned pf_off = (threadIdx.x & %du) * 64u + %du;" % (trip // 64 - 1, trip + SGPR_PREFETCH * trip))
    else:
        txt.append("template <> constexpr int kSgprAsmTrip<%d> = %d;" % (B, trip // 16))
        txt.append("__device__ __forceinline__ void sgpr_loop_asm_b%d(const float4* first, const float4* last, %s) {" % (B, sig))
    txt.append("  const char* q = reinterpret_cast<const char*>(first) - %d;     // biased: all immediate offsets positive" % trip)
    txt.append("  const char* qend = reinterpret_cast<const char*>(last) - %d;  // value of the pointer after the last trip's advance" % trip)
    txt.append("  asm volatile(")
    txt.append("\n".join(lines))
    txt.append("      : %s" % outs)
    txt.append("      : %s%s%s" % (ins, extra if ts else "", ', "v"(pf_off)' if pf else ""))
    txt.append("      : %s);" % ", ".join(clob))
    txt.append("}")
    return "\n".join(txt)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "nbody-demo-2023_amd", "csrc", "nbx_sgpr_loop.inc")
    parts = ["// nbx_sgpr_loop.inc -- GENERATED by tools/gen_sgpr_loop.py (see its docstring for the why); do not edit.",
             "// Included by nbx_kernels.hpp inside namespace nbx.  tests/test_isa_audit.py checks it is in sync with the generator.",
             "template <int B> constexpr int kSgprAsmTrip = 0;",
             emit(2, 8), emit(4, 4), emit(2, 8 * TS_TRIP_FACTOR, ts=True), emit(4, 4 * TS_TRIP_FACTOR, ts=True), emit(2, 8 * PF_TRIP_FACTOR, pf=True), emit(4, 4 * PF_TRIP_FACTOR_B4, pf=True), emit_jpair(JPAIR_GROUPS), ""]
    open(out, "w").write("\n".join(parts))


if __name__ == "__main__":
    main()
