#!/usr/bin/env python3
"""gen_jlane_loop.py -- writes nbody-demo-2023_amd/csrc/nbx_jlane_loop.inc: the hand-scheduled gfx950 main loop of
force_jlane_kernel (nbx_kernels.hpp; fp32, NB = 2, 4 or 8 bodies per wave).

Same reasons as tools/gen_sgpr_loop.py (one or two waves per SIMD pay a full issue slot for every scalar instruction and
for every s_nop the compiler pads a hazard with; 8-byte instructions want 8-byte alignment), same rules: the arithmetic
of pair2() bit for bit, per accumulator the records in ascending order, every producer at least two instructions away
from its consumer, 4-byte instructions in pairs.  What differs is where the operands live: here the BODIES are
wave-uniform (SGPR pairs, two bodies per packed operation, asm operands) and the j RECORDS are per lane, in two register
sets of four records that ping-pong -- the loads of one set (coalesced global_load_dwordx4, SGPR base + per-lane offset +
immediate) fly while the other set is applied.

A trip applies 8 records per lane (two sets); the caller passes the number of whole trips and finishes a remainder of
four records with the compiled loop.  The last trip requests one set past its end (blocks of the tail, or zero-filled
spare records behind the array: kSgprOverread).

Register plan (explicit, clobbered): accumulators v[40 : 40+6*NB/2*... ], temporaries, record sets -- see the constants.
"""
import os
import sys

NEG = "neg_lo:[0,1] neg_hi:[0,1]"
VBASE = 40
SP, SE, SEPS, SINC = 30, 28, 34, 36   # pointer pair, end pair, softening pair, increment
TRIP_BYTES = 8 * 1024                  # 8 blocks of 64 records x 16 bytes
WAIT = "s_waitcnt vmcnt(0)"


class Regs:
    def __init__(self, NB):
        self.NB = NB
        self.npairs = NB // 2
        self.acc = VBASE                               # 3 pairs per body pair
        self.tmp = self.acc + 6 * self.npairs          # 2 slots x 6 pairs
        self.setA = self.tmp + 24                      # 4 records x 4 dwords
        self.setB = self.setA + 16
        self.end = self.setB + 16

    def accp(self, p, k):
        b = self.acc + 6 * p + 2 * k
        return "v[%d:%d]" % (b, b + 1)

    def t(self, slot, k):
        b = self.tmp + 12 * slot + 2 * k
        return "v[%d:%d]" % (b, b + 1), "v%d" % b, "v%d" % (b + 1)

    def rec(self, s, d):
        b = (self.setA if s == 0 else self.setB) + 4 * d
        return "v[%d:%d]" % (b, b + 1), "v[%d:%d]" % (b + 2, b + 3), "v[%d:%d]" % (b, b + 3)


def record_ops(R, slot, rec_xy, rec_zw, sx, sy, sz, ax, ay, az):
    (dx, _, _), (dy, _, _), (dz, _, _), (r2, r2l, r2h), (q, _, _), (s, _, _) = (R.t(slot, k) for k in range(6))
    head = [
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[0,1] %s" % (dx, rec_xy, sx, NEG),
        "v_pk_add_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[1,1] %s" % (dy, rec_xy, sy, NEG),
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[0,1] %s" % (dz, rec_zw, sz, NEG),
        "v_pk_fma_f32 %s, %s, %s, s[%d:%d] op_sel_hi:[1,1,0]" % (r2, dz, dz, SEPS, SEPS + 1),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dy, dy, r2),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dx, dx, r2),
        "v_rsq_f32_e32 %s, %s" % (r2l, r2l),
        "v_rsq_f32_e32 %s, %s" % (r2h, r2h),
        "v_pk_mul_f32 %s, %s, %s" % (q, r2, r2),
        "v_pk_mul_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[1,1]" % (s, rec_zw, r2),
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


def set_ops(R, s, body):
    """VALU instructions applying the four records of register set s to all body pairs."""
    out = []
    if R.NB == 2:
        bx, by, bz = body[0]
        accs = (R.accp(0, 0), R.accp(0, 1), R.accp(0, 2))
        for d in (0, 2):  # two records in flight on the one body pair; record d's terms are added before record d+1's
            xy0, zw0, _ = R.rec(s, d)
            xy1, zw1, _ = R.rec(s, d + 1)
            ha, ta = record_ops(R, 0, xy0, zw0, bx, by, bz, *accs)
            hb, tb = record_ops(R, 1, xy1, zw1, bx, by, bz, *accs)
            out += zip2(ha, hb) + ta + tb
    else:
        for d in range(4):
            xy, zw, _ = R.rec(s, d)
            for p in range(0, R.npairs, 2):  # one record, two body pairs in flight
                ha, ta = record_ops(R, 0, xy, zw, *body[p], R.accp(p, 0), R.accp(p, 1), R.accp(p, 2))
                hb, tb = record_ops(R, 1, xy, zw, *body[p + 1], R.accp(p + 1, 0), R.accp(p + 1, 1), R.accp(p + 1, 2))
                out += zip2(ha, hb) + zip2(ta, tb)
    return out


def loads(R, s, voff, first_block):
    """Four coalesced loads: block b of the trip lives at pointer + 1024 b; blocks 4..7 use the second offset register."""
    out = []
    for d in range(4):
        blk = first_block + d
        reg = voff[0] if blk % 8 < 4 else voff[1]
        out.append("global_load_dwordx4 %s, %s, s[%d:%d] offset:%d" % (R.rec(s, d)[2], reg, SP, SP + 1, 1024 * (blk % 4)))
    return out


def check_alignment(body):
    off = 0
    for ins in body:
        op = ins.split()[0]
        size = 4 if (op in ("s_waitcnt", "s_add_u32", "s_addc_u32", "s_cmp_lg_u64", "s_cbranch_scc1", "s_nop") or op.endswith("_e32")) else 8
        assert size == 4 or off % 8 == 0, (ins, off)
        off += size
    return off


def check_distance(body):
    import re
    prev = set()
    for ins in body:
        if not ins.startswith("v_"):
            prev = set()
            continue
        ops = ins.split(None, 1)[1]
        regs = []
        for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", ops):
            if m.group(1):
                regs.append(frozenset("v%d" % r for r in range(int(m.group(1)), int(m.group(2)) + 1)))
            else:
                regs.append(frozenset(["v%s" % m.group(3)]))
        for s in regs[1:]:
            assert not (s & prev), "back-to-back dependency: " + ins
        prev = set(regs[0])


def emit(NB):
    R = Regs(NB)
    np_ = R.npairs
    # operand numbering: outputs 0 .. 3*np-1 (ax, ay, az per pair), then inputs: body pairs (x, y, z per pair), pointer, end, voff0, voff1
    n_out = 3 * np_
    body = [("%%%d" % (n_out + 3 * p), "%%%d" % (n_out + 3 * p + 1), "%%%d" % (n_out + 3 * p + 2)) for p in range(np_)]
    o_ptr, o_end, o_v0, o_v1 = n_out + 3 * np_, n_out + 3 * np_ + 1, n_out + 3 * np_ + 2, n_out + 3 * np_ + 3
    voff = ("%%%d" % o_v0, "%%%d" % o_v1)
    pro = ["s_mov_b64 s[%d:%d], %%%d" % (SP, SP + 1, o_ptr), "s_mov_b64 s[%d:%d], %%%d" % (SE, SE + 1, o_end),
           "s_mov_b32 s%d, 0x3a83126f" % SEPS, "s_mov_b32 s%d, 0x%x" % (SINC, TRIP_BYTES)]
    pro += ["v_mov_b32 v%d, 0" % r for r in range(R.acc, R.acc + 6 * np_)]
    pro += loads(R, 0, voff, 0) + [WAIT]
    loop = []
    loop += loads(R, 1, voff, 4)                                   # set B: blocks 4..7 of this trip
    loop += set_ops(R, 0, body)
    loop += [WAIT, "s_add_u32 s%d, s%d, s%d" % (SP, SP, SINC), "s_addc_u32 s%d, s%d, 0" % (SP + 1, SP + 1),
             "s_cmp_lg_u64 s[%d:%d], s[%d:%d]" % (SP, SP + 1, SE, SE + 1)]
    loop += loads(R, 0, voff, 0)                                   # set A of the NEXT trip (pointer already advanced)
    loop += set_ops(R, 1, body)
    loop += [WAIT, "s_cbranch_scc1 1b"]
    nbytes = check_alignment(loop)
    check_distance(loop)
    post = ["v_mov_b64 %%%d, %s" % (3 * p + k, R.accp(p, k)) for p in range(np_) for k in range(3)]
    lines = (['      "%s\\n"' % s for s in pro] + ['      ".p2align 3\\n"', '      "1:\\n"'] + ['      "%s\\n"' % s for s in loop] +
             ['      "%s\\n"' % s for s in post])
    clob = ['"v%d"' % r for r in range(VBASE, R.end)] + ['"s%d"' % r for r in (28, 29, 30, 31, 34, 35, 36)] + ['"scc"', '"memory"']
    nv = sum(1 for s in loop if s.startswith("v_"))
    outs = ", ".join('"=&v"(ax[%d]), "=&v"(ay[%d]), "=&v"(az[%d])' % (p, p, p) for p in range(np_))
    ins = ", ".join('"s"(xi[%d]), "s"(yi[%d]), "s"(zi[%d])' % (p, p, p) for p in range(np_)) + ', "s"(base), "s"(end), "v"(voff0), "v"(voff1)'
    txt = []
    txt.append("// NB = %d bodies per wave: 8 records per lane and trip, %d VALU + %d other instructions, %d bytes of loop body, VGPRs v[%d:%d]." %
               (NB, nv, len(loop) - nv, nbytes, VBASE, R.end - 1))
    txt.append("// Applies records 0 .. 8*trips-1 of this lane (lane l: array index l + 64 k) to the wave's bodies; the accumulators START AT ZERO")
    txt.append("// and are returned.  trips >= 1.  `records` = the record array (wave-uniform pointer).")
    txt.append("__device__ __forceinline__ void jlane_loop_asm_nb%d(const float4* records, int trips, const f32x2 (&xi)[%d], const f32x2 (&yi)[%d], const f32x2 (&zi)[%d],"
               % (NB, np_, np_, np_))
    txt.append("                                                   f32x2 (&ax)[%d], f32x2 (&ay)[%d], f32x2 (&az)[%d]) {" % (np_, np_, np_))
    txt.append("  const char* base = reinterpret_cast<const char*>(records);")
    txt.append("  const char* end = base + (size_t)trips * %d;" % TRIP_BYTES)
    txt.append("  const unsigned voff0 = 16u * __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), voff1 = voff0 + 4096u;")
    txt.append("  asm volatile(")
    txt.append("\n".join(lines))
    txt.append("      : %s" % outs)
    txt.append("      : %s" % ins)
    txt.append("      : %s);" % ", ".join(clob))
    txt.append("}")
    return "\n".join(txt)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "nbody-demo-2023_amd", "csrc", "nbx_jlane_loop.inc")
    parts = ["// nbx_jlane_loop.inc -- GENERATED by tools/gen_jlane_loop.py (see its docstring); do not edit.",
             "// Included by nbx_kernels.hpp inside namespace nbx.  tests/test_isa_audit.py checks it is in sync with the generator.",
             emit(2), emit(4), emit(8), ""]
    open(out, "w").write("\n".join(parts))


if __name__ == "__main__":
    main()
