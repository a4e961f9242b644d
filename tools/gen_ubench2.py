#!/usr/bin/env python3
"""gen_ubench2.py -- writes tools/ubench2.hip: lone-wave VALU issue / latency microbenchmarks for gfx950.

Every test is one kernel whose timed loop is a single asm block with explicit registers, so the
instruction order measured is exactly the order written here (hipcc schedules nothing inside).
Reported: shader cycles (s_memtime) per loop body, per instruction, and wall time, at 1/2/4/8 waves
per SIMD.  Used to design the instruction stream of the reference-order force kernel, which runs
one wave per SIMD when a rank owns 131072 of 1M bodies.

Register plan inside the asm block:
  v[2:3] v[4:5] v[6:7]      xi, yi, zi   (two i-bodies, packed)
  v[8:9] v[10:11] v[12:13]  ax, ay, az
  v[16 + 12*t ...]          temporaries of j-slot t: dx dy dz r2 q s  (6 pairs)
  s[36:67]                  8 j records {x, y, z, gm}
  s[34:35]                  softening (both halves)
"""
import sys

XI, YI, ZI, AX, AY, AZ = "v[2:3]", "v[4:5]", "v[6:7]", "v[8:9]", "v[10:11]", "v[12:13]"
NEG = "neg_lo:[0,1] neg_hi:[0,1]"


def tmp(t, k):
    b = 16 + 12 * t + 2 * k
    return "v[%d:%d]" % (b, b + 1)


def tmp_lo(t, k):
    return "v%d" % (16 + 12 * t + 2 * k)


def tmp_hi(t, k):
    return "v%d" % (16 + 12 * t + 2 * k + 1)


def jrec(u):
    """SGPR pairs of record u: (xy pair, zw pair)."""
    b = 36 + 4 * u
    return "s[%d:%d]" % (b, b + 1), "s[%d:%d]" % (b + 2, b + 3)


def pair_ops(t, u, dep=True):
    """The 14 instructions of one j record applied to the packed i-pair, in dependency order.
    t = temporary slot, u = record.  Returns a list of (text, is_trans)."""
    xy, zw = jrec(u)
    dx, dy, dz, r2, q, s = (tmp(t, k) for k in range(6))
    ops = [
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[0,1] %s" % (dx, xy, XI, NEG),
        "v_pk_add_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[1,1] %s" % (dy, xy, YI, NEG),
        "v_pk_add_f32 %s, %s, %s op_sel_hi:[0,1] %s" % (dz, zw, ZI, NEG),
        "v_pk_fma_f32 %s, %s, %s, s[34:35] op_sel_hi:[1,1,0]" % (r2, dz, dz),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dy, dy, r2),
        "v_pk_fma_f32 %s, %s, %s, %s" % (r2, dx, dx, r2),
        "v_rsq_f32_e32 %s, %s" % (tmp_lo(t, 3), tmp_lo(t, 3)),
        "v_rsq_f32_e32 %s, %s" % (tmp_hi(t, 3), tmp_hi(t, 3)),
        "v_pk_mul_f32 %s, %s, %s" % (q, r2, r2),
        "v_pk_mul_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[1,1]" % (s, zw, r2),
        "v_pk_mul_f32 %s, %s, %s" % (s, s, q),
        "v_pk_fma_f32 %s, %s, %s, %s" % (AX, dx, s, AX),
        "v_pk_fma_f32 %s, %s, %s, %s" % (AY, dy, s, AY),
        "v_pk_fma_f32 %s, %s, %s, %s" % (AZ, dz, s, AZ),
    ]
    return ops


def interleave(chains):
    """Round-robin merge of instruction lists (software pipelining by plain interleaving)."""
    out = []
    for k in range(max(len(c) for c in chains)):
        for c in chains:
            if k < len(c):
                out.append(c[k])
    return out


def skewed(nrec, depth, lag):
    """nrec records, `depth` temporary slots in flight, chain k+1 starts `lag` instructions after chain k."""
    chains = [pair_ops(u % depth, u) for u in range(nrec)]
    slots = {}
    for u, c in enumerate(chains):
        for k, op in enumerate(c):
            slots.setdefault(u * lag + k, []).append(op)
    out = []
    for key in sorted(slots):
        out.extend(slots[key])
    return out


def skewed_wrap(nrec, depth, lag):
    """Steady-state (modulo) form of skewed(): ops that would run past the end of the body wrap to its start."""
    period = nrec * lag
    keyed = []
    for u in range(nrec):
        for k, op in enumerate(pair_ops(u % depth, u)):
            keyed.append((((u * lag + k) % period), u, op))
    keyed.sort(key=lambda x: (x[0], x[1]))
    return [op for _, _, op in keyed]


TESTS = []


def test(name, body, ninstr, flops=0):
    TESTS.append((name, body, ninstr, flops))


P = ["v[%d:%d]" % (16 + 2 * k, 17 + 2 * k) for k in range(16)]
A = ["v%d" % (16 + k) for k in range(16)]

test("pk_fma indep x8", ["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k], P[k]) for k in range(8)], 8)
test("pk_fma chain dist1 x8", ["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[0], P[0]) for k in range(8)], 8)
test("pk_fma chain dist2 x8", ["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k % 2], P[k % 2]) for k in range(8)], 8)
test("pk_fma chain dist3 x9", ["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k % 3], P[k % 3]) for k in range(9)], 9)
test("pk_fma chain dist4 x8", ["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k % 4], P[k % 4]) for k in range(8)], 8)
test("pk_add sgpr-src indep x8", ["v_pk_add_f32 %s, s[36:37], %s op_sel_hi:[0,1] %s" % (P[k], P[k], NEG) for k in range(8)], 8)
test("pk_add sgpr-src odd elt x8", ["v_pk_add_f32 %s, s[36:37], %s op_sel:[1,0] op_sel_hi:[1,1] %s" % (P[k], P[k], NEG) for k in range(8)], 8)
test("pk_fma sgpr-src2 indep x8", ["v_pk_fma_f32 %s, %s, v[14:15], s[34:35] op_sel_hi:[1,1,0]" % (P[k], P[k]) for k in range(8)], 8)
test("pk_mul indep x8", ["v_pk_mul_f32 %s, %s, v[14:15]" % (P[k], P[k]) for k in range(8)], 8)
test("pk_fma + s_mov x8", sum([["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k], P[k]), "s_mov_b32 s20, s37"] for k in range(8)], []), 16)
test("pk_fma + 2 s_mov x8", sum([["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k], P[k]), "s_mov_b32 s20, s37", "s_mov_b32 s21, s38"] for k in range(8)], []), 24)
test("fma indep x16", ["v_fma_f32 %s, %s, v14, v6" % (A[k], A[k]) for k in range(16)], 16)
test("fma chain dist1 x8", ["v_fma_f32 %s, %s, v14, v6" % (A[0], A[0]) for k in range(8)], 8)
test("fma chain dist2 x8", ["v_fma_f32 %s, %s, v14, v6" % (A[k % 2], A[k % 2]) for k in range(8)], 8)
test("fma sgpr-src indep x16", ["v_fma_f32 %s, %s, s36, v6" % (A[k], A[k]) for k in range(16)], 16)
test("rsq indep x8", ["v_rsq_f32_e32 %s, %s" % (A[k], A[k]) for k in range(8)], 8)
test("rsq chain dist1 x8", ["v_rsq_f32_e32 %s, %s" % (A[0], A[0]) for k in range(8)], 8)
test("rsq,rsq,nop,pk_mul(dep) x4", sum([["v_rsq_f32_e32 %s, %s" % (A[2 * k], A[2 * k]), "v_rsq_f32_e32 %s, %s" % (A[2 * k + 1], A[2 * k + 1]), "s_nop 0",
                                         "v_pk_mul_f32 %s, %s, %s" % (P[k], P[k], P[k])] for k in range(4)], []), 16)
test("rsq,rsq,2pk,pk_mul(dep) x4", sum([["v_rsq_f32_e32 %s, %s" % (A[2 * k], A[2 * k]), "v_rsq_f32_e32 %s, %s" % (A[2 * k + 1], A[2 * k + 1]),
                                         "v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[8 + k], P[8 + k]), "v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[12 + k], P[12 + k]),
                                         "v_pk_mul_f32 %s, %s, %s" % (P[k], P[k], P[k])] for k in range(4)], []), 20)
test("6pk+rsq indep (12pk+2rsq)", (["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k], P[k]) for k in range(6)] + ["v_rsq_f32_e32 v40, v40"]
                                  + ["v_pk_fma_f32 %s, %s, v[14:15], v[6:7]" % (P[k], P[k]) for k in range(6)] + ["v_rsq_f32_e32 v41, v41"]), 14)
# the real pair arithmetic, 8 records per loop body
test("pairs: 1 record at a time", sum([pair_ops(0, u) for u in range(8)], []), 112, 8 * 128 * 20)
test("pairs: 2 interleaved", sum([interleave([pair_ops(0, u), pair_ops(1, u + 1)]) for u in range(0, 8, 2)], []), 112, 8 * 128 * 20)
test("pairs: 4 interleaved", sum([interleave([pair_ops(k, u + k) for k in range(4)]) for u in range(0, 8, 4)], []), 112, 8 * 128 * 20)
test("pairs: 8 interleaved", interleave([pair_ops(k, k) for k in range(8)]), 112, 8 * 128 * 20)
test("pairs: skew lag3 depth5", skewed(8, 5, 3), 112, 8 * 128 * 20)
test("pairs: skew lag4 depth4", skewed(8, 4, 4), 112, 8 * 128 * 20)
test("pairs: skew lag5 depth3", skewed(8, 3, 5), 112, 8 * 128 * 20)
test("pairs: skew lag7 depth2", skewed(8, 2, 7), 112, 8 * 128 * 20)
test("pairs: wrap lag3 depth5", skewed_wrap(8, 5, 3), 112, 8 * 128 * 20)
test("pairs: wrap lag4 depth4", skewed_wrap(8, 4, 4), 112, 8 * 128 * 20)
test("pairs: wrap lag5 depth3", skewed_wrap(8, 3, 5), 112, 8 * 128 * 20)
test("pairs: wrap lag7 depth2", skewed_wrap(8, 2, 7), 112, 8 * 128 * 20)
test("pairs: wrap lag2 depth8", skewed_wrap(8, 8, 2), 112, 8 * 128 * 20)
# with the scalar-load pipeline of the real kernel: two s_load_dwordx16 per 8 records into the OTHER half of a
# 64-SGPR ring would need renaming; here the loads target s[68:99] (never read) so only their issue cost shows
test("pairs: 4 interleaved + 2 s_load + waitcnt", ["s_load_dwordx16 s[68:83], s[24:25], 0x0", "s_load_dwordx16 s[84:99], s[24:25], 0x40"]
     + sum([interleave([pair_ops(k, u + k) for k in range(4)]) for u in range(0, 8, 4)], []) + ["s_waitcnt lgkmcnt(0)"], 115, 8 * 128 * 20)


def two_interleaved(recs):
    out = []
    for a in range(0, len(recs), 2):
        out += interleave([pair_ops(0, recs[a]), pair_ops(1, recs[a + 1])])
    return out


LD_A = ["s_load_dwordx16 s[36:51], s[24:25], 0x100", "s_load_dwordx16 s[52:67], s[24:25], 0x140"]
LD_B = ["s_load_dwordx16 s[68:83], s[24:25], 0x80", "s_load_dwordx16 s[84:99], s[24:25], 0xc0"]
LD_A8 = ["s_load_dwordx8 s[%d:%d], s[24:25], 0x%x" % (36 + 8 * k, 43 + 8 * k, 0x100 + 32 * k) for k in range(4)]
LD_B8 = ["s_load_dwordx8 s[%d:%d], s[24:25], 0x%x" % (68 + 8 * k, 75 + 8 * k, 0x80 + 32 * k) for k in range(4)]
WAIT = ["s_waitcnt lgkmcnt(0)"]
PTR = ["s_add_u32 s26, s26, 0x100", "s_addc_u32 s27, s27, 0"]
GA, GB = list(range(0, 8)), list(range(8, 16))


def full(name, body, recs):
    test(name, body, len(body), recs * 128 * 20)


full("trip16: no loads", two_interleaved(GA) + two_interleaved(GB), 16)
full("trip16: loads imm-offset, 2 waits, ptr add", LD_B + two_interleaved(GA) + WAIT + LD_A + two_interleaved(GB) + PTR + WAIT, 16)
full("trip16: as compiled today (4 addr pairs)", PTR + LD_B[:1] + PTR + LD_B[1:] + two_interleaved(GA) + PTR + PTR + WAIT + LD_A + two_interleaved(GB) + PTR[:1] + WAIT, 16)
full("trip16: x8 loads", LD_B8 + two_interleaved(GA) + WAIT + LD_A8 + two_interleaved(GB) + PTR + WAIT, 16)
half_a, half_b = two_interleaved(GA), two_interleaved(GB)
full("trip16: loads mid-stream", half_a[:28] + LD_B + half_a[28:] + WAIT + half_b[:28] + LD_A + half_b[28:] + PTR + WAIT, 16)
full("trip16: loads spread (1 per 2 records)", half_a[:28] + LD_B[:1] + half_a[28:84] + LD_B[1:] + half_a[84:] + WAIT + half_b[:28] + LD_A[:1] + half_b[28:84] + LD_A[1:] + half_b[84:] + PTR + WAIT, 16)
full("trip32: loads imm-offset, 4 waits", (LD_B + two_interleaved(GA) + WAIT + LD_A + two_interleaved(GB) + WAIT) + (LD_B + two_interleaved(GA) + WAIT + LD_A + two_interleaved(GB) + PTR + WAIT), 32)
full("trip16: loads only in group B regs (no rewrite of A)", LD_B + two_interleaved(GA) + WAIT + LD_B + two_interleaved(GA) + PTR + WAIT, 16)
full("trip16: waitcnt only, no loads", two_interleaved(GA) + WAIT + two_interleaved(GB) + PTR + WAIT, 16)

blkA = two_interleaved(GA)
full("E1: 1 x A block", blkA, 8)
full("E2: 2 x A blocks", blkA * 2, 16)
full("E3: 4 x A blocks", blkA * 4, 32)
full("E4: A + B (B initialised)", blkA + two_interleaved(GB), 16)
full("E5: 2 x A, s_nop between blocks", blkA + ["s_nop 0"] + blkA, 16)
full("E6: 2 x A, s_nop every 28 instr", sum([blkA[k:k + 28] + ["s_nop 0"] for k in range(0, 112, 28)], []) * 2, 16)
full("E7: 2 x A, waitcnt between blocks", blkA + WAIT + blkA + WAIT, 16)
full("E8: 3 x A blocks", blkA * 3, 24)
full("E9: 2 x A, s_nop every 14 instr", sum([blkA[k:k + 14] + ["s_nop 0"] for k in range(0, 112, 14)], []) * 2, 16)

HEADER = r'''// ubench2.hip -- GENERATED by tools/gen_ubench2.py; do not edit.  Lone-wave VALU issue / latency microbenchmarks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int ITERS = @ITERS@;
'''

KERNEL = r'''
__global__ __launch_bounds__(256) void k%d(float* out, unsigned long long* cyc, const float4* jsrc, float seed) {
  const float x0 = seed + 0.001f * (float)(threadIdx.x & 63);
  unsigned long long t0, t1;
  float r0, r1, r2;
  asm volatile(
      "s_mov_b64 s[24:25], %%6\n s_mov_b64 s[26:27], %%6\n"
      "v_add_f32 v2, 0x3e000000, %%5\n v_add_f32 v3, 0x3e800000, %%5\n v_add_f32 v4, 0x3f000000, %%5\n"
      "v_add_f32 v5, 0x3e4ccccd, %%5\n v_add_f32 v6, 0x3dcccccd, %%5\n v_add_f32 v7, 0x3f19999a, %%5\n"
      "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"
%s
      "s_mov_b32 s34, 0x3a83126f\n s_mov_b32 s35, 0x3a83126f\n"
%s
      "s_mov_b32 s22, %d\n"
      "s_memtime %%0\n s_waitcnt lgkmcnt(0)\n"
      "1:\n"
%s
      "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n"
      "s_memtime %%1\n s_waitcnt lgkmcnt(0)\n"
      "v_add_f32 %%2, v8, v9\n v_add_f32 %%3, v10, v11\n v_add_f32 %%4, v12, v13\n"
      : "=s"(t0), "=s"(t1), "=v"(r0), "=v"(r1), "=v"(r2)
      : "v"(x0), "s"(jsrc)
      : %s);
  if (r0 + r1 + r2 == 12345.678f) out[0] = r0;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
'''

MAIN = r'''
struct T { const char* name; void (*fn)(float*, unsigned long long*, const float4*, float); int ninstr; int flops; };
int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %%s CUs %%d; ITERS %%d; cycles are s_memtime ticks (shader clock), median over waves\n", prop.gcnArchName, cus, ITERS);
  float* out; float4* jsrc;
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&jsrc, 4096));
  { std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (i & 3) == 3 ? 1.0e-5f : 0.001f * (float)((i * 37) %% 1000);
    CK(hipMemcpy(jsrc, h.data(), 4096, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const T tests[] = {
%s
  };
  const int first = argc > 1 ? atoi(argv[1]) : 0, last = argc > 2 ? atoi(argv[2]) : 1 << 30;
  const int max_wps = argc > 3 ? atoi(argv[3]) : 8;
  int idx = -1;
  for (const T& t : tests) {
    ++idx;
    if (idx < first || idx > last) continue;
    for (int wps : {1, 2, 4, 8}) {
      if (wps > max_wps) continue;
      const int blocks = cus * wps;
      unsigned long long* cyc;
      CK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4));
      hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, jsrc, 0.25f);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, jsrc, 0.25f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(blocks * 4);
      CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2];
      const double per_body = med / ITERS;
      printf("%%-44s w/SIMD %%d  cyc/body %%8.1f  cyc/instr %%5.2f  cyc/instr/SIMD %%5.2f  wall %%7.3f ms  clock %%4.2f GHz", t.name, wps, per_body,
             per_body / t.ninstr, per_body / t.ninstr / wps, ms, med / (ms * 1e-3) * 1e-9);
      if (t.flops) printf("  %%5.1f %%%% of 157.3 TF", 100.0 * (double)blocks * 4 * ITERS * t.flops / (ms * 1e-3) / 157.3e12);
      printf("\n");
      CK(hipFree(cyc));
    }
  }
  return 0;
}
'''


def main():
    iters = 20000
    out = [HEADER.replace("@ITERS@", str(iters))]
    vinit = "".join('      "v_add_f32 v%d, 0x%08x, %%5\\n"\n' % (r, 0x3e000000 + 0x10000 * r) for r in range(16, 112))
    # j records: x,y,z in [0,1), gm ~ 1e-5
    sinit = ""
    for u in range(16):
        b = 36 + 4 * u
        vals = [0x3f000000 - 0x00100000 * (u % 8), 0x3e800000 + 0x00080000 * (u % 8), 0x3f400000 - 0x00040000 * (u % 8), 0x3727c5ac]
        sinit += '      "' + "".join("s_mov_b32 s%d, 0x%08x\\n " % (b + k, vals[k]) for k in range(4)) + '"\n'
    clob = ", ".join('"v%d"' % r for r in range(2, 112)) + ", " + ", ".join('"s%d"' % r for r in range(20, 100) if r not in (32, 33)) + ', "scc", "memory"'
    rows = []
    for i, (name, body, ninstr, flops) in enumerate(TESTS):
        assert len(body) == ninstr, (name, len(body), ninstr)
        btxt = "".join('      "%s\\n"\n' % ln for ln in body)
        out.append(KERNEL % (i, vinit, sinit, iters, btxt, clob))
        rows.append('    {"%s", k%d, %d, %d},' % (name, i, ninstr, flops))
    out.append(MAIN % "\n".join(rows))
    open(sys.argv[1] if len(sys.argv) > 1 else "ubench2.hip", "w").write("".join(out))


if __name__ == "__main__":
    main()
