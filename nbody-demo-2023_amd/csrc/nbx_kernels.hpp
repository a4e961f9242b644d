// nbx_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// The per-time-step work of the reference's GSimulation::start()
// (ver7/GSimulation.cpp:138-200):
//   force_kernel        all-pairs softened-gravity acceleration   (ver7:141-177)
//   integrate_kernel    v += a*dt; x += v*dt; m*v^2 partial sums  (ver7:178-198)
//   ke_reduce_kernel    ordered final sum of the partials         (ver7:179,200)
//   force_exact_kernel  validation: the reference build's arithmetic, bit for bit
//
// Data layout in HBM (all resident for the lifetime of a context):
//   posm[2][n_alloc]  {x, y, z, G*m}   one 16 B (fp32) / 32 B (fp64) record per body, double
//                     buffered: step s reads posm[cur] (every body, as j) and writes the owned
//                     slice of posm[cur^1]; entries >= n are {0,0,0,0} (zero mass => the pair term
//                     is exactly 0, so tiles need no bounds checks).
//   velm[own]         {vx, vy, vz, m}  owned slice only; velocities never leave their owner.
//   accp[S][own_pad]  {ax, ay, az, -}  partial accelerations when S workgroups split the j range.
//   ke_part[blocks]   fp64 block partials of sum m*v^2, reduced in fixed order (deterministic).
//
// Execution shape: 256-thread workgroups (4 wave64, one per SIMD).  Each lane keeps B i-bodies
// in registers (register blocking: one j record feeds B independent 13-instruction chains, which
// hides the v_rsq_f32 latency and amortises the j fetch).  The j records come either from an LDS
// tile (256 records, double buffered, every lane reads the same address => broadcast
// ds_read_b128, no bank conflicts) or from wave-uniform scalar loads (hand-pipelined
// s_load_dwordx16 into SGPRs: costs neither LDS bandwidth nor VGPRs).  No MFMA: the pair kernel is
// rsqrt/FMA-chain bound and its contraction forms cancel catastrophically in fp32.
//
// Two summation orders (include/nbx.h): REFERENCE = one accumulator per body over all
// j ascending (grid.y = 1; reproduces the reference's rounding noise, which is what parity means at
// n >= 262144), TREE = the four waves of a workgroup and grid.y workgroups each sum a j sub-range and
// the partials are added in fixed order (fastest, closest to an fp64 sum).
#pragma once
#include <hip/hip_runtime.h>

namespace nbx {

constexpr int kBlock = 256;  // threads per workgroup
constexpr int kTile = 256;   // j records per LDS tile (BASELINE.json configs[1]: "LDS j-tile=256")

template <typename T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

// ver7/GSimulation.cpp:126-127; the fp64 variant widens the float literals.
template <typename T> __host__ __device__ constexpr T softening2() { return (T)1.e-3f; }
template <typename T> __host__ __device__ constexpr T grav_const() { return (T)6.67259e-11f; }

// c/sqrt(x) with c = rsq_scale<T>().  fp32: the raw v_rsq_f32, c = 1 (<= 1 ulp; r2 >= 1e-3 so no
// denormal/zero handling is needed -- the ocml rsqrtf wrapper would add scaling code per pair).
// fp64: v_rsq_f64 is a ~2^-26 seed (measured 1.2e-8 on the accelerations); ONE Newton step
// y' = y/2 * (3 - x*y*y) takes it to ~3e-16 (measured 4.6e-15 on accelerations, 2.8e-15 on a
// 500-step kenergy trace; the gate is 1e-10).  The step's factor 1/2 is not applied here: the
// function returns 2/sqrt(x) and the records carry G*m/8 instead (exact power-of-two scaling,
// gm_prescale<double>()), which saves one multiply per pair: 3 VALU for the step instead of 4.
template <typename T> __host__ __device__ constexpr T gm_prescale() { return sizeof(T) == 8 ? (T)0.125 : (T)1; }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rsq(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double t = x * y;
  const double u = __builtin_fma(-t, y, 3.0);
  return y * u;
}

__device__ __forceinline__ float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }

// One pair: 3 sub, 3 FMA (r^2 + eps^2), 1 rsq, 3 mul (G*m_j * inv^3), 3 FMA (accumulate)
// = 12 VALU + 1 transcendental = the 20 "algorithmic" flops of DESIGN.md.
// gmj is the record's .w = G*m_j * gm_prescale<T>(), inv = rsq_scale * r2^-1/2 (see rsq above).
template <typename T>
__device__ __forceinline__ void pair(T xj, T yj, T zj, T gmj, T xi, T yi, T zi, T& ax, T& ay, T& az) {
  const T dx = xj - xi, dy = yj - yi, dz = zj - zi;
  const T r2 = fmaT(dx, dx, fmaT(dy, dy, fmaT(dz, dz, softening2<T>())));
  const T inv = rsq(r2);
  const T inv2 = inv * inv;
  const T s = (gmj * inv) * inv2;
  ax = fmaT(dx, s, ax);
  ay = fmaT(dy, s, ay);
  az = fmaT(dz, s, az);
}

// Two i-bodies per call on the packed-fp32 pipe (v_pk_add/mul/fma_f32): the j record is a
// scalar splat (op_sel), the i-bodies live in even-aligned register pairs.  12 packed VALU +
// 2 v_rsq_f32 per TWO pairs.  A/B'd against the scalar form in tools/kbench.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pair2(float xj, float yj, float zj, float gmj, f32x2 xi, f32x2 yi, f32x2 zi,
                                      f32x2& ax, f32x2& ay, f32x2& az) {
  const f32x2 dx = f32x2{xj, xj} - xi, dy = f32x2{yj, yj} - yi, dz = f32x2{zj, zj} - zi;
  const f32x2 e2 = {softening2<float>(), softening2<float>()};
  f32x2 r2 = __builtin_elementwise_fma(dz, dz, e2);
  r2 = __builtin_elementwise_fma(dy, dy, r2);
  r2 = __builtin_elementwise_fma(dx, dx, r2);
  f32x2 inv;
  inv.x = __builtin_amdgcn_rsqf(r2.x);
  inv.y = __builtin_amdgcn_rsqf(r2.y);
  const f32x2 inv2 = inv * inv;
  const f32x2 s = (f32x2{gmj, gmj} * inv) * inv2;
  ax = __builtin_elementwise_fma(dx, s, ax);
  ay = __builtin_elementwise_fma(dy, s, ay);
  az = __builtin_elementwise_fma(dz, s, az);
}

// pair2() twice, with the two instruction streams interleaved and pinned: (record a on bodies A) and (record b on bodies
// B), instruction k of the one followed by instruction k of the other, a scheduling barrier after each such couple.  Every
// result is consumed two or more instructions after it is produced, which is what the gfx940-family VALU needs (one wait
// state after a transcendental or packed result) -- hipcc left to itself schedules one 14-instruction chain at a time and
// pads it with ~1.4 s_nop per pair2 (15 % of the issue slots of the jlane loop).  Operations, association and, per
// accumulator, the order of additions are exactly pair2's.  SAME_ACC: A and B are the same bodies and share accumulators
// (two j records on one body pair): record a's terms are added before record b's.
template <bool SAME_ACC>
__device__ __forceinline__ void pair2_x2(float xa, float ya, float za, float gma, f32x2 xiA, f32x2 yiA, f32x2 ziA, f32x2& axA,
                                         f32x2& ayA, f32x2& azA, float xb, float yb, float zb, float gmb, f32x2 xiB, f32x2 yiB,
                                         f32x2 ziB, f32x2& axB, f32x2& ayB, f32x2& azB) {
#define NBX_PIN() __builtin_amdgcn_sched_barrier(0)
  const f32x2 e2 = {softening2<float>(), softening2<float>()};
  const f32x2 dxa = f32x2{xa, xa} - xiA, dxb = f32x2{xb, xb} - xiB; NBX_PIN();
  const f32x2 dya = f32x2{ya, ya} - yiA, dyb = f32x2{yb, yb} - yiB; NBX_PIN();
  const f32x2 dza = f32x2{za, za} - ziA, dzb = f32x2{zb, zb} - ziB; NBX_PIN();
  f32x2 ra = __builtin_elementwise_fma(dza, dza, e2), rb = __builtin_elementwise_fma(dzb, dzb, e2); NBX_PIN();
  ra = __builtin_elementwise_fma(dya, dya, ra); rb = __builtin_elementwise_fma(dyb, dyb, rb); NBX_PIN();
  ra = __builtin_elementwise_fma(dxa, dxa, ra); rb = __builtin_elementwise_fma(dxb, dxb, rb); NBX_PIN();
  f32x2 ia, ib;
  ia.x = __builtin_amdgcn_rsqf(ra.x); ib.x = __builtin_amdgcn_rsqf(rb.x); NBX_PIN();
  ia.y = __builtin_amdgcn_rsqf(ra.y); ib.y = __builtin_amdgcn_rsqf(rb.y); NBX_PIN();
  const f32x2 qa = ia * ia, qb = ib * ib; NBX_PIN();
  f32x2 sa = f32x2{gma, gma} * ia, sb = f32x2{gmb, gmb} * ib; NBX_PIN();
  sa = sa * qa; sb = sb * qb; NBX_PIN();
  if constexpr (SAME_ACC) {  // one set of accumulators: a's three updates, then b's (each waits three instructions for its input)
    axA = __builtin_elementwise_fma(dxa, sa, axA); ayA = __builtin_elementwise_fma(dya, sa, ayA); NBX_PIN();
    azA = __builtin_elementwise_fma(dza, sa, azA); axA = __builtin_elementwise_fma(dxb, sb, axA); NBX_PIN();
    ayA = __builtin_elementwise_fma(dyb, sb, ayA); azA = __builtin_elementwise_fma(dzb, sb, azA); NBX_PIN();
  } else {
    axA = __builtin_elementwise_fma(dxa, sa, axA); axB = __builtin_elementwise_fma(dxb, sb, axB); NBX_PIN();
    ayA = __builtin_elementwise_fma(dya, sa, ayA); ayB = __builtin_elementwise_fma(dyb, sb, ayB); NBX_PIN();
    azA = __builtin_elementwise_fma(dza, sa, azA); azB = __builtin_elementwise_fma(dzb, sb, azB); NBX_PIN();
  }
#undef NBX_PIN
}

// Separately rounded multiply and add, so the O(n) update rounds exactly like the reference's x86-64 baseline
// build (no FMA instruction there).  HIP's __fmul_rn / __fadd_rn are plain `*` / `+` and hipcc's
// default -ffp-contract=fast fuses them (seen in the ISA, caught by the NBX_KERNEL_EXACT bit-equality tests), hence
// the pragma: it clears the contract flag on exactly these operations and survives inlining.
template <typename T> __device__ __forceinline__ T mul_rn(T a, T b) {
#pragma clang fp contract(off)
  return a * b;
}
template <typename T> __device__ __forceinline__ T add_rn(T a, T b) {
#pragma clang fp contract(off)
  return a + b;
}

// Sum of one double per thread over the 256-thread workgroup, fixed order: wave64 shuffle tree,
// then the four wave sums through LDS.  Result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* lds4) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) lds4[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
  return r;
}

// ver7/GSimulation.cpp:181-197 for one body; returns m*(vx^2+vy^2+vz^2) (the reference's term,
// evaluated in T with the reference's association).
template <typename T>
__device__ __forceinline__ T euler_update(T ax, T ay, T az, T dt, typename V4<T>::type& p,
                                          typename V4<T>::type& v) {
  v.x = add_rn(v.x, mul_rn(ax, dt));
  v.y = add_rn(v.y, mul_rn(ay, dt));
  v.z = add_rn(v.z, mul_rn(az, dt));
  p.x = add_rn(p.x, mul_rn(v.x, dt));
  p.y = add_rn(p.y, mul_rn(v.y, dt));
  p.z = add_rn(p.z, mul_rn(v.z, dt));
  const T v2 = add_rn(add_rn(mul_rn(v.x, v.x), mul_rn(v.y, v.y)), mul_rn(v.z, v.z));
  return mul_rn(v.w, v2);
}

template <typename T>
struct ForceArgs {
  const typename V4<T>::type* posm;  // current positions, n_alloc records
  typename V4<T>::type* accp;        // [gridDim.y][own_pad] partial-acceleration slabs (EPI_SLAB)
  typename V4<T>::type* velm;        // EPI_ROW: owned velocities, updated in place
  typename V4<T>::type* posm_next;   // EPI_ROW: next position buffer (owned slice written)
  double* ke_part;                   // EPI_ROW: one partial per workgroup (blockIdx.x)
  int i_begin, i_count, own_pad;
  int j_per_split;                   // split y covers [y*jps, min((y+1)*jps, n_alloc)); multiple of kTile for the
                                     // LDS source, of 4*kSgprBatch (2*kSgprBatch per wave under WSPLIT) for the SGPR one
  int n_alloc;                       // multiple of kTile
  T dt;
  const typename V4<T>::type* posm_pairs;  // LOOP_ASM with one body per lane: the pair-interleaved copy of posm (pair_transpose_kernel)
  unsigned slice_bit;                // LOOP_ASM_TS: clock bit of the priority slices (kSliceBit unless NBX_SLICE_BIT overrides)
};

enum : int { JSRC_LDS = 1, JSRC_SGPR = 2 };
// records per scalar-load batch of the SGPR source (one s_load_dwordx16 = 64 B); a split's j range
// (a quarter of it under WSPLIT) must be a multiple of this
template <typename T> constexpr int kSgprBatch = 64 / (4 * (int)sizeof(T));
// spare records behind posm[n_alloc), zero-filled: the pipelined scalar loop requests one batch (16 records at most) past
// the end, the jlane kernel one trip of its widest prefetch (8 blocks of 64 records); nothing read there is ever applied
constexpr int kSgprOverread = 16 + 8 * 64;

// One 64-byte batch of j records held in 16 SGPRs, loaded by an asm s_load_dwordx16 the compiler cannot
// sink.  load() only requests; wait() is the first point at which the values may be read.
template <typename T> struct SgprBatch;
template <> struct SgprBatch<float> {
  typedef float v16 __attribute__((ext_vector_type(16)));
  v16 r;
  __device__ __forceinline__ void load_first(const void* p) {
    // s_nop: the address may have been produced by v_readfirstlane just before (VALU-written SGPR -> SMEM)
    asm volatile("s_nop 4\n\ts_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(p));
  }
  __device__ __forceinline__ void load(const void* p) { asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(p)); }
  __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }
  __device__ __forceinline__ float x(int u) const { return r[4 * u]; }
  __device__ __forceinline__ float y(int u) const { return r[4 * u + 1]; }
  __device__ __forceinline__ float z(int u) const { return r[4 * u + 2]; }
  __device__ __forceinline__ float w(int u) const { return r[4 * u + 3]; }
};
template <> struct SgprBatch<double> {
  typedef double v8 __attribute__((ext_vector_type(8)));
  v8 r;
  __device__ __forceinline__ void load_first(const void* p) {
    asm volatile("s_nop 4\n\ts_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(p));
  }
  __device__ __forceinline__ void load(const void* p) { asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(p)); }
  __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }
  __device__ __forceinline__ double x(int u) const { return r[4 * u]; }
  __device__ __forceinline__ double y(int u) const { return r[4 * u + 1]; }
  __device__ __forceinline__ double z(int u) const { return r[4 * u + 2]; }
  __device__ __forceinline__ double w(int u) const { return r[4 * u + 3]; }
};
// one s_waitcnt for a group of batches: names every destination "+s" so no consumer is scheduled above it
template <typename T, int G>
__device__ __forceinline__ void sgpr_wait(SgprBatch<T> (&b)[G]) {
  if constexpr (G == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0].r));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0].r), "+s"(b[1].r));
}

enum : int { MATH_SCALAR = 0, MATH_PACKED = 1 };
// How the SGPR kernel's j loop is scheduled: LOOP_CXX = hipcc schedules the C++ loop below; LOOP_ASM = the hand-scheduled
// gfx950 loop of nbx_sgpr_loop.inc (packed fp32, B = 2 or 4, no wave split): same operations in the same order, hence
// the same bits (tests compare the two), but no s_mov splats, one pointer update per trip and 8-byte aligned VOP3P code:
// worth 13 % when a SIMD holds a single wave, where every scalar instruction costs a full 4-cycle issue slot.
// With ONE body per lane LOOP_ASM is the two-j-records-per-packed-operation loop (sgpr_loop_asm_jpair): slices that leave less than
// one wave per SIMD at two bodies per lane (<= 65536 owned bodies) get twice the waves at 76 cycles per two pairs instead of 2 x 56;
// it reads the pair-interleaved copy of the records that pair_transpose_kernel rebuilds every step.
// LOOP_ASM_TS = the same loop with time-sliced wave priority, for shapes that put two waves on a SIMD for the whole launch
// (reference order, grid.y == 1, 257..512 workgroups on 256 CUs): this chip issues the waves of a SIMD in strict age order,
// so without it they run one after the other -- the older one leaves the loop at 0.50 of the kernel time -- and the younger
// one has nobody to fill its issue bubbles (tools/wave_fair.hip).  +4.5 % at n = 262144; nothing to gain
// with one wave per SIMD (-0.6 %: the six scalar instructions) or with three and more (profiles/r02_time_sliced_ab.txt).
// LOOP_ASM_PF = the same loop plus one L2-prefetch load per trip, for launches that leave ONE wave per SIMD (grid.x <= CUs, e.g. a rank
// that owns 131072 of 1M bodies): there the arithmetic of one ring group (512 cycles) is all the cover a scalar load gets, every wave of
// an XCD asks for the same line at about the same time, and what they all wait for is the first requester's Infinity-Cache round trip
// (~545 cycles).  +3.5 % at one wave per SIMD, -0.4 ... -1.4 % with two or more (profiles/r04_b2_prefetch_ab.txt).
enum : int { LOOP_CXX = 0, LOOP_ASM = 1, LOOP_ASM_TS = 2, LOOP_ASM_PF = 3 };
// clock bit (s_memrealtime counts 10 ns) that selects the favoured slot parity: slices of 2^16 x 10 ns = 0.66 ms.  Measured
// 2^14 ... 2^19 within 1 % of each other, 2^16-2^17 best; shorter slices lose to the time the unfavoured wave needs to reach
// its next decision, longer ones to the imbalance of the last slice.
constexpr unsigned kSliceBit = 1u << 16;
#include "nbx_sgpr_loop.inc"
// What a workgroup does with its accelerations:
//   EPI_SLAB  write them to its split's slab (the separate integrate_kernel, or nbx_accel, consumes the slabs)
//   EPI_ROW   single split (gridDim.y == 1): integrate its bodies directly, no slab
// (Round 1 also had a "last arriver integrates" epilogue for split shapes -- agent-scope release / ticket / acquire.  It was
// bit-equal but slower than the extra launch at every size, and force_jlane_kernel below now gives launch-bound sizes
// their single launch per step without any inter-workgroup hand-off; it was removed.)
enum : int { EPI_SLAB = 0, EPI_ROW = 1 };

// The B i-bodies a lane keeps in registers, and how one j record is applied to them.
template <typename T, int B, int MATH>
struct IBodies {
  T xi[B], yi[B], zi[B], ax[B], ay[B], az[B];
  __device__ __forceinline__ void set(int b, T x, T y, T z) {
    xi[b] = x; yi[b] = y; zi[b] = z;
    ax[b] = ay[b] = az[b] = (T)0;
  }
  __device__ __forceinline__ void apply(T xj, T yj, T zj, T gmj) {
#pragma unroll
    for (int b = 0; b < B; ++b) pair<T>(xj, yj, zj, gmj, xi[b], yi[b], zi[b], ax[b], ay[b], az[b]);
  }
  __device__ __forceinline__ void get(int b, T& x, T& y, T& z) const { x = ax[b]; y = ay[b]; z = az[b]; }
};

template <int B>
struct IBodies<float, B, MATH_PACKED> {
  static_assert(B % 2 == 0, "packed math needs an even number of bodies per lane");
  f32x2 xi[B / 2], yi[B / 2], zi[B / 2], ax[B / 2], ay[B / 2], az[B / 2];
  __device__ __forceinline__ void set(int b, float x, float y, float z) {
    xi[b / 2][b & 1] = x; yi[b / 2][b & 1] = y; zi[b / 2][b & 1] = z;
    ax[b / 2][b & 1] = 0.f; ay[b / 2][b & 1] = 0.f; az[b / 2][b & 1] = 0.f;
  }
  __device__ __forceinline__ void apply(float xj, float yj, float zj, float gmj) {
#pragma unroll
    for (int b = 0; b < B / 2; ++b) pair2(xj, yj, zj, gmj, xi[b], yi[b], zi[b], ax[b], ay[b], az[b]);
  }
  __device__ __forceinline__ void get(int b, float& x, float& y, float& z) const {
    x = ax[b / 2][b & 1]; y = ay[b / 2][b & 1]; z = az[b / 2][b & 1];
  }
  // all records of [first, last) in ascending order through the hand-scheduled loop (LOOP_ASM)
  __device__ __forceinline__ void apply_range_asm(const float4* first, const float4* last) {
    static_assert(B == 2 || B == 4, "the asm loop exists for 2 and 4 bodies per lane");
    if constexpr (B == 2) sgpr_loop_asm_b2(first, last, xi[0], yi[0], zi[0], ax[0], ay[0], az[0]);
    else sgpr_loop_asm_b4(first, last, xi[0], yi[0], zi[0], xi[1], yi[1], zi[1], ax[0], ay[0], az[0], ax[1], ay[1], az[1]);
  }
  // the same with the L2 prefetch (LOOP_ASM_PF)
  __device__ __forceinline__ void apply_range_asm_pf(const float4* first, const float4* last) {
    static_assert(B == 2 || B == 4, "the asm loop exists for 2 and 4 bodies per lane");
    if constexpr (B == 2) sgpr_loop_asm_b2_pf(first, last, xi[0], yi[0], zi[0], ax[0], ay[0], az[0]);
    else sgpr_loop_asm_b4_pf(first, last, xi[0], yi[0], zi[0], xi[1], yi[1], zi[1], ax[0], ay[0], az[0], ax[1], ay[1], az[1]);
  }
  // the same with time-sliced wave priority (LOOP_ASM_TS); slot_bit = slice_bit for waves in odd slots of their SIMD, else 0
  __device__ __forceinline__ void apply_range_asm_ts(const float4* first, const float4* last, unsigned slice_bit, unsigned slot_bit) {
    static_assert(B == 2 || B == 4, "the asm loop exists for 2 and 4 bodies per lane");
    if constexpr (B == 2) sgpr_loop_asm_b2_ts(first, last, slice_bit, slot_bit, xi[0], yi[0], zi[0], ax[0], ay[0], az[0]);
    else sgpr_loop_asm_b4_ts(first, last, slice_bit, slot_bit, xi[0], yi[0], zi[0], xi[1], yi[1], zi[1], ax[0], ay[0], az[0], ax[1], ay[1], az[1]);
  }
};

// ---------------------------------------------------------------------------------------------
// force_kernel: block 256 = 4 wave64.
//   WSPLIT == false: grid (ceil(i_count / (256*B)), S).  Lane t of workgroup bx owns bodies
//     i_begin + bx*256*B + b*256 + t, b = 0..B-1, and every wave walks the whole j range of split y.
//   WSPLIT == true (small n, SGPR source only): grid (ceil(i_count / (64*B)), S).  The four waves own
//     the SAME 64*B bodies (lane l: bx*64*B + b*64 + l) and each walks one quarter of the split's
//     j range; the four partial sums are added in wave order through LDS.  Four times the workgroups
//     for the same number of partial-acceleration slabs.
// ---------------------------------------------------------------------------------------------
template <typename T, int B, int JSRC, int EPI, int MINW, int MATH = MATH_SCALAR, bool WSPLIT = false, int LOOP = LOOP_CXX>
__global__ __launch_bounds__(kBlock, MINW) void force_kernel(const ForceArgs<T> a) {
  using T4 = typename V4<T>::type;
  static_assert(!WSPLIT || (JSRC == JSRC_SGPR && EPI != EPI_ROW), "wave split exists for the SGPR kernel with slabs only");
  static_assert(LOOP == LOOP_CXX || (JSRC == JSRC_SGPR && MATH == MATH_PACKED && sizeof(T) == 4 && (B == 2 || B == 4)) ||
                    (LOOP == LOOP_ASM && JSRC == JSRC_SGPR && MATH == MATH_SCALAR && sizeof(T) == 4 && B == 1 && !WSPLIT),
                "the hand-scheduled loop exists for the packed fp32 SGPR kernels with 2 or 4 bodies per lane, and as the two-records-per-operation loop for 1");
  const int t = threadIdx.x;
  constexpr int kStride = WSPLIT ? 64 : kBlock;  // distance between a lane's consecutive bodies
  const int base = blockIdx.x * (kStride * B) + (WSPLIT ? (t & 63) : t);

  IBodies<T, B, MATH> ib;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    int li = base + b * kStride;
    li = li < a.i_count ? li : a.i_count - 1;  // padded lanes shadow the last owned body
    const T4 p = a.posm[a.i_begin + li];
    ib.set(b, p.x, p.y, p.z);
  }

  int j0 = blockIdx.y * a.j_per_split;
  int j1 = j0 + a.j_per_split;
  j1 = j1 < a.n_alloc ? j1 : a.n_alloc;
  if constexpr (WSPLIT) {
    // wave-uniform quarter of [j0, j1); quarter length is a multiple of the load batch
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int q = a.j_per_split >> 2;
    j0 += wave * q;
    const int e = j0 + q;
    j1 = e < j1 ? e : j1;
  }

  if constexpr (JSRC == JSRC_LDS) {
    __shared__ T4 tile[2][kTile];
    T4 pre = a.posm[j0 + t];
    tile[0][t] = pre;
    __syncthreads();
    int cur = 0;
    for (int jt = j0; jt < j1; jt += kTile) {
      const bool more = jt + kTile < j1;
      if (more) pre = a.posm[jt + kTile + t];  // in flight under the tile's arithmetic
      const T4* tl = tile[cur];
#pragma unroll 8
      for (int jj = 0; jj < kTile; ++jj) {
        const T4 pj = tl[jj];  // same address in every lane: broadcast read
        ib.apply(pj.x, pj.y, pj.z, pj.w);
      }
      if (more) tile[cur ^ 1][t] = pre;
      __syncthreads();  // one barrier per tile: readers of `cur` done, writers of `cur^1` done
      cur ^= 1;
    }
  } else if constexpr (LOOP == LOOP_ASM) {
    // j range = a positive multiple of kSgprAsmTrip<B> records (the host rounds j_per_split to 64, to 256 under WSPLIT where a
    // wave walks a quarter of it; n_alloc is a multiple of 256)
    if constexpr (B == 1) {
      // one body per lane: two consecutive j records per packed operation, out of the pair-interleaved copy (same offsets)
      if (j0 < j1) sgpr_loop_asm_jpair(a.posm_pairs + j0, a.posm_pairs + j1, f32x2{ib.xi[0], ib.yi[0]}, f32x2{ib.zi[0], ib.zi[0]}, ((unsigned)t & (kSgprJpairPrefetchLines - 1u)) * 64u + kSgprJpairPrefetchBytes, ib.ax[0], ib.ay[0], ib.az[0]);
    } else {
      if (j0 < j1) ib.apply_range_asm(a.posm + j0, a.posm + j1);
    }
  } else if constexpr (LOOP == LOOP_ASM_PF) {
    if (j0 < j1) ib.apply_range_asm_pf(a.posm + j0, a.posm + j1);
  } else if constexpr (LOOP == LOOP_ASM_TS) {
    if (j0 < j1) {
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));  // bits 3:0 = the wave's slot on its SIMD
      const unsigned slice = __builtin_amdgcn_readfirstlane(a.slice_bit);
      ib.apply_range_asm_ts(a.posm + j0, a.posm + j1, slice, (hwid & 1u) ? slice : 0u);
      __builtin_amdgcn_s_setprio(0);
    }
  } else {
    // Wave-uniform j index => the records travel by s_load_dwordx16 (64 B = kSgprBatch records) into
    // SGPRs and feed the VALU as scalar operands: no LDS bandwidth, no VGPRs, no barrier.  Two
    // register batches ping-pong: the load of the NEXT batch is issued before the current one is
    // consumed, so the scalar-cache latency sits under ~200 VALU instructions even when a SIMD holds
    // a single wave (small n).  hipcc would sink a plain C++ prefetch back to its use, so the loads
    // are asm statements (guide section 5.7 form (ii): "=s" load, later a wait naming the destination
    // "+s"; SMEM returns out of order, hence lgkmcnt(0)).  The final trip over-reads one batch past
    // the split; the array carries kSgprOverread spare records for the last split.
    constexpr int U = kSgprBatch<T>;
    // G batches travel together.  SMEM returns out of order, so only lgkmcnt(0) is a usable wait: the cover a
    // request gets is the arithmetic of ONE group.  The wave-split kernel runs 8 waves per SIMD and needs <= 80 SGPRs
    // (G = 1); the plain SGPR kernel serves the reference-order shapes, which run 1-2 waves per SIMD when a rank owns
    // few bodies and need the longer cover of G = 2 (8 j records, 64 SGPRs of payload).
    constexpr int G = WSPLIT ? 1 : 2;
    const char* p = reinterpret_cast<const char*>(a.posm + j0);
    if (j0 < j1) {
      SgprBatch<T> ba[G];
      ba[0].load_first(p);
#pragma unroll
      for (int g = 1; g < G; ++g) ba[g].load(p + 64 * g);
      sgpr_wait<T, G>(ba);
      // Invariant at the loop head AND at the back edge: group `ba` has landed.  No asm-loaded value is in flight
      // across the back edge, so a register copy the compiler may insert for the loop-carried value can never read
      // (or be overtaken by) a pending scalar load.  tests/test_isa_audit.py checks the compiled code: nothing touches
      // a batch's SGPRs between its s_load and the next s_waitcnt lgkmcnt(0).
      for (int j = j0; j < j1; j += 2 * G * U) {
        SgprBatch<T> bb[G];
#pragma unroll
        for (int g = 0; g < G; ++g) bb[g].load(p + 64 * (G + g));
        __builtin_amdgcn_sched_barrier(0);  // keep the arithmetic below the requests (no operand ties it)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int u = 0; u < U; ++u) ib.apply(ba[g].x(u), ba[g].y(u), ba[g].z(u), ba[g].w(u));
        __builtin_amdgcn_sched_barrier(0);
        sgpr_wait<T, G>(bb);
#pragma unroll
        for (int g = 0; g < G; ++g) ba[g].load(p + 64 * (2 * G + g));  // next trip's first group (over-read on the last trip)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int u = 0; u < U; ++u) ib.apply(bb[g].x(u), bb[g].y(u), bb[g].z(u), bb[g].w(u));
        __builtin_amdgcn_sched_barrier(0);
        sgpr_wait<T, G>(ba);
        p += 128 * G;
      }
    }
  }

  if constexpr (EPI == EPI_ROW) {
    __shared__ double ksum[4];
    double ke = 0.0;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int li = base + b * kBlock;
      if (li < a.i_count) {
        T4 p = a.posm[a.i_begin + li];
        T4 v = a.velm[li];
        T axb, ayb, azb;
        ib.get(b, axb, ayb, azb);
        ke += (double)euler_update<T>(axb, ayb, azb, a.dt, p, v);
        a.velm[li] = v;
        a.posm_next[a.i_begin + li] = p;
      }
    }
    const double s = block_sum(ke, ksum);
    if (t == 0) a.ke_part[blockIdx.x] = s;
    return;
  }

  // ---- this workgroup's partial accelerations -> slab blockIdx.y -------------------------------
  T4* out = a.accp + (size_t)blockIdx.y * a.own_pad;
  if constexpr (WSPLIT) {
    __shared__ T red[3][3][B][64];  // [wave-1][component][body][lane]
    const int lane = t & 63, wave = t >> 6;
    if (wave > 0) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        T x, y, z;
        ib.get(b, x, y, z);
        red[wave - 1][0][b][lane] = x; red[wave - 1][1][b][lane] = y; red[wave - 1][2][b][lane] = z;
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int li = base + b * kStride;
        T4 r;
        ib.get(b, r.x, r.y, r.z);
#pragma unroll
        for (int w = 0; w < 3; ++w) { r.x += red[w][0][b][lane]; r.y += red[w][1][b][lane]; r.z += red[w][2][b][lane]; }
        r.w = (T)0;
        if (li < a.i_count) out[li] = r;
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int li = base + b * kBlock;
      if (li < a.i_count) {
        T4 r;
        ib.get(b, r.x, r.y, r.z);
        r.w = (T)0;
        out[li] = r;
      }
    }
  }

}

// ---------------------------------------------------------------------------------------------
// force_jlane_kernel (NBX_KERNEL_JLANE; fp32, tree order, launch-bound sizes): the roles of i and j swapped.
//   A WAVE owns NB bodies and holds them wave-uniformly (scalar loads -> SGPRs, two bodies per packed op).  Its 64
//   LANES each walk every 64th j record (lane l: j = l, l + 64, ...), one record per lane in VGPRs, requested D records
//   (one trip) ahead with coalesced 1-KiB loads.  Every lane ends with a partial sum for each of the NB bodies; a transpose through LDS
//   (one padded float4 column per body: conflict-free) lets lane t < NB add body t's 64 partials in lane order, and that
//   lane integrates the body at once (same euler_update as everywhere else) and contributes to the wave's energy partial.
// One launch per time step: no partial-acceleration slabs, no integrate kernel, no inter-workgroup hand-off -- which is
// what bounds n <= 16k (a step there was two dependent launches whose fixed costs exceeded the arithmetic: 21 us per
// step for 0.9 us of pair work at n = 2048).  Parallelism is ceil(own / NB) waves, so NB is chosen to give ~1024 waves
// (one per SIMD); the inner loop is the same 12 packed + 2 rsq instructions per two pairs as the other kernels.
// Summation order: j = lane (mod 64) ascending per lane, then lanes 0..63 in order -- a tree, like SGPRW's, so the kernel
// serves the tree-order range only (n <= 131072).  acc_only != 0: store the accelerations to accp instead
// of integrating (nbx_accel).
// ---------------------------------------------------------------------------------------------
#include "nbx_jlane_loop.inc"

// LOOP_ASM (NB = 2, 4, 8): whole trips of 8 records per lane go through the generated loop, a remainder of four records
// through the compiled one -- the same operations in the same order either way (tests compare the bits).
template <int NB, int D, int LOOP = LOOP_CXX>
__global__ __launch_bounds__(kBlock, 1) void force_jlane_kernel(const ForceArgs<float> a, const int acc_only) {
  static_assert(NB % 2 == 0 && NB >= 2 && NB <= 16 && D >= 1, "two bodies per packed operation; body state must fit the SGPR file");
  static_assert(LOOP == LOOP_CXX || NB <= 8, "the generated loop exists for 2, 4 and 8 bodies per wave");
  __shared__ float4 red[4][NB][65];  // [wave][body][lane], one float4 of padding per column: lanes t read 1040 B apart
  __shared__ double ksum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
  const int b0 = wave * NB;

  f32x2 xi[NB / 2], yi[NB / 2], zi[NB / 2], ax[NB / 2], ay[NB / 2], az[NB / 2];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int li = b0 + b;
    li = li < a.i_count ? li : a.i_count - 1;  // waves / bodies past the end shadow the last owned body
    const float4 p = a.posm[a.i_begin + li];   // wave-uniform index: a scalar load
    xi[b / 2][b & 1] = p.x; yi[b / 2][b & 1] = p.y; zi[b / 2][b & 1] = p.z;
    ax[b / 2][b & 1] = 0.f; ay[b / 2][b & 1] = 0.f; az[b / 2][b & 1] = 0.f;
  }

  // the body lane t < NB will integrate at the end: its position and velocity are requested now, so that the two loads
  // land under the j loop instead of in front of the epilogue (at n = 2048 the whole launch is a few microseconds)
  const int li = b0 + lane;
  const bool mine = lane < NB && li < a.i_count;
  float4 pe = make_float4(0.f, 0.f, 0.f, 0.f), ve = pe;
  if (mine) {
    pe = a.posm[a.i_begin + li];
    ve = a.velm[li];
  }

  // Two register sets of D records ping-pong: the loads of the NEXT D records are issued before the current D are
  // applied, so a request has D x NB/2 x 56 cycles of arithmetic to land under (one wave per SIMD has no other wave to
  // hide an L2 round trip behind).  Requests may run up to D blocks past the end of the array (zero-filled spare records,
  // kSgprOverread); what they return is never applied.
  static_assert(64 * D <= kSgprOverread - 16, "the farthest request is D blocks of 64 records past the array (main loop: ra at k + 2 D <= K; the tail requests rb only when part of it is applied)");
  const float4* pj = a.posm + lane;
  const int K = a.n_alloc >> 6;  // records per lane; n_alloc is a multiple of 256, so K >= 4
  float4 ra[D], rb[D];
  auto request = [&](float4 (&r)[D], int k0) {
#pragma unroll
    for (int d = 0; d < D; ++d) r[d] = pj[(size_t)64 * (k0 + d)];
  };
  // one record on body pairs (p, p + 1), or -- a wave with a single body pair -- two records on that pair: two pinned,
  // interleaved instruction streams either way (pair2_x2)
  auto apply_record = [&](const float4& r) {
    if constexpr (NB >= 4) {
#pragma unroll
      for (int p = 0; p < NB / 2; p += 2)
        pair2_x2<false>(r.x, r.y, r.z, r.w, xi[p], yi[p], zi[p], ax[p], ay[p], az[p], r.x, r.y, r.z, r.w, xi[p + 1], yi[p + 1], zi[p + 1],
                        ax[p + 1], ay[p + 1], az[p + 1]);
    } else {
      pair2(r.x, r.y, r.z, r.w, xi[0], yi[0], zi[0], ax[0], ay[0], az[0]);
    }
  };
  auto apply = [&](const float4 (&r)[D]) {
    if constexpr (NB >= 4) {
#pragma unroll
      for (int d = 0; d < D; ++d) apply_record(r[d]);
    } else {
      static_assert(NB >= 4 || D % 2 == 0, "a single body pair takes its records two at a time");
#pragma unroll
      for (int d = 0; d < D; d += 2)
        pair2_x2<true>(r[d].x, r[d].y, r[d].z, r[d].w, xi[0], yi[0], zi[0], ax[0], ay[0], az[0], r[d + 1].x, r[d + 1].y, r[d + 1].z,
                       r[d + 1].w, xi[0], yi[0], zi[0], ax[0], ay[0], az[0]);
    }
  };
  auto apply_some = [&](const float4 (&r)[D], int count) {  // wave-uniform count in [0, D]
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < count) apply_record(r[d]);
  };
  int k = 0;
  if constexpr (LOOP == LOOP_ASM) {
    const int trips = K >> 3;  // K is a multiple of 4: the remainder is 0 or 4 records
    if (trips > 0) {
      if constexpr (NB == 2) jlane_loop_asm_nb2(a.posm, trips, xi, yi, zi, ax, ay, az);
      else if constexpr (NB == 4) jlane_loop_asm_nb4(a.posm, trips, xi, yi, zi, ax, ay, az);
      else jlane_loop_asm_nb8(a.posm, trips, xi, yi, zi, ax, ay, az);
      k = trips << 3;
    }
    if (k < K) request(ra, k);
  } else {
    request(ra, 0);
    for (; k + 2 * D <= K; k += 2 * D) {
      request(rb, k + D);
      apply(ra);
      request(ra, k + 2 * D);
      apply(rb);
    }
  }
  if (k < K) {  // fewer than 2 D records left: ra holds records k .. k + D - 1
    // rb is requested only if any of it will be applied: the farthest request of the whole kernel is then the main loop's
    // ra at k + 2 D <= K, i.e. at most D blocks (64 D records) past the array -- what kSgprOverread reserves
    if (K - k > D) request(rb, k + D);
    apply_some(ra, K - k < D ? K - k : D);
    apply_some(rb, K - k - D > 0 ? K - k - D : 0);
  }

  // lane partials -> LDS columns; lane t < NB adds the 64 partials of body t in lane order
#pragma unroll
  for (int b = 0; b < NB; ++b) red[w][b][lane] = make_float4(ax[b / 2][b & 1], ay[b / 2][b & 1], az[b / 2][b & 1], 0.f);
  __builtin_amdgcn_wave_barrier();  // same wave, in-order LDS queue: the reads below see the writes above
  double ke = 0.0;
  if (mine) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll 8
    for (int l = 0; l < 64; ++l) {
      const float4 q = red[w][lane][l];
      sx += q.x; sy += q.y; sz += q.z;
    }
    if (acc_only) {
      a.accp[li] = make_float4(sx, sy, sz, 0.f);
    } else {
      ke = (double)euler_update<float>(sx, sy, sz, a.dt, pe, ve);
      a.velm[li] = ve;
      a.posm_next[a.i_begin + li] = pe;
    }
  }
  // one energy partial per workgroup, fixed order (wave shuffle tree, then the four waves)
  const double s = block_sum(ke, ksum);
  if (threadIdx.x == 0 && !acc_only) a.ke_part[blockIdx.x] = s;
}

// The fp64 form of force_jlane_kernel: same decomposition (a wave owns NB bodies wave-uniformly, its lanes split j, LDS
// transpose, the wave integrates its own bodies), plain fp64 arithmetic (pair<double>: there is no packed fp64), records
// of 32 bytes.  NB <= 8: eight bodies are 48 SGPRs of coordinates.
template <int NB, int D>
__global__ __launch_bounds__(kBlock, 1) void force_jlane_kernel_f64(const ForceArgs<double> a, const int acc_only) {
  static_assert(NB >= 1 && NB <= 8 && D >= 1 && 64 * D <= kSgprOverread - 16, "body state must fit the SGPR file; the farthest request is D blocks past the array");
  __shared__ double4 red[4][NB][65];  // [wave][body][lane] + one column of padding (2080 B between the lanes that read)
  __shared__ double ksum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + w);
  const int b0 = wave * NB;
  double xi[NB], yi[NB], zi[NB], ax[NB], ay[NB], az[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int li = b0 + b;
    li = li < a.i_count ? li : a.i_count - 1;
    const double4 p = a.posm[a.i_begin + li];  // wave-uniform index: scalar loads
    xi[b] = p.x; yi[b] = p.y; zi[b] = p.z;
    ax[b] = ay[b] = az[b] = 0.0;
  }
  const int li = b0 + lane;
  const bool mine = lane < NB && li < a.i_count;
  double4 pe = make_double4(0.0, 0.0, 0.0, 0.0), ve = pe;
  if (mine) {
    pe = a.posm[a.i_begin + li];
    ve = a.velm[li];
  }
  const double4* pj = a.posm + lane;
  const int K = a.n_alloc >> 6;
  double4 ra[D], rb[D];
  auto request = [&](double4 (&r)[D], int k0) {
#pragma unroll
    for (int d = 0; d < D; ++d) r[d] = pj[(size_t)64 * (k0 + d)];
  };
  auto apply_record = [&](const double4& r) {
#pragma unroll
    for (int b = 0; b < NB; ++b) pair<double>(r.x, r.y, r.z, r.w, xi[b], yi[b], zi[b], ax[b], ay[b], az[b]);
  };
  int k = 0;
  request(ra, 0);
  for (; k + 2 * D <= K; k += 2 * D) {
    request(rb, k + D);
#pragma unroll
    for (int d = 0; d < D; ++d) apply_record(ra[d]);
    request(ra, k + 2 * D);
#pragma unroll
    for (int d = 0; d < D; ++d) apply_record(rb[d]);
  }
  if (k < K) {
    if (K - k > D) request(rb, k + D);  // as in the fp32 kernel: never more than D blocks past the array
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (k + d < K) apply_record(ra[d]);
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (k + D + d < K) apply_record(rb[d]);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) red[w][b][lane] = make_double4(ax[b], ay[b], az[b], 0.0);
  __builtin_amdgcn_wave_barrier();
  double ke = 0.0;
  if (mine) {
    double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll 8
    for (int l = 0; l < 64; ++l) {
      const double4 q = red[w][lane][l];
      sx += q.x; sy += q.y; sz += q.z;
    }
    if (acc_only) {
      a.accp[li] = make_double4(sx, sy, sz, 0.0);
    } else {
      ke = euler_update<double>(sx, sy, sz, a.dt, pe, ve);
      a.velm[li] = ve;
      a.posm_next[a.i_begin + li] = pe;
    }
  }
  const double s = block_sum(ke, ksum);
  if (threadIdx.x == 0 && !acc_only) a.ke_part[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// force_exact_kernel (NBX_KERNEL_EXACT): the reference's acceleration loop with the reference's rounding.
// The pinned build of ver7 (g++ -O2, x86-64 baseline) compiles ver7/GSimulation.cpp:153-173 to a scalar,
// strictly sequential loop: r2 = ((dx*dx + dy*dy) + dz*dz) + eps, inv = 1.0f / sqrtf(r2) (IEEE sqrtss, divss),
// term = ((((d*G)*m)*inv)*inv)*inv, sum += term for j = 0..n-1, acc = 0 + sum.  One thread per body does exactly
// that: contraction off, HIP's correctly rounded fp32 sqrt and divide.  Validation path, not a fast path.
// ---------------------------------------------------------------------------------------------
// The reference's loop body, spelled once; the fp-contract pragma is lexical, so the text is instantiated inside
// each pragma's scope below.
#define NBX_EXACT_ROW()                                                            \
  T sx = (T)0, sy = (T)0, sz = (T)0;                                               \
  for (int j = 0; j < n; ++j) {                                                    \
    const T4 pj = posm[j];                                                         \
    const T m = mass[j];                                                           \
    const T dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;                  \
    const T r2 = ((dx * dx + dy * dy) + dz * dz) + eps;                            \
    const T inv = (T)1 / sqrt(r2);                                                 \
    sx = sx + ((((dx * G) * m) * inv) * inv) * inv;                                \
    sy = sy + ((((dy * G) * m) * inv) * inv) * inv;                                \
    sz = sz + ((((dz * G) * m) * inv) * inv) * inv;                                \
  }                                                                                \
  r.x = (T)0 + sx; r.y = (T)0 + sy; r.z = (T)0 + sz;

// CONTRACT = false: contraction off (the pinned reference build).  CONTRACT = true: the same source lines with FMA
// contraction allowed, i.e. another legitimate build of the reference (diagnostic NBX_KERNEL_EXACT_FMA).
template <typename T, bool CONTRACT>
__global__ __launch_bounds__(kBlock) void force_exact_kernel(const typename V4<T>::type* __restrict__ posm,
                                                             const T* __restrict__ mass,
                                                             typename V4<T>::type* __restrict__ accp, int i_begin,
                                                             int i_count, int n) {
  using T4 = typename V4<T>::type;
  const int li = blockIdx.x * kBlock + threadIdx.x;
  if (li >= i_count) return;
  const T4 pi = posm[i_begin + li];
  const T eps = softening2<T>(), G = grav_const<T>();
  T4 r;
  r.w = (T)0;
  if constexpr (CONTRACT) {
#pragma clang fp contract(fast)
    NBX_EXACT_ROW()
  } else {
#pragma clang fp contract(off)
    NBX_EXACT_ROW()
  }
  accp[li] = r;
}
#undef NBX_EXACT_ROW

// ---------------------------------------------------------------------------------------------
// pair_transpose_kernel: the pair-interleaved copy of the record array that sgpr_loop_asm_jpair reads.  Records 2k and 2k+1,
// {x0 y0 z0 w0}{x1 y1 z1 w1}, become {x0 x1 y0 y1}{z0 z1 w0 w1} at the same byte offset: a packed operand is one aligned 64-bit
// register pair, so the two records a packed instruction works on must be neighbours component by component.  One thread per
// pair, 32 B in and 32 B out, once per step in front of the force launch (n = 1M: 32 MiB of L2 / Infinity-Cache traffic, ~10 us
// against a 15 ms step); npairs = n_alloc / 2 (the spare records behind the array stay zero in both layouts).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pair_transpose_kernel(const float4* __restrict__ posm, float4* __restrict__ pairs, int npairs) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= npairs) return;
  const float4 a = posm[2 * k], b = posm[2 * k + 1];
  pairs[2 * k] = make_float4(a.x, b.x, a.y, b.y);
  pairs[2 * k + 1] = make_float4(a.z, b.z, a.w, b.w);
}

// ---------------------------------------------------------------------------------------------
// integrate_kernel: one body per thread; sums the S partial accelerations in split order.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void integrate_kernel(const typename V4<T>::type* __restrict__ posm_cur,
                                                           typename V4<T>::type* __restrict__ posm_next,
                                                           typename V4<T>::type* __restrict__ velm,
                                                           const typename V4<T>::type* __restrict__ accp,
                                                           int nsplit, int own_pad, int i_begin, int i_count,
                                                           T dt, double* __restrict__ ke_part) {
  using T4 = typename V4<T>::type;
  __shared__ double ksum[4];
  const int li = blockIdx.x * kBlock + threadIdx.x;
  double ke = 0.0;
  if (li < i_count) {
    T ax = (T)0, ay = (T)0, az = (T)0;
    for (int s = 0; s < nsplit; ++s) {
      const T4 q = accp[(size_t)s * own_pad + li];
      ax += q.x; ay += q.y; az += q.z;
    }
    T4 p = posm_cur[i_begin + li];
    T4 v = velm[li];
    ke = (double)euler_update<T>(ax, ay, az, dt, p, v);
    velm[li] = v;
    posm_next[i_begin + li] = p;
  }
  const double s = block_sum(ke, ksum);
  if (threadIdx.x == 0) ke_part[blockIdx.x] = s;
}

// One workgroup; thread t sums partials t, t+256, ... then the block tree: fixed order.
__global__ __launch_bounds__(kBlock) void ke_reduce_kernel(const double* __restrict__ ke_part, int nparts,
                                                           double* __restrict__ out) {
  __shared__ double ksum[4];
  double v = 0.0;
  for (int k = threadIdx.x; k < nparts; k += kBlock) v += ke_part[k];
  const double s = block_sum(v, ksum);
  if (threadIdx.x == 0) *out = s;
}

}  // namespace nbx
