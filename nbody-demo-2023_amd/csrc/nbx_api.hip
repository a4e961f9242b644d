// nbx_api.hip -- the C-ABI of include/nbx.h over the gfx950 kernels of nbx_kernels.hpp.
//
// One context = one GPU's share of the reference's GSimulation::start() loop
// (ver7/GSimulation.cpp:138-200): it owns bodies [i_begin, i_begin+i_count), keeps
// {x,y,z,G*m} of ALL bodies resident (double buffered) and {vx,vy,vz,m} of its own.
// No CPU fallback exists: without a HIP device every entry point fails with NBX_ERR_DEVICE.
#include <hip/hip_runtime.h>


#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "nbx_internal.hpp"
#include "nbx_kernels.hpp"

using namespace nbx;

using namespace nbx_detail;

namespace nbx_detail {
std::string& last_error() {
  thread_local std::string err;
  return err;
}
}  // namespace nbx_detail

namespace {

constexpr int kMaxProfiledLaunches = 8192;
// NBX_ORDER_AUTO: fp32 sums of more terms than this use the reference's order.  131072 x 500 steps agrees with the
// reference to 2e-5 in tree order (profiles/r01_validate_orders_n131072_s500.log); 262144 x 200 does not (1.3e-3).
constexpr int kTreeOrderMaxN = 131072;
// NBX_KERNEL_AUTO, tree order: contexts that own at most this many bodies step with ONE launch (force_jlane_kernel)
// (12288: 41 us against SGPRW's 48.  Round 3, profiles/r03_band_sweep.txt: between 12288 and 16384 the wave-split kernel falls back
// to two bodies per lane and, at sizes whose splits are not whole tiles, to the compiled loop -- 40-45 % -- while the one-launch kernel
// with 8 bodies per wave is 6-15 % ahead: 13000 50.3 vs 53.2 us, 15000 58.0 vs 65.6, 16000 62.8 vs 72.3.)  16384 ITSELF is excluded:
// there the two tie within 2 %, and SGPRW's summation tree (S = 32) is the one whose chaotic n = 16384 x 500 run -- BASELINE
// configs[1] -- stays inside the 1e-4 gate at every printed step (profiles/r02_config1_by_kernel.txt, r03_config1_by_shape.txt).
constexpr int kJlaneMaxOwn = 16383;
// Tree order, wave-split kernel: contexts that own up to this many bodies keep round 1's split rule (S = 32 at 16384).  Fewer
// splits are 5 % faster there (profiles/r03_band_sweep.txt: S = 4 or 8), but BASELINE.json configs[1] -- n = 16384 x 500 steps, 450 of
// them after the bounce -- is decided at the reference's own noise level, and of S = 2, 4, 8, 16, 32 only the tree of S = 32
// lands inside 1e-4 at every printed row (7.7e-5; the others 1.05e-4 ... 1.43e-4, two builds of the reference itself 1.3e-4:
// profiles/r03_config1_by_shape.txt).  tests/test_parity_gpu.py::test_config1_launch_shape_is_frozen pins it.
constexpr int kRound1SplitMaxOwn = 16384;
constexpr int kJlaneMaxOwnF64 = 12288;  // fp64 form: 92 us against 97 at 12288, SGPRW ahead at 16384 (profiles/r02_jlane_f64_ab.txt)

}  // namespace

namespace {

// ------------------------------------------------------------------------------------------
// kernel dispatch
// ------------------------------------------------------------------------------------------
template <typename T, int B, int JSRC, int EPI, int MATH, bool WS, int LOOP = LOOP_CXX>
void launch_force_t(const ForceArgs<T>& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((force_kernel<T, B, JSRC, EPI, 1, MATH, WS, LOOP>), grid, dim3(kBlock), 0, st, a);
}

// does a hand-scheduled (LOOP_ASM) instance exist for this combination?
template <typename T, int JSRC, int EPI, int MATH, bool WS>
constexpr bool kHasAsmLoop = sizeof(T) == 4 && JSRC == JSRC_SGPR && MATH == MATH_PACKED;

template <typename T>
using ForceLauncher = void (*)(const ForceArgs<T>&, dim3, hipStream_t);

template <typename T, int JSRC, int EPI, int MATH, bool WS>
ForceLauncher<T> pick_b(int B, int loop) {
  if constexpr (sizeof(T) == 4 && JSRC == JSRC_SGPR && MATH == MATH_SCALAR && !WS) {
    // one body per lane: the hand-scheduled loop is the two-j-records-per-operation one (sgpr_loop_asm_jpair)
    if (B == 1 && loop != LOOP_CXX) return launch_force_t<T, 1, JSRC, EPI, MATH, WS, LOOP_ASM>;
  }
  if constexpr (kHasAsmLoop<T, JSRC, EPI, MATH, WS>) {
    if constexpr (!WS && EPI == EPI_ROW) {  // time-sliced priority: the whole-launch-resident reference-order shapes only
      if (loop == LOOP_ASM_TS) {
        if (B == 2) return launch_force_t<T, 2, JSRC, EPI, MATH, WS, LOOP_ASM_TS>;
        if (B == 4) return launch_force_t<T, 4, JSRC, EPI, MATH, WS, LOOP_ASM_TS>;
        return nullptr;
      }
      if (loop == LOOP_ASM_PF) {  // L2 prefetch: same scope (one workgroup row, one wave per SIMD)
        if (B == 2) return launch_force_t<T, 2, JSRC, EPI, MATH, WS, LOOP_ASM_PF>;
        if (B == 4) return launch_force_t<T, 4, JSRC, EPI, MATH, WS, LOOP_ASM_PF>;
        return nullptr;
      }
    }
    if (loop == LOOP_ASM_TS || loop == LOOP_ASM_PF) loop = LOOP_ASM;  // e.g. nbx_accel's slab form of a time-sliced context
    if (loop == LOOP_ASM) {
      if (B == 2) return launch_force_t<T, 2, JSRC, EPI, MATH, WS, LOOP_ASM>;
      if (B == 4) return launch_force_t<T, 4, JSRC, EPI, MATH, WS, LOOP_ASM>;
      return nullptr;
    }
  } else if (loop != LOOP_CXX) {
    return nullptr;
  }
  switch (B) {
    case 1:
      if constexpr (MATH == MATH_SCALAR) return launch_force_t<T, 1, JSRC, EPI, MATH, WS>;
      return nullptr;
    // fp32 runs B >= 2 on the packed pipe only: the scalar-math B >= 2 instances are not built
    case 2:
      if constexpr (sizeof(T) == 8 || MATH == MATH_PACKED) return launch_force_t<T, 2, JSRC, EPI, MATH, WS>;
      return nullptr;
    case 4:
      if constexpr (sizeof(T) == 8 || MATH == MATH_PACKED) return launch_force_t<T, 4, JSRC, EPI, MATH, WS>;
      return nullptr;
    case 8:
      if constexpr (sizeof(T) == 4 && !WS && MATH == MATH_PACKED) return launch_force_t<T, 8, JSRC, EPI, MATH, WS>;
      return nullptr;
  }
  return nullptr;
}

template <typename T, int JSRC, int MATH, bool WS>
ForceLauncher<T> pick_epi(int B, int epi, int loop) {
  if (epi == EPI_ROW) {
    if constexpr (!WS) return pick_b<T, JSRC, EPI_ROW, MATH, WS>(B, loop);
    return nullptr;
  }
  return pick_b<T, JSRC, EPI_SLAB, MATH, WS>(B, loop);
}

template <typename T, int MATH>
ForceLauncher<T> pick(int B, int variant, int epi, int loop) {
  if (variant == NBX_KERNEL_SGPRW) return pick_epi<T, JSRC_SGPR, MATH, true>(B, epi, loop);
  if (variant == NBX_KERNEL_SGPR) return pick_epi<T, JSRC_SGPR, MATH, false>(B, epi, loop);
  return pick_epi<T, JSRC_LDS, MATH, false>(B, epi, loop);
}

// One body per lane on the plain SGPR kernel: its hand-scheduled loop packs two consecutive j records per operation and reads the
// pair-interleaved copy of the records (nbx_ctx::posm_pairs)
bool jpair_shape(const nbx_ctx* c) {
  return c->precision == 32 && c->variant == NBX_KERNEL_SGPR && c->math == MATH_SCALAR && c->B == 1 && c->jps % kSgprAsmTrip<1> == 0;
}

// Does the hand-scheduled loop exist for this shape?  (mirror of kHasAsmLoop for run-time shape decisions)
bool asm_loop_available(const nbx_ctx* c, int epi) {
  (void)epi;
  if (c->variant == NBX_KERNEL_JLANE) return c->precision == 32 && (c->B == 2 || c->B == 4 || c->B == 8);
  if (c->variant == NBX_KERNEL_SGPRW && c->jps % 256 != 0) return false;  // a wave walks a quarter of a split: whole trips only
  if (jpair_shape(c)) return true;
  return c->precision == 32 && (c->variant == NBX_KERNEL_SGPR || c->variant == NBX_KERNEL_SGPRW) && c->math == MATH_PACKED && (c->B == 2 || c->B == 4);
}

template <typename T>
ForceLauncher<T> pick_force(const nbx_ctx* c, int epi) {
  // nbx_accel runs the EPI_SLAB form of a context whose step kernel may be another epilogue: decide per call
  const int loop = (c->loop != LOOP_CXX && asm_loop_available(c, epi)) ? c->loop : LOOP_CXX;
  if constexpr (sizeof(T) == 4) {
    if (c->math == MATH_PACKED) return pick<float, MATH_PACKED>(c->B, c->variant, epi, loop);
  }
  return pick<T, MATH_SCALAR>(c->B, c->variant, epi, loop);
}


// Bodies per lane of the reference-order kernel (one chain per owned body, S = 1).  Its run time is quantised: the
// ceil(own / (256 B)) workgroups are spread over the CUs, and a launch takes as long as the fullest CU, which holds
// r = ceil(workgroups / CUs) of them.  Measured on MI355X at n = 1048576 with the hand-scheduled loop for B = 2 and 4
// (profiles/r02_reference_order_thresholds.txt), ms for r = 1, 2, 3, ...: B = 1, compiled loop: 31.0, 48.6, 70.3, 91, 112 (plain VALU ops);
// B = 2: 31.5 (30.0 with the L2 prefetch and the 256-record trips of LOOP_ASM_PF, round 4), 59.6, 88.5, 118;  B = 4: 59.3, 117.4, 175.6, 234.6 -- linear in r after
// the first workgroup.  With two workgroups on the fullest CU the time-sliced loop applies (LOOP_ASM_TS): B = 2, r = 2 then costs 58.0
// (profiles/r02_time_sliced_ab.txt: 57.98 ms for 262144 of 1M bodies), B = 4, r = 2 117.0.  Round 4: one body per lane with the
// two-j-records-per-operation loop (sgpr_loop_asm_jpair, `jpair`): 17.9 for r = 1 (65536 of 1M bodies: 48.8 % of the roofline against 28.4 %
// for the compiled loop and 27.8 % for B = 2 on half the CUs), 34.3 for r = 2 (profiles/r04_jpair_ab.txt) -- so it takes every slice of up to
// 256 x CUs = 65536 bodies, and B = 2 keeps 65537 ... 131072.  Pick the B with the smallest estimate; ties go to the larger B (fewer
// workgroups stream the j records).  Only the ratios matter, so the table serves every n.
int reference_order_bodies_per_lane(int own, int cus, int max_b, bool jpair, double* cost_out = nullptr) {
  struct Cost { int b; double first, next, two; };
  static const Cost kCost[] = {{1, 31.0, 20.2, 0.0}, {2, 30.0, 29.25, 58.0}, {4, 59.8, 58.2, 117.0}};
  static const Cost kJpair = {1, 17.9, 16.4, 0.0};  // one body per lane, two j records per packed operation
  int best = 1;
  double best_t = 0.0;
  for (const auto& k0 : kCost) {
    const auto& k = (k0.b == 1 && jpair) ? kJpair : k0;
    if (k.b > max_b) continue;
    const int wgs = ceil_div(own, kBlock * k.b);
    const int r = std::max(1, ceil_div(wgs, std::max(1, cus)));
    const double t = (r == 2 && k.two > 0.0) ? k.two : k.first + k.next * (r - 1);
    if (best_t == 0.0 || t <= best_t * 1.01) { best = k.b; best_t = std::min(t, best_t == 0.0 ? t : best_t); }
  }
  // Where two bodies per lane load every CU evenly too (twice the workgroups, all CUs with the same count), they win over four
  // by 2.5 % at 262144 owned bodies and tie from 524288 up (the younger wave of a SIMD fills the older one's issue bubbles;
  // profiles/r02_loop_ab_asm_vs_cxx.txt, same process on two boxes): take them.
  if (best == 4 && max_b >= 2 && ceil_div(own, kBlock * 2) % std::max(1, cus) == 0) best = 2;
  if (cost_out) *cost_out = best_t;
  return best;
}

// j-splits of the wave-split kernel with the hand-scheduled loop (round 3, profiles/r03_band_sweep.txt).  Round 1's rule -- 32
// workgroups per CU, i.e. S = 32 up to n = 65536 -- suited the compiler-scheduled loop, which needed eight waves per SIMD to
// hide its own bubbles.  The hand-scheduled loop is at its rate with two, and every extra split costs a slab write, a slab
// read by integrate_kernel and a shorter j loop per wave.  The launch lasts as long as the fullest CU: ceil(bi S / CUs)
// workgroups of 1/S of the j range each.  Take the S (power of two) that minimises that product; among equals the smallest
// S that still gives every CU two workgroups.  Measured optimum at every size tried: 24576 -> 8 (+2.6 % over S = 32),
// 32768 -> 4 (+1.9 %), 49152 -> 4 (+1.2 %), 65536 -> 2 (+1.2 %); three workgroups on half the CUs (24576 with S = 4) is 20 % slower.
int balanced_j_split(int bi, int cus, int max_s) {
  double best_cost = 0.0;
  for (int S = 1; S <= max_s; S *= 2) {
    const double cost = (double)ceil_div(bi * S, cus) / S;
    if (best_cost == 0.0 || cost < best_cost) best_cost = cost;
  }
  int pick = 0, largest = 1;
  for (int S = 1; S <= max_s; S *= 2) {
    if ((double)ceil_div(bi * S, cus) / S > best_cost * 1.0001) continue;
    largest = S;
    if (!pick && bi * S >= 2 * cus) pick = S;
  }
  return pick ? pick : largest;
}

// Launch shape.  Measured with tools/kbench on MI355X (profiles/r01_kbench_*): the force kernel is
// VALU-issue bound and wants all 8 wave slots of every SIMD filled, i.e. >= 8192 workgroups of 256
// threads (32 per CU).  Fastest shape from n = 2k to 1M: j records in SGPRs, the four waves of a
// workgroup sharing 64*B bodies and splitting the j range (NBX_KERNEL_SGPRW), B = 4 bodies per lane
// (two packed register pairs; B = 2 for short i ranges), plus S j-range splits across workgroups:
// 58-60 % of the fp32 roofline at n >= 64k, 52 % at 16k, vs 52 % / 36 % for B = 8 / LDS tile.
void auto_shape(nbx_ctx* c, const nbx_opts& o) {
  const int cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
  const int target_wgs = cus * 32;
  int variant = o.kernel_variant;
  if (variant == NBX_KERNEL_EXACT || variant == NBX_KERNEL_EXACT_FMA) {  // one thread per body, no blocking, no splits, separate integrate kernel
    c->B = 1; c->S = 1; c->jps = c->n_alloc; c->math = MATH_SCALAR; c->variant = variant; c->epi = EPI_SLAB;
    c->order = NBX_ORDER_REFERENCE;  // one accumulator per body, j ascending: it IS the reference's loop
    c->grid = dim3(ceil_div(c->i_count, kBlock), 1);
    return;
  }
  // Summation order (include/nbx.h).  The reference adds a body's n terms one after the other into one fp32
  // accumulator; from n = 262144 that sum carries ~1e-5 of rounding noise per step which heats the system (kenergy
  // +5e-4..1e-3 against an fp64 run; 1.6e-5 over 500 steps at n = 131072).  A tree of partial sums does not reproduce
  // that, a single accumulator per body in the same j order does (to 5e-5 / 5e-7, tools/validate_big.py) -- at the
  // price of one chain per owned body.  The noise is a property of the LENGTH of the sum, i.e. of n, not of how many
  // bodies this context owns: every rank of a sharded run takes the same decision.
  int order = o.summation_order;
  if (order != NBX_ORDER_REFERENCE && order != NBX_ORDER_TREE) {
    const bool shape_given = o.j_split > 0 || variant == NBX_KERNEL_SGPRW || variant == NBX_KERNEL_JLANE;  // tree-only shapes
    if (o.j_split == 1 && variant != NBX_KERNEL_SGPRW) order = NBX_ORDER_REFERENCE;
    // fp64 keeps the tree: its summation noise (~1e-13) is far below the 1e-10 fp64 gate in either order
    else order = (!shape_given && c->precision == 32 && c->n > kTreeOrderMaxN) ? NBX_ORDER_REFERENCE : NBX_ORDER_TREE;
  }
  c->order = order;
  if (order == NBX_ORDER_REFERENCE) {
    if (variant != NBX_KERNEL_LDS && variant != NBX_KERNEL_SGPR) variant = NBX_KERNEL_SGPR;
    int B = o.bodies_per_lane;
    const int maxBr = c->precision == 32 ? 8 : 4;
    if (B != 1 && B != 2 && B != 4 && B != 8) B = 0;
    if (B > maxBr) B = maxBr;
    // fp32, plain SGPR kernel, hand-scheduled loops allowed: one body per lane means sgpr_loop_asm_jpair
    const bool jpair = c->precision == 32 && variant == NBX_KERNEL_SGPR && o.inner_loop != NBX_LOOP_CXX;
    if (B == 0) B = reference_order_bodies_per_lane(c->i_count, cus, maxBr, jpair);
    c->B = B; c->S = 1; c->jps = c->n_alloc; c->variant = variant;
    c->math = (c->precision == 32 && B >= 2) ? MATH_PACKED : MATH_SCALAR;
    c->epi = o.fused_epilogue == 2 ? EPI_SLAB : EPI_ROW;
    c->grid = dim3(ceil_div(c->i_count, kBlock * B), 1);
    return;
  }
  // Launch-bound sizes (fp32): one launch per step with the lanes of a wave splitting j (force_jlane_kernel).  Bodies per
  // wave: the power of two that gives about one wave per SIMD (1024 waves), between 2 and 16.
  const int max_nb = c->precision == 32 ? 16 : 8;  // fp64 bodies take two SGPRs per coordinate
  const bool jlane_auto = variant == NBX_KERNEL_AUTO && o.j_split <= 0 && o.bodies_per_lane == 0 && o.fused_epilogue != 2 &&
                          c->i_count <= (c->precision == 32 ? kJlaneMaxOwn : kJlaneMaxOwnF64);
  if (variant == NBX_KERNEL_JLANE || jlane_auto) {
    int NB = o.bodies_per_lane;
    if ((NB != 2 && NB != 4 && NB != 8 && NB != 16) || NB > max_nb) {
      // A launch lasts as long as the fullest SIMD: ceil(waves / SIMDs) rounds of NB bodies each, times what a body-round costs
      // with that NB.  Measured per body-round and per j record, relative to NB = 8 (profiles/r03_jlane_band.txt: the same ratios
      // at 12288, 13000 and 16383 bodies): 2 bodies per wave 1.29 (every wave streams all j records and transposes through LDS
      // for two bodies only), 4 -> 1.06, 8 -> 1.00 (the generated loop), 16 -> 1.06 (compiled loop only).  Smallest product wins:
      // 2048 -> 2, 4096 -> 4, 8192 -> 8, 12288 -> 4 (three full rounds), 13000 ... 16383 -> 8 -- the measured optimum at each.
      // Round 2 counted body-rounds alone, which sent 13000 and 14336 to NB = 2 (56.6 us against 50.1 with 8).
      // fp64 (no generated loop, other ratios not measured): body-rounds alone, ties to the larger NB, as in round 2.
      long best = 0;
      for (int nb = 2; nb <= max_nb; nb *= 2) {
        const long weight = c->precision != 32 ? 100 : (nb == 2 ? 129 : nb == 8 ? 100 : 106);
        const long cost = (long)ceil_div(ceil_div(c->i_count, nb), cus * 4) * nb * weight;
        if (best == 0 || cost <= best) { best = cost; NB = nb; }
      }
    }
    c->B = NB; c->S = 1; c->jps = c->n_alloc; c->math = c->precision == 32 ? MATH_PACKED : MATH_SCALAR; c->variant = NBX_KERNEL_JLANE; c->epi = EPI_ROW;
    c->grid = dim3(ceil_div(ceil_div(c->i_count, NB), 4), 1);
    return;
  }
  if (variant != NBX_KERNEL_LDS && variant != NBX_KERNEL_SGPR && variant != NBX_KERNEL_SGPRW) variant = NBX_KERNEL_SGPRW;
  const int maxB = (c->precision == 32 && variant != NBX_KERNEL_SGPRW) ? 8 : 4;
  int B = o.bodies_per_lane;
  if (B != 1 && B != 2 && B != 4 && B != 8) B = 0;
  if (B > maxB) B = maxB;
  if (B == 0) B = c->i_count >= 16384 ? 4 : 2;
  const int iblk = (variant == NBX_KERNEL_SGPRW ? 64 : kBlock) * B;  // bodies per workgroup
  // j-range granularity of one split: a whole LDS tile / two pipelined SGPR batches (per wave)
  // (the hand-scheduled loop of the plain SGPR kernel walks whole trips of up to 64 records)
  // (round 3: where the balanced split rule applies -- few, long splits -- a split is a whole number of 256-record tiles, so
  // that every wave's quarter is whole trips of the hand-scheduled loop whatever n is: n = 50000 used to get 32 splits of 1568
  // records and, with them, the compiler-scheduled loop)
  // Only where tree order is what AUTO takes (n <= 131072).  Above that it runs on request alone -- asked for because it is closer to the
  // true sum than the reference's single chain -- and the number of chains is what buys that: n = 262144 keeps its 8 x 4, 1M its 2 x 4.
  const bool balanced = o.j_split <= 0 && c->precision == 32 && variant == NBX_KERNEL_SGPRW && c->i_count > kRound1SplitMaxOwn && c->n <= kTreeOrderMaxN;
  constexpr int kSgprGran = kSgprAsmTrip<1> > 64 ? kSgprAsmTrip<1> : 64;  // whole trips of every hand-scheduled loop of the plain SGPR kernel
  const int gran = variant == NBX_KERNEL_LDS ? kTile : (variant == NBX_KERNEL_SGPR ? kSgprGran : (balanced ? 256 : 32));
  static_assert(kSgprGran % kSgprAsmTrip<2> == 0 && kSgprGran % kSgprAsmTrip<4> == 0 && kSgprGran % kSgprAsmTrip<1> == 0 && kTile % kSgprGran == 0,
                "j ranges are whole trips of the asm loop");
  const int max_split = std::max(1, c->n_alloc / gran);
  int S = o.j_split;
  if (S <= 0) {
    const int bi = ceil_div(c->i_count, iblk);
    if (balanced) S = balanced_j_split(bi, cus, std::min(32, max_split));
    else S = std::min(32, ceil_div(target_wgs, bi));
    // the S partial-acceleration slabs are written and re-read every step: keep them <= 256 MiB
    while (S > 1 && (size_t)S * c->own_pad * c->rec > ((size_t)256 << 20)) S /= 2;
  }
  S = std::max(1, std::min(S, max_split));
  int jps = round_up(ceil_div(c->n_alloc, S), gran);
  S = ceil_div(c->n_alloc, jps);  // drop empty tail splits
  c->B = B;
  c->S = S;
  c->jps = jps;
  c->math = (c->precision == 32 && B >= 2) ? MATH_PACKED : MATH_SCALAR;
  c->variant = variant;
  // fused_epilogue: 0 auto, 1 on, 2 off.  A single split integrates directly (EPI_ROW); shapes with j-splits always
  // use the separate integrate kernel (one launch per step for small n is NBX_KERNEL_JLANE's job).
  if (o.fused_epilogue == 2) c->epi = EPI_SLAB;
  else if (S == 1 && variant != NBX_KERNEL_SGPRW) c->epi = EPI_ROW;
  else c->epi = EPI_SLAB;
  c->grid = dim3(ceil_div(c->i_count, iblk), S);
}

template <typename T>
int enqueue_force(nbx_ctx* c, int epi, double dt) {
  if (c->variant == NBX_KERNEL_EXACT || c->variant == NBX_KERNEL_EXACT_FMA) {
    using T4 = typename V4<T>::type;
    if (c->variant == NBX_KERNEL_EXACT)
      hipLaunchKernelGGL((force_exact_kernel<T, false>), c->grid, dim3(kBlock), 0, c->stream, (const T4*)c->posm[c->cur],
                         (const T*)c->mass_all, (T4*)c->accp, c->i_begin, c->i_count, c->n);
    else
      hipLaunchKernelGGL((force_exact_kernel<T, true>), c->grid, dim3(kBlock), 0, c->stream, (const T4*)c->posm[c->cur],
                         (const T*)c->mass_all, (T4*)c->accp, c->i_begin, c->i_count, c->n);
    HIP_TRY(hipGetLastError());
    return NBX_OK;
  }
  ForceLauncher<T> fn = c->variant == NBX_KERNEL_JLANE ? nullptr : pick_force<T>(c, epi);
  if (!fn && c->variant != NBX_KERNEL_JLANE) return fail(NBX_ERR_ARG, "no kernel instance for this bodies_per_lane / precision");
  ForceArgs<T> a{};
  using T4 = typename V4<T>::type;
  a.posm = (const T4*)c->posm[c->cur];
  a.accp = (T4*)c->accp;
  a.velm = (T4*)c->velm;
  a.posm_next = (T4*)c->posm[c->cur ^ 1];
  a.ke_part = c->ke_part;
  a.i_begin = c->i_begin;
  a.i_count = c->i_count;
  a.own_pad = c->own_pad;
  a.j_per_split = c->jps;
  a.n_alloc = c->n_alloc;
  a.dt = (T)dt;
  a.slice_bit = c->slice_bit;
  a.posm_pairs = (const T4*)c->posm_pairs;
  if constexpr (sizeof(T) == 4) {
    if (c->loop == LOOP_ASM && jpair_shape(c)) {  // this step's pair-interleaved copy of the records (all n_alloc of them: other ranks' blocks arrived by all-gather)
      const int npairs = c->n_alloc / 2;
      hipLaunchKernelGGL(pair_transpose_kernel, dim3(ceil_div(npairs, kBlock)), dim3(kBlock), 0, c->stream, (const float4*)c->posm[c->cur],
                         (float4*)c->posm_pairs, npairs);
    }
  }
  const bool prof = c->profiling && c->ev_used + 2 <= c->ev.size();
  if (prof) HIP_TRY(hipEventRecord(c->ev[c->ev_used], c->stream));
  if (c->variant == NBX_KERNEL_JLANE) {
    if constexpr (sizeof(T) == 4) {
      const int acc_only = epi == EPI_SLAB ? 1 : 0;  // nbx_accel asks for the slab form: accelerations only
      const bool hand = c->loop == LOOP_ASM;  // the generated main loop (NB <= 8); D = prefetch depth of the compiled loop / tail
      switch (c->B) {
        case 2:
          if (hand) hipLaunchKernelGGL((force_jlane_kernel<2, 8, LOOP_ASM>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only);
          else hipLaunchKernelGGL((force_jlane_kernel<2, 8>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only);
          break;
        case 4:
          if (hand) hipLaunchKernelGGL((force_jlane_kernel<4, 8, LOOP_ASM>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only);
          else hipLaunchKernelGGL((force_jlane_kernel<4, 8>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only);
          break;
        case 8:
          if (hand) hipLaunchKernelGGL((force_jlane_kernel<8, 4, LOOP_ASM>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only);
          else hipLaunchKernelGGL((force_jlane_kernel<8, 4>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only);
          break;
        default: hipLaunchKernelGGL((force_jlane_kernel<16, 4>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
      }
    } else {
      const int acc_only = epi == EPI_SLAB ? 1 : 0;
      switch (c->B) {
        case 2: hipLaunchKernelGGL((force_jlane_kernel_f64<2, 8>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
        case 4: hipLaunchKernelGGL((force_jlane_kernel_f64<4, 4>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
        default: hipLaunchKernelGGL((force_jlane_kernel_f64<8, 4>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
      }
    }
  } else {
    fn(a, c->grid, c->stream);
  }
  if (prof) {
    HIP_TRY(hipEventRecord(c->ev[c->ev_used + 1], c->stream));
    c->ev_used += 2;
  }
  HIP_TRY(hipGetLastError());
  return NBX_OK;
}

// energy partials one step of this context's shape writes: one per workgroup of the kernel that integrates
int step_ke_parts(const nbx_ctx* c) { return c->epi != EPI_SLAB ? (int)c->grid.x : ceil_div(c->i_count, kBlock); }

// one local step: force (+ integrate) into the next buffer; does not swap
template <typename T>
int enqueue_step(nbx_ctx* c, double dt) {
  using T4 = typename V4<T>::type;
  int rc = enqueue_force<T>(c, c->epi, dt);
  if (rc) return rc;
  if (c->epi != EPI_SLAB) {
    c->ke_parts = step_ke_parts(c);
  } else {
    const int blocks = step_ke_parts(c);
    hipLaunchKernelGGL((integrate_kernel<T>), dim3(blocks), dim3(kBlock), 0, c->stream,
                       (const T4*)c->posm[c->cur], (T4*)c->posm[c->cur ^ 1], (T4*)c->velm,
                       (const T4*)c->accp, c->S, c->own_pad, c->i_begin, c->i_count, (T)dt, c->ke_part);
    HIP_TRY(hipGetLastError());
    c->ke_parts = blocks;
  }
  return NBX_OK;
}

int enqueue_step_any(nbx_ctx* c, double dt) {
  return c->precision == 32 ? enqueue_step<float>(c, dt) : enqueue_step<double>(c, dt);
}

// A window of `unit` (even) steps captured once per buffer parity and replayed: the two launches of a
// step cost ~3.5 us each from the host but ~1.5 us as graph nodes (MI355X_MICROARCH.md, rows
// 'boundary' / 'graph-replay-floor'), which is what bounds n <= 16k.
int graph_unit_exec(nbx_ctx* c, int unit, double dt, hipGraphExec_t* out) {
  for (auto& g : c->graphs)
    if (g.steps == unit && g.parity == c->cur && g.dt == dt) { *out = g.exec; return NBX_OK; }
  const int cur0 = c->cur;
  hipGraph_t graph = nullptr;
  HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = NBX_OK;
  for (int s = 0; s < unit && rc == NBX_OK; ++s) {
    rc = enqueue_step_any(c, dt);
    c->cur ^= 1;
  }
  c->cur = cur0;
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc != NBX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return fail(NBX_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(NBX_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  c->graphs.push_back({unit, cur0, dt, exec});
  *out = exec;
  return NBX_OK;
}

int ensure_ke_cap(nbx_ctx* c, int need) {
  if (need <= c->ke_cap) return NBX_OK;
  if (c->ke_dev) HIP_TRY(hipFree(c->ke_dev));
  c->ke_dev = nullptr;
  c->ke_cap = 0;
  HIP_TRY(hipMalloc(&c->ke_dev, sizeof(double) * (size_t)need));
  c->ke_cap = need;
  return NBX_OK;
}

}  // namespace
int nbx_detail::enqueue_ke_reduce(nbx_ctx* c, int slot) {
  hipLaunchKernelGGL(ke_reduce_kernel, dim3(1), dim3(kBlock), 0, c->stream, (const double*)c->ke_part,
                     c->ke_parts, c->ke_dev + slot);
  HIP_TRY(hipGetLastError());
  return NBX_OK;
}
namespace {

int drain_profile(nbx_ctx* c) {
  for (size_t k = 0; k + 1 < c->ev_used; k += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]));
    c->force_ms_total += ms;
    c->force_timed += 1;
  }
  c->ev_used = 0;
  return NBX_OK;
}

}  // namespace
// What one force launch of this context would cost, in relative units, if the context owned `own` bodies instead: the cost table of
// reference_order_bodies_per_lane in reference order (a step function of `own`: the launch lasts as long as its fullest CU), and `own`
// itself in tree order, where j-splits keep the time close to proportional.  The tuner of nbx_group_retune uses the RATIO of two such
// values to predict what a move of the shares would do before it makes it.
double nbx_detail::model_force_cost(const nbx_ctx* c, int own) {
  if (own <= 0) return 0.0;
  if (c->order != NBX_ORDER_REFERENCE) return (double)own;
  const int cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
  const bool jpair = c->precision == 32 && c->variant == NBX_KERNEL_SGPR;
  double t = 0.0;
  (void)reference_order_bodies_per_lane(own, cus, c->precision == 32 ? 8 : 4, jpair, &t);
  return t;
}

int nbx_detail::use_device(nbx_ctx* c) {
  HIP_TRY(hipSetDevice(c->device));
  return NBX_OK;
}
namespace {

template <typename T>
int upload_t(nbx_ctx* c, const T* px, const T* py, const T* pz, const T* vx, const T* vy, const T* vz,
             const T* m) {
  using T4 = typename V4<T>::type;
  std::vector<T4> hp((size_t)c->n_alloc);
  const T G = grav_const<T>();
  for (int i = 0; i < c->n; ++i) {
    T4 r; r.x = px[i]; r.y = py[i]; r.z = pz[i]; r.w = (G * m[i]) * gm_prescale<T>();
    hp[i] = r;
  }
  for (int i = c->n; i < c->n_alloc; ++i) { T4 z; z.x = z.y = z.z = z.w = (T)0; hp[i] = z; }
  std::vector<T4> hv((size_t)c->own_pad);
  for (int k = 0; k < c->own_pad; ++k) {
    T4 r; r.x = r.y = r.z = r.w = (T)0;
    if (k < c->i_count) { const int i = c->i_begin + k; r.x = vx[i]; r.y = vy[i]; r.z = vz[i]; r.w = m[i]; }
    hv[k] = r;
  }
  HIP_TRY(hipMemcpyAsync(c->posm[0], hp.data(), sizeof(T4) * hp.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->posm[1], hp.data(), sizeof(T4) * hp.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->velm, hv.data(), sizeof(T4) * hv.size(), hipMemcpyHostToDevice, c->stream));
  if (c->mass_all) HIP_TRY(hipMemcpyAsync(c->mass_all, m, sizeof(T) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NBX_OK;
}

template <typename T>
int download_t(nbx_ctx* c, T* px, T* py, T* pz, T* vx, T* vy, T* vz) {
  using T4 = typename V4<T>::type;
  if (px || py || pz) {
    std::vector<T4> hp((size_t)c->n);
    HIP_TRY(hipMemcpyAsync(hp.data(), c->posm[c->cur], sizeof(T4) * hp.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->n; ++i) {
      if (px) px[i] = hp[i].x;
      if (py) py[i] = hp[i].y;
      if (pz) pz[i] = hp[i].z;
    }
  }
  if (vx || vy || vz) {
    std::vector<T4> hv((size_t)c->i_count);
    HIP_TRY(hipMemcpyAsync(hv.data(), c->velm, sizeof(T4) * hv.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->i_count; ++k) {
      const int i = c->i_begin + k;
      if (vx) vx[i] = hv[k].x;
      if (vy) vy[i] = hv[k].y;
      if (vz) vz[i] = hv[k].z;
    }
  }
  return NBX_OK;
}

template <typename T>
int accel_t(nbx_ctx* c, T* ax, T* ay, T* az) {
  using T4 = typename V4<T>::type;
  int rc = enqueue_force<T>(c, EPI_SLAB, 0.0);
  if (rc) return rc;
  std::vector<T4> h((size_t)c->S * c->own_pad);
  HIP_TRY(hipMemcpyAsync(h.data(), c->accp, sizeof(T4) * h.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int k = 0; k < c->i_count; ++k) {
    T sx = (T)0, sy = (T)0, sz = (T)0;
    for (int s = 0; s < c->S; ++s) {  // same order as integrate_kernel
      const T4& q = h[(size_t)s * c->own_pad + k];
      sx += q.x; sy += q.y; sz += q.z;
    }
    const int i = c->i_begin + k;
    if (ax) ax[i] = sx;
    if (ay) ay[i] = sy;
    if (az) az[i] = sz;
  }
  return NBX_OK;
}

}  // namespace

extern "C" {

const char* nbx_last_error(void) { return last_error().c_str(); }
int32_t nbx_abi_version(void) { return NBX_ABI_VERSION; }

int nbx_create(nbx_ctx** out, int32_t n, int32_t precision, const nbx_opts* opts) {
  return guarded("nbx_create", [&]() -> int {
  if (!out) return fail(NBX_ERR_ARG, "nbx_create: out is NULL");
  *out = nullptr;
  if (n <= 0) return fail(NBX_ERR_ARG, "nbx_create: n must be > 0");
  if (precision != 32 && precision != 64) return fail(NBX_ERR_ARG, "nbx_create: precision must be 32 or 64");
  nbx_opts o;
  std::memset(&o, 0, sizeof(o));
  o.device = -1;
  if (opts) {
    if (opts->struct_size != 0 && opts->struct_size != (int32_t)sizeof(nbx_opts))
      return fail(NBX_ERR_ARG, "nbx_create: nbx_opts.struct_size does not match this library");
    o = *opts;
  }
  if (o.i_begin < 0 || o.i_count < 0 || o.i_begin >= n || (long long)o.i_begin + o.i_count > n)
    return fail(NBX_ERR_ARG, "nbx_create: slice [i_begin, i_begin+i_count) is outside [0, n)");
  if (o.n_alloc != 0 && o.n_alloc < n) return fail(NBX_ERR_ARG, "nbx_create: n_alloc < n");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NBX_ERR_DEVICE, "nbx_create: no HIP device available (libnbx has no CPU path)");
  int dev = o.device;
  if (dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) return fail(NBX_ERR_DEVICE, "nbx_create: hipGetDevice failed");
  }
  if (dev >= ndev) return fail(NBX_ERR_ARG, "nbx_create: device ordinal out of range");

  nbx_ctx* c = new (std::nothrow) nbx_ctx();
  if (!c) return fail(NBX_ERR_ALLOC, "nbx_create: out of host memory");
  struct Owner { nbx_ctx* c; ~Owner() { nbx_destroy(c); } } owner{c};  // every failure path below frees the context
  c->device = dev;
  c->n = n;
  c->precision = precision;
  c->rec = precision == 32 ? sizeof(float4) : sizeof(double4);
  c->i_begin = o.i_begin;
  c->i_count = o.i_count == 0 ? n - o.i_begin : o.i_count;
  c->n_alloc = round_up(std::max(n, o.n_alloc), kTile);
  c->own_pad = round_up(c->i_count, kBlock);

#define CREATE_TRY(expr)                                                                 \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      std::string m_ = std::string("nbx_create: " #expr ": ") + hipGetErrorString(e_);   \
      return fail(e_ == hipErrorOutOfMemory ? NBX_ERR_ALLOC : NBX_ERR_DEVICE, m_);       \
    }                                                                                    \
  } while (0)

  CREATE_TRY(hipSetDevice(dev));
  CREATE_TRY(hipGetDeviceProperties(&c->prop, dev));
  if (o.external_stream) {
    c->stream = (hipStream_t)o.stream;  // may be NULL: the default stream
  } else {
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  auto_shape(c, o);
  if (o.inner_loop != NBX_LOOP_AUTO && o.inner_loop != NBX_LOOP_CXX && o.inner_loop != NBX_LOOP_ASM && o.inner_loop != NBX_LOOP_ASM_TS &&
      o.inner_loop != NBX_LOOP_ASM_PF)
    return fail(NBX_ERR_ARG, "nbx_create: inner_loop must be NBX_LOOP_AUTO, NBX_LOOP_CXX, NBX_LOOP_ASM, NBX_LOOP_ASM_TS or NBX_LOOP_ASM_PF");
  c->loop = (o.inner_loop != NBX_LOOP_CXX && asm_loop_available(c, c->epi)) ? LOOP_ASM : LOOP_CXX;
  if ((o.inner_loop == NBX_LOOP_ASM || o.inner_loop == NBX_LOOP_ASM_TS || o.inner_loop == NBX_LOOP_ASM_PF) && c->loop != LOOP_ASM)
    return fail(NBX_ERR_ARG, "nbx_create: no hand-scheduled loop for this shape (needs fp32; kernel_variant SGPR with 1, 2 or 4 bodies per lane, SGPRW with j_per_split a multiple of 256 and 2 or 4 bodies per lane, or JLANE with 2, 4 or 8 bodies per wave)");
  // Time-sliced wave priority (LOOP_ASM_TS) exists for the row-epilogue SGPR kernel: one workgroup row, every wave resident
  // from the first cycle to the last.  Auto takes it when the fullest CU holds exactly two workgroups, i.e. two waves per
  // SIMD: measured +4.5 % at 512 workgroups, +2.7 % at 384, -0.6 % with one wave per SIMD (nobody to alternate with, six
  // more scalar instructions per trip) and -0.6 ... +0.3 % with three, four or eight (profiles/r02_time_sliced_ab.txt).
  {
    const bool ts_shape = c->loop == LOOP_ASM && c->variant == NBX_KERNEL_SGPR && c->epi == EPI_ROW && c->B >= 2;
    if (o.inner_loop == NBX_LOOP_ASM_TS && !ts_shape)
      return fail(NBX_ERR_ARG, "nbx_create: NBX_LOOP_ASM_TS needs the single-row SGPR kernel (reference summation order or j_split 1, fp32, 2 or 4 bodies per lane)");
    const int cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    const bool two_per_simd = (int)c->grid.x > cus && (int)c->grid.x <= 2 * cus;
    if (ts_shape && (o.inner_loop == NBX_LOOP_ASM_TS || (o.inner_loop == NBX_LOOP_AUTO && two_per_simd))) c->loop = LOOP_ASM_TS;
    // L2 prefetch (LOOP_ASM_PF): the same kernels when the launch leaves one wave per SIMD -- a rank that owns 131072 of 1M bodies:
    // +3.5 %; with two or more waves per SIMD the other waves are the cover and it costs 0.4-1.4 % (profiles/r04_b2_prefetch_ab.txt)
    if (o.inner_loop == NBX_LOOP_ASM_PF && !ts_shape)
      return fail(NBX_ERR_ARG, "nbx_create: NBX_LOOP_ASM_PF needs the single-row SGPR kernel (reference summation order or j_split 1, fp32, 2 or 4 bodies per lane)");
    if (ts_shape && (o.inner_loop == NBX_LOOP_ASM_PF || (o.inner_loop == NBX_LOOP_AUTO && (int)c->grid.x <= cus))) c->loop = LOOP_ASM_PF;
  }
  c->slice_bit = kSliceBit;
  if (const char* e = getenv("NBX_SLICE_BIT")) {  // experiments: log2 of the slice length in 10 ns units
    const int k = atoi(e);
    if (k >= 4 && k <= 30) c->slice_bit = 1u << k;
  }
  // The jlane kernel's generated loop keeps four records per set in flight; with few bodies per wave that is too little
  // arithmetic to cover an L2 round trip when a SIMD holds a single wave, and the compiled loop (eight records per set) is
  // 3-4 % ahead there (profiles/r02_jlane_ab.txt).  Auto takes the generated loop where it measured faster: 8 bodies per wave,
  // or 4 with more than one wave per SIMD.
  if (o.inner_loop == NBX_LOOP_AUTO && c->variant == NBX_KERNEL_JLANE && c->loop == LOOP_ASM) {
    const int cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    const bool several_waves = ceil_div(c->i_count, c->B) > cus * 4;
    if (!(c->B == 8 || (c->B == 4 && several_waves))) c->loop = LOOP_CXX;
  }
  // use_graph: 0 auto (launch-bound sizes only: < ~0.3 ms of pair work per step), 1 on, 2 off;
  // capture needs a stream of our own
  {
    const double pairs = (double)c->i_count * (double)c->n;
    c->use_graph = c->own_stream && (o.use_graph == 1 || (o.use_graph == 0 && pairs < 1.5e9));
  }

  // + spare records: the pipelined SGPR loop requests one batch past the last split (never used)
  const size_t pos_bytes = c->rec * (size_t)(c->n_alloc + kSgprOverread);
  CREATE_TRY(hipMalloc(&c->posm[0], pos_bytes));
  CREATE_TRY(hipMalloc(&c->posm[1], pos_bytes));
  CREATE_TRY(hipMalloc(&c->velm, c->rec * (size_t)c->own_pad));
  CREATE_TRY(hipMalloc(&c->accp, c->rec * (size_t)c->own_pad * c->S));
  const int max_parts = std::max(ceil_div(c->i_count, kBlock), (int)c->grid.x);
  CREATE_TRY(hipMalloc(&c->ke_part, sizeof(double) * (size_t)max_parts));
  if (c->variant == NBX_KERNEL_EXACT || c->variant == NBX_KERNEL_EXACT_FMA) CREATE_TRY(hipMalloc(&c->mass_all, (c->rec / 4) * (size_t)c->n_alloc));
  if (c->loop == LOOP_ASM && jpair_shape(c)) {  // same size and the same zero-filled spare records as posm
    CREATE_TRY(hipMalloc(&c->posm_pairs, pos_bytes));
    CREATE_TRY(hipMemsetAsync(c->posm_pairs, 0, pos_bytes, c->stream));
  }
  CREATE_TRY(hipMemsetAsync(c->posm[0], 0, pos_bytes, c->stream));
  CREATE_TRY(hipMemsetAsync(c->posm[1], 0, pos_bytes, c->stream));
  CREATE_TRY(hipMemsetAsync(c->ke_part, 0, sizeof(double) * (size_t)max_parts, c->stream));
  CREATE_TRY(hipStreamSynchronize(c->stream));
#undef CREATE_TRY
  if (ensure_ke_cap(c, 64) != NBX_OK) return NBX_ERR_ALLOC;  // message set by ensure_ke_cap
  owner.c = nullptr;
  *out = c;
  last_error().clear();
  return NBX_OK;
  });
}

void nbx_destroy(nbx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);  // NULL = the default stream when the caller lent us that one
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
  if (c->posm[0]) (void)hipFree(c->posm[0]);
  if (c->posm[1]) (void)hipFree(c->posm[1]);
  if (c->velm) (void)hipFree(c->velm);
  if (c->accp) (void)hipFree(c->accp);
  if (c->ke_part) (void)hipFree(c->ke_part);
  if (c->mass_all) (void)hipFree(c->mass_all);
  if (c->posm_pairs) (void)hipFree(c->posm_pairs);
  if (c->ke_dev) (void)hipFree(c->ke_dev);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int nbx_upload(nbx_ctx* c, const void* px, const void* py, const void* pz, const void* vx, const void* vy,
               const void* vz, const void* m) {
  return guarded("nbx_upload", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_upload: ctx is NULL");
  if (!px || !py || !pz || !vx || !vy || !vz || !m) return fail(NBX_ERR_ARG, "nbx_upload: NULL array");
  int rc = use_device(c);
  if (rc) return rc;
  rc = c->precision == 32
           ? upload_t<float>(c, (const float*)px, (const float*)py, (const float*)pz, (const float*)vx,
                             (const float*)vy, (const float*)vz, (const float*)m)
           : upload_t<double>(c, (const double*)px, (const double*)py, (const double*)pz, (const double*)vx,
                              (const double*)vy, (const double*)vz, (const double*)m);
  if (rc) return rc;
  c->cur = 0;
  c->uploaded = true;
  c->pending_commit = false;
  c->ke_parts = 0;  // the partials on the device belong to the previous trajectory
  return NBX_OK;
  });
}

static int step_common(nbx_ctx* c, double dt, int32_t nsteps, double* ke_last, double* ke_trace) {
  return guarded("nbx_step", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_step: ctx is NULL");
  if (nsteps < 0) return fail(NBX_ERR_ARG, "nbx_step: nsteps < 0");
  if (!c->uploaded) return fail(NBX_ERR_STATE, "nbx_step: nbx_upload has not been called");
  if (c->i_begin != 0 || c->i_count != c->n)
    return fail(NBX_ERR_STATE, "nbx_step: context owns a slice; use nbx_step_local + exchange + nbx_commit");
  if (c->pending_commit) return fail(NBX_ERR_STATE, "nbx_step: a local step awaits nbx_commit");
  int rc = use_device(c);
  if (rc) return rc;
  if (ke_trace) {
    rc = ensure_ke_cap(c, std::max(nsteps, 1));
    if (rc) return rc;
  }
  int first = 0;
  if (c->use_graph && !ke_trace && !c->profiling && nsteps >= 4) {
    const int unit = std::min(nsteps & ~1, 50);  // one replay costs the host 10-16 us: 50 steps per replay keeps that under 0.3 us per step
    hipGraphExec_t exec = nullptr;
    rc = graph_unit_exec(c, unit, dt, &exec);
    if (rc) return rc;
    while (nsteps - first >= unit) {  // an even unit leaves the buffer parity unchanged
      HIP_TRY(hipGraphLaunch(exec, c->stream));
      first += unit;
      c->steps_done += unit;
      c->graph_replays += 1;
      // a replay enqueues through the captured nodes, not through enqueue_step: say here how many energy partials its
      // last step leaves behind (nbx_upload zeroes the count; a cached graph must not leave it at zero)
      c->ke_parts = step_ke_parts(c);
    }
    if (first == nsteps && ke_last) {  // the partials of the last captured step are in ke_part
      rc = enqueue_ke_reduce(c, 0);
      if (rc) return rc;
    }
  }
  for (int s = first; s < nsteps; ++s) {
    rc = enqueue_step_any(c, dt);
    if (rc) return rc;
    c->cur ^= 1;
    c->steps_done += 1;
    if (ke_trace) {
      rc = enqueue_ke_reduce(c, s);
      if (rc) return rc;
    } else if (ke_last && s == nsteps - 1) {
      rc = enqueue_ke_reduce(c, 0);
      if (rc) return rc;
    }
  }
  if (ke_trace && nsteps > 0) {
    HIP_TRY(hipMemcpyAsync(ke_trace, c->ke_dev, sizeof(double) * (size_t)nsteps, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int s = 0; s < nsteps; ++s) ke_trace[s] *= 0.5;  // ver7/GSimulation.cpp:200
  } else if (ke_last) {
    if (nsteps > 0 || c->ke_parts > 0) {
      if (nsteps == 0) {
        rc = enqueue_ke_reduce(c, 0);
        if (rc) return rc;
      }
      double sum = 0.0;
      HIP_TRY(hipMemcpyAsync(&sum, c->ke_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      *ke_last = 0.5 * sum;
    } else {
      *ke_last = 0.0;
    }
  }
  return NBX_OK;
  });
}

int nbx_step(nbx_ctx* c, double dt, int32_t nsteps, double* kenergy_out) {
  return step_common(c, dt, nsteps, kenergy_out, nullptr);
}

int nbx_step_trace(nbx_ctx* c, double dt, int32_t nsteps, double* ke_trace) {
  if (!ke_trace) return guarded("nbx_step_trace", [&]() -> int { return fail(NBX_ERR_ARG, "nbx_step_trace: ke_trace is NULL"); });
  return step_common(c, dt, nsteps, nullptr, ke_trace);
}

int nbx_step_local(nbx_ctx* c, double dt) {
  return guarded("nbx_step_local", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_step_local: ctx is NULL");
  if (!c->uploaded) return fail(NBX_ERR_STATE, "nbx_step_local: nbx_upload has not been called");
  if (c->pending_commit) return fail(NBX_ERR_STATE, "nbx_step_local: previous step not committed");
  int rc = use_device(c);
  if (rc) return rc;
  rc = enqueue_step_any(c, dt);
  if (rc) return rc;
  c->pending_commit = true;
  return NBX_OK;
  });
}

int nbx_exchange_buffer(nbx_ctx* c, void** dev_ptr, size_t* total_bytes, size_t* own_offset_bytes, size_t* own_bytes) {
  return guarded("nbx_exchange_buffer", [&]() -> int {
  if (!c || !dev_ptr) return fail(NBX_ERR_ARG, "nbx_exchange_buffer: NULL argument");
  // the buffer the last local step wrote (NEXT while a commit is pending, else current)
  *dev_ptr = c->posm[c->pending_commit ? (c->cur ^ 1) : c->cur];
  if (total_bytes) *total_bytes = c->rec * (size_t)c->n_alloc;
  if (own_offset_bytes) *own_offset_bytes = c->rec * (size_t)c->i_begin;
  if (own_bytes) *own_bytes = c->rec * (size_t)c->i_count;
  return NBX_OK;
  });
}

int nbx_commit(nbx_ctx* c) {
  return guarded("nbx_commit", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_commit: ctx is NULL");
  if (!c->pending_commit) return fail(NBX_ERR_STATE, "nbx_commit: no local step pending");
  c->cur ^= 1;
  c->pending_commit = false;
  c->steps_done += 1;
  return NBX_OK;
  });
}

int nbx_kenergy_partial(nbx_ctx* c, double* sum_mv2) {
  return guarded("nbx_kenergy_partial", [&]() -> int {
  if (!c || !sum_mv2) return fail(NBX_ERR_ARG, "nbx_kenergy_partial: NULL argument");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->ke_parts <= 0) {
    *sum_mv2 = 0.0;
    return NBX_OK;
  }
  rc = enqueue_ke_reduce(c, 0);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(sum_mv2, c->ke_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NBX_OK;
  });
}

int nbx_accel(nbx_ctx* c, void* ax, void* ay, void* az) {
  return guarded("nbx_accel", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_accel: ctx is NULL");
  if (!c->uploaded) return fail(NBX_ERR_STATE, "nbx_accel: nbx_upload has not been called");
  if (c->pending_commit) return fail(NBX_ERR_STATE, "nbx_accel: a local step awaits nbx_commit");
  int rc = use_device(c);
  if (rc) return rc;
  return c->precision == 32 ? accel_t<float>(c, (float*)ax, (float*)ay, (float*)az)
                            : accel_t<double>(c, (double*)ax, (double*)ay, (double*)az);
  });
}

int nbx_sync(nbx_ctx* c) {
  return guarded("nbx_sync", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_sync: ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NBX_OK;
  });
}

int nbx_download(nbx_ctx* c, void* px, void* py, void* pz, void* vx, void* vy, void* vz) {
  return guarded("nbx_download", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_download: ctx is NULL");
  if (!c->uploaded) return fail(NBX_ERR_STATE, "nbx_download: nbx_upload has not been called");
  int rc = use_device(c);
  if (rc) return rc;
  return c->precision == 32
             ? download_t<float>(c, (float*)px, (float*)py, (float*)pz, (float*)vx, (float*)vy, (float*)vz)
             : download_t<double>(c, (double*)px, (double*)py, (double*)pz, (double*)vx, (double*)vy, (double*)vz);
  });
}

int nbx_profile(nbx_ctx* c, int32_t enable) {
  return guarded("nbx_profile", [&]() -> int {
  if (!c) return fail(NBX_ERR_ARG, "nbx_profile: ctx is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  if (enable && c->ev.empty()) {
    c->ev.resize(2 * kMaxProfiledLaunches);
    for (auto& e : c->ev) e = nullptr;
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
  }
  if (!enable && c->profiling) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = drain_profile(c);
    if (rc) return rc;
  }
  if (enable && !c->profiling) {
    c->force_ms_total = 0.0;
    c->force_timed = 0;
    c->ev_used = 0;
  }
  c->profiling = enable != 0;
  return NBX_OK;
  });
}

int nbx_stats(nbx_ctx* c, nbx_stats_t* s) {
  return guarded("nbx_stats", [&]() -> int {
  if (!c || !s) return fail(NBX_ERR_ARG, "nbx_stats: NULL argument");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->ev_used) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = drain_profile(c);
    if (rc) return rc;
  }
  std::memset(s, 0, sizeof(*s));
  s->n = c->n; s->n_alloc = c->n_alloc; s->i_begin = c->i_begin; s->i_count = c->i_count;
  s->precision = c->precision; s->bodies_per_lane = c->B; s->j_split = c->S; s->j_tile = kTile;
  s->kernel_variant = c->variant; s->fused_epilogue = c->epi; s->summation_order = c->order;
  s->force_grid_x = c->grid.x; s->force_grid_y = c->grid.y; s->force_block = kBlock;
  s->cu_count = c->prop.multiProcessorCount; s->clock_mhz = c->prop.clockRate / 1000;
  s->steps_done = c->steps_done;
  s->force_launches_timed = c->force_timed;
  s->force_ms_total = c->force_ms_total;
  s->pairs_per_launch = (double)c->i_count * (double)c->n;
  s->graph_replays = c->graph_replays;
  s->use_graph = c->use_graph ? 1 : 0;
  s->inner_loop = c->loop == LOOP_ASM_TS ? NBX_LOOP_ASM_TS : c->loop == LOOP_ASM_PF ? NBX_LOOP_ASM_PF : c->loop == LOOP_ASM ? NBX_LOOP_ASM : NBX_LOOP_CXX;
  // some boxes report an empty marketing name; fall back to / append the ISA name
  std::snprintf(s->device_name, sizeof(s->device_name), "%s%s%s", c->prop.name, c->prop.name[0] ? " " : "",
                c->prop.gcnArchName);
  return NBX_OK;
  });
}

}  // extern "C"

static_assert(nbx::kTile == 256 && nbx::kBlock == 256, "nbx_group.hip aligns blocks to the kernels' 256-record j tile");
