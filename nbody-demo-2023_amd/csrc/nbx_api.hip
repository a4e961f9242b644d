// nbx_api.hip -- the C-ABI of include/nbx.h over the gfx950 kernels of nbx_kernels.hpp.
//
// One context = one GPU's share of the reference's GSimulation::start() loop
// (ver7/GSimulation.cpp:138-200): it owns bodies [i_begin, i_begin+i_count), keeps
// {x,y,z,G*m} of ALL bodies resident (double buffered) and {vx,vy,vz,m} of its own.
// No CPU fallback exists: without a HIP device every entry point fails with NBX_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is dlopen'ed (struct Rccl), never linked

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/nbx.h"
#include "nbx_kernels.hpp"

using namespace nbx;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(NBX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

// No C++ exception may cross the C boundary: every entry point that can allocate host memory runs inside this.
template <typename F>
int guarded(const char* where, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    try { return fail(NBX_ERR_ALLOC, std::string(where) + ": out of host memory"); } catch (...) { return NBX_ERR_ALLOC; }
  } catch (const std::exception& e) {
    try { return fail(NBX_ERR_STATE, std::string(where) + ": " + e.what()); } catch (...) { return NBX_ERR_STATE; }
  } catch (...) {
    return NBX_ERR_STATE;
  }
}

constexpr int kMaxProfiledLaunches = 8192;
// NBX_ORDER_AUTO: fp32 sums of more terms than this use the reference's order.  131072 x 500 steps agrees with the
// reference to 2e-5 in tree order (profiles/r01_validate_orders_n131072_s500.log); 262144 x 200 does not (1.3e-3).
constexpr int kTreeOrderMaxN = 131072;
// NBX_KERNEL_AUTO, tree order: contexts that own at most this many bodies step with ONE launch (force_jlane_kernel)
// (12288: 41 us against SGPRW's 48; at 16384 the two tie at 67 us, and SGPRW's summation tree happens to be the one whose
// chaotic n = 16384 x 500 run stays inside the 1e-4 gate at every printed step: profiles/r02_config1_by_kernel.txt)
constexpr int kJlaneMaxOwn = 12288;

}  // namespace

struct nbx_ctx {
  int n = 0, n_alloc = 0, i_begin = 0, i_count = 0, own_pad = 0, precision = 32;
  int B = 1, S = 1, jps = 0, variant = NBX_KERNEL_LDS, epi = EPI_SLAB, math = MATH_SCALAR, order = NBX_ORDER_TREE;
  int loop = LOOP_CXX;  // LOOP_ASM where the hand-scheduled j loop is in use
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  size_t rec = 16;  // bytes per {x,y,z,w} record
  void* posm[2] = {nullptr, nullptr};
  int cur = 0;
  void* velm = nullptr;
  void* accp = nullptr;
  double* ke_part = nullptr;
  void* mass_all = nullptr;        // NBX_KERNEL_EXACT only: m of every body (the records carry G*m)
  int ke_parts = 0;       // partials written by the last step
  double* ke_dev = nullptr;  // [ke_cap] reduced sums (sum m v^2)
  int ke_cap = 0;
  bool uploaded = false;
  bool pending_commit = false;
  long long steps_done = 0;
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev;  // pairs start/stop
  size_t ev_used = 0;
  double force_ms_total = 0.0;
  long long force_timed = 0;
  hipDeviceProp_t prop{};
  dim3 grid;
  // hipGraph replay of multi-step windows (launch-bound small n)
  bool use_graph = false;
  struct GraphUnit { int steps; int parity; double dt; hipGraphExec_t exec; };
  std::vector<GraphUnit> graphs;
  long long graph_replays = 0;
};

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
constexpr bool kHasAsmLoop = sizeof(T) == 4 && JSRC == JSRC_SGPR && MATH == MATH_PACKED && !WS;

template <typename T>
using ForceLauncher = void (*)(const ForceArgs<T>&, dim3, hipStream_t);

template <typename T, int JSRC, int EPI, int MATH, bool WS>
ForceLauncher<T> pick_b(int B, int loop) {
  if constexpr (kHasAsmLoop<T, JSRC, EPI, MATH, WS>) {
    if (loop == LOOP_ASM) {
      if (B == 2) return launch_force_t<T, 2, JSRC, EPI, MATH, WS, LOOP_ASM>;
      if (B == 4) return launch_force_t<T, 4, JSRC, EPI, MATH, WS, LOOP_ASM>;
      return nullptr;
    }
  } else if (loop == LOOP_ASM) {
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

// Does the hand-scheduled loop exist for this shape?  (mirror of kHasAsmLoop for run-time shape decisions)
bool asm_loop_available(const nbx_ctx* c, int epi) {
  (void)epi;
  return c->precision == 32 && c->variant == NBX_KERNEL_SGPR && c->math == MATH_PACKED && (c->B == 2 || c->B == 4);
}

template <typename T>
ForceLauncher<T> pick_force(const nbx_ctx* c, int epi) {
  // nbx_accel runs the EPI_SLAB form of a context whose step kernel may be another epilogue: decide per call
  const int loop = (c->loop == LOOP_ASM && asm_loop_available(c, epi)) ? LOOP_ASM : LOOP_CXX;
  if constexpr (sizeof(T) == 4) {
    if (c->math == MATH_PACKED) return pick<float, MATH_PACKED>(c->B, c->variant, epi, loop);
  }
  return pick<T, MATH_SCALAR>(c->B, c->variant, epi, loop);
}

int ceil_div(int a, int b) { return (a + b - 1) / b; }
int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Bodies per lane of the reference-order kernel (one chain per owned body, S = 1).  Its run time is quantised: the
// ceil(own / (256 B)) workgroups are spread over the CUs, and a launch takes as long as the fullest CU, which holds
// r = ceil(workgroups / CUs) of them.  Measured on MI355X at n = 1048576 with the hand-scheduled loop for B = 2 and 4
// (profiles/r02_reference_order_thresholds.txt), ms for r = 1, 2, 3, ...: B = 1: 31.0, 48.6, 70.3, 91, 112 (plain VALU ops);
// B = 2: 31.5, 59.6, 88.5, 118;  B = 4: 59.3, 117.4, 175.6, 234.6 -- linear in r after the first workgroup.  Pick the B with the smallest
// estimate; ties go to the larger B (fewer workgroups stream the j records).  Only the ratios matter, so the table
// serves every n.
int reference_order_bodies_per_lane(int own, int cus, int max_b) {
  static const struct { int b; double first, next; } kCost[] = {{1, 31.0, 20.2}, {2, 31.5, 28.8}, {4, 59.8, 58.2}};
  int best = 1;
  double best_t = 0.0;
  for (const auto& k : kCost) {
    if (k.b > max_b) continue;
    const int wgs = ceil_div(own, kBlock * k.b);
    const int r = std::max(1, ceil_div(wgs, std::max(1, cus)));
    const double t = k.first + k.next * (r - 1);
    if (best_t == 0.0 || t <= best_t * 1.01) { best = k.b; best_t = std::min(t, best_t == 0.0 ? t : best_t); }
  }
  return best;
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
    if (B == 0) B = reference_order_bodies_per_lane(c->i_count, cus, maxBr);
    c->B = B; c->S = 1; c->jps = c->n_alloc; c->variant = variant;
    c->math = (c->precision == 32 && B >= 2) ? MATH_PACKED : MATH_SCALAR;
    c->epi = o.fused_epilogue == 2 ? EPI_SLAB : EPI_ROW;
    c->grid = dim3(ceil_div(c->i_count, kBlock * B), 1);
    return;
  }
  // Launch-bound sizes (fp32): one launch per step with the lanes of a wave splitting j (force_jlane_kernel).  Bodies per
  // wave: the power of two that gives about one wave per SIMD (1024 waves), between 2 and 16.
  const bool jlane_auto = variant == NBX_KERNEL_AUTO && c->precision == 32 && o.j_split <= 0 && o.bodies_per_lane == 0 &&
                          o.fused_epilogue != 2 && c->i_count <= kJlaneMaxOwn;
  if ((variant == NBX_KERNEL_JLANE && c->precision == 32) || jlane_auto) {
    int NB = o.bodies_per_lane;
    if (NB != 2 && NB != 4 && NB != 8 && NB != 16) {
      // A launch lasts as long as the fullest SIMD: ceil(waves / SIMDs) rounds of NB bodies each.  Fewest body-rounds wins;
      // ties go to the larger NB (fewer waves stream the j records, fewer LDS transposes) -- the measured optimum at every
      // size from 2048 to 32768 (profiles/r02_jlane_ab.txt: 2048 -> 2, 4096 -> 4, 8192 -> 8, 12288 -> 4, 16384 -> 16).
      long best = 0;
      for (int nb = 2; nb <= 16; nb *= 2) {
        const long cost = (long)ceil_div(ceil_div(c->i_count, nb), cus * 4) * nb;
        if (best == 0 || cost <= best) { best = cost; NB = nb; }
      }
    }
    c->B = NB; c->S = 1; c->jps = c->n_alloc; c->math = MATH_PACKED; c->variant = NBX_KERNEL_JLANE; c->epi = EPI_ROW;
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
  const int gran = variant == NBX_KERNEL_LDS ? kTile : (variant == NBX_KERNEL_SGPR ? 64 : 32);
  static_assert(64 % kSgprAsmTrip<2> == 0 && 64 % kSgprAsmTrip<4> == 0 && kTile % 64 == 0, "j ranges are whole trips of the asm loop");
  const int max_split = std::max(1, c->n_alloc / gran);
  int S = o.j_split;
  if (S <= 0) {
    const int bi = ceil_div(c->i_count, iblk);
    S = std::min(32, ceil_div(target_wgs, bi));
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
  const bool prof = c->profiling && c->ev_used + 2 <= c->ev.size();
  if (prof) HIP_TRY(hipEventRecord(c->ev[c->ev_used], c->stream));
  if (c->variant == NBX_KERNEL_JLANE) {
    if constexpr (sizeof(T) == 4) {
      const int acc_only = epi == EPI_SLAB ? 1 : 0;  // nbx_accel asks for the slab form: accelerations only
      switch (c->B) {  // prefetch depth: enough records in flight to cover an L2 round trip with NB/2 x 56 cycles of work each
        case 2: hipLaunchKernelGGL((force_jlane_kernel<2, 8>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
        case 4: hipLaunchKernelGGL((force_jlane_kernel<4, 8>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
        case 8: hipLaunchKernelGGL((force_jlane_kernel<8, 4>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
        default: hipLaunchKernelGGL((force_jlane_kernel<16, 4>), c->grid, dim3(kBlock), 0, c->stream, a, acc_only); break;
      }
    } else {
      return fail(NBX_ERR_ARG, "NBX_KERNEL_JLANE exists in fp32 only");
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

// one local step: force (+ integrate) into the next buffer; does not swap
template <typename T>
int enqueue_step(nbx_ctx* c, double dt) {
  using T4 = typename V4<T>::type;
  int rc = enqueue_force<T>(c, c->epi, dt);
  if (rc) return rc;
  if (c->epi != EPI_SLAB) {
    c->ke_parts = c->grid.x;
  } else {
    const int blocks = ceil_div(c->i_count, kBlock);
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

int enqueue_ke_reduce(nbx_ctx* c, int slot) {
  hipLaunchKernelGGL(ke_reduce_kernel, dim3(1), dim3(kBlock), 0, c->stream, (const double*)c->ke_part,
                     c->ke_parts, c->ke_dev + slot);
  HIP_TRY(hipGetLastError());
  return NBX_OK;
}

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

int use_device(nbx_ctx* c) {
  HIP_TRY(hipSetDevice(c->device));
  return NBX_OK;
}

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

const char* nbx_last_error(void) { return g_err.c_str(); }
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
  if (o.inner_loop != NBX_LOOP_AUTO && o.inner_loop != NBX_LOOP_CXX && o.inner_loop != NBX_LOOP_ASM)
    return fail(NBX_ERR_ARG, "nbx_create: inner_loop must be NBX_LOOP_AUTO, NBX_LOOP_CXX or NBX_LOOP_ASM");
  c->loop = (o.inner_loop != NBX_LOOP_CXX && asm_loop_available(c, c->epi)) ? LOOP_ASM : LOOP_CXX;
  if (o.inner_loop == NBX_LOOP_ASM && c->loop != LOOP_ASM)
    return fail(NBX_ERR_ARG, "nbx_create: no hand-scheduled loop for this shape (needs fp32, kernel_variant SGPR, 2 or 4 bodies per lane)");
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
  CREATE_TRY(hipMemsetAsync(c->posm[0], 0, pos_bytes, c->stream));
  CREATE_TRY(hipMemsetAsync(c->posm[1], 0, pos_bytes, c->stream));
  CREATE_TRY(hipMemsetAsync(c->ke_part, 0, sizeof(double) * (size_t)max_parts, c->stream));
  CREATE_TRY(hipStreamSynchronize(c->stream));
#undef CREATE_TRY
  if (ensure_ke_cap(c, 64) != NBX_OK) return NBX_ERR_ALLOC;  // message set by ensure_ke_cap
  owner.c = nullptr;
  *out = c;
  g_err.clear();
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
    const int unit = std::min(nsteps & ~1, 20);
    hipGraphExec_t exec = nullptr;
    rc = graph_unit_exec(c, unit, dt, &exec);
    if (rc) return rc;
    while (nsteps - first >= unit) {  // an even unit leaves the buffer parity unchanged
      HIP_TRY(hipGraphLaunch(exec, c->stream));
      first += unit;
      c->steps_done += unit;
      c->graph_replays += 1;
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
  s->inner_loop = c->loop == LOOP_ASM ? NBX_LOOP_ASM : NBX_LOOP_CXX;
  // some boxes report an empty marketing name; fall back to / append the ISA name
  std::snprintf(s->device_name, sizeof(s->device_name), "%s%s%s", c->prop.name, c->prop.name[0] ? " " : "",
                c->prop.gcnArchName);
  return NBX_OK;
  });
}

}  // extern "C"

// =============================================================================================
// nbx_group: single-process multi-GPU driver (see include/nbx.h)
// =============================================================================================
namespace {

// The RCCL entry points used, resolved at run time so libnbx.so has no link-time dependency on librccl (and binds to
// the copy already in the process when a host such as PyTorch brought its own).  Every pointer takes its type from
// rccl.h's own prototype (decltype), so a signature or enum change in the header is a compile error here, not a
// silent ABI mismatch at the first multi-GPU run.
struct Rccl {
  typedef ncclComm_t comm_t;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  template <typename F> static void sym(void* h, const char* name, F& fn) { fn = reinterpret_cast<F>(dlsym(h, name)); }
  bool load() {
    if (ok) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    sym(h, "ncclCommInitAll", CommInitAll);
    sym(h, "ncclCommInitRank", CommInitRank);
    sym(h, "ncclGetUniqueId", GetUniqueId);
    sym(h, "ncclCommDestroy", CommDestroy);
    sym(h, "ncclGroupStart", GroupStart);
    sym(h, "ncclGroupEnd", GroupEnd);
    sym(h, "ncclAllGather", AllGather);
    sym(h, "ncclGetErrorString", GetErrorString);
    ok = CommInitAll && CommInitRank && GetUniqueId && CommDestroy && GroupStart && GroupEnd && AllGather;
    return ok;
  }
  std::string text(ncclResult_t e) const { return GetErrorString ? std::string(GetErrorString(e)) : std::string("RCCL error ") + std::to_string((int)e); }
};
Rccl g_rccl;
static_assert(std::is_same<decltype(Rccl::AllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value,
              "ncclAllGather is called as (send, recv, bytes, ncclChar, comm, stream)");
static_assert(sizeof(ncclUniqueId) == NBX_UNIQUE_ID_BYTES, "include/nbx.h promises callers the size of the rendezvous token");

}  // namespace

struct nbx_group {
  int n = 0, precision = 32, P = 0, block = 0, n_alloc = 0;
  int my_rank = -1;                  // >= 0: one-process-per-GPU group (nbx_group_create_rank): `rank` holds this process's context only
  std::vector<nbx_ctx*> rank;
  std::vector<int> dev;
  std::vector<hipEvent_t> done;      // rank r's NEXT block is complete (copy path)
  std::vector<Rccl::comm_t> comm;    // RCCL path
  bool use_rccl = false;
  double* ke_all = nullptr;          // rank groups: [P] sum m v^2 of every rank, all-gathered
  void* vel_stage = nullptr;         // rank groups, nbx_group_download: own velocities padded to `block` records
  void* vel_all = nullptr;           //   and the all-gathered [P * block] records
};

namespace {

// Balanced, tile-aligned blocks; ranks that would own nothing are dropped (P is reduced).  The same arithmetic as
// sharded.block_partition (tests/test_partition.py compares them) with that reduction applied.
void partition(int n, int n_ranks, int* P_out, int* block_out) {
  int P = n_ranks, block = 0;
  for (;; --P) {
    block = round_up(ceil_div(n, P), kTile);
    if (P == 1 || (long long)(P - 1) * block < n) break;
  }
  *P_out = P;
  *block_out = block;
}

int rccl_fail(const char* what, ncclResult_t e) { return fail(NBX_ERR_DEVICE, std::string(what) + ": " + g_rccl.text(e)); }

int group_exchange(nbx_group* g) {
  const size_t rec = g->rank[0]->rec;
  if (g->use_rccl) {
    // in place: rank r sends its own block, receives every block at its natural offset.  Once the group is open every
    // path reaches ncclGroupEnd: an early return would leave RCCL in group mode for the rest of the process.
    ncclResult_t e = g_rccl.GroupStart();
    if (e != ncclSuccess) return rccl_fail("ncclGroupStart", e);
    int rc = NBX_OK;
    for (size_t k = 0; k < g->rank.size() && rc == NBX_OK; ++k) {
      nbx_ctx* c = g->rank[k];
      const int r = g->my_rank >= 0 ? g->my_rank : (int)k;
      char* buf = (char*)c->posm[c->cur ^ 1];
      const hipError_t he = hipSetDevice(g->dev[k]);
      if (he != hipSuccess) { rc = fail(NBX_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(he)); break; }
      e = g_rccl.AllGather(buf + (size_t)r * g->block * rec, buf, (size_t)g->block * rec, ncclChar, g->comm[k], c->stream);
      if (e != ncclSuccess) rc = rccl_fail("ncclAllGather", e);
    }
    e = g_rccl.GroupEnd();
    if (rc == NBX_OK && e != ncclSuccess) rc = rccl_fail("ncclGroupEnd", e);
    return rc;
  }
  // copy path: every destination pulls every other rank's block, stream-ordered behind the producer's event
  for (int r = 0; r < g->P; ++r) {
    HIP_TRY(hipSetDevice(g->dev[r]));
    HIP_TRY(hipEventRecord(g->done[r], g->rank[r]->stream));
  }
  for (int q = 0; q < g->P; ++q) {
    nbx_ctx* dst = g->rank[q];
    HIP_TRY(hipSetDevice(g->dev[q]));
    for (int r = 0; r < g->P; ++r) {
      if (r == q) continue;
      nbx_ctx* src = g->rank[r];
      const size_t off = (size_t)src->i_begin * rec, bytes = (size_t)src->i_count * rec;
      HIP_TRY(hipStreamWaitEvent(dst->stream, g->done[r], 0));
      char* d = (char*)dst->posm[dst->cur ^ 1] + off;
      const char* s = (const char*)src->posm[src->cur ^ 1] + off;
      if (g->dev[q] == g->dev[r]) HIP_TRY(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, dst->stream));
      else HIP_TRY(hipMemcpyPeerAsync(d, g->dev[q], s, g->dev[r], bytes, dst->stream));
    }
  }
  return NBX_OK;
}

int check_group_args(const char* who, nbx_group** out, int n, int precision, int n_ranks, const nbx_opts* opts, nbx_opts* o) {
  if (!out) return fail(NBX_ERR_ARG, std::string(who) + ": out is NULL");
  *out = nullptr;
  if (n <= 0) return fail(NBX_ERR_ARG, std::string(who) + ": n must be > 0");
  if (n_ranks <= 0 || n_ranks > 64) return fail(NBX_ERR_ARG, std::string(who) + ": the number of ranks must be in 1..64");
  if (precision != 32 && precision != 64) return fail(NBX_ERR_ARG, std::string(who) + ": precision must be 32 or 64");
  std::memset(o, 0, sizeof(*o));
  if (opts) {
    if (opts->struct_size != 0 && opts->struct_size != (int32_t)sizeof(nbx_opts))
      return fail(NBX_ERR_ARG, std::string(who) + ": nbx_opts.struct_size does not match this library");
    *o = *opts;
  }
  o->stream = nullptr; o->external_stream = 0; o->use_graph = 2;  // every rank: own stream, plain launches
  return NBX_OK;
}

}  // namespace

extern "C" {

int nbx_partition(int32_t n, int32_t n_ranks, int32_t rank, int32_t* ranks_used, int32_t* block, int32_t* i_begin,
                  int32_t* i_count, int32_t* n_alloc) {
  return guarded("nbx_partition", [&]() -> int {
  if (n <= 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(NBX_ERR_ARG, "nbx_partition: need n > 0 and 0 <= rank < n_ranks");
  int P = 0, b = 0;
  partition(n, n_ranks, &P, &b);
  const long long lo = std::min<long long>((long long)rank * b, n), hi = std::min<long long>((long long)(rank + 1) * b, n);
  if (ranks_used) *ranks_used = P;
  if (block) *block = b;
  if (i_begin) *i_begin = (int32_t)lo;
  if (i_count) *i_count = rank < P ? (int32_t)(hi - lo) : 0;  // ranks >= P own nothing and take no part
  if (n_alloc) *n_alloc = P * b;
  return NBX_OK;
  });
}

int nbx_group_create(nbx_group** out, int32_t n, int32_t precision, int32_t n_ranks, const int32_t* devices,
                     const nbx_opts* opts) {
  return guarded("nbx_group_create", [&]() -> int {
  nbx_opts o;
  int rc = check_group_args("nbx_group_create", out, n, precision, n_ranks, opts, &o);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NBX_ERR_DEVICE, "nbx_group_create: no HIP device available (libnbx has no CPU path)");
  int P = 0, block = 0;
  partition(n, n_ranks, &P, &block);
  nbx_group* g = new (std::nothrow) nbx_group();
  if (!g) return fail(NBX_ERR_ALLOC, "nbx_group_create: out of host memory");
  struct Owner { nbx_group* g; ~Owner() { nbx_group_destroy(g); } } owner{g};  // every failure path below frees the group
  g->n = n; g->precision = precision; g->P = P; g->block = block; g->n_alloc = P * block;
  bool distinct = true;
  for (int r = 0; r < P; ++r) {
    const int d = devices ? devices[r] : r % ndev;
    if (d < 0 || d >= ndev) return fail(NBX_ERR_ARG, "nbx_group_create: device ordinal out of range");
    for (int q : g->dev) distinct = distinct && q != d;
    g->dev.push_back(d);
  }
  for (int r = 0; r < P; ++r) {
    o.device = g->dev[r];
    o.i_begin = r * block;
    o.i_count = std::min(n, (r + 1) * block) - r * block;
    o.n_alloc = g->n_alloc;
    nbx_ctx* c = nullptr;
    rc = nbx_create(&c, n, precision, &o);
    if (rc != NBX_OK) { const std::string m = g_err; return fail(rc, "nbx_group_create: rank " + std::to_string(r) + ": " + m); }
    g->rank.push_back(c);
  }
  const char* force = std::getenv("NBX_EXCHANGE");  // "copy" forces the peer-copy path, "rccl" insists on RCCL
  const bool insist = force && !std::strcmp(force, "rccl");  // also with a single rank: smoke-tests the RCCL binding
  const bool want_rccl = distinct && (P > 1 || insist) && !(force && !std::strcmp(force, "copy"));
  if (want_rccl && g_rccl.load()) {
    g->comm.assign(P, nullptr);
    const ncclResult_t e = g_rccl.CommInitAll(g->comm.data(), P, g->dev.data());
    if (e == ncclSuccess) g->use_rccl = true;
    else g->comm.clear();
  }
  if (insist && !g->use_rccl)
    return fail(NBX_ERR_DEVICE, "nbx_group_create: NBX_EXCHANGE=rccl but RCCL is unavailable for these devices");
  if (!g->use_rccl) {
    g->done.assign(P, nullptr);
    for (int r = 0; r < P; ++r) {
      if (hipSetDevice(g->dev[r]) != hipSuccess || hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming) != hipSuccess)
        return fail(NBX_ERR_DEVICE, "nbx_group_create: hipEventCreate failed");
      for (int q = 0; q < P; ++q)  // best effort: direct peer access speeds hipMemcpyPeerAsync up
        if (g->dev[q] != g->dev[r]) { (void)hipDeviceEnablePeerAccess(g->dev[q], 0); (void)hipGetLastError(); }
    }
  }
  owner.g = nullptr;
  *out = g;
  g_err.clear();
  return NBX_OK;
  });
}

int nbx_comm_unique_id(void* id_out) {
  return guarded("nbx_comm_unique_id", [&]() -> int {
  if (!id_out) return fail(NBX_ERR_ARG, "nbx_comm_unique_id: id_out is NULL");
  if (!g_rccl.load()) return fail(NBX_ERR_DEVICE, "nbx_comm_unique_id: librccl could not be loaded");
  ncclUniqueId id;
  const ncclResult_t e = g_rccl.GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
  std::memcpy(id_out, &id, sizeof id);
  return NBX_OK;
  });
}

int nbx_group_create_rank(nbx_group** out, int32_t n, int32_t precision, int32_t world, int32_t rank, const void* unique_id,
                          int32_t device, const nbx_opts* opts) {
  return guarded("nbx_group_create_rank", [&]() -> int {
  nbx_opts o;
  int rc = check_group_args("nbx_group_create_rank", out, n, precision, world, opts, &o);
  if (rc) return rc;
  if (rank < 0 || rank >= world) return fail(NBX_ERR_ARG, "nbx_group_create_rank: rank must be in [0, world)");
  if (!unique_id) return fail(NBX_ERR_ARG, "nbx_group_create_rank: unique_id is NULL");
  int P = 0, block = 0;
  partition(n, world, &P, &block);
  // every rank computes the same P: a world too large for n is refused by ALL ranks alike (nobody is left waiting in a collective)
  if (P != world)
    return fail(NBX_ERR_ARG, "nbx_group_create_rank: " + std::to_string(n) + " bodies give only " + std::to_string(P) +
                                 " non-empty blocks of 256-aligned size; start at most that many ranks");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NBX_ERR_DEVICE, "nbx_group_create_rank: no HIP device available (libnbx has no CPU path)");
  const int dev = device >= 0 ? device : rank % ndev;
  if (dev >= ndev) return fail(NBX_ERR_ARG, "nbx_group_create_rank: device ordinal out of range");
  if (!g_rccl.load()) return fail(NBX_ERR_DEVICE, "nbx_group_create_rank: librccl could not be loaded");
  nbx_group* g = new (std::nothrow) nbx_group();
  if (!g) return fail(NBX_ERR_ALLOC, "nbx_group_create_rank: out of host memory");
  struct Owner { nbx_group* g; ~Owner() { nbx_group_destroy(g); } } owner{g};
  g->n = n; g->precision = precision; g->P = P; g->block = block; g->n_alloc = P * block; g->my_rank = rank;
  g->dev.push_back(dev);
  o.device = dev;
  o.i_begin = rank * block;
  o.i_count = std::min(n, (rank + 1) * block) - rank * block;
  o.n_alloc = g->n_alloc;
  nbx_ctx* c = nullptr;
  rc = nbx_create(&c, n, precision, &o);
  if (rc != NBX_OK) { const std::string m = g_err; return fail(rc, "nbx_group_create_rank: rank " + std::to_string(rank) + ": " + m); }
  g->rank.push_back(c);
  HIP_TRY(hipSetDevice(dev));
  HIP_TRY(hipMalloc(&g->ke_all, sizeof(double) * (size_t)P));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  g->comm.assign(1, nullptr);
  const ncclResult_t e = g_rccl.CommInitRank(&g->comm[0], P, id, rank);
  if (e != ncclSuccess) { g->comm.clear(); return rccl_fail("ncclCommInitRank", e); }
  g->use_rccl = true;
  owner.g = nullptr;
  *out = g;
  g_err.clear();
  return NBX_OK;
  });
}

void nbx_group_destroy(nbx_group* g) {
  if (!g) return;
  for (nbx_ctx* c : g->rank) if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
  for (auto cm : g->comm) if (cm) (void)g_rccl.CommDestroy(cm);
  for (size_t r = 0; r < g->done.size(); ++r) if (g->done[r]) { (void)hipSetDevice(g->dev[r]); (void)hipEventDestroy(g->done[r]); }
  if (!g->dev.empty()) (void)hipSetDevice(g->dev[0]);
  if (g->ke_all) (void)hipFree(g->ke_all);
  if (g->vel_stage) (void)hipFree(g->vel_stage);
  if (g->vel_all) (void)hipFree(g->vel_all);
  for (nbx_ctx* c : g->rank) nbx_destroy(c);
  delete g;
}

int nbx_group_upload(nbx_group* g, const void* px, const void* py, const void* pz, const void* vx, const void* vy,
                     const void* vz, const void* m) {
  return guarded("nbx_group_upload", [&]() -> int {
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_upload: group is NULL");
  for (nbx_ctx* c : g->rank) {
    const int rc = nbx_upload(c, px, py, pz, vx, vy, vz, m);
    if (rc) return rc;
  }
  return NBX_OK;
  });
}

int nbx_group_step(nbx_group* g, double dt, int32_t nsteps, double* kenergy_out) {
  return guarded("nbx_group_step", [&]() -> int {
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_step: group is NULL");
  if (nsteps < 0) return fail(NBX_ERR_ARG, "nbx_group_step: nsteps < 0");
  for (int s = 0; s < nsteps; ++s) {
    for (nbx_ctx* c : g->rank) {
      const int rc = nbx_step_local(c, dt);
      if (rc) return rc;
    }
    if (g->P > 1 || g->use_rccl) {
      const int rc = group_exchange(g);
      if (rc) return rc;
    }
    for (nbx_ctx* c : g->rank) {
      const int rc = nbx_commit(c);
      if (rc) return rc;
    }
  }
  if (kenergy_out) {
    double sum = 0.0;
    if (g->my_rank >= 0) {
      // one process per GPU: every rank reduces its partial on the device, one 8-byte all-gather, and all ranks add
      // the P values in rank order -- the same number on every rank, independent of arrival order
      nbx_ctx* c = g->rank[0];
      int rc = use_device(c);
      if (rc) return rc;
      if (c->ke_parts > 0) {
        rc = enqueue_ke_reduce(c, 0);
        if (rc) return rc;
      } else {
        HIP_TRY(hipMemsetAsync(c->ke_dev, 0, sizeof(double), c->stream));
      }
      const ncclResult_t e = g_rccl.AllGather(c->ke_dev, g->ke_all, sizeof(double), ncclChar, g->comm[0], c->stream);
      if (e != ncclSuccess) return rccl_fail("ncclAllGather(kenergy)", e);
      std::vector<double> parts((size_t)g->P);
      HIP_TRY(hipMemcpyAsync(parts.data(), g->ke_all, sizeof(double) * parts.size(), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      for (double p : parts) sum += p;
    } else {
      for (nbx_ctx* c : g->rank) {  // rank order: deterministic
        double part = 0.0;
        const int rc = nbx_kenergy_partial(c, &part);
        if (rc) return rc;
        sum += part;
      }
      for (nbx_ctx* c : g->rank) {  // the exchange copies of the last step must have landed too
        const int rc = nbx_sync(c);
        if (rc) return rc;
      }
    }
    *kenergy_out = 0.5 * sum;
  }
  return NBX_OK;
  });
}

int nbx_group_download(nbx_group* g, void* px, void* py, void* pz, void* vx, void* vy, void* vz) {
  return guarded("nbx_group_download", [&]() -> int {
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_download: group is NULL");
  for (nbx_ctx* c : g->rank) {
    const int rc = nbx_sync(c);
    if (rc) return rc;
  }
  if (g->my_rank >= 0) {
    // collective: every rank calls it.  Positions are complete on every rank; velocities live with their owners, so the
    // owned blocks are all-gathered (padded to `block` records) -- afterwards every caller holds the full final state,
    // as rank 0 of the reference does after mpi_gather (ver5_all/GSimulation.cpp:186-214).
    nbx_ctx* c = g->rank[0];
    int rc = nbx_download(c, px, py, pz, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (!vx && !vy && !vz) return NBX_OK;
    rc = use_device(c);
    if (rc) return rc;
    const size_t rec = c->rec, blk = rec * (size_t)g->block;
    if (!g->vel_stage) HIP_TRY(hipMalloc(&g->vel_stage, blk));
    if (!g->vel_all) HIP_TRY(hipMalloc(&g->vel_all, blk * (size_t)g->P));
    HIP_TRY(hipMemsetAsync(g->vel_stage, 0, blk, c->stream));
    HIP_TRY(hipMemcpyAsync(g->vel_stage, c->velm, rec * (size_t)c->i_count, hipMemcpyDeviceToDevice, c->stream));
    const ncclResult_t e = g_rccl.AllGather(g->vel_stage, g->vel_all, blk, ncclChar, g->comm[0], c->stream);
    if (e != ncclSuccess) return rccl_fail("ncclAllGather(velocities)", e);
    std::vector<char> h(rec * (size_t)g->n);
    HIP_TRY(hipMemcpyAsync(h.data(), g->vel_all, h.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < g->n; ++i) {
      if (c->precision == 32) {
        const float4& q = reinterpret_cast<const float4*>(h.data())[i];
        if (vx) ((float*)vx)[i] = q.x;
        if (vy) ((float*)vy)[i] = q.y;
        if (vz) ((float*)vz)[i] = q.z;
      } else {
        const double4& q = reinterpret_cast<const double4*>(h.data())[i];
        if (vx) ((double*)vx)[i] = q.x;
        if (vy) ((double*)vy)[i] = q.y;
        if (vz) ((double*)vz)[i] = q.z;
      }
    }
    return NBX_OK;
  }
  for (size_t r = 0; r < g->rank.size(); ++r) {  // positions once (rank 0 holds all), velocities per owner
    const int rc = nbx_download(g->rank[r], r == 0 ? px : nullptr, r == 0 ? py : nullptr, r == 0 ? pz : nullptr, vx, vy, vz);
    if (rc) return rc;
  }
  return NBX_OK;
  });
}

int nbx_group_info(nbx_group* g, int32_t* n_ranks, int32_t* uses_rccl, int32_t rank, nbx_stats_t* rank_stats) {
  return guarded("nbx_group_info", [&]() -> int {
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_info: group is NULL");
  if (n_ranks) *n_ranks = g->P;
  if (uses_rccl) *uses_rccl = g->use_rccl ? 1 : 0;
  if (rank_stats) {
    if (rank < 0 || rank >= g->P) return fail(NBX_ERR_ARG, "nbx_group_info: rank out of range");
    // a rank group holds this process's context only: its statistics are returned whatever rank is asked for
    return nbx_stats(g->rank[g->my_rank >= 0 ? 0 : rank], rank_stats);
  }
  return NBX_OK;
  });
}

}  // extern "C"
