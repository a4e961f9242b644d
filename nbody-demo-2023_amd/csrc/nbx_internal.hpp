// nbx_internal.hpp -- what the translation units of libnbx.so share: the context object, error plumbing, small helpers.
// Not part of the C-ABI (include/nbx.h is); nothing here is visible outside the library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/nbx.h"

namespace nbx_detail {

std::string& last_error();  // thread-local text behind nbx_last_error() (defined in nbx_api.hip)

inline int fail(int code, const std::string& msg) {
  last_error() = msg;
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
inline int guarded(const char* where, F&& body) noexcept {
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

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace nbx_detail

struct nbx_ctx {
  int n = 0, n_alloc = 0, i_begin = 0, i_count = 0, own_pad = 0, precision = 32;
  int B = 1, S = 1, jps = 0, variant = NBX_KERNEL_LDS, epi = 0 /* nbx::EPI_SLAB */, math = 0 /* nbx::MATH_SCALAR */, order = NBX_ORDER_TREE;
  int loop = 0;  // nbx::LOOP_CXX; nbx::LOOP_ASM where the hand-scheduled j loop is in use; nbx::LOOP_ASM_TS with time-sliced wave priority
  unsigned slice_bit = 0;  // LOOP_ASM_TS: clock bit of the priority slices
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
  void* posm_pairs = nullptr;      // one body per lane + hand-scheduled loop only: pair-interleaved copy of posm[cur], rebuilt every step
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

namespace nbx_detail {
// shared by the context entry points (nbx_api.hip) and the groups (nbx_group.hip)
int use_device(nbx_ctx* c);
double model_force_cost(const nbx_ctx* c, int own);  // relative cost of one force launch if the context owned `own` bodies (the tuner's predictor)
int enqueue_ke_reduce(nbx_ctx* c, int slot);  // ke_part[0 .. ke_parts) -> ke_dev[slot], fixed order, on the context's stream
}  // namespace nbx_detail
