// nbx_group.hip -- multi-GPU drivers of include/nbx.h over the contexts of nbx_api.hip: one process with k GPUs
// (nbx_group_create) and one process per GPU (nbx_comm_unique_id + nbx_group_create_rank), both exchanging the freshly
// integrated position blocks with one in-place RCCL all-gather per time step.  Replaces init_mpi + mpi_bcast_all +
// mpi_gather_acc of the reference (ver5_all/GSimulation.cpp:93-115,170-214) and its OpenCL multi-device loop
// (opencl/Compute.cpp:241-284).
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is dlopen'ed (struct Rccl), never linked

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "nbx_internal.hpp"
#include "nbx_watchdog.hpp"

using namespace nbx_detail;

namespace {
constexpr double kRcclInitAllowanceSeconds = 30.0;  // added to the collective timeout for ncclCommInitRank (see there)
constexpr int kTile = 256;  // records per block alignment (= nbx::kTile: j tile of the kernels; checked in nbx_api.hip)
}  // namespace

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
  decltype(&ncclBroadcast) Broadcast = nullptr;  // weighted groups: blocks of unequal size, one broadcast per owner
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
    sym(h, "ncclBroadcast", Broadcast);
    sym(h, "ncclGetErrorString", GetErrorString);
    ok = CommInitAll && CommInitRank && GetUniqueId && CommDestroy && GroupStart && GroupEnd && AllGather && Broadcast;
    return ok;
  }
  std::string text(ncclResult_t e) const { return GetErrorString ? std::string(GetErrorString(e)) : std::string("RCCL error ") + std::to_string((int)e); }
};
Rccl g_rccl;
static_assert(std::is_same<decltype(Rccl::AllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value,
              "ncclAllGather is called as (send, recv, bytes, ncclChar, comm, stream)");
static_assert(kExitCollectiveTimeout == NBX_EXIT_COLLECTIVE_TIMEOUT, "include/nbx.h documents the watchdog's exit status");
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
  // watchdog bookkeeping (nbx_watchdog.hpp): steps enqueued since the last host synchronisation, and what one step took
  long long steps_unsynced = 0;
  double step_s_est = 0.0;           // seconds per step measured over the last fully synchronised nbx_group_step call; 0 = not yet
  // weighted groups (nbx_group_create_weighted): rank r owns [begin[r], begin[r] + count[r]), whole 256-record tiles in proportion
  // to its weight; the per-step exchange is one broadcast per owner instead of the equal-block all-gather
  bool weighted = false;
  std::vector<int> begin, count;
  std::vector<double> weight;
  nbx_opts opts{};                   // launch-shape options the contexts were made with (nbx_group_retune rebuilds them)
  std::vector<char> mass_host;       // the masses as uploaded (n elements of the group's precision): needed to re-upload after a retune
  bool uploaded = false;
  long long retunes = 0;
  bool broken = false;               // a retune failed half-way (contexts could not be rebuilt): only nbx_group_destroy is left
  // the tuner accepts improvements only: the shares in force before the last move and the slowest rank's time under them
  std::vector<int> prev_begin, prev_count;
  double prev_max_ms = 0.0;          // 0 = no move to judge
  bool tuning_frozen = false;        // a move made the step slower and was taken back: the shares stay where they are
};

namespace {

// Balanced, tile-aligned blocks; ranks that would own nothing are dropped (P is reduced).  The same arithmetic as
// sharded.block_partition (tests/test_sharded_gloo.py::test_check_world_matches_the_native_partition compares them) with that
// reduction applied.
void partition(int n, int n_ranks, int* P_out, int* block_out) {
  int P = n_ranks, block = 0;
  for (;; --P) {
    block = round_up(ceil_div(n, P), kTile);
    if (P == 1 || (long long)(P - 1) * block < n) break;
  }
  *P_out = P;
  *block_out = block;
}

// Weighted partition: the ceil(n / 256) tiles of 256 records are handed out in proportion to the weights (largest remainder, ties to
// the lower rank), every rank at least one tile; ranks beyond the number of tiles are dropped.  The reference's co-execution split
// gives device 0 `n * cpu_ratio` bodies and the rest to the other device (opencl/Compute.cpp:241-249); here every block stays a whole
// number of j tiles, which is what the kernels' zero-mass padding and the in-place exchange rely on.
int partition_weighted(int n, int n_ranks, const double* w, std::vector<int>* begin, std::vector<int>* count, int* n_alloc) {
  const int tiles = ceil_div(n, kTile);
  const int P = std::min(n_ranks, tiles);
  double sum = 0.0;
  for (int r = 0; r < P; ++r) {
    const double x = w ? w[r] : 1.0;
    if (!(x > 0.0) || !std::isfinite(x)) return fail(NBX_ERR_ARG, "weights must be finite and > 0");
    sum += x;
  }
  std::vector<double> ideal((size_t)P);
  std::vector<int> t((size_t)P);
  long long total = 0;
  for (int r = 0; r < P; ++r) {
    ideal[r] = (double)tiles * (w ? w[r] : 1.0) / sum;
    t[r] = std::max(1, (int)std::floor(ideal[r]));
    total += t[r];
  }
  while (total != tiles) {  // at most P passes each way: every rank is within one tile of its ideal share afterwards (or at the 1-tile floor)
    int pick = -1;
    double best = 0.0;
    for (int r = 0; r < P; ++r) {
      const double d = total < tiles ? ideal[r] - t[r] : t[r] - ideal[r];
      if (total > tiles && t[r] <= 1) continue;
      if (pick < 0 || d > best) { pick = r; best = d; }
    }
    if (pick < 0) return fail(NBX_ERR_STATE, "partition_weighted: cannot balance the tiles");  // cannot happen: P <= tiles
    t[pick] += total < tiles ? 1 : -1;
    total += total < tiles ? 1 : -1;
  }
  begin->assign((size_t)P, 0);
  count->assign((size_t)P, 0);
  int first = 0;
  for (int r = 0; r < P; ++r) {
    (*begin)[r] = first * kTile;
    (*count)[r] = std::min(n, (first + t[r]) * kTile) - first * kTile;
    first += t[r];
  }
  *n_alloc = tiles * kTile;
  return NBX_OK;
}

// Seconds of legitimately queued work in front of a synchronisation: the watchdog's deadline is its timeout PLUS this, so
// that a long print window is not mistaken for a dead peer.  Four times the measured step time once a window has been
// timed; before that a rate no shape is slower than (2e10 pair/s: the validation kernel runs at ~5e11).
double queued_allowance(const nbx_group* g) {
  if (g->steps_unsynced <= 0 || g->rank.empty()) return 0.0;
  double per_step = 4.0 * g->step_s_est;
  if (!(per_step > 0.0)) {
    per_step = 1e-3;
    for (const nbx_ctx* c : g->rank) per_step += (double)c->i_count * (double)c->n / 2e10;  // logical ranks may share a GPU: add them up
  }
  return per_step * (double)g->steps_unsynced;
}

int rccl_fail(const char* what, ncclResult_t e) { return fail(NBX_ERR_DEVICE, std::string(what) + ": " + g_rccl.text(e)); }

int group_exchange(nbx_group* g) {
  const size_t rec = g->rank[0]->rec;
  if (g->use_rccl && g->weighted) {
    // blocks of unequal size: one in-place broadcast per owner, all of them in one group (single-process groups only)
    ncclResult_t e = g_rccl.GroupStart();
    if (e != ncclSuccess) return rccl_fail("ncclGroupStart", e);
    int rc = NBX_OK;
    for (size_t k = 0; k < g->rank.size() && rc == NBX_OK; ++k) {
      nbx_ctx* c = g->rank[k];
      char* buf = (char*)c->posm[c->cur ^ 1];
      const hipError_t he = hipSetDevice(g->dev[k]);
      if (he != hipSuccess) { rc = fail(NBX_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(he)); break; }
      for (int r = 0; r < g->P && rc == NBX_OK; ++r) {
        char* blk = buf + (size_t)g->begin[r] * rec;
        e = g_rccl.Broadcast(blk, blk, (size_t)g->count[r] * rec, ncclChar, r, g->comm[k], c->stream);
        if (e != ncclSuccess) rc = rccl_fail("ncclBroadcast", e);
      }
    }
    e = g_rccl.GroupEnd();
    if (rc == NBX_OK && e != ncclSuccess) rc = rccl_fail("ncclGroupEnd", e);
    return rc;
  }
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

}  // extern "C"

namespace {

// (re)creates the contexts of a single-process group from g->begin / g->count
int make_contexts(nbx_group* g, const char* who) {
  for (nbx_ctx* c : g->rank) nbx_destroy(c);
  g->rank.clear();
  nbx_opts o = g->opts;
  for (int r = 0; r < g->P; ++r) {
    o.device = g->dev[r];
    o.i_begin = g->begin[r];
    o.i_count = g->count[r];
    o.n_alloc = g->n_alloc;
    nbx_ctx* c = nullptr;
    const int rc = nbx_create(&c, g->n, g->precision, &o);
    if (rc != NBX_OK) { const std::string m = last_error(); return fail(rc, std::string(who) + ": rank " + std::to_string(r) + ": " + m); }
    g->rank.push_back(c);
    if (g->weighted) {  // per-launch timing of the force kernel: what nbx_group_retune weighs the ranks by
      const int pc = nbx_profile(c, 1);
      if (pc != NBX_OK) return pc;
    }
  }
  return NBX_OK;
}

int create_single_process(const char* who, nbx_group** out, int32_t n, int32_t precision, int32_t n_ranks, const int32_t* devices,
                          bool weighted, const double* weights, const nbx_opts* opts) {
  nbx_opts o;
  int rc = check_group_args(who, out, n, precision, n_ranks, opts, &o);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NBX_ERR_DEVICE, std::string(who) + ": no HIP device available (libnbx has no CPU path)");
  nbx_group* g = new (std::nothrow) nbx_group();
  if (!g) return fail(NBX_ERR_ALLOC, std::string(who) + ": out of host memory");
  struct Owner { nbx_group* g; ~Owner() { nbx_group_destroy(g); } } owner{g};  // every failure path below frees the group
  g->n = n; g->precision = precision; g->opts = o; g->weighted = weighted;
  if (weighted) {
    rc = partition_weighted(n, n_ranks, weights, &g->begin, &g->count, &g->n_alloc);
    if (rc) { const std::string m = last_error(); return fail(rc, std::string(who) + ": " + m); }
    g->P = (int)g->begin.size();
    g->block = 0;  // no common block size
    for (int r = 0; r < g->P; ++r) g->weight.push_back(weights ? weights[r] : 1.0);
  } else {
    int P = 0, block = 0;
    partition(n, n_ranks, &P, &block);
    g->P = P; g->block = block; g->n_alloc = P * block;
    for (int r = 0; r < P; ++r) { g->begin.push_back(r * block); g->count.push_back(std::min(n, (r + 1) * block) - r * block); }
  }
  const int P = g->P;
  bool distinct = true;
  for (int r = 0; r < P; ++r) {
    const int d = devices ? devices[r] : r % ndev;
    if (d < 0 || d >= ndev) return fail(NBX_ERR_ARG, std::string(who) + ": device ordinal out of range");
    for (int q : g->dev) distinct = distinct && q != d;
    g->dev.push_back(d);
  }
  rc = make_contexts(g, who);
  if (rc) return rc;
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
    return fail(NBX_ERR_DEVICE, std::string(who) + ": NBX_EXCHANGE=rccl but RCCL is unavailable for these devices");
  if (!g->use_rccl) {
    g->done.assign(P, nullptr);
    for (int r = 0; r < P; ++r) {
      if (hipSetDevice(g->dev[r]) != hipSuccess || hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming) != hipSuccess)
        return fail(NBX_ERR_DEVICE, std::string(who) + ": hipEventCreate failed");
      for (int q = 0; q < P; ++q)  // best effort: direct peer access speeds hipMemcpyPeerAsync up
        if (g->dev[q] != g->dev[r]) { (void)hipDeviceEnablePeerAccess(g->dev[q], 0); (void)hipGetLastError(); }
    }
  }
  owner.g = nullptr;
  *out = g;
  last_error().clear();
  return NBX_OK;
}

}  // namespace

extern "C" {

int nbx_group_create(nbx_group** out, int32_t n, int32_t precision, int32_t n_ranks, const int32_t* devices,
                     const nbx_opts* opts) {
  return guarded("nbx_group_create", [&]() -> int {
    return create_single_process("nbx_group_create", out, n, precision, n_ranks, devices, false, nullptr, opts);
  });
}

int nbx_group_create_weighted(nbx_group** out, int32_t n, int32_t precision, int32_t n_ranks, const int32_t* devices,
                              const double* weights, const nbx_opts* opts) {
  return guarded("nbx_group_create_weighted", [&]() -> int {
    return create_single_process("nbx_group_create_weighted", out, n, precision, n_ranks, devices, true, weights, opts);
  });
}

int nbx_partition_weighted(int32_t n, int32_t n_ranks, const double* weights, int32_t rank, int32_t* ranks_used, int32_t* i_begin,
                           int32_t* i_count, int32_t* n_alloc) {
  return guarded("nbx_partition_weighted", [&]() -> int {
  if (n <= 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(NBX_ERR_ARG, "nbx_partition_weighted: need n > 0 and 0 <= rank < n_ranks");
  std::vector<int> b, c;
  int na = 0;
  const int rc = partition_weighted(n, n_ranks, weights, &b, &c, &na);
  if (rc) { const std::string m = last_error(); return fail(rc, "nbx_partition_weighted: " + m); }
  const int P = (int)b.size();
  if (ranks_used) *ranks_used = P;
  if (i_begin) *i_begin = rank < P ? b[rank] : n;
  if (i_count) *i_count = rank < P ? c[rank] : 0;
  if (n_alloc) *n_alloc = na;
  return NBX_OK;
  });
}

// New weights from what every rank achieved: its bodies per millisecond of force kernel, normalised to sum 1.
int nbx_tune_weights(int32_t n_ranks, const int32_t* i_count, const double* force_ms, double* weights_out) {
  return guarded("nbx_tune_weights", [&]() -> int {
  if (n_ranks <= 0 || !i_count || !force_ms || !weights_out) return fail(NBX_ERR_ARG, "nbx_tune_weights: NULL argument or no ranks");
  double sum = 0.0;
  for (int r = 0; r < n_ranks; ++r) {
    if (i_count[r] <= 0 || !(force_ms[r] > 0.0) || !std::isfinite(force_ms[r]))
      return fail(NBX_ERR_ARG, "nbx_tune_weights: every rank needs bodies and a positive measured time");
    sum += (double)i_count[r] / force_ms[r];
  }
  for (int r = 0; r < n_ranks; ++r) weights_out[r] = ((double)i_count[r] / force_ms[r]) / sum;
  return NBX_OK;
  });
}

int nbx_group_shares(nbx_group* g, int32_t* i_begin, int32_t* i_count, double* force_ms) {
  return guarded("nbx_group_shares", [&]() -> int {
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_shares: group is NULL");
  if (g->broken) return fail(NBX_ERR_STATE, "nbx_group_shares: a retune failed while rebuilding the contexts; destroy the group");
  for (int r = 0; r < g->P; ++r) {
    if (i_begin) i_begin[r] = g->begin[r];
    if (i_count) i_count[r] = g->count[r];
  }
  if (force_ms) {
    for (int r = 0; r < g->P; ++r) force_ms[r] = 0.0;
    for (size_t k = 0; k < g->rank.size(); ++k) {
      nbx_stats_t st;
      const int rc = nbx_stats(g->rank[k], &st);  // synchronises this rank's stream and drains its events
      if (rc) return rc;
      const int r = g->my_rank >= 0 ? g->my_rank : (int)k;
      force_ms[r] = st.force_launches_timed > 0 ? st.force_ms_total / (double)st.force_launches_timed : 0.0;
    }
  }
  return NBX_OK;
  });
}

int nbx_group_retune(nbx_group* g, const double* force_ms, int32_t* changed) {
  return guarded("nbx_group_retune", [&]() -> int {
  if (changed) *changed = 0;
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_retune: group is NULL");
  if (!g->weighted || g->my_rank >= 0) return fail(NBX_ERR_STATE, "nbx_group_retune: needs a single-process group made by nbx_group_create_weighted");
  if (g->broken) return fail(NBX_ERR_STATE, "nbx_group_retune: an earlier retune failed while rebuilding the contexts; destroy the group");
  if (!g->uploaded) return fail(NBX_ERR_STATE, "nbx_group_retune: nbx_group_upload has not been called");
  std::vector<double> ms((size_t)g->P), w((size_t)g->P);
  if (force_ms) {
    for (int r = 0; r < g->P; ++r) ms[r] = force_ms[r];
  } else {
    const int rc = nbx_group_shares(g, nullptr, nullptr, ms.data());
    if (rc) return rc;
    for (int r = 0; r < g->P; ++r)
      if (!(ms[r] > 0.0)) return NBX_OK;  // a rank without a timed launch since the last retune: nothing to weigh by, shares stay
  }
  // restart the measurement window whether or not the shares move
  auto restart_timing = [&]() -> int {
    for (nbx_ctx* x : g->rank) {
      int pc = nbx_profile(x, 0);
      if (pc == NBX_OK) pc = nbx_profile(x, 1);
      if (pc != NBX_OK) return pc;
    }
    return NBX_OK;
  };
  // The step lasts as long as the slowest rank.  A rank's time is NOT linear in its share: a reference-order launch lasts as long as
  // its fullest SIMD, so one body more than a whole number of waves per SIMD costs a whole extra wave there (131072 bodies of 1M: 30 ms,
  // 131073: 58 ms).  The rate-proportional move below cannot know that, so it is judged by its result: if the window after a move was
  // slower than the window before it, the move is taken back and the shares are left alone from then on.
  double cur_max = 0.0;
  for (int r = 0; r < g->P; ++r) cur_max = std::max(cur_max, ms[r]);
  std::vector<int> b, c;
  int na = g->n_alloc;
  int rc = NBX_OK;
  if (g->prev_max_ms > 0.0 && cur_max > 1.01 * g->prev_max_ms) {
    b = g->prev_begin; c = g->prev_count;   // back to the shares that were faster
    g->tuning_frozen = true;
    g->prev_max_ms = 0.0;
  } else {
    g->prev_max_ms = 0.0;
    if (g->tuning_frozen) return restart_timing();
    rc = nbx_tune_weights(g->P, g->count.data(), ms.data(), w.data());
    if (rc) return rc;
    rc = partition_weighted(g->n, g->P, w.data(), &b, &c, &na);
    if (rc) return rc;
    if ((int)b.size() != g->P || na != g->n_alloc || c == g->count) return restart_timing();  // same shares (or the 256-record tiles allow no finer step)
    // Predict before moving: rank r's time under the new shares = its measured time x model(new share) / model(present share), with the
    // library's own cost table as the model (a step function of the share in reference order).  A move that the model expects to make
    // the slowest rank slower -- e.g. across a one-workgroup-per-CU boundary -- is not made.
    double predicted = 0.0;
    for (int r = 0; r < g->P; ++r) {
      const nbx_ctx* x = g->rank[(size_t)r];
      const double now = model_force_cost(x, g->count[r]), then = model_force_cost(x, c[r]);
      predicted = std::max(predicted, now > 0.0 ? ms[r] * then / now : ms[r]);
    }
    if (predicted > 0.99 * cur_max) return restart_timing();
    g->prev_begin = g->begin; g->prev_count = g->count; g->prev_max_ms = cur_max;
  }
  // Shares move: velocities live with their owners, so the state goes through the host once -- positions from rank 0 (every rank holds
  // them all), velocities from each owner -- and comes back to contexts with the new slices.  Values are copied, never recomputed: the
  // trajectory is the same bit for bit in reference summation order, whoever owns a body (tests compare).
  const size_t es = g->precision == 32 ? sizeof(float) : sizeof(double);
  std::vector<char> h(6 * es * (size_t)g->n);
  char* a[6];
  for (int k = 0; k < 6; ++k) a[k] = h.data() + (size_t)k * es * (size_t)g->n;
  rc = nbx_group_download(g, a[0], a[1], a[2], a[3], a[4], a[5]);
  if (rc) return rc;
  g->begin = b; g->count = c;
  g->weight.assign((size_t)g->P, 0.0);
  for (int r = 0; r < g->P; ++r) g->weight[r] = (double)c[r] / (double)g->n;
  g->broken = true;  // until every new context exists and holds the state
  rc = make_contexts(g, "nbx_group_retune");
  if (rc) return rc;
  for (nbx_ctx* x : g->rank) {
    rc = nbx_upload(x, a[0], a[1], a[2], a[3], a[4], a[5], g->mass_host.data());
    if (rc) return rc;
  }
  g->broken = false;
  g->steps_unsynced = 0;
  g->step_s_est = 0.0;
  g->retunes += 1;
  if (changed) *changed = 1;
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
  for (int r = 0; r < P; ++r) { g->begin.push_back(r * block); g->count.push_back(std::min(n, (r + 1) * block) - r * block); }
  g->dev.push_back(dev);
  o.device = dev;
  o.i_begin = rank * block;
  o.i_count = std::min(n, (rank + 1) * block) - rank * block;
  o.n_alloc = g->n_alloc;
  nbx_ctx* c = nullptr;
  rc = nbx_create(&c, n, precision, &o);
  if (rc != NBX_OK) { const std::string m = last_error(); return fail(rc, "nbx_group_create_rank: rank " + std::to_string(rank) + ": " + m); }
  g->rank.push_back(c);
  HIP_TRY(hipSetDevice(dev));
  HIP_TRY(hipMalloc(&g->ke_all, sizeof(double) * (size_t)P));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  g->comm.assign(1, nullptr);
  Watchdog::instance().set_identity(rank, P);
  ncclResult_t e;
  {
    // blocks until all P ranks have called it.  RCCL's own set-up -- loading its kernels, topology detection, channel set-up --
    // is legitimate work in front of the hand-shake: 4-5 s for a world of one on a cold process (measured), more on eight GPUs
    double init_allowance = kRcclInitAllowanceSeconds;  // NBX_RCCL_INIT_ALLOWANCE=<seconds> overrides (large nodes; tests)
    if (const char* txt = std::getenv("NBX_RCCL_INIT_ALLOWANCE")) {
      char* end = nullptr;
      const double v = std::strtod(txt, &end);
      if (end != txt && v >= 0.0) init_allowance = v;
    }
    Watchdog::Scope bounded("ncclCommInitRank (nbx_group_create_rank)", init_allowance);
    e = g_rccl.CommInitRank(&g->comm[0], P, id, rank);
  }
  if (e != ncclSuccess) { g->comm.clear(); return rccl_fail("ncclCommInitRank", e); }
  g->use_rccl = true;
  owner.g = nullptr;
  *out = g;
  last_error().clear();
  return NBX_OK;
  });
}

void nbx_group_destroy(nbx_group* g) {
  if (!g) return;
  // draining the streams and tearing the communicators down waits for collectives in flight: bounded like the others
  Watchdog::Scope bounded("nbx_group_destroy (stream synchronisation + ncclCommDestroy)", queued_allowance(g));
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
  if (g->weighted && m) {  // nbx_group_retune re-uploads the state to contexts with other slices: it needs the masses again
    const size_t bytes = (g->precision == 32 ? sizeof(float) : sizeof(double)) * (size_t)g->n;
    g->mass_host.assign((const char*)m, (const char*)m + bytes);
  }
  g->uploaded = true;
  return NBX_OK;
  });
}

int nbx_group_step(nbx_group* g, double dt, int32_t nsteps, double* kenergy_out) {
  return guarded("nbx_group_step", [&]() -> int {
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_step: group is NULL");
  if (nsteps < 0) return fail(NBX_ERR_ARG, "nbx_group_step: nsteps < 0");
  if (g->broken) return fail(NBX_ERR_STATE, "nbx_group_step: a retune failed while rebuilding the contexts; destroy the group");
  const auto t_enter = std::chrono::steady_clock::now();
  const bool window_from_sync = g->steps_unsynced == 0;  // everything this call waits for was enqueued by this call
  g->steps_unsynced += nsteps;
  // Groups that exchange over RCCL are bounded from the FIRST enqueue on, not only at the final synchronisation: the first
  // collective of a communicator sets its channels up on the host inside ncclGroupEnd / ncclAllGather, and a full launch queue
  // blocks the host in hipLaunchKernel -- a peer that died after ncclCommInitRank would otherwise leave this rank in the loop
  // below, which never reaches an armed region (and never at all when kenergy_out == NULL).  Scopes nest: the inner ones stay.
  Watchdog::Scope bounded_enqueue(g->use_rccl, "nbx_group_step (enqueue: local steps + position all-gathers)", queued_allowance(g));
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
    // the one place a stepping group blocks: every all-gather enqueued above completes only if every rank took part
    Watchdog::Scope bounded("nbx_group_step (position all-gathers + kinetic energy: stream synchronisation)", queued_allowance(g));
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
    if (window_from_sync && nsteps > 0)
      g->step_s_est = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count() / nsteps;
    g->steps_unsynced = 0;
  }
  return NBX_OK;
  });
}

int nbx_group_download(nbx_group* g, void* px, void* py, void* pz, void* vx, void* vy, void* vz) {
  return guarded("nbx_group_download", [&]() -> int {
  if (!g) return fail(NBX_ERR_ARG, "nbx_group_download: group is NULL");
  if (g->broken) return fail(NBX_ERR_STATE, "nbx_group_download: a retune failed while rebuilding the contexts; destroy the group");
  Watchdog::Scope bounded("nbx_group_download (stream synchronisation + all-gather of the velocities)", queued_allowance(g));
  struct Synced { nbx_group* g; ~Synced() { g->steps_unsynced = 0; } } synced{g};
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

int nbx_collective_timeout(double seconds) {
  return guarded("nbx_collective_timeout", [&]() -> int {
  if (std::isnan(seconds)) return fail(NBX_ERR_ARG, "nbx_collective_timeout: seconds is NaN");
  Watchdog::instance().set_timeout(seconds);
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
