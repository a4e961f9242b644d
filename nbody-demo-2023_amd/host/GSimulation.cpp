// GSimulation.cpp -- host side of the MI355X drop-in for the reference's GSimulation.
//
// Mirrors the observable behaviour of ver7/GSimulation.cpp: defaults (:24-32), seed-42 initial
// conditions (:45-94), stdout table (:203-212, :236-240, :245-263), flop model (:133) and window
// statistics (:213-230).  The two per-step loops (:141-198) are NOT here: start() hands the
// particle store to libnbx (include/nbx.h) once, steps it on the GPU one print window at a time
// and copies the final state back, so `particles->*` end up as the reference leaves them.
#include "GSimulation.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/nbx.h"
#include "cpu_time.hpp"
#include "rendezvous.hpp"
#include "snapshot.hpp"

namespace {

const int kPrecisionBits = 8 * (int)sizeof(real_type);

int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

const char* kernel_name(int variant) {
  switch (variant) {
    case NBX_KERNEL_LDS: return "lds";
    case NBX_KERNEL_SGPR: return "sgpr";
    case NBX_KERNEL_SGPRW: return "sgprw";
    case NBX_KERNEL_EXACT: return "exact";
    case NBX_KERNEL_EXACT_FMA: return "exact-fma";
    case NBX_KERNEL_JLANE: return "jlane";
  }
  return "?";
}

void die_nbx(const char* where) {
  std::cerr << "nbody.x: " << where << " failed: " << nbx_last_error() << std::endl;
  std::exit(1);
}

real_type* alloc_array(int n) {
  void* p = NULL;
  const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(real_type);
  if (posix_memalign(&p, 64, bytes) != 0) {
    std::cerr << "nbody.x: out of host memory" << std::endl;
    std::exit(1);
  }
  return static_cast<real_type*>(p);
}

}  // namespace

GSimulation::GSimulation()
    : world_rank(0), world_size(1), npp(0), npp_global(NULL), particles(NULL), _kenergy(0), _totTime(0), _totFlops(0),
      _cpu_ratio(-1.f), _thread_dim0(0), _thread_dim1(0), _devices(0), _allocated(false), _alloc_n(0), _multiprocess(false),
      _master_addr("127.0.0.1"), _master_port(29417), _local_rank(-1) {
  read_world_env();  // no network yet: only who we are, so that rank 0 alone prints (as in ver5_all/main.cpp:58-61)
#ifndef NBX_BANNER_IN_MAIN  // ver7 prints the banner here (ver7/GSimulation.cpp:26-27), ver5_all in main()
  if (world_rank == 0) {
    std::cout << "===============================" << std::endl;
    std::cout << " Initialize Gravity Simulation" << std::endl;
  }
#endif
  set_npart(2000);
  set_nsteps(500);
  // (float)0.1 as in ver7/GSimulation.cpp:30; the fp64 variant (BASELINE.json configs[4]) widens the
  // float-rounded constants, so its time step is (double)0.1f = 0.100000001490116 -- the value nbx.DT and every fp64
  // fixture use -- not the double literal 0.1
  set_tstep(real_type(0.1f));
  set_sfreq(50);
}

GSimulation::~GSimulation() {
  release_store();
  std::free(npp_global);
}

// Who am I?  NBODY_WORLD / NBODY_RANK.  The variables torchrun exports (WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR /
// MASTER_PORT) are honoured only with NBODY_USE_TORCHRUN_ENV=1 in the environment --
//   NBODY_USE_TORCHRUN_ENV=1 python -m torch.distributed.run --no-python --nproc-per-node 8 ./nbody.x 1048576 100
// -- because they are generic: a lone nbody.x started by a torchrun worker (or inside a PyTorchJob pod) inherits them
// and must not wait two minutes for peers that were never launched.  Absent: one process, rank 0 of 1.  Nothing exits
// here (this runs in the constructor): a bad description is remembered and reported by init_mpi() / start().
void GSimulation::read_world_env() {
  const char* w = std::getenv("NBODY_WORLD");
  const char* r = std::getenv("NBODY_RANK");
  const bool torchrun = env_int("NBODY_USE_TORCHRUN_ENV", 0) != 0;
  if (!w && torchrun) { w = std::getenv("WORLD_SIZE"); r = std::getenv("RANK"); }
  world_size = (w && *w) ? std::atoi(w) : 1;
  world_rank = (r && *r) ? std::atoi(r) : 0;
  _multiprocess = w && *w;  // NBODY_WORLD=1 still takes the rank-group path (one-rank rehearsal of RCCL)
  _world_error.clear();
  if (world_size < 1 || world_rank < 0 || world_rank >= world_size) {
    _world_error = "bad world description (world " + std::to_string(world_size) + ", rank " + std::to_string(world_rank) + ")";
    world_size = 1; world_rank = 0; _multiprocess = false;  // so that the banner still has one author
  }
  const char* a = std::getenv("NBODY_MASTER_ADDR");
  if (!a && torchrun) a = std::getenv("MASTER_ADDR");
  if (a && *a) _master_addr = a;
  const char* p = std::getenv("NBODY_MASTER_PORT");
  if (!p && torchrun) p = std::getenv("MASTER_PORT");
  if (p && *p) _master_port = std::atoi(p);
  _local_rank = env_int("NBODY_LOCAL_RANK", torchrun ? env_int("LOCAL_RANK", -1) : -1);  // the generic LOCAL_RANK only under the opt-in
}

void GSimulation::require_world() const {
  if (_world_error.empty()) return;
  std::cerr << "nbody.x: " << _world_error << std::endl;
  std::exit(1);
}

// ver5_all/GSimulation.cpp:93-115.  The reference calls MPI_Init here and gives every rank n / size bodies (rank 0 the
// remainder as well).  Here: the environment says who we are (read_world_env), and the shares are the 256-aligned blocks
// of nbx_partition -- what each rank's GPU will own.  No network is touched before start().
void GSimulation::init_mpi() {
  read_world_env();
  require_world();
  std::free(npp_global);
  npp_global = static_cast<int*>(std::malloc(sizeof(int) * (size_t)world_size));
  npp = 0;
  for (int r = 0; r < world_size && npp_global; ++r) {
    int32_t cnt = 0;
    if (get_npart() > 0 && nbx_partition(get_npart(), world_size, r, NULL, NULL, NULL, &cnt, NULL)) die_nbx("nbx_partition");
    npp_global[r] = cnt;
    if (r == world_rank) npp = cnt;
  }
}

void GSimulation::set_number_of_particles(int N) { set_npart(N); }
void GSimulation::set_number_of_steps(int N) { set_nsteps(N); }

void GSimulation::allocate_store(int n) {
  release_store();
  particles = new ParticleSoA();
  real_type** slots[10] = {&particles->pos_x, &particles->pos_y, &particles->pos_z, &particles->vel_x,
                           &particles->vel_y, &particles->vel_z, &particles->acc_x, &particles->acc_y,
                           &particles->acc_z, &particles->mass};
  for (int k = 0; k < 10; ++k) *slots[k] = alloc_array(n);
  _allocated = true;
  _alloc_n = n;
}

void GSimulation::release_store() {
  if (!_allocated) return;
  real_type* arrays[10] = {particles->pos_x, particles->pos_y, particles->pos_z, particles->vel_x,
                           particles->vel_y, particles->vel_z, particles->acc_x, particles->acc_y,
                           particles->acc_z, particles->mass};
  for (int k = 0; k < 10; ++k) std::free(arrays[k]);
  delete particles;
  particles = NULL;
  _allocated = false;
  _alloc_n = 0;
}

// The four init_* keep the reference's names and order of use; the draws themselves are the
// bit-exact restatement in libnbx (nbx_ic_*), so the particles do not depend on the libstdc++
// this file happens to be compiled against.
void GSimulation::init_pos() {
  if (nbx_ic_pos(get_npart(), kPrecisionBits, particles->pos_x, particles->pos_y, particles->pos_z)) die_nbx("nbx_ic_pos");
}
void GSimulation::init_vel() {
  if (nbx_ic_vel(get_npart(), kPrecisionBits, particles->vel_x, particles->vel_y, particles->vel_z)) die_nbx("nbx_ic_vel");
}
void GSimulation::init_acc() {
  for (int i = 0; i < get_npart(); ++i) particles->acc_x[i] = particles->acc_y[i] = particles->acc_z[i] = real_type(0);
}
void GSimulation::init_mass() {
  if (nbx_ic_mass(get_npart(), kPrecisionBits, particles->mass)) die_nbx("nbx_ic_mass");
}

void GSimulation::init() {
  allocate_store(get_npart());
  init_pos();
  init_vel();
  init_acc();
  init_mass();
}

void GSimulation::print_header() {
  if (world_rank != 0) return;  // ver5_all/GSimulation.cpp:119
  std::cout << " nPart = " << get_npart() << "; "
            << "nSteps = " << get_nsteps() << "; "
            << "dt = " << get_tstep() << std::endl;
  const std::string rule(48, '-');
  std::cout << rule << std::endl;
  std::cout << " " << std::left << std::setw(8) << "s" << std::setw(8) << "dt" << std::setw(12) << "kenergy"
            << std::setw(12) << "time (s)" << std::setw(12) << "GFlops" << std::endl;
  std::cout << rule << std::endl;
}

void GSimulation::start() {
  const int n = get_npart();
  const int nsteps = get_nsteps();
  // NBODY_SFREQ=<k>: print (and synchronise) every k steps instead of the reference's fixed 50 (_sfreq, ver7/GSimulation.cpp:31,
  // set only by its constructor) -- bench.py's native cross-check uses short windows
  if (env_int("NBODY_SFREQ", 0) > 0) set_sfreq(env_int("NBODY_SFREQ", 0));
  const int sfreq = get_sfreq();
  const double dt = (double)get_tstep();
  require_world();

  // ver5_all device word (ver5_all/main.cpp:40-47): 1 = "cpu", 2 = "gpu", 3 = "cpu+gpu", 0 = not given.  libnbx has no
  // CPU engine by design (no fallback may stand in for the HIP path), so "cpu" is refused here -- also when the
  // reference's own main.cpp drives this class.  "cpu+gpu" is the co-execution split of the OpenCL back end: two devices share
  // the bodies by `cpu_ratio` (argv[4]; opencl/Compute.cpp:154-162,241-255), and a negative ratio means "tuning": the ratio is
  // stepped every print window (:317-321).  Its GPU-native reading (include/nbx.h, nbx_group_create_weighted): every device is a
  // GPU; with NBODY_GPUS=k >= 2 device 0 owns the share `cpu_ratio` and the other devices split the rest (exactly the reference's
  // arithmetic for two devices), and a negative or missing ratio tunes -- every printed row the shares are re-weighted from each
  // device's measured force-kernel time.  With a single GPU there is nothing to split: all bodies run there and the note below goes
  // to stderr, to the `#` lines behind the footer and to NBODY_JSON.
  if (_devices == 1) {
    std::cerr << "nbody.x: device \"cpu\" requested, but this build has no CPU engine (the hot path lives in libnbx on the GPU); use gpu"
              << std::endl;
    std::exit(1);
  }
  const bool root = world_rank == 0;  // every rank computes; rank 0 alone prints (ver5_all/GSimulation.cpp:119,136,162)

  init();
  // NBODY_RESTART=<file>: continue from a snapshot instead of the seed-42 initial conditions
  long long steps_before = 0;
  if (const char* rs = std::getenv("NBODY_RESTART")) {
    std::string err;
    if (n > 0 && !nbx_snapshot::load(rs, particles, n, &steps_before, &err)) {
      std::cerr << "nbody.x: restart failed: " << err << std::endl;
      std::exit(1);
    }
  }
  print_header();
#ifdef NBX_BANNER_IN_MAIN
  // The ver5_all surface: its HIP back end announces the block size between the header and the first row
  // (ver5_all/programming_models/hip/Compute.cpp:133-140; argv[5] = thread_dim0 overrides its default of 256).  The kernels
  // here are written for workgroups of 256 threads -- four wave64s, the j tile, the LDS reductions -- so the line always
  // says 256, and another request is answered on stderr instead of being dropped silently.
  if (root) {
    if (_thread_dim0 != 0 && _thread_dim0 != 256)
      std::cerr << "nbody.x: thread_dim0 = " << _thread_dim0 << " ignored: the gfx950 kernels use workgroups of 256 threads (dim1 = bodies per lane is honoured)"
                << std::endl;
    std::cout << "using block_size = " << 256 << std::endl;
  }
#endif

  if (n <= 0) {  // the reference would run zero-trip loops; nothing to hand to the GPU
    if (!root) return;
    std::cout << std::endl << "# Number Threads     : 1" << std::endl;
    std::cout << "# Total Time (s)     : 0" << std::endl;
    std::cout << "# Average Perfomance : " << std::nan("") << " +- " << std::nan("") << std::endl;
    std::cout << "===============================" << std::endl;
    return;
  }

  nbx_opts opts;
  std::memset(&opts, 0, sizeof(opts));
  opts.struct_size = (int32_t)sizeof(opts);
  opts.device = env_int("NBODY_DEVICE", -1);
  opts.bodies_per_lane = env_int("NBODY_BPL", _thread_dim1 > 0 ? _thread_dim1 : 0);
  opts.j_split = env_int("NBODY_JSPLIT", 0);
  opts.fused_epilogue = env_int("NBODY_FUSED", 0);
  const char* kv = std::getenv("NBODY_KERNEL");
  if (kv && !std::strcmp(kv, "sgpr")) opts.kernel_variant = NBX_KERNEL_SGPR;
  if (kv && !std::strcmp(kv, "lds")) opts.kernel_variant = NBX_KERNEL_LDS;
  if (kv && !std::strcmp(kv, "sgprw")) opts.kernel_variant = NBX_KERNEL_SGPRW;
  if (kv && !std::strcmp(kv, "jlane")) opts.kernel_variant = NBX_KERNEL_JLANE;
  if (kv && !std::strcmp(kv, "exact")) opts.kernel_variant = NBX_KERNEL_EXACT;  // bit-for-bit the CPU ver7 arithmetic

  // NBODY_LOOP=asm|asm_ts|cxx: the hand-scheduled j loop (default wherever it exists; asm_ts = with time-sliced wave
  // priority, the default where a SIMD holds several reference-order waves) or the compiler-scheduled one -- same bits
  if (const char* lp = std::getenv("NBODY_LOOP")) {
    if (!std::strcmp(lp, "asm")) opts.inner_loop = NBX_LOOP_ASM;
    if (!std::strcmp(lp, "asm_ts")) opts.inner_loop = NBX_LOOP_ASM_TS;
    if (!std::strcmp(lp, "asm_pf")) opts.inner_loop = NBX_LOOP_ASM_PF;
    if (!std::strcmp(lp, "cxx")) opts.inner_loop = NBX_LOOP_CXX;
  }
  // NBODY_ORDER=reference|tree: how each body's pair terms are summed (include/nbx.h summation_order); default auto
  if (const char* ord = std::getenv("NBODY_ORDER")) {
    if (!std::strcmp(ord, "reference")) opts.summation_order = NBX_ORDER_REFERENCE;
    if (!std::strcmp(ord, "tree")) opts.summation_order = NBX_ORDER_TREE;
  }

  // NBODY_GPUS=k: block-partition the bodies over k GPUs of this node (one all-gather of positions per step);
  // k larger than the device count gives logical ranks sharing devices.  Default: one context on one GPU.
  const int gpus = env_int("NBODY_GPUS", 1);
  // Unequal shares (single process, k >= 2): NBODY_WEIGHTS=w0,w1,... fixes them, NBODY_TUNE=1 re-weights them every printed row from
  // the measured force-kernel times; the ver5_all device word cpu+gpu maps onto the same two things (see above).
  std::vector<double> weights;
  bool tune = env_int("NBODY_TUNE", 0) != 0;
  if (const char* ws = std::getenv("NBODY_WEIGHTS")) {
    const char* q = ws;
    while (*q) {
      char* end = NULL;
      const double v = std::strtod(q, &end);
      if (end == q) break;
      weights.push_back(v);
      q = (*end == ',') ? end + 1 : end;
    }
    if ((int)weights.size() != gpus) {
      std::cerr << "nbody.x: NBODY_WEIGHTS needs " << gpus << " comma-separated numbers (NBODY_GPUS=" << gpus << "), got " << weights.size() << std::endl;
      std::exit(1);
    }
  }
  std::string split_note;  // what became of cpu+gpu / cpu_ratio: stderr, the `#` lines behind the footer, NBODY_JSON
  if (_devices == 3) {
    if (_multiprocess || gpus < 2) {
      split_note = _multiprocess ? "cpu+gpu: one process per GPU uses equal blocks, cpu_ratio ignored (unequal shares need the single-process form, NBODY_GPUS=k)"
                                 : "cpu+gpu: one GPU and no CPU engine, all bodies on the GPU, cpu_ratio ignored (NBODY_GPUS=k >= 2 splits by it)";
    } else if (weights.empty()) {
      if (_cpu_ratio > 0.f && _cpu_ratio < 1.f) {
        weights.assign((size_t)gpus, (1.0 - (double)_cpu_ratio) / (double)(gpus - 1));
        weights[0] = (double)_cpu_ratio;
        split_note = "cpu+gpu: device 0 owns the share cpu_ratio = " + std::to_string(_cpu_ratio) + ", the other " + std::to_string(gpus - 1) + " device(s) the rest";
      } else {
        tune = true;  // negative (the reference's tuning mode), missing, or a ratio that would leave a device without bodies
        split_note = "cpu+gpu: tuning -- shares re-weighted every printed row from each device's measured force-kernel time";
      }
    }
    if (root && !split_note.empty()) std::cerr << "nbody.x: " << split_note << std::endl;
  }
  if (root && (tune || !weights.empty()) && (_multiprocess || gpus < 2))
    std::cerr << "nbody.x: NBODY_WEIGHTS / NBODY_TUNE apply to the single-process form with NBODY_GPUS >= 2 only; equal blocks are used" << std::endl;
  const bool weighted = !_multiprocess && gpus >= 2 && (tune || !weights.empty());
  nbx_ctx* ctx = NULL;
  nbx_group* grp = NULL;
  if (_multiprocess) {
    // One process per GPU (the reference's MPI mode, ver5_all/GSimulation.cpp:93-115 + cpu/Compute.cpp:47-58): every
    // rank has drawn the same seed-42 particles above, so nothing needs broadcasting (the reference re-broadcasts nine
    // arrays every step); rank 0's communicator token travels over one TCP round (rendezvous.hpp), then RCCL carries the
    // per-step all-gather.  A world too large for n is refused by every rank before any network is touched.
    int32_t used = 0;
    if (nbx_partition(n, world_size, world_rank, &used, NULL, NULL, NULL, NULL)) die_nbx("nbx_partition");
    if (used != world_size) {
      std::cerr << "nbody.x: " << n << " bodies fill only " << used << " blocks of 256-aligned size; start at most that many ranks (world is "
                << world_size << ")" << std::endl;
      std::exit(1);
    }
    // NBODY_COLLECTIVE_TIMEOUT (seconds; default 120, 0 = off): libnbx's watchdog on the blocking collectives below --
    // a rank whose peer died after the rendezvous ends with status NBX_EXIT_COLLECTIVE_TIMEOUT instead of hanging in
    // ncclCommInitRank or in a window's synchronisation (the reference's MPI mode hangs: ver5_all/GSimulation.cpp:170-214)
    if (const char* ct = std::getenv("NBODY_COLLECTIVE_TIMEOUT")) {
      if (*ct) {  // a number of seconds, 0 = off; anything unparsable is refused: "off" read as 0.0 used to switch the bound off silently
        char* end = NULL;
        const double v = std::strtod(ct, &end);
        while (end && (*end == ' ' || *end == '\t')) ++end;
        if (end == ct || (end && *end)) {
          std::cerr << "nbody.x: NBODY_COLLECTIVE_TIMEOUT=\"" << ct << "\" is not a number of seconds (0 switches the bound off)" << std::endl;
          std::exit(1);
        }
        if (nbx_collective_timeout(v)) die_nbx("nbx_collective_timeout");
      }
    }
    char token[NBX_UNIQUE_ID_BYTES];
    std::memset(token, 0, sizeof token);
    const bool token_ok = !root || nbx_comm_unique_id(token) == NBX_OK;
    if (!token_ok) std::cerr << "nbody.x: nbx_comm_unique_id failed: " << nbx_last_error() << std::endl;
    const int32_t sig[3] = {n, nsteps, kPrecisionBits};
    std::string rerr;
    if (!nbx_rendezvous::exchange(world_rank, world_size, _master_addr, _master_port, sig, token, env_int("NBODY_RENDEZVOUS_TIMEOUT", 120), &rerr,
                                  token_ok)) {
      std::cerr << "nbody.x: rendezvous failed: " << rerr << std::endl;
      std::exit(1);
    }
    const int dev = opts.device >= 0 ? opts.device : _local_rank;  // NBODY_DEVICE, else LOCAL_RANK, else rank % device count
    if (nbx_group_create_rank(&grp, n, kPrecisionBits, world_size, world_rank, token, dev, &opts)) die_nbx("nbx_group_create_rank");
    if (nbx_group_upload(grp, particles->pos_x, particles->pos_y, particles->pos_z, particles->vel_x, particles->vel_y,
                         particles->vel_z, particles->mass))
      die_nbx("nbx_group_upload");
  } else if (weighted) {
    if (nbx_group_create_weighted(&grp, n, kPrecisionBits, gpus, NULL, weights.empty() ? NULL : weights.data(), &opts)) die_nbx("nbx_group_create_weighted");
    if (nbx_group_upload(grp, particles->pos_x, particles->pos_y, particles->pos_z, particles->vel_x, particles->vel_y,
                         particles->vel_z, particles->mass))
      die_nbx("nbx_group_upload");
  } else if (gpus > 1 || std::getenv("NBX_EXCHANGE")) {
    if (nbx_group_create(&grp, n, kPrecisionBits, gpus, NULL, &opts)) die_nbx("nbx_group_create");
    if (nbx_group_upload(grp, particles->pos_x, particles->pos_y, particles->pos_z, particles->vel_x, particles->vel_y,
                         particles->vel_z, particles->mass))
      die_nbx("nbx_group_upload");
  } else {
    if (nbx_create(&ctx, n, kPrecisionBits, &opts)) die_nbx("nbx_create");
    if (nbx_upload(ctx, particles->pos_x, particles->pos_y, particles->pos_z, particles->vel_x, particles->vel_y,
                   particles->vel_z, particles->mass))
      die_nbx("nbx_upload");
  }

  _totTime = 0.;
  const double nd = double(n);
  const double gflops = 1e-9 * ((11. + 18.) * nd * nd + nd * 19.);  // the reference's flop model
  double av = 0.0, dev = 0.0;
  int nf = 0;
  struct Window { int step; double kenergy, seconds; };
  std::vector<Window> windows;  // for NBODY_JSON: every printed row at full precision
  int retunes = 0;              // windows after which the tuner moved the shares

  CPUTime time;
  const double t0 = time.start();
  int done = 0;
  while (done < nsteps) {
    const int todo = (nsteps - done >= sfreq) ? sfreq : nsteps - done;
    const bool printed = (todo == sfreq);
    double ke = 0.0;
    const double w0 = time.start();
    if (grp) {
      double tail_ke = 0.0;  // asking for the energy is what synchronises a group
      if (nbx_group_step(grp, dt, todo, printed ? &ke : &tail_ke)) die_nbx("nbx_group_step");
    } else {
      if (nbx_step(ctx, dt, todo, printed ? &ke : NULL)) die_nbx("nbx_step");
      if (!printed && nbx_sync(ctx)) die_nbx("nbx_sync");
    }
    const double w1 = time.stop();
    done += todo;
    if (!printed) break;
    _kenergy = (real_type)ke;
    nf += 1;
    const double wt = w1 - w0;
    windows.push_back(Window{done, ke, wt});
    if (weighted && tune) {
      // the reference prints its ratio in front of every row of a tuning run (opencl/Compute.cpp:317-319) and then steps it by 0.01;
      // here the line shows device 0's share of the window just timed, and the next window runs on shares re-weighted by measurement
      std::vector<int32_t> cnt((size_t)gpus);
      if (nbx_group_shares(grp, NULL, cnt.data(), NULL)) die_nbx("nbx_group_shares");
      if (root) std::printf("cpu/gpu ratio = %f\n", (double)cnt[0] / (double)n);
      std::fflush(stdout);
      int32_t changed = 0;
      if (done < nsteps && nbx_group_retune(grp, NULL, &changed)) die_nbx("nbx_group_retune");
      retunes += changed;
    }
    if (root)
      std::cout << " " << std::left << std::setw(8) << done << std::left << std::setprecision(5) << std::setw(8)
              << done * get_tstep() << std::left << std::setprecision(5) << std::setw(12) << _kenergy << std::left
              << std::setprecision(5) << std::setw(12) << wt << std::left << std::setprecision(5) << std::setw(12)
              << gflops * sfreq / wt << std::endl;
    if (nf > 2) {
      av += gflops * sfreq / wt;
      dev += gflops * sfreq * gflops * sfreq / (wt * wt);
    }
  }
  const double t1 = time.stop();
  _totTime = (t1 - t0);
  _totFlops = gflops * nsteps;

  av /= (double)(nf - 2);
  dev = std::sqrt(dev / (double)(nf - 2) - av * av);

  nbx_stats_t st;
  int32_t ranks = 1, rccl = 0;
  std::string shares_txt;  // weighted groups: "b0 b1 ..." bodies per device at the end of the run
  // leave particles->* as the reference does after its last step (acc zeroed by the update loop)
  if (grp) {
    if (nbx_group_info(grp, &ranks, &rccl, 0, &st)) die_nbx("nbx_group_info");
    if (weighted) {
      std::vector<int32_t> cnt((size_t)ranks);
      if (nbx_group_shares(grp, NULL, cnt.data(), NULL)) die_nbx("nbx_group_shares");
      for (int r = 0; r < ranks; ++r) shares_txt += (r ? " " : "") + std::to_string(cnt[r]);
    }
    if (nbx_group_download(grp, particles->pos_x, particles->pos_y, particles->pos_z, particles->vel_x, particles->vel_y,
                           particles->vel_z))
      die_nbx("nbx_group_download");
    nbx_group_destroy(grp);
  } else {
    if (nbx_stats(ctx, &st)) die_nbx("nbx_stats");
    if (nbx_download(ctx, particles->pos_x, particles->pos_y, particles->pos_z, particles->vel_x, particles->vel_y,
                     particles->vel_z))
      die_nbx("nbx_download");
    nbx_destroy(ctx);
  }
  init_acc();
  if (!root) return;  // the other ranks hold the same final state; rank 0 reports (ver5_all/GSimulation.cpp:162)
  // NBODY_SNAPSHOT=<file>: keep the final state (the reference drops it at exit)
  if (const char* sp = std::getenv("NBODY_SNAPSHOT")) {
    std::string err;
    if (!nbx_snapshot::save(sp, particles, n, steps_before + nsteps, &err)) {
      std::cerr << "nbody.x: snapshot failed: " << err << std::endl;
      std::exit(1);
    }
  }

  std::cout << std::endl;
  std::cout << "# Number Threads     : " << 1 << std::endl;
  std::cout << "# Total Time (s)     : " << _totTime << std::endl;
  std::cout << "# Average Perfomance : " << av << " +- " << dev << std::endl;
  std::cout << "===============================" << std::endl;
  // extra lines AFTER the reference's footer, so line-wise diffs of the reference part still match
  std::cout << "# Device             : " << st.device_name << " (" << st.cu_count << " CUs), fp" << st.precision
            << ", kernel " << kernel_name(st.kernel_variant)
            << ", " << (st.summation_order == NBX_ORDER_REFERENCE ? "reference-order" : "tree") << " sums"
            << (st.inner_loop == NBX_LOOP_ASM ? (st.bodies_per_lane == 1 && st.kernel_variant == NBX_KERNEL_SGPR ? " (hand-scheduled loop, two j records per packed operation)" : " (hand-scheduled loop)")
                : st.inner_loop == NBX_LOOP_ASM_TS ? " (hand-scheduled loop, time-sliced wave priority)" : st.inner_loop == NBX_LOOP_ASM_PF ? " (hand-scheduled loop, L2 prefetch)" : "") << ", bodies/lane " << st.bodies_per_lane << ", j-split " << st.j_split << ", grid " << st.force_grid_x << "x"
            << st.force_grid_y << std::endl;
  if (weighted)
    std::cout << "# GPUs / shares      : " << ranks << " devices own " << shares_txt << " bodies (" << (tune ? "tuned: " + std::to_string(retunes) + " re-weighting(s) from measured force-kernel times" : "fixed weights")
              << "), one in-place broadcast per owner and step over " << (rccl ? "RCCL" : "device-to-device copies") << std::endl;
  else if (ranks > 1 || _multiprocess)
    std::cout << "# GPUs / ranks       : " << ranks << " x " << st.i_count << " bodies" << (_multiprocess ? " (one process per rank)" : "")
              << ", position all-gather per step over " << (rccl ? "RCCL" : "device-to-device copies") << std::endl;
  if (!split_note.empty()) std::cout << "# Device word        : " << split_note << std::endl;
  // NBODY_JSON=<file>: the same facts as one machine-readable line (does not touch stdout)
  if (const char* jp = std::getenv("NBODY_JSON")) {
    if (FILE* jf = std::fopen(jp, "w")) {
      const double pps = (_totTime > 0) ? nd * nd * nsteps / _totTime : 0.0;
      char avtxt[40];  // fewer than three print windows: the reference's average is nan (nf - 2 <= 0), JSON has no nan
      if (std::isfinite(av)) std::snprintf(avtxt, sizeof avtxt, "%.9g", av);
      else std::snprintf(avtxt, sizeof avtxt, "null");
      std::fprintf(jf,
                   "{\"n\": %d, \"steps\": %d, \"precision\": %d, \"ranks\": %d, \"exchange\": \"%s\", \"total_time_s\": %.9g, "
                   "\"pair_per_s_total\": %.9g, \"gflops_avg_reference_convention\": %s, \"kenergy_last_printed\": %.17g, "
                   "\"kernel\": \"%s\", \"bodies_per_lane\": %d, \"j_split\": %d, \"grid\": [%d, %d], \"device\": \"%s\", "
                   "\"one_process_per_rank\": %s, \"uses_rccl\": %s, \"inner_loop\": %d, \"shares\": \"%s\", \"tuned\": %s, \"retunes\": %d, "
                   "\"device_word_note\": \"%s\", \"windows\": [",
                   n, nsteps, st.precision, (int)ranks, ranks > 1 ? (rccl ? "rccl" : "copy") : "none", _totTime, pps, avtxt,
                   (double)_kenergy, kernel_name(st.kernel_variant),
                   st.bodies_per_lane, st.j_split, st.force_grid_x, st.force_grid_y, st.device_name, _multiprocess ? "true" : "false",
                   rccl ? "true" : "false", (int)st.inner_loop, shares_txt.c_str(), (weighted && tune) ? "true" : "false", retunes, split_note.c_str());
      // every printed row: the step, the energy as computed (fp64 sum of the ranks' partials, before the narrowing to
      // real_type that the table shows) and the window's wall time
      for (size_t k = 0; k < windows.size(); ++k)
        std::fprintf(jf, "%s{\"step\": %d, \"kenergy\": %.17g, \"seconds\": %.9g}", k ? ", " : "", windows[k].step, windows[k].kenergy,
                     windows[k].seconds);
      std::fprintf(jf, "]}\n");
      std::fclose(jf);
    }
  }
  if (nf > 2) {
    const double pairs_per_s = av / 29.0 * 1e9;  // GFlops(29/pair) -> pair/s, integration term ignored
    std::cout << "# Pair rate          : " << pairs_per_s * 1e-9 << " G pair/s = "
              << 100.0 * 20.0 * pairs_per_s / (st.precision == 32 ? 157.3e12 : 78.6e12)
              << " % of the fp" << st.precision << " vector roofline (20 flop/pair)" << std::endl;
  }
}
