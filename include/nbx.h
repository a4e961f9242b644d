/*
 * nbx.h -- C-ABI of libnbx.so: the MI355X (gfx950) implementation of the
 * GSimulation::start() hot path of NTHU-SC/nbody-demo-2023.
 *
 * Drop-in boundary.  In the reference the plug-in point is "a translation unit
 * that defines void GSimulation::start()" (ver5_all/Makefile:104,
 * ver5_all/programming_models/hip/Compute.cpp:65; inline in ver7 at
 * ver7/GSimulation.cpp:96-242).  Everything start() does per time step lives
 * behind the entry points below; the host keeps allocation of the ParticleSoA
 * arrays, the init_* functions, timing and printing (see INTEGRATION.md for the
 * start() a maintainer would write against this header).
 *
 * Conventions: plain C, no C++/torch types; every function returns an int
 * status (NBX_OK == 0, negative = error, text via nbx_last_error()); no
 * exception crosses the boundary.  A context is driven by one host thread at a
 * time.  All host arrays are caller-owned SoA arrays of `n` elements of the
 * context's precision (float for 32, double for 64), exactly the reference's
 * ParticleSoA members (ver7/Particle.hpp:43-58).
 */
#ifndef NBX_H
#define NBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBX_ABI_VERSION 1

enum {
  NBX_OK = 0,
  NBX_ERR_ARG = -1,     /* null pointer, n <= 0, precision not in {32,64}, bad slice ... */
  NBX_ERR_DEVICE = -2,  /* no HIP device / HIP runtime error (text in nbx_last_error) */
  NBX_ERR_STATE = -3,   /* call out of order (e.g. step before upload) */
  NBX_ERR_ALLOC = -4
};

enum { NBX_ORDER_AUTO = 0, NBX_ORDER_REFERENCE = 1, NBX_ORDER_TREE = 2 };
enum { NBX_LOOP_AUTO = 0, NBX_LOOP_CXX = 1, NBX_LOOP_ASM = 2, NBX_LOOP_ASM_TS = 3, NBX_LOOP_ASM_PF = 4 };

/* kernel_variant values */
enum {
  NBX_KERNEL_AUTO = 0,  /* chosen by size and summation order (csrc/nbx_api.hip: auto_shape): tree order -- NBX_KERNEL_JLANE below
                           16384 owned bodies (fp64: up to 12288), NBX_KERNEL_SGPRW from there; reference order (fp32 runs of n > 131072) -- NBX_KERNEL_SGPR
                           with one j range and the row epilogue.  nbx_stats reports what was taken */
  NBX_KERNEL_LDS = 1,   /* j-tile (256 records) staged in LDS, broadcast ds_read_b128 (the north-star design) */
  NBX_KERNEL_SGPR = 2,  /* j records fetched by pipelined wave-uniform scalar loads into SGPRs */
  NBX_KERNEL_SGPRW = 3, /* as SGPR, and the 4 waves of a workgroup share 64*B bodies and split the j range */
  NBX_KERNEL_EXACT = 4, /* validation only.  Measured cost against the default kernel (profiles/r03_exact_mode_cost.txt): 5.3x at
                           n = 1048576, 6.5x at 262144, 7.9x at 131072, 14x at 65536, 27x at 32768, 44x at 16384, 72x at 2000 -- one
                           thread per body is n / 256 workgroups, which fill the 256 CUs only from n = 65536 up.
                           The arithmetic of the reference's pinned build, bit for bit --
                           one thread per body, j strictly ascending, unfused IEEE multiply/add in the source's
                           association, correctly rounded sqrt and divide, G and m multiplied separately.  Positions
                           and velocities then equal the CPU ver7 run exactly (tests compare CRC-32 of whole arrays) */
  NBX_KERNEL_JLANE = 6, /* tree order, launch-bound sizes (auto below 16384 owned bodies; fp64 up to 12288): a wave owns `bodies_per_lane`
                           (2, 4, 8 or, fp32 only, 16) bodies wave-uniformly and its 64 lanes split the j records; lane partials meet in LDS
                           and the wave integrates its bodies itself -- ONE launch per time step, no slabs, no integrate kernel */
  NBX_KERNEL_EXACT_FMA = 5 /* diagnostic: NBX_KERNEL_EXACT with FMA contraction allowed -- what a -march=native / icpc -xAVX2
                           build of the same reference loop computes.  Used to show how far two builds of the REFERENCE
                           drift apart in the chaotic regime (tools/validate_big.py) */
};

typedef struct nbx_ctx nbx_ctx;

/*
 * Options (all zero == defaults).  Replaces the compile-time / argv knobs of the
 * reference's GPU back ends: block size argv[5] (hip/Compute.cpp:133-139), the
 * MPI slice [start,end) (cpu/Compute.cpp:47-58), real_type (types.hpp:21).
 */
typedef struct nbx_opts {
  int32_t struct_size;     /* = sizeof(nbx_opts); 0 is accepted as "this version" */
  int32_t device;          /* HIP device ordinal; -1 = keep the current device */
  void*   stream;          /* hipStream_t to enqueue on when external_stream != 0 (NULL then means the
                              default stream); ignored otherwise: the context creates its own stream */
  int32_t i_begin;         /* first body this context owns (integrates) */
  int32_t i_count;         /* bodies owned; 0 = all n (single-GPU) */
  int32_t n_alloc;         /* length of the device {x,y,z,G*m} array, >= n; 0 = n rounded up to the
                              j-tile.  Ranks of one job pass the same value (= ranks * block). */
  int32_t bodies_per_lane; /* register blocking of i-bodies: 1,2,4,8; 0 = auto (reference order: by a measured cost model of the launch --
                              1 for slices of up to 256 x CUs bodies and wherever 256-body workgroups load the CUs more evenly, else 2 or 4) */
  int32_t j_split;         /* workgroups sharing one i-block, each summing a j-range; 0 = auto.  Explicit splits of NBX_KERNEL_LDS and
                              NBX_KERNEL_SGPR are whole 256-record tiles, of NBX_KERNEL_SGPRW multiples of 32 records */
  int32_t kernel_variant;  /* NBX_KERNEL_* */
  int32_t fused_epilogue;  /* 0 / 1 = integrate inside the force kernel where the shape allows it (j_split == 1 without
                              wave split, and NBX_KERNEL_JLANE); shapes with j-splits always run the separate integrate
                              kernel.  2 = always the separate integrate kernel */
  int32_t use_graph;       /* 0 = auto, 1 = replay multi-step windows from a hipGraph, 2 = plain launches */
  int32_t external_stream; /* 0 = own non-blocking stream; 1 = enqueue everything on `stream` (caller-owned) */
  int32_t summation_order; /* how a body's n pair terms are added up:
                              1 = NBX_ORDER_REFERENCE: one fp32 (fp64) accumulator per body, j strictly ascending, exactly the
                                  reference's loop order (ver7/GSimulation.cpp:156-173).  Reproduces the reference's own
                                  rounding noise, which at n >= 262144 shifts kenergy by 5e-4..1e-3: measured against the
                                  bit-exact NBX_KERNEL_EXACT, kenergy stays within 5e-5 (n=262144 x 200 steps) / 5e-7
                                  (n=1048576 x 100); needs j_split = 1, so parallelism = owned bodies
                              2 = NBX_ORDER_TREE: partial sums per wave and per j-split, added in fixed order: fastest and
                                  ~40x closer to an fp64 sum, but NOT the reference's rounding
                              0 = auto: REFERENCE for fp32 runs of n > 131072 bodies (the noise grows with the length of the
                                  sum, whatever slice of the bodies this context owns), TREE up to there (the two agree
                                  with the reference within the 1e-4 gate: 2e-5 over 500 steps at n = 131072) and for fp64
                                  (its summation noise is ~1e-13, far inside the 1e-10 gate either way) */
  int32_t inner_loop;      /* scheduling of the SGPR kernel's j loop: 0 = NBX_LOOP_AUTO (the hand-scheduled gfx950 loop wherever
                              an instance exists: fp32, kernel_variant SGPR or SGPRW, 2 or 4 bodies per lane, whole trips per wave; with
                              kernel_variant SGPR and ONE body per lane it is the two-j-records-per-packed-operation loop -- what AUTO
                              takes for reference-order slices of up to 256 x CUs bodies, which leave less than one wave per SIMD
                              at two bodies per lane: 46.7 % of the roofline for 65536 of 262144 bodies against 28.3 %), 1 = NBX_LOOP_CXX
                              (always the compiler-scheduled C++ loop), 2 = NBX_LOOP_ASM (fail if no instance fits the shape),
                              3 = NBX_LOOP_ASM_TS (the hand-scheduled loop with time-sliced wave priority: the waves sharing a SIMD
                              take turns as the favoured one instead of running one after the other; single-row SGPR kernel only --
                              AUTO takes it there when the fullest CU holds exactly two workgroups, where it measured +3 ... +4.5 %),
                              4 = NBX_LOOP_ASM_PF (the hand-scheduled loop plus one L2-prefetch load per trip; same kernel scope as
                              ASM_TS; AUTO takes it when the launch leaves one wave per SIMD -- a rank that owns 131072 of 1M
                              bodies, +3.5 % -- and never with more).  All loops perform the same operations in
                              the same order: results are bit-identical */
  int32_t reserved[2];
} nbx_opts;

typedef struct nbx_stats_t {
  int32_t n, n_alloc, i_begin, i_count, precision;
  int32_t bodies_per_lane, j_split, j_tile, kernel_variant;
  int32_t fused_epilogue;      /* 0 separate integrate kernel, 1 integrated by the force kernel itself */
  int32_t summation_order;     /* NBX_ORDER_REFERENCE or NBX_ORDER_TREE actually in use */
  int32_t force_grid_x, force_grid_y, force_block;
  int32_t cu_count, clock_mhz;
  int64_t steps_done;          /* time steps executed since create */
  int64_t force_launches_timed;/* force-kernel launches covered by force_ms_total */
  double  force_ms_total;      /* sum of HIP-event durations of those launches (profiling on) */
  double  pairs_per_launch;    /* i_count * n */
  char    device_name[64];
  int64_t graph_replays;       /* hipGraph launches issued by nbx_step (each covers up to 50 steps) */
  int32_t use_graph;           /* 1 if nbx_step replays windows from a hipGraph */
  int32_t inner_loop;          /* NBX_LOOP_CXX, NBX_LOOP_ASM, NBX_LOOP_ASM_TS or NBX_LOOP_ASM_PF actually in use by the step kernel */
} nbx_stats_t;

/* Text of the last error on the calling thread ("" if none). Never NULL. */
const char* nbx_last_error(void);
int32_t nbx_abi_version(void);

/*
 * Create a context for n bodies at `precision` (32 | 64).  Allocates all device
 * state; replaces hipMalloc x7 + host aligned_alloc of hip/Compute.cpp:71-104.
 */
int nbx_create(nbx_ctx** out, int32_t n, int32_t precision, const nbx_opts* opts);
void nbx_destroy(nbx_ctx* ctx); /* NULL-safe */

/*
 * Host SoA -> device.  Replaces the per-step H2D copies of hip/Compute.cpp:111-119,148-150
 * (done ONCE here; state then stays resident).  All seven arrays have n elements; positions
 * and mass of ALL bodies are read, velocities only of the owned slice.  Packs {x,y,z,G*m} and
 * {vx,vy,vz,m} with G = 6.67259e-11f (ver7/GSimulation.cpp:127).
 */
int nbx_upload(nbx_ctx* ctx, const void* pos_x, const void* pos_y, const void* pos_z,
               const void* vel_x, const void* vel_y, const void* vel_z, const void* mass);

/*
 * nsteps time steps of ver7/GSimulation.cpp:138-200 (acceleration over all pairs, then
 * v += a*dt; x += v*dt; kinetic energy).  Asynchronous unless kenergy_out != NULL, in which
 * case it synchronises and stores _kenergy (= 0.5 * sum m v^2, ver7:200) after the LAST step.
 * Only valid when the context owns all n bodies (no exchange step needed).
 */
int nbx_step(nbx_ctx* ctx, double dt, int32_t nsteps, double* kenergy_out);

/* As nbx_step, but stores _kenergy after EVERY step into ke_trace[0..nsteps) and synchronises. */
int nbx_step_trace(nbx_ctx* ctx, double dt, int32_t nsteps, double* ke_trace);

/*
 * Sharded stepping (one context per GPU, block partition over i as in the reference's MPI
 * slices, cpu/Compute.cpp:47-58, and OpenCL device split, opencl/Compute.cpp:241-255):
 *   nbx_step_local  : forces of the owned slice against all n_alloc resident bodies, then the
 *                     Euler update of the owned slice into the NEXT position buffer.
 *   nbx_exchange_buffer : device pointer / sizes of that NEXT buffer so the caller can run the
 *                     in-place all-gather (RCCL ncclAllGather / torch.distributed) on `stream`.
 *   nbx_commit      : make NEXT current.
 *   nbx_kenergy_partial : sum over owned bodies of m*v^2 after the last local step (synchronises);
 *                     the caller adds the partials over ranks and multiplies by 0.5.
 */
int nbx_step_local(nbx_ctx* ctx, double dt);
int nbx_exchange_buffer(nbx_ctx* ctx, void** dev_ptr, size_t* total_bytes, size_t* own_offset_bytes,
                        size_t* own_bytes);
int nbx_commit(nbx_ctx* ctx);
int nbx_kenergy_partial(nbx_ctx* ctx, double* sum_mv2);

/*
 * Accelerations of the owned bodies at the CURRENT positions, without integrating -- the
 * whole job of the reference's GPU kernels (hip/Compute.cpp:27-62 + D2H :160-162).
 * acc_* are host arrays of n elements; only [i_begin, i_begin+i_count) is written.
 */
int nbx_accel(nbx_ctx* ctx, void* acc_x, void* acc_y, void* acc_z);

int nbx_sync(nbx_ctx* ctx);

/*
 * Device -> host SoA: positions of all n bodies (current buffer), velocities of the owned slice.
 * Any pointer may be NULL to skip that array.  Leaves `particles->*` as the reference's start()
 * leaves them (ver7/GSimulation.cpp:179-198).
 */
int nbx_download(nbx_ctx* ctx, void* pos_x, void* pos_y, void* pos_z, void* vel_x, void* vel_y,
                 void* vel_z);

/*
 * Single-process multi-GPU: one host thread drives `n_ranks` contexts, rank r on devices[r] owning the
 * r-th i-block (block = ceil(n / n_ranks) rounded up to 256 records; the reference's MPI / OpenCL split,
 * ver5_all/GSimulation.cpp:93-115, opencl/Compute.cpp:241-255).  Per time step: every rank's local step on
 * its own stream, then ONE all-gather of the freshly integrated position blocks -- RCCL ncclAllGather
 * (in place, grouped over the ranks; librccl is loaded on first use) when the devices are distinct, peer /
 * device-to-device copies when `devices` repeats an ordinal (logical ranks on one GPU, used by the tests)
 * or when RCCL cannot be loaded -- then commit.  Replaces mpi_bcast_all + mpi_gather_acc
 * (ver5_all/GSimulation.cpp:170-214).  Kinetic energy: rank partials added on the host when asked for.
 * devices == NULL: rank r -> device r % hipGetDeviceCount().  opts (nullable) supplies the launch-shape fields;
 * its slice / stream / device fields are ignored.
 */
typedef struct nbx_group nbx_group;
int nbx_group_create(nbx_group** out, int32_t n, int32_t precision, int32_t n_ranks, const int32_t* devices,
                     const nbx_opts* opts);
void nbx_group_destroy(nbx_group* g);
int nbx_group_upload(nbx_group* g, const void* pos_x, const void* pos_y, const void* pos_z, const void* vel_x,
                     const void* vel_y, const void* vel_z, const void* mass);
int nbx_group_step(nbx_group* g, double dt, int32_t nsteps, double* kenergy_out);
int nbx_group_download(nbx_group* g, void* pos_x, void* pos_y, void* pos_z, void* vel_x, void* vel_y, void* vel_z);
/* ranks actually used (n_ranks is reduced so that no rank is empty), 1 if the exchange runs over RCCL, and the
 * per-rank statistics of rank `rank`.  Any out pointer may be NULL. */
int nbx_group_info(nbx_group* g, int32_t* n_ranks, int32_t* uses_rccl, int32_t rank, nbx_stats_t* rank_stats);

/*
 * The block partition every multi-rank form uses (host arithmetic only, no GPU needed): block = ceil(n / P) rounded up
 * to 256 records, rank r owns [r*block, min(n, (r+1)*block)), all ranks hold n_alloc = P*block records.  P = n_ranks
 * reduced until no rank is empty (*ranks_used); ranks >= P get i_count = 0.  Replaces the reference's
 * `npp = n / world_size (+ n % world_size on rank 0)` (ver5_all/GSimulation.cpp:99-108), which its slice loops apply
 * correctly only when n % world_size == 0 (cpu/Compute.cpp:50-51).  Any out pointer may be NULL.
 */
int nbx_partition(int32_t n, int32_t n_ranks, int32_t rank, int32_t* ranks_used, int32_t* block, int32_t* i_begin,
                  int32_t* i_count, int32_t* n_alloc);

/*
 * Unequal shares -- the GPU-native reading of the reference's co-execution split (ver5_all/main.cpp:40-54 feeds `cpu_ratio` to
 * opencl/Compute.cpp:154-162: a fixed ratio, or "tuning" when it is negative; :241-255 turns it into per-device shares and offsets;
 * :317-321 steps it every print window).  There the two devices are a CPU and a GPU; here they are GPUs that hold clocks up to 10 %
 * apart under this load (one binary: 0.553 ... 0.609 of the roofline over the boxes of the pool), and with equal blocks the slowest
 * device sets every step.
 *   nbx_partition_weighted   : the ceil(n / 256) tiles of 256 records handed out in proportion to `weights` (largest remainder,
 *                              every rank at least one tile; weights NULL = equal).  All ranks hold n_alloc = 256 * tiles records.
 *   nbx_group_create_weighted: nbx_group_create with that partition (single process, k GPUs).  The per-step exchange becomes one
 *                              in-place ncclBroadcast per owner (grouped), or the same peer copies as before; every context times
 *                              its force launches (HIP events), which is what the tuner weighs the devices by.
 *   nbx_group_shares         : who owns what right now, and each rank's mean force-kernel time since the last retune (ms; 0 = none).
 *   nbx_tune_weights         : host arithmetic of the tuner: weight_r = i_count_r / force_ms_r, normalised -- every rank's measured
 *                              rate.  Fed back into nbx_partition_weighted the ranks' times meet within one tile.
 *   nbx_group_retune         : measure (force_ms == NULL) or take the caller's per-rank times, compute new shares and, if any
 *                              boundary moves, carry the state over to contexts with the new slices (once through the host;
 *                              values are copied, never recomputed: in reference summation order the trajectory is the same bit
 *                              for bit whoever owns a body).  *changed = 1 if the shares moved.  Call it between windows, e.g.
 *                              after every nbx_group_step that asked for the energy (nbody.x: every printed row, like the reference).
 *                              A rank's time is not linear in its share (a reference-order launch lasts as long as its fullest
 *                              SIMD: one body beyond a whole number of waves per SIMD costs a whole extra wave), so (1) a move is
 *                              first predicted -- each rank's measured time scaled by the library's own launch-cost table for the
 *                              old and the new share -- and not made unless the slowest rank is expected to gain at least 1 %, and
 *                              (2) every move made is judged by the window after it: if the slowest rank got slower by more than
 *                              1 %, the previous shares come back (*changed = 1) and stay.  The tuner accepts improvements only.
 */
int nbx_partition_weighted(int32_t n, int32_t n_ranks, const double* weights, int32_t rank, int32_t* ranks_used, int32_t* i_begin,
                           int32_t* i_count, int32_t* n_alloc);
int nbx_group_create_weighted(nbx_group** out, int32_t n, int32_t precision, int32_t n_ranks, const int32_t* devices,
                              const double* weights, const nbx_opts* opts);
int nbx_group_shares(nbx_group* g, int32_t* i_begin /* [ranks] */, int32_t* i_count /* [ranks] */, double* force_ms /* [ranks] */);
int nbx_tune_weights(int32_t n_ranks, const int32_t* i_count, const double* force_ms, double* weights_out);
int nbx_group_retune(nbx_group* g, const double* force_ms /* [ranks] or NULL = measured */, int32_t* changed);

/*
 * One process per GPU (the reference's MPI mode: init_mpi + mpi_bcast_all + mpi_gather_acc,
 * ver5_all/GSimulation.cpp:93-115,170-214; slices cpu/Compute.cpp:47-58).  Rank 0 obtains a rendezvous token with
 * nbx_comm_unique_id (ncclGetUniqueId) and ships its NBX_UNIQUE_ID_BYTES bytes to the other ranks by any means (the
 * drop-in uses a TCP socket, host/rendezvous.hpp); every rank then calls nbx_group_create_rank (ncclCommInitRank) with
 * the same n / precision / world and its own rank.  The returned group is driven with the nbx_group_* calls above,
 * which become collective: each process steps its own block, the per-step exchange is one in-place ncclAllGather of the
 * position blocks, kenergy_out is the same number on every rank (8-byte all-gather, summed in rank order) and
 * nbx_group_download leaves the full final state on every rank.  `device` < 0 picks rank % device count.  A world
 * larger than the number of non-empty blocks is refused identically on all ranks.
 */
#define NBX_UNIQUE_ID_BYTES 128
int nbx_comm_unique_id(void* id_out /* NBX_UNIQUE_ID_BYTES bytes */);
int nbx_group_create_rank(nbx_group** out, int32_t n, int32_t precision, int32_t world, int32_t rank, const void* unique_id,
                          int32_t device, const nbx_opts* opts);

/*
 * Bound on every BLOCKING collective of a group.  The reference's MPI mode hangs for ever when a rank dies (init_mpi,
 * mpi_bcast_all, mpi_gather_acc: ver5_all/GSimulation.cpp:93-115,170-214), and so does RCCL: a rank whose peer is gone sits
 * in ncclCommInitRank or in the stream synchronisation behind an all-gather without error.  libnbx watches those calls
 * -- ncclCommInitRank in nbx_group_create_rank, the synchronisation of nbx_group_step when kenergy_out != NULL,
 * nbx_group_download, nbx_group_destroy -- from a host thread: when one of them has not returned `seconds` after it
 * should have (the limit is `seconds` plus four times the measured duration of the steps still queued in front of it; for
 * ncclCommInitRank `seconds` plus 30 -- NBX_RCCL_INIT_ALLOWANCE in the environment -- for RCCL's own set-up, which takes
 * 4-5 s even for a world of one),
 * the thread writes which call is stuck on which rank to stderr and ends the process with
 * _exit(NBX_EXIT_COLLECTIVE_TIMEOUT): a collective cannot be abandoned from inside the process, so the bound is on the
 * process; there is no re-exec and no retry.  Process-wide.  Default: the environment's NBX_COLLECTIVE_TIMEOUT, else 120;
 * seconds <= 0 switches the watchdog off.  nbody.x sets it from NBODY_COLLECTIVE_TIMEOUT.
 */
#define NBX_EXIT_COLLECTIVE_TIMEOUT 75
int nbx_collective_timeout(double seconds);

/*
 * Seed-42 initial conditions of ver7/GSimulation.cpp:45-94, bit-exact and independent of the
 * host's libstdc++: mt19937(42) re-created per array family, libstdc++-11's
 * uniform_real_distribution<float> restated (one 32-bit draw per value).  Host-only (no GPU
 * needed).  precision 32 writes float arrays; 64 writes the SAME fp32-drawn values widened to
 * double (SURVEY.md 8c variant B).  Arrays hold n elements.
 */
int nbx_ic_pos(int32_t n, int32_t precision, void* pos_x, void* pos_y, void* pos_z);
int nbx_ic_vel(int32_t n, int32_t precision, void* vel_x, void* vel_y, void* vel_z);
int nbx_ic_mass(int32_t n, int32_t precision, void* mass);

/* Per-launch HIP-event timing of the force kernel (on the context's stream). */
int nbx_profile(nbx_ctx* ctx, int32_t enable);
int nbx_stats(nbx_ctx* ctx, nbx_stats_t* out);

#ifdef __cplusplus
}
#endif
#endif /* NBX_H */
