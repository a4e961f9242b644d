#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the GSimulation::start() hot path on MI355X.

A "step" is one time step of ver7/GSimulation.cpp:138-200 over all n bodies: the all-pairs force
kernel, the Euler update and the kinetic-energy partials (plus, at N > 1, the all-gather of the
position blocks).  State is resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload: N == 1 -> BASELINE.json configs[2] (n = 262144, fp32: the roofline configuration the metric's
target is quoted on); N > 1 -> configs[3] (n = 1048576 block-partitioned over the N ranks, one RCCL
all-gather of the positions per step).  Prints ONE JSON line on rank 0.

An N > 1 line validates itself (all outside the timed region): `parity` = the same sharded simulation restarted
from the seed-42 state, 10 steps against the reference's own trace; `ranks` = force-kernel and all-gather times
of every rank (min / mean / max, skew, bytes, devices); and the PRODUCT's own two multi-GPU forms run on the same
GPUs, timed and compared with the torch.distributed figure: `native_single_process` = one nbody.x driving all N
GPUs from one host thread (NBODY_GPUS=N: nbx_group_create -> ncclCommInitAll -> grouped in-place ncclAllGather per
step; `value_native_single_process`), `native_rank_group` = nbody.x per rank (TCP rendezvous, ncclCommInitRank,
in-place ncclAllGather inside libnbx; `value_native_rank_group`).  The N > 1 line also carries both denominators of
a scaling record: `cpu_baseline` (rank 0, before it touches the GPU) and `one_gpu_at_multi_gpu_n` (rank 0's GPU
alone at the same n, the other ranks idle at a barrier).

An N = 1 line carries `multi_gpu_slice_proxy`: what ONE rank of 2 / 4 / 8 computes per step at n = 262144, 524288 and
1048576 on this one GPU, next to the one-GPU step at the same n: the multi-GPU speed-up before communication and skew.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "nbody-demo-2023_amd")
for _p in (PKG,):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
PEAK_FP64_VECTOR_TFLOPS = 78.6
FLOP_PER_PAIR = 20               # 3 sub + 3 FMA + rsqrt + 3 mul + 3 FMA


def usable_cpus():
    """Threads this job may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, p = open(path).read().split()[:2]
            if q != "max":
                n = max(1, min(n, int(float(q) / float(p) + 0.5)))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = max(1, min(n, int(q / p + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(kind, n_gpu, precision, build="pinned"):
    """The reference's CPU path timed on this box's host cores, on a bounded sample.

    kind "reference": oracle/_ref/ver7_trace.x = the reference's unmodified ver7 source compiled in the
    build container (pinned flags), timing itself with its own CPUTime around both loops, one row per
    step.  kind "port": the oracle's C restatement (same loops, same flags) called through ctypes.
    pair/s of this O(n^2) loop does not depend on n beyond cache effects, so the
    sample uses the largest n <= n_gpu whose few steps fit ~10-30 s of CPU time.
    """
    threads = usable_cpus()
    rate_guess = 0.2e9 * threads  # pair/s, from the survey's 1.7 G pair/s on 8 vCPU
    n_cpu = n_gpu
    steps = 3
    while n_cpu > 16384 and steps * float(n_cpu) ** 2 / rate_guess > 25.0:
        n_cpu //= 2
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="spread", OMP_PLACES="threads")
    exe = os.path.join(ROOT, "oracle", "_ref", ("ver7_trace_o3.x" if build == "o3" else "ver7_trace.x") if precision == 32 else "ver7_trace_f64.x")
    under_profiler = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY"))
    if kind == "auto":
        kind = "reference" if (os.path.exists(exe) and not under_profiler) else "port"
    if kind == "reference":
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "cpu.json")
            subprocess.run([exe, str(n_cpu), str(steps), out, "1", "1"], env=env, check=True, timeout=600,
                           stdout=subprocess.DEVNULL)
            secs = json.load(open(out))["step_seconds"]
        t = statistics.median(secs[1:]) if len(secs) > 1 else secs[0]
        what = ("reference ver7 (unmodified source, g++ -O3 -march=x86-64-v3 -fopenmp: best-effort flags, FMA contraction on)" if build == "o3" else
                "reference ver7 (unmodified source, g++ -O2 -fopenmp, built in the build container)")
    else:
        os.environ.setdefault("OMP_NUM_THREADS", str(threads))  # no OMP_PROC_BIND here: it would pin THIS process
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import numpy as np
        import oracle as O
        s = O.init_state(n_cpu)
        if precision == 64:
            s = s.astype(np.float64)
        secs = []
        for _ in range(steps):
            t0 = time.perf_counter()
            O.run(s, 1)
            secs.append(time.perf_counter() - t0)
        t = statistics.median(secs[1:]) if len(secs) > 1 else secs[0]
        what = "oracle C restatement of ver7 (gcc -O2 -fopenmp)"
    return {
        "value": float(n_cpu) ** 2 / t, "unit": "pair/s", "cores": threads, "kind": kind,
        # `cores` = the threads the baseline really ran on (affinity mask capped by the cgroup quota); the box itself has more
        "cores_on_box": os.cpu_count(),
        "sample": "%s, n=%d, %d steps (first discarded), median step %.3f s, fp%d, OMP_NUM_THREADS=%d of %s logical CPUs on the box"
                  % (what, n_cpu, steps, t, precision, threads, os.cpu_count()),
    }


PARITY_FIXTURES = {(262144, 32): "ver7_f32_n262144_s7.json", (16384, 32): "ver7_f32_n16384_s500.json",
                   (2000, 32): "ver7_f32_n2000_s500.json", (1000, 32): "ver7_f32_n1000_s100.json", (262144, 64): "ver7_f64_n262144_s3.json",
                   (16384, 64): "ver7_f64_n16384_s60.json", (1048576, 32): "ver7_f32_n1048576_s10.json"}


def parity_fixture(n, precision):
    """The reference's per-step kenergy trace for this n, if tests/golden/ holds one (made by oracle/gen_golden.py from the
    reference's unmodified ver7 source)."""
    name = PARITY_FIXTURES.get((n, precision))
    if not name:
        return None, None
    return name, json.load(open(os.path.join(ROOT, "tests", "golden", name)))


# a second build of the SAME reference source (oracle/Makefile: -O3 -march=x86-64-v3, FMA contraction on) for the one BASELINE
# configuration that runs deep into the chaotic regime: configs[1], n = 16384 x 500 steps
SECOND_BUILD_FIXTURES = {(16384, 32): "ver7_f32o3_n16384_s500.json"}


def divergence_vs_reference_builds(ke, pinned, second, sfreq=50):
    """The divergence-vs-step curve SURVEY.md 7.2 asks for: this run's per-step kenergy against the pinned build of the
    reference, beside how far a second build of the reference itself is from the pinned one.  All three are rounding-level
    variants of one chaotic system (the cloud bounces at step ~53 when n = 16384), so the honest statement of parity is "no
    farther from the reference than the reference is from itself", step by step -- with the 1e-4 gate applied to the rows
    the program prints (every sfreq-th step: ver7/GSimulation.cpp:203-212)."""
    k = min(len(ke), len(pinned), len(second))
    rel = lambda a, b: [abs(x - y) / abs(y) for x, y in zip(a[:k], b[:k])]
    e_gpu, e_gpu2, spread = rel(ke, pinned), rel(ke, second), rel(second, pinned)
    rows = list(range(sfreq, k + 1, sfreq))
    return {
        "steps": k,
        "printed_rows": {str(s): {"gpu_vs_pinned_build": e_gpu[s - 1], "gpu_vs_second_build": e_gpu2[s - 1],
                                  "second_build_vs_pinned_build": spread[s - 1]} for s in rows},
        "max_over_printed_rows": {"gpu_vs_pinned_build": max([e_gpu[s - 1] for s in rows], default=None),
                                  "second_build_vs_pinned_build": max([spread[s - 1] for s in rows], default=None)},
        "max_over_all_steps": {"gpu_vs_pinned_build": max(e_gpu), "gpu_vs_second_build": max(e_gpu2), "second_build_vs_pinned_build": max(spread)},
        "first_step_above_1e-5": {"gpu_vs_pinned_build": next((i + 1 for i, x in enumerate(e_gpu) if x > 1e-5), None),
                                  "second_build_vs_pinned_build": next((i + 1 for i, x in enumerate(spread) if x > 1e-5), None)},
        "per_step": {"gpu_vs_pinned_build": e_gpu, "second_build_vs_pinned_build": spread},
    }


def parity_probe(nbx, n, precision, full=False):
    """Cheap in-run check against the reference's golden trace when a fixture exists for this n.  `full` (bench.py --bodies
    16384, i.e. BASELINE.json configs[1]): the whole fixture, all 500 steps, with the divergence curve beside the
    reference-vs-reference spread."""
    import numpy as np
    name, g = parity_fixture(n, precision)
    if not name:
        return None
    second = SECOND_BUILD_FIXTURES.get((n, precision)) if full else None
    k = g["nsteps"] if second else min(7, g["nsteps"])
    with nbx.Context(n, precision) as c:
        c.upload(nbx.initial_conditions(n, precision))
        ke = c.step_trace(k)
        st = c.stats()
    ref = np.array(g["kenergy"][:k])
    out = {"fixture": name, "steps": k, "max_rel_kenergy_err": float((abs(ke - ref) / ref).max())}
    if second:
        g2 = json.load(open(os.path.join(ROOT, "tests", "golden", second)))
        d = divergence_vs_reference_builds([float(x) for x in ke], g["kenergy"], g2["kenergy"])
        d.pop("per_step")
        out.update({"second_build_fixture": second, "gate_printed_rows": 1e-4,
                    "pass_printed_rows": d["max_over_printed_rows"]["gpu_vs_pinned_build"] < 1e-4, "divergence": d,
                    "shape": {"kernel": st["kernel_variant"], "bodies_per_lane": st["bodies_per_lane"], "j_split": st["j_split"], "inner_loop": st["inner_loop"]}})
    return out


def parity_probe_sharded(sim, ic, n, precision, rank):
    """N > 1: the SAME sharded simulation the timed region used (same ranks, same collective), restarted from the seed-42
    state and stepped 10 times with the energy all-reduced after every step -- compared on rank 0 with the reference's own
    trace (n = 1048576: 10 steps of the unmodified ver7 binary, tests/golden/ver7_f32_n1048576_s10.json).  Collective: every
    rank calls it.  Returns (record for the JSON line or None, the per-step energies)."""
    name, g = parity_fixture(n, precision)
    k = min(10, g["nsteps"]) if g else 10
    sim.upload(ic)
    ke = []
    for _ in range(k):
        sim.step(1)
        ke.append(sim.kenergy())
    if rank != 0 or not g:
        return None, ke
    ref = g["kenergy"][:k]
    err = [abs(a - b) / abs(b) for a, b in zip(ke, ref)]
    return {"fixture": name, "steps": k, "max_rel_kenergy_err": max(err), "rel_kenergy_err_per_step": err,
            "gate": 1e-4, "pass": max(err) < 1e-4, "ranks": sim.world}, ke


def _reserve_port():
    """A free TCP port and the socket that holds it: bound with SO_REUSEADDR and never listening, so that nbody.x's rendezvous (which
    sets SO_REUSEADDR too) can bind and listen on the same port while nobody else can take it in between (bind-then-close
    was a time-of-check / time-of-use race).  Keep the socket until the children are done."""
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    return s, s.getsockname()[1]


def native_rank_group_check(dist, rank, world, device, n, precision, ke_torch_path, window=10):
    """The drop-in's OWN multi-process path on the same GPUs: one `nbody.x` child per rank (NBODY_WORLD / NBODY_RANK), i.e.
    host/rendezvous.hpp (TCP hand-over of the RCCL token) -> nbx_group_create_rank (ncclCommInitRank) -> one in-place
    ncclAllGather per step inside libnbx (csrc/nbx_group.hip) -> collective energy and download: the counterpart of the
    reference's init_mpi / mpi_bcast_all / mpi_gather_acc (ver5_all/GSimulation.cpp:93-115,170-214).  bench.py's timed
    region exchanges through torch.distributed; without this leg the native path would never run on more than one GPU.

    A child per rank rather than an in-process group: whatever happens in there (an RCCL refusal, the collective watchdog
    ending a stuck rank with status 75) ends the CHILD; this process goes on and prints its line.  Two windows of `window`
    steps: the first carries RCCL's lazy channel set-up, the second is the timing.  Collective: every rank calls it."""
    hosts = [None] * world
    dist.all_gather_object(hosts, socket.gethostname())
    if len(set(hosts)) > 1:
        # the children rendezvous on 127.0.0.1: ranks on other nodes would wait out NBODY_RENDEZVOUS_TIMEOUT for nothing
        return {"skipped": "ranks on %d hosts: this leg starts its children on one node only" % len(set(hosts))} if rank == 0 else None
    hold, p0 = _reserve_port() if rank == 0 else (None, None)
    port = [p0]
    dist.broadcast_object_list(port, src=0)
    exe = os.path.join(PKG, "host", "nbody.x" if precision == 32 else "nbody_fp64.x")
    res = {"rank": rank, "returncode": None}
    with tempfile.TemporaryDirectory() as td:
        jpath = os.path.join(td, "native.json")
        # the child runs the drop-in's DEFAULTS: no knob of the caller's shell leaks into it (NBODY_DEVICE would put every rank
        # on one GPU, NBODY_KERNEL / NBODY_ORDER would change what is compared)
        env = {k: v for k, v in os.environ.items() if not k.startswith("NBODY_") and k not in ("NBX_EXCHANGE", "NBX_SLICE_BIT")}
        env.update(NBODY_WORLD=str(world), NBODY_RANK=str(rank), NBODY_LOCAL_RANK=str(device), NBODY_MASTER_ADDR="127.0.0.1",
                   NBODY_MASTER_PORT=str(port[0]), NBODY_SFREQ=str(window), NBODY_JSON=jpath, NBODY_COLLECTIVE_TIMEOUT="60",
                   NBODY_RENDEZVOUS_TIMEOUT="60")
        t0 = time.perf_counter()
        try:
            p = subprocess.run([exe, str(n), str(2 * window)], env=env, capture_output=True, text=True, timeout=420)
            res["returncode"] = p.returncode
            if p.returncode != 0:
                res["stderr_tail"] = p.stderr[-400:]
            if rank == 0 and p.returncode == 0:
                res["json"] = json.load(open(jpath))
        except Exception as e:  # a child that could not even be started or timed out: reported, never raised
            res["error"] = "%s: %s" % (type(e).__name__, e)
        res["wall_s"] = time.perf_counter() - t0
    allres = [None] * world
    dist.all_gather_object(allres, res)
    if hold is not None:
        hold.close()
    if rank != 0:
        return None
    out = {"binary": os.path.relpath(exe, ROOT), "ranks": world, "steps": 2 * window, "returncodes": [r["returncode"] for r in allres],
           "path": "host/rendezvous.hpp -> nbx_group_create_rank (ncclCommInitRank) -> in-place ncclAllGather per step (csrc/nbx_group.hip)"}
    bad = [r for r in allres if r["returncode"] != 0]
    if bad:
        out["error"] = "; ".join("rank %d: %s" % (r["rank"], r.get("error") or r.get("stderr_tail") or "status %r" % r["returncode"]) for r in bad)
        return out
    try:  # whatever the child wrote, this leg never takes the bench line down with it
        j = allres[0]["json"]
        w = j["windows"]
        out.update({"exchange": j["exchange"], "uses_rccl": j["uses_rccl"], "one_process_per_rank": j["one_process_per_rank"],
                    "ms_per_step": 1e3 * w[1]["seconds"] / window, "ms_per_step_first_window": 1e3 * w[0]["seconds"] / window,
                    "pair_per_s": float(n) * n * window / w[1]["seconds"], "kenergy_step%d" % window: w[0]["kenergy"],
                    "wall_s_max_over_ranks": max(r["wall_s"] for r in allres)})
        if ke_torch_path is not None:
            # same partition, same kernels, the ranks' fp64 energy partials added in rank order (native) or by the all-reduce (torch)
            d = abs(w[0]["kenergy"] - ke_torch_path) / abs(ke_torch_path)
            out["rel_diff_vs_torch_path"] = d
            out["kenergy_equal_to_torch_path"] = bool(d < 1e-12)
        name, g = parity_fixture(n, precision)
        if g and g["nsteps"] >= window:
            out["rel_kenergy_err_vs_reference_step%d" % window] = abs(w[0]["kenergy"] - g["kenergy"][window - 1]) / g["kenergy"][window - 1]
    except Exception as e:
        out["error"] = "could not read the child's report: %s: %s" % (type(e).__name__, e)
    return out


def time_one_gpu(nbx, sharded, n, precision, opts, steps=3):
    """A few whole steps of ONE context that owns all n bodies (the denominator of every multi-GPU speed-up at this n)."""
    big = sharded.ShardedSimulation(n, precision, dist=None, **opts)
    big.upload(nbx.initial_conditions(n, precision))
    big.step(1)
    big.sync()
    tb = time.perf_counter()
    big.step(steps)
    big.sync()
    t = (time.perf_counter() - tb) / steps
    big.close()
    return t


def slice_proxy_cell(nbx, n, P, precision, opts, ic, t_one_gpu, budget_s=0.25):
    """What ONE rank of P computes per step at n bodies -- rank 0's block of the library's partition (nbx_partition) against all
    n_alloc resident records -- timed on this one GPU: the P-GPU step before communication and skew.  NOT a P-GPU measurement."""
    used, block, i_begin, i_count, n_alloc = nbx.partition(n, P, 0)
    with nbx.Context(n, precision, i_begin=i_begin, i_count=i_count, n_alloc=n_alloc, **opts) as c:
        c.upload(ic)
        for _ in range(2):
            c.step_local(); c.commit()
        c.sync()
        tb = time.perf_counter()
        c.step_local(); c.commit()
        c.sync()
        reps = max(3, min(50, int(budget_s / max(time.perf_counter() - tb, 1e-5))))
        c.profile(True)
        tb = time.perf_counter()
        for _ in range(reps):
            c.step_local(); c.commit()
        c.sync()
        t_rank = (time.perf_counter() - tb) / reps
        st = c.stats()
    peak = (PEAK_FP32_VECTOR_TFLOPS if precision == 32 else PEAK_FP64_VECTOR_TFLOPS) * 1e12
    return {"n_bodies": n, "gpus": P, "measured": False, "bodies_owned": i_count, "steps": reps, "ms_per_step": 1e3 * t_rank,
            "force_kernel_ms": st["force_ms_total"] / max(1, st["force_launches_timed"]),
            "roofline_frac": FLOP_PER_PAIR * float(i_count) * float(n) / t_rank / peak,
            "bodies_per_lane": st["bodies_per_lane"], "grid": [st["force_grid_x"], st["force_grid_y"]],
            "kernel": {1: "lds", 2: "sgpr", 3: "sgprw", 6: "jlane"}.get(st["kernel_variant"], "?"), "inner_loop": LOOP_NAMES.get(st["inner_loop"], "?"),
            "one_gpu_ms_per_step": 1e3 * t_one_gpu,
            "implied_%dgpu_speedup_before_communication" % P: t_one_gpu / t_rank,
            "note": "measured on ONE GPU with a %d-body slice; not a %d-GPU measurement" % (i_count, P)}


def native_single_process_check(dist, rank, world, n, precision, ke_torch_path, window=10):
    """The product's single-process multi-GPU form -- the design SURVEY.md section 5 specified -- on the GPUs of this job: rank 0
    alone starts ONE `nbody.x` with NBODY_GPUS=<world> (one host thread, a context and a stream per GPU: nbx_group_create ->
    ncclCommInitAll -> per step every rank's local step, then one grouped in-place ncclAllGather: csrc/nbx_group.hip), the other
    ranks idle at the gather below.  Replaces mpi_bcast_all + mpi_gather_acc of ver5_all/GSimulation.cpp:170-214 without any
    second process.  A child, so that an RCCL refusal or the collective watchdog ends the child and not this line.  Two windows of
    `window` steps: the first carries RCCL's lazy channel set-up, the second is the timing.  With a world of one the child is told
    NBX_EXCHANGE=rccl: a one-device communicator, every collective of the path with one participant (1-GPU rehearsal).
    Collective: every rank calls it."""
    exe = os.path.join(PKG, "host", "nbody.x" if precision == 32 else "nbody_fp64.x")
    res = None
    if rank == 0:
        res = {"binary": os.path.relpath(exe, ROOT), "gpus": world, "steps": 2 * window,
               "path": "NBODY_GPUS=%d: one process, one host thread -> nbx_group_create (ncclCommInitAll) -> nbx_group_step: local steps + one grouped in-place "
                       "ncclAllGather per step (csrc/nbx_group.hip)" % world}
        with tempfile.TemporaryDirectory() as td:
            jpath = os.path.join(td, "single.json")
            env = {k: v for k, v in os.environ.items() if not k.startswith("NBODY_") and k not in ("NBX_EXCHANGE", "NBX_SLICE_BIT", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
            env.update(NBODY_GPUS=str(world), NBODY_SFREQ=str(window), NBODY_JSON=jpath, NBODY_COLLECTIVE_TIMEOUT="60")
            if world == 1:
                env["NBX_EXCHANGE"] = "rccl"
            t0 = time.perf_counter()
            try:
                p = subprocess.run([exe, str(n), str(2 * window)], env=env, capture_output=True, text=True, timeout=420)
                res["returncode"] = p.returncode
                if p.returncode != 0:
                    res["error"] = p.stderr[-400:] or "status %r" % p.returncode
                else:
                    j = json.load(open(jpath))
                    w = j["windows"]
                    res.update({"exchange": j["exchange"], "uses_rccl": j["uses_rccl"], "ranks": j["ranks"], "one_process_per_rank": j["one_process_per_rank"],
                                "ms_per_step": 1e3 * w[1]["seconds"] / window, "ms_per_step_first_window": 1e3 * w[0]["seconds"] / window,
                                "pair_per_s": float(n) * n * window / w[1]["seconds"], "kenergy_step%d" % window: w[0]["kenergy"],
                                "bodies_per_lane": j["bodies_per_lane"], "grid": j["grid"]})
                    if ke_torch_path is not None:
                        d = abs(w[0]["kenergy"] - ke_torch_path) / abs(ke_torch_path)
                        res["rel_diff_vs_torch_path"] = d
                        res["kenergy_equal_to_torch_path"] = bool(d < 1e-12)
                    name, g = parity_fixture(n, precision)
                    if g and g["nsteps"] >= window:
                        res["rel_kenergy_err_vs_reference_step%d" % window] = abs(w[0]["kenergy"] - g["kenergy"][window - 1]) / g["kenergy"][window - 1]
            except Exception as e:  # reported, never raised
                res["error"] = "%s: %s" % (type(e).__name__, e)
            res["wall_s"] = time.perf_counter() - t0
    box = [res]
    dist.broadcast_object_list(box, src=0)  # the other ranks wait here, their GPUs idle, until rank 0's child has finished
    return box[0] if rank == 0 else None


LOOP_NAMES = {1: "cxx", 2: "asm", 3: "asm_ts", 4: "asm_pf"}


def ran_shape(st):
    """The launch shape in the vocabulary of profiles/roofline_traffic.json's `profiled_shape` (tools/roofline_summary.py)."""
    return {"bodies_per_lane": st["bodies_per_lane"], "epilogue": "row" if st["fused_epilogue"] == 1 else "slab", "j_split": st["j_split"],
            "wave_split": st["kernel_variant"] == 3, "loop": LOOP_NAMES.get(st["inner_loop"], "?")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--bodies", dest="n", type=int, default=0,
                    help="override the body count (under torch.distributed.run spell it --bodies: its parser takes --n for an abbreviation of its own options)")
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--cpu-baseline", default="auto", choices=("auto", "reference", "port", "none"))
    ap.add_argument("--bodies-per-lane", type=int, default=0)
    ap.add_argument("--j-split", type=int, default=0)
    ap.add_argument("--kernel", default="auto", choices=("auto", "lds", "sgpr", "sgprw", "jlane"))
    ap.add_argument("--order", default="auto", choices=("auto", "reference", "tree"),
                    help="summation order of a body's pair terms (include/nbx.h): reference = the CPU loop's order")
    a = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL prints a version banner on
    # rank 0 at communicator creation), so file descriptor 1 points at stderr until the line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    n = a.n or (262144 if a.gpus == 1 else 1048576)
    workload = ("BASELINE.json configs[4]: 1xMI355X nPart=262144 fp64 variant" if (n == 262144 and a.gpus == 1 and a.precision == 64) else
                "BASELINE.json configs[2]: 1xMI355X nPart=262144 fp32" if (n == 262144 and a.gpus == 1) else
                "BASELINE.json configs[3]: nPart=1048576 block-partitioned, all-gather(pos) per step" if n == 1048576 else
                "nPart=%d" % n)

    # CPU baseline first: it may start a child process, so it runs before this process touches the GPU
    cpu = cpu_o3 = None
    if rank == 0 and a.cpu_baseline != "none":  # N > 1 too: a scaling record then carries its own CPU denominator
        cpu = cpu_baseline(a.cpu_baseline, n, a.precision)
        # SURVEY.md 8d asks for both builds of the reference: the pinned -O2 one above (the oracle's flags) and a best-effort one
        if cpu["kind"] == "reference" and a.precision == 32 and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ver7_trace_o3.x")):
            cpu_o3 = cpu_baseline("reference", n, a.precision, build="o3")

    import torch
    import torch.distributed as dist
    import nbx
    import sharded

    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    # NBX_BENCH_FORCE_DIST=1 rehearses the RCCL path with a 1-rank group (the only way to run it on a 1-GPU box)
    force_dist = bool(os.environ.get("NBX_BENCH_FORCE_DIST")) and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NBX_BENCH_BACKEND=gloo: rehearsal of the N>1 code path with several ranks sharing one GPU
        backend = os.environ.get("NBX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    opts = dict(bodies_per_lane=a.bodies_per_lane, j_split=a.j_split, summation_order={"auto": 0, "reference": 1, "tree": 2}[a.order],
                kernel_variant={"auto": 0, "lds": 1, "sgpr": 2, "sgprw": 3, "jlane": 6}[a.kernel])

    parity = parity_probe(nbx, n, a.precision, full=bool(a.n)) if (rank == 0 and not use_dist) else None  # N > 1: parity_probe_sharded below

    ic = nbx.initial_conditions(n, a.precision)  # synthetic: the reference's seed-42 generator
    sim = sharded.ShardedSimulation(n, a.precision, dist=dist if use_dist else None, force_collective=force_dist, **opts)
    sim.upload(ic)

    def fence():
        sim.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    sim.step(a.warmup)
    fence()
    sim.engine.ctx.profile(True)  # HIP events around every force launch, on the context's own stream
    t0 = time.perf_counter()
    sim.step(a.steps)
    fence()
    t1 = time.perf_counter()
    elapsed = own_elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = sim.engine.ctx.stats()
    ke = sim.kenergy()

    # N > 1: what every rank did, gathered to rank 0 -- a sub-6x result must be diagnosable from the line alone
    breakdown = native = native_single = same_n_multi = None
    if use_dist:
        # the all-gathers are timed in a pass of their own, AFTER the timed region (two event records per step inside
        # the headline's clock were small but unmeasured): the same number of steps, events on the stream the kernel and the
        # collective are ordered on
        sim.engine.ctx.profile(False)
        sim.profile_exchange(True)
        sim.step(a.steps)
        fence()
        xms = sim.exchange_ms()
        sim.profile_exchange(False)
        report = {"rank": rank, "device": torch.cuda.current_device(), "host": socket.gethostname(), "bodies_owned": st["i_count"],
                  "force_ms_mean": st["force_ms_total"] / max(1, st["force_launches_timed"]), "allgather_ms": xms, "elapsed_s": own_elapsed}
        reports = sharded.gather_rank_reports(dist, report)
        if rank == 0:
            breakdown = sharded.summarise_rank_reports(reports, sim.block * sim.rec)
            breakdown["backend"] = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (rehearsal: ranks share a GPU, exchange staged through the host)")
            breakdown["world_seen"] = dist.get_world_size()
            breakdown["allgather_share_of_step"] = breakdown["allgather_ms_per_step"]["mean"] / (1e3 * elapsed / a.steps)
        # parity of THIS multi-rank form against the reference's own trace, and the native drop-in path on the same GPUs
        parity, ke_trace = parity_probe_sharded(sim, ic, n, a.precision, rank)
        if dist.get_backend() == "nccl" and not os.environ.get("NBX_BENCH_NO_NATIVE"):
            native = native_rank_group_check(dist, rank, dist.get_world_size(), torch.cuda.current_device(), n, a.precision,
                                             ke_trace[9] if len(ke_trace) >= 10 else None)
        elif rank == 0:
            native = {"skipped": "rehearsal backend %s: RCCL refuses several ranks on one device" % dist.get_backend()}
        # the product's single-process form over all the job's GPUs (rank 0 starts it, the others idle), and rank 0's GPU alone
        # at the same n: the denominator of this line's speed-up, measured in the same run
        if dist.get_backend() == "nccl" and not os.environ.get("NBX_BENCH_NO_NATIVE"):
            native_single = native_single_process_check(dist, rank, dist.get_world_size(), n, a.precision,
                                                        ke_trace[9] if len(ke_trace) >= 10 else None)
        elif rank == 0:
            native_single = {"skipped": "rehearsal backend %s: the ranks share one device" % dist.get_backend()}
        if rank == 0 and not os.environ.get("NBX_BENCH_NO_ONE_GPU"):
            t_one = time_one_gpu(nbx, sharded, n, a.precision, opts, steps=3 if n >= 524288 else 10)
            same_n_multi = {"n_bodies": n, "steps": 3 if n >= 524288 else 10, "value": float(n) * n / t_one, "unit": "pair/s", "ms_per_step": 1e3 * t_one,
                            "device": torch.cuda.current_device(), "note": "rank 0's GPU alone, all n bodies in one context, the other ranks idle"}
        dist.barrier()

    # N == 1 only: the multi-GPU runs use configs[3]'s n = 1048576; time a few steps of it on this one GPU as
    # well, so that scaling can also be read at equal n (pair/s is nearly flat in n here, see profiles/r01_sweep_*)
    same_n = None
    rank8 = rank4 = rank2 = None
    other_order = None
    if world == 1 and a.order == "auto" and not a.j_split and a.kernel in ("auto",):
        # the same workload with the OTHER summation order (see DESIGN.md "Summation order"): reference order is what
        # matches CPU ver7's kenergy at this size, tree order is the fastest and the closest to an fp64 sum
        alt = 2 if st["summation_order"] == 1 else 1
        o2 = dict(opts, summation_order=alt)
        s2 = sharded.ShardedSimulation(n, a.precision, dist=None, **o2)
        s2.upload(ic)
        s2.step(a.warmup)
        s2.sync()
        tb = time.perf_counter()
        s2.step(a.steps)
        s2.sync()
        el2 = time.perf_counter() - tb
        st2 = s2.engine.ctx.stats()
        other_order = {"summation_order": {1: "reference", 2: "tree"}[alt], "value": float(n) * n * a.steps / el2, "unit": "pair/s",
                       "roofline_frac": FLOP_PER_PAIR * float(n) * n * a.steps / el2 / ((PEAK_FP32_VECTOR_TFLOPS if a.precision == 32 else PEAK_FP64_VECTOR_TFLOPS) * 1e12),
                       "kernel": {1: "lds", 2: "sgpr", 3: "sgprw", 6: "jlane"}.get(st2["kernel_variant"], "?"), "bodies_per_lane": st2["bodies_per_lane"],
                       "j_split": st2["j_split"]}
        s2.close()
    proxy = None
    if world == 1 and not a.n and a.precision == 32:
        # What ONE rank of the 2-, 4- and 8-GPU forms computes per step at n = 262144, 524288 and 1048576 (configs[3]) -- its block of the
        # partition against all resident records -- on this GPU, next to one GPU stepping all n bodies: the P-GPU speed-up over one GPU at
        # the same n, before communication and skew, is the ratio of the two (at n = 1M, P = 8 the all-gather is 2 MiB sent / 14 MiB
        # received per rank and step).  Reference order gives one chain per owned body, so small slices leave SIMDs idle: these are the
        # cells a multi-GPU table will show, reported here so that no scaling record is needed to see them.
        t_same = {262144: elapsed / a.steps}
        cells = []
        for nn in (262144, 524288, 1048576):
            if nn not in t_same:
                t_same[nn] = time_one_gpu(nbx, sharded, nn, 32, opts, steps=3)
            ic_nn = ic if nn == n else nbx.initial_conditions(nn, 32)
            for P in (2, 4, 8):
                cells.append(slice_proxy_cell(nbx, nn, P, 32, opts, ic_nn, t_same[nn]))
        t_big = t_same[1048576]
        same_n = {"n_bodies": 1048576, "steps": 3, "value": 1048576.0 ** 2 / t_big, "unit": "pair/s", "ms_per_step": 1e3 * t_big}
        proxy = {"what": "one rank's block of n bodies on P GPUs, timed on ONE GPU (plain launches, no exchange); speed-up = one GPU at the same n / that",
                 "one_gpu_ms_per_step": {str(k): 1e3 * v for k, v in t_same.items()}, "cells": cells}
        by = {(c["n_bodies"], c["gpus"]): c for c in cells}
        rank8, rank4, rank2 = by[(1048576, 8)], by[(1048576, 4)], by[(1048576, 2)]

    if rank == 0:
        pairs_per_step = float(n) * float(n)
        value = pairs_per_step * a.steps / elapsed
        peak = PEAK_FP32_VECTOR_TFLOPS if a.precision == 32 else PEAK_FP64_VECTOR_TFLOPS
        mix_ceiling = 0.625 if a.precision == 32 else 1280.0 / (76 * 32)
        launch_ms = st["force_ms_total"] / max(1, st["force_launches_timed"])
        achieved = FLOP_PER_PAIR * st["pairs_per_launch"] / (launch_ms * 1e-3) * 1e-12 if launch_ms > 0 else 0.0
        # algorithmic HBM bytes of ONE force launch of the shape that ran: every record once, plus the owned
        # block's {v,m} read, {v,m} and new position written when the kernel integrates itself (row epilogue), or the S
        # partial-acceleration slabs when a separate integrate kernel follows
        rec = 16 if a.precision == 32 else 32
        own = st["i_count"]
        alg_bytes = float(rec) * n + (3.0 * rec * own if st["fused_epilogue"] == 1 else float(rec) * own * st["j_split"])
        # PMC counters cannot be collected inside this run; `traffic` is the committed result of scripts/profile.sh and is
        # emitted ONLY when that profile was taken on the very launch shape that ran here (else null + the reason)
        traffic = traffic_x1 = None
        traffic_shape = None
        traffic_note = "no profiles/roofline_traffic.json"
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic_shape = tj.get("profiled_shape")
                same_job = tj.get("n") == n and tj.get("gpus", 1) == a.gpus and tj.get("precision", 32) == a.precision
                if same_job and traffic_shape == ran_shape(st):
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_x1 = tj.get("hbm_bytes_per_launch_fetch_x1")
                    traffic_note = "PMC passes of %s on this shape" % tj.get("source", "scripts/profile.sh")
                elif same_job:
                    traffic_note = "the committed PMC profile was taken on another launch shape (%s); this run used %s" % (json.dumps(traffic_shape), json.dumps(ran_shape(st)))
                else:
                    traffic_note = "the committed PMC profile is for n=%s gpus=%s fp%s" % (tj.get("n"), tj.get("gpus", 1), tj.get("precision", 32))
            except Exception as e:
                traffic = None
                traffic_note = "unreadable profiles/roofline_traffic.json: %s" % e
        line = {
            "metric": "pair-interactions/s", "value": value, "unit": "pair/s", "n_gpus": a.gpus,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            # total work does not grow with N: every N > 1 runs configs[3] (n = 1048576 in all); N = 1 runs configs[2], the size
            # the metric is quoted on, and also reports one GPU at n = 1048576 ("one_gpu_at_multi_gpu_n") for an equal-n reading
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if a.precision == 32 else "f64", "data": "synthetic",
            "config": {"workload": workload, "n_bodies": n, "bodies_per_gpu": st["i_count"],
                       "parallelism": "i-block x%d" % a.gpus, "j_tile": st["j_tile"],
                       "bodies_per_lane": st["bodies_per_lane"], "j_split": st["j_split"],
                       "summation_order": {1: "reference", 2: "tree"}.get(st["summation_order"], "?"),
                       "kernel": {1: "lds", 2: "sgpr", 3: "sgprw", 4: "exact", 6: "jlane"}.get(st["kernel_variant"], "?"),
                       "inner_loop": {1: "compiler-scheduled", 2: "hand-scheduled asm" + (", two j records per packed operation" if st["bodies_per_lane"] == 1 and st["kernel_variant"] == 2 else ""),
                                      3: "hand-scheduled asm, time-sliced wave priority", 4: "hand-scheduled asm, L2 prefetch"}.get(st["inner_loop"], "?"),
                       "grid": [st["force_grid_x"], st["force_grid_y"]], "block": st["force_block"]},
            "gflops_reference_convention": 1e-9 * (29.0 * pairs_per_step + 19.0 * n) * a.steps / elapsed,
            "kenergy_after_run": ke,
            "roofline": {"bound": "valu", "kernel": "nbx::force_kernel", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         # PMC passes of scripts/profile.sh, per launch: `traffic` = 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950
                         # correction, calibrated for 16-B/lane vector loads: an upper bound for this kernel's scalar loads),
                         # `traffic_fetch_x1` = FETCH_SIZE + WRITE_SIZE as counted; both far below what 8 TB/s would move
                         "traffic_fetch_x1": traffic_x1, "traffic_profiled_shape": traffic_shape, "traffic_shape_ran": ran_shape(st),
                         "traffic_note": traffic_note,
                         "algorithmic_bytes": alg_bytes, "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                         "algorithmic_GBps": alg_bytes / (launch_ms * 1e-3) * 1e-9 if launch_ms > 0 else None,
                         "flop_per_pair": FLOP_PER_PAIR, "pairs_per_launch": st["pairs_per_launch"],
                         # the most this instruction mix can reach at the spec clock: fp32 12 packed
                         # ops + 2 quarter-rate v_rsq_f32 per 2 pairs = 64 issue cycles for 40 of 64 creditable flop/cycle;
                         # fp64 15 ops + one 16-cycle v_rsq_f64 per pair
                         "mix_ceiling_frac": mix_ceiling, "frac_of_mix_ceiling": achieved / peak / mix_ceiling,
                         "launch_ms_avg": launch_ms, "launches_timed": st["force_launches_timed"],
                         "note": "fp%d vector FMA roofline (north_star: FMA/rsqrt-bound, no MFMA); for fp32 the 157.3 "
                                 "TFLOP/s is also the dense f32 MFMA peak.  HBM is not the bound: see DESIGN.md" % a.precision},
            "device": st["device_name"], "cu_count": st["cu_count"],
        }
        if parity:
            line["parity"] = parity
        if breakdown:
            line["ranks"] = breakdown
        if native:
            line["native_rank_group"] = native
        if same_n:
            line["one_gpu_at_multi_gpu_n"] = same_n
        if same_n_multi:
            line["one_gpu_at_multi_gpu_n"] = same_n_multi
            line["speedup_vs_one_gpu_same_n"] = same_n_multi["ms_per_step"] / (1e3 * elapsed / a.steps)
        if use_dist:
            # the three ways this job can be run, side by side: torch.distributed driving libnbx (the timed region above), the product's
            # single-process group (one host thread, all GPUs) and the product's one-process-per-GPU group
            line["value_torch_distributed"] = value
            line["value_native_single_process"] = (native_single or {}).get("pair_per_s")
            line["value_native_rank_group"] = (native or {}).get("pair_per_s")
        if native_single:
            line["native_single_process"] = native_single
        if proxy:
            line["multi_gpu_slice_proxy"] = proxy
        if rank8:
            line["one_rank_of_8_at_1m"] = rank8
            line["one_rank_of_4_at_1m"] = rank4
            line["one_rank_of_2_at_1m"] = rank2
        if other_order:
            line["other_summation_order"] = other_order
        if cpu:
            line["cpu_baseline"] = cpu
            line["gpu_over_cpu"] = value / cpu["value"]
        if cpu_o3:
            line["cpu_baseline_o3"] = cpu_o3
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)

    sim.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
