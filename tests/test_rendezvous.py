"""Start-up rendezvous of the one-process-per-GPU drop-in (host/rendezvous.hpp) -- CPU only.

The data path of that mode (ncclCommInitRank + all-gathers) needs one GPU per rank and cannot run here; what can be
checked anywhere is everything before it: the token reaches every rank unchanged, a mis-launched rank is refused instead
of deadlocking the first collective, nobody waits for ever, and nbody.x refuses a world that leaves a rank without bodies.
"""
import os
import socket
import subprocess
import time

import pytest

from conftest import PKG, ROOT

HOST = os.path.join(PKG, "host")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rdv") / "rendezvous_driver.x")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", HOST, os.path.join(ROOT, "tests", "rendezvous_driver.cpp"), "-o", exe, "-lpthread"])
    return exe


def _run(driver, specs):
    """specs: list of argument lists (one per process); returns [(rc, stdout)] in the same order."""
    procs = []
    for k, a in enumerate(specs):
        if k == 1:
            time.sleep(0.3)  # rank 0 listens before the others connect in most runs; late listeners are retried anyway
        procs.append(subprocess.Popen([driver] + [str(x) for x in a], stdout=subprocess.PIPE, text=True))
    return [(p.wait(timeout=60), p.stdout.read().strip()) for p in procs]


def test_token_reaches_every_rank(driver):
    port = _free_port()
    want = "ok " + "".join("%02x" % ((i * 7 + 3) & 255) for i in range(128))
    res = _run(driver, [[r, 4, port, 1000] for r in range(4)])
    assert all(rc == 0 and out == want for rc, out in res), res


def test_ranks_started_before_rank_zero_keep_retrying(driver):
    port = _free_port()
    procs = [subprocess.Popen([driver, str(r), "3", str(port), "1000"], stdout=subprocess.PIPE, text=True) for r in (1, 2)]
    time.sleep(1.0)
    procs.insert(0, subprocess.Popen([driver, "0", "3", str(port), "1000"], stdout=subprocess.PIPE, text=True))
    res = [(p.wait(timeout=60), p.stdout.read().strip()) for p in procs]
    assert all(rc == 0 and out.startswith("ok 030a11") for rc, out in res), res


def test_rank_with_another_problem_size_is_refused(driver):
    port = _free_port()
    res = _run(driver, [[0, 2, port, 1000, 5], [1, 2, port, 2000, 5]])
    assert res[0][0] == 1 and "does not belong to this job" in res[0][1]
    assert res[1][0] == 1 and "refused by rank 0" in res[1][1]


def test_one_verdict_for_the_whole_job(driver):
    """rank 1 is fine and says hello FIRST, rank 2 belongs to another job.  Rank 0 answers nobody before it
    has heard everybody, so rank 1 is refused as well -- it must not be sent into ncclCommInitRank with a token for a
    communicator that can never form."""
    port = _free_port()
    p0 = subprocess.Popen([driver, "0", "3", str(port), "1000", "20"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)
    p1 = subprocess.Popen([driver, "1", "3", str(port), "1000", "20"], stdout=subprocess.PIPE, text=True)
    time.sleep(1.0)
    assert p1.poll() is None  # still waiting for the verdict: nothing was answered early
    t0 = time.time()
    p2 = subprocess.Popen([driver, "2", "3", str(port), "2000", "20"], stdout=subprocess.PIPE, text=True)
    res = [(p.wait(timeout=60), p.stdout.read().strip()) for p in (p0, p1, p2)]
    assert time.time() - t0 < 10
    assert res[0][0] == 1 and "does not belong to this job" in res[0][1]
    assert all(rc == 1 and "refused by rank 0" in out for rc, out in res[1:]), res


def test_a_wrong_hello_takes_no_seat_and_late_ranks_are_still_told(driver):
    """a duplicate (or foreign) hello used to count as a seat, so with world = 3 a duplicate rank 1 + the real rank 1 ended
    the accept loop before rank 2 had connected: rank 2 then found the port closed and retried for its whole timeout.  Now the wrong
    hello refuses the job but takes no seat, and rank 0 keeps listening for a short grace period: rank 2, arriving a second later,
    is refused at once -- long before its own 30 s timeout."""
    port = _free_port()
    p0 = subprocess.Popen([driver, "0", "3", str(port), "1000", "30"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)
    p1 = subprocess.Popen([driver, "1", "3", str(port), "1000", "30"], stdout=subprocess.PIPE, text=True)
    pd = subprocess.Popen([driver, "1", "3", str(port), "1000", "30"], stdout=subprocess.PIPE, text=True)   # the duplicate
    time.sleep(1.0)
    t0 = time.time()
    p2 = subprocess.Popen([driver, "2", "3", str(port), "1000", "30"], stdout=subprocess.PIPE, text=True)   # still on its way
    res = [(p.wait(timeout=60), p.stdout.read().strip()) for p in (p0, p1, pd, p2)]
    assert time.time() - t0 < 8, res
    assert res[0][0] == 1 and "duplicate rank" in res[0][1]
    assert all(rc == 1 and "refused by rank 0" in out for rc, out in res[1:]), res


def test_a_silent_client_costs_one_second_not_ten(driver):
    port = _free_port()
    p0 = subprocess.Popen([driver, "0", "2", str(port), "1000", "30"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.5)
    silent = [socket.create_connection(("127.0.0.1", port), timeout=5) for _ in range(3)]   # connect and say nothing
    t0 = time.time()
    p1 = subprocess.Popen([driver, "1", "2", str(port), "1000", "30"], stdout=subprocess.PIPE, text=True)
    res = [(p.wait(timeout=60), p.stdout.read().strip()) for p in (p0, p1)]
    for c in silent:
        c.close()
    assert all(rc == 0 and out.startswith("ok 030a11") for rc, out in res), res
    assert time.time() - t0 < 8     # three stalled connections in front of the real rank: ~3 s, not ~30


def test_a_stray_connection_on_the_port_does_not_end_the_job(driver):
    """A client that is not a rank (no magic word, or a short write) is dropped; the rendezvous completes."""
    port = _free_port()
    p0 = subprocess.Popen([driver, "0", "2", str(port), "1000", "20"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.5)
    for junk in (b"GET / HTTP/1.0\r\n\r\n" + b"x" * 64, b"\x00\x01"):
        s = socket.create_connection(("127.0.0.1", port), timeout=5)
        s.sendall(junk)
        s.close()
    p1 = subprocess.Popen([driver, "1", "2", str(port), "1000", "20"], stdout=subprocess.PIPE, text=True)
    res = [(p.wait(timeout=60), p.stdout.read().strip()) for p in (p0, p1)]
    assert all(rc == 0 and out.startswith("ok 030a11") for rc, out in res), res


def test_missing_rank_times_out_instead_of_hanging(driver):
    port = _free_port()
    t0 = time.time()
    res = _run(driver, [[0, 3, port, 1000, 2], [1, 3, port, 1000, 2]])  # rank 2 never starts
    assert time.time() - t0 < 30
    assert res[0][0] == 1 and "timed out waiting for 1 rank" in res[0][1]


def test_rank_zero_failure_is_passed_on(driver):
    port = _free_port()
    res = _run(driver, [[0, 3, port, 1000, 10, 0], [1, 3, port, 1000, 10], [2, 3, port, 1000, 10]])
    assert res[0][0] == 1 and all(rc == 1 and "rank 0 could not initialise" in out for rc, out in res[1:]), res


def test_nbody_x_refuses_a_world_with_empty_ranks_on_every_rank():
    exe = os.path.join(HOST, "nbody.x")
    for rank in range(3):  # 300 bodies = 2 blocks of 256: three ranks cannot all own bodies; no network is touched
        p = subprocess.run([exe, "300", "10"], env=dict(os.environ, NBODY_WORLD="3", NBODY_RANK=str(rank)), capture_output=True, text=True, timeout=60)
        assert p.returncode == 1 and "start at most that many ranks" in p.stderr
        assert ("Initialize Gravity Simulation" in p.stdout) == (rank == 0)  # only rank 0 prints


def test_inherited_torchrun_variables_do_not_start_the_multi_process_mode():
    """WORLD_SIZE / RANK are generic (any child of a torchrun worker inherits them).  Without the opt-in a lone
    nbody.x is a one-process run: it prints its banner as rank 0 and fails for the one reason this container offers (no
    HIP device) -- at once, not after waiting two minutes for seven peers.  With NBODY_USE_TORCHRUN_ENV=1 the same
    variables are obeyed: rank 3 of 8 is silent and refuses 300 bodies for 8 ranks."""
    exe = os.path.join(HOST, "nbody.x")
    inherited = dict(os.environ, WORLD_SIZE="8", RANK="3", LOCAL_RANK="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    t0 = time.time()
    p = subprocess.run([exe, "300", "10"], env=inherited, capture_output=True, text=True, timeout=60)
    assert time.time() - t0 < 20
    assert "Initialize Gravity Simulation" in p.stdout and "start at most that many ranks" not in p.stderr
    if not os.path.exists("/dev/kfd"):
        assert p.returncode == 1 and "no HIP device" in p.stderr
    p = subprocess.run([exe, "300", "10"], env=dict(inherited, NBODY_USE_TORCHRUN_ENV="1"), capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and p.stdout == "" and "start at most that many ranks" in p.stderr


def test_bad_world_description_is_reported_by_start_not_by_the_constructor(tmp_path):
    """The constructor never exits the process (an embedding program may build a GSimulation it never starts)."""
    src = tmp_path / "ctor_probe.cpp"
    src.write_text('#include "GSimulation.hpp"\n#include <cstring>\nint main(int argc, char** argv) { GSimulation sim; std::cout << "constructed" << std::endl;\n'
                   '  if (argc > 1 && !std::strcmp(argv[1], "mpi")) sim.init_mpi();\n  if (argc > 1 && !std::strcmp(argv[1], "start")) sim.start();\n  return 0; }\n')
    exe = str(tmp_path / "ctor_probe.x")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", HOST, str(src), os.path.join(HOST, "GSimulation.cpp"), "-o", exe,
                           "-L" + PKG, "-lnbx", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, NBODY_WORLD="2", NBODY_RANK="5")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "constructed" in p.stdout
    for how in ("mpi", "start"):
        p = subprocess.run([exe, how], env=env, capture_output=True, text=True, timeout=60)
        assert p.returncode == 1 and "constructed" in p.stdout and "bad world description (world 2, rank 5)" in p.stderr


def test_init_mpi_shares_follow_the_native_partition(nbx, tmp_path):
    """GSimulation::init_mpi() (the ver5_all surface) reports npp / npp_global = the blocks nbx_partition hands the GPUs."""
    src = tmp_path / "mpi_probe.cpp"
    src.write_text('#include "GSimulation.hpp"\nint main() { GSimulation sim; sim.set_number_of_particles(4099); sim.init_mpi();\n'
                   '  std::cout << sim.world_rank << " " << sim.world_size << " " << sim.npp;\n'
                   '  for (int r = 0; r < sim.world_size; ++r) std::cout << " " << sim.npp_global[r];\n  std::cout << std::endl; return 0; }\n')
    exe = str(tmp_path / "mpi_probe.x")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-DNBX_BANNER_IN_MAIN", "-I", HOST, str(src), os.path.join(HOST, "GSimulation.cpp"), "-o", exe,
                           "-L" + PKG, "-lnbx", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    for world in (1, 3, 8):
        for rank in (0, world - 1):
            out = subprocess.check_output([exe], env=dict(os.environ, NBODY_WORLD=str(world), NBODY_RANK=str(rank)), text=True).split()
            shares = [nbx.partition(4099, world, r)[3] for r in range(world)]
            assert [int(x) for x in out] == [rank, world, shares[rank]] + shares


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU failure mode of the one-process-per-GPU drop-in")
def test_two_process_nbody_x_without_a_gpu_ends_on_both_ranks_at_once():
    """Rank 0 cannot make an RCCL token without a HIP device: it says so, answers rank 1's hello with a refusal, and both
    processes exit 1 within seconds -- nobody waits out the rendezvous timeout, nothing falls back to a CPU path."""
    exe = os.path.join(HOST, "nbody.x")
    port = str(_free_port())
    env = dict(os.environ, NBODY_WORLD="2", NBODY_MASTER_PORT=port, NBODY_RENDEZVOUS_TIMEOUT="60")
    t0 = time.time()
    p1 = subprocess.Popen([exe, "3000", "10"], env=dict(env, NBODY_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    p0 = subprocess.Popen([exe, "3000", "10"], env=dict(env, NBODY_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    o0, e0 = p0.communicate(timeout=90)
    o1, e1 = p1.communicate(timeout=90)
    assert time.time() - t0 < 45
    assert p0.returncode == 1 and "nbx_comm_unique_id failed" in e0 and "told to stop" in e0
    assert p1.returncode == 1 and "rank 0 could not initialise" in e1
    assert "Initialize Gravity Simulation" in o0 and o1 == ""  # only rank 0 prints
