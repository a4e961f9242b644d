// watchdog_driver.cpp -- test harness for csrc/nbx_watchdog.hpp (CPU only; the header has no HIP in it).
//   watchdog_driver stuck <timeout_s>                      a "collective" that never returns: the process must end with 75
//   watchdog_driver ok <timeout_s>                         thousands of short scopes, then one of half the timeout: exit 0
//   watchdog_driver off                                    timeout 0: a scope that outlives everything is left alone
//   watchdog_driver allowance <timeout_s> <allowance_s> <sleep_s>   a scope longer than the timeout but inside timeout + allowance
//   watchdog_driver nested <timeout_s>                     the outer scope's deadline holds while inner scopes come and go
//   watchdog_driver enqueue <timeout_s> <rccl 0|1>         the shape of nbx_group_step: a conditional outer scope around the enqueue loop (armed
//                                                          only for groups that exchange over RCCL), the stand-in blocks INSIDE the loop, before
//                                                          the inner scope at the synchronisation point is ever reached
//   watchdog_driver rdv <rank> <world> <port> <timeout_s>  the start-up rendezvous of nbody.x (host/rendezvous.hpp), after which
//                                                          rank 0 enters a collective its peers never join (they exit at once,
//                                                          as a rank that died after the rendezvous)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "nbx_watchdog.hpp"
#include "rendezvous.hpp"

using nbx_detail::Watchdog;

static void block_for_ever() {
  int fds[2];
  if (pipe(fds) != 0) std::exit(3);
  char c;
  for (;;) {
    const ssize_t k = read(fds[0], &c, 1);  // nobody ever writes: what a collective with a dead peer looks like
    (void)k;
  }
}

static void nap(double s) { std::this_thread::sleep_for(std::chrono::duration<double>(s)); }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string mode = argv[1];
  if (mode == "stuck" && argc > 2) {
    Watchdog::instance().set_timeout(std::atof(argv[2]));
    Watchdog::instance().set_identity(0, 2);
    Watchdog::Scope bounded("ncclCommInitRank (test stand-in)");
    block_for_ever();
  }
  if (mode == "ok" && argc > 2) {
    const double t = std::atof(argv[2]);
    Watchdog::instance().set_timeout(t);
    for (int k = 0; k < 20000; ++k) { Watchdog::Scope s("short scope"); }
    { Watchdog::Scope s("half the timeout"); nap(0.5 * t); }
    nap(1.5 * t);  // nothing armed: the watcher must stay quiet however long the process goes on
    std::printf("done\n");
    return 0;
  }
  if (mode == "off") {
    Watchdog::instance().set_timeout(0.0);
    Watchdog::Scope s("unbounded");
    nap(1.0);
    std::printf("done\n");
    return 0;
  }
  if (mode == "allowance" && argc > 4) {
    Watchdog::instance().set_timeout(std::atof(argv[2]));
    Watchdog::Scope s("a long print window", std::atof(argv[3]));
    nap(std::atof(argv[4]));
    std::printf("done\n");
    return 0;
  }
  if (mode == "nested" && argc > 2) {
    Watchdog::instance().set_timeout(std::atof(argv[2]));
    Watchdog::instance().set_identity(1, 4);
    Watchdog::Scope outer("outer call");
    for (;;) { Watchdog::Scope inner("inner call"); nap(0.05); }  // inner scopes must not push the outer deadline back
  }
  if (mode == "enqueue" && argc > 3) {
    const double t = std::atof(argv[2]);
    const bool rccl = std::atoi(argv[3]) != 0;
    Watchdog::instance().set_timeout(t);
    Watchdog::instance().set_identity(1, 2);
    Watchdog::Scope enqueue(rccl, "nbx_group_step (enqueue: local steps + position all-gathers)", 0.0);
    for (int s = 0; s < 3; ++s) {
      if (s == 1) {              // "ncclGroupEnd of the first all-gather with a dead peer" / "a full launch queue"
        if (rccl) block_for_ever();
        nap(2.0 * t);            // a copy-path group has nothing that can wait for a peer: nothing is armed, nothing fires
      }
    }
    { Watchdog::Scope sync("nbx_group_step (stream synchronisation)"); }
    std::printf("done\n");
    return 0;
  }
  if (mode == "rdv" && argc > 5) {
    const int rank = std::atoi(argv[2]), world = std::atoi(argv[3]), port = std::atoi(argv[4]);
    const int32_t sig[3] = {1000, 100, 32};
    char token[nbx_rendezvous::kTokenBytes];
    std::memset(token, rank == 0 ? 7 : 0, sizeof token);
    std::string err;
    if (!nbx_rendezvous::exchange(rank, world, "127.0.0.1", port, sig, token, 20, &err)) {
      std::printf("error %s\n", err.c_str());
      return 1;
    }
    std::printf("rendezvous ok\n");
    std::fflush(stdout);
    if (rank != 0) return 0;  // "dies" right after the rendezvous
    Watchdog::instance().set_timeout(std::atof(argv[5]));
    Watchdog::instance().set_identity(rank, world);
    Watchdog::Scope bounded("ncclCommInitRank (test stand-in)");
    block_for_ever();
  }
  return 2;
}
