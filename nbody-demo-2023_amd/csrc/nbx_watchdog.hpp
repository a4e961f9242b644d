// nbx_watchdog.hpp -- a bound on every blocking collective of a multi-rank group.
//
// The reference's multi-process mode simply hangs when a rank dies: init_mpi / mpi_bcast_all / mpi_gather_acc
// (ver5_all/GSimulation.cpp:93-115,170-214) block in MPI for ever.  RCCL behaves the same way: a rank whose peer is gone
// (or never came) sits inside ncclCommInitRank, or in the stream synchronisation behind an all-gather, with no error and
// no end.  A collective cannot be un-stuck from inside the process, so the bound is on the PROCESS: a host thread
// watches the one blocking call that is armed at a time and, when its deadline passes, writes which call is stuck on
// which rank to stderr and ends the process with _exit(NBX_EXIT_COLLECTIVE_TIMEOUT) -- no re-exec, no retry, no
// destructors (they would block on the same dead communicator).  The job's launcher sees a non-zero status from every
// surviving rank within the timeout instead of a job that never ends.
//
// Header-only and free of HIP so that tests/watchdog_driver.cpp can exercise it on a machine without a GPU.
#pragma once

#include <pthread.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace nbx_detail {

constexpr int kExitCollectiveTimeout = 75;  // = NBX_EXIT_COLLECTIVE_TIMEOUT (include/nbx.h); EX_TEMPFAIL of sysexits.h
constexpr double kDefaultCollectiveTimeout = 120.0;

// Plain pthreads with a CLOCK_MONOTONIC condition variable: the deadline is immune to wall-clock steps, and
// pthread_cond_timedwait is a call ThreadSanitizer understands (libstdc++'s steady-clock wait_until goes through
// pthread_cond_clockwait, which GCC 11's runtime does not intercept; tests/test_watchdog.py runs this under TSan).
class Watchdog {
 public:
  // Leaked on purpose: the watcher thread may still be waiting when static destructors run at exit.
  static Watchdog& instance() {
    static Watchdog* w = new Watchdog();
    return *w;
  }

  // Seconds a blocking collective may take beyond the work known to be queued in front of it; <= 0 switches the
  // watchdog off (a stuck collective then stays stuck, as in the reference).
  void set_timeout(double seconds) {
    Lock lk(mu_);
    timeout_ = seconds;
    configured_ = true;
  }
  double timeout() {
    Lock lk(mu_);
    return timeout_locked();
  }
  void set_identity(int rank, int world) {
    Lock lk(mu_);
    rank_ = rank;
    world_ = world;
  }

  // Arms the deadline `timeout + allowance_s` from now for the call named `what` (a string literal or otherwise
  // outliving the scope).  Scopes nest: only the outermost one arms and disarms.
  class Scope {
   public:
    explicit Scope(const char* what, double allowance_s = 0.0) { Watchdog::instance().arm(what, allowance_s); }
    // a scope that only exists under a run-time condition (e.g. "this group exchanges over RCCL")
    Scope(bool enabled, const char* what, double allowance_s) : on_(enabled) { if (on_) Watchdog::instance().arm(what, allowance_s); }
    ~Scope() { if (on_) Watchdog::instance().disarm(); }
    Scope(const Scope&) = delete;
    Scope& operator=(const Scope&) = delete;

   private:
    bool on_ = true;
  };

 private:
  struct Lock {
    pthread_mutex_t& m;
    explicit Lock(pthread_mutex_t& mu) : m(mu) { pthread_mutex_lock(&m); }
    ~Lock() { pthread_mutex_unlock(&m); }
  };

  Watchdog() {
    pthread_mutex_init(&mu_, nullptr);
    pthread_condattr_t a;
    pthread_condattr_init(&a);
    pthread_condattr_setclock(&a, CLOCK_MONOTONIC);
    pthread_cond_init(&cv_, &a);
    pthread_condattr_destroy(&a);
  }

  static double now_s() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
  }

  double timeout_locked() {
    if (!configured_) {
      configured_ = true;
      timeout_ = kDefaultCollectiveTimeout;
      if (const char* e = std::getenv("NBX_COLLECTIVE_TIMEOUT")) {
        char* end = nullptr;
        const double v = std::strtod(e, &end);
        if (end != e) timeout_ = v;
      }
    }
    return timeout_;
  }

  void arm(const char* what, double allowance_s) {
    Lock lk(mu_);
    if (depth_++ > 0) return;
    const double t = timeout_locked();
    if (!(t > 0.0)) return;
    if (!(allowance_s >= 0.0)) allowance_s = 0.0;
    what_ = what;
    limit_s_ = t + allowance_s;
    armed_at_ = now_s();
    deadline_ = armed_at_ + limit_s_;
    armed_ = true;
    gen_ += 1;
    if (!started_) {
      pthread_t th;
      pthread_attr_t at;
      pthread_attr_init(&at);
      pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
      started_ = pthread_create(&th, &at, &Watchdog::entry, this) == 0;  // no thread, no bound: as with the watchdog off
      pthread_attr_destroy(&at);
    }
    pthread_cond_broadcast(&cv_);
  }

  void disarm() {
    Lock lk(mu_);
    if (--depth_ > 0) return;
    if (!armed_) return;
    armed_ = false;
    gen_ += 1;
    pthread_cond_broadcast(&cv_);
  }

  static void* entry(void* self) {
    static_cast<Watchdog*>(self)->run();
    return nullptr;
  }

  void run() {
    Lock lk(mu_);
    for (;;) {
      while (!armed_) pthread_cond_wait(&cv_, &mu_);
      const unsigned long long gen = gen_;
      timespec until;
      until.tv_sec = (time_t)deadline_;
      until.tv_nsec = (long)((deadline_ - (double)until.tv_sec) * 1e9);
      bool expired = false;
      while (armed_ && gen_ == gen && !expired) expired = pthread_cond_timedwait(&cv_, &mu_, &until) != 0 && now_s() >= deadline_;
      if (!(expired && armed_ && gen_ == gen)) continue;  // finished in time, or another call is armed by now
      // Expired with the same call still armed.  Format into a local buffer, one write(2), _exit: the main thread is
      // blocked inside RCCL / HIP and must not be waited for.
      char msg[640];
      const double waited = now_s() - armed_at_;
      if (world_ > 0)
        std::snprintf(msg, sizeof msg,
                      "libnbx: rank %d of %d has been inside %s for %.0f s (limit %.0f s: NBX_COLLECTIVE_TIMEOUT / nbx_collective_timeout plus the "
                      "work queued in front of it): a peer rank is gone or never arrived.  Ending this process with status %d.\n",
                      rank_, world_, what_, waited, limit_s_, kExitCollectiveTimeout);
      else
        std::snprintf(msg, sizeof msg,
                      "libnbx: %s has not returned for %.0f s (limit %.0f s: NBX_COLLECTIVE_TIMEOUT / nbx_collective_timeout plus the work queued "
                      "in front of it).  Ending this process with status %d.\n",
                      what_, waited, limit_s_, kExitCollectiveTimeout);
      const ssize_t ignored = ::write(2, msg, std::strlen(msg));
      (void)ignored;
      ::_exit(kExitCollectiveTimeout);
    }
  }

  pthread_mutex_t mu_;
  pthread_cond_t cv_;
  bool configured_ = false, started_ = false, armed_ = false;
  double timeout_ = kDefaultCollectiveTimeout, limit_s_ = 0.0, armed_at_ = 0.0, deadline_ = 0.0;
  int depth_ = 0, rank_ = -1, world_ = 0;
  unsigned long long gen_ = 0;
  const char* what_ = "";
};

}  // namespace nbx_detail
