// cpu_time.hpp -- wall-clock helper with the reference's interface (ver7/cpu_time.hpp:30-48:
// class CPUTime, start()/stop() both return "now" in seconds as a double).  The reference sums
// absolute gettimeofday() epochs, which loses microseconds at 1.7e9 s; this one reads the
// monotonic clock relative to the first call, so window times of a few ms stay exact.
#ifndef NBX_HOST_CPU_TIME_HPP
#define NBX_HOST_CPU_TIME_HPP

#include <chrono>

class CPUTime {
 public:
  CPUTime() : origin_(clock::now()) {}
  double start() { return now(); }
  double stop() { return now(); }

 private:
  typedef std::chrono::steady_clock clock;
  clock::time_point origin_;
  double now() const { return std::chrono::duration<double>(clock::now() - origin_).count(); }
};

#endif
