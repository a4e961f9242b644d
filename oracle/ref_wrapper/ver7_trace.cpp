// ver7_trace.cpp -- TEST INFRASTRUCTURE (build container only).
//
// Wrapper translation unit around the reference's UNMODIFIED ver7 source
// (/root/reference/ver7/GSimulation.cpp, pulled in by #include at compile time;
// no reference source is copied into this repository).  It exposes what the
// reference keeps private / prints with 5 digits, so that golden vectors can be
// produced from the reference itself:
//   * per-step kinetic energy with 9 (fp32) / 17 (fp64) significant digits,
//   * the initial SoA arrays (bit-exact pins: CRC-32, fp64 sum, samples),
//   * the state after the last step (tolerance pins).
//
// Build: see oracle/Makefile (target ref).  Output goes to oracle/_ref/ only.
//
// -DREF_F64 builds the fp64 oracle of SURVEY.md 8c variant (B): the same source
// with `float` -> `double` and `sqrtf` -> `sqrt`, while the initial conditions
// remain the fp32-drawn ones.  The latter is done by routing the source's
// std::uniform_real_distribution through a proxy that draws in real fp32 and
// keeps the two products the init functions apply to the draw (`* 1.0e-3f`,
// `n *`) in fp32 as well, so pos/vel/mass are exactly the fp32 arrays widened.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include <stdlib.h>
#include <mm_malloc.h>
#include <omp.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <sys/types.h>

typedef float f32;

#ifdef REF_F64
struct F32Draw {
  f32 v;
  operator double() const { return (double)v; }
};
static inline double operator*(F32Draw a, f32 b) { return (double)(a.v * b); }
static inline double operator*(double a, F32Draw b) { return (double)((f32)a * b.v); }
namespace std {
template <class T>
class f32_proxy_uniform_real_distribution {
 public:
  f32_proxy_uniform_real_distribution(double a, double b) : d_((f32)a, (f32)b) {}
  template <class G>
  F32Draw operator()(G &g) {
    F32Draw r;
    r.v = d_(g);
    return r;
  }

 private:
  std::uniform_real_distribution<f32> d_;
};
}  // namespace std
#define uniform_real_distribution f32_proxy_uniform_real_distribution
#define float double
#define sqrtf sqrt
#define TRACE_PREC 17
#else
#define TRACE_PREC 9
#endif

#define private public
#define setprecision(x) setprecision(TRACE_PREC)
#define setw(x) setw(28)
#include "GSimulation.cpp"
#undef private
#undef setprecision
#undef setw

static uint32_t crc32_ieee(const void *data, size_t len) {
  static uint32_t table[256];
  static bool have = false;
  if (!have) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : (c >> 1);
      table[i] = c;
    }
    have = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  const unsigned char *p = (const unsigned char *)data;
  for (size_t i = 0; i < len; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

static void dump_array(FILE *f, const char *name, const real_type *a, int n, int nsample,
                       bool last) {
  double sum = 0;
  for (int i = 0; i < n; ++i) sum += (double)a[i];
  fprintf(f, "    \"%s\": {\"crc32\": \"%08x\", \"sum\": %.17g, \"last\": %.17g, \"first\": [", name,
          crc32_ieee(a, (size_t)n * sizeof(real_type)), sum, (double)a[n - 1]);
  for (int i = 0; i < nsample && i < n; ++i) fprintf(f, "%s%.17g", i ? ", " : "", (double)a[i]);
  fprintf(f, "]}%s\n", last ? "" : ",");
}

static void dump_state(FILE *f, const char *key, ParticleSoA *p, int n, int nsample, bool last) {
  fprintf(f, "  \"%s\": {\n", key);
  dump_array(f, "pos_x", p->pos_x, n, nsample, false);
  dump_array(f, "pos_y", p->pos_y, n, nsample, false);
  dump_array(f, "pos_z", p->pos_z, n, nsample, false);
  dump_array(f, "vel_x", p->vel_x, n, nsample, false);
  dump_array(f, "vel_y", p->vel_y, n, nsample, false);
  dump_array(f, "vel_z", p->vel_z, n, nsample, false);
  dump_array(f, "mass", p->mass, n, nsample, true);
  fprintf(f, "  }%s\n", last ? "" : ",");
}

// usage: ver7_trace.x <n> <nsteps> <out.json> [nsample] [dt_mode]
//   dt_mode (REF_F64 only): 0 = leave set_tstep(0.1) as the ctor did (double 0.1),
//                           1 = (double)(float)0.1  (the fp32 runs' dt, widened)
int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s n nsteps out.json [nsample] [dt_mode]\n", argv[0]);
    return 2;
  }
  const int n = atoi(argv[1]);
  const int nsteps = atoi(argv[2]);
  const char *out = argv[3];
  const int nsample = argc > 4 ? atoi(argv[4]) : 8;
  const int dt_mode = argc > 5 ? atoi(argv[5]) : 1;

  FILE *f = fopen(out, "w");
  if (!f) return 3;

  std::streambuf *saved = std::cout.rdbuf();
  std::ostringstream sink;
  std::cout.rdbuf(sink.rdbuf());

  // initial state: a zero-step run performs allocation + the four init_* only
  {
    GSimulation sim0;
    sim0.set_number_of_particles(n);
    sim0.set_number_of_steps(0);
    sim0.start();
    fprintf(f, "{\n  \"n\": %d, \"nsteps\": %d, \"precision\": %d, \"threads\": %d,\n", n, nsteps,
            (int)(8 * sizeof(real_type)), omp_get_max_threads());
    dump_state(f, "init", sim0.particles, n, nsample, false);
  }

  sink.str("");
  GSimulation sim;
  sim.set_number_of_particles(n);
  sim.set_number_of_steps(nsteps);
  sim.set_sfreq(1);
#ifdef REF_F64
  if (dt_mode == 1) sim.set_tstep((double)(f32)0.1);
#else
  (void)dt_mode;
#endif
  fprintf(f, "  \"dt\": %.17g,\n", (double)sim.get_tstep());
  sim.start();
  std::cout.rdbuf(saved);

  // rows: " s  s*dt  kenergy  time  gflops" -- keep column 3
  std::vector<std::string> ke, secs;
  std::istringstream in(sink.str());
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ls(line);
    std::string a, b, c, d, e;
    if (!(ls >> a >> b >> c >> d >> e)) continue;
    char *endp = 0;
    long s = strtol(a.c_str(), &endp, 10);
    if (*endp != 0 || s < 1) continue;
    ke.push_back(c);
    secs.push_back(d);  // the reference's own (ts1 - ts0) for this step: both loops, no printing
  }
  fprintf(f, "  \"kenergy\": [");
  for (size_t i = 0; i < ke.size(); ++i) fprintf(f, "%s%s", i ? ", " : "", ke[i].c_str());
  fprintf(f, "],\n");
  fprintf(f, "  \"step_seconds\": [");
  for (size_t i = 0; i < secs.size(); ++i) fprintf(f, "%s%s", i ? ", " : "", secs[i].c_str());
  fprintf(f, "],\n");
  fprintf(f, "  \"kenergy_last_member\": %.17g,\n", (double)sim._kenergy);
  dump_state(f, "final", sim.particles, n, nsample, true);
  fprintf(f, "}\n");
  fclose(f);
  if ((int)ke.size() != nsteps) {
    fprintf(stderr, "parsed %zu rows, expected %d\n", ke.size(), nsteps);
    return 4;
  }
  return 0;
}
