/*
 * nbody_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference hot path (NTHU-SC/nbody-demo-2023, ver7):
 * seed-42 initial conditions, all-pairs softened-gravity acceleration,
 * semi-implicit Euler update and kinetic energy.  Written from the behaviour
 * of the reference, not copied from it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product path
 * (libnbx.so / nbody.x) never links or calls it.
 *
 * Parity status: PINNED.  Checked against (a) the reference's own ver7 source
 * compiled in the build container (oracle/_ref, recipe in oracle/Makefile) and
 * (b) the golden vectors that recipe produced (tests/golden/, generator
 * oracle/gen_golden.py).  The reference ships no tests or fixtures of its own.
 *
 * Build (same flags as the pinned reference build, see oracle/Makefile):
 *   gcc -std=c11 -O2 -fopenmp -ffp-contract=off -fPIC -shared
 *
 * Reference anchors are cited per function as ver7/<file>:<lines>.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* MT19937 (ISO C++ [rand.predef] mt19937: w=32 n=624 m=397 r=31             */
/* a=0x9908b0df u=11 d=0xffffffff s=7 b=0x9d2c5680 t=15 c=0xefc60000 l=18    */
/* f=1812433253), the engine behind std::mt19937 gen(42) at                  */
/* ver7/GSimulation.cpp:48,62,87.                                            */
/* ------------------------------------------------------------------------ */
typedef struct {
  uint32_t mt[624];
  int idx;
} orc_mt19937;

static void mt_seed(orc_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

static void mt_refill(orc_mt19937 *g) {
  for (int k = 0; k < 624; ++k) {
    uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
    uint32_t v = g->mt[(k + 397) % 624] ^ (y >> 1);
    if (y & 1u) v ^= 0x9908b0dfu;
    g->mt[k] = v;
  }
  g->idx = 0;
}

static uint32_t mt_next(orc_mt19937 *g) {
  if (g->idx >= 624) mt_refill(g);
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/*
 * std::uniform_real_distribution<float>(a,b)(gen) as libstdc++ 11 evaluates it
 * (the arithmetic behind ver7/GSimulation.cpp:49,63,88):
 * generate_canonical<float,24> takes ONE 32-bit draw, u = float(draw)/2^32 in
 * float arithmetic, clamps u>=1 to nextafterf(1,0), then returns u*(b-a)+a.
 */
static float uniform_f32(orc_mt19937 *g, float a, float b) {
  float sum = (float)mt_next(g) * 1.0f;
  float u = sum / 4294967296.0f;
  if (u >= 1.0f) u = nextafterf(1.0f, 0.0f);
  return u * (b - a) + a;
}

uint32_t orc_mt19937_first(uint32_t seed) {
  orc_mt19937 g;
  mt_seed(&g, seed);
  return mt_next(&g);
}

/* ver7/GSimulation.cpp:45-57 -- pos_{x,y,z}[i] = U(0,1), drawn x,y,z per body */
void orc_init_pos(int n, float *px, float *py, float *pz) {
  orc_mt19937 g;
  mt_seed(&g, 42u);
  for (int i = 0; i < n; ++i) {
    px[i] = uniform_f32(&g, 0.0f, 1.0f);
    py[i] = uniform_f32(&g, 0.0f, 1.0f);
    pz[i] = uniform_f32(&g, 0.0f, 1.0f);
  }
}

/* ver7/GSimulation.cpp:59-71 -- vel = U(-1,1) * 1.0e-3f, fresh generator */
void orc_init_vel(int n, float *vx, float *vy, float *vz) {
  orc_mt19937 g;
  mt_seed(&g, 42u);
  for (int i = 0; i < n; ++i) {
    vx[i] = uniform_f32(&g, -1.0f, 1.0f) * 1.0e-3f;
    vy[i] = uniform_f32(&g, -1.0f, 1.0f) * 1.0e-3f;
    vz[i] = uniform_f32(&g, -1.0f, 1.0f) * 1.0e-3f;
  }
}

/* ver7/GSimulation.cpp:83-94 -- mass[i] = (float)n * U(0,1), fresh generator */
void orc_init_mass(int n, float *mass) {
  orc_mt19937 g;
  mt_seed(&g, 42u);
  float fn = (float)n;
  for (int i = 0; i < n; ++i) mass[i] = fn * uniform_f32(&g, 0.0f, 1.0f);
}

/* ------------------------------------------------------------------------ */
/* fp32 path                                                                 */
/* ------------------------------------------------------------------------ */

/*
 * ver7/GSimulation.cpp:141-177 -- acc_i += sum_j d * G * m_j * inv^3 for the
 * bodies i in [i0,i1) against ALL j in [0,n), j == i included (adds 0).
 * Products associate left to right as in the source (:170-172); the inner
 * loop is an omp simd reduction, so the association of the j-sum is the
 * vectoriser's, as in the reference build.  [i0,i1) restates the slice
 * semantics of ver5_all/programming_models/cpu/Compute.cpp:47-58.
 */
void orc_accel_f32(int n, int i0, int i1, const float *px, const float *py, const float *pz,
                   const float *mass, float *ax, float *ay, float *az) {
  const float softeningSquared = 1.e-3f; /* ver7:126 */
  const float G = 6.67259e-11f;          /* ver7:127 */
#pragma omp parallel for
  for (int i = i0; i < i1; i++) {
    float ax_i = ax[i], ay_i = ay[i], az_i = az[i];
    const float xi = px[i], yi = py[i], zi = pz[i];
#pragma omp simd reduction(+ : ax_i, ay_i, az_i)
    for (int j = 0; j < n; j++) {
      float dx = px[j] - xi;
      float dy = py[j] - yi;
      float dz = pz[j] - zi;
      float distanceSqr = dx * dx + dy * dy + dz * dz + softeningSquared;
      float distanceInv = 1.0f / sqrtf(distanceSqr);
      ax_i += dx * G * mass[j] * distanceInv * distanceInv * distanceInv;
      ay_i += dy * G * mass[j] * distanceInv * distanceInv * distanceInv;
      az_i += dz * G * mass[j] * distanceInv * distanceInv * distanceInv;
    }
    ax[i] = ax_i;
    ay[i] = ay_i;
    az[i] = az_i;
  }
}

/*
 * ver7/GSimulation.cpp:178-200 -- v += a*dt; x += v*dt (updated v); a = 0;
 * energy += m*(vx^2+vy^2+vz^2) in a float accumulator under an OpenMP
 * reduction; returns that float sum (caller applies 0.5 as the reference does).
 */
float orc_integrate_f32(int i0, int i1, float dt, float *px, float *py, float *pz, float *vx,
                        float *vy, float *vz, float *ax, float *ay, float *az,
                        const float *mass) {
  float energy = 0;
#pragma omp parallel for reduction(+ : energy)
  for (int i = i0; i < i1; ++i) {
    vx[i] += ax[i] * dt;
    vy[i] += ay[i] * dt;
    vz[i] += az[i] * dt;
    px[i] += vx[i] * dt;
    py[i] += vy[i] * dt;
    pz[i] += vz[i] * dt;
    ax[i] = 0.f;
    ay[i] = 0.f;
    az[i] = 0.f;
    energy += mass[i] * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
  }
  return energy;
}

/* ver7/GSimulation.cpp:200 -- _kenergy = 0.5 * energy (double product, float store) */
float orc_kenergy_from_sum_f32(float energy) { return (float)(0.5 * (double)energy); }

/*
 * ver7/GSimulation.cpp:138-200 -- nsteps time steps on caller-owned SoA
 * arrays; ke_trace[s-1] (nullable) receives _kenergy after step s.
 */
void orc_run_f32(int n, int nsteps, float dt, float *px, float *py, float *pz, float *vx,
                 float *vy, float *vz, float *ax, float *ay, float *az, const float *mass,
                 float *ke_trace) {
  for (int s = 1; s <= nsteps; ++s) {
    orc_accel_f32(n, 0, n, px, py, pz, mass, ax, ay, az);
    float e = orc_integrate_f32(0, n, dt, px, py, pz, vx, vy, vz, ax, ay, az, mass);
    if (ke_trace) ke_trace[s - 1] = orc_kenergy_from_sum_f32(e);
  }
}

/* ------------------------------------------------------------------------ */
/* fp64 path: the reference with real_type = double (ver7/types.hpp:21),      */
/* sqrtf -> sqrt and the float literals widened, run on the fp32-drawn        */
/* initial conditions (SURVEY.md 8c variant B).                               */
/* ------------------------------------------------------------------------ */
void orc_accel_f64(int n, int i0, int i1, const double *px, const double *py, const double *pz,
                   const double *mass, double *ax, double *ay, double *az) {
  const double softeningSquared = (double)1.e-3f;
  const double G = (double)6.67259e-11f;
#pragma omp parallel for
  for (int i = i0; i < i1; i++) {
    double ax_i = ax[i], ay_i = ay[i], az_i = az[i];
    const double xi = px[i], yi = py[i], zi = pz[i];
#pragma omp simd reduction(+ : ax_i, ay_i, az_i)
    for (int j = 0; j < n; j++) {
      double dx = px[j] - xi;
      double dy = py[j] - yi;
      double dz = pz[j] - zi;
      double distanceSqr = dx * dx + dy * dy + dz * dz + softeningSquared;
      double distanceInv = 1.0 / sqrt(distanceSqr);
      ax_i += dx * G * mass[j] * distanceInv * distanceInv * distanceInv;
      ay_i += dy * G * mass[j] * distanceInv * distanceInv * distanceInv;
      az_i += dz * G * mass[j] * distanceInv * distanceInv * distanceInv;
    }
    ax[i] = ax_i;
    ay[i] = ay_i;
    az[i] = az_i;
  }
}

double orc_integrate_f64(int i0, int i1, double dt, double *px, double *py, double *pz,
                         double *vx, double *vy, double *vz, double *ax, double *ay, double *az,
                         const double *mass) {
  double energy = 0;
#pragma omp parallel for reduction(+ : energy)
  for (int i = i0; i < i1; ++i) {
    vx[i] += ax[i] * dt;
    vy[i] += ay[i] * dt;
    vz[i] += az[i] * dt;
    px[i] += vx[i] * dt;
    py[i] += vy[i] * dt;
    pz[i] += vz[i] * dt;
    ax[i] = 0.;
    ay[i] = 0.;
    az[i] = 0.;
    energy += mass[i] * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
  }
  return energy;
}

void orc_run_f64(int n, int nsteps, double dt, double *px, double *py, double *pz, double *vx,
                 double *vy, double *vz, double *ax, double *ay, double *az, const double *mass,
                 double *ke_trace) {
  for (int s = 1; s <= nsteps; ++s) {
    orc_accel_f64(n, 0, n, px, py, pz, mass, ax, ay, az);
    double e = orc_integrate_f64(0, n, dt, px, py, pz, vx, vy, vz, ax, ay, az, mass);
    if (ke_trace) ke_trace[s - 1] = 0.5 * e;
  }
}

/* (float)0.1 -- the value `real_type _tstep` holds after set_tstep(0.1), ver7:30 */
float orc_default_dt_f32(void) { return (float)0.1; }
