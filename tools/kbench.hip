// kbench.hip -- development harness (not shipped in libnbx.so): A/B of force_kernel variants on
// one GPU, interleaved rounds in ONE process (guide rule 24), random data, HIP-event timing.
//   usage: kbench.x [n=262144] [rounds=5] [S1,S2,...]   (every compiled variant x every j-split S)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "kb_common.hpp"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int kTile = 256, kBlock = 256;

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 262144;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  if (n % kTile) { fprintf(stderr, "n must be a multiple of %d\n", kTile); return 2; }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s  CUs %d  clock %d MHz  n %d\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, n);

  std::vector<float4> h(n);
  std::mt19937 g(7);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  for (auto& p : h) { p.x = U(g); p.y = U(g); p.z = U(g); p.w = 6.67259e-11f * n * U(g); }
  float4 *posm, *accp;
  const int maxS = 128;
  CK(hipMalloc(&posm, sizeof(float4) * (n + 16)));
  CK(hipMalloc(&accp, sizeof(float4) * (size_t)n * maxS));
  CK(hipMemcpy(posm, h.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  std::vector<int> splits;
  {
    std::string sl = argc > 3 ? argv[3] : "4,8";
    size_t p0 = 0;
    while (p0 < sl.size()) {
      size_t p1 = sl.find(',', p0);
      if (p1 == std::string::npos) p1 = sl.size();
      splits.push_back(atoi(sl.substr(p0, p1 - p0).c_str()));
      p0 = p1 + 1;
    }
  }
  std::vector<Variant> base, vs;
  reg_noslp(base);
  reg_slp(base);
  if (n % 1024 == 0 && n >= 8192) reg_sym(vs);  // experimental symmetric kernel: S fixed at 1 (its launcher ignores the grid)
  for (auto& b : base)
    for (int S : splits) {
      if (S > n / (b.B < 0 ? 64 : kTile)) continue;
      Variant v = b;
      v.S = S;
      v.name += " S" + std::to_string(S);
      vs.push_back(v);
    }

  // reference result for a cross-check: variant 0
  std::vector<float4> ref(n), got(n);
  auto run = [&](Variant& v, bool timeit) {
    KbArgs a{posm, accp, n, 0};
    const bool ws = v.B < 0;
    const int Bv = ws ? -v.B : v.B;
    const int gran = ws ? 64 : kTile;  // WSPLIT: quarter of jps must be a multiple of 16
    int jps = (n / v.S + gran - 1) / gran * gran;
    a.jps = jps;
    int S = (n + jps - 1) / jps;
    const int ib = ws ? 64 * Bv : kBlock * Bv;
    dim3 grid((n + ib - 1) / ib, S);
    if (timeit) CK(hipEventRecord(e0, st));
    v.launch(a, grid, st);
    if (timeit) {
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms);
    } else {
      CK(hipStreamSynchronize(st));
    }
    return S;
  };
  auto gather = [&](int S, std::vector<float4>& out) {
    std::vector<float4> tmp((size_t)n * S);
    CK(hipMemcpy(tmp.data(), accp, sizeof(float4) * (size_t)n * S, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
      float4 r{0, 0, 0, 0};
      for (int s = 0; s < S; ++s) { r.x += tmp[(size_t)s * n + i].x; r.y += tmp[(size_t)s * n + i].y; r.z += tmp[(size_t)s * n + i].z; }
      out[i] = r;
    }
  };
  {
    int S = run(vs[0], false);
    gather(S, ref);
  }
  double amax = 0;
  for (auto& r : ref) amax = std::max({amax, (double)fabsf(r.x), (double)fabsf(r.y), (double)fabsf(r.z)});
  for (auto& v : vs) {  // warm + correctness
    int S = run(v, false);
    gather(S, got);
    double err = 0;
    for (int i = 0; i < n; ++i)
      err = std::max({err, (double)fabsf(got[i].x - ref[i].x), (double)fabsf(got[i].y - ref[i].y), (double)fabsf(got[i].z - ref[i].z)});
    printf("check %-22s max|da|/|a|inf = %.3e\n", v.name.c_str(), err / amax);
  }
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) run(v, true);

  const double pairs = (double)n * n;
  printf("%-22s %10s %10s %12s %8s\n", "variant", "med ms", "min ms", "Gpair/s", "roof%");
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    double med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    double pps = pairs / (med * 1e-3);
    printf("%-22s %10.3f %10.3f %12.1f %8.1f\n", v.name.c_str(), med, mn, pps * 1e-9, 100.0 * 20.0 * pps / 157.3e12);
  }
  return 0;
}
