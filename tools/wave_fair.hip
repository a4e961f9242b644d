// wave_fair.hip -- when do the waves that share a SIMD finish?  Runs the generated reference-order loop
// (sgpr_loop_asm_b2, nbx_sgpr_loop.inc) at configs[2]'s launch shape (n = 262144 records, 512 workgroups of 256 threads:
// two workgroups per CU, two waves per SIMD) and records, per wave, the constant-rate clock (s_memrealtime, 100 MHz) at
// loop entry and exit together with HW_ID / XCC_ID.  Waves of one SIMD that issue in strict age order finish one after the
// other (the older one at about half the kernel time); fair issue makes both end together.
//   hipcc --offload-arch=gfx950 -O3 -o tools/wave_fair.x tools/wave_fair.hip && tools/wave_fair.x [n] [workgroups] [mode]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../nbody-demo-2023_amd/csrc/nbx_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

using nbx::f32x2;

__global__ __launch_bounds__(256, 1) void probe(const float4* __restrict__ posm, int n, uint64_t* __restrict__ out,
                                                float4* __restrict__ sink, int mode, int chunk, int kbit) {
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 2;
  const float4 p0 = posm[i0 % n], p1 = posm[(i0 + 1) % n];
  f32x2 xi = {p0.x, p1.x}, yi = {p0.y, p1.y}, zi = {p0.z, p1.z};
  f32x2 ax = {0.f, 0.f}, ay = ax, az = ax;
  uint32_t hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
  if (mode == 1 && (hwid & 1)) __builtin_amdgcn_s_setprio(3);  // odd wave slots of a SIMD first
  const uint64_t t0 = wall_clock64();
  if (mode < 2) {
    nbx::sgpr_loop_asm_b2(posm, posm + n, xi, yi, zi, ax, ay, az);
  } else {
    // time-sliced priority: the loop in chunks; before each chunk the wave reads the 100 MHz clock and takes priority 3
    // when bit `kbit` of it equals the parity of its slot on the SIMD, priority 0 otherwise -- so that, of two waves sharing a
    // SIMD, each is the favoured one for half of the time and the other fills its issue bubbles
    const int par = __builtin_amdgcn_readfirstlane((int)(hwid & 1));
    for (int j0 = 0; j0 < n; j0 += chunk) {
      const int phase = (int)((wall_clock64() >> kbit) & 1);
      if (__builtin_amdgcn_readfirstlane(phase) == par) __builtin_amdgcn_s_setprio(3);
      else __builtin_amdgcn_s_setprio(0);
      const int j1 = j0 + chunk < n ? j0 + chunk : n;
      nbx::sgpr_loop_asm_b2(posm + j0, posm + j1, xi, yi, zi, ax, ay, az);
    }
  }
  const uint64_t t1 = wall_clock64();
  __builtin_amdgcn_s_setprio(0);
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    out[4 * w + 0] = t0;
    out[4 * w + 1] = t1;
    out[4 * w + 2] = hwid;
    out[4 * w + 3] = xcc;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = make_float4(ax.x + ax.y, ay.x + ay.y, az.x + az.y, 0.f);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 262144;
  const int wgs = argc > 2 ? std::atoi(argv[2]) : 512;
  const int mode = argc > 3 ? std::atoi(argv[3]) : 0;
  const int chunk = argc > 4 ? std::atoi(argv[4]) : 1024;  // records per priority decision (mode 2); multiple of 64
  const int kbit = argc > 5 ? std::atoi(argv[5]) : 14;      // clock bit that selects the favoured slot parity (2^kbit x 10 ns)
  if (chunk <= 0 || chunk % 64 != 0 || n % chunk != 0) { std::fprintf(stderr, "chunk must divide n and be a multiple of 64\n"); return 1; }
  if (n <= 0 || n % 64 != 0 || wgs <= 0) { std::fprintf(stderr, "n must be a positive multiple of 64\n"); return 1; }
  std::vector<float4> h((size_t)n + nbx::kSgprOverread);
  uint32_t s = 12345u;
  for (int i = 0; i < n; ++i) {
    auto u = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    h[i] = make_float4(u(), u(), u(), 1e-6f * u());
  }
  for (size_t i = n; i < h.size(); ++i) h[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 *d_pos, *d_sink;
  uint64_t* d_out;
  const int waves = wgs * 4;
  CK(hipMalloc(&d_pos, h.size() * sizeof(float4)));
  CK(hipMalloc(&d_sink, (size_t)wgs * 256 * sizeof(float4)));
  CK(hipMalloc(&d_out, (size_t)waves * 4 * sizeof(uint64_t)));
  CK(hipMemcpy(d_pos, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<uint64_t> o((size_t)waves * 4);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    probe<<<wgs, 256>>>(d_pos, n, d_out, d_sink, mode, chunk, kbit);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(o.data(), d_out, o.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t tmin = ~0ull, tmax = 0;
    for (int w = 0; w < waves; ++w) { tmin = std::min(tmin, o[4 * w]); tmax = std::max(tmax, o[4 * w + 1]); }
    const double span = (double)(tmax - tmin);
    // histogram of loop-exit times as a fraction of the span, and of start times
    int hist_end[10] = {0}, hist_start[10] = {0};
    std::map<uint64_t, std::vector<int>> simd;  // (xcc, se, sh, cu, simd) -> waves
    for (int w = 0; w < waves; ++w) {
      const double fe = (double)(o[4 * w + 1] - tmin) / span, fs = (double)(o[4 * w] - tmin) / span;
      hist_end[std::min(9, (int)(fe * 10))]++;
      hist_start[std::min(9, (int)(fs * 10))]++;
      const uint64_t id = o[4 * w + 2], key = (o[4 * w + 3] << 32) | (id & 0xff30u);
      simd[key].push_back(w);
    }
    std::printf("rep %d mode %d chunk %d kbit %d: n %d, %d workgroups, kernel %.3f ms, first entry -> last exit %.3f ms\n", rep, mode, chunk, kbit, n, wgs, ms, span * 1e-5);
    std::printf("  loop entry, tenths of the span:");
    for (int b = 0; b < 10; ++b) std::printf(" %5d", hist_start[b]);
    std::printf("\n  loop exit,  tenths of the span:");
    for (int b = 0; b < 10; ++b) std::printf(" %5d", hist_end[b]);
    std::printf("\n");
    // per SIMD: waves sharing it, their own durations relative to the span
    std::map<size_t, int> per;
    double first_end = 0, last_end = 0, dur_first = 0, dur_last = 0;
    int pairs = 0, same_parity = 0;
    for (auto& kv : simd) {
      per[kv.second.size()]++;
      if (kv.second.size() == 2) {
        int a = kv.second[0], b = kv.second[1];
        if (o[4 * a + 1] > o[4 * b + 1]) std::swap(a, b);
        first_end += (double)(o[4 * a + 1] - tmin) / span;
        last_end += (double)(o[4 * b + 1] - tmin) / span;
        dur_first += (double)(o[4 * a + 1] - o[4 * a]) / span;
        dur_last += (double)(o[4 * b + 1] - o[4 * b]) / span;
        ++pairs;
        if (((o[4 * a + 2] ^ o[4 * b + 2]) & 1) == 0) ++same_parity;
      }
    }
    std::printf("  SIMDs by number of waves they ran:");
    for (auto& kv : per) std::printf("  %zu waves: %d", kv.first, kv.second);
    std::printf("\n");
    if (pairs)
      std::printf("  SIMDs with two waves (%d): the earlier one exits at %.3f of the span (in the loop for %.3f), the later at %.3f (%.3f)\n",
                  pairs, first_end / pairs, dur_first / pairs, last_end / pairs, dur_last / pairs);
    if (pairs) std::printf("  pairs whose two wave slots have the same parity: %d\n", same_parity);
  }
  return 0;
}
