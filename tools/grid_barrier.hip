// grid_barrier.hip -- what does a hand-rolled grid-wide barrier cost on MI355X?  (The question behind a device-resident multi-step
// loop for launch-bound sizes: one hipGraph kernel node per step costs ~4.5 us of dispatch at n = 2048.)
// 256 workgroups x 256 threads, one per CU, all resident.  Each round: __syncthreads, thread 0 releases (agent scope: buffer_wbl2),
// adds 1 to a counter in device memory, spins on it (bounded: a wave that gives up sets an abort flag that everybody polls), acquires
// (buffer_inv), __syncthreads.  Modes: 0 = barrier only, no fences; 1 = with release/acquire fences; 2 = fences + every thread writes 16 B
// before and reads another workgroup's 16 B after (the position exchange of a step).
//   hipcc --offload-arch=gfx950 -O3 -o tools/grid_barrier.x tools/grid_barrier.hip && tools/grid_barrier.x
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr unsigned kMaxSpins = 1u << 20;  // ~1 s at ~1 us per poll: every wave leaves the loop, whatever the others do

__global__ __launch_bounds__(256, 1) void rounds(unsigned* bar, int* abort_flag, int iters, int mode, float4* buf0, float4* buf1,
                                                 unsigned long long* cycles, float4* sink) {
  const unsigned nwg = gridDim.x;
  unsigned target = 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    float4* wr = (it & 1) ? buf1 : buf0;
    if (mode == 2) wr[blockIdx.x * 256 + threadIdx.x] = make_float4((float)it, (float)blockIdx.x, (float)threadIdx.x, 1.f);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (mode >= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      target += nwg;
      unsigned spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > kMaxSpins || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      if (mode >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // uniform enough: everybody sees it within a round
    if (mode == 2) {
      if (mode >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const float4 r = wr[((blockIdx.x + 97) % nwg) * 256 + threadIdx.x];
      acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
  const int wgs = argc > 2 ? std::atoi(argv[2]) : 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  if (wgs <= 0 || wgs > prop.multiProcessorCount) { std::fprintf(stderr, "workgroups must be 1..%d (one per CU: all resident)\n", prop.multiProcessorCount); return 1; }
  unsigned* bar;
  int* abort_flag;
  float4 *b0, *b1, *sink;
  unsigned long long* cyc;
  CK(hipMalloc(&bar, 4));
  CK(hipMalloc(&abort_flag, 4));
  CK(hipMalloc(&b0, (size_t)wgs * 256 * 16));
  CK(hipMalloc(&b1, (size_t)wgs * 256 * 16));
  CK(hipMalloc(&sink, (size_t)wgs * 256 * 16));
  CK(hipMalloc(&cyc, (size_t)wgs * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode <= 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(bar, 0, 4));
      CK(hipMemset(abort_flag, 0, 4));
      CK(hipEventRecord(e0));
      rounds<<<wgs, 256>>>(bar, abort_flag, iters, mode, b0, b1, cyc, sink);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      int ab = 0;
      unsigned count = 0;
      CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&count, bar, 4, hipMemcpyDeviceToHost));
      std::vector<float4> h((size_t)wgs * 256);
      CK(hipMemcpy(h.data(), sink, h.size() * 16, hipMemcpyDeviceToHost));
      // mode 2: thread t of workgroup b read, in round it, what workgroup (b + 97) % wgs wrote in that round: sum of it, of that id, of t
      bool ok = true;
      if (mode == 2 && !ab) {
        const double sum_it = 0.5 * (double)iters * (iters - 1);
        for (int b = 0; b < wgs && ok; ++b)
          for (int t = 0; t < 256; t += 85) {
            const float4 v = h[(size_t)b * 256 + t];
            ok = ok && v.w == (float)iters && v.y == (float)iters * (float)((b + 97) % wgs) && v.z == (float)iters * (float)t && std::abs(v.x - sum_it) < 1e-3 * sum_it + 1;
          }
      }
      std::printf("mode %d rep %d: %d workgroups, %d rounds, kernel %.3f ms = %.3f us per round; counter %u (expected %u); aborted %d%s\n", mode, rep, wgs, iters, ms,
                  1e3 * ms / iters, count, (unsigned)wgs * (unsigned)iters, ab, mode == 2 ? (ok ? "; data seen: ok" : "; data seen: WRONG") : "");
    }
  }
  return 0;
}
