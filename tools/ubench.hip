// ubench.hip -- development harness: VALU issue-rate microbenchmark on gfx950.
// Measures cycles per wave-instruction per SIMD for v_fma_f32, v_pk_fma_f32, v_rsq_f32 and the
// n-body pair mix, at 1..8 waves per SIMD, from s_memtime deltas (clock independent) and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;

// 16 independent instructions per loop iteration
// HALF: only lanes 0..31 of every wave64 execute the loop (EXEC = low half): does a half-populated wave issue faster?
template <int OP, bool HALF = false>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;
  const float m = 1.0000001f, c = 1e-9f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (!HALF || (threadIdx.x & 63) < 32)
  for (int i = 0; i < ITERS; ++i) {
    if constexpr (OP == 0) {  // 16 x v_fma_f32
      asm volatile(
          "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
          "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
          "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
          "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2),
            "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
          : "v"(m), "v"(c));
    } else if constexpr (OP == 1) {  // 8 x v_pk_fma_f32 on register pairs (16 fp32 FMAs)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
      f2 mm = {m, m}, cc = {c, c};
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
          : "v"(mm), "v"(cc));
      a0 = p0.x; b0 = p0.y; a1 = p1.x; b1 = p1.y; a2 = p2.x; b2 = p2.y; a3 = p3.x; b3 = p3.y;
      a4 = p4.x; b4 = p4.y; a5 = p5.x; b5 = p5.y; a6 = p6.x; b6 = p6.y; a7 = p7.x; b7 = p7.y;
    } else if constexpr (OP == 2) {  // 16 x v_rsq_f32
      asm volatile(
          "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n"
          "v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n v_rsq_f32 %8, %8\n v_rsq_f32 %9, %9\n v_rsq_f32 %10, %10\n v_rsq_f32 %11, %11\n"
          "v_rsq_f32 %12, %12\n v_rsq_f32 %13, %13\n v_rsq_f32 %14, %14\n v_rsq_f32 %15, %15\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2),
            "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
    } else if constexpr (OP == 3) {  // pair mix: 12 v_fma_f32 + 1 v_rsq_f32 (+3 more fma to make 16)
      asm volatile(
          "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
          "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_rsq_f32 %6, %6\n v_fma_f32 %7, %7, %16, %17\n"
          "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
          "v_fma_f32 %12, %12, %16, %17\n v_rsq_f32 %13, %13\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2),
            "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
          : "v"(m), "v"(c));
    } else if constexpr (OP == 4) {  // packed pair mix: 6 v_pk_fma_f32 + 1 v_rsq... -> 12 pk + 2 rsq per 14 instr
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5};
      f2 mm = {m, m}, cc = {c, c};
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_rsq_f32 %6, %6\n"
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_rsq_f32 %7, %7\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(a6), "+v"(a7)
          : "v"(mm), "v"(cc));
      a0 = p0.x; b0 = p0.y; a1 = p1.x; b1 = p1.y; a2 = p2.x; b2 = p2.y; a3 = p3.x; b3 = p3.y; a4 = p4.x; b4 = p4.y; a5 = p5.x; b5 = p5.y;
    } else if constexpr (OP == 5) {  // 16 x v_fma_f32 with one SGPR operand
      asm volatile(
          "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
          "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
          "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
          "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2),
            "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
          : "s"(m), "v"(c));
    } else if constexpr (OP == 6) {  // 8 x v_pk_mul_f32 + 8 x v_pk_add_f32
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
      f2 mm = {m, m}, cc = {c, c};
      asm volatile(
          "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
          "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
          "v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %9\n v_pk_add_f32 %3, %3, %9\n"
          "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
          : "v"(mm), "v"(cc));
      a0 = p0.x; b0 = p0.y; a1 = p1.x; b1 = p1.y; a2 = p2.x; b2 = p2.y; a3 = p3.x; b3 = p3.y;
      a4 = p4.x; b4 = p4.y; a5 = p5.x; b5 = p5.y; a6 = p6.x; b6 = p6.y; a7 = p7.x; b7 = p7.y;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  if (r == 12345.678f) out[0] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, bool HALF = false>
void run(const char* name, int instr_per_iter, int flops_per_iter, int cus) {
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD == blocks of 256 per CU
    int blocks = cus * wps;
    CK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4));
    hipLaunchKernelGGL((k<OP, HALF>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<OP, HALF>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    std::vector<unsigned long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    // per SIMD: wps waves each issue ITERS*instr_per_iter instructions in `med` cycles
    double cyc_per_instr = med / ((double)ITERS * instr_per_iter * wps);
    double tflops = (double)blocks * 256 * ITERS * flops_per_iter / (ms * 1e-3) * 1e-12;
    printf("%-26s waves/SIMD %d  cyc/wave-instr/SIMD %6.2f  wall %8.3f ms  %7.1f TFLOP/s  eff.clock %.2f GHz\n", name, wps,
           cyc_per_instr, ms, tflops, med / (ms * 1e-3) * 1e-9);
    CK(hipFree(cyc));
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs %d clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
  run<0>("v_fma_f32 x16", 16, 32, cus);
  run<5>("v_fma_f32 (sgpr src) x16", 16, 32, cus);
  run<1>("v_pk_fma_f32 x8", 8, 32, cus);
  run<6>("v_pk_mul+v_pk_add x16", 16, 32, cus);
  run<2>("v_rsq_f32 x16", 16, 16, cus);
  run<3>("14 fma + 2 rsq", 16, 30, cus);
  run<4>("12 pk_fma + 2 rsq", 14, 50, cus);
  // same streams with only the low 32 lanes active (flops counted for the active half)
  run<4, true>("12 pk_fma + 2 rsq, EXEC lo32", 14, 25, cus);
  run<0, true>("v_fma_f32 x16, EXEC lo32", 16, 16, cus);
  return 0;
}
