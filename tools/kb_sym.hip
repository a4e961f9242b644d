// kb_sym.hip -- EXPERIMENT (development harness only): Newton's-third-law force kernel.
// Round-1 result (profiles/r01_kbench_sym_prototype.txt): correct (agrees with the shipped kernel to 1.5e-6 of |a|inf)
// but SLOWER as compiled: 17.6 ms vs 15.7 ms at n = 262144.  hipcc needs 126-168+ VGPRs for it (3-4 waves/SIMD, or
// scratch spills under a tighter launch bound), where the shipped kernel runs 8 waves/SIMD in 28 VGPRs.  The 1.6x
// arithmetic advantage needs hand register allocation to materialise; starting point for a later round.
// Every unordered pair of an (I,J) tile of the upper triangle is evaluated once and applied to both bodies.
// See DESIGN.md section 10 for the shape; this file measures whether it pays on MI355X.
#include <hip/hip_runtime.h>
#include "kb_common.hpp"

namespace sym {

constexpr int kT = 1024;  // bodies per tile side
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dpp_add(float x, int) { return x; }

template <int CTRL>
__device__ __forceinline__ float dpp_sum(float x) {
  return x + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
// sum over the 8 lanes of an aligned group of 8 (result in every lane of the group)
__device__ __forceinline__ float red8(float x) {
  x = dpp_sum<0xB1>(x);   // quad_perm:[1,0,3,2]
  x = dpp_sum<0x4E>(x);   // quad_perm:[2,3,0,1]
  x = dpp_sum<0x141>(x);  // row_half_mirror
  return x;
}

template <bool SYM>
__device__ __forceinline__ void pair2sym(float xj, float yj, float zj, float gmj, f32x2 xi, f32x2 yi, f32x2 zi, f32x2 gmi,
                                         f32x2& axi, f32x2& ayi, f32x2& azi, float& axj, float& ayj, float& azj) {
  const f32x2 dx = f32x2{xj, xj} - xi, dy = f32x2{yj, yj} - yi, dz = f32x2{zj, zj} - zi;
  const f32x2 e2 = {1.e-3f, 1.e-3f};
  f32x2 r2 = __builtin_elementwise_fma(dz, dz, e2);
  r2 = __builtin_elementwise_fma(dy, dy, r2);
  r2 = __builtin_elementwise_fma(dx, dx, r2);
  f32x2 inv;
  inv.x = __builtin_amdgcn_rsqf(r2.x);
  inv.y = __builtin_amdgcn_rsqf(r2.y);
  const f32x2 inv3 = (inv * inv) * inv;
  const f32x2 fi = f32x2{gmj, gmj} * inv3;
  axi = __builtin_elementwise_fma(dx, fi, axi);
  ayi = __builtin_elementwise_fma(dy, fi, ayi);
  azi = __builtin_elementwise_fma(dz, fi, azi);
  if constexpr (SYM) {
    const f32x2 fj = gmi * inv3;
    axj = __builtin_fmaf(-fj.x, dx.x, axj); axj = __builtin_fmaf(-fj.y, dx.y, axj);
    ayj = __builtin_fmaf(-fj.x, dy.x, ayj); ayj = __builtin_fmaf(-fj.y, dy.y, ayj);
    azj = __builtin_fmaf(-fj.x, dz.x, azj); azj = __builtin_fmaf(-fj.y, dz.y, azj);
  }
}

template <bool SYM, int NP>
__device__ __forceinline__ void tile_body(const float4* __restrict__ posm, int I, int J, float (*accI)[kT], float (*accJ)[kT]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int u = lane & 7, v = lane >> 3;
#pragma unroll 1
  for (int phase = 0; phase < 4; ++phase) {
    const int Iq = wave, Jq = (wave + phase) & 3;
    constexpr int kStrip = NP * 16;
#pragma unroll 1
    for (int strip = 0; strip < 256 / kStrip; ++strip) {
      const int ib = Iq * 256 + strip * kStrip;  // local index of the strip's first body
      f32x2 xi[NP], yi[NP], zi[NP], gi[NP], ax[NP], ay[NP], az[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float4 a = posm[I * kT + ib + (2 * p) * 8 + u];
        const float4 b = posm[I * kT + ib + (2 * p + 1) * 8 + u];
        xi[p] = f32x2{a.x, b.x}; yi[p] = f32x2{a.y, b.y}; zi[p] = f32x2{a.z, b.z}; gi[p] = f32x2{a.w, b.w};
        ax[p] = ay[p] = az[p] = f32x2{0.f, 0.f};
      }
      const float4* jp = posm + J * kT + Jq * 256 + v * 2;
      float4 n0 = jp[0], n1 = jp[1];
#pragma unroll 1
      for (int js = 0; js < 16; ++js) {
        const float4 j0 = n0, j1 = n1;
        if (js + 1 < 16) { n0 = jp[(js + 1) * 16]; n1 = jp[(js + 1) * 16 + 1]; }
        float a0x = 0.f, a0y = 0.f, a0z = 0.f, a1x = 0.f, a1y = 0.f, a1z = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          pair2sym<SYM>(j0.x, j0.y, j0.z, j0.w, xi[p], yi[p], zi[p], gi[p], ax[p], ay[p], az[p], a0x, a0y, a0z);
          pair2sym<SYM>(j1.x, j1.y, j1.z, j1.w, xi[p], yi[p], zi[p], gi[p], ax[p], ay[p], az[p], a1x, a1y, a1z);
          __builtin_amdgcn_sched_barrier(0);  // at most two pair evaluations interleaved: bounds the live temporaries
        }
        if constexpr (SYM) {
          a0x = red8(a0x); a0y = red8(a0y); a0z = red8(a0z);
          a1x = red8(a1x); a1y = red8(a1y); a1z = red8(a1z);
          if (u == 0) {
            const int jl = Jq * 256 + js * 16 + v * 2;
            accJ[0][jl] += a0x; accJ[1][jl] += a0y; accJ[2][jl] += a0z;
            accJ[0][jl + 1] += a1x; accJ[1][jl + 1] += a1y; accJ[2][jl + 1] += a1z;
          }
        }
      }
      // reduce the strip's a_i over v (lanes u, u+8, ..., u+56) and add into the tile's accI
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float sx = ax[p][h], sy = ay[p][h], sz = az[p][h];
#pragma unroll
          for (int m = 8; m < 64; m <<= 1) { sx += __shfl_xor(sx, m, 64); sy += __shfl_xor(sy, m, 64); sz += __shfl_xor(sz, m, 64); }
          if (v == 0) {
            const int il = ib + (2 * p + h) * 8 + u;
            accI[0][il] += sx; accI[1][il] += sy; accI[2][il] += sz;
          }
        }
      }
    }
    __syncthreads();
  }
}

// grid (nb, nb); slab layout [nb][n] float4: slab[k][body] = contribution of block k to `body`
template <int NP, int MINW>
__global__ __launch_bounds__(256, MINW) void sym_force_kernel(const float4* __restrict__ posm, float4* __restrict__ slab, int n) {
  const int I = blockIdx.x, J = blockIdx.y;
  if (J < I) return;
  __shared__ float accI[3][kT];
  __shared__ float accJ[3][kT];
  const int t = threadIdx.x;
  for (int k = t; k < 3 * kT; k += 256) { (&accI[0][0])[k] = 0.f; (&accJ[0][0])[k] = 0.f; }
  __syncthreads();
  if (I == J) tile_body<false, NP>(posm, I, J, accI, accJ);
  else tile_body<true, NP>(posm, I, J, accI, accJ);
  for (int b = t; b < kT; b += 256) {
    slab[(size_t)J * n + I * kT + b] = float4{accI[0][b], accI[1][b], accI[2][b], 0.f};
    if (I != J) slab[(size_t)I * n + J * kT + b] = float4{accJ[0][b], accJ[1][b], accJ[2][b], 0.f};
  }
}

// acc[body] = sum_k slab[k][body]  (what integrate_kernel does with nsplit = nb)
__global__ void slab_sum_kernel(const float4* __restrict__ slab, float4* __restrict__ out, int n, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sx = 0, sy = 0, sz = 0;
  for (int k = 0; k < nb; ++k) { const float4 q = slab[(size_t)k * n + i]; sx += q.x; sy += q.y; sz += q.z; }
  out[i] = float4{sx, sy, sz, 0.f};
}

}  // namespace sym

static float4* g_slab = nullptr;
static size_t g_slab_n = 0;

template <int NP, int MINW>
static void launch_sym(const KbArgs& k, dim3, hipStream_t st) {
  const int nb = k.n / sym::kT;
  if (g_slab_n != (size_t)k.n) {
    if (g_slab) (void)hipFree(g_slab);
    (void)hipMalloc(&g_slab, sizeof(float4) * (size_t)nb * k.n);
    g_slab_n = k.n;
  }
  hipLaunchKernelGGL((sym::sym_force_kernel<NP, MINW>), dim3(nb, nb), dim3(256), 0, st, k.posm, g_slab, k.n);
  hipLaunchKernelGGL(sym::slab_sum_kernel, dim3((k.n + 255) / 256), dim3(256), 0, st, g_slab, k.accp, k.n, nb);
}

void reg_sym(std::vector<Variant>& vs) {
  vs.push_back({"SYM Bi8 Bj2 w3   ", 1, 1, launch_sym<4, 3>, {}});
  vs.push_back({"SYM Bi4 Bj2 w4   ", 1, 1, launch_sym<2, 4>, {}});
}
