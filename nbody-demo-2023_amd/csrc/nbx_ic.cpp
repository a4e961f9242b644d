// nbx_ic.cpp -- host-side initial conditions of the reference (ver7/GSimulation.cpp:45-94).
//
// The reference draws from std::mt19937 gen(42) through std::uniform_real_distribution<float>.
// mt19937 is fixed by the C++ standard; the distribution is implementation-defined, so the
// libstdc++-11 arithmetic the reference was built with is written out here:
// one 32-bit draw d per value, u = float(d) / 2^32 evaluated in float, u >= 1 clamped to the
// largest float below 1, result u * (b - a) + a in float.
#include <cmath>
#include <cstdint>

#include "../../include/nbx.h"

namespace {

class Mt19937 {
 public:
  explicit Mt19937(uint32_t seed) : pos_(kN) {
    s_[0] = seed;
    for (uint32_t i = 1; i < kN; ++i) s_[i] = 1812433253u * (s_[i - 1] ^ (s_[i - 1] >> 30)) + i;
  }
  uint32_t operator()() {
    if (pos_ == kN) twist();
    uint32_t y = s_[pos_++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    return y ^ (y >> 18);
  }

 private:
  static const uint32_t kN = 624, kM = 397;
  uint32_t s_[kN];
  uint32_t pos_;
  void twist() {
    for (uint32_t k = 0; k < kN; ++k) {
      const uint32_t y = (s_[k] & 0x80000000u) | (s_[(k + 1) % kN] & 0x7fffffffu);
      s_[k] = s_[(k + kM) % kN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    pos_ = 0;
  }
};

inline float uniform(Mt19937& g, float a, float b) {
  float u = static_cast<float>(g()) / 4294967296.0f;
  if (u >= 1.0f) u = std::nextafter(1.0f, 0.0f);
  return u * (b - a) + a;
}

template <typename T>
inline void put(void* base, int i, float v) { static_cast<T*>(base)[i] = static_cast<T>(v); }

int bad(int32_t n, int32_t precision) { return n < 0 || (precision != 32 && precision != 64); }

}  // namespace

extern "C" {

int nbx_ic_pos(int32_t n, int32_t precision, void* x, void* y, void* z) {
  if (bad(n, precision) || !x || !y || !z) return NBX_ERR_ARG;
  Mt19937 g(42u);
  for (int i = 0; i < n; ++i) {
    const float a = uniform(g, 0.0f, 1.0f), b = uniform(g, 0.0f, 1.0f), c = uniform(g, 0.0f, 1.0f);
    if (precision == 32) { put<float>(x, i, a); put<float>(y, i, b); put<float>(z, i, c); }
    else { put<double>(x, i, a); put<double>(y, i, b); put<double>(z, i, c); }
  }
  return NBX_OK;
}

int nbx_ic_vel(int32_t n, int32_t precision, void* x, void* y, void* z) {
  if (bad(n, precision) || !x || !y || !z) return NBX_ERR_ARG;
  Mt19937 g(42u);
  for (int i = 0; i < n; ++i) {
    const float a = uniform(g, -1.0f, 1.0f) * 1.0e-3f;
    const float b = uniform(g, -1.0f, 1.0f) * 1.0e-3f;
    const float c = uniform(g, -1.0f, 1.0f) * 1.0e-3f;
    if (precision == 32) { put<float>(x, i, a); put<float>(y, i, b); put<float>(z, i, c); }
    else { put<double>(x, i, a); put<double>(y, i, b); put<double>(z, i, c); }
  }
  return NBX_OK;
}

int nbx_ic_mass(int32_t n, int32_t precision, void* m) {
  if (bad(n, precision) || !m) return NBX_ERR_ARG;
  Mt19937 g(42u);
  const float fn = static_cast<float>(n);
  for (int i = 0; i < n; ++i) {
    const float v = fn * uniform(g, 0.0f, 1.0f);
    if (precision == 32) put<float>(m, i, v); else put<double>(m, i, v);
  }
  return NBX_OK;
}

}  // extern "C"
