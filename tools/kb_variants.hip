// kb_variants.hip -- compiled twice by tools/Makefile: default flags (KB_NS=nbx_slp) and
// -fno-slp-vectorize (KB_NS=nbx_noslp), so the compiler's automatic v_pk_* packing can be A/B'd.
#define nbx KB_NS
#include "../nbody-demo-2023_amd/csrc/nbx_kernels.hpp"
#include "kb_common.hpp"

using namespace nbx;

template <int B, int JSRC, int MINW, int MATH, bool WS = false>
static void launch_f32(const KbArgs& k, dim3 grid, hipStream_t st) {
  ForceArgs<float> a{};
  a.posm = k.posm; a.accp = k.accp; a.i_begin = 0; a.i_count = k.n; a.own_pad = k.n; a.n_alloc = k.n;
  a.j_per_split = k.jps;
  hipLaunchKernelGGL((force_kernel<float, B, JSRC, false, MINW, MATH, WS>), grid, dim3(kBlock), 0, st, a);
}

#define ADD(B_, J_, M_) \
  vs.push_back({std::string(KB_TAG) + (J_ == JSRC_LDS ? " lds " : " sgpr") + " B" #B_ + (M_ ? " pk" : "   "), B_, 0, launch_f32<B_, J_, 1, M_>, {}})

void KB_REGISTER(std::vector<Variant>& vs) {
#ifndef KB_WITH_PK
  ADD(1, JSRC_LDS, 0); ADD(2, JSRC_LDS, 0); ADD(4, JSRC_LDS, 0); ADD(8, JSRC_LDS, 0);
  ADD(2, JSRC_SGPR, 0); ADD(4, JSRC_SGPR, 0); ADD(8, JSRC_SGPR, 0);
#else
  ADD(1, JSRC_LDS, 0); ADD(2, JSRC_LDS, 0); ADD(4, JSRC_LDS, 0); ADD(8, JSRC_LDS, 0);
  ADD(2, JSRC_LDS, 1);  ADD(4, JSRC_LDS, 1);  ADD(8, JSRC_LDS, 1);
  ADD(2, JSRC_SGPR, 1); ADD(4, JSRC_SGPR, 1); ADD(8, JSRC_SGPR, 1);
  vs.push_back({std::string(KB_TAG) + " sgprW B1   ", -1, 0, launch_f32<1, JSRC_SGPR, 1, 0, true>, {}});
  vs.push_back({std::string(KB_TAG) + " sgprW B2 pk", -2, 0, launch_f32<2, JSRC_SGPR, 1, 1, true>, {}});
  vs.push_back({std::string(KB_TAG) + " sgprW B4 pk", -4, 0, launch_f32<4, JSRC_SGPR, 1, 1, true>, {}});
  vs.push_back({std::string(KB_TAG) + " sgprW B6 pk", -6, 0, launch_f32<6, JSRC_SGPR, 1, 1, true>, {}});
  vs.push_back({std::string(KB_TAG) + " sgprW B8 pk", -8, 0, launch_f32<8, JSRC_SGPR, 1, 1, true>, {}});
  vs.push_back({std::string(KB_TAG) + " sgprW B4 sc", -4, 0, launch_f32<4, JSRC_SGPR, 1, 0, true>, {}});
#endif
}
