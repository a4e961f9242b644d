// kb_common.hpp -- shared declarations of the kbench development harness
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <string>
#include <vector>

struct KbArgs {  // mirrors nbx::ForceArgs<float> without depending on the namespace
  const float4* posm; float4* accp; int n; int jps;
};
struct Variant {
  std::string name;
  int B, S;
  std::function<void(const KbArgs&, dim3, hipStream_t)> launch;
  std::vector<float> ms;
};
void reg_slp(std::vector<Variant>&);
void reg_noslp(std::vector<Variant>&);
void reg_sym(std::vector<Variant>&);
