// Internal (non-ABI) declarations shared by the translation units of libeae.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/eae.h"

int eae_set_error(int code, const char* msg);   // records the message for eae_last_error(), returns code

struct ConvArgs;
int eae_launch_conv_s2(const ConvArgs& a, int cin, int cout, int src, int epi, hipStream_t st);
int eae_launch_deconv_s2(const ConvArgs& a, int cin, int cout, int src, int epi, hipStream_t st);
int eae_conv_s2_ntiles(int kind, int B, int Hin, int Win, int cin = 0);   // statistics partials per channel = workgroups along grid.x

#define EAE_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return eae_set_error(-3, hipGetErrorString(e__)); } while (0)

// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute belongs to the device's code object, so a
// process-wide "done" flag would skip it for a context created later on another device.
#include <utility>
#include <vector>
inline hipError_t eae_smem_attr(const void* func, size_t bytes) {
  static thread_local std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  for (const auto& d : done) if (d.first == func && d.second == dev) return hipSuccess;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.emplace_back(func, dev);
  return e;
}

#include "eae_group.h"
#include "eae_misc.h"
#include "eae_head.h"
// optional bracket around the MAIN kernel of a launcher that enqueues more than one (weight gradient + slice reduction)
struct EaeProfHook { void (*begin)(void* user, hipStream_t st); void (*end)(void* user, hipStream_t st); void* user; };
struct SrcDesc;
struct EdgeArgs; struct Deconv4Args; struct WgradArgs; struct FcNtArgs; struct FcTnArgs;
int eae_launch_edge_conv(hipStream_t st, int src3_kind, int epi, const EdgeArgs& a);
int eae_edge_tiles(int B, int H, int W);
int eae_launch_edge_wgrad(hipStream_t st, int src3_kind, const void* src3, int B, int H, int W, const SrcDesc& side, int smode,
                          float* scratch, long long scratch_floats, float* dw, const EaeProfHook* hook = nullptr,
                          const struct BnBwdFold* bfold = nullptr, unsigned* sig = nullptr, unsigned sig_val = 0,
                          int (*mid)(void*, GateArgs*) = nullptr, void* mid_user = nullptr);
int eae_launch_deconv4_loss(hipStream_t st, int smode, const Deconv4Args& a);
int eae_launch_wgrad_s2(hipStream_t st, const WgradArgs& a, int cs, int cb, int smode, int bmode, float* scratch,
                        long long scratch_floats, float* dw, const EaeProfHook* hook = nullptr);
int eae_launch_fc_nt(hipStream_t st, const FcNtArgs& a, int amode, int epi, int ksplit);
int eae_launch_fc_reduce(hipStream_t st, const float* part, int nsl, int M, int N, const float* bias, const float* addend,
                         const float* addend2, float* out);
int eae_launch_sigmoid_bwd(hipStream_t st, const float* x_hat, const float* dx_hat, void* g4, float* part, int B, int H, int W);
int eae_launch_fc_tn(hipStream_t st, const FcTnArgs& a, int pmode, int qmode);
