// Host-side dispatch of the generic stride-2 conv / transposed-conv implicit-GEMM kernels.
#include "eae_internal.h"
#include "eae_conv.cuh"

namespace {

template <int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  auto kern = conv_s2_kernel<CIN, COUT, BN, TW, TH, NI, SRC, EPI>;
  constexpr size_t smem = conv_s2_smem<CIN, COUT, BN, TW, TH, NI>();
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eae_set_error(-3, hipGetErrorString(e));
    attr_done = true;
  }
  const int Hout = a.Hin / 2, Wout = a.Win / 2;
  const int groups = (a.B + NI - 1) / NI;
  dim3 grid(groups * (Hout / TH) * (Wout / TW), COUT / BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
  EAE_LAUNCH_CHECK();
  return 0;
}

template <int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
int launch_deconv(const ConvArgs& a, hipStream_t st) {
  auto kern = deconv_s2_kernel<CIN, COUT, BN, TW, TH, NI, SRC, EPI>;
  constexpr size_t smem = deconv_s2_smem<CIN, COUT, BN, TW, TH, NI>();
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return eae_set_error(-3, hipGetErrorString(e));
    attr_done = true;
  }
  const int groups = (a.B + NI - 1) / NI;
  dim3 grid(groups * (a.Hin / TH) * (a.Win / TW), COUT / BN);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
  EAE_LAUNCH_CHECK();
  return 0;
}

// geometry by output width (conv) / input width (deconv)
template <int CIN, int COUT, int BN, int SRC, int EPI>
int conv_geo(const ConvArgs& a, hipStream_t st) {
  const int Hout = a.Hin / 2, Wout = a.Win / 2;
  if (Wout % 16 == 0 && Hout % 8 == 0) return launch_conv<CIN, COUT, BN, 16, 8, 1, SRC, EPI>(a, st);
  if (Wout == 8 && Hout == 8) return launch_conv<CIN, COUT, BN, 8, 8, 2, SRC, EPI>(a, st);
  if (Wout == 4 && Hout == 4) return launch_conv<CIN, COUT, BN, 4, 4, 8, SRC, EPI>(a, st);
  return eae_set_error(-2, "conv_s2: unsupported spatial size (output must be 4x4, 8x8 or a multiple of 8x16)");
}

template <int CIN, int COUT, int BN, int SRC, int EPI>
int deconv_geo(const ConvArgs& a, hipStream_t st) {
  if (a.Win % 8 == 0 && a.Hin % 4 == 0) return launch_deconv<CIN, COUT, BN, 8, 4, 1, SRC, EPI>(a, st);
  if (a.Win == 4 && a.Hin == 4) return launch_deconv<CIN, COUT, BN, 4, 4, 2, SRC, EPI>(a, st);
  return eae_set_error(-2, "deconv_s2: unsupported spatial size (input must be 4x4 or a multiple of 4x8)");
}

}  // namespace

// conv kernel instantiations used by the path:
//   forward of enc.conv2/3/4      : (32,64) (64,128) (128,256)  SRC_BNRELU / EPI_FWD
//   backward-data of dec.deconv3/2: (32,64) (64,128)            SRC_BNBWD  / EPI_MASK
//   backward-data of dec.deconv1  : (128,256)                   SRC_BNBWD  / EPI_PLAIN
int eae_launch_conv_s2(const ConvArgs& a, int cin, int cout, int src, int epi, hipStream_t st) {
  if (a.B <= 0 || (a.Hin & 1) || (a.Win & 1)) return eae_set_error(-2, "conv_s2: bad shape");
#define CASE(CI, CO, BN_, S, E) if (cin == CI && cout == CO && src == S && epi == E) return conv_geo<CI, CO, BN_, S, E>(a, st)
  CASE(32, 64, 64, SRC_BNRELU, EPI_FWD);
  CASE(64, 128, 128, SRC_BNRELU, EPI_FWD);
  CASE(128, 256, 128, SRC_BNRELU, EPI_FWD);
  CASE(32, 64, 64, SRC_BNBWD, EPI_MASK);
  CASE(64, 128, 128, SRC_BNBWD, EPI_MASK);
  CASE(128, 256, 128, SRC_BNBWD, EPI_PLAIN);
  CASE(32, 64, 64, SRC_RAW, EPI_FWD);          // plain conv (tests / generic use)
#undef CASE
  return eae_set_error(-2, "conv_s2: no kernel instantiated for this (cin, cout, src, epilogue)");
}

// deconv kernel instantiations used by the path:
//   forward of dec.deconv1/2/3      : (256,128) SRC_RAW, (128,64) (64,32) SRC_BNRELU / EPI_FWD
//   backward-data of enc.conv4/3/2  : (256,128) (128,64) (64,32)  SRC_BNBWD / EPI_MASK
int eae_launch_deconv_s2(const ConvArgs& a, int cin, int cout, int src, int epi, hipStream_t st) {
  if (a.B <= 0) return eae_set_error(-2, "deconv_s2: bad shape");
#define CASE(CI, CO, BN_, S, E) if (cin == CI && cout == CO && src == S && epi == E) return deconv_geo<CI, CO, BN_, S, E>(a, st)
  CASE(256, 128, 128, SRC_RAW, EPI_FWD);
  CASE(128, 64, 64, SRC_BNRELU, EPI_FWD);
  CASE(64, 32, 32, SRC_BNRELU, EPI_FWD);
  CASE(256, 128, 128, SRC_BNBWD, EPI_MASK);
  CASE(128, 64, 64, SRC_BNBWD, EPI_MASK);
  CASE(64, 32, 32, SRC_BNBWD, EPI_MASK);
  CASE(64, 32, 32, SRC_RAW, EPI_FWD);          // plain deconv (tests / generic use)
#undef CASE
  return eae_set_error(-2, "deconv_s2: no kernel instantiated for this (cin, cout, src, epilogue)");
}

int eae_conv_s2_ntiles(int kind, int B, int Hin, int Win) {
  if (kind == 0) {
    int Hout = Hin / 2, Wout = Win / 2;
    if (Wout % 16 == 0 && Hout % 8 == 0) return B * (Hout / 8) * (Wout / 16);
    if (Wout == 8) return (B + 1) / 2;
    if (Wout == 4) return (B + 7) / 8;
    return -1;
  }
  if (Win % 8 == 0 && Hin % 4 == 0) return B * (Hin / 4) * (Win / 8);
  if (Win == 4) return (B + 1) / 2;
  return -1;
}
