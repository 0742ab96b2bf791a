// Host-side dispatch of the unified stride-2 conv / transposed-conv implicit-GEMM kernel (eae_igemm.hip.h).
#include "eae_internal.h"
#include "eae_igemm.hip.h"
#include "eae_igemm2.hip.h"
#ifdef EAE_IGEMM_MT       // multi-tile variant of the 16-wide geometries: built, parity-green, measured SLOWER -- A/B builds only
#include "eae_igemm_mt.hip.h"
#endif
#include <cstdlib>

namespace {

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
int launch(const ConvArgs& a, hipStream_t st) {   // NOLINT
  auto kern = igemm_s2_kernel<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI>;
  auto kern_g = igemm_s2_kernel_g<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI>;
  constexpr size_t smem = igemm_smem<KIND, BN, TW, TH, NI>();
  EAE_HIP(eae_smem_attr(reinterpret_cast<const void*>(eae_rec ? (const void*)kern_g : (const void*)kern), smem));
  const int Hpos = (KIND == KIND_CONV) ? a.Hin / 2 : a.Hin, Wpos = (KIND == KIND_CONV) ? a.Win / 2 : a.Win;
  const int groups = (a.B + NI - 1) / NI;
  const int ntiles = groups * (Hpos / TH) * (Wpos / TW);
  dim3 grid(ntiles * (COUT / BN));        // 1-D: the kernel maps ids to (tile, channel block) XCD-aware
  ConvArgs b = a;
  b.ntiles = ntiles;
  eae_launch(kern, kern_g, grid, dim3(256), smem, st, b);
  EAE_LAUNCH_CHECK();
  return 0;
}

#ifdef EAE_IGEMM_MT
// Workgroups of a multi-tile launch (eae_igemm_mt.hip.h) with `nvb` (tile, channel block) pairs: at most EAE_IG_WGS_PER_CU (default 2)
// x 256 CUs, a multiple of 8 * NB (all tiles of a workgroup then share its XCD run and its channel block), with equal shares where
// the tile count allows.
static int ig_mt_grid(int nvb, int nb) {
  static const int per_cu = getenv("EAE_IG_WGS_PER_CU") ? atoi(getenv("EAE_IG_WGS_PER_CU")) : 2;
  const int cap = (per_cu > 0 ? per_cu : 2) * 256, q = 8 * nb;
  if (nvb <= cap || nvb % q != 0) return nvb;
  for (int t = (nvb + cap - 1) / cap; t <= 256; ++t)
    if (nvb % t == 0 && (nvb / t) % q == 0) return nvb / t;
  return cap / q * q > 0 ? cap / q * q : q;
}
// EAE_IG_MT=0: one tile per workgroup for the 16-wide geometries too (rounds 1-3; A/B switch)
static bool ig_mt_on() { static const bool v = !(getenv("EAE_IG_MT") && atoi(getenv("EAE_IG_MT")) == 0); return v; }
// EAE_IG_MT_TH4=1: the 32 <-> 64-channel layers on 16 x 4-position tiles (twice the tiles per workgroup: a longer pipeline at B=512)
static bool ig_mt_th4() { static const bool v = getenv("EAE_IG_MT_TH4") && atoi(getenv("EAE_IG_MT_TH4")) != 0; return v; }

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int SRC, int EPI>
int launch_mt(const ConvArgs& a, hipStream_t st) {   // NOLINT
  EAE_NO_GROUP("the multi-tile implicit-GEMM kernel");
  void (*kern)(ConvArgs) = igemm_mt_kernel<KIND, CIN, COUT, BN, TW, TH, SRC, EPI>;
  if (a.qs) kern = igemm8_mt_kernel<KIND, CIN, COUT, BN, TW, TH, SRC, EPI>;
  constexpr size_t smem = igemm_mt_smem<KIND, BN, TW, TH>();
  EAE_HIP(eae_smem_attr(reinterpret_cast<const void*>(kern), smem));
  const int Hpos = (KIND == KIND_CONV) ? a.Hin / 2 : a.Hin, Wpos = (KIND == KIND_CONV) ? a.Win / 2 : a.Win;
  const int ntiles = a.B * (Hpos / TH) * (Wpos / TW);
  ConvArgs b = a;
  b.ntiles = ntiles;
  hipLaunchKernelGGL(kern, dim3(ig_mt_grid(ntiles * (COUT / BN), COUT / BN)), dim3(256), smem, st, b);
  EAE_LAUNCH_CHECK();
  return 0;
}
#endif

// fp8 variant (BASELINE config 5): same geometry, operands converted to fp8 (ConvArgs::qs set by the engine); 16-wide tiles only
template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
int launch8(const ConvArgs& a, hipStream_t st) {   // NOLINT
  EAE_NO_GROUP("the fp8 implicit-GEMM kernel");
  auto kern = igemm8_s2_kernel<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI>;
  constexpr size_t smem = igemm_smem<KIND, BN, TW, TH, NI>();
  EAE_HIP(eae_smem_attr(reinterpret_cast<const void*>(kern), smem));
  const int Hpos = (KIND == KIND_CONV) ? a.Hin / 2 : a.Hin, Wpos = (KIND == KIND_CONV) ? a.Win / 2 : a.Win;
  const int ntiles = a.B * (Hpos / TH) * (Wpos / TW);
  ConvArgs b = a;
  b.ntiles = ntiles;
  hipLaunchKernelGGL(kern, dim3(ntiles * (COUT / BN)), dim3(256), smem, st, b);
  EAE_LAUNCH_CHECK();
  return 0;
}

// Wave-specialised kernel (eae_igemm2.hip.h) for the multi-chunk layers on the small maps.  Inside the training step (rocprofv3,
// B=512, us, igemm2 vs one-role kernel): conv 128->256 forward 16.7 vs 18.3 and deconv 256->128 forward 16.5 vs 18.2 win; conv
// 64->128 forward 18.5 vs 16.8, deconv 128->64 forward 22.6 vs 19.6, and every backward-data use (31.3 vs 26.2, 48.8 vs 33.6: a
// 512-thread workgroup owns the whole CU and collides with the weight-gradient kernels beside it) lose.  So: EAE_IGEMM2 unset / 1 =
// the two winning forward layers only, 2 = every layer it is instantiated for, 0 = one-role kernel everywhere.
static int igemm2_mode() { static const int v = getenv("EAE_IGEMM2") ? atoi(getenv("EAE_IGEMM2")) : 1; return v; }
template <int KIND, int CIN, int EPI>
static bool igemm2_on() {
  constexpr bool wins = (EPI == EPI_FWD) && ((KIND == KIND_CONV && CIN == 128) || (KIND == KIND_DECONV && CIN == 256));
  return igemm2_mode() >= 2 || (igemm2_mode() == 1 && wins);
}

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI, int NBL>
int launch2(const ConvArgs& a, hipStream_t st) {   // NOLINT
  auto kern = igemm2_s2_kernel<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI, NBL>;
  auto kern_g = igemm2_s2_kernel_g<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI, NBL>;
  constexpr size_t smem = igemm2_smem<KIND, BN, TW, TH, NI, NBL>();
  EAE_HIP(eae_smem_attr(reinterpret_cast<const void*>(eae_rec ? (const void*)kern_g : (const void*)kern), smem));
  const int Hpos = (KIND == KIND_CONV) ? a.Hin / 2 : a.Hin, Wpos = (KIND == KIND_CONV) ? a.Win / 2 : a.Win;
  const int groups = (a.B + NI - 1) / NI;
  const int ntiles = groups * (Hpos / TH) * (Wpos / TW);
  dim3 grid(ntiles * (NBL > 1 ? 1 : COUT / BN));
  ConvArgs b = a;
  b.ntiles = ntiles;
  eae_launch(kern, kern_g, grid, dim3(512), smem, st, b);
  EAE_LAUNCH_CHECK();
  return 0;
}

// Small-tile geometries for the 8x8 / 4x4 maps: 64-position tiles for the conv kind, 32-channel blocks for the transposed kind --
// twice the workgroups, each with half the work.  They pay when the 128-position grid leaves most of the 256 CUs empty (ms per step,
// small vs large: B=64 0.255 vs 0.280, B=128 0.288 vs 0.310, B=256 0.369 vs 0.371) and lose at B=512, so a layer takes them when its
// large-tile grid has fewer than 256 workgroups.  EAE_IG_SMALL=<mask> overrides (bit 0: conv kind, bit 1: transposed kind).
static int ig_small_env() { static const int v = getenv("EAE_IG_SMALL") ? atoi(getenv("EAE_IG_SMALL")) : -1; return v; }
// (eae_geo_mult: members of a grouped step -- K contexts' launches run as one, so the grid that decides is K times the member's)
static bool conv_small(int B, int Wp, int cout) {
  if (ig_small_env() >= 0) return (ig_small_env() & 1) != 0;
  B *= eae_geo_mult;
  const int nt = (Wp == 8) ? (B + 1) / 2 : (B + 7) / 8;
  return nt * (cout / 64) < 256;
}
static bool deconv_small(int B, int Win, int cout) {
  if (ig_small_env() >= 0) return (ig_small_env() & 2) != 0;
  B *= eae_geo_mult;
  const int nt = (Win == 8) ? B : (B + 3) / 4;
  return nt * (cout / 64) < 256;
}

// geometry by the size of the position grid (conv: output map; deconv: input map)
template <int CIN, int COUT, int BN, int SRC, int EPI>
int conv_geo(const ConvArgs& a, hipStream_t st) {
  const int Hp = a.Hin / 2, Wp = a.Win / 2;
  if (Wp % 16 == 0 && Hp % 8 == 0) {
#ifdef EAE_IGEMM_MT
    if constexpr (CIN == 32) { if (ig_mt_on() && ig_mt_th4()) return launch_mt<KIND_CONV, CIN, COUT, BN, 16, 4, SRC, EPI>(a, st); }
    if (ig_mt_on()) return launch_mt<KIND_CONV, CIN, COUT, BN, 16, 8, SRC, EPI>(a, st);
#endif
    return a.qs ? launch8<KIND_CONV, CIN, COUT, BN, 16, 8, 1, SRC, EPI>(a, st) : launch<KIND_CONV, CIN, COUT, BN, 16, 8, 1, SRC, EPI>(a, st);
  }
  if (a.qs) return eae_set_error(-2, "conv_s2: the fp8 variant needs output maps that are multiples of 8 x 16");
  if constexpr (CIN >= 64) {
    constexpr int NBL = (CIN == 64) ? COUT / BN : 1;       // two chunks: both stay resident, the workgroup loops over the channel blocks
    const bool small = (Wp == 8 || Wp == 4) && conv_small(a.B, Wp, COUT);
    if (igemm2_on<KIND_CONV, CIN, EPI>() && !small) {
      if (Wp == 8 && Hp == 8) return launch2<KIND_CONV, CIN, COUT, BN, 8, 8, 2, SRC, EPI, NBL>(a, st);
      if (Wp == 4 && Hp == 4) return launch2<KIND_CONV, CIN, COUT, BN, 4, 4, 8, SRC, EPI, NBL>(a, st);
    }
    if (Wp == 8 && Hp == 8 && small) return launch<KIND_CONV, CIN, COUT, BN, 8, 8, 1, SRC, EPI>(a, st);
    if (Wp == 4 && Hp == 4 && small) return launch<KIND_CONV, CIN, COUT, BN, 4, 4, 4, SRC, EPI>(a, st);
  }
  if (Wp == 8 && Hp == 8) return launch<KIND_CONV, CIN, COUT, BN, 8, 8, 2, SRC, EPI>(a, st);
  if (Wp == 4 && Hp == 4) return launch<KIND_CONV, CIN, COUT, BN, 4, 4, 8, SRC, EPI>(a, st);
  return eae_set_error(-2, "conv_s2: unsupported spatial size (output must be 4x4, 8x8 or a multiple of 8x16)");
}

// The 64->32 transposed kind on 16x8-position tiles launches B*2 workgroups of 3 per CU at 64x64 images: 1024 workgroups on 768
// slots at B=512, i.e. TWO rounds of workgroups (deconv3 forward 22 us, conv2 backward-data 44 us for 50 / 100 MB).  16x4 tiles
// (half the accumulators: 4 per CU) make it 2048 half-size workgroups on 1024 slots -- measured SLOWER (28.1 vs 22.4 us stand-alone,
// 0.569 vs 0.562 ms per step): the round count is not what bounds this kernel.  Kept reachable with EAE_DECONV64_TH4=1.
static bool deconv64_th4() { static const bool v = getenv("EAE_DECONV64_TH4") != nullptr; return v; }

template <int CIN, int COUT, int BN, int SRC, int EPI>
int deconv_geo(const ConvArgs& a, hipStream_t st) {
  if constexpr (CIN == 64) {
    if (a.Win % 16 == 0 && a.Hin % 4 == 0 && deconv64_th4()) return launch<KIND_DECONV, CIN, COUT, BN, 16, 4, 1, SRC, EPI>(a, st);
  }
  if (a.Win % 16 == 0 && a.Hin % 8 == 0) {
#ifdef EAE_IGEMM_MT
    if constexpr (CIN == 64) { if (ig_mt_on() && ig_mt_th4()) return launch_mt<KIND_DECONV, CIN, COUT, BN, 16, 4, SRC, EPI>(a, st); }
    if (ig_mt_on()) return launch_mt<KIND_DECONV, CIN, COUT, BN, 16, 8, SRC, EPI>(a, st);
#endif
    return a.qs ? launch8<KIND_DECONV, CIN, COUT, BN, 16, 8, 1, SRC, EPI>(a, st) : launch<KIND_DECONV, CIN, COUT, BN, 16, 8, 1, SRC, EPI>(a, st);
  }
  if (a.qs) return eae_set_error(-2, "deconv_s2: the fp8 variant needs input maps that are multiples of 8 x 16");
  if constexpr (CIN >= 128) {
    if (igemm2_on<KIND_DECONV, CIN, EPI>()) {
      if (a.Win == 8 && a.Hin == 8) return launch2<KIND_DECONV, CIN, COUT, BN, 8, 8, 1, SRC, EPI, 1>(a, st);
      if (a.Win == 4 && a.Hin == 4) return launch2<KIND_DECONV, CIN, COUT, BN, 4, 4, 4, SRC, EPI, 1>(a, st);
    }
  }
  if (a.Win == 8 && a.Hin == 8) return launch<KIND_DECONV, CIN, COUT, BN, 8, 8, 1, SRC, EPI>(a, st);
  if (a.Win == 4 && a.Hin == 4) return launch<KIND_DECONV, CIN, COUT, BN, 4, 4, 4, SRC, EPI>(a, st);
  return eae_set_error(-2, "deconv_s2: unsupported spatial size (input must be 4x4, 8x8 or a multiple of 8x16)");
}

}  // namespace

// conv-kind instantiations used by the path:
//   forward of enc.conv2/3/4      : (32,64) (64,128) (128,256)  SRC_BNRELU / EPI_FWD
//   backward-data of dec.deconv3/2: (32,64) (64,128)            SRC_BNBWD  / EPI_MASK
//   backward-data of dec.deconv1  : (128,256)                   SRC_BNBWD  / EPI_PLAIN
int eae_launch_conv_s2(const ConvArgs& a, int cin, int cout, int src, int epi, hipStream_t st) {
  if (a.B <= 0 || (a.Hin & 1) || (a.Win & 1)) return eae_set_error(-2, "conv_s2: bad shape");
#define CASE(CI, CO, S, E) if (cin == CI && cout == CO && src == S && epi == E) return conv_geo<CI, CO, 64, S, E>(a, st)
  CASE(32, 64, SRC_BNRELU, EPI_FWD);
  CASE(64, 128, SRC_BNRELU, EPI_FWD);
  CASE(128, 256, SRC_BNRELU, EPI_FWD);
  CASE(32, 64, SRC_BNBWD, EPI_MASK);
  CASE(64, 128, SRC_BNBWD, EPI_MASK);
  CASE(128, 256, SRC_BNBWD, EPI_PLAIN);
  CASE(32, 64, SRC_RAW, EPI_FWD);          // plain conv (tests / generic use)
#undef CASE
  return eae_set_error(-2, "conv_s2: no kernel instantiated for this (cin, cout, src, epilogue)");
}

// deconv-kind instantiations used by the path:
//   forward of dec.deconv1/2/3      : (256,128) SRC_RAW, (128,64) (64,32) SRC_BNRELU / EPI_FWD
//   backward-data of enc.conv4/3/2  : (256,128) (128,64) (64,32)  SRC_BNBWD / EPI_MASK
int eae_launch_deconv_s2(const ConvArgs& a, int cin, int cout, int src, int epi, hipStream_t st) {
  if (a.B <= 0) return eae_set_error(-2, "deconv_s2: bad shape");
#define CASE(CI, CO, BN_, S, E) if (cin == CI && cout == CO && src == S && epi == E) return deconv_geo<CI, CO, BN_, S, E>(a, st)
  if ((a.Win == 8 || a.Win == 4) && a.Hin == a.Win && cout >= 64 && deconv_small(a.B, a.Win, cout)) {
    CASE(256, 128, 32, SRC_RAW, EPI_FWD);
    CASE(128, 64, 32, SRC_BNRELU, EPI_FWD);
    CASE(256, 128, 32, SRC_BNBWD, EPI_MASK);
    CASE(128, 64, 32, SRC_BNBWD, EPI_MASK);
  }
  CASE(256, 128, 64, SRC_RAW, EPI_FWD);
  CASE(128, 64, 64, SRC_BNRELU, EPI_FWD);
  CASE(64, 32, 32, SRC_BNRELU, EPI_FWD);
  CASE(256, 128, 64, SRC_BNBWD, EPI_MASK);
  CASE(128, 64, 64, SRC_BNBWD, EPI_MASK);
  CASE(64, 32, 32, SRC_BNBWD, EPI_MASK);
  CASE(64, 32, 32, SRC_RAW, EPI_FWD);          // plain deconv (tests / generic use)
#undef CASE
  return eae_set_error(-2, "deconv_s2: no kernel instantiated for this (cin, cout, src, epilogue)");
}

// number of per-workgroup statistics partials (= grid.x) for a given shape
int eae_conv_s2_ntiles(int kind, int B, int Hin, int Win, int cin) {
  if (kind == 1 && cin == 64 && Win % 16 == 0 && Hin % 4 == 0 && deconv64_th4()) return B * (Hin / 4) * (Win / 16);
#ifdef EAE_IGEMM_MT
  if (kind == 1 && cin == 64 && Win % 16 == 0 && Hin % 8 == 0 && ig_mt_on() && ig_mt_th4()) return B * (Hin / 4) * (Win / 16);
  if (kind == 0 && cin == 32 && (Win / 2) % 16 == 0 && (Hin / 2) % 8 == 0 && ig_mt_on() && ig_mt_th4()) return B * (Hin / 8) * (Win / 32);
#endif
  if (kind == 0) {
    int Hp = Hin / 2, Wp = Win / 2;
    if (Wp % 16 == 0 && Hp % 8 == 0) return B * (Hp / 8) * (Wp / 16);
    if (cin >= 64 && Wp == Hp && (Wp == 8 || Wp == 4) && conv_small(B, Wp, 2 * cin)) {      // (every instantiated conv layer doubles the channels)
      if (Wp == 8) return B;
      return (B + 3) / 4;
    }
    if (Wp == 8 && Hp == 8) return (B + 1) / 2;
    if (Wp == 4 && Hp == 4) return (B + 7) / 8;
    return -1;
  }
  if (Win % 16 == 0 && Hin % 8 == 0) return B * (Hin / 8) * (Win / 16);
  if (Win == 8 && Hin == 8) return B;
  if (Win == 4 && Hin == 4) return (B + 3) / 4;
  return -1;
}
