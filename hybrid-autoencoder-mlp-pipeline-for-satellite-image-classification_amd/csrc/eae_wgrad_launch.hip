// Host-side dispatch of the generic 3x3 stride-2 weight-gradient kernel.
#include "eae_internal.h"
#include <cstdio>
#include <cstdlib>
#include "eae_wgrad.hip.h"

namespace {
template <int CS, int CB, int TW, int TH, int NI, int SM, int BM>
int launch(const WgradArgs& a0, int ntiles, float* scratch, long long scratch_floats, float* dw, hipStream_t st, const EaeProfHook* hook) {
  WgradArgs a = a0;
  constexpr int nblk = (CS / 64) * (CB / 32);
  constexpr long long sz = (long long)CS * CB * 9;
  // 12-wave workgroups (4 consumer + 8 producer waves), one per CU.  Every workgroup writes a partial of its 64 x 32 x 9 block, so
  // their number also sets the split-K traffic and the work of the reduction behind the kernel.  Inside the train step these
  // kernels run beside the backward-data chain, and what counts is how many CUs they take from it: 64 workgroups measured best for
  // every layer (ms per step at B=512 / B=64: 32: 0.599 / 0.241, 48: 0.540, 64: 0.500-0.512 / 0.244, 80: 0.515, 128: 0.509-0.527 /
  // 0.271, 256: 0.542).  EAE_WGRAD_WGS=<n> or <n one-block layers>,<n wider layers> overrides.
  static int wgs_one = 64, wgs_wide = 64, wgs_last = 64;
  static const bool parsed = [] {
    if (const char* e = getenv("EAE_WGRAD_WGS")) {
      int a = 0, b = 0, c = 0;
      const int n = sscanf(e, "%d,%d,%d", &a, &b, &c);
      if (n >= 1 && a > 0) { wgs_one = a; wgs_wide = (n >= 2 && b > 0) ? b : a; wgs_last = (n >= 3 && c > 0) ? c : wgs_one; }
    }
    return true;
  }();
  (void)parsed;
  // enc.conv2's weight gradient is the last one of the step: it runs in the tail, beside conv2's backward-data and conv1's weight gradient only
  constexpr bool LAST = (CS == 64 && (SM == SRC_BNBWD || SM == SRC_RAWG));
  int wgs = LAST ? wgs_last : (nblk == 1 ? wgs_one : wgs_wide);
  if (eae_geo_mult > 1) wgs = wgs / eae_geo_mult > 8 ? wgs / eae_geo_mult : 8;      // member of a grouped step: the grid sizes are per launch
  int slices = wgs / nblk;
  if (slices < 1) slices = 1;
  if (slices > ntiles) slices = ntiles;
  while ((long long)slices * sz > scratch_floats && slices > 1) slices >>= 1;
  if ((long long)slices * sz > scratch_floats) return eae_set_error(-2, "wgrad: scratch too small");
  a.ntiles = ntiles;
  a.tiles_per_block = (ntiles + slices - 1) / slices;
  slices = (ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
  a.part = slices == 1 ? dw : scratch;      // one slice: its "partial" is the result (same layout), no reduction launch
  a.nslices = slices;
  constexpr size_t smem = WgGeo<TW, TH, NI>::smem();
  void (*kern)(WgradArgs) = wgrad_s2_kernel<CS, CB, TW, TH, NI, SM, BM>;
  void (*kern_g)(GroupPack<WgradArgs>, int) = wgrad_s2_kernel_g<CS, CB, TW, TH, NI, SM, BM>;
  if (a.qs) {                                // fp8 variant (BASELINE config 5): built for the 16 x 8 tiles only
    EAE_NO_GROUP("the fp8 weight-gradient kernel");
    if constexpr (TW == 16) kern = wgrad8_s2_kernel<CS, CB, TW, TH, NI, SM, BM>;
    else return eae_set_error(-2, "wgrad: the fp8 variant needs small maps that are multiples of 8 x 16");
  }
  EAE_HIP(eae_smem_attr(eae_rec ? reinterpret_cast<const void*>(kern_g) : reinterpret_cast<const void*>(kern), smem));
  if (hook) hook->begin(hook->user, st);
  eae_launch(kern, kern_g, dim3(slices * nblk), dim3(WG_THREADS), smem, st, a);
  if (hook) hook->end(hook->user, st);
  EAE_LAUNCH_CHECK();
  if (slices == 1) return 0;
  launch_reduce_slices(st, scratch, slices, (long)(sz / 4), dw, 1.0f);
  EAE_LAUNCH_CHECK();
  return 0;
}

template <int CS, int CB, int SM, int BM>
int geo(const WgradArgs& a, float* scratch, long long sf, float* dw, hipStream_t st, const EaeProfHook* hook) {
  if (a.Ws % 16 == 0 && a.Hs % 8 == 0) return launch<CS, CB, 16, 8, 1, SM, BM>(a, a.B * (a.Hs / 8) * (a.Ws / 16), scratch, sf, dw, st, hook);
  if (a.Ws == 8 && a.Hs == 8) return launch<CS, CB, 8, 8, 2, SM, BM>(a, (a.B + 1) / 2, scratch, sf, dw, st, hook);
  if (a.Ws == 4 && a.Hs == 4) return launch<CS, CB, 4, 4, 8, SM, BM>(a, (a.B + 7) / 8, scratch, sf, dw, st, hook);
  return eae_set_error(-2, "wgrad: unsupported spatial size");
}
}  // namespace

// instantiations used by the path
//   conv2/3/4   : small = dy  (BNBWD), big = input activation (BNRELU): (64,32) (128,64) (256,128)
//   deconv3/2   : small = input activation (BNRELU), big = dOut (BNBWD): (64,32) (128,64)
//   deconv1     : small = dec.fc output (RAW),      big = dOut (BNBWD): (256,128)
int eae_launch_wgrad_s2(hipStream_t st, const WgradArgs& a, int cs, int cb, int smode, int bmode, float* scratch,
                        long long scratch_floats, float* dw, const EaeProfHook* hook) {
#define CASE(S, B_, SM, BM) if (cs == S && cb == B_ && smode == SM && bmode == BM) return geo<S, B_, SM, BM>(a, scratch, scratch_floats, dw, st, hook)
  CASE(64, 32, SRC_BNBWD, SRC_BNRELU);
  CASE(128, 64, SRC_BNBWD, SRC_BNRELU);
  CASE(256, 128, SRC_BNBWD, SRC_BNRELU);
  CASE(64, 32, SRC_BNRELU, SRC_BNBWD);
  CASE(128, 64, SRC_BNRELU, SRC_BNBWD);
  CASE(256, 128, SRC_RAW, SRC_BNBWD);
  CASE(64, 32, SRC_RAW, SRC_RAW);       // plain (tests)
  // the train step: the gradient operand is the dy tensor the layer's backward-data kernel wrote while staging it (SRC_RAWG)
  CASE(64, 32, SRC_RAWG, SRC_BNRELU);
  CASE(128, 64, SRC_RAWG, SRC_BNRELU);
  CASE(256, 128, SRC_RAWG, SRC_BNRELU);
  CASE(64, 32, SRC_BNRELU, SRC_RAWG);
  CASE(128, 64, SRC_BNRELU, SRC_RAWG);
  CASE(256, 128, SRC_RAW, SRC_RAWG);
#undef CASE
  return eae_set_error(-2, "wgrad: no kernel instantiated for this (cs, cb, modes)");
}
