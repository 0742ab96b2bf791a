// Host-side dispatch of the generic 3x3 stride-2 weight-gradient kernel.
#include "eae_internal.h"
#include "eae_wgrad.hip.h"

namespace {
template <int CS, int CB, int TW, int TH, int NI, int SM, int BM>
int launch(const WgradArgs& a0, int ntiles, float* scratch, long long scratch_floats, float* dw, hipStream_t st, const EaeProfHook* hook) {
  WgradArgs a = a0;
  constexpr int nblk = (CS / 64) * (CB / 32);
  constexpr long long sz = (long long)CS * CB * 9;
  // One 12-wave workgroup per CU (4 consumer + 8 producer waves).  EAE_WGRAD_WGS workgroups in all: every one writes a partial of
  // its 64 x 32 x 9 block, so the count also sets the split-K traffic.
  // The 64x32 layers (one block, 67-84 MB of operands) are bandwidth-bound: every CU.  The wider layers (4 / 16 blocks, 17-42 MB)
  // are bound by per-workgroup overheads and by the partial writes themselves (256 workgroups = 18.9 MB of partials for 21 MB of
  // operands): half as many workgroups, twice the tiles each.
  static const int wgs_env = getenv("EAE_WGRAD_WGS") ? atoi(getenv("EAE_WGRAD_WGS")) : 0;
  const int wgs = wgs_env ? wgs_env : (nblk == 1 ? 256 : 128);
  int slices = wgs / nblk;
  if (slices < 1) slices = 1;
  if (slices > ntiles) slices = ntiles;
  while ((long long)slices * sz > scratch_floats && slices > 1) slices >>= 1;
  if ((long long)slices * sz > scratch_floats) return eae_set_error(-2, "wgrad: scratch too small");
  a.ntiles = ntiles;
  a.tiles_per_block = (ntiles + slices - 1) / slices;
  slices = (ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
  a.part = scratch;
  a.nslices = slices;
  auto kern = wgrad_s2_kernel<CS, CB, TW, TH, NI, SM, BM>;
  constexpr size_t smem = WgGeo<TW, TH, NI>::smem();
  EAE_HIP(eae_smem_attr(reinterpret_cast<const void*>(kern), smem));
  if (hook) hook->begin(hook->user, st);
  hipLaunchKernelGGL(kern, dim3(slices * nblk), dim3(WG_THREADS), smem, st, a);
  if (hook) hook->end(hook->user, st);
  EAE_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_slices_kernel, dim3(reduce_slices_grid(sz / 4)), dim3(256), 0, st, scratch, slices, (long)(sz / 4), dw, 1.0f);
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
#undef CASE
  return eae_set_error(-2, "wgrad: no kernel instantiated for this (cs, cb, modes)");
}
