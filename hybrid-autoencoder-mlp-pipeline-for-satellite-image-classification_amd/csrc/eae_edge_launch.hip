// Host-side dispatch of the 3-channel edge-layer kernels (enc.conv1, dec.deconv4).
#include "eae_internal.h"
#include <cstdlib>
#include "eae_edge.hip.h"
#include "eae_wgrad.hip.h"

static int check_edge_shape(int B, int H, int W) {
  if (B <= 0 || H % 8 || W % 64) return eae_set_error(-2, "edge layer: image height must be a multiple of 8 and width of 64");
  return 0;
}

int eae_launch_edge_conv(hipStream_t st, int src3_kind, int epi, const EdgeArgs& a0) {
  if (int rc = check_edge_shape(a0.B, a0.H, a0.W)) return rc;
  EdgeArgs a = a0;
  dim3 grid(a.B * (a.H / 2 / E_TH) * (a.W / 2 / E_TW));
  a.c.ntiles = (int)grid.x;
#define CASE(S, E) if (src3_kind == S && epi == E) { eae_launch(edge_conv_kernel<S, E>, edge_conv_kernel_g<S, E>, grid, dim3(256), 0, st, a); EAE_LAUNCH_CHECK(); return 0; }
  CASE(SRC3_NCHW_F32, EPI_FWD)
  CASE(SRC3_NHWC4_BF16, EPI_MASK)
  CASE(SRC3_NHWC4_BF16, EPI_PLAIN)
  CASE(SRC3_NCHW_F32, EPI_PLAIN)
#undef CASE
  return eae_set_error(-2, "edge_conv: combination not instantiated");
}

int eae_edge_tiles(int B, int H, int W) { return B * (H / 2 / E_TH) * (W / 2 / E_TW); }

// dw [32][3][3][3] = reduce over blocks of the per-block partials. scratch must hold nblocks*864 floats.
int eae_launch_edge_wgrad(hipStream_t st, int src3_kind, const void* src3, int B, int H, int W, const SrcDesc& side, int smode,
                          float* scratch, long long scratch_floats, float* dw, const EaeProfHook* hook, const BnBwdFold* bfold,
                          unsigned* sig, unsigned sig_val, int (*mid)(void*, GateArgs*), void* mid_user) {
  if (int rc = check_edge_shape(B, H, W)) return rc;
  // the kernel addresses both operands through buffer descriptors with 32-bit byte offsets below OOB_OFF
  if ((long long)B * H * W * 16 >= 0x7fffff00LL) return eae_set_error(-2, "edge_wgrad: batch too large for one launch (2 GB per operand)");
  EdgeWgradArgs a;
  a.bfold = bfold ? *bfold : BnBwdFold();
  a.sig = sig; a.sig_val = sig_val;
  a.src3 = src3; a.B = B; a.H = H; a.W = W; a.side = side; a.part = scratch;
  a.ntiles = eae_edge_tiles(B, H, W);
  // Workgroups: each writes a partial, so their number also sets the reduction behind the kernel.  Round 4 re-sweep at B=512 (ms per
  // step, main / side cap): 1024 / 1024 0.4875-0.4891, 512 / 1024 0.4832, 512 / 512 0.4821, 512 / 256 0.4794-0.4795, 768 / 256 0.4788,
  // 512 / 128 0.4836, 640 / 192 0.4887, 256 main 0.4930.
  static const int cap = getenv("EAE_EDGE_WGRAD_BLOCKS") ? atoi(getenv("EAE_EDGE_WGRAD_BLOCKS")) : 512;
  // a launch beside the backward-data chain (deconv4's weight gradient, side stream) takes fewer CUs from it with fewer blocks
  static const int cap_side = getenv("EAE_EDGE_WGRAD_SIDE_BLOCKS") ? atoi(getenv("EAE_EDGE_WGRAD_SIDE_BLOCKS")) : 256;
  int lim = (src3_kind == SRC3_NHWC4_BF16) ? cap_side : cap;
  if (eae_geo_mult > 1) lim = lim / eae_geo_mult > 32 ? lim / eae_geo_mult : 32;      // member of a grouped step: the caps are per launch
  int nblocks = a.ntiles < lim ? a.ntiles : lim;
  while ((long long)nblocks * 864 > scratch_floats && nblocks > 1) nblocks /= 2;
  a.tiles_per_block = (a.ntiles + nblocks - 1) / nblocks;
  nblocks = (a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
  if ((long long)nblocks * 864 > scratch_floats) return eae_set_error(-2, "edge_wgrad: scratch too small");
#define CASE(S, M) if (src3_kind == S && smode == M) { if (hook) hook->begin(hook->user, st); eae_launch(edge_wgrad_kernel<S, M>, edge_wgrad_kernel_g<S, M>, dim3(nblocks), dim3(256), 0, st, a); if (hook) hook->end(hook->user, st); EAE_LAUNCH_CHECK(); goto reduce; }
  CASE(SRC3_NCHW_F32, SRC_BNBWD)
  CASE(SRC3_NHWC4_BF16, SRC_BNRELU)
  CASE(SRC3_NCHW_F32, SRC_RAW)
#undef CASE
  return eae_set_error(-2, "edge_wgrad: combination not instantiated");
reduce:
  GateArgs tail = GateArgs();      // mid(): the caller's work between the two launches; a gate it returns is waited for in the reduction's tail
  if (mid) { if (int rc = mid(mid_user, &tail)) return rc; }
  {
    const ReduceTallArgs ra = {scratch, nblocks, (long)(864 / 4), dw, tail};
    eae_launch(reduce_slices_tall_kernel, reduce_slices_tall_kernel_g, dim3((864 / 4 + 3) / 4), dim3(256), 0, st, ra);
  }
  EAE_LAUNCH_CHECK();
  return 0;
}

int eae_launch_deconv4_loss(hipStream_t st, int smode, const Deconv4Args& a) {
  if (a.B <= 0 || a.Hin % E_TH || a.Win % E_TW) return eae_set_error(-2, "deconv4: input must be a multiple of 4 x 32");
  dim3 grid(a.B * (a.Hin / E_TH) * (a.Win / E_TW));
  if (smode == SRC_BNRELU) eae_launch(deconv4_loss_kernel<SRC_BNRELU>, deconv4_loss_kernel_g<SRC_BNRELU>, grid, dim3(256), 0, st, a);
  else if (smode == SRC_RAW) eae_launch(deconv4_loss_kernel<SRC_RAW>, deconv4_loss_kernel_g<SRC_RAW>, grid, dim3(256), 0, st, a);
  else return eae_set_error(-2, "deconv4: source mode not instantiated");
  EAE_LAUNCH_CHECK();
  return 0;
}

// g4[n,oy,ox,c] = bf16(dx_hat * x_hat * (1 - x_hat))   (backward of nn.Sigmoid, R.md:383, for an externally supplied dL/dx_hat)
// fp32 NCHW in, bf16 NHWC4 out; part[block][4] = {0, sum g(c=0), sum g(c=1), sum g(c=2)} (bias gradient of deconv4)
__global__ EAE_NO_PK __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ xh, const float* __restrict__ dxh, bf16_t* __restrict__ g4,
                                                           float* __restrict__ part, long npix_total, long plane) {
  __shared__ float red[4][4];
  const long p = (long)blockIdx.x * 256 + threadIdx.x;      // pixel index n*H*W + oy*W + ox
  float g[3] = {0.f, 0.f, 0.f};
  if (p < npix_total) {
    const long n = p / plane, r = p % plane;
    uint32_t w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float x = xh[(n * 3 + c) * plane + r];
      w[c] = f2bf(dxh[(n * 3 + c) * plane + r] * x * (1.0f - x));
      g[c] = bf2f(w[c]);
    }
    *reinterpret_cast<uint2*>(g4 + p * 4) = make_uint2(w[0] | (w[1] << 16), w[2]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) g[c] += __shfl_xor(g[c], o);
    if (lane == 0) red[wave][c + 1] = g[c];
  }
  __syncthreads();
  if (threadIdx.x < 4)
    part[(size_t)blockIdx.x * 4 + threadIdx.x] = threadIdx.x == 0 ? 0.f : red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

int eae_launch_sigmoid_bwd(hipStream_t st, const float* x_hat, const float* dx_hat, void* g4, float* part, int B, int H, int W) {
  const long plane = (long)H * W, tot = plane * B;
  EAE_NO_GROUP("sigmoid_bwd");
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x_hat, dx_hat, (bf16_t*)g4, part, tot, plane);
  EAE_LAUNCH_CHECK();
  return 0;
}

#ifdef EAE_STAMPS
extern "C" int eae_debug_set_edge(void* p, int block) {
  unsigned long long* q = static_cast<unsigned long long*>(p);
  EAE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_edge_dbg), &q, sizeof(q)));
  EAE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_edge_dbg_block), &block, sizeof(block)));
  return 0;
}
#endif
