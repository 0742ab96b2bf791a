// Kernels for the two 3-channel "edge" layers, which carry the most bytes and the least math (hard parts):
//   enc.conv1  (R.md:292)  x fp32 NCHW [B,3,H,W]            -> y1 [B,H/2,W/2,32]
//   dec.deconv4 (R.md:382) a3 [B,H/2,W/2,32] -> sigmoid -> x_hat [B,3,H,W], fused with MSE loss and its gradient
// K = 27 is padded to 32 in LDS only (one v_mfma_f32_16x16x32_bf16 K-step), N = 3 is handled by computing the four
// sub-pixel phases jointly (N = 4 phases x 3 channels = 12 of 16 MFMA columns), never by padding in HBM.
#pragma once
#include "eae_common.hip.h"
#include "eae_igemm.hip.h"

#ifdef EAE_STAMPS          // diagnostic build: s_memtime stamps of one workgroup of the edge kernels (tools/kstamp3.py)
__device__ unsigned long long* g_edge_dbg = nullptr;
__device__ int g_edge_dbg_block = 0;
#define EDGE_STAMP(i) do { if (g_edge_dbg && (int)blockIdx.x == g_edge_dbg_block && threadIdx.x == 0) { \
    unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); g_edge_dbg[i] = t__; } } while (0)
#else
#define EDGE_STAMP(i) do {} while (0)
#endif

enum { SRC3_NCHW_F32 = 0,     // fp32 planar image (the loader contract)
       SRC3_NHWC4_BF16 = 1 }; // bf16 pixels padded to 4 channels (gradient of the pre-sigmoid output)

constexpr int E_TH = 4, E_TW = 32;                 // 128 output pixels (conv view) / 128 input positions (deconv view)
constexpr int E_PH = 2 * E_TH + 1, E_PW = 2 * E_TW + 1;   // 9 x 65 patch of the 3-channel tensor
constexpr int E_PATCH = E_PH * E_PW * 4;           // bf16 elements ([row][col][4])
constexpr int E_AT = 128 * PIX_STRIDE;             // im2col tile [128][32 + pad]

// Stage the 3-channel patch (rows iy0.., cols ix0..) as bf16 [E_PH][E_PW][4] with zero padding outside the image.
// Every pixel of the patch is written exactly once, as a whole 8-byte [c0, c1, c2, 0] pixel: a thread takes 4 consecutive pixels of
// a row -- for the planar fp32 source one float4 per colour plane -- and writes them with four 8-byte stores (the patch starts one
// pixel left of a 16-byte boundary: the halo column).  Round 2 cleared the patch, synchronised, and scattered 2-byte elements plane
// by plane (12 ds_write_b16 per float4 triple): by the stamps 4.8-8.5 K of conv1's 9.5-14.6 K workgroup cycles.  One barrier, at the end.
// (split into a load and a write half so that a loop over tiles can keep the NEXT tile's raw values in flight: edge_wgrad_kernel)
template <int SRC3> struct Patch3Regs { float4 v[3]; };                 // planar fp32: one float4 per colour plane (halo threads: .x only)
template <> struct Patch3Regs<1> { uint4 v[2]; };                      // bf16 NHWC4: up to two 16-byte pieces per thread
template <int SRC3>
__device__ __forceinline__ void patch3_load(const void* src, int n, int H, int W, int iy0, int ix0, Patch3Regs<SRC3>& r) {
  const int tid = threadIdx.x;
  if constexpr (SRC3 == SRC3_NCHW_F32) {
    const float* x = static_cast<const float*>(src);
#pragma unroll
    for (int c = 0; c < 3; ++c) r.v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    // interior columns ix0+1 .. ix0+64 are image columns (ix0 = 2*tx0-1, tx0 multiple of 32 -> 16-byte aligned rows)
    if (tid < E_PH * 16) {
      const int c4 = tid & 15, rr = tid >> 4;
      const int iy = iy0 + rr, ix = ix0 + 1 + c4 * 4;
      if (iy >= 0 && iy < H) {
#pragma unroll
        for (int c = 0; c < 3; ++c) r.v[c] = *reinterpret_cast<const float4*>(x + (((size_t)n * 3 + c) * H + iy) * W + ix);
      }
    } else if (tid < E_PH * 16 + E_PH) {          // left halo column: a real pixel for tiles not at the image edge, else zero padding
      const int iy = iy0 + tid - E_PH * 16;
      if (ix0 >= 0 && iy >= 0 && iy < H) {
#pragma unroll
        for (int c = 0; c < 3; ++c) r.v[c].x = x[(((size_t)n * 3 + c) * H + iy) * W + ix0];
      }
    }
  } else {
    const bf16_t* g = static_cast<const bf16_t*>(src);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * 256;
      r.v[k] = make_uint4(0, 0, 0, 0);
      if (i < E_PH * 32) {                        // 2 pixels (16 B) per piece
        const int c2 = i & 31, rr = i >> 5;
        const int iy = iy0 + rr, ix = ix0 + 1 + c2 * 2;
        if (iy >= 0 && iy < H) r.v[k] = *reinterpret_cast<const uint4*>(g + (((size_t)n * H + iy) * W + ix) * 4);
      } else if (i < E_PH * 32 + E_PH) {          // the halo column
        const int iy = iy0 + i - E_PH * 32;
        if (ix0 >= 0 && iy >= 0 && iy < H) {
          const uint2 h = *reinterpret_cast<const uint2*>(g + (((size_t)n * H + iy) * W + ix0) * 4);
          r.v[k].x = h.x; r.v[k].y = h.y;
        }
      }
    }
  }
}
template <int SRC3>
__device__ __forceinline__ void patch3_write(bf16_t* p3, const Patch3Regs<SRC3>& r) {
  const int tid = threadIdx.x;
  if constexpr (SRC3 == SRC3_NCHW_F32) {
    if (tid < E_PH * 16) {
      const int c4 = tid & 15, rr = tid >> 4;
      uint2* d = reinterpret_cast<uint2*>(p3 + (rr * E_PW + 1 + c4 * 4) * 4);
      d[0] = make_uint2(pack2(r.v[0].x, r.v[1].x), f2bf(r.v[2].x));
      d[1] = make_uint2(pack2(r.v[0].y, r.v[1].y), f2bf(r.v[2].y));
      d[2] = make_uint2(pack2(r.v[0].z, r.v[1].z), f2bf(r.v[2].z));
      d[3] = make_uint2(pack2(r.v[0].w, r.v[1].w), f2bf(r.v[2].w));
    } else if (tid < E_PH * 16 + E_PH) {
      *reinterpret_cast<uint2*>(p3 + ((tid - E_PH * 16) * E_PW) * 4) = make_uint2(pack2(r.v[0].x, r.v[1].x), f2bf(r.v[2].x));
    }
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * 256;
      if (i < E_PH * 32) {
        const int c2 = i & 31, rr = i >> 5;
        uint2* d = reinterpret_cast<uint2*>(p3 + (rr * E_PW + 1 + c2 * 2) * 4);
        d[0] = make_uint2(r.v[k].x, r.v[k].y);
        d[1] = make_uint2(r.v[k].z, r.v[k].w);
      } else if (i < E_PH * 32 + E_PH) {
        *reinterpret_cast<uint2*>(p3 + ((i - E_PH * 32) * E_PW) * 4) = make_uint2(r.v[k].x, r.v[k].y);
      }
    }
  }
}
template <int SRC3>
__device__ __forceinline__ void stage_patch3(const void* src, bf16_t* p3, int n, int H, int W, int iy0, int ix0) {
  Patch3Regs<SRC3> r;
  patch3_load<SRC3>(src, n, H, W, iy0, ix0, r);
  patch3_write<SRC3>(p3, r);
  __syncthreads();
}

// Build the im2col tile [128 pixels][32 k] (k = tap*3 + c, zero for k >= 27) from the staged patch.
__device__ __forceinline__ void build_im2col27(const bf16_t* p3, bf16_t* at) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + i * 256;
    int m = q >> 2, kg = q & 3;
    int ty = m / E_TW, tx = m % E_TW;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      uint32_t e[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int k = kg * 8 + j + u;
        int tap = k / 3, c = k % 3;
        int ky = tap / 3, kx = tap % 3;
        e[u] = (k < 27) ? (uint32_t)p3[((2 * ty + ky) * E_PW + 2 * tx + kx) * 4 + c] : 0u;
      }
      w[j >> 1] = e[0] | (e[1] << 16);
    }
    *reinterpret_cast<uint4*>(at + m * PIX_STRIDE + kg * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// out[m][32] = im2col27(src3)[m][32] . Wp[32][32]^T      (conv1 forward; backward-data of deconv4)
// ---------------------------------------------------------------------------------------------------------------
struct EdgeArgs {
  const void* src3;        // fp32 NCHW [B,3,H,W] or bf16 NHWC4 [B,H,W,4]
  int B, H, W;             // spatial size of the 3-channel tensor
  ConvArgs c;              // wpack [32][64] (k = tap*4 + c, zero where c == 3 or k >= 36), bias, out [B,H/2,W/2,32], stat_part, yprev, prev_coef
};

// No im2col tile: with k = tap*4 + c (c = 0..3, the 4th channel and k >= 36 are zero in the weights: K = 64, two MFMA k-steps)
// a lane's 8 consecutive k are TWO whole pixels of the staged [row][col][4] patch, so the pixel operand is two ds_read_b64
// straight from the patch.  The weights are the MFMA A operand: an accumulator lane holds 4 consecutive output channels of
// one pixel (8-byte tile writes instead of 16 two-byte ones).
template <int SRC3, int EPI>
__device__ __forceinline__ void edge_conv_body(const EdgeArgs& a) {
  __shared__ __attribute__((aligned(16))) bf16_t p3[E_PATCH];
  // output tile [128][40] (10 KB); the statistics reduction scratch [2][64][32] floats (16 KB) reuses the same memory: TileEpilogue::end()
  // starts with a barrier behind the last read of the tile (31.3 -> 20.7 KB of LDS per block: 7 instead of 5 blocks per CU)
  __shared__ __attribute__((aligned(16))) float red[2 * 64 * 32];
  static_assert(sizeof(float) * 2 * 64 * 32 >= sizeof(bf16_t) * E_AT, "the tile must fit the reduction scratch");
  bf16_t* const at = reinterpret_cast<bf16_t*>(red);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hout = a.H >> 1, Wout = a.W >> 1;
  const int tiles_x = Wout / E_TW, tiles_y = Hout / E_TH;
  int t = blockIdx.x;
  const int txb = t % tiles_x; t /= tiles_x;
  const int tyb = t % tiles_y; t /= tiles_y;
  const int n = t;
  const int kgl = lane >> 4;
  eae_signal(a.c.sig, a.c.sig_val);
  // weight fragments first (independent of the patch): A[channel][k], 2 m-tiles x 2 k-steps
  bf16x8 wf[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      wf[mt][ks] = *reinterpret_cast<const bf16x8*>(a.c.wpack + (mt * 16 + (lane & 15)) * 64 + ks * 32 + kgl * 8);
  EDGE_STAMP(16);
  stage_patch3<SRC3>(a.src3, p3, n, a.H, a.W, 2 * tyb * E_TH - 1, 2 * txb * E_TW - 1);      // ends with a barrier
  EDGE_STAMP(17);
  f32x4 acc[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int m = (wave * 2 + nt) * 16 + (lane & 15);                // this lane's pixel of the n-tile
    const int ty = m / E_TW, tx = m % E_TW;
    const bf16_t* pb = p3 + ((2 * ty) * E_PW + 2 * tx) * 4;
    // k-step 0: taps 2*kgl and 2*kgl+1; k-step 1: tap 8 (lane group 0 only), everything else multiplies zero weights
    const int t0 = 2 * kgl, t1 = 2 * kgl + 1;
    union { bf16x8 v; uint2 h[2]; } f0, f1;
    f0.h[0] = *reinterpret_cast<const uint2*>(pb + ((t0 / 3) * E_PW + (t0 % 3)) * 4);
    f0.h[1] = *reinterpret_cast<const uint2*>(pb + ((t1 / 3) * E_PW + (t1 % 3)) * 4);
    f1.h[0] = (kgl == 0) ? *reinterpret_cast<const uint2*>(pb + (2 * E_PW + 2) * 4) : make_uint2(0, 0);
    f1.h[1] = make_uint2(0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      acc[mt][nt] = mfma16(wf[mt][0], f0.v, (f32x4){0.f, 0.f, 0.f, 0.f});
      acc[mt][nt] = mfma16(wf[mt][1], f1.v, acc[mt][nt]);
    }
  }
  // D[channel][pixel]: col = lane & 15 = pixel, rows (lane >> 4) * 4 + r = 4 consecutive channels
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int ch = mt * 16 + (lane >> 4) * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_FWD) bv = *reinterpret_cast<const float4*>(a.c.bias + ch);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int m = (wave * 2 + nt) * 16 + (lane & 15);
      uint2 w2;
      w2.x = pk2((f32x2){acc[mt][nt][0] + bv.x, acc[mt][nt][1] + bv.y});
      w2.y = pk2((f32x2){acc[mt][nt][2] + bv.z, acc[mt][nt][3] + bv.w});
      *reinterpret_cast<uint2*>(at + m * 40 + ch) = w2;
    }
  }
  EDGE_STAMP(18);
  __syncthreads();
  auto rowmap = [=](int row) -> long {
    int ty = row / E_TW, tx = row % E_TW;
    return (((long)n * Hout + (tyb * E_TH + ty)) * Wout + (txb * E_TW + tx)) * 32;
  };
  tile_epilogue<32, 32, EPI>(a.c, at, red, 0, blockIdx.x, 128, rowmap);
  EDGE_STAMP(19);
}
template <int SRC3, int EPI>
__global__ __launch_bounds__(256) void edge_conv_kernel(EdgeArgs a) { edge_conv_body<SRC3, EPI>(a); }
template <int SRC3, int EPI>
__global__ __launch_bounds__(256) void edge_conv_kernel_g(GroupPack<EdgeArgs> p, int gz) { edge_conv_body<SRC3, EPI>(group_args<EdgeArgs>(gz)); }

// ---------------------------------------------------------------------------------------------------------------
// R[k][c] = sum_m im2col27(src3)[m][k] * T(side)[m][c]     (weight gradient of conv1 and of deconv4)
// Each block sweeps `tiles_per_block` 128-pixel tiles and writes one fp32 partial in the REFERENCE layout
// [c][3][3][3] (index c*27 + cx*9 + tap with k = tap*3 + cx), summed later by reduce_slices (deterministic).
// ---------------------------------------------------------------------------------------------------------------
struct EdgeWgradArgs {
  const void* src3;
  int B, H, W;
  SrcDesc side;           // [B,H/2,W/2,32] tensor with its load transform
  float* part;            // [nblocks][32*27]
  int tiles_per_block, ntiles;
  BnBwdFold bfold;        // SRC_BNBWD side: coefficient table from the layer's backward accumulators (eae_common.hip.h)
  unsigned* sig;          // progress word of the caller's stream, published when the kernel starts (ConvArgs::sig); nullptr: none
  unsigned sig_val;
};

template <int SRC3, int SMODE>
__device__ __forceinline__ void edge_wgrad_body(const EdgeWgradArgs& a) {
  eae_signal(a.sig, a.sig_val);
  __shared__ __attribute__((aligned(16))) bf16_t p3[E_PATCH];
  // the two operand tiles; the cross-wave reduction image of the epilogue (16 KB) reuses them (41.9 -> 26 KB of LDS per block:
  // 6 instead of 3 blocks per CU)
  static_assert(2 * E_AT * sizeof(bf16_t) >= 4 * 2 * 2 * 64 * 4 * sizeof(float), "reduction image must fit the operand tiles");
  __shared__ __attribute__((aligned(16))) bf16_t tiles[2 * E_AT];
  bf16_t* const at = tiles;
  bf16_t* const st = tiles + E_AT;
  float (*racc)[2][2][64 * 4] = reinterpret_cast<float (*)[2][2][64 * 4]>(tiles);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hout = a.H >> 1, Wout = a.W >> 1;
  const int tiles_x = Wout / E_TW, tiles_y = Hout / E_TH;
  const int kgs = tid & 3;
  ChanCoef<SMODE> cc;
  {
    __shared__ float coef_tab[SMODE == SRC_BNBWD ? 3 * 32 : 4];
    const float* coefp = a.side.coef;
    if (SMODE == SRC_BNBWD && a.bfold.acc != nullptr) {      // folded BatchNorm-backward finalize (this kernel is the layer's only consumer)
      BnFoldRegsB fr;
      bn_fold_bwd_load<32>(a.bfold, fr);
      bn_fold_bwd_finish<32>(a.bfold, fr, coef_tab, reinterpret_cast<long long*>(&racc[0][0][0][0]), blockIdx.x == 0);
      coefp = coef_tab;
    }
    cc.load(coefp, 32, kgs * 8);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  // Tile loop with the NEXT tile's raw values (side pieces + 3-channel patch) requested before this tile is multiplied: round 2
  // loaded, staged and multiplied one tile after the other, every tile exposing a memory round trip (conv1's weight gradient is the
  // last kernel of the backward: 32 us on the critical path).
  const int t_begin = blockIdx.x * a.tiles_per_block;
  int t_end = t_begin + a.tiles_per_block;
  if (t_end > a.ntiles) t_end = a.ntiles;
  RawPiece<SMODE> raw[2];
  Patch3Regs<SRC3> pr;
  auto request = [&](int t) __attribute__((always_inline)) {
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y; t /= tiles_y;
    const int n = t;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = (tid + i * 256) >> 2;
      int ty = m / E_TW, tx = m % E_TW;
      size_t off = ((((size_t)n * Hout + tyb * E_TH + ty) * Wout) + txb * E_TW + tx) * 32 + kgs * 8;
      load_piece<SMODE>(a.side, off, true, raw[i]);
    }
    patch3_load<SRC3>(a.src3, n, a.H, a.W, 2 * tyb * E_TH - 1, 2 * txb * E_TW - 1, pr);
  };
  if (t_begin < t_end) request(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();                 // the previous tile's fragment reads (at, st) and im2col reads (p3) are done
    patch3_write<SRC3>(p3, pr);
    uint4 sv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) sv[i] = transform_piece<SMODE>(raw[i], true, cc);
    if (t + 1 < t_end) request(t + 1);
    __syncthreads();                 // the 3-channel patch is complete
    build_im2col27(p3, at);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = (tid + i * 256) >> 2;
      *reinterpret_cast<uint4*>(st + m * PIX_STRIDE + kgs * 8) = sv[i];
    }
    __syncthreads();
    // wave w reduces pixels 32w .. 32w+31
    const int r_lo = wave * 32 + 8 * g + q, r_hi = r_lo + 4;
    bf16x8 ka[2], sb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      ka[it] = tr_frag(at + r_lo * PIX_STRIDE + it * 16 + 4 * p, at + r_hi * PIX_STRIDE + it * 16 + 4 * p);
      sb[it] = tr_frag(st + r_lo * PIX_STRIDE + it * 16 + 4 * p, st + r_hi * PIX_STRIDE + it * 16 + 4 * p);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) acc[it][jt] = mfma16(ka[it], sb[jt], acc[it][jt]);
  }
  // cross-wave reduction (fixed order) and store in reference layout
  __syncthreads();            // every wave has read its fragments of the last tile: the image may overwrite the tiles
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) racc[wave][it][jt][lane * 4 + r] = acc[it][jt][r];
  __syncthreads();
  // 4 (it,jt) tiles x 256 values = 1024 outputs; thread handles 4
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int e = tid + i * 256;
    int tile = e >> 8, idx = e & 255;
    int it = tile >> 1, jt = tile & 1;
    float v = racc[0][it][jt][idx] + racc[1][it][jt][idx] + racc[2][it][jt][idx] + racc[3][it][jt][idx];
    int l = idx >> 2, r = idx & 3;
    int k = it * 16 + (l >> 4) * 4 + r;     // D row  = k index (im2col side)
    int c = jt * 16 + (l & 15);             // D col  = side channel
    if (k < 27) {
      int tap = k / 3, cx = k % 3;
      a.part[(size_t)blockIdx.x * (32 * 27) + c * 27 + cx * 9 + tap] = v;
    }
  }
}
template <int SRC3, int SMODE>
__global__ __launch_bounds__(256) void edge_wgrad_kernel(EdgeWgradArgs a) { edge_wgrad_body<SRC3, SMODE>(a); }
template <int SRC3, int SMODE>
__global__ __launch_bounds__(256) void edge_wgrad_kernel_g(GroupPack<EdgeWgradArgs> p, int gz) { edge_wgrad_body<SRC3, SMODE>(group_args<EdgeWgradArgs>(gz)); }

// ---------------------------------------------------------------------------------------------------------------
// deconv4 forward (all four phases jointly) + sigmoid + MSE loss + its gradient      (R.md:382-383, 622, 649)
//   s[n,oy,ox,co] = b[co] + sum over the 2x2 input neighbourhood ;  x_hat = sigmoid(s)
//   loss partial  = sum (x_hat - x)^2 ;  g4 = gscale * (x_hat - x) * x_hat * (1 - x_hat)   (gscale = alpha*2/numel)
// ---------------------------------------------------------------------------------------------------------------
struct Deconv4Args {
  SrcDesc src;             // a3 = BNRELU(u3)  [B,Hin,Win,32]
  const bf16_t* wjoint;    // [16][128]  n = phase*3+co, k = nb*32+ci
  const float* bias;       // [3]
  const float* x;          // target fp32 NCHW [B,3,2Hin,2Win] or nullptr (forward only)
  float* x_hat;            // fp32 NCHW or nullptr
  bf16_t* g4;              // bf16 NHWC4 [B,2Hin,2Win,4] or nullptr
  float* loss_part;        // [ntiles][4]: sum diff^2, sum g (co = 0,1,2)   or nullptr
  float gscale;
  int B, Hin, Win;
  BnFold fold;             // BNRELU source: coefficient table of deconv3's BatchNorm from its accumulators
};

template <int SRC>
__device__ __forceinline__ void deconv4_loss_body(const Deconv4Args& a) {
  constexpr int PH = E_TH + 1, PW = E_TW + 1, NPIX = PH * PW;       // 5 x 33 input pixels
  constexpr int NPA = (NPIX * 4 + 255) / 256;
  // the pre-sigmoid tile `sl` (8.7 KB) reuses the patch (13.2 KB) once every wave has read its fragments: 8 blocks per CU
  __shared__ __attribute__((aligned(16))) bf16_t patch[NPIX * PIX_STRIDE];
  static_assert(sizeof(bf16_t) * NPIX * PIX_STRIDE >= sizeof(float) * 128 * 17, "the tile must fit the patch");
  float* const sl = reinterpret_cast<float*>(patch);
  __shared__ float redl[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Hout = a.Hin * 2, Wout = a.Win * 2;
  const int tiles_x = a.Win / E_TW, tiles_y = a.Hin / E_TH;
  int t = blockIdx.x;
  const int txb = t % tiles_x; t /= tiles_x;
  const int tyb = t % tiles_y; t /= tiles_y;
  const int n = t;
  const int iy0 = tyb * E_TH, ix0 = txb * E_TW;
  const int kgs = tid & 3, kgl = lane >> 4;
  ChanCoef<SRC> cc;
  BnFoldRegs fr;
  const bool folded = SRC == SRC_BNRELU && a.fold.acc != nullptr;
  EDGE_STAMP(0);
  if (folded) bn_fold_load<32>(a.fold, fr);       // before the patch loads: results return in issue order
  RawPiece<SRC> raw[NPA];
  bool val[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    int qq = tid + i * 256;
    int pix = qq >> 2;
    int pr = pix / PW, pc = pix % PW;
    int iy = iy0 + pr, ix = ix0 + pc;
    val[i] = (pix < NPIX) && (iy < a.Hin) && (ix < a.Win);
    size_t off = (((size_t)n * a.Hin + iy) * a.Win + ix) * 32 + kgs * 8;
    load_piece<SRC>(a.src, off, val[i], raw[i]);
  }
  {   // coefficient table of the 32-channel source layer (folded BatchNorm finalize) while the loads are in flight
    __shared__ float coef_tab[4 * 32];
    const float* coefp = a.src.coef;
    if (folded) {
      bn_fold_fwd_finish<32>(a.fold, fr, coef_tab, reinterpret_cast<long long*>(sl), blockIdx.x == 0);
      coefp = coef_tab;
    }
    cc.load(coefp, 32, kgs * 8);
  }
  EDGE_STAMP(1);
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    int qq = tid + i * 256;
    if (qq < NPIX * 4)
      *reinterpret_cast<uint4*>(patch + (qq >> 2) * PIX_STRIDE + kgs * 8) = transform_piece<SRC>(raw[i], val[i], cc);
  }
  EDGE_STAMP(2);
  __syncthreads();
  EDGE_STAMP(3);
  f32x4 acc[2];
  acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    bf16x8 bfr = *reinterpret_cast<const bf16x8*>(a.wjoint + (lane & 15) * 128 + nb * 32 + kgl * 8);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      int pos = (wave * 2 + mi) * 16 + (lane & 15);
      int ty = pos / E_TW, tx = pos % E_TW;
      bf16x8 af = *reinterpret_cast<const bf16x8*>(patch + ((ty + (nb >> 1)) * PW + tx + (nb & 1)) * PIX_STRIDE + kgl * 8);
      acc[mi] = mfma16(af, bfr, acc[mi]);
    }
  }
  EDGE_STAMP(4);
  __syncthreads();                 // every wave has read its patch fragments: the tiles below overwrite the patch
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) sl[((wave * 2 + mi) * 16 + (lane >> 4) * 4 + r) * 17 + (lane & 15)] = acc[mi][r];
  __syncthreads();
  EDGE_STAMP(5);
  // Elementwise pass, one thread per OUTPUT PIXEL (2 pixels per thread: oy = tid / 64 + 4 i, ox = tid % 64), all three channels: the
  // target / x_hat accesses stay coalesced per plane (a wave = one 64-pixel row segment of a plane), and the gradient pixel
  // [g0, g1, g2, 0] leaves as ONE 8-byte NHWC4 store.  (Round 2 went element by element in NCHW order -- 6 elements per thread, each
  // with its own 64-bit address arithmetic -- and staged the gradient tile through LDS with 2-byte scatter writes, a clear and two
  // more barriers: by the stamps this phase was 7.7-12 K of a workgroup's 23-38 K cycles, all of it VALU issue at 8 workgroups per CU.)
  float lsum = 0.f, gsum[3] = {0.f, 0.f, 0.f};
  const float b0 = a.bias[0], b1 = a.bias[1], b2 = a.bias[2];
  const size_t plane = (size_t)Hout * Wout;
  const int ox = tid & 63;
  float xt[2][3];                    // the six target values of this thread, requested together (the stores below may not be reordered against loads)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int co = 0; co < 3; ++co)
      xt[i][co] = a.x ? a.x[((size_t)n * 3 + co) * plane + (size_t)(2 * iy0 + (tid >> 6) + 4 * i) * Wout + 2 * ix0 + ox] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = (tid >> 6) + 4 * i;
    const int pos = (oy >> 1) * E_TW + (ox >> 1), ph = (oy & 1) * 2 + (ox & 1);
    const float* sp = sl + pos * 17 + ph * 3;
    const float s3[3] = {sp[0] + b0, sp[1] + b1, sp[2] + b2};
    const size_t pix = (size_t)(2 * iy0 + oy) * Wout + 2 * ix0 + ox;
    const size_t gi = (size_t)n * 3 * plane + pix;
    uint32_t gb[3] = {0u, 0u, 0u};
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const float xh = 1.0f / (1.0f + __expf(-s3[co]));
      if (a.x_hat) a.x_hat[gi + co * plane] = xh;
      if (a.x) {
        const float d = xh - xt[i][co];
        lsum = fmaf(d, d, lsum);
        if (a.g4) {
          gb[co] = f2bf(a.gscale * d * xh * (1.0f - xh));
          gsum[co] += bf2f(gb[co]);
        }
      }
    }
    if (a.g4) *reinterpret_cast<uint2*>(a.g4 + ((size_t)n * plane + pix) * 4) = make_uint2(gb[0] | (gb[1] << 16), gb[2]);
  }
  EDGE_STAMP(6);
  if (a.loss_part) {
    float v[4] = {lsum, gsum[0], gsum[1], gsum[2]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v[k] += __shfl_xor(v[k], o);
      if (lane == 0) redl[wave][k] = v[k];
    }
    __syncthreads();
    if (tid < 4) a.loss_part[(size_t)blockIdx.x * 4 + tid] = redl[0][tid] + redl[1][tid] + redl[2][tid] + redl[3][tid];
  }
  EDGE_STAMP(7);
}
template <int SRC>
__global__ __launch_bounds__(256) void deconv4_loss_kernel(Deconv4Args a) { deconv4_loss_body<SRC>(a); }
template <int SRC>
__global__ __launch_bounds__(256) void deconv4_loss_kernel_g(GroupPack<Deconv4Args> p, int gz) { deconv4_loss_body<SRC>(group_args<Deconv4Args>(gz)); }
