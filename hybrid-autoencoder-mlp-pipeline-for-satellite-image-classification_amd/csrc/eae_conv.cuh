// Implicit-GEMM kernels for the 3x3 stride-2 convolution family on NHWC bf16 activations (gfx950).
//
//   conv_s2_kernel   : out[n,oy,ox,co] = sum_{ky,kx,ci} T(in)[n,2oy-1+ky,2ox-1+kx,ci] * W[co][ky,kx][ci]
//                      = forward of nn.Conv2d(3x3,s2,p1) (R.md:292-304) and backward-data of nn.ConvTranspose2d.
//   deconv_s2_kernel : out[n,2i+py,2j+px,co] = sum_{taps(py,px),ci} T(in)[n,i+dy,j+dx,ci] * W[co][ky,kx][ci]
//                      = forward of nn.ConvTranspose2d(3x3,s2,p1,op1) (R.md:370-382) and backward-data of nn.Conv2d;
//                        the four sub-pixel phases are computed from one staged input patch (no zero insertion).
//
// Both stage the (transformed) input patch of a 128-row output tile once in LDS -- BatchNorm-apply+ReLU or
// BatchNorm-backward-apply happen once per input element while staging -- and read MFMA A fragments straight out of
// the patch (the im2col matrix is never materialised).  Weights of the current 32-channel K-chunk are staged in LDS
// as [n][tap][32].  4 waves in a 2(M) x 2(N) grid, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
// Epilogue: accumulators -> bf16 tile in LDS -> 16-byte coalesced NHWC stores, with the per-channel reductions that
// BatchNorm needs (forward: sum y, sum y^2; backward: sum g, sum g*xhat) taken from the values actually stored and
// written as deterministic per-tile partials (no atomics).
#pragma once
#include "eae_common.cuh"

struct ConvArgs {
  SrcDesc src;
  const bf16_t* wpack;     // [COUT][9][CIN] bf16 (tap = ky*3+kx)
  const float* bias;       // [COUT] (EPI_FWD) or nullptr
  bf16_t* out;             // NHWC bf16
  float* stat_part;        // [ntiles][2][COUT] or nullptr (no statistics, e.g. eval mode)
  const bf16_t* yprev;     // EPI_MASK: raw pre-BN tensor at the output positions
  const float* prev_coef;  // EPI_MASK: [4][COUT] s,t,mean,invstd of that BN
  int B, Hin, Win;         // input spatial size (conv: out = Hin/2; deconv: out = 2*Hin)
};

constexpr int PIX_STRIDE = 40;       // bf16 elements per staged pixel: 32 channels + 8 pad (80 B) -> conflict-free A reads
constexpr int W_STRIDE = 9 * 32 + 8; // bf16 elements per staged weight row (592 B)

// ---------------------------------------------------------------------------------------------------------------
// shared epilogue: tile [128][BN] (bf16, LDS) -> global, plus statistics partials
// rowmap(row) -> element offset of that output pixel (channel 0) in the NHWC tensor, or -1 if the row is invalid
// ---------------------------------------------------------------------------------------------------------------
template <int COUT, int BN, int EPI, class RowMap>
__device__ __forceinline__ void tile_epilogue(const ConvArgs& a, bf16_t* tile, float* red, int n0, int tile_id,
                                              int nrows, RowMap rowmap) {
  constexpr int TS = BN + 8;             // tile row stride (bf16)
  constexpr int CPR = BN / 8;            // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;         // rows per pass
  const int tid = threadIdx.x;
  const int c = tid % CPR, r0 = tid / CPR;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  float ps[8], pt[8], pm[8], pi[8];
  if (EPI == EPI_MASK) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int ch = n0 + c * 8 + j;
      ps[j] = a.prev_coef[ch]; pt[j] = a.prev_coef[COUT + ch];
      pm[j] = a.prev_coef[2 * COUT + ch]; pi[j] = a.prev_coef[3 * COUT + ch];
    }
  }
  for (int row = r0; row < nrows; row += RPP) {
    long off = rowmap(row);
    if (off < 0) continue;
    uint4 v = *reinterpret_cast<const uint4*>(tile + row * TS + c * 8);
    size_t g = (size_t)off + n0 + c * 8;
    if (EPI == EPI_FWD) {
      float f[8];
      unpack8(v, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]); }
    } else if (EPI == EPI_MASK) {
      uint4 yv = *reinterpret_cast<const uint4*>(a.yprev + g);
      float f[8], y[8];
      unpack8(v, f);
      unpack8(yv, y);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float act = fmaf(ps[j], y[j], pt[j]);
        f[j] = act > 0.f ? f[j] : 0.f;
        float xh = (y[j] - pm[j]) * pi[j];
        s1[j] += f[j];
        s2[j] = fmaf(f[j], xh, s2[j]);
      }
      v = pack8(f);     // exact: f are bf16 values or zero
    }
    *reinterpret_cast<uint4*>(a.out + g) = v;
  }
  if (EPI == EPI_PLAIN || a.stat_part == nullptr) return;
  // deterministic reduction over the RPP row-groups that share a channel chunk
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[(0 * RPP + r0) * BN + c * 8 + j] = s1[j];
    red[(1 * RPP + r0) * BN + c * 8 + j] = s2[j];
  }
  __syncthreads();
  if (tid < 2 * BN) {
    int which = tid / BN, ch = tid % BN;
    float acc = 0.f;
    for (int r = 0; r < RPP; ++r) acc += red[(which * RPP + r) * BN + ch];
    a.stat_part[((size_t)tile_id * 2 + which) * COUT + n0 + ch] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// conv 3x3 stride 2 pad 1
// tile = NI images x TH x TW output pixels (NI*TH*TW == 128)
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
__global__ __launch_bounds__(256) void conv_s2_kernel(ConvArgs a) {
  static_assert(NI * TH * TW == 128, "tile must hold 128 output pixels");
  static_assert(CIN % 32 == 0 && COUT % BN == 0 && (BN == 64 || BN == 128), "shape");
  constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1, NPIX = NI * PH * PW;
  constexpr int NT = BN / 32;                       // n-tiles per wave
  constexpr int NPA = (NPIX * 4 + 255) / 256;       // 16-byte patch pieces per thread
  constexpr int NWP = (BN * 36 + 255) / 256;        // 16-byte weight pieces per thread
  constexpr int PATCH_ELEMS = NPIX * PIX_STRIDE;
  constexpr int TILE_ELEMS = 128 * (BN + 8);
  constexpr int REGION0 = (PATCH_ELEMS > TILE_ELEMS ? PATCH_ELEMS : TILE_ELEMS);
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  bf16_t* patch = smem;
  bf16_t* wl = smem + REGION0;                      // [BN][W_STRIDE]; reused as the fp32 reduction scratch afterwards

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int Hout = a.Hin >> 1, Wout = a.Win >> 1;
  const int tiles_x = Wout / TW, tiles_y = Hout / TH;
  int t = blockIdx.x;
  const int txb = t % tiles_x; t /= tiles_x;
  const int tyb = t % tiles_y; t /= tiles_y;
  const int img0 = t * NI;
  const int n0 = blockIdx.y * BN;
  const int iy0 = 2 * tyb * TH - 1, ix0 = 2 * txb * TW - 1;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // patch pixel (for tap 0,0) of the A-fragment row owned by this lane in each of the wave's 4 m-tiles
  int pixbase[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    int m = (wm * 4 + mi) * 16 + (lane & 15);
    int img = m / (TH * TW), ty = (m / TW) % TH, tx = m % TW;
    pixbase[mi] = (img * PH + 2 * ty) * PW + 2 * tx;
  }
  const int kgl = lane >> 4;        // k-group of the lane inside an MFMA (8 channels)
  const int kgs = tid & 3;          // k-group staged by this thread (256 % 4 == 0 -> fixed per thread)

  for (int chunk = 0; chunk < CIN / 32; ++chunk) {
    if (chunk) __syncthreads();
    ChanCoef<SRC> cc;
    cc.load(a.src.coef, CIN, chunk * 32 + kgs * 8);
    // ---- stage the input patch (transform applied once per element)
    RawPiece<SRC> raw[NPA];
    bool val[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int q = tid + i * 256;
      int pix = q >> 2;
      int img = pix / (PH * PW), rem = pix % (PH * PW);
      int pr = rem / PW, pc = rem % PW;
      int iy = iy0 + pr, ix = ix0 + pc, n = img0 + img;
      val[i] = (pix < NPIX) && (n < a.B) && (iy >= 0) && (iy < a.Hin) && (ix >= 0) && (ix < a.Win);
      size_t off = (((size_t)n * a.Hin + iy) * a.Win + ix) * CIN + chunk * 32 + kgs * 8;
      load_piece<SRC>(a.src, off, val[i], raw[i]);
    }
    // ---- stage the weights of this K-chunk: [BN][9 taps][32 ch]
    uint4 wraw[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      int q = tid + i * 256;
      int n = q / 36, tap = (q >> 2) % 9;
      wraw[i] = make_uint4(0, 0, 0, 0);
      if (q < BN * 36)
        wraw[i] = *reinterpret_cast<const uint4*>(a.wpack + ((size_t)(n0 + n) * 9 + tap) * CIN + chunk * 32 + kgs * 8);
    }
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int q = tid + i * 256;
      if (q < NPIX * 4)
        *reinterpret_cast<uint4*>(patch + (q >> 2) * PIX_STRIDE + kgs * 8) = transform_piece<SRC>(raw[i], val[i], cc);
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      int q = tid + i * 256;
      int n = q / 36, tap = (q >> 2) % 9;
      if (q < BN * 36) *reinterpret_cast<uint4*>(wl + n * W_STRIDE + tap * 32 + kgs * 8) = wraw[i];
    }
    __syncthreads();
    // ---- 9 taps x one K=32 MFMA step
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * PW + (tap % 3);
      bf16x8 af[4], bfr[NT];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        af[mi] = *reinterpret_cast<const bf16x8*>(patch + (pixbase[mi] + toff) * PIX_STRIDE + kgl * 8);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        bfr[ni] = *reinterpret_cast<const bf16x8*>(wl + (wn * (BN / 2) + ni * 16 + (lane & 15)) * W_STRIDE + tap * 32 + kgl * 8);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma16(af[mi], bfr[ni], acc[mi][ni]);
    }
  }
  // ---- epilogue
  __syncthreads();
  bf16_t* tile = smem;
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    int col = wn * (BN / 2) + ni * 16 + (lane & 15);
    float bv = (EPI == EPI_FWD) ? a.bias[n0 + col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = (wm * 4 + mi) * 16 + (lane >> 4) * 4 + r;
        tile[row * (BN + 8) + col] = (bf16_t)f2bf(acc[mi][ni][r] + bv);
      }
  }
  __syncthreads();
  const int B = a.B;
  auto rowmap = [=](int row) -> long {
    int img = row / (TH * TW), ty = (row / TW) % TH, tx = row % TW;
    int n = img0 + img;
    if (n >= B) return -1;
    return (((long)n * Hout + (tyb * TH + ty)) * Wout + (txb * TW + tx)) * COUT;
  };
  tile_epilogue<COUT, BN, EPI>(a, tile, reinterpret_cast<float*>(wl), n0, blockIdx.x, 128, rowmap);
}

template <int CIN, int COUT, int BN, int TW, int TH, int NI>
constexpr size_t conv_s2_smem() {
  constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1, NPIX = NI * PH * PW;
  constexpr int PATCH_ELEMS = NPIX * PIX_STRIDE, TILE_ELEMS = 128 * (BN + 8);
  constexpr int REGION0 = (PATCH_ELEMS > TILE_ELEMS ? PATCH_ELEMS : TILE_ELEMS);
  constexpr size_t wbytes = (size_t)BN * W_STRIDE * 2;
  constexpr size_t redbytes = (size_t)2 * (256 / (BN / 8)) * BN * 4;
  return (size_t)REGION0 * 2 + (wbytes > redbytes ? wbytes : redbytes);
}

// ---------------------------------------------------------------------------------------------------------------
// transposed conv 3x3 stride 2 pad 1 output_padding 1, all four phases from one input patch
// tile = NI images x TH x TW INPUT positions (NI*TH*TW == 32) -> 128 output pixels
// M-row ordering inside the tile: row = phase*32 + position  (an MFMA m-tile never mixes phases)
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
__global__ __launch_bounds__(256) void deconv_s2_kernel(ConvArgs a) {
  static_assert(NI * TH * TW == 32, "tile must hold 32 input positions");
  static_assert(CIN % 32 == 0 && COUT % BN == 0 && (BN == 32 || BN == 64 || BN == 128), "shape");
  constexpr int PH = TH + 1, PW = TW + 1, NPIX = NI * PH * PW;
  constexpr int NT = BN / 32;                       // n-tiles per wave  (BN=32: one n-tile, waves split N as 2 x 16)
  constexpr int NPA = (NPIX * 4 + 255) / 256;
  constexpr int NWP = (BN * 36 + 255) / 256;
  constexpr int PATCH_ELEMS = NPIX * PIX_STRIDE;
  constexpr int TILE_ELEMS = 128 * (BN + 8);
  constexpr int REGION0 = (PATCH_ELEMS > TILE_ELEMS ? PATCH_ELEMS : TILE_ELEMS);
  constexpr int NTW = (NT > 0 ? NT : 1);
  constexpr int NHALF = (BN >= 32 ? BN / 2 : BN);
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  bf16_t* patch = smem;
  bf16_t* wl = smem + REGION0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int Hout = a.Hin * 2, Wout = a.Win * 2;
  const int tiles_x = a.Win / TW, tiles_y = a.Hin / TH;
  int t = blockIdx.x;
  const int txb = t % tiles_x; t /= tiles_x;
  const int tyb = t % tiles_y; t /= tiles_y;
  const int img0 = t * NI;
  const int n0 = blockIdx.y * BN;
  const int iy0 = tyb * TH, ix0 = txb * TW;

  // wave (wm, wn): positions m-tile wm (16 positions) for all 4 phases; n half wn
  f32x4 acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int pixbase;
  {
    int pos = wm * 16 + (lane & 15);
    int img = pos / (TH * TW), ty = (pos / TW) % TH, tx = pos % TW;
    pixbase = (img * PH + ty) * PW + tx;
  }
  const int kgl = lane >> 4;
  const int kgs = tid & 3;
  constexpr int NCOLW = (BN == 32 ? 16 : BN / 2);   // columns per wave

  for (int chunk = 0; chunk < CIN / 32; ++chunk) {
    if (chunk) __syncthreads();
    ChanCoef<SRC> cc;
    cc.load(a.src.coef, CIN, chunk * 32 + kgs * 8);
    RawPiece<SRC> raw[NPA];
    bool val[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int q = tid + i * 256;
      int pix = q >> 2;
      int img = pix / (PH * PW), rem = pix % (PH * PW);
      int pr = rem / PW, pc = rem % PW;
      int iy = iy0 + pr, ix = ix0 + pc, n = img0 + img;
      val[i] = (pix < NPIX) && (n < a.B) && (iy < a.Hin) && (ix < a.Win);
      size_t off = (((size_t)n * a.Hin + iy) * a.Win + ix) * CIN + chunk * 32 + kgs * 8;
      load_piece<SRC>(a.src, off, val[i], raw[i]);
    }
    uint4 wraw[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      int q = tid + i * 256;
      int n = q / 36, tap = (q >> 2) % 9;
      wraw[i] = make_uint4(0, 0, 0, 0);
      if (q < BN * 36)
        wraw[i] = *reinterpret_cast<const uint4*>(a.wpack + ((size_t)(n0 + n) * 9 + tap) * CIN + chunk * 32 + kgs * 8);
    }
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int q = tid + i * 256;
      if (q < NPIX * 4)
        *reinterpret_cast<uint4*>(patch + (q >> 2) * PIX_STRIDE + kgs * 8) = transform_piece<SRC>(raw[i], val[i], cc);
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      int q = tid + i * 256;
      int n = q / 36, tap = (q >> 2) % 9;
      if (q < BN * 36) *reinterpret_cast<uint4*>(wl + n * W_STRIDE + tap * 32 + kgs * 8) = wraw[i];
    }
    __syncthreads();
    // the four neighbour offsets (dy,dx) of a position; tap (ky,kx) reads neighbour dy = (ky==0), dx = (kx==0) and
    // feeds phase py = (ky!=1), px = (kx!=1)   [oy = 2*iy - 1 + ky]
    bf16x8 af[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
      af[nb] = *reinterpret_cast<const bf16x8*>(patch + (pixbase + (nb >> 1) * PW + (nb & 1)) * PIX_STRIDE + kgl * 8);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int nb = ((ky == 0) ? 2 : 0) + ((kx == 0) ? 1 : 0);
      const int ph = ((ky != 1) ? 2 : 0) + ((kx != 1) ? 1 : 0);
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) {
        bf16x8 bfr = *reinterpret_cast<const bf16x8*>(wl + (wn * NCOLW + ni * 16 + (lane & 15)) * W_STRIDE + tap * 32 + kgl * 8);
        acc[ph][ni] = mfma16(af[nb], bfr, acc[ph][ni]);
      }
    }
  }
  __syncthreads();
  bf16_t* tile = smem;
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) {
    int col = wn * NCOLW + ni * 16 + (lane & 15);
    float bv = (EPI == EPI_FWD) ? a.bias[n0 + col] : 0.f;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = ph * 32 + wm * 16 + (lane >> 4) * 4 + r;
        tile[row * (BN + 8) + col] = (bf16_t)f2bf(acc[ph][ni][r] + bv);
      }
  }
  __syncthreads();
  const int B = a.B;
  auto rowmap = [=](int row) -> long {
    int ph = row >> 5, pos = row & 31;
    int img = pos / (TH * TW), ty = (pos / TW) % TH, tx = pos % TW;
    int n = img0 + img;
    if (n >= B) return -1;
    int oy = 2 * (iy0 + ty) + (ph >> 1), ox = 2 * (ix0 + tx) + (ph & 1);
    return (((long)n * Hout + oy) * Wout + ox) * COUT;
  };
  tile_epilogue<COUT, BN, EPI>(a, tile, reinterpret_cast<float*>(wl), n0, blockIdx.x, 128, rowmap);
}

template <int CIN, int COUT, int BN, int TW, int TH, int NI>
constexpr size_t deconv_s2_smem() {
  constexpr int PH = TH + 1, PW = TW + 1, NPIX = NI * PH * PW;
  constexpr int PATCH_ELEMS = NPIX * PIX_STRIDE, TILE_ELEMS = 128 * (BN + 8);
  constexpr int REGION0 = (PATCH_ELEMS > TILE_ELEMS ? PATCH_ELEMS : TILE_ELEMS);
  constexpr size_t wbytes = (size_t)BN * W_STRIDE * 2;
  constexpr size_t redbytes = (size_t)2 * (256 / (BN / 8)) * BN * 4;
  return (size_t)REGION0 * 2 + (wbytes > redbytes ? wbytes : redbytes);
}
