// Weight-gradient kernel of the 3x3 stride-2 conv / transposed-conv layers (aten::convolution_backward's wgrad,
// 39 % of the reference's CPU step time).
//
//   R[tap][cs][cb] = sum over positions (n,oy,ox) of  S[n,oy,ox,cs] * Bg[n,2oy-1+ky,2ox-1+kx,cb]
//
//   conv   layer: S = dy (gradient of the conv output, small map), Bg = input activation (big map)  -> dW[co=cs][ci=cb][ky][kx]
//   deconv layer: S = input activation (small map), Bg = gradient of the deconv output (big map)    -> dW[ci=cs][co=cb][ky][kx]
// i.e. both land directly in the reference's parameter layout [cs][cb][3][3].
//
// The reduction runs over pixels, which are the *strided* dimension of NHWC tensors, so both MFMA operands are read
// with the hardware transposing LDS read (ds_read_b64_tr_b16): the S tile and the Bg patch sit in LDS in their natural
// [pixel][channel] order (staged with 16-byte loads, load transforms applied once per element) and each lane supplies
// the addresses of the gathered patch rows of its tap.
// One block owns a (64 cs) x (32 cb) x 9-tap output block and a slice of the positions; it writes an fp32 partial in
// the reference layout, partials are summed in a fixed order by reduce_slices_kernel (deterministic, no atomics).
#pragma once
#include "eae_common.cuh"
#include "eae_igemm.cuh"

struct WgradArgs {
  SrcDesc small, big;
  float* part;            // [nslices][9][CS][CB]
  int B, Hs, Ws;          // small-map spatial size (big map = 2Hs x 2Ws)
  int tiles_per_block, ntiles, nslices;
};

constexpr int S_STRIDE = 72;   // bf16 elements per staged S row: 64 channels + 8 pad (144 B)

// Register budget of two workgroups per CU (<= 256 VGPR + AGPR) although the grid only has one per CU: the kernel shares the
// CUs with the backward-data chain of the main stream, and at 242 + 72 registers it kept those kernels' workgroups off the
// SIMDs (step 0.585 -> 0.571 ms; a budget of three spills 300-450 bytes and costs 0.80 ms).
template <int CS, int CB, int TW, int TH, int NI, int SMODE, int BMODE>
__global__ __launch_bounds__(256, 2) void wgrad_s2_kernel(WgradArgs a) {
  static_assert(NI * TH * TW == 128, "tile must hold 128 positions");
  static_assert(CS % 64 == 0 && CB % 32 == 0, "shape");
  constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1, NPIX = NI * PH * PW;
  constexpr int NPA = (NPIX * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  bf16_t* patch = smem;                          // [NPIX][PIX_STRIDE]
  bf16_t* sl = smem + NPIX * PIX_STRIDE;         // [128][S_STRIDE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup -> (position slice, output block): the (CS/64)*(CB/32) output blocks of one slice read the same tiles, so they
  // sit on the same XCD, adjacent in dispatch order (ids go round-robin over the 8 XCDs), and share them through that L2
  constexpr int NBLK = (CS / 64) * (CB / 32);
  int slice, oblk;
  {
    const int bid = blockIdx.x;
    if (NBLK == 1) { slice = bid; oblk = 0; }
    else if (a.nslices % 8 == 0) { const int xcd = bid & 7, idx = bid >> 3; oblk = idx % NBLK; slice = (idx / NBLK) * 8 + xcd; }
    else { oblk = bid % NBLK; slice = bid / NBLK; }
  }
  const int cs0 = (oblk / (CB / 32)) * 64, cb0 = (oblk % (CB / 32)) * 32;
  const int Hb = a.Hs * 2, Wb = a.Ws * 2;
  const int tiles_x = a.Ws / TW, tiles_y = a.Hs / TH;
  const int kgs4 = tid & 3, kgs8 = tid & 7;
  ChanCoef<BMODE> ccb;
  ccb.load(a.big.coef, CB, cb0 + kgs4 * 8);
  ChanCoef<SMODE> ccs;
  ccs.load(a.small.coef, CS, cs0 + kgs8 * 8);
  // wave: cs-tiles {2*(wave&1), +1} x cb-tile (wave>>1) x 9 taps
  const int it0 = 2 * (wave & 1), jt = wave >> 1;
  f32x4 acc[9][2];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) acc[t9][0] = acc[t9][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;

  for (int ti = 0; ti < a.tiles_per_block; ++ti) {
    int t = slice * a.tiles_per_block + ti;
    if (t >= a.ntiles) break;
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y; t /= tiles_y;
    const int img0 = t * NI;
    const int iy0 = 2 * tyb * TH - 1, ix0 = 2 * txb * TW - 1;
    __syncthreads();
    // ---- big-map patch (32 channels cb0..cb0+31)
    RawPiece<BMODE> raw[NPA];
    bool val[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int qq = tid + i * 256;
      int pix = qq >> 2;
      int img = pix / (PH * PW), rem = pix % (PH * PW);
      int pr = rem / PW, pc = rem % PW;
      int iy = iy0 + pr, ix = ix0 + pc, n = img0 + img;
      val[i] = (pix < NPIX) && (n < a.B) && (iy >= 0) && (iy < Hb) && (ix >= 0) && (ix < Wb);
      size_t off = (((size_t)n * Hb + iy) * Wb + ix) * CB + cb0 + kgs4 * 8;
      load_piece<BMODE>(a.big, off, val[i], raw[i]);
    }
    // ---- small-map tile [128 positions][64 channels cs0..]
    RawPiece<SMODE> sraw[4];
    bool sval[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int qq = tid + i * 256;
      int m = qq >> 3;
      int img = m / (TH * TW), ty = (m / TW) % TH, tx = m % TW;
      int n = img0 + img;
      sval[i] = n < a.B;
      size_t off = (((size_t)n * a.Hs + tyb * TH + ty) * a.Ws + txb * TW + tx) * CS + cs0 + kgs8 * 8;
      load_piece<SMODE>(a.small, off, sval[i], sraw[i]);
    }
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int qq = tid + i * 256;
      if (qq < NPIX * 4)
        *reinterpret_cast<uint4*>(patch + (qq >> 2) * PIX_STRIDE + kgs4 * 8) = transform_piece<BMODE>(raw[i], val[i], ccb);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int qq = tid + i * 256;
      *reinterpret_cast<uint4*>(sl + (qq >> 3) * S_STRIDE + kgs8 * 8) = transform_piece<SMODE>(sraw[i], sval[i], ccs);
    }
    __syncthreads();
    // ---- 4 K-steps of 32 positions
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int m_lo = ks * 32 + 8 * g + q, m_hi = m_lo + 4;
      bf16x8 sa[2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
        sa[ii] = tr_frag(sl + m_lo * S_STRIDE + (it0 + ii) * 16 + 4 * p, sl + m_hi * S_STRIDE + (it0 + ii) * 16 + 4 * p);
      int pb_lo, pb_hi;
      {
        int img = m_lo / (TH * TW), ty = (m_lo / TW) % TH, tx = m_lo % TW;
        pb_lo = (img * PH + 2 * ty) * PW + 2 * tx;
        img = m_hi / (TH * TW); ty = (m_hi / TW) % TH; tx = m_hi % TW;
        pb_hi = (img * PH + 2 * ty) * PW + 2 * tx;
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * PW + (tap % 3);
        bf16x8 bb = tr_frag(patch + (pb_lo + toff) * PIX_STRIDE + jt * 16 + 4 * p,
                            patch + (pb_hi + toff) * PIX_STRIDE + jt * 16 + 4 * p);
        acc[tap][0] = mfma16(sa[0], bb, acc[tap][0]);
        acc[tap][1] = mfma16(sa[1], bb, acc[tap][1]);
      }
    }
  }
  // ---- store the partial as [tap][cs][cb] (16 consecutive cb per lane group -> 64-byte segments); the reduce kernel
  //      permutes into the reference layout [cs][cb][3][3]
  float* out = a.part + (size_t)slice * ((size_t)CS * CB * 9);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int cs = cs0 + (it0 + ii) * 16 + (lane >> 4) * 4 + r;
        int cb = cb0 + jt * 16 + (lane & 15);
        out[((size_t)tap * CS + cs) * CB + cb] = acc[tap][ii][r];
      }
}

template <int TW, int TH, int NI>
constexpr size_t wgrad_smem() {
  constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1, NPIX = NI * PH * PW;
  return (size_t)(NPIX * PIX_STRIDE + 128 * S_STRIDE) * 2;
}

// Deterministic slice reductions: block = 16 float4 lanes x 16 slice lanes; every thread sums its slices in order, then
// the 16 slice lanes are combined in a fixed order through LDS.
template <class Store>
__device__ __forceinline__ void reduce_slices_body(const float* __restrict__ part, int nslices, long n4, Store store) {
  __shared__ float4 red[16][16];
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + lx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4)
    for (int sidx = ly; sidx < nslices; sidx += 16) {
      float4 v = reinterpret_cast<const float4*>(part)[(long)sidx * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[ly][lx] = s;
  __syncthreads();
  if (ly == 0 && i < n4) {
    float4 r = red[0][lx];
#pragma unroll
    for (int k = 1; k < 16; ++k) { r.x += red[k][lx].x; r.y += red[k][lx].y; r.z += red[k][lx].z; r.w += red[k][lx].w; }
    store(i, r);
  }
}

// out[i] = sum_s part[s][i]
static __global__ EAE_NO_PK __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ part, int nslices, long n4,
                                                                   float* __restrict__ out, float scale) {
  reduce_slices_body(part, nslices, n4, [=](long i, float4 r) {
    r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
    reinterpret_cast<float4*>(out)[i] = r;
  });
}

// out[cs][cb][tap] = sum_s part[s][tap][cs][cb]; thread = 4 consecutive cb of one (tap, cs)
static __global__ EAE_NO_PK __launch_bounds__(256) void reduce_slices_perm_kernel(const float* __restrict__ part, int nslices, int CS, int CB,
                                                                        float* __restrict__ out) {
  reduce_slices_body(part, nslices, (long)9 * CS * CB / 4, [=](long i, float4 r) {
    long e = i * 4;
    int cb = (int)(e % CB); long t2 = e / CB; int cs = (int)(t2 % CS); int tap = (int)(t2 / CS);
    float* o = out + ((size_t)cs * CB + cb) * 9 + tap;
    o[0] = r.x; o[9] = r.y; o[18] = r.z; o[27] = r.w;
  });
}

static inline unsigned reduce_slices_grid(long n4) { return (unsigned)((n4 + 15) / 16); }

// Tall variant for few outputs and many slices (conv1 / deconv4 weight gradients: 216 float4, 512 slices): 4 float4 columns
// x 64 slice lanes per block -> 4x the blocks and a quarter of the serial loads per thread; fixed summation order.
static __global__ EAE_NO_PK __launch_bounds__(256) void reduce_slices_tall_kernel(const float* __restrict__ part, int nslices, long n4,
                                                                                float* __restrict__ out) {
  __shared__ float4 red[64][4];
  const int lx = threadIdx.x & 3, ly = threadIdx.x >> 2;
  const long i = (long)blockIdx.x * 4 + lx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4)
    for (int sidx = ly; sidx < nslices; sidx += 64) {
      float4 v = reinterpret_cast<const float4*>(part)[(long)sidx * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[ly][lx] = s;
  __syncthreads();
  if (ly == 0 && i < n4) {
    float4 r = red[0][lx];
    for (int k = 1; k < 64; ++k) { r.x += red[k][lx].x; r.y += red[k][lx].y; r.z += red[k][lx].z; r.w += red[k][lx].w; }
    reinterpret_cast<float4*>(out)[i] = r;
  }
}
