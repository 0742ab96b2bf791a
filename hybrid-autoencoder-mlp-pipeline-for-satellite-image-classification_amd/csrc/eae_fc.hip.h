// GEMMs of the two latent projections (enc.fc: nn.Linear(256*h*w, L), R.md:309; dec.fc: nn.Linear(L, 256*h*w), R.md:365).
// Activations stay NHWC inside the engine, so the flattened feature index is k' = p*256 + c (p = pixel, c = channel)
// while the reference's nn.Flatten index is c*P + p; the permutation lives in the packed weights (eae_misc.hip).
//
//   fc_nt_kernel : C[m][n] = sum_k T(A)[m][k] * Wp[n][k]        (forward projections, backward-data)
//   fc_tn_kernel : R[i][j] = sum_b T(P)[b][i] * T(Q)[b][j]      (weight gradients; reduction over the batch -> both
//                                                                operands via the transposing LDS read)
#pragma once
#include "eae_common.hip.h"
#include "eae_igemm.hip.h"

enum { FCE_PARTIAL = 0,    // fp32 partial [kslice][M][N]       (split-K)
       FCE_BIAS_BF16 = 1,  // bf16(acc + bias[n]) -> [M][N]
       FCE_MASK = 2 };     // ReLU mask of the BN output at the same position + BN-backward partial sums

struct FcNtArgs {
  SrcDesc a;               // A [M][K]; SRC_F32: p0 is a float*; BNRELU: channel = k % 256
  const bf16_t* w;         // [N][K]
  int M, N, K;
  int klen;                // K-range per grid.z slice (multiple of 64)
  float* part;             // FCE_PARTIAL
  ConvArgs c;              // FCE_BIAS_BF16 / FCE_MASK: out, bias ([N]), stat_part, yprev, prev_coef ([4][256]);
                           // c.fold: BNRELU source -- coefficient table of the 256-channel source layer from its accumulators
                           // (no field of its own: eight of these blocks must fit one grouped launch, eae_group.h)
};

constexpr int FC_KC = 64;
constexpr int FC_LS = FC_KC + 8;   // LDS row stride (bf16)

template <int AMODE>
__device__ __forceinline__ uint4 fc_load_a(const SrcDesc& s, size_t off, bool valid, const float* coef, int ch0) {
  if (!valid) return make_uint4(0, 0, 0, 0);
  if (AMODE == SRC_F32) {
    const float* f = reinterpret_cast<const float*>(s.p0) + off;
    float4 lo = *reinterpret_cast<const float4*>(f), hi = *reinterpret_cast<const float4*>(f + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return pack8(v);
  }
  uint4 r = *reinterpret_cast<const uint4*>(s.p0 + off);
  if (AMODE == SRC_BNRELU) {
    float x[8];
    unpack8(r, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = relu_nan(fmaf(coef[ch0 + j], x[j], coef[256 + ch0 + j]));
    r = pack8(x);
  }
  return r;
}

// raw 16-byte piece of 8 consecutive operand elements as it comes back from memory (fp32 sources: two float4), and its transform
template <int MODE> struct FcRaw { uint4 v; };
template <> struct FcRaw<SRC_F32> { float4 lo, hi; };
// `off` must be in range for every thread (callers clamp the row of out-of-range threads): the load is UNCONDITIONAL -- a load under a
// per-thread condition makes hipcc branch around it and drain the whole vector-memory queue (s_waitcnt vmcnt(0)) at the join, which
// serialises a prefetch ring; out-of-range rows are zeroed by fc_finish_raw instead
template <int MODE>
__device__ __forceinline__ void fc_load_raw(const SrcDesc& s, size_t off, FcRaw<MODE>& r) {
  if constexpr (MODE == SRC_F32) {
    const float* f = reinterpret_cast<const float*>(s.p0) + off;
    r.lo = *reinterpret_cast<const float4*>(f);
    r.hi = *reinterpret_cast<const float4*>(f + 4);
  } else {
    r.v = *reinterpret_cast<const uint4*>(s.p0 + off);
  }
}
template <int MODE>
__device__ __forceinline__ uint4 fc_finish_raw(const FcRaw<MODE>& r, bool valid, const float* cs, const float* ct) {
  const uint32_t m = valid ? 0xffffffffu : 0u;
  uint4 o;
  if constexpr (MODE == SRC_F32) {
    float v[8] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w};
    o = pack8(v);
  } else if constexpr (MODE == SRC_BNRELU) {
    float x[8];
    unpack8(r.v, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = relu_nan(fmaf(cs[j], x[j], ct[j]));
    o = pack8(x);
  } else {
    o = r.v;
  }
  return make_uint4(o.x & m, o.y & m, o.z & m, o.w & m);
}

// tile 128 (M) x 64 (N); 4 waves 2x2 (each 64 x 32); K chunks of 64
// (bz: the workgroup's K-slice index -- blockIdx.z alone, blockIdx.z % gz in a grouped launch)
template <int AMODE, int EPI>
__device__ __forceinline__ void fc_nt_body(const FcNtArgs& a, const int bz) {
  eae_signal(a.c.sig, a.c.sig_val);
  __shared__ __attribute__((aligned(16))) bf16_t al[128 * FC_LS];      // A chunk; later the output tile [128][72]
  __shared__ __attribute__((aligned(16))) bf16_t wl[64 * FC_LS];
  __shared__ __attribute__((aligned(16))) float red[2 * 32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 64, kbeg = bz * a.klen;
  const int kgl = lane >> 4, kg8 = tid & 7;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __shared__ float coef_tab[(AMODE == SRC_BNRELU) ? 4 * 256 : 4];
  const float* coefp = a.a.coef;
  if (AMODE == SRC_BNRELU && a.c.fold.acc != nullptr) {
    bn_fold_fwd<256>(a.c.fold, coef_tab, reinterpret_cast<long long*>(red), blockIdx.x == 0 && blockIdx.y == 0 && bz == 0);
    coefp = coef_tab;
  }
  if constexpr (EPI == FCE_PARTIAL) {      // the split-K projections: several chunks per workgroup
    // K loop with a one-deep register prefetch: the raw 16-byte pieces of chunk c+1 are requested before the MFMAs of chunk c (the
    // load -> LDS -> barrier -> MFMA loop of rounds 1-2 exposed a full memory round trip per 64-wide chunk: 16 of them per workgroup at
    // the 256x256 shape).  Loads are unconditional (row clamped, value masked in fc_finish_raw); the branch around the prefetch is
    // uniform over the workgroup.
    const int nch = a.klen / FC_KC;
    // (named registers and a macro -- not arrays in a lambda or a loop inside the macro: both forms left the prefetched pieces in scratch)
    FcRaw<AMODE> ra0, ra1, ra2, ra3;
    uint4 rw0, rw1;
    const int arow0 = tid >> 3, arow1 = (tid + 256) >> 3, arow2 = (tid + 512) >> 3, arow3 = (tid + 768) >> 3;
    const size_t ga0 = (size_t)(m0 + arow0 < a.M ? m0 + arow0 : a.M - 1) * a.K + kg8 * 8, ga1 = (size_t)(m0 + arow1 < a.M ? m0 + arow1 : a.M - 1) * a.K + kg8 * 8;
    const size_t ga2 = (size_t)(m0 + arow2 < a.M ? m0 + arow2 : a.M - 1) * a.K + kg8 * 8, ga3 = (size_t)(m0 + arow3 < a.M ? m0 + arow3 : a.M - 1) * a.K + kg8 * 8;
    const size_t gw0 = (size_t)(n0 + arow0) * a.K + kg8 * 8, gw1 = (size_t)(n0 + arow1) * a.K + kg8 * 8;
#define FC_NT_ISSUE(K0) do { \
      fc_load_raw<AMODE>(a.a, ga0 + (K0), ra0); fc_load_raw<AMODE>(a.a, ga1 + (K0), ra1); \
      fc_load_raw<AMODE>(a.a, ga2 + (K0), ra2); fc_load_raw<AMODE>(a.a, ga3 + (K0), ra3); \
      rw0 = *reinterpret_cast<const uint4*>(a.w + gw0 + (K0)); rw1 = *reinterpret_cast<const uint4*>(a.w + gw1 + (K0)); \
    } while (0)
    FC_NT_ISSUE(kbeg);
    for (int ci = 0; ci < nch; ++ci) {
      const int k0 = kbeg + ci * FC_KC;
      if (ci) __syncthreads();
      const float* cs_ = coefp + ((k0 + kg8 * 8) & 255);
      *reinterpret_cast<uint4*>(al + arow0 * FC_LS + kg8 * 8) = fc_finish_raw<AMODE>(ra0, (m0 + arow0) < a.M, cs_, cs_ + 256);
      *reinterpret_cast<uint4*>(al + arow1 * FC_LS + kg8 * 8) = fc_finish_raw<AMODE>(ra1, (m0 + arow1) < a.M, cs_, cs_ + 256);
      *reinterpret_cast<uint4*>(al + arow2 * FC_LS + kg8 * 8) = fc_finish_raw<AMODE>(ra2, (m0 + arow2) < a.M, cs_, cs_ + 256);
      *reinterpret_cast<uint4*>(al + arow3 * FC_LS + kg8 * 8) = fc_finish_raw<AMODE>(ra3, (m0 + arow3) < a.M, cs_, cs_ + 256);
      *reinterpret_cast<uint4*>(wl + arow0 * FC_LS + kg8 * 8) = rw0;
      *reinterpret_cast<uint4*>(wl + arow1 * FC_LS + kg8 * 8) = rw1;
      if (ci + 1 < nch) FC_NT_ISSUE(k0 + FC_KC);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bfr[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          af[mi] = *reinterpret_cast<const bf16x8*>(al + ((wm * 4 + mi) * 16 + (lane & 15)) * FC_LS + ks * 32 + kgl * 8);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          bfr[ni] = *reinterpret_cast<const bf16x8*>(wl + (wn * 32 + ni * 16 + (lane & 15)) * FC_LS + ks * 32 + kgl * 8);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma16(af[mi], bfr[ni], acc[mi][ni]);
      }
    }
#undef FC_NT_ISSUE
  } else {      // one to four chunks (K = latent width): the plain loop (the prefetching form measured 19.5 vs 8.3 us for enc.fc backward-data)
    for (int k0 = kbeg; k0 < kbeg + a.klen; k0 += FC_KC) {
      if (k0 != kbeg) __syncthreads();
      uint4 av[4], wv[2];
  #pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = (tid + i * 256) >> 3;
        int k = k0 + kg8 * 8;
        av[i] = fc_load_a<AMODE>(a.a, (size_t)(m0 + row) * a.K + k, (m0 + row) < a.M, coefp, k & 255);
      }
  #pragma unroll
      for (int i = 0; i < 2; ++i) {
        int row = (tid + i * 256) >> 3;
        wv[i] = *reinterpret_cast<const uint4*>(a.w + (size_t)(n0 + row) * a.K + k0 + kg8 * 8);
      }
  #pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(al + ((tid + i * 256) >> 3) * FC_LS + kg8 * 8) = av[i];
  #pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(wl + ((tid + i * 256) >> 3) * FC_LS + kg8 * 8) = wv[i];
      __syncthreads();
  #pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bfr[2];
  #pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          af[mi] = *reinterpret_cast<const bf16x8*>(al + ((wm * 4 + mi) * 16 + (lane & 15)) * FC_LS + ks * 32 + kgl * 8);
  #pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          bfr[ni] = *reinterpret_cast<const bf16x8*>(wl + (wn * 32 + ni * 16 + (lane & 15)) * FC_LS + ks * 32 + kgl * 8);
  #pragma unroll
        for (int mi = 0; mi < 4; ++mi)
  #pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma16(af[mi], bfr[ni], acc[mi][ni]);
      }
    }
  }
  if (EPI == FCE_PARTIAL) {
    float* out = a.part + (size_t)bz * a.M * a.N;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = m0 + (wm * 4 + mi) * 16 + (lane >> 4) * 4 + r;
          int col = n0 + wn * 32 + ni * 16 + (lane & 15);
          if (row < a.M) out[(size_t)row * a.N + col] = acc[mi][ni][r];
        }
    return;
  }
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    int col = wn * 32 + ni * 16 + (lane & 15);
    float bv = (EPI == FCE_BIAS_BF16) ? a.c.bias[n0 + col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = (wm * 4 + mi) * 16 + (lane >> 4) * 4 + r;
        al[row * FC_LS + col] = (bf16_t)f2bf(acc[mi][ni][r] + bv);
      }
  }
  __syncthreads();
  const int M = a.M, N = a.N;
  const int pcol = n0 & ~255, c0 = n0 & 255;       // pixel base column / channel offset inside the 256-channel pixel
  auto rowmap = [=](int row) -> long { return (m0 + row) < M ? (long)(m0 + row) * N + pcol : -1; };
  const int tile_id = blockIdx.x * (N / 256) + (n0 >> 8);
  if (EPI == FCE_MASK) tile_epilogue<256, 64, EPI_MASK>(a.c, al, red, c0, tile_id, 128, rowmap);
  else tile_epilogue<256, 64, EPI_PLAIN>(a.c, al, red, c0, tile_id, 128, rowmap);
}
template <int AMODE, int EPI>
__global__ __launch_bounds__(256) void fc_nt_kernel(FcNtArgs a) { fc_nt_body<AMODE, EPI>(a, (int)blockIdx.z); }
template <int AMODE, int EPI>
__global__ __launch_bounds__(256) void fc_nt_kernel_g(GroupPack<FcNtArgs> p, int gz) { fc_nt_body<AMODE, EPI>(group_args<FcNtArgs>(gz), (int)(blockIdx.z % (unsigned)gz)); }

// z (or dz) = bias + sum over K-slices of the split-K partials (+ optional addends), fp32; 16 float4 lanes x 16 slice
// lanes per block, fixed summation order
struct FcReduceArgs { const float* part; int nsl, M, N; const float* bias; const float* addend; const float* addend2; float* out; };
static __device__ __forceinline__ EAE_NO_PK void fc_splitk_reduce_body(const FcReduceArgs& a) {
  const float* __restrict__ part = a.part; const int nsl = a.nsl, M = a.M, N = a.N;
  const float* __restrict__ bias = a.bias; const float* __restrict__ addend = a.addend; const float* __restrict__ addend2 = a.addend2;
  float* __restrict__ out = a.out;
  __shared__ float4 red[16][16];
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const long n4 = (long)M * N / 4;
  const long i = (long)blockIdx.x * 16 + lx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4)
    for (int k0 = ly; k0 < nsl; k0 += 16 * 4) {          // 4 loads in flight, summed in slice order
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + 16 * q;
        v[q] = k < nsl ? reinterpret_cast<const float4*>(part)[(long)k * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
  red[ly][lx] = s;
  __syncthreads();
  if (ly == 0 && i < n4) {
    float4 r = red[0][lx];
#pragma unroll
    for (int k = 1; k < 16; ++k) { r.x += red[k][lx].x; r.y += red[k][lx].y; r.z += red[k][lx].z; r.w += red[k][lx].w; }
    if (bias) { float4 b = *reinterpret_cast<const float4*>(bias + (i * 4) % N); r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w; }
    if (addend) { float4 b = reinterpret_cast<const float4*>(addend)[i]; r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w; }
    if (addend2) { float4 b = reinterpret_cast<const float4*>(addend2)[i]; r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w; }
    reinterpret_cast<float4*>(out)[i] = r;
  }
}
static __global__ EAE_NO_PK __launch_bounds__(256) void fc_splitk_reduce_kernel(FcReduceArgs a) { fc_splitk_reduce_body(a); }
static __global__ EAE_NO_PK __launch_bounds__(256) void fc_splitk_reduce_kernel_g(GroupPack<FcReduceArgs> p, int gz) { fc_splitk_reduce_body(group_args<FcReduceArgs>(gz)); }

// ---------------------------------------------------------------------------------------------------------------
// R[i][j] = sum_b P[b][i] * Q[b][j]; block = 64 (i) x 64 (j); whole batch reduced in chunks of 64 rows.
// Also emits colsum_P[i] = sum_b P[b][i] (bias gradient) from the blocks with blockIdx.y == 0.
// out_mode 0: R is [I][J] row-major, row index permuted   i' = p*256+c  ->  c*Pn + p   (dec.fc weight [4096][L])
// out_mode 1: R is [I][J] row-major, column index permuted j' = p*256+c ->  c*Pn + p   (enc.fc weight [L][4096])
// ---------------------------------------------------------------------------------------------------------------
struct FcTnArgs {
  SrcDesc p, q;            // P [Bt][I], Q [Bt][J]   (F32: p0 is float*; BNRELU: channel = col % 256)
  int Bt, I, J;
  float* out;              // reference-layout weight gradient
  float* colsum;           // bias gradient (reference order) or nullptr
  int out_mode, Pn;        // Pn = pixels per image of the flattened map
};


// The reduction runs over the batch in chunks of 64 rows; one block walks ALL chunks, so its time used to be (number of chunks) x
// (memory latency + staging + a handful of MFMAs): 8 exposed round trips at B=512 (22 us inside the step for 8 MB of operands).  The raw
// pieces of the next FC_TN_D chunks are now kept in flight in a register ring (compile-time slots), the transform / LDS staging of chunk
// c runs when its pieces have arrived, and the slot is re-issued for chunk c + FC_TN_D before the MFMAs of chunk c.
constexpr int FC_TN_D = 4;
template <int PMODE, int QMODE>
__device__ __forceinline__ void fc_tn_body(const FcTnArgs& a) {
  __shared__ __attribute__((aligned(16))) bf16_t pl[64 * FC_LS];
  __shared__ __attribute__((aligned(16))) bf16_t ql[64 * FC_LS];
  __shared__ float csum[32][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  const int kg8 = tid & 7;
  // out_mode 1 (enc.fc's weight gradient: columns permuted j' = p*256 + c -> c*Pn + p): the 64 columns of a tile are 16 consecutive channels
  // at 4 consecutive positions (column jl <-> position p0 + jl/16, channel c0 + jl%16) instead of 64 consecutive channels at one position,
  // so a lane's four accumulator tiles are four CONSECUTIVE output floats and the epilogue writes one 16-byte piece per row instead
  // of four 4-byte ones Pn floats apart (256x256 inputs: 141 us for this kernel, a 67 MB fp32 output written 4 bytes per KB).
  const bool pj = a.out_mode == 1 && (a.Pn & 3) == 0 && a.J == 256 * a.Pn;
  const int pj_c0 = (blockIdx.y & 15) * 16, pj_p0 = (blockIdx.y >> 4) * 4;
  const int qcol = pj ? (pj_p0 + (kg8 >> 1)) * 256 + pj_c0 + (kg8 & 1) * 8 : j0 + kg8 * 8;       // first of this thread's 8 Q columns
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  // BatchNorm-apply coefficients of this thread's 8 channels (fixed over the batch loop)
  float pcs[8], pct[8], qcs[8], qct[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pcs[j] = pct[j] = qcs[j] = qct[j] = 0.f;
    if (PMODE == SRC_BNRELU) { const int ch = ((i0 + kg8 * 8) & 255) + j; pcs[j] = a.p.coef[ch]; pct[j] = a.p.coef[256 + ch]; }
    if (QMODE == SRC_BNRELU) { const int ch = (qcol & 255) + j; qcs[j] = a.q.coef[ch]; qct[j] = a.q.coef[256 + ch]; }
  }
  const int nchunks = (a.Bt + 63) / 64;
  FcRaw<PMODE> rp[FC_TN_D][2];
  FcRaw<QMODE> rq[FC_TN_D][2];
  auto issue = [&](FcRaw<PMODE> (&xp)[2], FcRaw<QMODE> (&xq)[2], int c) __attribute__((always_inline)) {
    const int b0 = (c < nchunks ? c : nchunks - 1) * 64;         // (past the end: the last chunk again, never consumed)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = b0 + ((tid + i * 256) >> 3);
      row = row < a.Bt ? row : a.Bt - 1;                          // clamped: the load is unconditional, the value is zeroed in fc_finish_raw
      fc_load_raw<PMODE>(a.p, (size_t)row * a.I + i0 + kg8 * 8, xp[i]);
      fc_load_raw<QMODE>(a.q, (size_t)row * a.J + qcol, xq[i]);
    }
  };
#pragma unroll
  for (int d = 0; d < FC_TN_D; ++d) issue(rp[d], rq[d], d);
  for (int c0 = 0; c0 < nchunks; c0 += FC_TN_D) {
#pragma unroll
    for (int d = 0; d < FC_TN_D; ++d) {
      const int c = c0 + d;
      if (c >= nchunks) break;                         // uniform over the block
      if (c) __syncthreads();                          // the previous chunk's fragment reads are done
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid + i * 256) >> 3;
        const bool v = (c * 64 + row) < a.Bt;
        const uint4 pv = fc_finish_raw<PMODE>(rp[d][i], v, pcs, pct);
        const uint4 qv = fc_finish_raw<QMODE>(rq[d][i], v, qcs, qct);
        float f[8];
        unpack8(pv, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += f[j];
        *reinterpret_cast<uint4*>(pl + row * FC_LS + kg8 * 8) = pv;
        *reinterpret_cast<uint4*>(ql + row * FC_LS + kg8 * 8) = qv;
      }
      issue(rp[d], rq[d], c + FC_TN_D);            // unconditional (clamped): no branch around loads inside the ring
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r_lo = ks * 32 + 8 * g + q, r_hi = r_lo + 4;
        bf16x8 pa = tr_frag(pl + r_lo * FC_LS + wave * 16 + 4 * p, pl + r_hi * FC_LS + wave * 16 + 4 * p);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          bf16x8 qb = tr_frag(ql + r_lo * FC_LS + jt * 16 + 4 * p, ql + r_hi * FC_LS + jt * 16 + 4 * p);
          acc[jt] = mfma16(pa, qb, acc[jt]);
        }
      }
    }
  }
  const int Pn = a.Pn;
  auto perm = [=](int k2) -> int { return (k2 & 255) * Pn + (k2 >> 8); };
  if (pj) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wave * 16 + (lane >> 4) * 4 + r;
      const size_t o = (size_t)i * a.J + (size_t)(pj_c0 + (lane & 15)) * Pn + pj_p0;
      *reinterpret_cast<float4*>(a.out + o) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
  } else {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = i0 + wave * 16 + (lane >> 4) * 4 + r;
        int j = j0 + jt * 16 + (lane & 15);
        size_t o = (a.out_mode == 0) ? (size_t)perm(i) * a.J + j : (size_t)i * a.J + perm(j);
        a.out[o] = acc[jt][r];
      }
  }
  if (a.colsum && blockIdx.y == 0) {
    // thread (row-group tid>>3, chunk kg8) holds partial column sums of its rows; reduce over the 32 row-groups
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[tid >> 3][kg8 * 8 + j] = cs[j];
    __syncthreads();
    if (tid < 64) {
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += csum[r][tid];
      int i = i0 + tid;
      a.colsum[a.out_mode == 0 ? perm(i) : i] = s;
    }
  }
}
template <int PMODE, int QMODE>
__global__ __launch_bounds__(256) void fc_tn_kernel(FcTnArgs a) { fc_tn_body<PMODE, QMODE>(a); }
template <int PMODE, int QMODE>
__global__ __launch_bounds__(256) void fc_tn_kernel_g(GroupPack<FcTnArgs> p, int gz) { fc_tn_body<PMODE, QMODE>(group_args<FcTnArgs>(gz)); }
