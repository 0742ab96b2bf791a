// Small kernels: BatchNorm finalisation (forward statistics -> apply coefficients + running stats, backward
// reductions -> dgamma/dbeta + apply coefficients), weight packing (fp32 master -> bf16 kernel layouts),
// fused multi-tensor Adam, loss bookkeeping.
#include "eae_internal.h"
#include "eae_common.hip.h"
#include "eae_misc.h"


// Sum the partials of ONE channel: part = [2][C][ntiles] (channel-major, so the reads are contiguous); 256 threads stride
// over the tiles with fp64 accumulation, then a fixed-order reduction: xor-butterfly inside each wave (every lane ends with
// the wave's sum), the four wave sums added in wave order -> bitwise reproducible, one barrier instead of a 9-step LDS tree.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void reduce_partials_ch(const float* __restrict__ part, int ntiles, int C, int ch, double* red,
                                                   double& a, double& b) {
  const int tid = threadIdx.x;
  const float* p1 = part + (size_t)ch * ntiles;
  const float* p2 = part + ((size_t)C + ch) * ntiles;
  double s1 = 0.0, s2 = 0.0;
  for (int t = tid; t < ntiles; t += 256) { s1 += (double)p1[t]; s2 += (double)p2[t]; }
  s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
  if ((tid & 63) == 0) { red[tid >> 6] = s1; red[4 + (tid >> 6)] = s2; }
  __syncthreads();
  a = ((red[0] + red[1]) + red[2]) + red[3];
  b = ((red[4] + red[5]) + red[6]) + red[7];
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm forward finalize.  part = [2][C][ntiles] (sum y, sum y^2) -> coef [4][C] = s, t, mean, invstd
//   s = gamma*invstd, t = beta - mean*s  so that  BN(y) = s*y + t           (nn.BatchNorm2d, eps 1e-5, R.md:293)
// running_mean/var updated with momentum (unbiased variance), num_batches_tracked += 1   
// grid = C blocks (one channel each) of 256 threads.
// ---------------------------------------------------------------------------------------------------------------
__global__ EAE_NO_PK __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int ntiles, int C, float count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* running_mean, float* running_var, long long* nbt,
                                                           float momentum, float eps, float* __restrict__ coef) {
  __shared__ double red[512];
  const int tid = threadIdx.x;
  const int ch = blockIdx.x;
  double a, b;
  reduce_partials_ch(part, ntiles, C, ch, red, a, b);
  if (tid == 0) {
    double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0.0) var = 0.0;
    float invstd = 1.0f / sqrtf((float)var + eps);
    float s = gamma[ch] * invstd;
    coef[ch] = s;
    coef[C + ch] = beta[ch] - (float)mean * s;
    coef[2 * C + ch] = (float)mean;
    coef[3 * C + ch] = invstd;
    if (running_mean) {
      double unb = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mean;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
    }
  }
  if (nbt && blockIdx.x == 0 && tid == 0) *nbt += 1;
}

// eval mode: coefficients from the running statistics
struct BnEvalCoefArgs { int C; const float* gamma; const float* beta; const float* rm; const float* rv; float eps; float* coef; };
__device__ __forceinline__ EAE_NO_PK void bn_eval_coef_body(const BnEvalCoefArgs& a) {
  const int C = a.C;
  int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  float invstd = 1.0f / sqrtf(a.rv[ch] + a.eps);
  float s = a.gamma[ch] * invstd;
  a.coef[ch] = s;
  a.coef[C + ch] = a.beta[ch] - a.rm[ch] * s;
  a.coef[2 * C + ch] = a.rm[ch];
  a.coef[3 * C + ch] = invstd;
}
__global__ EAE_NO_PK void bn_eval_coef_kernel(BnEvalCoefArgs a) { bn_eval_coef_body(a); }
__global__ EAE_NO_PK void bn_eval_coef_kernel_g(GroupPack<BnEvalCoefArgs> p, int gz) { bn_eval_coef_body(group_args<BnEvalCoefArgs>(gz)); }

// BatchNorm backward finalize.  part = [2][C][ntiles] (sum g, sum g*xhat), g = ReLU-masked upstream gradient.
//   dbeta = sum g ; dgamma = sum g*xhat ;  dy = A*g + B*y + Cc  with
//   A = gamma*invstd, B = -A*invstd*dgamma/N, Cc = -A*dbeta/N + A*invstd*mean*dgamma/N   (native_batch_norm_backward)
__global__ EAE_NO_PK __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int ntiles, int C, float count,
                                                               const float* __restrict__ gamma, const float* __restrict__ coef_fwd,
                                                               float* dgamma, float* dbeta, float* __restrict__ coef_bwd) {
  __shared__ double red[512];
  const int tid = threadIdx.x;
  const int ch = blockIdx.x;
  double a, b;
  reduce_partials_ch(part, ntiles, C, ch, red, a, b);
  if (tid == 0) {
    float db = (float)a, dg = (float)b;
    if (dbeta) dbeta[ch] = db;
    if (dgamma) dgamma[ch] = dg;
    float mean = coef_fwd[2 * C + ch], invstd = coef_fwd[3 * C + ch];
    float A = gamma[ch] * invstd;
    float Bc = -A * invstd * dg / count;
    coef_bwd[ch] = A;
    coef_bwd[C + ch] = Bc;
    coef_bwd[2 * C + ch] = -A * db / count - Bc * mean;
  }
}

// SyncBN form of the backward finalize, split at the point where the replicas exchange their sums (dp.py):
//   reduce: part -> sums[2][C] in fp64 (this replica's sum g, sum g*xhat) and its LOCAL dgamma / dbeta (the gradient exchange sums them)
//   coef  : A, B, C from the GLOBAL sums (after the all-reduce) and the global element count
__global__ EAE_NO_PK __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ part, int ntiles, int C, double* __restrict__ sums,
                                                             float* dgamma, float* dbeta) {
  __shared__ double red[512];
  const int ch = blockIdx.x;
  double a, b;
  reduce_partials_ch(part, ntiles, C, ch, red, a, b);
  if (threadIdx.x == 0) {
    sums[ch] = a; sums[C + ch] = b;
    if (dbeta) dbeta[ch] = (float)a;
    if (dgamma) dgamma[ch] = (float)b;
  }
}
__global__ EAE_NO_PK void bn_bwd_coef_kernel(const double* __restrict__ sums, int C, float count, const float* __restrict__ gamma,
                                             const float* __restrict__ coef_fwd, float* __restrict__ coef_bwd) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  const float db = (float)sums[ch], dg = (float)sums[C + ch];
  const float mean = coef_fwd[2 * C + ch], invstd = coef_fwd[3 * C + ch];
  const float A = gamma[ch] * invstd;
  const float Bc = -A * invstd * dg / count;
  coef_bwd[ch] = A;
  coef_bwd[C + ch] = Bc;
  coef_bwd[2 * C + ch] = -A * db / count - Bc * mean;
}
int eae_launch_bn_bwd_reduce(hipStream_t st, const float* part, int ntiles, int C, double* sums, float* dgamma, float* dbeta) {
  EAE_NO_GROUP("bn_bwd_reduce_kernel");
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C), dim3(256), 0, st, part, ntiles, C, sums, dgamma, dbeta);
  EAE_LAUNCH_CHECK();
  return 0;
}
int eae_launch_bn_bwd_coef(hipStream_t st, const double* sums, int C, long long count, const float* gamma, const float* coef_fwd,
                           float* coef_bwd) {
  EAE_NO_GROUP("bn_bwd_coef_kernel");
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, st, sums, C, (float)count, gamma, coef_fwd, coef_bwd);
  EAE_LAUNCH_CHECK();
  return 0;
}

int eae_launch_bn_finalize(hipStream_t st, const float* part, int ntiles, int C, long long count, const float* gamma,
                           const float* beta, float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef) {
  EAE_NO_GROUP("bn_finalize_kernel");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, st, part, ntiles, C, (float)count, gamma, beta, rm,
                     rv, nbt, momentum, eps, coef);
  EAE_LAUNCH_CHECK();
  return 0;
}
int eae_launch_bn_eval_coef(hipStream_t st, int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                            float eps, float* coef) {
  const BnEvalCoefArgs ba = {C, gamma, beta, rm, rv, eps, coef};
  eae_launch(bn_eval_coef_kernel, bn_eval_coef_kernel_g, dim3((C + 63) / 64), dim3(64), 0, st, ba);
  EAE_LAUNCH_CHECK();
  return 0;
}
int eae_launch_bn_bwd_finalize(hipStream_t st, const float* part, int ntiles, int C, long long count, const float* gamma,
                               const float* coef_fwd, float* dgamma, float* dbeta, float* coef_bwd) {
  EAE_NO_GROUP("bn_bwd_finalize_kernel");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, part, ntiles, C, (float)count, gamma,
                     coef_fwd, dgamma, dbeta, coef_bwd);
  EAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Device-side stream dependencies.  gate_kernel: one lane polls up to 4 progress words (relaxed agent-scope loads, s_sleep
// between polls) until each has reached its value; the kernels behind it on the same stream then see everything the signalling
// stream had completed when it published the value (kernel-boundary acquire).  Values only grow (wrap-safe signed compare).
// The spin is bounded by WALL-CLOCK time (s_memrealtime, 100 MHz; default 30 s, EAE_GATE_TIMEOUT_MS): a caller's stream that is
// stalled for seconds in front of the step (a late data-parallel peer inside the previous step's all-reduce, a long copy) is waited
// for.  A gate that does give up records the value it was waiting for in the STICKY word `timeout`: the work behind it then ran too
// early, so every later optimizer kernel of the context refuses to update the parameters and writes NaN losses
// (adam_kernel `bad`), the steppers raise (eae_gate_timeouts()).  signal_kernel publishes a value from a stream that has no other
// kernel to carry it.
// ---------------------------------------------------------------------------------------------------------------
__global__ EAE_NO_PK __launch_bounds__(64) void gate_kernel(GateArgs g) {
  if (threadIdx.x == 0) gate_wait(g);
}
__global__ EAE_NO_PK __launch_bounds__(64) void gate_kernel_g(GroupPack<GateArgs> p, int gz) {      // grouped twins (eae_group.h): one wave per member
  if (threadIdx.x == 0) gate_wait(group_args<GateArgs>(gz));
}
struct SignalArgs { unsigned* word; unsigned val; };
__global__ EAE_NO_PK void signal_kernel(SignalArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(a.word, a.val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ EAE_NO_PK void signal_kernel_g(GroupPack<SignalArgs> p, int gz) {
  const SignalArgs a = group_args<SignalArgs>(gz);
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(a.word, a.val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int eae_launch_gate(hipStream_t st, const GateArgs& g) {
  eae_launch(gate_kernel, gate_kernel_g, dim3(1), dim3(64), 0, st, g);
  EAE_LAUNCH_CHECK();
  return 0;
}
int eae_launch_signal(hipStream_t st, unsigned* word, unsigned val) {
  const SignalArgs sa = {word, val};
  eae_launch(signal_kernel, signal_kernel_g, dim3(1), dim3(64), 0, st, sa);
  EAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight packing: one launch converts every fp32 master weight into the bf16 layouts the kernels read.
// ---------------------------------------------------------------------------------------------------------------
// (32-bit unsigned index arithmetic: destination and source element counts stay below 2^32, and a 64-bit division costs ~5x a
// 32-bit one on this target -- the all-64-bit version of this kernel took 14 us per step)
__device__ __forceinline__ float pack_fetch(const PackDesc& d, const float* __restrict__ src, unsigned i) {
  const unsigned d0 = (unsigned)d.d0, d1 = (unsigned)d.d1, d2 = (unsigned)d.d2;
  switch (d.mode) {
    case PACK_3x3_P1: {   // dst [A][9][B]  <- src [A][B][3][3]
      unsigned b = i % d1, r = i / d1, tap = r % 9u, a = r / 9u;
      return src[(a * d1 + b) * 9u + tap];
    }
    case PACK_3x3_P2: {   // dst [B][9][A]  <- src [A][B][3][3]
      unsigned a = i % d0, r = i / d0, tap = r % 9u, b = r / 9u;
      return src[(a * d1 + b) * 9u + tap];
    }
    case PACK_K27: {      // dst [A][32] with k = tap*3 + c (zero for k >= 27)  <- src [A][3][3][3]
      unsigned k = i & 31u, a = i >> 5;
      if (k >= 27u) return 0.f;
      unsigned tap = k / 3u, c = k % 3u;
      return src[(a * 3u + c) * 9u + tap];
    }
    case PACK_K36: {      // dst [A][64] with k = tap*4 + c (zero for c == 3 and k >= 36)  <- src [A][3][3][3]   (edge_conv_kernel)
      unsigned k = i & 63u, a = i >> 6;
      if (k >= 36u || (k & 3u) == 3u) return 0.f;
      return src[(a * 3u + (k & 3u)) * 9u + (k >> 2)];
    }
    case PACK_DECONV4_JOINT: {   // dst [16][128]: n = phase*3+co, k = nb*32+ci  <- src [32 ci][3 co][3][3]
      int k = i & 127; int n = i >> 7;
      if (n >= 12) return 0.f;
      int ph = n / 3, co = n % 3, nb = k >> 5, ci = k & 31;
      int py = ph >> 1, px = ph & 1, dy = nb >> 1, dx = nb & 1;
      int ky = py == 0 ? (dy == 0 ? 1 : -1) : (dy == 0 ? 2 : 0);
      int kx = px == 0 ? (dx == 0 ? 1 : -1) : (dx == 0 ? 2 : 0);
      if (ky < 0 || kx < 0) return 0.f;
      return src[(ci * 3 + co) * 9 + ky * 3 + kx];
    }
    // FC modes: d0 = latent width PADDED to a multiple of 64, d.lv = the real width (rows / columns >= lv are zero; the source
    // tensor has lv of them)
    case PACK_FC_ROWMAJOR_KPERM: {   // dst [R][K'] with k' = p*Cc + c  <- src [R][K] with k = c*P + p   (d0=R, d1=Cc, d2=P)
      unsigned K = d1 * d2, k2 = i % K, r = i / K;
      if (r >= (unsigned)d.lv) return 0.f;
      unsigned c = k2 % d1, p = k2 / d1;
      return src[r * K + c * d2 + p];
    }
    case PACK_FC_TRANS_KPERM: {      // dst [K'][R]  <- src [R][K]   (transpose + permute)
      unsigned r = i % d0, k2 = i / d0, K = d1 * d2;
      if (r >= (unsigned)d.lv) return 0.f;
      unsigned c = k2 % d1, p = k2 / d1;
      return src[r * K + c * d2 + p];
    }
    case PACK_FC_ROWPERM: {          // dst [J'][L] with j' = p*Cc + c <- src [J][L] with j = c*P + p   (d0=L, d1=Cc, d2=P)
      unsigned l = i % d0, j2 = i / d0;
      if (l >= (unsigned)d.lv) return 0.f;
      unsigned c = j2 % d1, p = j2 / d1;
      return src[(c * d2 + p) * (unsigned)d.lv + l];
    }
    case PACK_FC_ROWPERM_TRANS: {    // dst [L][J']  <- src [J][L]
      unsigned J = d1 * d2, j2 = i % J, l = i / J;
      if (l >= (unsigned)d.lv) return 0.f;
      unsigned c = j2 % d1, p = j2 / d1;
      return src[(c * d2 + p) * (unsigned)d.lv + l];
    }
    case PACK_PAD_COLS: {            // dst [R][d0] <- src [R][lv], zero beyond column lv
      unsigned l = i % d0, r = i / d0;
      return l < (unsigned)d.lv ? src[r * (unsigned)d.lv + l] : 0.f;
    }
    default: return src[i];          // PACK_COPY
  }
}

struct PackAllArgs { const PackDesc* descs; const float* params; uint8_t* pack_base; Fp8State* q; unsigned* clear_word; const unsigned short* blkmap; int blk_begin; };
__device__ __forceinline__ EAE_NO_PK void pack_all_body(const PackAllArgs& pa) {
  const PackDesc* __restrict__ descs = pa.descs; const float* __restrict__ params = pa.params; uint8_t* __restrict__ pack_base = pa.pack_base;
  Fp8State* __restrict__ q = pa.q; unsigned* __restrict__ clear_word = pa.clear_word;
  // the workgroup's descriptor and its place among the descriptor's workgroups: 2-D launch (y = descriptor, the same number of
  // workgroups for each) or flattened (blkmap: a descriptor has as many workgroups as its work needs -- the 2-D form spent most of its
  // workgroups on the ~20 small descriptors while a handful looped over the four 262 K-element projections)
  const bool flat = pa.blkmap != nullptr;
  const unsigned gb = flat ? (unsigned)pa.blk_begin + blockIdx.x : 0u;
  const PackDesc d = descs[flat ? (unsigned)pa.blkmap[gb] : blockIdx.y];
  const unsigned bx = flat ? gb - (unsigned)d.blk0 : blockIdx.x, nbx = flat ? (unsigned)d.nblk : gridDim.x;
  if (clear_word != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *clear_word = 0u;
  const float* src = params + d.src_off;
  const unsigned count = (unsigned)d.count;
  if (d.q_layer >= 0) {        // e4m3 bytes = sat(w * s_w); also reports max |w| for the next step's scale (delayed scaling)
    const float sw = q->s_w[d.q_layer];
    float mx = 0.f;
    for (unsigned i = bx * 256u + threadIdx.x; i < count; i += nbx * 256u) {
      const float v = pack_fetch(d, src, i);
      mx = fmaxf(mx, fabsf(v));
      const float sv = fminf(fmaxf(v * sw, -448.f), 448.f);
      pack_base[d.dst_off + i] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(sv, 0.f, 0, false) & 0xff);
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&q->amax_w[d.q_layer][((bx * 4 + (threadIdx.x >> 6)) & (FP8_AMAX_SLOTS - 1)) * FP8_AMAX_STRIDE], __float_as_uint(mx));
    return;
  }
  if ((d.mode == PACK_3x3_P1 || d.mode == PACK_3x3_P2) && !d.out_f32) {
    // 3x3 weights W[a][b][3][3]: one thread per (a, b) reads its 9 consecutive floats and writes them one tap-row apart; the lanes of
    // a wave run over b for P1 = [a][9][b] and over a for P2 = [b][9][a], so every store instruction of a wave is contiguous and
    // every source line is read once (the element-per-thread mapping gathered one float per line and lane for P2)
    const unsigned A = (unsigned)d.d0, Bq = (unsigned)d.d1;
    const bool p1 = d.mode == PACK_3x3_P1;
    bf16_t* dst = reinterpret_cast<bf16_t*>(pack_base + d.dst_off);
    for (unsigned t = bx * 256u + threadIdx.x; t < A * Bq; t += nbx * 256u) {
      const unsigned a = p1 ? t / Bq : t % A, b = p1 ? t % Bq : t / A;
      const float* sp = src + ((size_t)a * Bq + b) * 9;
      float w[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) w[tap] = sp[tap];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        dst[p1 ? ((size_t)a * 9 + tap) * Bq + b : ((size_t)b * 9 + tap) * A + a] = (bf16_t)f2bf(w[tap]);
    }
    return;
  }
  if (d.mode == PACK_FC_ROWPERM_TRANS && !d.out_f32 && (unsigned)d.d2 * ((unsigned)d.d1 / 64) * ((unsigned)d.d0 / 64) >= 1024u) {
    // dec.fc's transposed layout dst[l][p*Cc + c] <- src[(c*P + p)][l]: 64 x 64 tiles through LDS (rows of the source are read whole,
    // rows of the destination are written in 128-byte pieces); element-per-thread gathers one float per source line and lane.
    // Only for the big shapes (>= 1024 tiles: 256x256 inputs): at 64x64 the 64 tiles are too few blocks (11.4 vs 8.9 us per step).
    // (round 4: 16-byte loads and 16-byte stores -- a thread converts 8 consecutive destination elements -- instead of one float per
    //  lane in and one bf16 per lane out: a 2-byte store costs ~12x a 16-byte one per byte, the kernel was store-instruction-bound)
    __shared__ float tile[64][65];
    const unsigned Lp = (unsigned)d.d0, Cc = (unsigned)d.d1, P = (unsigned)d.d2, lv = (unsigned)d.lv, J = Cc * P;
    const unsigned ncc = Cc / 64, nlc = Lp / 64, ntiles = P * ncc * nlc;
    const bool vec = (lv & 3u) == 0;
    bf16_t* dst = reinterpret_cast<bf16_t*>(pack_base + d.dst_off);
    for (unsigned tl = bx; tl < ntiles; tl += nbx) {
      const unsigned lc = tl % nlc, cch = (tl / nlc) % ncc, p = tl / (nlc * ncc);
      const unsigned c0 = cch * 64, l0 = lc * 64;
      __syncthreads();
#pragma unroll
      for (unsigned k = 0; k < 4; ++k) {         // 64 rows (ci) x 16 float4 (l)
        const unsigned idx = threadIdx.x + 256u * k, l4 = (idx & 15u) * 4u, ci = idx >> 4;
        const float* sp = src + ((size_t)(c0 + ci) * P + p) * lv + l0 + l4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec && l0 + l4 + 3 < lv) v = *reinterpret_cast<const float4*>(sp);
        else { if (l0 + l4 < lv) v.x = sp[0]; if (l0 + l4 + 1 < lv) v.y = sp[1]; if (l0 + l4 + 2 < lv) v.z = sp[2]; if (l0 + l4 + 3 < lv) v.w = sp[3]; }
        tile[ci][l4] = v.x; tile[ci][l4 + 1] = v.y; tile[ci][l4 + 2] = v.z; tile[ci][l4 + 3] = v.w;
      }
      __syncthreads();
#pragma unroll
      for (unsigned k = 0; k < 2; ++k) {         // 64 destination rows (l) x 8 pieces of 8 consecutive channels
        const unsigned idx = threadIdx.x + 256u * k, c8 = (idx & 7u) * 8u, l = idx >> 3;
        uint4 o;
        o.x = pack2(tile[c8][l], tile[c8 + 1][l]); o.y = pack2(tile[c8 + 2][l], tile[c8 + 3][l]);
        o.z = pack2(tile[c8 + 4][l], tile[c8 + 5][l]); o.w = pack2(tile[c8 + 6][l], tile[c8 + 7][l]);
        *reinterpret_cast<uint4*>(dst + (size_t)(l0 + l) * J + (size_t)p * Cc + c0 + c8) = o;
      }
    }
    return;
  }
  if ((d.mode == PACK_FC_ROWMAJOR_KPERM || d.mode == PACK_FC_TRANS_KPERM) && !d.out_f32 && (d.d2 & 63) == 0 && (d.d1 & 63) == 0 && (d.d0 & 63) == 0 &&
      (unsigned)d.d0 * ((unsigned)d.d1 / 64) * ((unsigned)d.d2 / 64) >= 1024u) {
    // enc.fc's two layouts on the big shapes (256x256 inputs: 16.7 M weights, P = 256 positions): 64 x 64 tiles through LDS.
    //   row-major form  dst[r][p*Cc + c] <- src[r][c*P + p]: for one r, a tile of 64 channels x 64 positions;
    //   transposed form dst[p*Cc + c][r] <- src[r][c*P + p]: for one c, a tile of 64 rows x 64 positions.
    // Either way the source is read in 256-byte runs along p and the destination written in 128-byte runs (the thread-per-(r, c)
    // mapping below touches 64 source lines per load instruction: 256 us of the 2.1 ms config-5 step went into this kernel).
    __shared__ float tile[64][65];
    const unsigned R = (unsigned)d.d0, Cc = (unsigned)d.d1, P = (unsigned)d.d2, K = Cc * P, lv = (unsigned)d.lv;
    const bool rowmajor = d.mode == PACK_FC_ROWMAJOR_KPERM;
    const unsigned npc = P / 64, nA = rowmajor ? Cc / 64 : R / 64, nfix = rowmajor ? R : Cc, ntiles = nfix * nA * npc;
    bf16_t* dst = reinterpret_cast<bf16_t*>(pack_base + d.dst_off);
    for (unsigned tl = bx; tl < ntiles; tl += nbx) {
      const unsigned pc = tl % npc, ac = (tl / npc) % nA, fix = tl / (npc * nA);
      const unsigned p0 = pc * 64, a0 = ac * 64;                 // a = channel (row-major form) or row (transposed form)
      __syncthreads();
#pragma unroll
      for (unsigned k = 0; k < 4; ++k) {         // 64 rows (a) x 16 float4 along p (16-byte loads: see the transposed dec.fc form above)
        const unsigned idx = threadIdx.x + 256u * k, p4 = (idx & 15u) * 4u, ai = idx >> 4;
        const unsigned r = rowmajor ? fix : a0 + ai, c = rowmajor ? a0 + ai : fix;
        const float4 v = r < lv ? *reinterpret_cast<const float4*>(src + (size_t)r * K + (size_t)c * P + p0 + p4) : make_float4(0.f, 0.f, 0.f, 0.f);
        tile[ai][p4] = v.x; tile[ai][p4 + 1] = v.y; tile[ai][p4 + 2] = v.z; tile[ai][p4 + 3] = v.w;
      }
      __syncthreads();
#pragma unroll
      for (unsigned k = 0; k < 2; ++k) {         // 64 destination rows (p) x 8 pieces of 8 consecutive a
        const unsigned idx = threadIdx.x + 256u * k, a8 = (idx & 7u) * 8u, pp = idx >> 3;
        uint4 o;
        o.x = pack2(tile[a8][pp], tile[a8 + 1][pp]); o.y = pack2(tile[a8 + 2][pp], tile[a8 + 3][pp]);
        o.z = pack2(tile[a8 + 4][pp], tile[a8 + 5][pp]); o.w = pack2(tile[a8 + 6][pp], tile[a8 + 7][pp]);
        const size_t oo = rowmajor ? (size_t)fix * K + (size_t)(p0 + pp) * Cc + a0 + a8
                                   : ((size_t)(p0 + pp) * Cc + fix) * R + a0 + a8;
        *reinterpret_cast<uint4*>(dst + oo) = o;
      }
    }
    return;
  }
  if ((d.mode == PACK_FC_ROWMAJOR_KPERM || d.mode == PACK_FC_TRANS_KPERM) && (d.d2 & 3) == 0 && !d.out_f32) {
    // enc.fc's two layouts (k' = p*Cc + c <- k = c*P + p).  One thread per (row r, channel c) reads its P consecutive source floats
    // (whole cache lines, each read once) and writes them P rows apart; the lanes of a wave run over c (row-major form) or over r
    // (transposed form), so every store instruction of a wave covers 128 contiguous bytes.  The element-per-thread mapping of
    // pack_fetch reads one float per 64-byte line and lane: 16x the L2 traffic for these two packs.
    const unsigned R = (unsigned)d.d0, Cc = (unsigned)d.d1, P = (unsigned)d.d2, K = Cc * P, lv = (unsigned)d.lv;
    const bool rowmajor = d.mode == PACK_FC_ROWMAJOR_KPERM;
    bf16_t* dst = reinterpret_cast<bf16_t*>(pack_base + d.dst_off);
    for (unsigned t = bx * 256u + threadIdx.x; t < R * Cc; t += nbx * 256u) {
      const unsigned r = rowmajor ? t / Cc : t % R, c = rowmajor ? t % Cc : t / R;
      const float* sp = src + (size_t)r * K + (size_t)c * P;
      for (unsigned p = 0; p < P; p += 4) {
        const float4 v = r < lv ? *reinterpret_cast<const float4*>(sp + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (unsigned q = 0; q < 4; ++q) {
          const size_t o = rowmajor ? (size_t)r * K + (size_t)(p + q) * Cc + c : ((size_t)(p + q) * Cc + c) * R + r;
          dst[o] = (bf16_t)f2bf(e[q]);
        }
      }
    }
    return;
  }
  if (d.mode == PACK_FC_ROWPERM && !d.out_f32 && (d.lv & 7) == 0 && (d.d0 & 7) == 0) {
    // dec.fc's row-permuted layout dst[p*Cc + c][l] <- src[c*P + p][l]: whole rows move, so a thread takes 8 consecutive l (two 16-byte
    // loads, one 16-byte store) instead of one float in and one bf16 out (16.7 M elements at 256x256 inputs)
    const unsigned Lp = (unsigned)d.d0, Cc = (unsigned)d.d1, P = (unsigned)d.d2, lv = (unsigned)d.lv, n8 = count / 8u, l8n = Lp / 8u;
    bf16_t* dst = reinterpret_cast<bf16_t*>(pack_base + d.dst_off);
    for (unsigned i = bx * 256u + threadIdx.x; i < n8; i += nbx * 256u) {
      const unsigned l = (i % l8n) * 8u, j2 = i / l8n, cc = j2 % Cc, pp = j2 / Cc;
      uint4 o = make_uint4(0, 0, 0, 0);
      if (l < lv) {
        const float4* sp = reinterpret_cast<const float4*>(src + (size_t)(cc * P + pp) * lv + l);
        const float4 a = sp[0], b = sp[1];
        o.x = pack2(a.x, a.y); o.y = pack2(a.z, a.w); o.z = pack2(b.x, b.y); o.w = pack2(b.z, b.w);
      }
      *reinterpret_cast<uint4*>(dst + (size_t)i * 8u) = o;
    }
    return;
  }
  for (unsigned i = bx * 256u + threadIdx.x; i < count; i += nbx * 256u) {
    float v = pack_fetch(d, src, i);
    if (d.out_f32) reinterpret_cast<float*>(pack_base + d.dst_off)[i] = v;
    else reinterpret_cast<bf16_t*>(pack_base + d.dst_off)[i] = (bf16_t)f2bf(v);
  }
}
__global__ EAE_NO_PK __launch_bounds__(256) void pack_all_kernel(PackAllArgs a) { pack_all_body(a); }
__global__ EAE_NO_PK __launch_bounds__(256) void pack_all_kernel_g(GroupPack<PackAllArgs> p, int gz) { pack_all_body(group_args<PackAllArgs>(gz)); }

int eae_launch_pack_all(hipStream_t st, const PackDesc* descs_dev, int ndesc, const float* params, void* pack_base, Fp8State* q, unsigned* clear_word,
                        int blocks_per_desc) {
  const PackAllArgs pa = {descs_dev, params, (uint8_t*)pack_base, q, clear_word, nullptr, 0};
  if (blocks_per_desc <= 0) blocks_per_desc = 256;
  // member of a grouped step: the launch carries eae_geo_mult members' descriptors (8 x 256 x ~20 workgroups, most of them with a few
  // hundred elements, took 54 us: the dispatch of 41 K workgroups, not the 62 MB they move)
  if (eae_geo_mult > 1) blocks_per_desc = blocks_per_desc / eae_geo_mult > 16 ? blocks_per_desc / eae_geo_mult : 16;
  eae_launch(pack_all_kernel, pack_all_kernel_g, dim3(blocks_per_desc, ndesc), dim3(256), 0, st, pa);
  EAE_LAUNCH_CHECK();
  return 0;
}

// Workgroups of a descriptor in the flattened launch: one pass of its loop per workgroup (2048 destination elements, one 64 x 64
// tile, 256 (a, b) pairs of a 3 x 3 weight, 256 (row, channel) pairs of a projection), at most 1024.
static int pack_desc_blocks(const PackDesc& d) {
  long long items;
  const bool big_rpt = d.mode == PACK_FC_ROWPERM_TRANS && !d.out_f32 && (long long)d.d2 * (d.d1 / 64) * (d.d0 / 64) >= 1024;
  const bool big_kp = (d.mode == PACK_FC_ROWMAJOR_KPERM || d.mode == PACK_FC_TRANS_KPERM) && !d.out_f32 && (d.d2 & 63) == 0 && (d.d1 & 63) == 0 &&
                      (d.d0 & 63) == 0 && (long long)d.d0 * (d.d1 / 64) * (d.d2 / 64) >= 1024;
  if (d.q_layer >= 0) items = (d.count + 2047) / 2048;
  else if ((d.mode == PACK_3x3_P1 || d.mode == PACK_3x3_P2) && !d.out_f32) items = ((long long)d.d0 * d.d1 + 255) / 256;
  else if (big_rpt) items = (long long)d.d2 * (d.d1 / 64) * (d.d0 / 64);
  else if (big_kp) items = (long long)d.d0 * (d.d1 / 64) * (d.d2 / 64);
  else if ((d.mode == PACK_FC_ROWMAJOR_KPERM || d.mode == PACK_FC_TRANS_KPERM) && (d.d2 & 3) == 0 && !d.out_f32) items = ((long long)d.d0 * d.d1 + 255) / 256;
  else items = (d.count + 2047) / 2048;
  if (items < 1) items = 1;
  if (items > 1024) items = 1024;
  return (int)items;
}
int eae_pack_assign_blocks(PackDesc* descs, int ndesc, unsigned short* blkmap, int cap) {
  int tot = 0;
  for (int i = 0; i < ndesc; ++i) {
    descs[i].blk0 = tot; descs[i].nblk = pack_desc_blocks(descs[i]);
    for (int b = 0; b < descs[i].nblk; ++b) { if (tot + b >= cap) return -1; blkmap[tot + b] = (unsigned short)i; }
    tot += descs[i].nblk;
  }
  return tot;
}
int eae_launch_pack_flat(hipStream_t st, const PackDesc* descs_all, const unsigned short* blkmap, int blk_begin, int blk_count, const float* params,
                         void* pack_base, Fp8State* q, unsigned* clear_word) {
  if (blk_count <= 0) return 0;
  const PackAllArgs pa = {descs_all, params, (uint8_t*)pack_base, q, clear_word, blkmap, blk_begin};
  eae_launch(pack_all_kernel, pack_all_kernel_g, dim3(blk_count), dim3(256), 0, st, pa);
  EAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// fp8 variant: delayed scaling.  Once per step (after every kernel that reads the current scales has finished) the scales of
// the NEXT step are derived from the maxima this step's kernels reported:  s = 2^floor(log2(MAX / (2 * amax)))  (a factor 2 of
// head room because the operand of the next step is not the one measured), MAX = 448 (e4m3: activations, weights) or 57344
// (e5m2: gradients).  A tensor whose maximum is 0 or not finite keeps its scale.  The maxima are cleared for the next step.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fp8_scale_from(unsigned amax_bits, float maxv, float keep) {
  const float amax = __uint_as_float(amax_bits);
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return keep;
  const float e = floorf(log2f(maxv / (2.f * amax)));
  return exp2f(fminf(fmaxf(e, -60.f), 60.f));
}
__global__ EAE_NO_PK void fp8_scales_kernel(Fp8State* q) {
  static_assert(FP8_AMAX_SLOTS % 64 == 0, "whole rounds of one slot per lane");
  const int lane = threadIdx.x;
  unsigned ma[6], mg[6], mw[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    ma[i] = mg[i] = mw[i] = 0u;
#pragma unroll
    for (int s0 = 0; s0 < FP8_AMAX_SLOTS; s0 += 64) {
      const int w_ = (s0 + lane) * FP8_AMAX_STRIDE;
      const unsigned a_ = q->amax_act[i][w_], g_ = q->amax_grad[i][w_], x_ = q->amax_w[i][w_];
      ma[i] = a_ > ma[i] ? a_ : ma[i]; mg[i] = g_ > mg[i] ? g_ : mg[i]; mw[i] = x_ > mw[i] ? x_ : mw[i];
      q->amax_act[i][w_] = 0; q->amax_grad[i][w_] = 0; q->amax_w[i][w_] = 0;
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {       // (non-negative float bits order like unsigned integers)
      const unsigned a = __shfl_xor(ma[i], sh), g = __shfl_xor(mg[i], sh), w = __shfl_xor(mw[i], sh);
      ma[i] = a > ma[i] ? a : ma[i]; mg[i] = g > mg[i] ? g : mg[i]; mw[i] = w > mw[i] ? w : mw[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (lane != i) continue;
    q->s_act[i] = fp8_scale_from(ma[i], 448.f, q->s_act[i]);
    q->s_grad[i] = fp8_scale_from(mg[i], 57344.f, q->s_grad[i]);
    q->s_w[i] = fp8_scale_from(mw[i], 448.f, q->s_w[i]);
    const float sa = q->s_act[i], sg = q->s_grad[i], sw = q->s_w[i];
    q->qs_fwd[i][0] = 1.f / sa; q->qs_fwd[i][1] = 1.f / (sa * sw);
    q->qs_bwd[i][0] = 1.f / sg; q->qs_bwd[i][1] = 1.f / (sg * sw);
    // weight gradient: conv layers (i < 3): small operand = output gradient, big = input activation; transposed layers: the reverse
    const float ss = i < 3 ? sg : sa, sb = i < 3 ? sa : sg;
    q->qs_wg[i][0] = 1.f / ss; q->qs_wg[i][1] = 1.f / sb; q->qs_wg[i][2] = 1.f / (ss * sb); q->qs_wg[i][3] = 0.f;
  }
}
int eae_launch_fp8_scales(hipStream_t st, Fp8State* q) {
  EAE_NO_GROUP("fp8_scales_kernel");
  hipLaunchKernelGGL(fp8_scales_kernel, dim3(1), dim3(64), 0, st, q);
  EAE_LAUNCH_CHECK();
  return 0;
}
void eae_fp8_state_init(Fp8State* h) {
  *h = Fp8State();
  for (int i = 0; i < 6; ++i) {
    h->s_act[i] = h->s_grad[i] = h->s_w[i] = 1.f;
    h->qs_fwd[i][0] = h->qs_fwd[i][1] = h->qs_bwd[i][0] = h->qs_bwd[i][1] = 1.f;
    h->qs_wg[i][0] = h->qs_wg[i][1] = h->qs_wg[i][2] = 1.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused multi-tensor Adam over the flat fp32 arenas (torch.optim.Adam defaults, R.md:624; L2 decay for the MLP, R.md:2625)
//   g += wd*p ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------------------------
// One element of the update with every rounding step written out (no fp contraction left to the compiler): the eager step's
// adam_kernel and the graph path's adam_dyn_kernel must agree bit for bit (tests/test_gpu_ae.py::test_graph_replay_equals_eager),
// and a refactoring of either kernel must not move a rounding -- round 4 found the compiler fusing  b2*v + ((1-b2)*g)*g  as
// fma(g, t, b2*v) in one of them and as fma(b2, v, g*t) in the other after the first was turned into an inlined body.
__device__ __forceinline__ EAE_NO_PK void adam_update(float& p, float gj, float& m, float& v, float b1, float b2, float step_size, float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
  m = __builtin_fmaf(1.f - b1, gj - m, m);                   // exp_avg.lerp_(grad, 1-beta1)
  const float t = (1.f - b2) * gj;
  v = __builtin_fmaf(b2, v, gj * t);                         // mul_(beta2).addcmul_(grad, grad, 1-beta2)
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = __builtin_fmaf(-step_size, m / denom, p);
}
struct AdamArgs {
  float* p; const float* g; float* m; float* v; long n4; float b1, b2, step_size, bc2_sqrt, eps, wd, gscale; uint4* zbuf; long zn16;
  const unsigned* bad; const unsigned* bad2; float* nan_out; int nan_fill;
};
__device__ __forceinline__ EAE_NO_PK void adam_body(const AdamArgs& aa) {
  float* __restrict__ p = aa.p; const float* __restrict__ g = aa.g; float* __restrict__ m = aa.m; float* __restrict__ v = aa.v;
  const long n4 = aa.n4; const float b1 = aa.b1, b2 = aa.b2, step_size = aa.step_size, bc2_sqrt = aa.bc2_sqrt, eps = aa.eps, wd = aa.wd, gscale = aa.gscale;
  uint4* __restrict__ zbuf = aa.zbuf; const long zn16 = aa.zn16;
  const unsigned* __restrict__ bad = aa.bad; const unsigned* __restrict__ bad2 = aa.bad2; float* __restrict__ nan_out = aa.nan_out;
  const int nan_fill = aa.nan_fill;
  // side job: clear the BatchNorm statistics accumulators for the next step (every consumer of this step has finished)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zn16; i += (long)gridDim.x * 256) zbuf[i] = make_uint4(0, 0, 0, 0);
  // a side-stream gate of this context has timed out at some point (sticky word): gradients may have been computed from stale
  // activations -- no update, and the step's loss scalars become NaN so that the run cannot go on unnoticed
  const bool stale = bad != nullptr && __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  const bool diverged = bad2 != nullptr && __hip_atomic_load(bad2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  if (stale || diverged) {
    if (nan_out != nullptr && blockIdx.x == 0 && threadIdx.x < 3) nan_out[threadIdx.x] = __builtin_nanf("");
    // nan_fill (eae_config.nan_exact / EAE_NAN_EXACT=1): a DIVERGED step does to the replica what it does to the reference's model --
    // loss.backward() on a non-finite loss hands every parameter a NaN gradient and Adam writes NaN into the parameter and both moments
    // (R.md:653-654); the default leaves the last finite parameters in place.  Never for a stale step.
    if (nan_fill && diverged && !stale) {
      const float4 nn = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        reinterpret_cast<float4*>(p)[i] = nn; reinterpret_cast<float4*>(m)[i] = nn; reinterpret_cast<float4*>(v)[i] = nn;
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float wp = wd * P[j];
      adam_update(P[j], __builtin_fmaf(G[j], gscale, wp), M[j], V[j], b1, b2, step_size, bc2_sqrt, eps);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}
__global__ EAE_NO_PK __launch_bounds__(256) void adam_kernel(AdamArgs a) { adam_body(a); }
__global__ EAE_NO_PK __launch_bounds__(256) void adam_kernel_g(GroupPack<AdamArgs> p, int gz) { adam_body(group_args<AdamArgs>(gz)); }

// Same update with the per-step scalars (lr/bias_correction1, sqrt(bias_correction2), weight decay) read from device memory,
// so that a captured hipGraph of the whole train step can be replayed while the step count advances.
__global__ EAE_NO_PK __launch_bounds__(256) void adam_dyn_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n4, float b1, float b2, float eps,
                                                        const float* __restrict__ dyn, const unsigned* __restrict__ bad,
                                                        const unsigned* __restrict__ bad2, float* __restrict__ nan_out) {
  if ((bad != nullptr && __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ||
      (bad2 != nullptr && __hip_atomic_load(bad2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {     // see adam_kernel
    if (nan_out != nullptr && blockIdx.x == 0 && threadIdx.x < 3) nan_out[threadIdx.x] = __builtin_nanf("");
    return;
  }
  const float step_size = dyn[0], bc2_sqrt = dyn[1], wd = dyn[2];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      adam_update(P[j], __builtin_fmaf(wd, P[j], G[j]), M[j], V[j], b1, b2, step_size, bc2_sqrt, eps);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}
__global__ EAE_NO_PK void set_dyn_kernel(float* dyn, float a, float b, float c) { dyn[0] = a; dyn[1] = b; dyn[2] = c; }

int eae_launch_adam_dyn(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double b1, double b2, double eps,
                        const float* dyn, const unsigned* bad, const unsigned* bad2, float* nan_out) {
  if (n % 4) return eae_set_error(-2, "adam: arena length must be a multiple of 4");
  long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  EAE_NO_GROUP("adam_dyn_kernel");
  hipLaunchKernelGGL(adam_dyn_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n4, (float)b1, (float)b2, (float)eps, dyn, bad, bad2, nan_out);
  EAE_LAUNCH_CHECK();
  return 0;
}
int eae_launch_set_dyn(hipStream_t st, float* dyn, double lr, double b1, double b2, double wd, long long step) {
  double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  EAE_NO_GROUP("set_dyn_kernel");
  hipLaunchKernelGGL(set_dyn_kernel, dim3(1), dim3(1), 0, st, dyn, (float)(lr / bc1), (float)sqrt(bc2), (float)wd);
  EAE_LAUNCH_CHECK();
  return 0;
}

int eae_launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2,
                    double eps, double wd, long long step) {
  return eae_launch_adam_scaled(st, p, g, m, v, n, lr, b1, b2, eps, wd, step, 1.0f);
}

int eae_launch_adam_scaled(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2,
                           double eps, double wd, long long step, float gscale, void* zero_buf, long long zero_bytes,
                           const unsigned* bad, const unsigned* bad2, float* nan_out, int nan_fill, int max_blocks) {
  if (n % 4) return eae_set_error(-2, "adam: arena length must be a multiple of 4");
  double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (max_blocks <= 0) max_blocks = 2048;
  if (blocks > max_blocks) blocks = max_blocks;
  if (eae_geo_mult > 1) blocks = blocks / eae_geo_mult > 64 ? blocks / eae_geo_mult : (blocks < 64 ? blocks : 64);      // member of a grouped step
  const AdamArgs aa = {p, g, m, v, n4, (float)b1, (float)b2, (float)(lr / bc1), (float)sqrt(bc2), (float)eps, (float)wd, gscale,
                       (uint4*)zero_buf, (long)(zero_bytes / 16), bad, bad2, nan_out, nan_fill};
  eae_launch(adam_kernel, adam_kernel_g, dim3(blocks), dim3(256), 0, st, aa);
  EAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// On-device input staging: the reference's training transform (R.md:225-230)
//   RandomHorizontalFlip -> RandomCrop(64, padding=4) -> ToTensor -> AddGaussianNoise(0, 0.03)   (eval: ToTensor only, R.md:232-234)
// fused into one kernel: uint8 HWC [B,H,W,3] -> fp32 NCHW [B,3,H,W] (what conv1 reads), so a batch crosses PCIe as 12 KB/img.
//   out[n,c,y,x] = img[n, y+top-4, flip ? W-1-(x+left-4) : x+left-4, c] / 255 (0 outside) + std * N(0,1)
// Per-image (flip, top, left) and the noise come from a counter-based Philox4x32-10 stream keyed by (seed, step) unless
// explicit `params` [B][3] / `noise` [B,3,H,W] are supplied (parity tests).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* o) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ EAE_NO_PK __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                       int train, float std, unsigned long long seed, unsigned long long step,
                                                       const int* __restrict__ params, const float* __restrict__ noise) {
  const long plane = (long)H * W;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= plane * B) return;
  const int n = (int)(p / plane), r = (int)(p % plane), y = r / W, x = r % W;
  int flip = 0, top = 4, left = 4;
  if (train) {
    if (params) { flip = params[n * 3]; top = params[n * 3 + 1]; left = params[n * 3 + 2]; }
    else {
      uint32_t o[4];
      philox4((uint32_t)n, 0x5eedu, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
      flip = o[0] >> 31; top = (int)(((unsigned long long)o[1] * 9) >> 32); left = (int)(((unsigned long long)o[2] * 9) >> 32);
    }
  }
  const int sy = y + top - 4;
  int sx = x + left - 4;
  const bool inside = sy >= 0 && sy < H && sx >= 0 && sx < W;
  if (flip) sx = W - 1 - sx;
  float v[3] = {0.f, 0.f, 0.f};
  if (inside) {
    const uint8_t* q = in + (((long)n * H + sy) * W + sx) * 3;
    v[0] = q[0] / 255.0f; v[1] = q[1] / 255.0f; v[2] = q[2] / 255.0f;   // ToTensor: .div(255), exact
  }
  if (train && std != 0.f) {
    if (noise) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = fmaf(std, noise[((long)n * 3 + c) * plane + r], v[c]);
    } else {
      uint32_t o[4];
      philox4((uint32_t)p, (uint32_t)(p >> 32) ^ 0xA5A5u, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, ~(uint32_t)(seed >> 32), o);
      // Box-Muller: two uniforms -> two normals (three needed: a second pair from o[2], o[3])
      float u1 = ((o[0] >> 8) + 1) * (1.0f / 16777216.0f), u2 = (o[1] >> 8) * (1.0f / 16777216.0f);
      float u3 = ((o[2] >> 8) + 1) * (1.0f / 16777216.0f), u4 = (o[3] >> 8) * (1.0f / 16777216.0f);
      float r1 = sqrtf(-2.0f * __logf(u1)), r2 = sqrtf(-2.0f * __logf(u3));
      v[0] = fmaf(std, r1 * __cosf(6.28318530718f * u2), v[0]);
      v[1] = fmaf(std, r1 * __sinf(6.28318530718f * u2), v[1]);
      v[2] = fmaf(std, r2 * __cosf(6.28318530718f * u4), v[2]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[((long)n * 3 + c) * plane + r] = v[c];
}

int eae_launch_augment(hipStream_t st, const void* in_u8, float* out, int B, int H, int W, int train, float std, unsigned long long seed,
                       unsigned long long step, const int* params, const float* noise) {
  if (!in_u8 || !out || B <= 0 || H <= 0 || W <= 0) return eae_set_error(-2, "augment: bad argument");
  const long tot = (long)B * H * W;
  EAE_NO_GROUP("augment_kernel");
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const uint8_t*)in_u8, out, B, H, W, train, std,
                     seed, step, params, noise);
  EAE_LAUNCH_CHECK();
  return 0;
}
