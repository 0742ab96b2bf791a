// Classification head of SupervisedAutoencoder (R.md:423-427): Linear(L,128) -> ReLU -> Linear(128,C), fused with
// CrossEntropyLoss (R.md:623, 650) and the whole backward of the head, in fp32 (9.5 K MAC per image: pure latency).
// One block handles 16 (8 for wide latents) batch rows: forward, softmax/CE, dlogits, dh, dz and the per-block partial weight gradients,
// which are laid out exactly like the four head tensors in the parameter arena so one reduce_slices call finishes them.
#include "eae_internal.h"
#include "eae_common.hip.h"
#include "eae_head.h"
#include <utility>
#include <vector>

constexpr int HC = 64;  // logits row stride in LDS = largest class count

// Round 3: the round-2 kernel (8 rows per block, every operand read from LDS one float at a time, W1 staged with scalar loads and
// an integer division per element) took 38-40 us for 19 K MAC per image.  Now: HR_ = 16 rows per block (8 when the latent is wider
// than 128: W1 alone is then 130 KB of LDS), 16-byte global and LDS accesses, and every product register-blocked so that an LDS
// float4 feeds 8-32 FMAs:
//   h_pre [HR][128] = z W1^T        thread = 2 rows x 4 columns (j, j+32, j+64, j+96), K in steps of 4
//   dz    [HR][L]   = dh W1         thread = 1 row x 4 consecutive k (strided over [HR][L/4]), j = 0..127
//   dW1   [128][L]  = dh^T z        thread = 4 j x 4 k tiles (strided over [32][L/4]), r = 0..HR-1
// Row strides are padded to L + 4 / 132 floats: 16-byte aligned rows whose float4 reads by 16 consecutive lanes fall on 16 different
// bank slots.  Arithmetic stays fp32 fmaf chains (the head is fp32 end to end: <= 5e-5 vs the reference's goldens).
template <int HR_>
__device__ __forceinline__ EAE_NO_PK void head_body(const HeadArgs& a) {     // <= 96 VGPRs: the kernel runs beside the decoder on a side stream, and a wave-specialised igemm2 workgroup (2 x 200 registers per SIMD lane) must still fit beside one of its waves -- at 132 registers dec.deconv1's forward waited for the head's 32 CUs (15 -> 28 us)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int L = a.L, C = a.C, LS = L + 4, HS = 132;
  float* w1 = sm;                    // [128][LS]
  float* zt = w1 + 128 * LS;         // [HR_][LS]
  float* w2 = zt + HR_ * LS;         // [C][HS]
  float* hp = w2 + ((C * HS + 3) & ~3);   // [HR_][HS]  pre-activation
  float* dh = hp + HR_ * HS;         // [HR_][HS]
  float* lg = dh + HR_ * HS;         // [HR_][HC]   logits, then dlogits
  float* b1 = lg + HR_ * HC;         // [128]
  float* b2 = b1 + 128;              // [HC]
  float* rl = b2 + HC;               // [HR_] per-row loss, [HR_] per-row correct
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * HR_;
  const int L4 = L >> 2;
  for (int i = tid; i < 128 * L4; i += 256) {
    const int j = i / L4, k4 = i - j * L4;
    *reinterpret_cast<float4*>(w1 + j * LS + k4 * 4) = *reinterpret_cast<const float4*>(a.w1 + (size_t)j * L + k4 * 4);
  }
  for (int i = tid; i < HR_ * L4; i += 256) {
    const int r = i / L4, k4 = i - r * L4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < a.B) v = *reinterpret_cast<const float4*>(a.z + (size_t)(r0 + r) * L + k4 * 4);
    *reinterpret_cast<float4*>(zt + r * LS + k4 * 4) = v;
  }
  for (int i = tid; i < C * 32; i += 256) {
    const int c = i >> 5, j4 = i & 31;
    *reinterpret_cast<float4*>(w2 + c * HS + j4 * 4) = *reinterpret_cast<const float4*>(a.w2 + (size_t)c * 128 + j4 * 4);
  }
  if (tid < 128) b1[tid] = a.b1[tid];
  if (tid < C) b2[tid] = a.b2[tid];
  __syncthreads();
  {   // h_pre[r][j]: 2 rows x 4 columns per thread
    constexpr int RT = HR_ / 2;                 // row pairs
    const int ct = tid & 31, rt = tid >> 5;     // 8 row groups x 32 column groups
    if (rt < RT) {
      float acc[2][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[0][q] = b1[ct + 32 * q]; acc[1][q] = acc[0][q]; }
#pragma unroll 2
      for (int k = 0; k < L; k += 4) {
        const float4 z0 = *reinterpret_cast<const float4*>(zt + (2 * rt) * LS + k), z1 = *reinterpret_cast<const float4*>(zt + (2 * rt + 1) * LS + k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 w = *reinterpret_cast<const float4*>(w1 + (ct + 32 * q) * LS + k);
          acc[0][q] = fmaf(z0.x, w.x, acc[0][q]); acc[0][q] = fmaf(z0.y, w.y, acc[0][q]);
          acc[0][q] = fmaf(z0.z, w.z, acc[0][q]); acc[0][q] = fmaf(z0.w, w.w, acc[0][q]);
          acc[1][q] = fmaf(z1.x, w.x, acc[1][q]); acc[1][q] = fmaf(z1.y, w.y, acc[1][q]);
          acc[1][q] = fmaf(z1.z, w.z, acc[1][q]); acc[1][q] = fmaf(z1.w, w.w, acc[1][q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { hp[(2 * rt) * HS + ct + 32 * q] = acc[0][q]; hp[(2 * rt + 1) * HS + ct + 32 * q] = acc[1][q]; }
    }
  }
  __syncthreads();
  for (int i = tid; i < HR_ * C; i += 256) {   // logits
    const int r = i / C, c = i - r * C;
    float s = b2[c];
#pragma unroll 4
    for (int j = 0; j < 128; j += 4) {
      const float4 h = *reinterpret_cast<const float4*>(hp + r * HS + j), w = *reinterpret_cast<const float4*>(w2 + c * HS + j);
      s = fmaf(relu_nan(h.x), w.x, s); s = fmaf(relu_nan(h.y), w.y, s); s = fmaf(relu_nan(h.z), w.z, s); s = fmaf(relu_nan(h.w), w.w, s);
    }
    lg[r * HC + c] = s;
  }
  __syncthreads();
  if (tid < HR_) {   // softmax + CE per row
    int r = r0 + tid;
    float loss = 0.f, correct = 0.f;
    if (r < a.B) {
      float mx = lg[tid * HC];
      int am = 0;
      for (int c = 1; c < C; ++c) if (lg[tid * HC + c] > mx) { mx = lg[tid * HC + c]; am = c; }
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(lg[tid * HC + c] - mx);
      float lse = logf(se) + mx;
      if (a.logits) for (int c = 0; c < C; ++c) a.logits[(size_t)r * C + c] = lg[tid * HC + c];
      if (a.dlogits_in) {
        for (int c = 0; c < C; ++c) lg[tid * HC + c] = a.dlogits_in[(size_t)r * C + c];
      } else if (a.labels) {
        int lab = (int)a.labels[r];
        loss = lse - lg[tid * HC + lab];
        correct = (am == lab) ? 1.f : 0.f;
        for (int c = 0; c < C; ++c) {
          float pr = expf(lg[tid * HC + c] - lse);
          lg[tid * HC + c] = (pr - (c == lab ? 1.f : 0.f)) * a.inv_batch;
        }
      }
    } else {
      for (int c = 0; c < C; ++c) lg[tid * HC + c] = 0.f;
    }
    rl[tid] = loss; rl[HR_ + tid] = correct;
  }
  __syncthreads();
  if (tid == 0 && a.loss_part) {
    float s = 0.f, cr = 0.f;
    for (int r = 0; r < HR_; ++r) { s += rl[r]; cr += rl[HR_ + r]; }
    a.loss_part[blockIdx.x * 2] = s; a.loss_part[blockIdx.x * 2 + 1] = cr;
  }
  if ((!a.labels && !a.dlogits_in) || !a.grad_part) return;
  // dh[r][j] = (h_pre > 0) * sum_c dlogits[r][c] * W2[c][j]
  for (int i = tid; i < HR_ * 128; i += 256) {
    const int r = i >> 7, j = i & 127;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) s = fmaf(lg[r * HC + c], w2[c * HS + j], s);
    dh[r * HS + j] = hp[r * HS + j] > 0.f ? s : 0.f;
  }
  __syncthreads();
  // dz_cls[r][k..k+3] = sum_j dh[r][j] * W1[j][k..k+3]
  for (int i = tid; i < HR_ * L4; i += 256) {
    const int r = i / L4, k4 = i - r * L4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < 128; j += 4) {        // 5 LDS reads in flight per 16 FMAs (a one-read-per-iteration loop exposes the LDS latency 128 times)
      const float4 d0 = *reinterpret_cast<const float4*>(dh + r * HS + j);
      const float d[4] = {d0.x, d0.y, d0.z, d0.w};
      float4 w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const float4*>(w1 + (j + q) * LS + k4 * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s.x = fmaf(d[q], w[q].x, s.x); s.y = fmaf(d[q], w[q].y, s.y); s.z = fmaf(d[q], w[q].z, s.z); s.w = fmaf(d[q], w[q].w, s.w);
      }
    }
    if (r0 + r < a.B) *reinterpret_cast<float4*>(a.dz + (size_t)(r0 + r) * L + k4 * 4) = s;
  }
  // partial weight gradients of this block, arena order: W1 [128][L], b1 [128], W2 [C][128], b2 [C] (+pad)
  float* gp = a.grad_part + (size_t)blockIdx.x * a.grad_stride;
  for (int i = tid; i < 32 * L4; i += 256) {            // dW1: tiles of 4 j x 4 k
    const int jt = i / L4, k4 = i - jt * L4;
    float4 s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = 0; r < HR_; ++r) {
      const float4 zv = *reinterpret_cast<const float4*>(zt + r * LS + k4 * 4);
      const float4 d0 = *reinterpret_cast<const float4*>(dh + r * HS + jt * 4);
      const float d[4] = {d0.x, d0.y, d0.z, d0.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s[q].x = fmaf(d[q], zv.x, s[q].x); s[q].y = fmaf(d[q], zv.y, s[q].y); s[q].z = fmaf(d[q], zv.z, s[q].z); s[q].w = fmaf(d[q], zv.w, s[q].w);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(gp + (size_t)(jt * 4 + q) * L + k4 * 4) = s[q];
  }
  if (tid < 128) {
    float s = 0.f;
    for (int r = 0; r < HR_; ++r) s += dh[r * HS + tid];
    gp[128 * L + tid] = s;
  }
  for (int i = tid; i < C * 128; i += 256) {
    int c = i >> 7, j = i & 127;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < HR_; ++r) s = fmaf(lg[r * HC + c], relu_nan(hp[r * HS + j]), s);
    gp[128 * L + 128 + i] = s;
  }
  if (tid < ((C + 3) & ~3)) {
    float s = 0.f;
    if (tid < C) for (int r = 0; r < HR_; ++r) s += lg[r * HC + tid];
    gp[128 * L + 128 + C * 128 + tid] = s;
  }
}
template <int HR_>
__global__ EAE_NO_PK __launch_bounds__(256, 5) void head_kernel(HeadArgs a) { head_body<HR_>(a); }
template <int HR_>
__global__ EAE_NO_PK __launch_bounds__(256, 5) void head_kernel_g(GroupPack<HeadArgs> p, int gz) { head_body<HR_>(group_args<HeadArgs>(gz)); }

static int head_rows(int L) { return L <= 128 ? 16 : 8; }
int eae_head_blocks(int B, int L) { const int hr = head_rows(L); return (B + hr - 1) / hr; }

int eae_launch_head(hipStream_t st, const HeadArgs& a) {
  if (a.L > 256 || a.C > HC || a.L % 4) return eae_set_error(-2, "head: latent_dim must be <= 256 (multiple of 4), classes <= 64");
  const int hr = head_rows(a.L);
  const size_t smem = sizeof(float) * ((size_t)128 * (a.L + 4) + (size_t)hr * (a.L + 4) + ((a.C * 132 + 3) & ~3) + 2 * hr * 132 + hr * HC + 128 + HC + 2 * hr);
  // the dynamic-LDS limit is an attribute of the kernel ON ONE DEVICE and the request grows with the latent width: keep the
  // largest size granted per (kernel, device) (a process-wide static skipped the attribute for an engine on a second device)
  void (*kern)(HeadArgs) = hr == 16 ? head_kernel<16> : head_kernel<8>;
  void (*kern_g)(GroupPack<HeadArgs>, int) = hr == 16 ? head_kernel_g<16> : head_kernel_g<8>;
  {
    static thread_local std::vector<std::pair<std::pair<const void*, int>, size_t>> granted;
    int dev = 0;
    EAE_HIP(hipGetDevice(&dev));
    const void* f = eae_rec ? reinterpret_cast<const void*>(kern_g) : reinterpret_cast<const void*>(kern);
    size_t* have = nullptr;
    for (auto& g : granted) if (g.first.first == f && g.first.second == dev) have = &g.second;
    if (!have || *have < smem) {
      EAE_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      if (have) *have = smem; else granted.push_back({{f, dev}, smem});
    }
  }
  eae_launch(kern, kern_g, dim3((a.B + hr - 1) / hr), dim3(256), (unsigned)smem, st, a);
  EAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Loss bookkeeping on the device (the reference syncs with loss.item() every step, R.md:656-657; here the
// sample-weighted sums of R.md:656-660 are accumulated on the device and read back once per epoch).
//   accum[0] += loss*B, accum[1] += mse*B, accum[2] += ce*B, accum[3] += B, accum[4] += correct
//   last[0..2] = loss, mse, ce of this step
// ---------------------------------------------------------------------------------------------------------------
// fixed-order reduction: xor butterfly inside each wave (every lane ends with the wave's sum), the four wave sums added in wave order
__device__ __forceinline__ double lf_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
struct LossFinArgs { const float* mse_part; int n_mse; const float* ce_part; int n_ce; float alpha, inv_numel, B; float* db4; float* accum; float* last; const unsigned* poison; };
__device__ __forceinline__ EAE_NO_PK void loss_finalize_body(const LossFinArgs& a) {
  const float* mse_part = a.mse_part; const int n_mse = a.n_mse; const float* ce_part = a.ce_part; const int n_ce = a.n_ce;
  const float alpha = a.alpha, inv_numel = a.inv_numel, B = a.B;
  float* db4 = a.db4; float* accum = a.accum; float* last = a.last; const unsigned* poison = a.poison;
  __shared__ double red[4][6];
  const int tid = threadIdx.x;
  double s[6] = {0, 0, 0, 0, 0, 0};
  // 8 loads in flight per thread: with one load per loop iteration every iteration exposed a full memory round trip (16 of them at
  // B=512: that, not the reduction, was this kernel's 16 us)
  // (ONE 1024-thread block with every row requested in the first round was measured too: 29 us in situ instead of 11 -- a 16-wave
  //  workgroup waits for a CU with that many free wave slots beside the register-filling weight-gradient workgroups)
  for (int i0 = tid; i0 < n_mse; i0 += 256 * 8) {
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * 256;
      v[q] = i < n_mse ? reinterpret_cast<const float4*>(mse_part)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { s[0] += v[q].x; s[1] += v[q].y; s[2] += v[q].z; s[3] += v[q].w; }
  }
  for (int i = tid; i < n_ce; i += 256) { s[4] += ce_part[i * 2]; s[5] += ce_part[i * 2 + 1]; }
#pragma unroll
  for (int k = 0; k < 6; ++k) {       // (round 2: a 9-step LDS tree with 8 barriers over fp64[256][6], 16-18 us inside the step)
    const double w = lf_wave_sum(s[k]);
    if ((tid & 63) == 0) red[tid >> 6][k] = w;
  }
  __syncthreads();
  if (tid == 0) {
    double r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    float mse = (float)(r[0] * inv_numel);
    float cem = n_ce ? (float)(r[4] / B) : 0.f;
    float loss = alpha * mse + cem;
    // a BatchNorm layer of this step saw non-finite statistics (bn_fold_fwd_finish): the losses read NaN like the reference's
    if (poison != nullptr && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) loss = mse = cem = __builtin_nanf("");
    if (db4) { db4[0] = (float)r[1]; db4[1] = (float)r[2]; db4[2] = (float)r[3]; }
    if (accum) { accum[0] += loss * B; accum[1] += mse * B; accum[2] += cem * B; accum[3] += B; accum[4] += (float)r[5]; }
    if (last) { last[0] = loss; last[1] = mse; last[2] = cem; }
  }
}
__global__ EAE_NO_PK __launch_bounds__(256) void loss_finalize_kernel(LossFinArgs a) { loss_finalize_body(a); }
__global__ EAE_NO_PK __launch_bounds__(256) void loss_finalize_kernel_g(GroupPack<LossFinArgs> p, int gz) { loss_finalize_body(group_args<LossFinArgs>(gz)); }

int eae_launch_loss_finalize(hipStream_t st, const float* mse_part, int n_mse, const float* ce_part, int n_ce, float alpha,
                             double numel, int B, float* db4, float* accum, float* last, const unsigned* poison) {
  const LossFinArgs la = {mse_part, n_mse, ce_part, n_ce, alpha, (float)(1.0 / numel), (float)B, db4, accum, last, poison};
  eae_launch(loss_finalize_kernel, loss_finalize_kernel_g, dim3(1), dim3(256), 0, st, la);
  EAE_LAUNCH_CHECK();
  return 0;
}

// op-level helper: out2[0] = mean CE over the batch, out2[1] = number of correct argmax (fixed-order sum of the block partials)
__global__ EAE_NO_PK void ce_mean_kernel(const float* ce_part, int n, float B, float* out2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0, c = 0.0;
  for (int i = 0; i < n; ++i) { s += ce_part[2 * i]; c += ce_part[2 * i + 1]; }
  out2[0] = (float)(s / B); out2[1] = (float)c;
}
int eae_launch_ce_mean(hipStream_t st, const float* ce_part, int n, int B, float* out2) {
  EAE_NO_GROUP("ce_mean");
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(64), 0, st, ce_part, n, (float)B, out2);
  EAE_LAUNCH_CHECK();
  return 0;
}

