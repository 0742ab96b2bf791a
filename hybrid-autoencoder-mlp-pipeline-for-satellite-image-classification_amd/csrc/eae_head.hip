// Classification head of SupervisedAutoencoder (R.md:423-427): Linear(L,128) -> ReLU -> Linear(128,C), fused with
// CrossEntropyLoss (R.md:623, 650) and the whole backward of the head, in fp32 (9.5 K MAC per image: pure latency).
// One block handles HR = 8 batch rows: forward, softmax/CE, dlogits, dh, dz and the per-block partial weight gradients,
// which are laid out exactly like the four head tensors in the parameter arena so one reduce_slices call finishes them.
#include "eae_internal.h"
#include "eae_common.hip.h"
#include "eae_head.h"

constexpr int HR = 8;   // batch rows per block
constexpr int HC = 64;  // logits row stride in LDS = largest class count

__global__ EAE_NO_PK __launch_bounds__(256) void head_kernel(HeadArgs a) {
  extern __shared__ float sm[];
  const int L = a.L, C = a.C, LS = L + 1;
  float* w1 = sm;                    // [128][L+1]
  float* zt = w1 + 128 * LS;         // [HR][L]
  float* w2 = zt + HR * L;           // [C][128]
  float* hp = w2 + C * 128;          // [32][129]  pre-activation
  float* dh = hp + HR * 129;         // [HR][129]
  float* lg = dh + HR * 129;         // [HR][HC]   logits, then dlogits
  float* b1 = lg + HR * HC;          // [128]
  float* b2 = b1 + 128;              // [HC]
  float* rl = b2 + HC;               // [HR] per-row loss, [HR] per-row correct
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * HR;
  for (int i = tid; i < 128 * L; i += 256) w1[(i / L) * LS + (i % L)] = a.w1[i];
  for (int i = tid; i < HR * L; i += 256) {
    int r = r0 + i / L;
    zt[i] = r < a.B ? a.z[(size_t)r * L + (i % L)] : 0.f;
  }
  for (int i = tid; i < C * 128; i += 256) w2[i] = a.w2[i];
  if (tid < 128) b1[tid] = a.b1[tid];
  if (tid < C) b2[tid] = a.b2[tid];
  __syncthreads();
  {   // h_pre[r][j]
    const int j = tid & 127, rh = tid >> 7;
    constexpr int RH = HR / 2;
    float accv[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) accv[r] = b1[j];
    for (int k = 0; k < L; ++k) {
      float w = w1[j * LS + k];
#pragma unroll
      for (int r = 0; r < RH; ++r) accv[r] = fmaf(zt[(rh * RH + r) * L + k], w, accv[r]);
    }
#pragma unroll
    for (int r = 0; r < RH; ++r) hp[(rh * RH + r) * 129 + j] = accv[r];
  }
  __syncthreads();
  for (int i = tid; i < HR * C; i += 256) {   // logits
    int r = i / C, c = i % C;
    float s = b2[c];
    for (int j = 0; j < 128; ++j) s = fmaf(fmaxf(hp[r * 129 + j], 0.f), w2[c * 128 + j], s);
    lg[r * HC + c] = s;
  }
  __syncthreads();
  if (tid < HR) {   // softmax + CE per row
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
    rl[tid] = loss; rl[HR + tid] = correct;
  }
  __syncthreads();
  if (tid == 0 && a.loss_part) {
    float s = 0.f, cr = 0.f;
    for (int r = 0; r < HR; ++r) { s += rl[r]; cr += rl[HR + r]; }
    a.loss_part[blockIdx.x * 2] = s; a.loss_part[blockIdx.x * 2 + 1] = cr;
  }
  if ((!a.labels && !a.dlogits_in) || !a.grad_part) return;
  {   // dh[r][j] = (h_pre > 0) * sum_c dlogits[r][c] * W2[c][j]
    const int j = tid & 127, rh = tid >> 7;
    for (int r = rh * (HR / 2); r < (rh + 1) * (HR / 2); ++r) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) s = fmaf(lg[r * HC + c], w2[c * 128 + j], s);
      dh[r * 129 + j] = hp[r * 129 + j] > 0.f ? s : 0.f;
    }
  }
  __syncthreads();
  // dz_cls[r][k] = sum_j dh[r][j] * W1[j][k]
  for (int i = tid; i < HR * L; i += 256) {
    int r = i / L, k = i % L;
    float s = 0.f;
    for (int j = 0; j < 128; ++j) s = fmaf(dh[r * 129 + j], w1[j * LS + k], s);
    if (r0 + r < a.B) a.dz[(size_t)(r0 + r) * L + k] = s;
  }
  // partial weight gradients of this block, arena order: W1 [128][L], b1 [128], W2 [C][128], b2 [C] (+pad)
  float* gp = a.grad_part + (size_t)blockIdx.x * a.grad_stride;
  for (int i = tid; i < 128 * L; i += 256) {
    int j = i / L, k = i % L;
    float s = 0.f;
    for (int r = 0; r < HR; ++r) s = fmaf(dh[r * 129 + j], zt[r * L + k], s);
    gp[i] = s;
  }
  if (tid < 128) {
    float s = 0.f;
    for (int r = 0; r < HR; ++r) s += dh[r * 129 + tid];
    gp[128 * L + tid] = s;
  }
  for (int i = tid; i < C * 128; i += 256) {
    int c = i / 128, j = i % 128;
    float s = 0.f;
    for (int r = 0; r < HR; ++r) s = fmaf(lg[r * HC + c], fmaxf(hp[r * 129 + j], 0.f), s);
    gp[128 * L + 128 + i] = s;
  }
  if (tid < ((C + 3) & ~3)) {
    float s = 0.f;
    if (tid < C) for (int r = 0; r < HR; ++r) s += lg[r * HC + tid];
    gp[128 * L + 128 + C * 128 + tid] = s;
  }
}

int eae_launch_head(hipStream_t st, const HeadArgs& a) {
  if (a.L > 256 || a.C > HC || a.L % 4) return eae_set_error(-2, "head: latent_dim must be <= 256 (multiple of 4), classes <= 64");
  size_t smem = sizeof(float) * ((size_t)128 * (a.L + 1) + HR * a.L + a.C * 128 + 2 * HR * 129 + HR * HC + 128 + HC + 2 * HR);
  static size_t attr = 0;
  if (smem > attr) {
    EAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = smem;
  }
  hipLaunchKernelGGL(head_kernel, dim3((a.B + HR - 1) / HR), dim3(256), smem, st, a);
  EAE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Loss bookkeeping on the device (the reference syncs with loss.item() every step, R.md:656-657; here the
// sample-weighted sums of R.md:656-660 are accumulated on the device and read back once per epoch).
//   accum[0] += loss*B, accum[1] += mse*B, accum[2] += ce*B, accum[3] += B, accum[4] += correct
//   last[0..2] = loss, mse, ce of this step
// ---------------------------------------------------------------------------------------------------------------
__global__ EAE_NO_PK __launch_bounds__(256) void loss_finalize_kernel(const float* mse_part, int n_mse, const float* ce_part, int n_ce, float alpha,
                                                             float inv_numel, float B, float* db4, float* accum, float* last) {
  __shared__ double red[256][6];
  const int tid = threadIdx.x;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < n_mse; i += 256) {
    float4 v = reinterpret_cast<const float4*>(mse_part)[i];
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
  }
  for (int i = tid; i < n_ce; i += 256) { s[4] += ce_part[i * 2]; s[5] += ce_part[i * 2 + 1]; }
  for (int k = 0; k < 6; ++k) red[tid][k] = s[k];
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) for (int k = 0; k < 6; ++k) red[tid][k] += red[tid + o][k];
    __syncthreads();
  }
  if (tid == 0) {
    float mse = (float)(red[0][0] * inv_numel);
    float cem = n_ce ? (float)(red[0][4] / B) : 0.f;
    float loss = alpha * mse + cem;
    if (db4) { db4[0] = (float)red[0][1]; db4[1] = (float)red[0][2]; db4[2] = (float)red[0][3]; }
    if (accum) { accum[0] += loss * B; accum[1] += mse * B; accum[2] += cem * B; accum[3] += B; accum[4] += (float)red[0][5]; }
    if (last) { last[0] = loss; last[1] = mse; last[2] = cem; }
  }
}

int eae_launch_loss_finalize(hipStream_t st, const float* mse_part, int n_mse, const float* ce_part, int n_ce, float alpha,
                             double numel, int B, float* db4, float* accum, float* last) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, mse_part, n_mse, ce_part, n_ce, alpha,
                     (float)(1.0 / numel), (float)B, db4, accum, last);
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
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(64), 0, st, ce_part, n, (float)B, out2);
  EAE_LAUNCH_CHECK();
  return 0;
}

int eae_head_blocks(int B) { return (B + HR - 1) / HR; }
