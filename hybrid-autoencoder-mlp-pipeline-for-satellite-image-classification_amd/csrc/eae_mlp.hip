// External latent-space classifier of the reference (R.md:2549-2566):
//   Linear(in,128) -> BatchNorm1d(128) -> ReLU -> Dropout(0.3) -> Linear(128,64) -> BatchNorm1d(64) -> ReLU -> Linear(64,C)
// and its training / evaluation steps (R.md:2639-2668).  17 K MAC per sample: everything is launch latency, so one
// training iteration (zero_grad, forward, CrossEntropy, backward, Adam with L2 weight decay, accuracy bookkeeping) is ONE
// kernel launch executed by a single 1024-thread workgroup (BatchNorm1d needs whole-batch statistics anyway); weights
// stay L1/L2-resident, activations live in a small global workspace, arithmetic is fp32 (matches the reference's dtype).
// Evaluation (running statistics, rows independent) uses one block per 64 rows.
#include "eae_internal.h"
#include "eae_common.hip.h"
#include <cmath>

namespace {
constexpr int H1 = 128, H2 = 64, T = 1024;
constexpr float BN_EPS = 1e-5f, BN_MOM = 0.1f;

struct MlpArgs {
  const float* x; const long long* labels;
  int B, IN, C;
  float *P, *G, *M, *V;
  long long off[11];
  float* bnrun;            // rm1[128] rv1[128] rm2[64] rv2[64]
  long long* nbt;          // [2]
  float *h1, *a1, *h2, *a2, *dlog, *g2, *g1;   // workspace
  int train, backward, adam;
  float step_size, bc2_sqrt, b1, b2, eps, wd;
  unsigned long long seed, step;
  const float* drop_mask;
  const float* dlog_in;    // externally supplied dL/dlogits [B][C] (autograd path) or nullptr
  int update_running;      // 0: do not touch running statistics / num_batches_tracked (recompute pass of the autograd path)
  float p_drop;
  float* logits; float* stats;    // stats: += loss*B, += B, += correct
};

// Philox4x32-10 (counter-based): keep-mask of nn.Dropout, keyed by (seed, optimisation step), counter = element index
__device__ __forceinline__ uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
__device__ float philox_uniform(unsigned long long seed, unsigned long long step, uint32_t idx) {
  uint32_t c0 = idx, c1 = (uint32_t)step, c2 = (uint32_t)(step >> 32), c3 = 0x9E3779B9u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = mulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = mulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// per-column batch statistics of v[rows][W] (two-pass), 1024 threads = W columns x (1024/W) row lanes
__device__ void col_stats(const float* v, int rows, int W, float* s_mean, float* s_var, float* red) {
  const int tid = threadIdx.x, lanes = T / W, col = tid % W, rl = tid / W;
  float s = 0.f;
  for (int r = rl; r < rows; r += lanes) s += v[r * W + col];
  red[rl * W + col] = s;
  __syncthreads();
  if (tid < W) { float a = 0.f; for (int i = 0; i < lanes; ++i) a += red[i * W + tid]; s_mean[tid] = a / rows; }
  __syncthreads();
  const float m = s_mean[col];
  s = 0.f;
  for (int r = rl; r < rows; r += lanes) { float d = v[r * W + col] - m; s = fmaf(d, d, s); }
  red[rl * W + col] = s;
  __syncthreads();
  if (tid < W) { float a = 0.f; for (int i = 0; i < lanes; ++i) a += red[i * W + tid]; s_var[tid] = a / rows; }
  __syncthreads();
}

// column sums of g[rows][W] and of g*xhat with xhat = (h-mean)*invstd
__device__ void col_sums2(const float* g, const float* h, const float* s_mean, const float* s_inv, int rows, int W, float* o1,
                          float* o2, float* red) {
  const int tid = threadIdx.x, lanes = T / W, col = tid % W, rl = tid / W;
  float a = 0.f, b = 0.f;
  for (int r = rl; r < rows; r += lanes) {
    float gv = g[r * W + col];
    a += gv;
    b = fmaf(gv, (h[r * W + col] - s_mean[col]) * s_inv[col], b);
  }
  red[rl * W + col] = a; red[T + rl * W + col] = b;
  __syncthreads();
  if (tid < W) {
    float x = 0.f, y = 0.f;
    for (int i = 0; i < lanes; ++i) { x += red[i * W + tid]; y += red[T + i * W + tid]; }
    o1[tid] = x; o2[tid] = y;
  }
  __syncthreads();
}

__global__ EAE_NO_PK __launch_bounds__(T) void mlp_kernel(MlpArgs a) {
  __shared__ float red[2 * T];
  __shared__ float mean1[H1], var1[H1], inv1[H1], mean2[H2], var2[H2], inv2[H2], c1[H1], c2[H1];
  const int tid = threadIdx.x;
  const int IN = a.IN, C = a.C;
  // rows handled by this block (training: the whole batch in block 0; eval: 64 rows per block)
  const int r0 = a.train ? 0 : blockIdx.x * 64;
  const int nb = a.train ? a.B : min(64, a.B - r0);
  const float* x = a.x + (size_t)r0 * IN;
  const float *W1 = a.P + a.off[0], *b1 = a.P + a.off[1], *g1w = a.P + a.off[2], *be1 = a.P + a.off[3];
  const float *W2 = a.P + a.off[4], *b2 = a.P + a.off[5], *g2w = a.P + a.off[6], *be2 = a.P + a.off[7];
  const float *W3 = a.P + a.off[8], *b3 = a.P + a.off[9];
  float *h1 = a.h1 + (size_t)r0 * H1, *a1 = a.a1 + (size_t)r0 * H1, *h2 = a.h2 + (size_t)r0 * H2, *a2 = a.a2 + (size_t)r0 * H2;
  float* dlog = a.dlog + (size_t)r0 * 16;
  // ---- layer 1
  for (int i = tid; i < nb * H1; i += T) {
    int b = i / H1, j = i % H1;
    float s = b1[j];
    for (int k = 0; k < IN; ++k) s = fmaf(x[b * IN + k], W1[j * IN + k], s);
    h1[i] = s;
  }
  __syncthreads();
  if (a.train) {
    col_stats(h1, nb, H1, mean1, var1, red);
    if (tid < H1) {
      inv1[tid] = 1.0f / sqrtf(var1[tid] + BN_EPS);
      float unb = nb > 1 ? var1[tid] * nb / (nb - 1) : var1[tid];
      if (a.update_running) {
        a.bnrun[tid] = (1.f - BN_MOM) * a.bnrun[tid] + BN_MOM * mean1[tid];
        a.bnrun[H1 + tid] = (1.f - BN_MOM) * a.bnrun[H1 + tid] + BN_MOM * unb;
      }
    }
    if (tid == 0 && a.nbt && a.update_running) { a.nbt[0] += 1; a.nbt[1] += 1; }
  } else if (tid < H1) {
    mean1[tid] = a.bnrun[tid];
    inv1[tid] = 1.0f / sqrtf(a.bnrun[H1 + tid] + BN_EPS);
  }
  __syncthreads();
  const float keep_scale = 1.0f / (1.0f - a.p_drop);
  for (int i = tid; i < nb * H1; i += T) {
    int j = i % H1;
    float o = fmaf(g1w[j], (h1[i] - mean1[j]) * inv1[j], be1[j]);
    float v = fmaxf(o, 0.f);
    if (a.train && a.p_drop > 0.f) {
      float keep;
      if (a.drop_mask) keep = a.drop_mask[(size_t)r0 * H1 + i];
      else keep = philox_uniform(a.seed, a.step, (uint32_t)i) >= a.p_drop ? 1.f : 0.f;
      v = v * keep * keep_scale;
    }
    a1[i] = v;
  }
  __syncthreads();
  // ---- layer 2
  for (int i = tid; i < nb * H2; i += T) {
    int b = i / H2, j = i % H2;
    float s = b2[j];
    for (int k = 0; k < H1; ++k) s = fmaf(a1[b * H1 + k], W2[j * H1 + k], s);
    h2[i] = s;
  }
  __syncthreads();
  if (a.train) {
    col_stats(h2, nb, H2, mean2, var2, red);
    if (tid < H2) {
      inv2[tid] = 1.0f / sqrtf(var2[tid] + BN_EPS);
      float unb = nb > 1 ? var2[tid] * nb / (nb - 1) : var2[tid];
      if (a.update_running) {
        a.bnrun[2 * H1 + tid] = (1.f - BN_MOM) * a.bnrun[2 * H1 + tid] + BN_MOM * mean2[tid];
        a.bnrun[2 * H1 + H2 + tid] = (1.f - BN_MOM) * a.bnrun[2 * H1 + H2 + tid] + BN_MOM * unb;
      }
    }
  } else if (tid < H2) {
    mean2[tid] = a.bnrun[2 * H1 + tid];
    inv2[tid] = 1.0f / sqrtf(a.bnrun[2 * H1 + H2 + tid] + BN_EPS);
  }
  __syncthreads();
  for (int i = tid; i < nb * H2; i += T) {
    int j = i % H2;
    a2[i] = fmaxf(fmaf(g2w[j], (h2[i] - mean2[j]) * inv2[j], be2[j]), 0.f);
  }
  __syncthreads();
  // ---- layer 3 + softmax / CE
  for (int i = tid; i < nb * C; i += T) {
    int b = i / C, c = i % C;
    float s = b3[c];
    for (int k = 0; k < H2; ++k) s = fmaf(a2[b * H2 + k], W3[c * H2 + k], s);
    dlog[b * 16 + c] = s;
    if (a.logits) a.logits[(size_t)(r0 + b) * C + c] = s;
  }
  __syncthreads();
  if (a.dlog_in) {
    for (int i = tid; i < nb * C; i += T) dlog[(i / C) * 16 + (i % C)] = a.dlog_in[(size_t)r0 * C + i];
    __syncthreads();
  }
  if (!a.labels && !a.dlog_in) return;
  float loss = 0.f, corr = 0.f;
  if (!a.dlog_in)
  for (int b = tid; b < nb; b += T) {
    float* l = dlog + b * 16;
    float mx = l[0]; int am = 0;
    for (int c = 1; c < C; ++c) if (l[c] > mx) { mx = l[c]; am = c; }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(l[c] - mx);
    float lse = logf(se) + mx;
    int lab = (int)a.labels[r0 + b];
    loss += lse - l[lab];
    corr += (am == lab) ? 1.f : 0.f;
    for (int c = 0; c < C; ++c) l[c] = (expf(l[c] - lse) - (c == lab ? 1.f : 0.f)) / (float)a.B;
  }
  red[tid] = loss; red[T + tid] = corr;
  __syncthreads();
  if (tid == 0 && a.stats) {
    float s = 0.f, cr = 0.f;
    const int lim = nb < T ? nb : T;
    for (int i = 0; i < lim; ++i) { s += red[i]; cr += red[T + i]; }
    if (a.train) { a.stats[0] += s; a.stats[1] += (float)nb; a.stats[2] += cr; }   // sum_b CE_b = mean CE * B
    else { atomicAdd(&a.stats[0], s); atomicAdd(&a.stats[1], (float)nb); atomicAdd(&a.stats[2], cr); }
  }
  if (!a.backward) return;
  // =========================================================================================== backward (train only)
  float *G = a.G;
  float *g2 = a.g2, *g1 = a.g1;
  // layer 3: dW3, db3, da2 -> do2
  for (int i = tid; i < C * H2; i += T) {
    int c = i / H2, k = i % H2;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s = fmaf(dlog[b * 16 + c], a2[b * H2 + k], s);
    G[a.off[8] + i] = s;
  }
  if (tid < C) { float s = 0.f; for (int b = 0; b < nb; ++b) s += dlog[b * 16 + tid]; G[a.off[9] + tid] = s; }
  for (int i = tid; i < nb * H2; i += T) {
    int b = i / H2, k = i % H2;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(dlog[b * 16 + c], W3[c * H2 + k], s);
    g2[i] = a2[i] > 0.f ? s : 0.f;
  }
  __syncthreads();
  // BN2 backward
  col_sums2(g2, h2, mean2, inv2, nb, H2, c1, c2, red);     // c1 = dbeta, c2 = dgamma
  if (tid < H2) { G[a.off[7] + tid] = c1[tid]; G[a.off[6] + tid] = c2[tid]; }
  for (int i = tid; i < nb * H2; i += T) {
    int j = i % H2;
    float xh = (h2[i] - mean2[j]) * inv2[j];
    g2[i] = g2w[j] * inv2[j] / nb * (nb * g2[i] - c1[j] - xh * c2[j]);
  }
  __syncthreads();
  // layer 2: dW2, db2, da1 -> do1
  for (int i = tid; i < H2 * H1; i += T) {
    int j = i / H1, k = i % H1;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s = fmaf(g2[b * H2 + j], a1[b * H1 + k], s);
    G[a.off[4] + i] = s;
  }
  if (tid < H2) { float s = 0.f; for (int b = 0; b < nb; ++b) s += g2[b * H2 + tid]; G[a.off[5] + tid] = s; }
  for (int i = tid; i < nb * H1; i += T) {
    int b = i / H1, k = i % H1;
    float s = 0.f;
    for (int j = 0; j < H2; ++j) s = fmaf(g2[b * H2 + j], W2[j * H1 + k], s);
    // d(ReLU o Dropout): a1 > 0 iff the unit was kept and its BN output was positive
    float keep = (!a.train || a.p_drop <= 0.f) ? 1.f : keep_scale;
    g1[i] = a1[i] > 0.f ? s * keep : 0.f;
  }
  __syncthreads();
  col_sums2(g1, h1, mean1, inv1, nb, H1, c1, c2, red);
  if (tid < H1) { G[a.off[3] + tid] = c1[tid]; G[a.off[2] + tid] = c2[tid]; }
  for (int i = tid; i < nb * H1; i += T) {
    int j = i % H1;
    float xh = (h1[i] - mean1[j]) * inv1[j];
    g1[i] = g1w[j] * inv1[j] / nb * (nb * g1[i] - c1[j] - xh * c2[j]);
  }
  __syncthreads();
  for (int i = tid; i < H1 * IN; i += T) {
    int j = i / IN, k = i % IN;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s = fmaf(g1[b * H1 + j], x[b * IN + k], s);
    G[a.off[0] + i] = s;
  }
  if (tid < H1) { float s = 0.f; for (int b = 0; b < nb; ++b) s += g1[b * H1 + tid]; G[a.off[1] + tid] = s; }
  __syncthreads();
  if (!a.adam) return;
  // ---- Adam with coupled L2 weight decay (torch.optim.Adam(lr, weight_decay=1e-4), R.md:2625)
  for (long i = tid; i < a.off[10]; i += T) {
    float p = a.P[i], g = G[i] + a.wd * p, m = a.M[i], v = a.V[i];
    m = m + (1.f - a.b1) * (g - m);
    v = a.b2 * v + (1.f - a.b2) * g * g;
    a.P[i] = p - a.step_size * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
    a.M[i] = m; a.V[i] = v;
  }
}
}  // namespace

struct eae_mlp {
  int IN, C, Bm;
  long long off[11], bnoff[5];
  float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr, *bnrun = nullptr;
  long long* nbt = nullptr;
  long long adam_step = 0;
  void* ws = nullptr;
  float *h1, *a1, *h2, *a2, *dlog, *g2, *g1;
};

static long long r4(long long n) { return (n + 3) & ~3LL; }

extern "C" int eae_mlp_layout(int input_dim, int num_classes, long long* param_off, long long* bn_off) {
  if (input_dim <= 0 || input_dim > 1024 || num_classes <= 0 || num_classes > 16) return eae_set_error(EAE_ERR_ARG, "mlp: input_dim in 1..1024, classes in 1..16");
  const long long sz[10] = {128LL * input_dim, 128, 128, 128, 64 * 128, 64, 64, 64, 64LL * num_classes, num_classes};
  long long o = 0;
  for (int i = 0; i < 10; ++i) { if (param_off) param_off[i] = o; o += r4(sz[i]); }
  if (param_off) param_off[10] = o;
  if (bn_off) { bn_off[0] = 0; bn_off[1] = 128; bn_off[2] = 256; bn_off[3] = 320; bn_off[4] = 384; }
  return 0;
}

extern "C" int eae_mlp_create(int input_dim, int num_classes, int max_batch, eae_mlp** out) {
  if (!out || max_batch <= 0) return eae_set_error(EAE_ERR_ARG, "mlp_create: bad argument");
  eae_mlp* m = new eae_mlp();
  if (int rc = eae_mlp_layout(input_dim, num_classes, m->off, m->bnoff)) { delete m; return rc; }
  m->IN = input_dim; m->C = num_classes; m->Bm = max_batch;
  size_t n = (size_t)max_batch * (128 * 3 + 64 * 3 + 16) * 4;
  hipError_t e = hipMalloc(&m->ws, n);
  if (e != hipSuccess) { delete m; return eae_set_error(EAE_ERR_HIP, hipGetErrorString(e)); }
  float* p = (float*)m->ws;
  m->h1 = p; p += (size_t)max_batch * 128; m->a1 = p; p += (size_t)max_batch * 128; m->g1 = p; p += (size_t)max_batch * 128;
  m->h2 = p; p += (size_t)max_batch * 64; m->a2 = p; p += (size_t)max_batch * 64; m->g2 = p; p += (size_t)max_batch * 64;
  m->dlog = p;
  *out = m;
  return 0;
}

extern "C" int eae_mlp_destroy(eae_mlp* m) {
  if (!m) return 0;
  hipDeviceSynchronize();
  if (m->ws) hipFree(m->ws);
  delete m;
  return 0;
}

extern "C" int eae_mlp_bind(eae_mlp* m, float* params, float* grads, float* adam_m, float* adam_v, float* bn_running, long long* bn_nbt) {
  if (!m || !params || !bn_running) return eae_set_error(EAE_ERR_ARG, "mlp_bind: ctx, params and bn_running are required");
  m->P = params; m->G = grads; m->M = adam_m; m->V = adam_v; m->bnrun = bn_running; m->nbt = bn_nbt;
  return 0;
}
extern "C" int eae_mlp_set_adam_step(eae_mlp* m, long long s) { if (!m) return eae_set_error(EAE_ERR_ARG, "mlp is NULL"); m->adam_step = s; return 0; }

static int mlp_launch(eae_mlp* m, hipStream_t st, const float* x, const long long* labels, int B, int train, int backward, int adam,
                      float lr, float wd, unsigned long long seed, const float* drop_mask, float* logits, float* stats,
                      const float* dlog_in = nullptr) {
  if (!m || !x) return eae_set_error(EAE_ERR_ARG, "mlp: NULL argument");
  if (!m->P) return eae_set_error(EAE_ERR_STATE, "eae_mlp_bind has not been called");
  if (B <= 0 || B > m->Bm) return eae_set_error(EAE_ERR_ARG, "mlp: batch size outside 1..max_batch");
  if (train && B < 2) return eae_set_error(EAE_ERR_ARG, "mlp: BatchNorm1d in training mode needs more than 1 sample per batch");
  if (backward && (!m->G || (!labels && !dlog_in))) return eae_set_error(EAE_ERR_STATE, "mlp: gradient arena and labels (or dlogits) required");
  if (adam && (!m->M || !m->V)) return eae_set_error(EAE_ERR_STATE, "mlp: Adam moment arenas required");
  MlpArgs a;
  a.x = x; a.labels = labels; a.B = B; a.IN = m->IN; a.C = m->C;
  a.P = m->P; a.G = m->G; a.M = m->M; a.V = m->V;
  for (int i = 0; i < 11; ++i) a.off[i] = m->off[i];
  a.bnrun = m->bnrun; a.nbt = train ? m->nbt : nullptr;
  a.h1 = m->h1; a.a1 = m->a1; a.h2 = m->h2; a.a2 = m->a2; a.dlog = m->dlog; a.g2 = m->g2; a.g1 = m->g1;
  a.train = train; a.backward = backward; a.adam = adam;
  a.b1 = 0.9f; a.b2 = 0.999f; a.eps = 1e-8f; a.wd = wd; a.step_size = 0.f; a.bc2_sqrt = 1.f;
  if (adam) {
    m->adam_step += 1;
    double bc1 = 1.0 - std::pow(0.9, (double)m->adam_step), bc2 = 1.0 - std::pow(0.999, (double)m->adam_step);
    a.step_size = (float)(lr / bc1); a.bc2_sqrt = (float)std::sqrt(bc2);
  }
  a.seed = seed; a.step = (unsigned long long)m->adam_step; a.drop_mask = drop_mask; a.p_drop = 0.3f;
  a.logits = logits; a.stats = stats; a.dlog_in = dlog_in; a.update_running = dlog_in ? 0 : 1;
  const int grid = train ? 1 : (B + 63) / 64;
  hipLaunchKernelGGL(mlp_kernel, dim3(grid), dim3(T), 0, st, a);
  EAE_LAUNCH_CHECK();
  return 0;
}

extern "C" int eae_mlp_forward(eae_mlp* m, void* stream, const float* x, int B, int train, unsigned long long seed,
                               const float* drop_mask, float* logits) {
  return mlp_launch(m, (hipStream_t)stream, x, nullptr, B, train, 0, 0, 0.f, 0.f, seed, drop_mask, logits, nullptr);
}
extern "C" int eae_mlp_train_step(eae_mlp* m, void* stream, const float* x, const long long* labels, int B, float lr, float weight_decay,
                                  unsigned long long seed, const float* drop_mask, float* logits, float* stats) {
  return mlp_launch(m, (hipStream_t)stream, x, labels, B, 1, 1, 1, lr, weight_decay, seed, drop_mask, logits, stats);
}
extern "C" int eae_mlp_eval_step(eae_mlp* m, void* stream, const float* x, const long long* labels, int B, float* logits, float* stats) {
  return mlp_launch(m, (hipStream_t)stream, x, labels, B, 0, 0, 0, 0.f, 0.f, 0, nullptr, logits, stats);
}

// loss.backward() for a torch-side loss on the logits (R.md:2644-2645): recomputes the train-mode forward of the SAME batch
// (same dropout mask: the Philox key is (seed, optimisation step); running statistics are not touched again) and
// back-propagates the supplied dL/dlogits; gradients land in the gradient arena.
extern "C" int eae_mlp_backward(eae_mlp* m, void* stream, const float* x, int B, unsigned long long seed, const float* drop_mask,
                                const float* dlogits) {
  if (!dlogits) return eae_set_error(EAE_ERR_ARG, "mlp_backward: dlogits is NULL");
  return mlp_launch(m, (hipStream_t)stream, x, nullptr, B, 1, 1, 0, 0.f, 0.f, seed, drop_mask, nullptr, nullptr, dlogits);
}
