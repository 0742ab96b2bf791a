#pragma once
#include <hip/hip_runtime.h>

struct HeadArgs {
  const float* z;          // [B][L] fp32 latent
  const float *w1, *b1, *w2, *b2;   // classifier.0 [128][L],[128]; classifier.2 [C][128],[C]  (fp32 master weights)
  const long long* labels; // int64 [B] or nullptr (forward only)
  const float* dlogits_in; // [B][C] externally supplied dL/dlogits (autograd path) or nullptr
  int B, L, C;
  float inv_batch;         // 1/B (CrossEntropyLoss mean reduction)
  float* logits;           // [B][C] or nullptr
  float* dz;               // [B][L] gradient of the head w.r.t. z
  float* grad_part;        // [nblocks][grad_stride] partial weight gradients in arena order, or nullptr
  long long grad_stride;
  float* loss_part;        // [nblocks][2]: sum of per-row CE, number of correct argmax
};
int eae_launch_head(hipStream_t st, const HeadArgs& a);
int eae_head_blocks(int B, int L);     // partial rows of loss_part / grad_part = blocks of the launch for latent width L
int eae_launch_ce_mean(hipStream_t st, const float* ce_part, int n, int B, float* out2);
int eae_launch_loss_finalize(hipStream_t st, const float* mse_part, int n_mse, const float* ce_part, int n_ce, float alpha,
                             double numel, int B, float* db4, float* accum, float* last, const unsigned* poison = nullptr);
