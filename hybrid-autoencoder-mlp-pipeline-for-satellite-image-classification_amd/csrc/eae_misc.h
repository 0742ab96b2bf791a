#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { PACK_COPY = 0, PACK_3x3_P1, PACK_3x3_P2, PACK_K27, PACK_DECONV4_JOINT, PACK_FC_ROWMAJOR_KPERM, PACK_FC_TRANS_KPERM,
       PACK_FC_ROWPERM, PACK_FC_ROWPERM_TRANS, PACK_K36, PACK_PAD_COLS };

struct PackDesc {
  long long src_off;   // element offset into the fp32 parameter arena
  long long dst_off;   // byte offset into the pack arena
  long long count;     // destination elements
  int mode, d0, d1, d2;
  int out_f32;         // 1: destination is fp32 (permuted biases), 0: bf16
  int lv;              // FC modes / PACK_PAD_COLS: number of REAL entries along the latent dimension d0 (the rest of d0 is zero padding)
};

int eae_launch_bn_finalize(hipStream_t st, const float* part, int ntiles, int C, long long count, const float* gamma,
                           const float* beta, float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef);
int eae_launch_bn_eval_coef(hipStream_t st, int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                            float eps, float* coef);
int eae_launch_bn_bwd_finalize(hipStream_t st, const float* part, int ntiles, int C, long long count, const float* gamma,
                               const float* coef_fwd, float* dgamma, float* dbeta, float* coef_bwd);
int eae_launch_bn_bwd_reduce(hipStream_t st, const float* part, int ntiles, int C, double* sums, float* dgamma, float* dbeta);
int eae_launch_bn_bwd_coef(hipStream_t st, const double* sums, int C, long long count, const float* gamma, const float* coef_fwd,
                           float* coef_bwd);
struct GateArgs { unsigned* word[4]; unsigned want[4]; int n; unsigned* timeout; };
int eae_launch_gate(hipStream_t st, const GateArgs& g);
int eae_launch_signal(hipStream_t st, unsigned* word, unsigned val);
int eae_launch_pack_all(hipStream_t st, const PackDesc* descs_dev, int ndesc, const float* params, void* pack_base);
int eae_launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2,
                    double eps, double wd, long long step);
int eae_launch_adam_dyn(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double b1, double b2, double eps,
                        const float* dyn);
int eae_launch_set_dyn(hipStream_t st, float* dyn, double lr, double b1, double b2, double wd, long long step);
int eae_launch_adam_scaled(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2,
                           double eps, double wd, long long step, float gscale, void* zero_buf = nullptr, long long zero_bytes = 0);
int eae_launch_augment(hipStream_t st, const void* in_u8, float* out, int B, int H, int W, int train, float std, unsigned long long seed,
                       unsigned long long step, const int* params, const float* noise);
