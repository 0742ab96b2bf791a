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
  int q_layer;         // >= 0: destination is e4m3 bytes scaled by the weight scale of 3x3 layer q_layer (fp8 variant, Fp8State below)
  // flattened launch (eae_pack_assign_blocks): the descriptor owns workgroups [blk0, blk0 + nblk) of the list it belongs to, nblk
  // proportional to its work; 0 / 0 = the 2-D launch (blocks_per_desc workgroups for every descriptor)
  int blk0, nblk;
};

// fp8 variant (BASELINE config 5) -- delayed scaling state in device memory, one per context.  Index i = 3x3 layer in W3 order
// (conv2, conv3, conv4, deconv1, deconv2, deconv3).  A scale MULTIPLIES a value before its conversion to fp8.
constexpr int FP8_AMAX_SLOTS = 64;       // (256 slots measured no better: 2.22 vs 2.19 ms per config-5 step)
constexpr int FP8_AMAX_STRIDE = 32;     // words between two slots: 128 bytes, so that the slots of one reporter class sit in different cache lines / channels
struct Fp8State {
  // maxima reported by this step's kernels, FP8_AMAX_SLOTS words each (a reporter picks one by its workgroup index: atomics on one
  // word serialise at the memory side); fp8_scales_kernel takes the maximum over the words
  unsigned amax_act[6][FP8_AMAX_SLOTS * FP8_AMAX_STRIDE];    // largest |input activation operand| of layer i seen by the forward kernel (bf16 bits << 16)
  unsigned amax_grad[6][FP8_AMAX_SLOTS * FP8_AMAX_STRIDE];   // largest |output-gradient operand| of layer i seen by the backward-data kernel
  unsigned amax_w[6][FP8_AMAX_SLOTS * FP8_AMAX_STRIDE];      // largest |weight| of layer i seen by the pack kernel
  float s_act[6], s_grad[6], s_w[6];
  float qs_fwd[6][2];      // ConvArgs::qs of the forward kernel:        1/s_act,  1/(s_act*s_w)
  float qs_bwd[6][2];      // ConvArgs::qs of the backward-data kernel:  1/s_grad, 1/(s_grad*s_w)
  float qs_wg[6][4];       // WgradArgs::qs: 1/s_small, 1/s_big, 1/(s_small*s_big)
};
void eae_fp8_state_init(Fp8State* host);
int eae_launch_fp8_scales(hipStream_t st, Fp8State* q);

int eae_launch_bn_finalize(hipStream_t st, const float* part, int ntiles, int C, long long count, const float* gamma,
                           const float* beta, float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef);
int eae_launch_bn_eval_coef(hipStream_t st, int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                            float eps, float* coef);
int eae_launch_bn_bwd_finalize(hipStream_t st, const float* part, int ntiles, int C, long long count, const float* gamma,
                               const float* coef_fwd, float* dgamma, float* dbeta, float* coef_bwd);
int eae_launch_bn_bwd_reduce(hipStream_t st, const float* part, int ntiles, int C, double* sums, float* dgamma, float* dbeta);
int eae_launch_bn_bwd_coef(hipStream_t st, const double* sums, int C, long long count, const float* gamma, const float* coef_fwd,
                           float* coef_bwd);
// limit_ticks: bound of the spin in s_memrealtime ticks (100 MHz), 0 = unbounded; timeout: STICKY word that receives the value the
// gate gave up waiting for (the optimizer kernels refuse to update while it is set, eae_gate_timeouts())
struct GateArgs { unsigned* word[4]; unsigned want[4]; int n; unsigned* timeout; unsigned long long limit_ticks; };
int eae_launch_gate(hipStream_t st, const GateArgs& g);
// the wait itself (ONE lane calls it): gate_kernel's body, also run in the tail of conv1's weight-gradient reduction
__device__ __forceinline__ void gate_wait(const GateArgs& g) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < g.n; ++k) {
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(g.word[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - g.want[k]) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if ((++spins & 255u) == 0 && g.limit_ticks != 0 && __builtin_amdgcn_s_memrealtime() - t0 > g.limit_ticks) {
        atomicExch(g.timeout, g.want[k] ? g.want[k] : 1u);
        return;
      }
    }
  }
}
int eae_launch_signal(hipStream_t st, unsigned* word, unsigned val);
// blocks_per_desc: workgroups per descriptor (grid.x; every descriptor loops over its elements / tiles with that stride)
int eae_launch_pack_all(hipStream_t st, const PackDesc* descs_dev, int ndesc, const float* params, void* pack_base, Fp8State* q = nullptr, unsigned* clear_word = nullptr,
                        int blocks_per_desc = 256);
// Flattened form: workgroups [blk_begin, blk_begin + blk_count) of the list `descs_all` (whole list, device) that eae_pack_assign_blocks
// laid out; blkmap[b] (device) = index of the descriptor that owns workgroup b.
int eae_pack_assign_blocks(PackDesc* descs_host, int ndesc, unsigned short* blkmap_host, int blkmap_cap);      // returns the total, fills blk0 / nblk / the map
int eae_launch_pack_flat(hipStream_t st, const PackDesc* descs_all, const unsigned short* blkmap, int blk_begin, int blk_count, const float* params,
                         void* pack_base, Fp8State* q = nullptr, unsigned* clear_word = nullptr);
int eae_launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2,
                    double eps, double wd, long long step);
int eae_launch_adam_dyn(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double b1, double b2, double eps,
                        const float* dyn, const unsigned* bad = nullptr, const unsigned* bad2 = nullptr, float* nan_out = nullptr);
int eae_launch_set_dyn(hipStream_t st, float* dyn, double lr, double b1, double b2, double wd, long long step);
int eae_launch_adam_scaled(hipStream_t st, float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2,
                           double eps, double wd, long long step, float gscale, void* zero_buf = nullptr, long long zero_bytes = 0,
                           const unsigned* bad = nullptr, const unsigned* bad2 = nullptr, float* nan_out = nullptr, int nan_fill = 0,
                           int max_blocks = 0 /* 0: 2048; the grid-stride loop covers the rest */);
int eae_launch_augment(hipStream_t st, const void* in_u8, float* out, int B, int H, int W, int train, float std, unsigned long long seed,
                       unsigned long long step, const int* params, const float* noise);
