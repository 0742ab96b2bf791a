// Host-side dispatch of the latent-projection GEMMs.
#include "eae_internal.h"
#include "eae_fc.hip.h"

int eae_launch_fc_nt(hipStream_t st, const FcNtArgs& a0, int amode, int epi, int ksplit) {
  FcNtArgs a = a0;
  a.c.ntiles = ((a.M + 127) / 128) * (a.N / 256 > 0 ? a.N / 256 : 1);
  if (a.N % 64 || a.K % 64 || a.klen % 64 || a.klen * ksplit != a.K) return eae_set_error(-2, "fc_nt: N, K, klen must be multiples of 64");
  if (epi != FCE_PARTIAL && (a.N % 256 || ksplit != 1)) return eae_set_error(-2, "fc_nt: fused epilogues need N % 256 == 0 and no split-K");
  dim3 grid((a.M + 127) / 128, a.N / 64, ksplit);
#define CASE(AM, E) if (amode == AM && epi == E) { eae_launch(fc_nt_kernel<AM, E>, fc_nt_kernel_g<AM, E>, grid, dim3(256), 0, st, a); EAE_LAUNCH_CHECK(); return 0; }
  CASE(SRC_BNRELU, FCE_PARTIAL)     // enc.fc forward
  CASE(SRC_RAW, FCE_PARTIAL)        // dec.fc backward-data
  CASE(SRC_F32, FCE_BIAS_BF16)      // dec.fc forward
  CASE(SRC_F32, FCE_MASK)           // enc.fc backward-data
#undef CASE
  return eae_set_error(-2, "fc_nt: combination not instantiated");
}

int eae_launch_fc_reduce(hipStream_t st, const float* part, int nsl, int M, int N, const float* bias, const float* addend,
                         const float* addend2, float* out) {
  long n4 = (long)M * N / 4;
  const FcReduceArgs ra = {part, nsl, M, N, bias, addend, addend2, out};
  eae_launch(fc_splitk_reduce_kernel, fc_splitk_reduce_kernel_g, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, ra);
  EAE_LAUNCH_CHECK();
  return 0;
}

int eae_launch_fc_tn(hipStream_t st, const FcTnArgs& a, int pmode, int qmode) {
  if (a.I % 64 || a.J % 64) return eae_set_error(-2, "fc_tn: I and J must be multiples of 64");
  dim3 grid(a.I / 64, a.J / 64);
#define CASE(PM, QM) if (pmode == PM && qmode == QM) { eae_launch(fc_tn_kernel<PM, QM>, fc_tn_kernel_g<PM, QM>, grid, dim3(256), 0, st, a); EAE_LAUNCH_CHECK(); return 0; }
  CASE(SRC_RAW, SRC_F32)       // dec.fc weight gradient: P = g_d0, Q = z
  CASE(SRC_F32, SRC_BNRELU)    // enc.fc weight gradient: P = dz,   Q = BNRELU(y4)
#undef CASE
  return eae_set_error(-2, "fc_tn: combination not instantiated");
}
