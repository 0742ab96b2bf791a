// Hardware probe for the fp8 path: (1) semantics of v_cvt_scalef32_pk_fp8_bf16's scale operand and its saturation, (2) that
// v_mfma_f32_16x16x32_fp8_fp8 takes its 8 K-values per lane in the same order as the bf16 form (k = 8*(lane>>4) + j).
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/probe_fp8.hip -o tools/probe/probe_fp8 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ long cvt8(bf16x8 v, float scale) {
  s16x2 lo = {0, 0}, hi = {0, 0};
  bf16x2 a = {v[0], v[1]}, b = {v[2], v[3]}, c = {v[4], v[5]}, d = {v[6], v[7]};
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, a, scale, false);
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, b, scale, true);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, c, scale, false);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, d, scale, true);
  union { s16x2 h[2]; long l; } u; u.h[0] = lo; u.h[1] = hi;
  return u.l;
}
__global__ void cvt_probe(const float* in, float scale, unsigned char* out, int ovfl) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");     // MODE.FP16_OVFL: out-of-range conversions clamp to +-max
  bf16x8 v;
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)in[j];
  long r = cvt8(v, scale);
  for (int j = 0; j < 8; ++j) out[j] = (unsigned char)(r >> (8 * j));
}
// A[16][32] (row m), B[16][32] (row n): D[m][n] = sum_k A[m][k] B[n][k]; lane l supplies row (l & 15), k = 8*(l>>4)+j
__global__ void mfma_probe(const float* A, const float* B, float* Dbf, float* Df8) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = (__bf16)B[(l & 15) * 32 + 8 * (l >> 4) + j]; }
  f32x4 z = {0, 0, 0, 0};
  f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, z, 0, 0, 0);
  f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(cvt8(a, 1.0f), cvt8(b, 1.0f), z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { Dbf[l * 4 + r] = d1[r]; Df8[l * 4 + r] = d2[r]; }
}
int main() {
  float h[8] = {1.0f, 0.5f, -3.0f, 448.0f, 1000.0f, 0.001953125f, 0.0009765625f, 17.0f};
  float* din; unsigned char* dout; unsigned char ho[8];
  hipMalloc(&din, 32); hipMalloc(&dout, 8);
  hipMemcpy(din, h, 32, hipMemcpyHostToDevice);
  for (int ov = 0; ov < 2; ++ov)
  for (float s : {1.0f, 2.0f, 0.5f}) {
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(1), 0, 0, din, s, dout, ov);
    hipMemcpy(ho, dout, 8, hipMemcpyDeviceToHost);
    printf("ovfl %d scale %.2f:", ov, s);
    for (int j = 0; j < 8; ++j) printf(" %g->0x%02x", h[j], ho[j]);
    printf("\n");
  }
  float A[512], B[512], D1[256], D2[256];
  srand(1);
  const float vals[8] = {0.f, 0.5f, 1.f, -1.f, 2.f, -0.25f, 1.5f, -3.f};      // exact in e4m3 and bf16
  for (int i = 0; i < 512; ++i) { A[i] = vals[rand() & 7]; B[i] = vals[rand() & 7]; }
  float *dA, *dB, *dD1, *dD2;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD1, 1024); hipMalloc(&dD2, 1024);
  hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD1, dD2);
  hipMemcpy(D1, dD1, 1024, hipMemcpyDeviceToHost); hipMemcpy(D2, dD2, 1024, hipMemcpyDeviceToHost);
  double md = 0;
  for (int i = 0; i < 256; ++i) md = fmax(md, fabs(D1[i] - D2[i]));
  printf("mfma bf16 vs fp8 (operands exact in e4m3): max |diff| = %g (D[0..3] = %g %g %g %g)\n", md, D1[0], D1[1], D1[2], D1[3]);
  return 0;
}
