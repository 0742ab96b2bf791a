// Diagnostic victims for the co-residency anomaly: a workgroup fills its LDS / registers with a known pattern, idles for a
// while, then checks the pattern.  Launched on a second stream while the igemm kernels run on the first.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void lds_victim(unsigned* report, int words, int spin) {
  extern __shared__ unsigned sm[];
  const int tid = threadIdx.x;
  for (int i = tid; i < words; i += 256) sm[i] = 0xA5000000u ^ (unsigned)(i * 2654435761u) ^ blockIdx.x;
  __syncthreads();
  for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
  __syncthreads();
  unsigned bad = 0, first = 0xffffffffu, got = 0;
  for (int i = tid; i < words; i += 256) {
    unsigned e = 0xA5000000u ^ (unsigned)(i * 2654435761u) ^ blockIdx.x;
    unsigned v = sm[i];
    if (v != e) { if (!bad) { first = i; got = v; } ++bad; }
  }
  if (bad) {
    unsigned slot = atomicAdd(report, 1u);
    if (slot < 60) { report[4 + slot * 4] = blockIdx.x; report[5 + slot * 4] = first; report[6 + slot * 4] = got; report[7 + slot * 4] = bad; }
  }
}

extern "C" __global__ __launch_bounds__(256) void reg_victim(unsigned* report, int spin) {
  unsigned r[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) r[i] = (threadIdx.x * 48 + i) * 2246822519u ^ blockIdx.x;
#pragma unroll
  for (int i = 0; i < 48; ++i) asm volatile("" : "+v"(r[i]));
  for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(64);
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < 48; ++i) { asm volatile("" : "+v"(r[i])); if (r[i] != ((threadIdx.x * 48 + i) * 2246822519u ^ blockIdx.x)) ++bad; }
  if (bad) atomicAdd(report + 1, bad);
}

// a compute victim shaped like the head's first phases: LDS-resident weights, fp32 FMAs, result checksum per block
extern "C" __global__ __launch_bounds__(256) void fma_victim(const float* w, const float* z, float* out, int L, unsigned* report) {
  extern __shared__ float sf[];
  float* w1 = sf; float* zt = sf + 128 * (L + 1);
  const int tid = threadIdx.x;
  for (int i = tid; i < 128 * L; i += 256) w1[(i / L) * (L + 1) + (i % L)] = w[i];
  for (int i = tid; i < 8 * L; i += 256) zt[i] = z[(size_t)blockIdx.x * 8 * L + i];
  __syncthreads();
  const int j = tid & 127, rh = tid >> 7;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < L; ++k) {
    float ww = w1[j * (L + 1) + k];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaf(zt[(rh * 4 + r) * L + k], ww, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[((size_t)blockIdx.x * 8 + rh * 4 + r) * 128 + j] = acc[r];
  // is the LDS image what memory holds?  (re-read memory past the caches)
  unsigned badw = 0, badz = 0;
  for (int i = tid; i < 128 * L; i += 256) if (w1[(i / L) * (L + 1) + (i % L)] != __builtin_nontemporal_load(w + i)) ++badw;
  for (int i = tid; i < 8 * L; i += 256) if (zt[i] != __builtin_nontemporal_load(z + (size_t)blockIdx.x * 8 * L + i)) ++badz;
  if (report && badw) atomicAdd(report + 8, badw);
  if (report && badz) atomicAdd(report + 9, badz);
}

extern "C" int victim_lds(void* stream, unsigned* report, int blocks, int bytes, int spin) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(lds_victim), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  hipLaunchKernelGGL(lds_victim, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, report, bytes / 4, spin);
  return (int)hipGetLastError();
}
extern "C" int victim_reg(void* stream, unsigned* report, int blocks, int spin) {
  hipLaunchKernelGGL(reg_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, report, spin);
  return (int)hipGetLastError();
}
extern "C" int victim_fma(void* stream, const float* w, const float* z, float* out, int blocks, int L, unsigned* report) {
  int bytes = (128 * (L + 1) + 8 * L) * 4;
  hipLaunchKernelGGL(fma_victim, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, w, z, out, L, report);
  return (int)hipGetLastError();
}

// global-load victim: buf[i] = hash(i) written by the host; every thread re-reads a strided set many times and checks it
extern "C" __global__ __launch_bounds__(256) void gload_victim(const unsigned* buf, int n, int iters, unsigned* report) {
  const int tid = threadIdx.x;
  unsigned bad = 0, firsti = 0, got = 0;
  for (int it = 0; it < iters; ++it)
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
      unsigned v = __builtin_nontemporal_load(buf + i);
      unsigned e = (unsigned)i * 2654435761u ^ 0x5bd1e995u;
      if (v != e) { if (!bad) { firsti = i; got = v; } ++bad; }
    }
  if (bad) {
    unsigned slot = atomicAdd(report + 2, 1u);
    if (slot < 30) { report[128 + slot * 4] = blockIdx.x * 256 + tid; report[129 + slot * 4] = firsti; report[130 + slot * 4] = got; report[131 + slot * 4] = bad; }
  }
}
// LDS-read victim: pattern written from registers, then summed many times with ds_read_b32 by every thread (conflict-free and
// broadcast reads, like the head's inner loop); the sum is known in closed form
extern "C" __global__ __launch_bounds__(256) void ldsread_victim(int iters, unsigned* report) {
  __shared__ unsigned sm[128 * 65 + 512];
  const int tid = threadIdx.x;
  for (int i = tid; i < 128 * 65 + 512; i += 256) sm[i] = (unsigned)i * 40503u + 7u;
  __syncthreads();
  const int j = tid & 127, rh = tid >> 7;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned s = 0, e = 0;
    for (int k = 0; k < 64; ++k) {
      unsigned a = sm[j * 65 + k], b = sm[128 * 65 + (rh * 4) * 64 + k];
      s += a ^ b;
      e += ((unsigned)(j * 65 + k) * 40503u + 7u) ^ ((unsigned)(128 * 65 + (rh * 4) * 64 + k) * 40503u + 7u);
    }
    if (s != e) ++bad;
  }
  if (bad) {
    unsigned slot = atomicAdd(report + 3, 1u);
    if (slot < 30) { report[256 + slot * 2] = blockIdx.x * 256 + tid; report[257 + slot * 2] = bad; }
  }
}
extern "C" int victim_gload(void* stream, const unsigned* buf, int n, int iters, unsigned* report, int blocks) {
  hipLaunchKernelGGL(gload_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, n, iters, report);
  return (int)hipGetLastError();
}
extern "C" int victim_ldsread(void* stream, int iters, unsigned* report, int blocks) {
  hipLaunchKernelGGL(ldsread_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, report);
  return (int)hipGetLastError();
}

// ---- synthetic aggressors: a loop of ONE instruction kind (to find which one perturbs the fma victim) ----
typedef __attribute__((ext_vector_type(8))) __bf16 vbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 vbf16x4;
typedef __attribute__((ext_vector_type(4))) float vf32x4;
typedef __attribute__((ext_vector_type(2))) float vf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 vbf16x2;

extern "C" __global__ __launch_bounds__(256) void agg_kernel(float* sink, int iters, int what) {
  vf32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (vf32x4){0.f, 0.f, 0.f, 0.f};
  vbf16x8 a8, b8; vbf16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.001f * (threadIdx.x + i)); b8[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  vf32x2 p = {0.5f + threadIdx.x, 1.5f}, q = {1.0001f, 0.9999f}, r = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (what == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
    } else if (what == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, a4), __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, b4), acc[i], 0, 0, 0);
    } else if (what == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { vbf16x2 c = __builtin_convertvector(p, vbf16x2); p[0] += (float)c[0] * 1e-6f; p[1] += (float)c[1] * 1e-6f; }
    } else if (what == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { r = p * q + r; p = r * q + p; }
    } else if (what == 4) {        // 16-bit / SDWA integer ops
      unsigned short h0 = (unsigned short)(threadIdx.x * 3 + it), h1 = (unsigned short)(threadIdx.x * 7 + 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) { h0 = (unsigned short)(h0 * h1 + (h1 >> 3)); h1 = (unsigned short)(h1 - h0 * 5); }
      p[0] += (float)h0 * 1e-9f; p[1] += (float)h1 * 1e-9f;
    } else if (what == 5) {        // wide LDS traffic + barriers
      __shared__ __attribute__((aligned(16))) float lbuf[256 * 8 + 64];
      float4 v = {p[0], p[1], q[0], q[1]};
      *reinterpret_cast<float4*>(lbuf + threadIdx.x * 4) = v;
      __syncthreads();
      float4 u = *reinterpret_cast<const float4*>(lbuf + ((threadIdx.x * 5 + it) & 255) * 4);
      __syncthreads();
      p[0] += u.x * 1e-9f; p[1] += u.w * 1e-9f;
    } else if (what == 6) {        // 128-bit global load/store streaming (sink must hold blocks*256*4 floats)
      float4* g = reinterpret_cast<float4*>(sink) + (size_t)blockIdx.x * 256 + threadIdx.x;
      float4 u = *g; u.x += 1.f; *g = u;
    } else if (what == 8) {        // MFMA whose C operand is the inline constant 0 (first MFMA of an accumulation chain)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vf32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, (vf32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[i] += t;
      }
    } else if (what == 10 || what == 11) {   // as 8, with extra wait states between the MFMA and the first read of its result
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vbf16x8 ai = a8; ai[0] = (__bf16)(float)(i + it);         // distinct MFMAs (no CSE)
        vf32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ai, b8, (vf32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (what == 11) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(t));
        else asm volatile("" : "+v"(t));
        acc[i] += t;
      }
    } else if (what == 9) {        // same chain shape with the zero in registers
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vf32x4 zr = {0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(zr));
        vf32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, zr, 0, 0, 0);
        acc[i] += t;
      }
    } else if (what == 7) {        // accumulators forced into AGPRs
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\ts_nop 4\n\tv_accvgpr_read_b32 %0, a1" : "+v"(p[0]) :: "a0", "a1");
      }
    }
  }
  float s = p[0] + p[1] + r[0] + r[1];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f && what != 6) sink[threadIdx.x] = s;
}
extern "C" int aggressor(void* stream, float* sink, int blocks, int iters, int what) {
  hipLaunchKernelGGL(agg_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters, what);
  return (int)hipGetLastError();
}

// register-only packed-FMA victim: the fma victim's arithmetic (v_pk_fma_f32 with op_sel broadcasts) without any LDS or memory
// operand inside the loop; mode 1 = the same arithmetic with scalar v_fma_f32
extern "C" __global__ __launch_bounds__(256) void pk_victim(float* out, int iters, int mode) {
  const int tid = threadIdx.x;
  vf32x2 a01 = {0.001f * tid, 0.002f * tid + 0.1f}, a23 = {0.003f * tid - 0.2f, 0.0005f * tid + 0.3f};
  vf32x2 b = {1.0f + 1e-3f * (tid & 15), 0.5f - 1e-3f * (tid >> 4)};
  vf32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {
      acc01 = a01 * (vf32x2){b[0], b[0]} + acc01;      // op_sel_hi:[1,0,1]
      acc23 = a23 * (vf32x2){b[0], b[0]} + acc23;
      acc01 = a01 * (vf32x2){b[1], b[1]} + acc01;      // op_sel:[0,1,0]
      acc23 = a23 * (vf32x2){b[1], b[1]} + acc23;
    } else {
      acc01[0] = __builtin_fmaf(a01[0], b[0], acc01[0]); acc01[1] = __builtin_fmaf(a01[1], b[0], acc01[1]);
      acc23[0] = __builtin_fmaf(a23[0], b[0], acc23[0]); acc23[1] = __builtin_fmaf(a23[1], b[0], acc23[1]);
      acc01[0] = __builtin_fmaf(a01[0], b[1], acc01[0]); acc01[1] = __builtin_fmaf(a01[1], b[1], acc01[1]);
      acc23[0] = __builtin_fmaf(a23[0], b[1], acc23[0]); acc23[1] = __builtin_fmaf(a23[1], b[1], acc23[1]);
      asm volatile("" : "+v"(acc01), "+v"(acc23));
    }
    acc01 *= (vf32x2){0.999f, 0.999f}; acc23 *= (vf32x2){0.999f, 0.999f};
    b[0] += 1e-6f; b[1] -= 1e-6f;
  }
  size_t o = ((size_t)blockIdx.x * 256 + tid) * 4;
  out[o] = acc01[0]; out[o + 1] = acc01[1]; out[o + 2] = acc23[0]; out[o + 3] = acc23[1];
}
extern "C" int victim_pk(void* stream, float* out, int blocks, int iters, int mode) {
  hipLaunchKernelGGL(pk_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, mode);
  return (int)hipGetLastError();
}

// pk victim with a padded register allocation (keeps `pad` live registers) -> does the victim's allocation size matter?
template <int PAD>
__global__ __launch_bounds__(256) void pk_victim_pad(float* out, int iters) {
  const int tid = threadIdx.x;
  float padr[PAD];
#pragma unroll
  for (int i = 0; i < PAD; ++i) { padr[i] = tid * 0.25f + i; asm volatile("" : "+v"(padr[i])); }
  vf32x2 a01 = {0.001f * tid, 0.002f * tid + 0.1f}, a23 = {0.003f * tid - 0.2f, 0.0005f * tid + 0.3f};
  vf32x2 b = {1.0f + 1e-3f * (tid & 15), 0.5f - 1e-3f * (tid >> 4)};
  vf32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    acc01 = a01 * (vf32x2){b[0], b[0]} + acc01;
    acc23 = a23 * (vf32x2){b[0], b[0]} + acc23;
    acc01 = a01 * (vf32x2){b[1], b[1]} + acc01;
    acc23 = a23 * (vf32x2){b[1], b[1]} + acc23;
    acc01 *= (vf32x2){0.999f, 0.999f}; acc23 *= (vf32x2){0.999f, 0.999f};
    b[0] += 1e-6f; b[1] -= 1e-6f;
  }
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < PAD; ++i) { asm volatile("" : "+v"(padr[i])); ps += padr[i] - (tid * 0.25f + i); }   // must be exactly 0
  size_t o = ((size_t)blockIdx.x * 256 + tid) * 4;
  out[o] = acc01[0] + ps; out[o + 1] = acc01[1]; out[o + 2] = acc23[0]; out[o + 3] = acc23[1];
}
extern "C" int victim_pk_pad(void* stream, float* out, int blocks, int iters, int pad) {
  if (pad == 100) hipLaunchKernelGGL((pk_victim_pad<100>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL((pk_victim_pad<230>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}

// victim and aggressor inside ONE kernel (same code object, same register allocation): even workgroups run the packed-FMA
// loop, odd workgroups the MFMA -> VALU-read loop
extern "C" __global__ __launch_bounds__(256) void mixed_kernel(float* out, float* sink, int iters) {
  const int tid = threadIdx.x;
  if (blockIdx.x & 1) {
    vf32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (vf32x4){0.f, 0.f, 0.f, 0.f};
    vbf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.001f * (tid + i)); b8[i] = (__bf16)(0.002f * (tid * 3 + i)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        vf32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, (vf32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[i] += t;
      }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[tid] = s;
    return;
  }
  vf32x2 a01 = {0.001f * tid, 0.002f * tid + 0.1f}, a23 = {0.003f * tid - 0.2f, 0.0005f * tid + 0.3f};
  vf32x2 b = {1.0f + 1e-3f * (tid & 15), 0.5f - 1e-3f * (tid >> 4)};
  vf32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    acc01 = a01 * (vf32x2){b[0], b[0]} + acc01;
    acc23 = a23 * (vf32x2){b[0], b[0]} + acc23;
    acc01 = a01 * (vf32x2){b[1], b[1]} + acc01;
    acc23 = a23 * (vf32x2){b[1], b[1]} + acc23;
    acc01 *= (vf32x2){0.999f, 0.999f}; acc23 *= (vf32x2){0.999f, 0.999f};
    b[0] += 1e-6f; b[1] -= 1e-6f;
  }
  size_t o = ((size_t)(blockIdx.x >> 1) * 256 + tid) * 4;
  out[o] = acc01[0]; out[o + 1] = acc01[1]; out[o + 2] = acc23[0]; out[o + 3] = acc23[1];
}
extern "C" int mixed(void* stream, float* out, float* sink, int blocks, int iters) {
  hipLaunchKernelGGL(mixed_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, sink, iters);
  return (int)hipGetLastError();
}

// pk victim whose register ALLOCATION is raised by clobbering a high VGPR once (nothing is kept there)
#define PK_ALLOC_VICTIM(N) PK_ALLOC_VICTIM2(N, "v_mov_b32 v" #N ", 0", "v" #N)
#define PK_ALLOC_VICTIM2(N, INSTR, CLOB)                                                                     \
  extern "C" __global__ __launch_bounds__(256) void pk_victim_alloc##N(float* out, int iters) {             \
    asm volatile(INSTR ::: CLOB);                                                                            \
    const int tid = threadIdx.x;                                                                             \
    vf32x2 a01 = {0.001f * tid, 0.002f * tid + 0.1f}, a23 = {0.003f * tid - 0.2f, 0.0005f * tid + 0.3f};     \
    vf32x2 b = {1.0f + 1e-3f * (tid & 15), 0.5f - 1e-3f * (tid >> 4)};                                       \
    vf32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};                                                           \
    for (int it = 0; it < iters; ++it) {                                                                     \
      acc01 = a01 * (vf32x2){b[0], b[0]} + acc01; acc23 = a23 * (vf32x2){b[0], b[0]} + acc23;                \
      acc01 = a01 * (vf32x2){b[1], b[1]} + acc01; acc23 = a23 * (vf32x2){b[1], b[1]} + acc23;                \
      acc01 *= (vf32x2){0.999f, 0.999f}; acc23 *= (vf32x2){0.999f, 0.999f};                                  \
      b[0] += 1e-6f; b[1] -= 1e-6f;                                                                          \
    }                                                                                                        \
    size_t o = ((size_t)blockIdx.x * 256 + tid) * 4;                                                         \
    out[o] = acc01[0]; out[o + 1] = acc01[1]; out[o + 2] = acc23[0]; out[o + 3] = acc23[1];                  \
  }
PK_ALLOC_VICTIM2(0, "s_nop 0", "memory") PK_ALLOC_VICTIM2(1, "v_nop", "memory") PK_ALLOC_VICTIM2(2, "s_nop 0", "v19") PK_ALLOC_VICTIM(19) PK_ALLOC_VICTIM(23) PK_ALLOC_VICTIM(27) PK_ALLOC_VICTIM(29) PK_ALLOC_VICTIM(31) PK_ALLOC_VICTIM(47) PK_ALLOC_VICTIM(59) PK_ALLOC_VICTIM(61) PK_ALLOC_VICTIM(63) PK_ALLOC_VICTIM(71) PK_ALLOC_VICTIM(79) PK_ALLOC_VICTIM(95) PK_ALLOC_VICTIM(127)
extern "C" int victim_pk_alloc(void* stream, float* out, int blocks, int iters, int n) {
  hipStream_t st = (hipStream_t)stream;
#define L(N) if (n == N) hipLaunchKernelGGL(pk_victim_alloc##N, dim3(blocks), dim3(256), 0, st, out, iters);
  L(0) L(1) L(2) L(19) L(23) L(27) L(29) L(59) L(61) L(31) L(47) L(63) L(71) L(79) L(95) L(127)
#undef L
  return (int)hipGetLastError();
}

// A/B on the instruction form: the same loop with the b[1]-broadcast written as inline asm, either with op_sel:[0,1,0]
// (low result lane reads the HIGH source register) or by first copying b[1] into a pair's low register (no op_sel)
#define PK_FORM_VICTIM(NAME, STMT)                                                                           \
  extern "C" __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                            \
    const int tid = threadIdx.x;                                                                             \
    vf32x2 a01 = {0.001f * tid, 0.002f * tid + 0.1f}, a23 = {0.003f * tid - 0.2f, 0.0005f * tid + 0.3f};     \
    vf32x2 b = {1.0f + 1e-3f * (tid & 15), 0.5f - 1e-3f * (tid >> 4)};                                       \
    vf32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};                                                           \
    for (int it = 0; it < iters; ++it) {                                                                     \
      acc01 = a01 * (vf32x2){b[0], b[0]} + acc01; acc23 = a23 * (vf32x2){b[0], b[0]} + acc23;                \
      STMT                                                                                                   \
      acc01 *= (vf32x2){0.999f, 0.999f}; acc23 *= (vf32x2){0.999f, 0.999f};                                  \
      b[0] += 1e-6f; b[1] -= 1e-6f;                                                                          \
    }                                                                                                        \
    size_t o = ((size_t)blockIdx.x * 256 + tid) * 4;                                                         \
    out[o] = acc01[0]; out[o + 1] = acc01[1]; out[o + 2] = acc23[0]; out[o + 3] = acc23[1];                  \
  }
PK_FORM_VICTIM(pk_form_opsel,
  asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel:[0,1,0]"
               : "+v"(acc01), "+v"(acc23) : "v"(a01), "v"(a23), "v"(b));)
PK_FORM_VICTIM(pk_form_copy,
  vf32x2 bb; bb[0] = b[1]; bb[1] = b[1];
  asm volatile("v_pk_fma_f32 %0, %2, %4, %0\n\tv_pk_fma_f32 %1, %3, %4, %1"
               : "+v"(acc01), "+v"(acc23) : "v"(a01), "v"(a23), "v"(bb));)
extern "C" int victim_pk_form(void* stream, float* out, int blocks, int iters, int form) {
  if (form == 0) hipLaunchKernelGGL(pk_form_opsel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(pk_form_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
