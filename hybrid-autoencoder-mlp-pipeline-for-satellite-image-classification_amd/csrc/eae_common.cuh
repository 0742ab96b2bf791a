// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the conv-autoencoder path.
// Wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16 (lane maps verified on hardware by tools/probe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;     // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) short s16x4;

// activation "sources": how a logical NHWC activation tensor is materialised when it is loaded
enum { SRC_RAW = 0,      // bf16 tensor as stored
       SRC_BNRELU = 1,   // max(0, s[c]*y + t[c])            (BatchNorm apply + ReLU fused into the consumer's load)
       SRC_BNBWD = 2,    // A[c]*g + B[c]*y + C[c]           (BatchNorm backward apply fused into the consumer's load)
       SRC_F32 = 3 };    // fp32 tensor, converted to bf16 on load

// epilogues of the conv-like kernels
enum { EPI_FWD = 0,      // + bias, store raw bf16, per-channel sum / sum-of-squares partials (BatchNorm batch statistics)
       EPI_MASK = 1,     // ReLU mask from the previous layer's BN output, store masked grad, sum g / sum g*xhat partials
       EPI_PLAIN = 2 };  // store bf16

struct SrcDesc {
  const bf16_t* p0;    // RAW: tensor; BNRELU: raw pre-BN tensor y; BNBWD: masked gradient g
  const bf16_t* p1;    // BNBWD: raw pre-BN tensor y
  const float* coef;   // BNRELU: [4][C] = s, t, mean, invstd ; BNBWD: [3][C] = A, B, C
};

__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {           // round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
  __bf16 b = (__bf16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf2f(v.x & 0xffffu); f[1] = bf2f(v.x >> 16);
  f[2] = bf2f(v.y & 0xffffu); f[3] = bf2f(v.y >> 16);
  f[4] = bf2f(v.z & 0xffffu); f[5] = bf2f(v.z >> 16);
  f[6] = bf2f(v.w & 0xffffu); f[7] = bf2f(v.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2(f[0], f[1]); v.y = pack2(f[2], f[3]); v.z = pack2(f[4], f[5]); v.w = pack2(f[6], f[7]);
  return v;
}

// Per-thread coefficients of 8 consecutive channels for a source transform.
template <int MODE> struct ChanCoef {
  float a[8], b[8], c[8];
  __device__ __forceinline__ void load(const float* coef, int C, int ch0) {
    if (MODE == SRC_BNRELU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = coef[ch0 + j]; b[j] = coef[C + ch0 + j]; }
    } else if (MODE == SRC_BNBWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = coef[ch0 + j]; b[j] = coef[C + ch0 + j]; c[j] = coef[2 * C + ch0 + j]; }
    }
  }
};

// Raw 16-byte loads of one 8-channel piece (second tensor only for BNBWD).
template <int MODE> struct RawPiece { uint4 v0, v1; };

template <int MODE>
__device__ __forceinline__ void load_piece(const SrcDesc& s, size_t off, bool valid, RawPiece<MODE>& r) {
  r.v0 = make_uint4(0, 0, 0, 0);
  r.v1 = make_uint4(0, 0, 0, 0);
  if (valid) {
    r.v0 = *reinterpret_cast<const uint4*>(s.p0 + off);
    if (MODE == SRC_BNBWD) r.v1 = *reinterpret_cast<const uint4*>(s.p1 + off);
  }
}

// Apply the source transform to a raw piece; out-of-image pieces are exact zeros (zero padding applies to the
// transformed activation, not to the stored tensor).
template <int MODE>
__device__ __forceinline__ uint4 transform_piece(const RawPiece<MODE>& r, bool valid, const ChanCoef<MODE>& cc) {
  if (MODE == SRC_RAW) return r.v0;
  if (!valid) return make_uint4(0, 0, 0, 0);
  float x[8], o[8];
  unpack8(r.v0, x);
  if (MODE == SRC_BNRELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaxf(fmaf(cc.a[j], x[j], cc.b[j]), 0.0f);
  } else {
    float y[8];
    unpack8(r.v1, y);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaf(cc.a[j], x[j], fmaf(cc.b[j], y[j], cc.c[j]));
  }
  return pack8(o);
}

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Transposed fragment read: LDS image stored [reduction row][16 columns...]; the lane receives, for column (lane&15),
// the 8 reduction rows 8*(lane>>4) .. +7.  `row_ptr_lo/hi` = this lane's row pointers per the ds_read_b64_tr_b16 rule:
// lane 4q+p of each 16-lane group supplies row q (lo: rows 0..3, hi: rows 4..7 of the group's 8), columns 4p..4p+3.
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* row_ptr_lo, const bf16_t* row_ptr_hi) {
  typedef s16x4 __attribute__((address_space(3))) * lp;
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(row_ptr_lo));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(row_ptr_hi));
  return u.v;
}

#define EAE_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return eae_set_error(-3, hipGetErrorString(e__)); } while (0)
