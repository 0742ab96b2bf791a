// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the conv-autoencoder path.
// Wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16 (lane maps verified on hardware by tools/probe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "eae_group.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;     // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) short s16x4;

// activation "sources": how a logical NHWC activation tensor is materialised when it is loaded
enum { SRC_RAW = 0,      // bf16 tensor as stored
       SRC_BNRELU = 1,   // max(0, s[c]*y + t[c])            (BatchNorm apply + ReLU fused into the consumer's load)
       SRC_BNBWD = 2,    // A[c]*g + B[c]*y + C[c]           (BatchNorm backward apply fused into the consumer's load)
       SRC_F32 = 3,      // fp32 tensor, converted to bf16 on load
       SRC_RAWG = 4 };   // bf16 GRADIENT tensor as stored (the BatchNorm-backward-applied dy a backward-data kernel wrote while it staged
                         // its patch, ConvArgs::dy_out): loads like SRC_RAW; the fp8 variants convert it to e5m2 like SRC_BNBWD

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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one dword = 2 bf16 <-> 2 fp32 (register pair, so that the arithmetic becomes v_pk_fma_f32)
__device__ __forceinline__ f32x2 up2(uint32_t w) { return (f32x2){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ uint32_t pk2(f32x2 v) {      // v_cvt_pk_bf16_f32 (round-to-nearest-even)
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t relu2(uint32_t p) { // ReLU on two packed bf16: v_pk_max_i16 with 0 (negative <=> sign bit)
  s16x2 v = __builtin_bit_cast(s16x2, p);
  s16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, z));
}

// fp32 ReLU that lets a NaN through like torch.relu does (fmaxf returns the other operand): a diverged run stays loud downstream
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

// Per-thread coefficients of 8 consecutive channels for a source transform (as 4 fp32 pairs).
template <int MODE> struct ChanCoef {
  f32x2 a[4], b[4], c[4];
  __device__ __forceinline__ void load(const float* coef, int C, int ch0) {
    if (MODE == SRC_BNRELU || MODE == SRC_BNBWD) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = *reinterpret_cast<const f32x2*>(coef + ch0 + 2 * j);
        b[j] = *reinterpret_cast<const f32x2*>(coef + C + ch0 + 2 * j);
        if (MODE == SRC_BNBWD) c[j] = *reinterpret_cast<const f32x2*>(coef + 2 * C + ch0 + 2 * j);
      }
    }
  }
};

// Buffer resources (V#) of a source: loads through them are bounds-checked by the hardware (out-of-range -> 0), so the
// zero padding of a patch needs no exec-mask branch: an invalid piece just uses the out-of-range offset OOB_OFF.
constexpr uint32_t OOB_OFF = 0x7fffff00u;
struct SrcRsrc {
  __amdgpu_buffer_rsrc_t r0, r1;
  template <int MODE> __device__ __forceinline__ void init(const SrcDesc& s) {
    r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(s.p0), 0, 0x7fffff00, 0x00020000);
    if (MODE == SRC_BNBWD) r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(s.p1), 0, 0x7fffff00, 0x00020000);
  }
};

// Raw 16-byte loads of one 8-channel piece (second tensor only for BNBWD).
template <int MODE> struct RawPiece { u32x4 v0, v1; };

// byte_off = byte offset of the piece inside the tensor, or OOB_OFF for a piece outside the image
// AUX = cache-policy bits of the load (0 = default, 2 = nt: streamed once, do not keep in L2)
// `soff` = wave-uniform byte offset added by the hardware AFTER the range check of byte_off (an out-of-range byte_off stays out of
// range whatever soff is): a compile-time soff folds into the instruction's immediate, so stepping through channel chunks costs no
// vector instruction and no select on the validity of the piece
template <int MODE, int AUX = 0>
__device__ __forceinline__ void load_piece_b(const SrcRsrc& rs, uint32_t byte_off, RawPiece<MODE>& r, int soff = 0) {
  r.v0 = __builtin_amdgcn_raw_buffer_load_b128(rs.r0, byte_off, soff, AUX);
  if (MODE == SRC_BNBWD) r.v1 = __builtin_amdgcn_raw_buffer_load_b128(rs.r1, byte_off, soff, AUX);
}

// compatibility form (element offset + validity flag)
template <int MODE>
__device__ __forceinline__ void load_piece(const SrcDesc& s, size_t off, bool valid, RawPiece<MODE>& r) {
  SrcRsrc rs;
  rs.init<MODE>(s);
  load_piece_b<MODE>(rs, valid ? (uint32_t)(off * 2) : OOB_OFF, r);
}

// Apply the source transform to a raw piece; out-of-image pieces are exact zeros (zero padding applies to the
// transformed activation, not to the stored tensor).  20 VALU instructions per 8 elements for BN-apply+ReLU.
template <int MODE>
__device__ __forceinline__ uint4 transform_piece(const RawPiece<MODE>& r, bool valid, const ChanCoef<MODE>& cc) {
  uint4 o;
  if (MODE == SRC_RAW || MODE == SRC_RAWG) { o.x = r.v0[0]; o.y = r.v0[1]; o.z = r.v0[2]; o.w = r.v0[3]; return o; }
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (MODE == SRC_BNRELU) w[j] = relu2(pk2(up2(r.v0[j]) * cc.a[j] + cc.b[j]));
    else w[j] = pk2(up2(r.v0[j]) * cc.a[j] + (up2(r.v1[j]) * cc.b[j] + cc.c[j]));
  }
  const uint32_t m = valid ? 0xffffffffu : 0u;
  o.x = w[0] & m; o.y = w[1] & m; o.z = w[2] & m; o.w = w[3] & m;
  return o;
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

// Kernels WITHOUT matrix instructions that may be co-resident with the MFMA kernels (second stream, other processes' work)
// are built without packed-FP32 VALU instructions: on gfx950 / ROCm 7.2 code with v_pk_fma_f32 / v_pk_mul_f32 op_sel chains
// returned wrong low-half results in lanes 48-63 when an MFMA kernel shared the SIMD (tools/race_*.py:
// reproduced with a register-only victim; the same source built with this attribute is immune).
#define EAE_NO_PK __attribute__((target("no-packed-fp32-ops")))

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm finalize without a kernel of its own (every dependent launch costs ~6 us on this part).
//   producer: instead of one partial row per workgroup it adds its per-channel sums, converted to FIXED POINT, into one of
//             `copies` accumulator sets [copies][2][C] with 64-bit integer atomics (integer addition is associative: the result
//             does not depend on arrival order, so this stays bitwise reproducible; `copies` sets keep the contention low);
//   consumer: every workgroup of the NEXT kernel sums the copies for all C channels in its prologue (<= 64*2*C 8-byte loads
//             over 256 threads) and builds the coefficient table in LDS; workgroup 0 also stores the table for the kernels
//             that need it later and updates the running statistics.
// ---------------------------------------------------------------------------------------------------------------
struct BnAcc {
  unsigned long long* acc;   // [copies][2][C]; zero before the producer runs; nullptr: per-tile partials + finalize kernel
  int copies;                // power of two
  float scale;               // fixed-point scale
  unsigned long long* flag;  // [C] sticky "a non-finite partial of this channel was seen" words (cleared with the accumulators)
};
struct BnFold {
  const unsigned long long* acc;   // nullptr: the coefficients come from SrcDesc::coef
  int copies;
  float inv_scale, count, momentum, eps;
  const float* gamma; const float* beta;
  float* rm; float* rv; long long* nbt;
  float* coef_out;           // [4][C]: s, t, mean, invstd
  const unsigned long long* flag;  // [C] the layer's non-finite flags (BnAcc::flag)
  unsigned* poison;          // step-wide sticky word (cleared with the accumulators): the writer workgroup sets it when a channel is flagged
};
__device__ __forceinline__ void bn_acc_add(const BnAcc& b, int C, int tile_id, int which, int ch, float v) {
  unsigned long long* p = b.acc + ((size_t)(tile_id & (b.copies - 1)) * 2 + which) * C + ch;
  // a non-finite partial (diverged run) must poison the statistics like it does in floating point.  It cannot travel inside the
  // wrapping integer sum (round 2 added 2^61 per poisoned partial: eight of them -- every workgroup of a diverged step -- cancel
  // mod 2^64): it sets the channel's sticky flag word instead, and the consumers turn a set flag into NaN coefficients of that channel
  const float sv = v * b.scale;
  if (fabsf(sv) < 9.0e18f) __hip_atomic_fetch_add(p, (unsigned long long)llrintf(sv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (b.flag != nullptr) __hip_atomic_fetch_or(b.flag + ch, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Split in two so that the accumulator loads can be ISSUED before the kernel's own first loads (vector-memory results return in
// issue order: a prologue whose loads queue behind the patch loads would wait for all of them) and CONSUMED after those are in
// flight.  copies <= 4 * (256 / C)  (at most 4 accumulator sets per thread).
constexpr int BN_FOLD_K = 4;
constexpr int BN_FOLD_KB = 2;      // backward tables: the consumers hold two source tensors' raw pieces in registers meanwhile
template <int K> struct BnFoldRegsT { long long v[2][K]; unsigned long long flag; };
typedef BnFoldRegsT<BN_FOLD_K> BnFoldRegs;
typedef BnFoldRegsT<BN_FOLD_KB> BnFoldRegsB;
// `tid` = index of the calling thread among the 256 threads that build the table (default: the whole 256-thread workgroup)
template <int C, int K>
__device__ __forceinline__ void bn_fold_load_acc(const unsigned long long* acc, int copies, const unsigned long long* flag, BnFoldRegsT<K>& r, int tid) {
  static_assert(C <= 256 && 256 % C == 0, "channel count");
  constexpr int G = 256 / C;
  const int ch = tid % C, grp = tid / C;
  r.flag = flag != nullptr ? flag[ch] : 0ull;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int k = grp + j * G;
    const bool ok = k < copies;
    r.v[0][j] = ok ? (long long)acc[((size_t)k * 2 + 0) * C + ch] : 0;
    r.v[1][j] = ok ? (long long)acc[((size_t)k * 2 + 1) * C + ch] : 0;
  }
}
template <int C>
__device__ __forceinline__ void bn_fold_load(const BnFold& f, BnFoldRegs& r, int tid = threadIdx.x) {
  bn_fold_load_acc<C, BN_FOLD_K>(f.acc, f.copies, f.flag, r, tid);
}
// All 256 threads of the workgroup.  table = LDS float [4][C]; red = LDS long long [2][256] (may alias any idle buffer).
// (contains two workgroup barriers: every wave of the workgroup has to pass them, table builders or not)
template <int C>
__device__ __forceinline__ void bn_fold_fwd_finish(const BnFold& f, const BnFoldRegs& r, float* table, long long* red, bool writer,
                                                   int tid = threadIdx.x) {
  constexpr int G = 256 / C;
  const int ch = tid % C, grp = tid / C;
  long long s1 = 0, s2 = 0;
#pragma unroll
  for (int j = 0; j < BN_FOLD_K; ++j) { s1 += r.v[0][j]; s2 += r.v[1][j]; }
  red[tid] = s1; red[256 + tid] = s2;
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int g = 1; g < G; ++g) { s1 += red[g * C + ch]; s2 += red[256 + g * C + ch]; }
    const bool poisoned = r.flag != 0ull;
    const double a = poisoned ? (double)__builtin_nanf("") : (double)s1 * (double)f.inv_scale, b = (double)s2 * (double)f.inv_scale;
    const double mean = a / f.count;
    double var = b / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = 1.0f / sqrtf((float)var + f.eps);
    const float s = f.gamma[ch] * invstd, t = f.beta[ch] - (float)mean * s;
    table[ch] = s; table[C + ch] = t; table[2 * C + ch] = (float)mean; table[3 * C + ch] = invstd;
    // The NaN coefficients poison this channel's activations, but not reliably what follows: the packed-bf16 ReLU of the consumers
    // is an integer max, which turns a negative-signed NaN into 0 (measured: a NaN weight of enc.conv2 left every later layer and the
    // loss finite).  The step-wide word makes the step loud whatever the data does: loss_finalize and the optimizer kernel read it.
    if (writer && poisoned && f.poison != nullptr) __hip_atomic_fetch_or(f.poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (writer) {
      f.coef_out[ch] = s; f.coef_out[C + ch] = t; f.coef_out[2 * C + ch] = (float)mean; f.coef_out[3 * C + ch] = invstd;
      if (f.rm) {
        const double unb = f.count > 1.f ? var * (double)f.count / ((double)f.count - 1.0) : var;
        f.rm[ch] = (1.f - f.momentum) * f.rm[ch] + f.momentum * (float)mean;
        f.rv[ch] = (1.f - f.momentum) * f.rv[ch] + f.momentum * (float)unb;
      }
      if (f.nbt && ch == 0) *f.nbt += 1;
    }
  }
  __syncthreads();
}
template <int C>
__device__ __forceinline__ void bn_fold_fwd(const BnFold& f, float* table, long long* red, bool writer) {
  BnFoldRegs r;
  bn_fold_load<C>(f, r);
  bn_fold_fwd_finish<C>(f, r, table, red, writer);
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm BACKWARD finalize without a kernel of its own (the 7 finalize launches were 6.2 us each on the dependency chain),
// same scheme as the forward one: the producer (the kernel whose epilogue applies the ReLU mask and takes sum g, sum g*xhat)
// adds its workgroup sums in fixed point to [copies][2][C] integer accumulators; EVERY consumer workgroup (backward-data kernel
// on the main stream, weight-gradient kernel on a side stream) sums the copies in its prologue, behind its own first loads, and
// builds the table [3][C] = A, B, Cc of bn_bwd_finalize_kernel in LDS.  Workgroup 0 of the main-stream consumer also stores
// dgamma / dbeta (and the table, for the per-op entry points and the diagnostics).  Integer sums: independent of the arrival
// order, so bitwise reproducible.  2^42 fixed point: workgroup sums from 2.3e-13 to 2e6 in magnitude (beyond: NaN).
// (A "last workgroup of the producer finalizes" variant was measured first: waiting for the atomics' return values and drawing a
// ticket cost every producer workgroup two device-scope round trips at its end: 0.558 vs 0.5225 ms per step.)
// ---------------------------------------------------------------------------------------------------------------
struct BnBwdFold {
  const unsigned long long* acc;   // nullptr: the coefficients come from SrcDesc::coef (bn_bwd_finalize_kernel wrote them)
  int copies;
  float inv_scale, count;
  const float* gamma; const float* coef_fwd;       // [C], [4][C]
  float* dgamma; float* dbeta; float* coef_out;    // [C], [C], [3][C]: written by the `writer` workgroup (each may be nullptr)
  float* dbias;              // eval-mode backward only: gradient of the bias in FRONT of this BatchNorm, A[c] * sum g (train mode: zero)
  const unsigned long long* flag;  // [C] the layer's non-finite flags (BnAcc::flag of the backward accumulators)
  unsigned* poison;          // step-wide sticky word, as in BnFold
};
template <int C>
__device__ __forceinline__ void bn_fold_bwd_load(const BnBwdFold& f, BnFoldRegsB& r, int tid = threadIdx.x) {
  bn_fold_load_acc<C, BN_FOLD_KB>(f.acc, f.copies, f.flag, r, tid);     // copies <= BN_FOLD_KB * (256 / C)
}
// 256 threads (tid 0..255).  table = LDS float [3][C]; red = LDS long long [2][256] (may alias any idle buffer).  Two barriers.
template <int C>
__device__ __forceinline__ void bn_fold_bwd_finish(const BnBwdFold& f, const BnFoldRegsB& r, float* table, long long* red, bool writer,
                                                   int tid = threadIdx.x) {
  constexpr int G = 256 / C;
  const int ch = tid % C, grp = tid / C;
  long long s1 = 0, s2 = 0;
#pragma unroll
  for (int j = 0; j < BN_FOLD_KB; ++j) { s1 += r.v[0][j]; s2 += r.v[1][j]; }
  red[tid] = s1; red[256 + tid] = s2;
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int g = 1; g < G; ++g) { s1 += red[g * C + ch]; s2 += red[256 + g * C + ch]; }
    const bool poisoned = r.flag != 0ull;
    const float db = poisoned ? __builtin_nanf("") : (float)((double)s1 * (double)f.inv_scale);
    const float dg = (float)((double)s2 * (double)f.inv_scale);
    const float mean = f.coef_fwd[2 * C + ch], invstd = f.coef_fwd[3 * C + ch];
    const float A = f.gamma[ch] * invstd;
    const float Bc = -A * invstd * dg / f.count;
    const float Cc = -A * db / f.count - Bc * mean;
    table[ch] = A; table[C + ch] = Bc; table[2 * C + ch] = Cc;
    if (writer && poisoned && f.poison != nullptr) __hip_atomic_fetch_or(f.poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (writer) {
      if (f.dbeta) f.dbeta[ch] = db;
      if (f.dgamma) f.dgamma[ch] = dg;
      if (f.coef_out) { f.coef_out[ch] = A; f.coef_out[C + ch] = Bc; f.coef_out[2 * C + ch] = Cc; }
      if (f.dbias) f.dbias[ch] = A * db;
    }
  }
  __syncthreads();
}

// first thread of the grid publishes the stream's progress value (relaxed agent-scope store: bypasses the non-coherent caches).
// Every z-plane's first workgroup stores: in a grouped launch (eae_group.h) the planes belong to different members, each with its own
// word; the K-slices of a split-K launch store the same value to the same word.
__device__ __forceinline__ void eae_signal(unsigned* sig, unsigned val) {
  if (sig != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    __hip_atomic_store(sig, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same store by the workgroup the caller elects (kernels whose workgroup ids are remapped: the grouped twins' XCD-aware map)
__device__ __forceinline__ void eae_signal_first(unsigned* sig, unsigned val, bool first) {
  if (sig != nullptr && first && threadIdx.x == 0) __hip_atomic_store(sig, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define EAE_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return eae_set_error(-3, hipGetErrorString(e__)); } while (0)
