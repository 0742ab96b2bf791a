// Unified implicit-GEMM kernel for the 3x3 stride-2 convolution family on NHWC bf16 activations (gfx950).
//
//   KIND_CONV   : out[n,oy,ox,co] = sum_{ky,kx,ci} T(in)[n,2oy-1+ky,2ox-1+kx,ci] * W[co][ky,kx][ci]
//                 = forward of nn.Conv2d(3x3,s2,p1) (R.md:292-304) and backward-data of nn.ConvTranspose2d.
//   KIND_DECONV : out[n,2i+py,2j+px,co] = sum_{taps(py,px),ci} T(in)[n,i+dy,j+dx,ci] * W[co][ky,kx][ci]
//                 = forward of nn.ConvTranspose2d(3x3,s2,p1,op1) (R.md:370-382) and backward-data of nn.Conv2d; the four
//                   sub-pixel phases are computed from ONE staged input patch (no zero insertion, no atomics).
//
// Structure (one workgroup = 4 waves = P positions x BN output channels):
//   * the (transformed) input patch of the tile's K-chunk (32 channels) is staged ONCE in LDS -- BatchNorm-apply+ReLU or
//     BatchNorm-backward-apply happen once per input element while staging -- and the MFMA pixel operand is read
//     straight out of the patch (the im2col matrix is never materialised);
//   * the weight operand is NOT staged: each wave reads its own 16-channel fragment rows [cout][tap][32ch] (64 B
//     contiguous per lane group) directly from L1/L2, all taps of a chunk prefetched into registers while the patch
//     loads are in flight.  Waves split the output channels (one 16-column n-tile each), so no weight fragment is
//     fetched twice per workgroup, and the LDS footprint is just the patch -> 2-3 workgroups per CU overlap their
//     load / MFMA / epilogue phases;
//   * MFMA = v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand, so an accumulator lane holds 4 consecutive
//     output channels of one pixel: the epilogue packs them to 8 bytes -> LDS tile -> 16-byte coalesced NHWC stores;
//   * BatchNorm reductions (forward: sum y, sum y^2; backward: sum g, sum g*xhat) are taken from the values actually
//     stored, accumulated over all phases, and written as ONE deterministic partial per workgroup (no atomics).
#pragma once
#include "eae_common.hip.h"

struct ConvArgs {
  SrcDesc src;
  const bf16_t* wpack;     // [COUT][9][CIN] bf16 (tap = ky*3+kx)
  const float* bias;       // [COUT] (EPI_FWD) or nullptr
  bf16_t* out;             // NHWC bf16
  float* stat_part;        // [2][COUT][ntiles] (channel-major: the finalize kernels read one channel contiguously) or nullptr
  int ntiles;              // number of statistics partials per channel = workgroups along grid.x (set by the launcher)
  const bf16_t* yprev;     // EPI_MASK: raw pre-BN tensor at the output positions
  const float* prev_coef;  // EPI_MASK: [4][COUT] s,t,mean,invstd of that BN
  int B, Hin, Win;         // input spatial size (conv: out = Hin/2; deconv: out = 2*Hin)
  // Progress word of the caller's stream: the first thread of the grid stores `sig_val` there when the kernel STARTS (i.e. after
  // everything enqueued before it on its stream has completed).  Gate kernels on the engine's side streams poll that word, so a
  // hand-over to a side stream needs no event record on the dependency chain (each one cost it ~5 us of bubble).
  unsigned* sig;
  unsigned sig_val;
  BnAcc bacc;              // statistics go to fixed-point accumulators instead of stat_part (finalize folded into the consumer)
  BnFold fold;             // SRC_BNRELU: build the source layer's coefficient table from its accumulators
  BnBwdFold bfold;         // SRC_BNBWD: build the source layer's backward coefficient table from its accumulators
  // fp8 variant (igemm8_s2_kernel, BASELINE config 5): wpack = e4m3 bytes [COUT][9][CIN];  qs[0] = 1 / (scale of the pixel
  // operand) -- the fragments are converted bf16 -> e4m3 (activations) / e5m2 (gradients) with v_cvt_scalef32_pk_*_bf16, which
  // DIVIDES by its scale operand --, qs[1] = 1 / (pixel scale * weight scale), applied to the accumulators;  amax: the largest
  // |staged value| (bf16 bits << 16, atomicMax) for the next step's scale (delayed scaling, eae_fp8.hip)
  const float* qs;
  unsigned* amax;
  int amax_mask;          // amax is an array of amax_mask + 1 slots (a power of two): workgroup t reports into slot t & amax_mask
  int amax_stride;        // words between two slots (the engine: 32 = one 128-byte line per slot; per-op calls: 0 slots -> unused)
  // SRC_BNBWD only: the BatchNorm-backward-applied gradient dy = A*g + B*y + C, exactly as staged (bf16, the source tensor's NHWC
  // layout), is ALSO stored here by channel block 0 of every tile, so that the layer's weight-gradient kernel reads ONE plain
  // tensor instead of transforming g and y again (SRC_RAWG, eae_wgrad.hip.h).  nullptr: off
  bf16_t* dy_out;
#ifdef EAE_STAMPS
  unsigned long long* dbg; // diagnostic build only: s_memtime stamps of workgroup `dbg_block`, wave 0
  int dbg_block;
#endif
};

#ifdef EAE_STAMPS
#define EAE_STAMP(i) do { if (a.dbg && (int)blockIdx.x == a.dbg_block && blockIdx.y == 0 && threadIdx.x == 0) { \
    unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); a.dbg[i] = t__; } } while (0)
// start / end stamp (s_memrealtime: 100 MHz, one clock for the whole device) of EVERY workgroup (dbg[64 + 2*wg], +1) and the XCC/CU it ran on (dbg[64 + 2*8192 + wg])
#define EAE_STAMP_WG(k) do { if (a.dbg && blockIdx.y == 0 && threadIdx.x == 0 && blockIdx.x < 8192) { \
    unsigned long long t__; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); a.dbg[64 + 2 * blockIdx.x + (k)] = t__; \
    if ((k) == 0) { unsigned id__; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id__)); unsigned xcc__; \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc__)); a.dbg[64 + 2 * 8192 + blockIdx.x] = ((unsigned long long)xcc__ << 32) | id__; } } } while (0)
// stamp by the first lane of an arbitrary wave (igemm2: consumer wave 0 = thread 0, producer wave 4 = thread 256)
#define EAE_STAMP_T(i, t) do { if (a.dbg && (int)blockIdx.x == a.dbg_block && blockIdx.y == 0 && threadIdx.x == (t)) { \
    unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); a.dbg[i] = t__; } } while (0)
#else
#define EAE_STAMP(i) do {} while (0)
#define EAE_STAMP_WG(k) do {} while (0)
#define EAE_STAMP_T(i, t) do {} while (0)
#endif

enum { KIND_CONV = 0, KIND_DECONV = 1 };
constexpr int PIX_STRIDE = 40;       // edge / wgrad kernels: bf16 elements per staged pixel, 32 channels + 8 pad (80 B)
// igemm patch: 2-D image [img][row][PWS pixels][32 ch], 64 B per pixel and no padding bytes (a 16x8 conv tile then fits
// 4 workgroups per CU).  Inside every 256-byte group of 4 pixels of a row the pixel slot j and the 16-byte chunk c are permuted so
// that the 16-lane groups of a ds_read_b128 fragment read ({0-3,12-15,20-27}, ...: 16 positions x 4 k-groups) hit 16 different
// 16-byte bank slots.  Rows and images are multiples of 256 bytes apart, so which permutation works depends on how the 16
// positions of an m-tile spread over columns and rows -- one per geometry class, found by exhaustive search and checked for
// every (m-tile, tap) of every geometry by tools/lds_conflicts.py (4 LDS cycles per read = conflict-free everywhere; the
// round-2 layout, class 0 for every geometry, cost 8 on the 4x4 conv tiles and on every transposed tile, 12-16 on the 4x4
// transposed ones -- the deep layers were LDS-bound on those reads):
//   class 0  conv, 16-wide tiles      : j ^ r, c ^ r                      (r = column / 4)
//   class 1  conv, 8- / 4-wide tiles  : j ^ (row / 2), c ^ 2r             (an m-tile spans 2-4 rows; rows alias mod 256 B)
//   class 2  transposed, 16-wide tiles: j, c ^ 2r
//   class 3  transposed, 8- / 4-wide  : j, c ^ 2 row
// Classes 0 and 2 depend on the column only: a fragment address is (per-lane, per-kx base) + (compile-time row offset), the row
// part folds into the ds_read immediate.  Classes 1 and 3 depend on (lane row + compile-time row) mod 4 / mod 2: one lane base per
// residue (frag_lane's `var`).
template <int SWZ>
__device__ __forceinline__ int swz_px(int col, int kg, int row) {     // element offset of (column, 8-channel group) inside a patch row
  const int r = col >> 2, j = col & 3;
  int jj, cc;
  if (SWZ == 0) { jj = (j ^ r) & 3; cc = (kg ^ r) & 3; }
  else if (SWZ == 1) { jj = (j ^ (row >> 1)) & 3; cc = (kg ^ (2 * r)) & 3; }
  else if (SWZ == 2) { jj = j; cc = (kg ^ (2 * r)) & 3; }
  else { jj = j; cc = (kg ^ (2 * row)) & 3; }
  return ((col & ~3) | jj) * 32 + cc * 8;
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogue helper: rows of a bf16 LDS tile [rows][BN+8] -> global with 16-byte stores, accumulating the per-channel
// reductions of the values stored.  One instance per thread; thread (r0 = tid / CPR, c = tid % CPR) owns 8 channels.
// ---------------------------------------------------------------------------------------------------------------
template <int COUT, int BN, int EPI, bool VALU_STATS = true>
struct TileEpilogue {
  static constexpr int TS = BN + 8, CPR = BN / 8, RPP = 256 / CPR;
  f32x2 s1[4], s2[4], ps[4], pt[4], pmi[4], pi[4];
  int c, r0;
  __device__ __forceinline__ void begin(const ConvArgs& a, int n0) {
    c = threadIdx.x % CPR; r0 = threadIdx.x / CPR;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = (f32x2){0.f, 0.f}; s2[j] = (f32x2){0.f, 0.f}; }
    if (EPI == EPI_MASK) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int ch = n0 + c * 8 + 2 * j;
        ps[j] = *reinterpret_cast<const f32x2*>(a.prev_coef + ch);
        pt[j] = *reinterpret_cast<const f32x2*>(a.prev_coef + COUT + ch);
        f32x2 m = *reinterpret_cast<const f32x2*>(a.prev_coef + 2 * COUT + ch);
        pi[j] = *reinterpret_cast<const f32x2*>(a.prev_coef + 3 * COUT + ch);
        pmi[j] = -m * pi[j];                      // xhat = y*invstd - mean*invstd
      }
    }
  }
  // rowmap(row) -> element offset of that output pixel (channel 0) in the NHWC tensor, or -1 if the row is invalid
  template <class RowMap>
  __device__ __forceinline__ void rows(const ConvArgs& a, const bf16_t* tile, int n0, int nrows, RowMap rowmap) {
    for (int row = r0; row < nrows; row += RPP) {
      long off = rowmap(row);
      if (off < 0) continue;
      uint4 v = *reinterpret_cast<const uint4*>(tile + row * TS + c * 8);
      size_t g = (size_t)off + n0 + c * 8;
      if (EPI == EPI_FWD && VALU_STATS) {
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { f32x2 f = up2(w[j]); s1[j] += f; s2[j] += f * f; }
      } else if (EPI == EPI_MASK) {
        uint4 yv = *reinterpret_cast<const uint4*>(a.yprev + g);
        uint32_t w[4] = {v.x, v.y, v.z, v.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x2 y = up2(yw[j]);
          f32x2 act = y * ps[j] + pt[j];
          uint32_t m = (act.x > 0.f ? 0x0000ffffu : 0u) | (act.y > 0.f ? 0xffff0000u : 0u);
          w[j] &= m;                               // exact: masked values are bf16 values or zero
          f32x2 f = up2(w[j]);
          s1[j] += f;
          s2[j] += f * (y * pi[j] + pmi[j]);
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      *reinterpret_cast<uint4*>(a.out + g) = v;
    }
  }
  // EPI_MASK: the previous layer's raw tensor at this thread's rows of one pass, requested ahead of the pass (one exposed memory
  // latency per pass instead of one per row group: the row loop below loads, waits, masks and stores one row group at a time --
  // 8 serialised round trips in conv2's backward-data, 38 % of its workgroup lifetime by the stamps)
  template <int NIT, class RowMap>
  __device__ __forceinline__ void load_prev(const ConvArgs& a, int n0, int nrows, RowMap rowmap, uint4 (&yv)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = r0 + it * RPP;
      const long off = row < nrows ? rowmap(row) : -1;
      yv[it] = off >= 0 ? *reinterpret_cast<const uint4*>(a.yprev + (size_t)off + n0 + c * 8) : make_uint4(0, 0, 0, 0);
    }
  }
  template <int NIT, class RowMap>
  __device__ __forceinline__ void rows_pre(const ConvArgs& a, const bf16_t* tile, int n0, int nrows, RowMap rowmap, const uint4 (&yv)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = r0 + it * RPP;
      const long off = row < nrows ? rowmap(row) : -1;
      if (off < 0) continue;
      uint4 v = *reinterpret_cast<const uint4*>(tile + row * TS + c * 8);
      const size_t g = (size_t)off + n0 + c * 8;
      uint32_t w[4] = {v.x, v.y, v.z, v.w}, yw[4] = {yv[it].x, yv[it].y, yv[it].z, yv[it].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 y = up2(yw[j]);
        f32x2 act = y * ps[j] + pt[j];
        uint32_t m = (act.x > 0.f ? 0x0000ffffu : 0u) | (act.y > 0.f ? 0xffff0000u : 0u);
        w[j] &= m;
        f32x2 f = up2(w[j]);
        s1[j] += f;
        s2[j] += f * (y * pi[j] + pmi[j]);
      }
      *reinterpret_cast<uint4*>(a.out + g) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  // deterministic reduction over the RPP row-groups that share a channel chunk; red = [2][RPP][BN] floats of LDS
  __device__ __forceinline__ void end(const ConvArgs& a, float* red, int n0, int tile_id) {
    if (EPI == EPI_PLAIN || (a.stat_part == nullptr && a.bacc.acc == nullptr) || (EPI == EPI_FWD && !VALU_STATS)) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f32x2*>(red + (0 * RPP + r0) * BN + c * 8 + 2 * j) = s1[j];
      *reinterpret_cast<f32x2*>(red + (1 * RPP + r0) * BN + c * 8 + 2 * j) = s2[j];
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 2 * BN) {
      int which = tid / BN, ch = tid % BN;
      float acc = 0.f;
      for (int r = 0; r < RPP; ++r) acc += red[(which * RPP + r) * BN + ch];
      if (a.bacc.acc) bn_acc_add(a.bacc, COUT, tile_id, which, n0 + ch, acc);
      else a.stat_part[((size_t)which * COUT + n0 + ch) * a.ntiles + tile_id] = acc;
    }
  }
};

// backwards-compatible free function used by the edge and FC kernels (single-pass tiles)
template <int COUT, int BN, int EPI, class RowMap>
__device__ __forceinline__ void tile_epilogue(const ConvArgs& a, bf16_t* tile, float* red, int n0, int tile_id, int nrows,
                                              RowMap rowmap) {
  TileEpilogue<COUT, BN, EPI> e;
  e.begin(a, n0);
  if constexpr (EPI == EPI_MASK) {      // (every caller has 128-row tiles) all rows of the previous layer's tensor requested at once
    constexpr int RPP = 256 / (BN / 8), NIT = (128 + RPP - 1) / RPP;
    uint4 yv[NIT];
    e.template load_prev<NIT>(a, n0, nrows, rowmap, yv);
    e.template rows_pre<NIT>(a, tile, n0, nrows, rowmap, yv);
  } else {
    e.rows(a, tile, n0, nrows, rowmap);
  }
  e.end(a, red, n0, tile_id);
}

// ---------------------------------------------------------------------------------------------------------------
// tile geometry: NI images x TH x TW positions (conv: output pixels; deconv: input positions), P = NI*TH*TW
// ---------------------------------------------------------------------------------------------------------------
template <int KIND, int TW, int TH, int NI>
struct Geo {
  static constexpr int P = NI * TH * TW;
  static constexpr int PH = (KIND == KIND_CONV) ? 2 * TH + 1 : TH + 1;
  static constexpr int PW = (KIND == KIND_CONV) ? 2 * TW + 1 : TW + 1;
  static constexpr int NPIX = NI * PH * PW;
  static constexpr int PWS = (PW + 3) & ~3;                        // pixels of LDS space per patch row
  static constexpr int RS = PWS * 32;                              // elements per patch row
  static constexpr int MUL = (KIND == KIND_CONV) ? 2 : 1;          // patch pixels per tile position
  static constexpr int NKX = (KIND == KIND_CONV) ? 3 : 2;          // distinct column offsets
  static constexpr int RPM = (16 / TW) ? (16 / TW) : 1;            // tile rows per 16-position m-tile
  static constexpr int NPH = (KIND == KIND_CONV) ? 1 : 4;          // output phases
  static constexpr int NOFF = (KIND == KIND_CONV) ? 9 : 4;         // distinct patch offsets
#ifdef EAE_SWZ_OLD      // diagnostic builds: the round-2 layout for every geometry (tools/ab.py A/B)
  static constexpr int SWZ = 0;
#else
  static constexpr int SWZ = (KIND == KIND_CONV) ? (TW == 16 ? 0 : 1) : (TW == 16 ? 2 : 3);       // swizzle class (swz_px)
#endif
  static constexpr int NVAR = (SWZ == 1) ? 4 : (SWZ == 3) ? 2 : 1;                                   // lane bases per column offset
  __device__ static __forceinline__ int swz(int col, int kg, int row) { return swz_px<SWZ>(col, kg, row); }
  // residue class of a compile-time row offset `radd` (patch rows added to the lane's own row MUL * ty)
  __device__ static constexpr int var_of(int radd) { return (SWZ == 1) ? ((radd >> 1) & 3) : (SWZ == 3) ? (radd & 1) : 0; }
  // patch pixel of position m for offset 0
  // per-lane part: i = lane & 15 (position inside the m-tile), wbase = index of the wave's first m-tile, column offset kx,
  // var = residue class of the compile-time row offset the base will be used with (var_of)
  __device__ static __forceinline__ int frag_lane(int i, int wbase, int kx, int kg, int var = 0) {
    int img, ty, tx = i % TW;
    if (TH * TW <= 16) { img = wbase; ty = i / TW; }
    else { int row = wbase * RPM + i / TW; img = row / TH; ty = row % TH; }
    const int srow = (SWZ == 1) ? MUL * ty + 2 * var : MUL * ty + var;      // any row with the same residue as (lane row + offset)
    return (img * PH + MUL * ty) * RS + swz(MUL * tx + kx, kg, srow);
  }
  // tap (ky,kx) -> patch offset id / phase.  deconv: oy = 2*iy - 1 + ky  =>  ky=1: (py=0, dy=0); ky=2: (py=1, dy=0); ky=0: (py=1, dy=1)
  __device__ static constexpr int tap_off(int tap) {
    return (KIND == KIND_CONV) ? tap : ((tap / 3 == 0) ? 2 : 0) + ((tap % 3 == 0) ? 1 : 0);
  }
  __device__ static constexpr int tap_phase(int tap) {
    return (KIND == KIND_CONV) ? 0 : ((tap / 3 != 1) ? 2 : 0) + ((tap % 3 != 1) ? 1 : 0);
  }
  __device__ static constexpr int off_row(int o) { return (KIND == KIND_CONV) ? o / 3 : o >> 1; }   // row / column of offset id o
  __device__ static constexpr int off_col(int o) { return (KIND == KIND_CONV) ? o % 3 : o & 1; }
  // compile-time part of a fragment address: m-tile mi (inside the wave's MT m-tiles), offset id o
  __device__ static constexpr int frag_const(int mi, int o) {
    return (TH * TW <= 16) ? (mi * PH + off_row(o)) * RS
                           : (((mi * RPM) / TH) * PH + MUL * ((mi * RPM) % TH) + off_row(o)) * RS;
  }
  // patch rows (inside the image) that frag_const adds to the lane's own row: selects the lane base (var_of)
  __device__ static constexpr int frag_radd(int mi, int o) {
    return (TH * TW <= 16) ? off_row(o) : MUL * ((mi * RPM) % TH) + off_row(o);
  }
};

// waves per SIMD the register allocation aims at (= workgroups per CU): the 32<->64-channel layers launch 1024 workgroups
// at B=512, which only fit the 256 CUs in ONE round at 4 per CU (128 VGPRs, <= 40 KB LDS)
// (the 64->32 transposed kind keeps 4 phases x 4 m-tiles of accumulators: 3 per CU is what fits without spilling)
// A 64-position tile of the transposed kind (2 m-tiles x 4 phases of accumulators per wave) fits the budget of 4 per CU as well.
constexpr int ig_occ(int kind, int cin, int cout, int P = 128) { return cin > 64 ? 1 : (cin * cout > 2048 ? 2 : ((kind == 0 || P == 64) ? 4 : 3)); }

// fp8 helpers (lane maps of the fp8 MFMA forms = the bf16 form's, conversion semantics and saturation: tools/probe/probe_fp8.hip)
template <bool E5M2>
__device__ __forceinline__ long cvt8(const bf16x8& v, float inv_scale) {
  s16x2 lo = {0, 0}, hi = {0, 0};
  const bf16x2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]}, p2 = {v[4], v[5]}, p3 = {v[6], v[7]};
  if (E5M2) {
    lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(lo, p0, inv_scale, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(lo, p1, inv_scale, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(hi, p2, inv_scale, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(hi, p3, inv_scale, true);
  } else {
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, p0, inv_scale, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, p1, inv_scale, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, p2, inv_scale, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, p3, inv_scale, true);
  }
  union { s16x2 h[2]; long l; } u;
  u.h[0] = lo; u.h[1] = hi;
  return u.l;
}
// MODE.FP16_OVFL: out-of-range fp8 conversions clamp to +-max instead of producing NaN
__device__ __forceinline__ void fp8_saturate_mode() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
// running max of |bf16| over the dwords of a staged piece (two packed 16-bit lanes)
__device__ __forceinline__ uint32_t amax_pk(uint32_t run, const uint4& o) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 r = __builtin_bit_cast(u16x2, run);
  r = __builtin_elementwise_max(r, __builtin_bit_cast(u16x2, o.x & 0x7fff7fffu));
  r = __builtin_elementwise_max(r, __builtin_bit_cast(u16x2, o.y & 0x7fff7fffu));
  r = __builtin_elementwise_max(r, __builtin_bit_cast(u16x2, o.z & 0x7fff7fffu));
  r = __builtin_elementwise_max(r, __builtin_bit_cast(u16x2, o.w & 0x7fff7fffu));
  return __builtin_bit_cast(uint32_t, r);
}

// Q = 0: bf16 operands;  Q = 1: fp8 operands (weights e4m3 from an fp8 pack, pixel fragments converted in registers from the bf16
// patch: e4m3 for activations, e5m2 for SRC_BNBWD gradients), fp32 accumulation either way
template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI, int Q>
__device__ __forceinline__ void igemm_body(const ConvArgs& a, const int li_given = -1) {     // li_given >= 0: the workgroup's logical id (grouped twin, group_xcd_map)
  using G = Geo<KIND, TW, TH, NI>;
  static_assert(G::P == 64 || G::P == 128, "tile must hold 64 or 128 positions");
  static_assert(CIN % 32 == 0 && COUT % BN == 0 && (BN == 32 || BN == 64), "shape");
  constexpr int P = G::P, PH = G::PH, PW = G::PW, NPIX = G::NPIX, NPH = G::NPH;
  constexpr int WN = BN / 16, WM = 4 / WN;          // waves along N (one 16-col n-tile each) / along M
  constexpr int MT = (P / 16) / WM;                 // m-tiles (16 positions) per wave
  constexpr int NPA = (NPIX * 4 + 255) / 256;       // 16-byte patch pieces per thread
  constexpr int TS = BN + 8;
  constexpr bool ROWSWEEP = (TW == 16 && NI == 1);      // m-tile == one tile row: sweep the patch rows (see the MFMA loop)
  static_assert(Q == 0 || ROWSWEEP, "the fp8 variant is built for the 16-wide tiles only");
  constexpr bool QG = (SRC == SRC_BNBWD);               // fp8 variant: the pixel operand is a gradient -> e5m2
  if (Q) fp8_saturate_mode();
  const float q_inv = Q ? a.qs[0] : 1.f;
  float q_out = Q ? a.qs[1] : 1.f;
  asm volatile("" : "+v"(q_out));                 // its own register, not the high half of the (q_inv, q_out) load
  const f32x2 q_out2 = {q_out, q_out};
  uint32_t amax_run = 0;
  constexpr int PFB = ig_occ(KIND, CIN, COUT, G::P) >= 4 ? 1 : 2;   // pixel-fragment register buffers (double buffering costs 4*MT VGPRs)
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  bf16_t* patch = smem;                             // [NPIX][PIX_STRIDE]; reused as the output tile [P][TS] + reduction scratch

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int Hout = (KIND == KIND_CONV) ? a.Hin >> 1 : a.Hin * 2, Wout = (KIND == KIND_CONV) ? a.Win >> 1 : a.Win * 2;
  const int Hpos = (KIND == KIND_CONV) ? Hout : a.Hin, Wpos = (KIND == KIND_CONV) ? Wout : a.Win;   // position grid
  const int tiles_x = Wpos / TW, tiles_y = Hpos / TH;
  // Workgroup -> (tile, channel block).  The COUT/BN channel blocks of one tile read the same input patch: they are placed on
  // the SAME XCD (workgroup ids go round-robin over the 8 XCDs) and next to each other in dispatch order, so the second one
  // finds the patch in that XCD's L2 instead of fetching it again over the fabric.
  constexpr int NB = COUT / BN;
  int tile_id, nblk;
  {
    const int bid = blockIdx.x;
    if (li_given >= 0) { nblk = NB == 1 ? 0 : li_given % NB; tile_id = NB == 1 ? li_given : li_given / NB; }
    else if (NB == 1) { tile_id = bid; nblk = 0; }
    else if ((a.ntiles * NB) % 8 == 0) { const int li = (bid & 7) * ((a.ntiles * NB) >> 3) + (bid >> 3); nblk = li % NB; tile_id = li / NB; }   // contiguous run of (tile, block) pairs per XCD
    else { nblk = bid % NB; tile_id = bid / NB; }
  }
  int t = tile_id;
  const int txb = t % tiles_x; t /= tiles_x;
  const int tyb = t % tiles_y; t /= tiles_y;
  const int img0 = t * NI;
  const int n0 = nblk * BN;
  const int iy0 = (KIND == KIND_CONV) ? 2 * tyb * TH - 1 : tyb * TH, ix0 = (KIND == KIND_CONV) ? 2 * txb * TW - 1 : txb * TW;

  const bool first_wg = li_given >= 0 ? li_given == 0 : blockIdx.x == 0;      // elected workgroup: publishes the progress value, writes the folded tables
  eae_signal_first(a.sig, a.sig_val, first_wg);
  EAE_STAMP(0);
  EAE_STAMP_WG(0);
  f32x4 acc[NPH][MT];
#pragma unroll
  for (int i = 0; i < NPH; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int kgl = lane >> 4;        // k-group of the lane inside an MFMA (8 channels)
  static_assert(TH * TW <= 16 || (TH % (MT * G::RPM) == 0) || (WM == 1 && (MT * G::RPM) % TH == 0), "m-tile rows must not straddle images");
  int lbase[G::NKX][G::NVAR];       // swizzled fragment base of this lane per column offset (and row residue); rows are immediate offsets
#pragma unroll
  for (int kx = 0; kx < G::NKX; ++kx)
#pragma unroll
    for (int v = 0; v < G::NVAR; ++v) lbase[kx][v] = G::frag_lane(lane & 15, wm * MT, kx, kgl, v);
  const int kgs = tid & 3;          // k-group staged by this thread (256 % 4 == 0 -> fixed per thread)
  // this lane's weight-fragment row: output channel n0 + wn*16 + (lane&15), 8 input channels kgl*8..
  const bf16_t* wrow = a.wpack + (size_t)(n0 + wn * 16 + (lane & 15)) * 9 * CIN + kgl * 8;
  const uint8_t* wrow8 = reinterpret_cast<const uint8_t*>(a.wpack) + (size_t)(n0 + wn * 16 + (lane & 15)) * 9 * CIN + kgl * 8;
  SrcRsrc rs;
  rs.init<SRC>(a.src);
  // The 32 <-> 64-channel instances run at 3-4 workgroups per CU on 128-168 registers with 72 of them holding raw pieces: keeping the
  // piece offsets alive until the staging loop cost them 100-290 bytes of scratch (dec.deconv3's backward-data: 35 -> 53 us in the
  // step).  They are built without the store; the engine never asks them for dy (eae_api.hip: dy_mask).
  constexpr bool DY = SRC == SRC_BNBWD && CIN * COUT > 2048;
  const bool wr_dy = DY && a.dy_out != nullptr && nblk == 0;
  // (a workgroup that does not store dy -- no dy_out, or not channel block 0 -- gets a descriptor of ZERO bytes: its stores are all
  //  out of range and dropped by the hardware, so the staging loop needs no branch around them; a branch per piece cut the loop into
  //  basic blocks and cost the 128-register instances scratch)
  __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(DY ? a.dy_out : (bf16_t*)nullptr, 0, wr_dy ? 0x7fffff00 : 0, 0x00020000);

  // ---- patch pieces of this thread: piece q = tid + 256*i covers patch pixel q/4, channels kgs*8.. of the current
  //      K-chunk; consecutive i advance the pixel by 64 -> (img, row, col) are updated incrementally.  Byte offsets are
  //      32-bit; out-of-image pieces use the hardware-checked out-of-range offset (no branches); both are chunk-independent.
  constexpr int NC = CIN / 32;
  uint32_t boff[NPA];
  int loff[NPA];                    // LDS element offset of the piece
  bool val[NPA];
  {
    constexpr int DR = (64 / PW) % PH, DC = 64 % PW, DI = 64 / (PH * PW);
    int pix = tid >> 2;
    int img = pix / (PH * PW), rem = pix % (PH * PW);
    int pr = rem / PW, pc = rem % PW;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int iy = iy0 + pr, ix = ix0 + pc, n = img0 + img;
      val[i] = (tid + i * 256 < NPIX * 4) && (n < a.B) && ((unsigned)iy < (unsigned)a.Hin) && ((unsigned)ix < (unsigned)a.Win);
      boff[i] = val[i] ? ((uint32_t)((n * a.Hin + iy) * a.Win + ix) * CIN + kgs * 8) * 2u : OOB_OFF;
      loff[i] = (img * PH + pr) * G::RS + G::swz(pc, kgs, pr);
      pc += DC; pr += DR; img += DI;
      if (pc >= PW) { pc -= PW; pr += 1; }
      if (pr >= PH) { pr -= PH; img += 1; }
    }
  }
  // Software pipeline over the K-chunks: the raw patch pieces of chunk c+1 are requested in slices between the MFMA groups
  // of chunk c (register double-buffering), so their latency hides behind the matrix pipe; the 9 weight fragments of chunk
  // c+1 are requested right after chunk c's last MFMA (into the same registers) and arrive while the patch is being staged.
  bf16x8 wf[9];
  long wq[9];
  RawPiece<SRC> raw[NPA];
  ChanCoef<SRC> cc;
  // prefetch requests are issued in NOFF slices, one after each offset's MFMA group, so that the vector-memory pipe
  // (the 64 B/clk L1 path is the scarce resource of the multi-chunk layers) drains while the matrix pipe works
  __shared__ float coef_tab[(SRC == SRC_BNRELU) ? 4 * CIN : (SRC == SRC_BNBWD) ? 3 * CIN : 4];
  const float* coefp = a.src.coef;
  auto issue_slice = [&](int chunk, int o, bool with_coef = true) {
    if (o == 0 && with_coef) cc.load(coefp, CIN, chunk * 32 + kgs * 8);
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      if (i % G::NOFF == o) load_piece_b<SRC>(rs, boff[i], raw[i], chunk * 64);      // (boff is OOB_OFF for a piece outside the image)
  };
  auto issue = [&](int chunk, bool with_coef) {
#pragma unroll
    for (int o = 0; o < G::NOFF; ++o) issue_slice(chunk, o, with_coef);
  };
  auto load_w = [&](int chunk) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (Q) wq[tap] = *reinterpret_cast<const long*>(wrow8 + tap * CIN + chunk * 32);
      else wf[tap] = *reinterpret_cast<const bf16x8*>(wrow + tap * CIN + chunk * 32);
    }
  };
  // accumulator loads first, then the first chunk's weight / patch loads: the table is built while those are in flight
  BnFoldRegs fr;
  BnFoldRegsB frb;
  const bool folded = SRC == SRC_BNRELU && a.fold.acc != nullptr;
  const bool folded_b = SRC == SRC_BNBWD && a.bfold.acc != nullptr;
  // The 32-channel conv kind with two source tensors (dec.deconv3 backward-data) has 72 registers of raw pieces in flight at 4
  // waves per SIMD: building the table underneath them spilled pieces to scratch (a reload waits for every load issued before it:
  // 10-15 us per workgroup, measured with the stamps).  That instance builds its table BEFORE it requests the patch.
  constexpr bool EARLY_TABLE = SRC == SRC_BNBWD && KIND == KIND_CONV && CIN == 32;
  if (folded) bn_fold_load<CIN>(a.fold, fr);
  if (folded_b) bn_fold_bwd_load<CIN>(a.bfold, frb);
  if (EARLY_TABLE && folded_b) {
    bn_fold_bwd_finish<CIN>(a.bfold, frb, coef_tab, reinterpret_cast<long long*>(smem), first_wg);
    coefp = coef_tab;
  }
  if (SRC != SRC_BNBWD) load_w(0);
  issue(0, false);
  if (folded) {
    bn_fold_fwd_finish<CIN>(a.fold, fr, coef_tab, reinterpret_cast<long long*>(smem), first_wg);
    coefp = coef_tab;
  }
  if (!EARLY_TABLE && folded_b) {
    bn_fold_bwd_finish<CIN>(a.bfold, frb, coef_tab, reinterpret_cast<long long*>(smem), first_wg);
    coefp = coef_tab;
  }
  // two source tensors' raw pieces are in registers while the table is built: the weight fragments (L2-resident, needed only at
  // the first MFMA) are requested behind it, or the builder's temporaries spill
  if (SRC == SRC_BNBWD) load_w(0);
  cc.load(coefp, CIN, kgs * 8);
#pragma unroll
  for (int chunk = 0; chunk < NC; ++chunk) {
    EAE_STAMP(8 + chunk * 4 + 0);
    // ---- stage the input patch of this chunk (transform applied once per element)
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int q = tid + i * 256;
      if (q < NPIX * 4) {
        const uint4 o = transform_piece<SRC>(raw[i], val[i], cc);
        if (Q) amax_run = amax_pk(amax_run, o);
        *reinterpret_cast<uint4*>(patch + loff[i]) = o;
        // (Halo pieces are stored too: every workgroup that stages an element computes the same bits from the
        //  same g, y and table, so the duplicate stores -- 10-25 % of a map -- are harmless, and the store reuses the load's offset:
        //  a mask of owned pieces cost the 128-register instances 100-280 bytes of scratch)
        if (DY)
          // (the chunk offset goes into the VECTOR offset, never into soffset: for a 16-byte store whose soffset is an SGPR -- 128 and
          //  192 are no inline constants -- hipcc (ROCm 7.2) omits the wait state between the store and a following VALU write of its
          //  data registers, and on gfx950 the third dword of sporadic pieces of chunks 2 and 3 then came out as the NEXT piece's
          //  intermediate: tools/debug_det.py found the stored dy differing from run to run in exactly those dwords)
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){o.x, o.y, o.z, o.w}, rs_dy, boff[i] + chunk * 64, 0, 0);
      }
    }
    EAE_STAMP(8 + chunk * 4 + 1);
    __syncthreads();
    EAE_STAMP(8 + chunk * 4 + 2);
    EAE_STAMP(8 + chunk * 4 + 3);
    // ---- MFMAs: for every distinct patch offset, read the pixel fragments once and feed all taps that use it.  The
    //      fragments of offset o+1 are requested before the MFMAs of offset o (register double buffer), so the LDS latency
    //      hides behind the matrix pipe even with a single wave per SIMD.
    if constexpr (ROWSWEEP) {
      // 16-wide tiles: an m-tile is one tile row, so the fragment of patch row R, column offset kx serves every (m-tile, ky)
      // with MUL*mi + ky == R.  Sweeping the patch rows reads each fragment ONCE (conv: 51 instead of 72 ds_read_b128 per
      // chunk, deconv: 18 instead of 32) and needs only 2 x NKX fragment registers (row R+1 is requested before row R's MFMAs).
      constexpr int NROWS = G::MUL * MT + 1;
      bf16x8 rf[2][G::NKX];
#pragma unroll
      for (int kx = 0; kx < G::NKX; ++kx) rf[0][kx] = *reinterpret_cast<const bf16x8*>(patch + lbase[kx][0]);
#pragma unroll
      for (int R = 0; R < NROWS; ++R) {
        if (R + 1 < NROWS) {
#pragma unroll
          for (int kx = 0; kx < G::NKX; ++kx) rf[(R + 1) & 1][kx] = *reinterpret_cast<const bf16x8*>(patch + lbase[kx][0] + (R + 1) * G::RS);
        }
        // pin the order "fragments of row R+1 requested, then the MFMAs of row R": left alone, hipcc sinks the ds_read_b128 next to
        // their MFMAs (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma: a full LDS round trip per MFMA, seen in the ISA of several instances)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cx = 0; cx < G::NKX; ++cx) {
          long rq = 0;
          if (Q) rq = cvt8<QG>(rf[R & 1][cx], q_inv);       // 4 conversions per fragment, each fragment feeds 2-4 MFMAs
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            constexpr int dummy = 0; (void)dummy;
            const int o = G::tap_off(tap), ry = G::off_row(o);
            if (G::off_col(o) != cx || R < ry || (R - ry) % G::MUL != 0 || (R - ry) / G::MUL >= MT) continue;
            const int mi = (R - ry) / G::MUL;
            f32x4& d = acc[G::tap_phase(tap)][mi];
            if (Q) d = QG ? __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(wq[tap], rq, d, 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wq[tap], rq, d, 0, 0, 0);
            else d = mfma16(wf[tap], rf[R & 1][cx], d);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (chunk + 1 < NC && R < G::NOFF) issue_slice(chunk + 1, R);
      }
    } else {
    bf16x8 pf[PFB][MT];
  #pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        pf[0][mi] = *reinterpret_cast<const bf16x8*>(patch + lbase[G::off_col(0)][G::var_of(G::frag_radd(mi, 0))] + G::frag_const(mi, 0));
  #pragma unroll
      for (int o = 0; o < G::NOFF; ++o) {
        if (PFB == 2 && o + 1 < G::NOFF) {
  #pragma unroll
          for (int mi = 0; mi < MT; ++mi)
            pf[(o + 1) % PFB][mi] = *reinterpret_cast<const bf16x8*>(patch + lbase[G::off_col(o + 1)][G::var_of(G::frag_radd(mi, o + 1))] + G::frag_const(mi, o + 1));
        }
        if (PFB == 2) __builtin_amdgcn_sched_barrier(0);      // fragments of offset o+1 requested BEFORE the MFMAs of offset o (see above)
  #pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (G::tap_off(tap) != o) continue;
  #pragma unroll
          for (int mi = 0; mi < MT; ++mi) acc[G::tap_phase(tap)][mi] = mfma16(wf[tap], pf[o % PFB][mi], acc[G::tap_phase(tap)][mi]);
        }
        if (PFB == 2) __builtin_amdgcn_sched_barrier(0);
        if (PFB == 1 && o + 1 < G::NOFF) {
  #pragma unroll
          for (int mi = 0; mi < MT; ++mi)
            pf[0][mi] = *reinterpret_cast<const bf16x8*>(patch + lbase[G::off_col(o + 1)][G::var_of(G::frag_radd(mi, o + 1))] + G::frag_const(mi, o + 1));
        }
        if (chunk + 1 < NC) issue_slice(chunk + 1, o);
      }
    }
    if (chunk + 1 < NC) { load_w(chunk + 1); __syncthreads(); }
  }
  // ---- epilogue: per phase, accumulators (lane = pixel, 4 consecutive channels in regs) -> LDS tile -> global
  EAE_STAMP(4);
  __syncthreads();
  EAE_STAMP(5);
  // The transposed kind handles its two horizontal phases (px = 0, 1) in ONE pass: tile row 2*m + px, so consecutive rows are
  // horizontally adjacent output pixels and the stores (and the reads of the previous layer's tensor in the mask epilogue)
  // cover whole 128-byte lines instead of every other 64 bytes (conv2 backward-data: 144 -> ~105 MB of HBM traffic per launch).
  constexpr int PHG = (KIND == KIND_CONV) ? 1 : 2, R2 = P * PHG;
  bf16_t* tile = smem;
  float* red = reinterpret_cast<float*>(smem);     // aliases the tile: TileEpilogue::end() starts with a barrier after the last rows() pass
  TileEpilogue<COUT, BN, EPI, false> epi;
  epi.begin(a, n0);
  auto rowmap_of = [=](int pass) {
    return [=](int row2) -> long {
      const int row = row2 / PHG, px = row2 % PHG;
      int img = row / (TH * TW), ty = (row / TW) % TH, tx = row % TW;
      int n = img0 + img;
      if (n >= a.B) return -1;
      int oy = (KIND == KIND_CONV) ? tyb * TH + ty : 2 * (tyb * TH + ty) + pass;
      int ox = (KIND == KIND_CONV) ? txb * TW + tx : 2 * (txb * TW + tx) + px;
      return (((long)n * Hout + oy) * Wout + ox) * COUT;
    };
  };
  constexpr int NIT = (R2 + 256 / (BN / 8) - 1) / (256 / (BN / 8));
  uint4 yv[(EPI == EPI_MASK) ? 2 : 1][(EPI == EPI_MASK) ? NIT : 1];
  if constexpr (EPI == EPI_MASK) epi.template load_prev<NIT>(a, n0, R2, rowmap_of(0), yv[0]);
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == EPI_FWD) bv = *reinterpret_cast<const float4*>(a.bias + n0 + wn * 16 + kgl * 4);
  const int B = a.B;
  // BatchNorm batch statistics of the forward epilogue run on the matrix cores: with Y = the stored bf16 tile [P][16],
  //   sum_p Y[p][j]   = (ones^T . Y)[.][j]         (any row of an MFMA with an all-ones A operand)
  //   sum_p Y[p][j]^2 = diag(Y^T . Y)[j]           (A and B are the same transposed-read fragment)
  // wave w < BN/16 owns the 16 channels of n-tile w and sweeps all P rows of every phase.
  const bool do_stats = (EPI == EPI_FWD) && (a.stat_part != nullptr || a.bacc.acc != nullptr) && wave < BN / 16;
  f32x4 st1 = (f32x4){0.f, 0.f, 0.f, 0.f}, st2 = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
  for (int pass = 0; pass < NPH / PHG; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int px = 0; px < PHG; ++px) {
      const int ph = pass * PHG + px;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        int row = (wm * MT + mi) * 16 + (lane & 15);
        uint2 w2;
        if (Q) {      // (the scale as an explicit register pair: a packed multiply whose LOW lane reads the HIGH half of a
                      //  (q_inv, q_out) pair is the op_sel form tests/test_isa_guard.py forbids)
          w2.x = pk2((f32x2){acc[ph][mi][0], acc[ph][mi][1]} * q_out2 + (f32x2){bv.x, bv.y});
          w2.y = pk2((f32x2){acc[ph][mi][2], acc[ph][mi][3]} * q_out2 + (f32x2){bv.z, bv.w});
        } else {
        w2.x = pk2((f32x2){acc[ph][mi][0] + bv.x, acc[ph][mi][1] + bv.y});
        w2.y = pk2((f32x2){acc[ph][mi][2] + bv.z, acc[ph][mi][3] + bv.w});
        }
        if (NI > 1 && img0 + row / (TH * TW) >= B) w2 = make_uint2(0, 0);   // images past the batch must not enter the statistics
        *reinterpret_cast<uint2*>(tile + (row * PHG + px) * TS + wn * 16 + kgl * 4) = w2;
      }
    }
    EAE_STAMP(40 + pass * 4 + 0);
    __syncthreads();
    EAE_STAMP(40 + pass * 4 + 1);
    if (do_stats) {
#pragma unroll
      for (int ks = 0; ks < R2 / 32; ++ks) {
        const bf16_t* lo = tile + (ks * 32 + 8 * tg + tq) * TS + wave * 16 + 4 * tp;
        bf16x8 fr = tr_frag(lo, lo + 4 * TS);
        st1 = mfma16(ones, fr, st1);
        st2 = mfma16(fr, fr, st2);
      }
    }
    EAE_STAMP(40 + pass * 4 + 2);
    if constexpr (EPI == EPI_MASK) {
      if (pass + 1 < NPH / PHG) epi.template load_prev<NIT>(a, n0, R2, rowmap_of(pass + 1), yv[(pass + 1) & 1]);
      epi.template rows_pre<NIT>(a, tile, n0, R2, rowmap_of(pass), yv[pass & 1]);
    } else {
      epi.rows(a, tile, n0, R2, rowmap_of(pass));
    }
    EAE_STAMP(40 + pass * 4 + 3);
  }
  EAE_STAMP(6);
  if (do_stats) {
    // st1: every accumulator row holds the column sums -> lanes 0..15 (row group 0, register 0)
    // st2: the diagonal element of column j sits in lane 16*(j>>2) + j, register j&3
    const float d2 = tp == 0 ? st2[0] : tp == 1 ? st2[1] : tp == 2 ? st2[2] : st2[3];
    if (a.bacc.acc) {
      if (tg == 0) bn_acc_add(a.bacc, COUT, tile_id, 0, n0 + wave * 16 + (lane & 15), st1[0]);
      if (tg == tq) bn_acc_add(a.bacc, COUT, tile_id, 1, n0 + wave * 16 + (lane & 15), d2);
    } else {
      float* sp = a.stat_part + (size_t)(n0 + wave * 16 + (lane & 15)) * a.ntiles + tile_id;
      if (tg == 0) sp[0] = st1[0];
      if (tg == tq) sp[(size_t)COUT * a.ntiles] = d2;
    }
  }
  epi.end(a, red, n0, tile_id);
  if (Q && a.amax != nullptr && nblk == 0) {      // the channel blocks of a tile stage the same patch: one of them reports
    const uint32_t m16 = (amax_run & 0xffffu) > (amax_run >> 16) ? (amax_run & 0xffffu) : (amax_run >> 16);
    uint32_t wmax = m16;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = __shfl_xor(wmax, sh); wmax = o > wmax ? o : wmax; }
    // agent-scope atomics on ONE word serialise at the memory side (~50 ns each: 4096 workgroups of a 256x256 batch made this kernel
    // 4x slower than its bf16 twin, profiles/r03_bench_c5fp8_kernel_stats.csv before the change); the engine hands in 64 words
    if (lane == 0 && wmax != 0) atomicMax(a.amax + (size_t)(tile_id & a.amax_mask) * a.amax_stride, wmax << 16);
  }
  EAE_STAMP(7);
  EAE_STAMP_WG(1);
}

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
__global__ __launch_bounds__(256, ig_occ(KIND, CIN, COUT, NI * TH * TW)) void igemm_s2_kernel(ConvArgs a) {
  igemm_body<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI, 0>(a);
}
// grouped twin (eae_group.h): workgroup z runs the body with member z's arguments
template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
__global__ __launch_bounds__(256, ig_occ(KIND, CIN, COUT, NI * TH * TW)) void igemm_s2_kernel_g(GroupPack<ConvArgs> p, int gz) {
  unsigned member; int li;
  group_xcd_map(member, li);           // a member's tiles on one XCD: its weights and the channel blocks' shared patches stay in that L2
  igemm_body<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI, 0>(group_args_of<ConvArgs>(member), li);
}
template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI>
__global__ __launch_bounds__(256, ig_occ(KIND, CIN, COUT, NI * TH * TW)) void igemm8_s2_kernel(ConvArgs a) {
  igemm_body<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI, 1>(a);
}

template <int KIND, int BN, int TW, int TH, int NI>
constexpr size_t igemm_smem() {
  using G = Geo<KIND, TW, TH, NI>;
  constexpr size_t patch = (size_t)NI * G::PH * G::RS * 2;
  constexpr size_t tile_b = (size_t)G::P * (KIND == KIND_CONV ? 1 : 2) * (BN + 8) * 2, red_b = (size_t)2 * (256 / (BN / 8)) * BN * 4;
  constexpr size_t tile = tile_b > red_b ? tile_b : red_b;       // the reduction scratch reuses the tile region
  return patch > tile ? patch : tile;
}
