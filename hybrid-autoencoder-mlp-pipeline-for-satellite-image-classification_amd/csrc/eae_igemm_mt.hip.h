// Multi-tile variant of the implicit-GEMM kernel (eae_igemm.hip.h) for the BIG maps: 16-wide single-image tiles (16 x 8 or 16 x 4
// positions), i.e. the 32 <-> 64-channel layers at 64 x 64 inputs and every 3x3 layer at 256 x 256.
//
// Why.  With one tile per workgroup and every workgroup resident from the start, a launch runs its phases in lockstep over the whole
// chip: all workgroups request their patches at once (tools/kstamp.py, B=512: conv2 forward waits 7 of its 11 us of workgroup lifetime
// for 33 MB), then all multiply, then all store -- the read side of the memory system idles while the write side works and vice versa,
// and the matrix pipe idles through both.  Here a workgroup owns SEVERAL tiles (virtual block ids blockIdx.x, + gridDim.x, ...; the
// launcher keeps gridDim.x a multiple of 8 * NB, so all of them sit in the same XCD run and have the same channel block) and runs a
// flat software pipeline over its (tile, K-chunk) steps with TWO LDS buffers:
//
//     step s:   MFMA(s) out of buffer s & 1                          raw pieces of step s+1 in flight (requested in step s-1)
//               stage(s+1) into buffer (s+1) & 1                     (load transform once per element; waits for those pieces)
//               request the raw pieces of step s+2                   (they have an epilogue, a barrier and an MFMA phase to arrive)
//               last chunk of a tile: epilogue of that tile          (LDS tile in buffer s & 1 -> 16-byte stores, statistics)
//               barrier
//
// The stage of the next step sits BEFORE the epilogue's stores on purpose: hipcc waits vmcnt(0) wherever a loop-carried load is
// consumed (tools/waitcnt_audit.py), and a stage placed behind the stores -- the first version of this kernel, one LDS buffer -- drained
// the stores it had just issued: slower than one tile per workgroup.  In this order a drain only ever waits for loads that have had a
// whole step to arrive and for stores that are a whole step old.
#pragma once
#include "eae_igemm.hip.h"

template <int KIND, int BN, int TW, int TH>
constexpr size_t igemm_mt_smem() { return 2 * igemm_smem<KIND, BN, TW, TH, 1>(); }

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int SRC, int EPI, int Q>
__device__ __forceinline__ void igemm_mt_body(const ConvArgs& a) {
  constexpr int NI = 1;
  using G = Geo<KIND, TW, TH, NI>;
  static_assert(TW == 16, "the multi-tile kernel is built for the 16-wide single-image tiles");
  static_assert(G::P == 64 || G::P == 128, "tile must hold 64 or 128 positions");
  static_assert(CIN % 32 == 0 && COUT % BN == 0 && (BN == 32 || BN == 64), "shape");
  constexpr int P = G::P, PH = G::PH, PW = G::PW, NPIX = G::NPIX, NPH = G::NPH;
  constexpr int WN = BN / 16, WM = 4 / WN;          // waves along N (one 16-col n-tile each) / along M
  constexpr int MT = (P / 16) / WM;                 // m-tiles (16 positions) per wave
  constexpr int NPA = (NPIX * 4 + 255) / 256;       // 16-byte patch pieces per thread
  constexpr int TS = BN + 8;
  constexpr int NC = CIN / 32;
  constexpr int BUF_ELEMS = (int)(igemm_smem<KIND, BN, TW, TH, 1>() / 2);
  constexpr bool QG = (SRC == SRC_BNBWD);               // fp8 variant: the pixel operand is a gradient -> e5m2
  if (Q) fp8_saturate_mode();
  const float q_inv = Q ? a.qs[0] : 1.f;
  float q_out = Q ? a.qs[1] : 1.f;
  asm volatile("" : "+v"(q_out));                 // its own register, not the high half of the (q_inv, q_out) load
  const f32x2 q_out2 = {q_out, q_out};
  uint32_t amax_run = 0;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int Hout = (KIND == KIND_CONV) ? a.Hin >> 1 : a.Hin * 2, Wout = (KIND == KIND_CONV) ? a.Win >> 1 : a.Win * 2;
  const int Hpos = (KIND == KIND_CONV) ? Hout : a.Hin, Wpos = (KIND == KIND_CONV) ? Wout : a.Win;   // position grid
  const int tiles_x = Wpos / TW, tiles_y = Hpos / TH;
  constexpr int NB = COUT / BN;
  const int nvb = a.ntiles * NB;                    // (tile, channel block) pairs of the launch
  const bool xcd_map = NB > 1 && nvb % 8 == 0;
  const int gstep = (int)gridDim.x;
  struct TileAt { int tile_id, txb, tyb, img0; };
  int nblk;
  auto tile_of = [&](int vb, int& blk) -> TileAt {
    int tile;
    if (NB == 1) { tile = vb; blk = 0; }
    else if (xcd_map) { const int li = (vb & 7) * (nvb >> 3) + (vb >> 3); blk = li % NB; tile = li / NB; }   // contiguous run of (tile, block) pairs per XCD
    else { blk = vb % NB; tile = vb / NB; }
    TileAt r;
    r.tile_id = tile;
    int t = tile;
    r.txb = t % tiles_x; t /= tiles_x;
    r.tyb = t % tiles_y; t /= tiles_y;
    r.img0 = t;
    return r;
  };
  TileAt cur = tile_of((int)blockIdx.x, nblk);
  const int n0 = nblk * BN;

  eae_signal(a.sig, a.sig_val);
  f32x4 acc[NPH][MT];
  const int kgl = lane >> 4;        // k-group of the lane inside an MFMA (8 channels)
  int lbase[G::NKX];                // swizzled fragment base of this lane per column offset; rows are immediate offsets (classes 0 / 2)
  static_assert(G::NVAR == 1, "16-wide tiles use the column-only swizzle classes");
#pragma unroll
  for (int kx = 0; kx < G::NKX; ++kx) lbase[kx] = G::frag_lane(lane & 15, wm * MT, kx, kgl, 0);
  const int kgs = tid & 3;          // k-group staged by this thread
  const bf16_t* wrow = a.wpack + (size_t)(n0 + wn * 16 + (lane & 15)) * 9 * CIN + kgl * 8;
  const uint8_t* wrow8 = reinterpret_cast<const uint8_t*>(a.wpack) + (size_t)(n0 + wn * 16 + (lane & 15)) * 9 * CIN + kgl * 8;
  SrcRsrc rs;
  rs.init<SRC>(a.src);
  constexpr bool DY = SRC == SRC_BNBWD && CIN * COUT > 2048;      // (see igemm_body)
  const bool wr_dy = DY && a.dy_out != nullptr && nblk == 0;
  __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(DY ? a.dy_out : (bf16_t*)nullptr, 0, wr_dy ? 0x7fffff00 : 0, 0x00020000);

  // Piece -> (row, column) of the patch is a property of the THREAD: byte offset relative to the tile's first patch pixel, LDS offset,
  // and three bit masks over the pieces (inside the patch / on the halo row / on the halo column that falls outside the tensor for
  // tiles at its border: the row above and the column left of a conv patch, the row below and the column right of a transposed one).
  uint32_t relb[NPA];
  int loff[NPA];
  uint32_t m_in = 0, m_hr = 0, m_hc = 0;
  {
    constexpr int DR = (64 / PW) % PH, DC = 64 % PW;
    int pix = tid >> 2;
    int pr = pix / PW, pc = pix % PW;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      if (tid + i * 256 < NPIX * 4) m_in |= 1u << i;
      if (pr == ((KIND == KIND_CONV) ? 0 : PH - 1)) m_hr |= 1u << i;
      if (pc == ((KIND == KIND_CONV) ? 0 : PW - 1)) m_hc |= 1u << i;
      relb[i] = ((uint32_t)(pr * a.Win + pc) * CIN + kgs * 8) * 2u;
      loff[i] = pr * G::RS + G::swz(pc, kgs, pr);
      pc += DC; pr += DR;
      if (pc >= PW) { pc -= PW; pr += 1; }
    }
  }
  auto base_of = [&](const TileAt& t) -> uint32_t {      // (wrapping arithmetic: a conv patch starts one row and one pixel in front of its tile)
    const int iy0 = (KIND == KIND_CONV) ? 2 * t.tyb * TH - 1 : t.tyb * TH, ix0 = (KIND == KIND_CONV) ? 2 * t.txb * TW - 1 : t.txb * TW;
    return (uint32_t)(((t.img0 * a.Hin + iy0) * a.Win + ix0) * CIN) * 2u;
  };
  auto vmask_of = [&](const TileAt& t) -> uint32_t {
    uint32_t vm = m_in;
    if ((KIND == KIND_CONV) ? t.tyb == 0 : t.tyb == tiles_y - 1) vm &= ~m_hr;
    if ((KIND == KIND_CONV) ? t.txb == 0 : t.txb == tiles_x - 1) vm &= ~m_hc;
    return vm;
  };

  bf16x8 wf[9];
  long wq[9];
  RawPiece<SRC> raw[NPA];
  ChanCoef<SRC> cc;
  __shared__ float coef_tab[(SRC == SRC_BNRELU) ? 4 * CIN : (SRC == SRC_BNBWD) ? 3 * CIN : 4];
  const float* coefp = a.src.coef;
  auto issue = [&](uint32_t base, uint32_t vm, int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPA; ++i) load_piece_b<SRC>(rs, ((vm >> i) & 1u) ? base + relb[i] : OOB_OFF, raw[i], chunk * 64);
  };
  auto load_w = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (Q) wq[tap] = *reinterpret_cast<const long*>(wrow8 + tap * CIN + chunk * 32);
      else wf[tap] = *reinterpret_cast<const bf16x8*>(wrow + tap * CIN + chunk * 32);
    }
  };
  // load transform of the raw pieces in registers (tile with mask vm at byte offset base, channel chunk `chunk`) into a buffer
  auto stage = [&](bf16_t* buf, uint32_t base, uint32_t vm, int chunk) __attribute__((always_inline)) {
    if (NC > 1) cc.load(coefp, CIN, chunk * 32 + kgs * 8);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int q = tid + i * 256;
      if (q < NPIX * 4) {
        const uint4 o = transform_piece<SRC>(raw[i], (vm >> i) & 1u, cc);
        if (Q) amax_run = amax_pk(amax_run, o);
        *reinterpret_cast<uint4*>(buf + loff[i]) = o;
        if (DY)      // (chunk offset in the VECTOR offset: tests/test_isa_guard.py, the SGPR-soffset store hazard)
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){o.x, o.y, o.z, o.w}, rs_dy, ((vm >> i) & 1u) ? base + relb[i] + chunk * 64 : OOB_OFF, 0, 0);
      }
    }
  };

  // ---- prologue: accumulator loads, the first step's pieces, the coefficient table, the first stage, the second step's pieces
  BnFoldRegs fr;
  BnFoldRegsB frb;
  const bool folded = SRC == SRC_BNRELU && a.fold.acc != nullptr;
  const bool folded_b = SRC == SRC_BNBWD && a.bfold.acc != nullptr;
  if (folded) bn_fold_load<CIN>(a.fold, fr);
  if (folded_b) bn_fold_bwd_load<CIN>(a.bfold, frb);
  const bool has1 = (int)blockIdx.x + gstep < nvb;
  int blk_unused;
  TileAt nxt = has1 ? tile_of((int)blockIdx.x + gstep, blk_unused) : cur;
  issue(base_of(cur), vmask_of(cur), 0);
  if (folded) {
    bn_fold_fwd_finish<CIN>(a.fold, fr, coef_tab, reinterpret_cast<long long*>(smem + BUF_ELEMS), blockIdx.x == 0);
    coefp = coef_tab;
  }
  if (folded_b) {
    bn_fold_bwd_finish<CIN>(a.bfold, frb, coef_tab, reinterpret_cast<long long*>(smem + BUF_ELEMS), blockIdx.x == 0);
    coefp = coef_tab;
  }
  load_w(0);
  cc.load(coefp, CIN, kgs * 8);
  stage(smem, base_of(cur), vmask_of(cur), 0);
  // step 1 = (cur, chunk 1) or (next tile, chunk 0)
  if (NC > 1) issue(base_of(cur), vmask_of(cur), 1);
  else if (has1) issue(base_of(nxt), vmask_of(nxt), 0);
  __syncthreads();

  // epilogue state that does not depend on the tile
  constexpr int PHG = (KIND == KIND_CONV) ? 1 : 2, R2 = P * PHG;
  TileEpilogue<COUT, BN, EPI, false> epi;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == EPI_FWD) bv = *reinterpret_cast<const float4*>(a.bias + n0 + wn * 16 + kgl * 4);
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const bool do_stats = (EPI == EPI_FWD) && (a.stat_part != nullptr || a.bacc.acc != nullptr) && wave < BN / 16;
  int par = 0;                                      // buffer of the step being multiplied

  for (int vb = (int)blockIdx.x;;) {                // ---- tiles of this workgroup
    const int vb1 = vb + gstep, vb2 = vb + 2 * gstep;
    const bool h1 = vb1 < nvb, h2 = vb2 < nvb;
    const TileAt t1 = h1 ? tile_of(vb1, blk_unused) : cur;
    const TileAt t2 = h2 ? tile_of(vb2, blk_unused) : cur;
    const int tile_id = cur.tile_id, txb = cur.txb, tyb = cur.tyb, img0 = cur.img0;
#pragma unroll
    for (int i = 0; i < NPH; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int chunk = 0; chunk < NC; ++chunk) {
      bf16_t* const patch = smem + par * BUF_ELEMS;
      bf16_t* const other = smem + (par ^ 1) * BUF_ELEMS;
      // ---- MFMAs of this step: an m-tile is one tile row, so the fragment of patch row R, column offset kx serves every (m-tile, ky)
      //      with MUL*mi + ky == R; row R+1 is requested before row R's MFMAs (order pinned: see igemm_body)
      {
        constexpr int NROWS = G::MUL * MT + 1;
        bf16x8 rf[2][G::NKX];
#pragma unroll
        for (int kx = 0; kx < G::NKX; ++kx) rf[0][kx] = *reinterpret_cast<const bf16x8*>(patch + lbase[kx]);
#pragma unroll
        for (int R = 0; R < NROWS; ++R) {
          if (R + 1 < NROWS) {
#pragma unroll
            for (int kx = 0; kx < G::NKX; ++kx) rf[(R + 1) & 1][kx] = *reinterpret_cast<const bf16x8*>(patch + lbase[kx] + (R + 1) * G::RS);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int cx = 0; cx < G::NKX; ++cx) {
            long rq = 0;
            if (Q) rq = cvt8<QG>(rf[R & 1][cx], q_inv);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
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
        }
      }
      // ---- the next step's patch into the other buffer, then the request for the step after it
      const bool last = chunk + 1 == NC;
      if (!last) {
        stage(other, base_of(cur), vmask_of(cur), chunk + 1);
        if (chunk + 2 < NC) issue(base_of(cur), vmask_of(cur), chunk + 2);
        else if (h1) issue(base_of(t1), vmask_of(t1), chunk + 2 - NC);
      } else if (h1) {
        stage(other, base_of(t1), vmask_of(t1), 0);
        if (NC > 1) issue(base_of(t1), vmask_of(t1), 1);
        else if (h2) issue(base_of(t2), vmask_of(t2), 0);
      }
      if (last) {
        // ---- epilogue of this tile: accumulators (lane = pixel, 4 consecutive channels in regs) -> LDS tile in THIS step's buffer -> global
        __syncthreads();           // every wave has read its fragments out of `patch`
        bf16_t* tile = patch;
        float* red = reinterpret_cast<float*>(patch);
        epi.begin(a, n0);
        auto rowmap_of = [=](int pass) {
          return [=](int row2) -> long {
            const int row = row2 / PHG, px = row2 % PHG;
            const int ty = (row / TW) % TH, tx = row % TW;
            if (img0 >= a.B) return -1;
            const int oy = (KIND == KIND_CONV) ? tyb * TH + ty : 2 * (tyb * TH + ty) + pass;
            const int ox = (KIND == KIND_CONV) ? txb * TW + tx : 2 * (txb * TW + tx) + px;
            return (((long)img0 * Hout + oy) * Wout + ox) * COUT;
          };
        };
        constexpr int NIT = (R2 + 256 / (BN / 8) - 1) / (256 / (BN / 8));
        uint4 yv[(EPI == EPI_MASK) ? 2 : 1][(EPI == EPI_MASK) ? NIT : 1];
        if constexpr (EPI == EPI_MASK) epi.template load_prev<NIT>(a, n0, R2, rowmap_of(0), yv[0]);
        f32x4 st1 = (f32x4){0.f, 0.f, 0.f, 0.f}, st2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pass = 0; pass < NPH / PHG; ++pass) {
          if (pass) __syncthreads();
#pragma unroll
          for (int px = 0; px < PHG; ++px) {
            const int ph = pass * PHG + px;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
              const int row = (wm * MT + mi) * 16 + (lane & 15);
              uint2 w2;
              if (Q) {
                w2.x = pk2((f32x2){acc[ph][mi][0], acc[ph][mi][1]} * q_out2 + (f32x2){bv.x, bv.y});
                w2.y = pk2((f32x2){acc[ph][mi][2], acc[ph][mi][3]} * q_out2 + (f32x2){bv.z, bv.w});
              } else {
                w2.x = pk2((f32x2){acc[ph][mi][0] + bv.x, acc[ph][mi][1] + bv.y});
                w2.y = pk2((f32x2){acc[ph][mi][2] + bv.z, acc[ph][mi][3] + bv.w});
              }
              *reinterpret_cast<uint2*>(tile + (row * PHG + px) * TS + wn * 16 + kgl * 4) = w2;
            }
          }
          __syncthreads();
          if (do_stats) {
#pragma unroll
            for (int ks = 0; ks < R2 / 32; ++ks) {
              const bf16_t* lo = tile + (ks * 32 + 8 * tg + tq) * TS + wave * 16 + 4 * tp;
              bf16x8 frg = tr_frag(lo, lo + 4 * TS);
              st1 = mfma16(ones, frg, st1);
              st2 = mfma16(frg, frg, st2);
            }
          }
          if constexpr (EPI == EPI_MASK) {
            if (pass + 1 < NPH / PHG) epi.template load_prev<NIT>(a, n0, R2, rowmap_of(pass + 1), yv[(pass + 1) & 1]);
            epi.template rows_pre<NIT>(a, tile, n0, R2, rowmap_of(pass), yv[pass & 1]);
          } else {
            epi.rows(a, tile, n0, R2, rowmap_of(pass));
          }
        }
        if (do_stats) {
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
      }
      // weight fragments of the next step (L2-resident; reloaded rather than kept alive under the epilogue: igemm_body's MULTI note)
      if (!last) load_w(chunk + 1);
      else if (h1) load_w(0);
      __syncthreads();             // the next step's patch is staged; this step's buffer (patch, LDS tile, statistics scratch) is free
      par ^= 1;
    }
    if (!h1) break;
    cur = t1; vb = vb1;
  }                                // ---- tiles of this workgroup
  if (Q && a.amax != nullptr && nblk == 0) {      // the channel blocks of a tile stage the same patch: one of them reports
    const uint32_t m16 = (amax_run & 0xffffu) > (amax_run >> 16) ? (amax_run & 0xffffu) : (amax_run >> 16);
    uint32_t wmax = m16;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = __shfl_xor(wmax, sh); wmax = o > wmax ? o : wmax; }
    if (lane == 0 && wmax != 0) atomicMax(a.amax + (size_t)(cur.tile_id & a.amax_mask) * a.amax_stride, wmax << 16);
  }
}

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int SRC, int EPI>
__global__ __launch_bounds__(256, (CIN > 64 ? 1 : 2)) void igemm_mt_kernel(ConvArgs a) {
  igemm_mt_body<KIND, CIN, COUT, BN, TW, TH, SRC, EPI, 0>(a);
}
template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int SRC, int EPI>
__global__ __launch_bounds__(256, (CIN > 64 ? 1 : 2)) void igemm8_mt_kernel(ConvArgs a) {
  igemm_mt_body<KIND, CIN, COUT, BN, TW, TH, SRC, EPI, 1>(a);
}
