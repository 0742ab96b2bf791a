// Wave-specialised form of the stride-2 implicit-GEMM kernel (eae_igemm.hip.h) for the layers whose reduction runs over SEVERAL
// 32-channel chunks (CIN >= 64: enc.conv3/4, dec.deconv1/2 and their backward-data twins; R.md:300-304, 370-374).
//
// In-kernel stamps of the one-role kernel (tools/kstamp.py, round 2) for enc.conv4 at B=512: a workgroup lives 39 K cycles for
// 4.6 K cycles of MFMA work -- per chunk its four waves (one per SIMD) first run the load transforms of the patch on the VALU
// (3.4 K cycles), then feed the matrix cores from LDS (4.9 K cycles), one after the other, and more workgroups per CU did not
// help (EAE_IG_SMALL: 16.4 -> 18.3 us).  Here a workgroup is 8 waves, two per SIMD, with two roles:
//   * waves 0-3, the CONSUMERS, keep the accumulators and the weight fragments of TWO chunks (the next chunk's weights are
//     requested while the current one is multiplied), read the pixel fragments of chunk c out of patch buffer c&1 and run the
//     MFMAs; after the last chunk they run the epilogue of the one-role kernel unchanged (store, masks, BatchNorm reductions);
//   * waves 4-7, the PRODUCERS, keep the raw 16-byte pieces of the next TWO chunks in flight in two register sets, apply the
//     load transform of chunk c+1 and write it to patch buffer (c+1)&1 while the consumers multiply chunk c, then retire.
// One barrier per chunk; every SIMD runs a matrix wave beside a VALU / memory wave.
// NBL > 1 (two-chunk layers whose patch for ALL chunks fits the two buffers): the workgroup loops over the COUT/BN channel
// blocks with the patch staged once, instead of COUT/BN workgroups staging the same patch each.
#pragma once
#include "eae_igemm.hip.h"

template <int KIND, int BN, int TW, int TH, int NI, int NBL>
constexpr size_t igemm2_smem() {
  using G = Geo<KIND, TW, TH, NI>;
  constexpr size_t patch = (size_t)NI * G::PH * G::RS * 2;
  constexpr size_t tile_b = (size_t)G::P * (KIND == KIND_CONV ? 1 : 2) * (BN + 8) * 2, red_b = (size_t)2 * (256 / (BN / 8)) * BN * 4;
  constexpr size_t tile = tile_b > red_b ? tile_b : red_b;
  // one channel block per workgroup: the output tile reuses the patch buffers; several: the patches stay resident beside it
  return NBL > 1 ? 2 * patch + tile : (2 * patch > tile ? 2 * patch : tile);
}

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI, int NBL>
__device__ __forceinline__ void igemm2_body(const ConvArgs& a, const int li_given = -1) {
  using G = Geo<KIND, TW, TH, NI>;
  static_assert(G::P == 64 || G::P == 128, "tile must hold 64 or 128 positions");
  static_assert(CIN % 32 == 0 && CIN >= 64 && COUT % BN == 0 && (BN == 32 || BN == 64), "shape");
  constexpr int P = G::P, PH = G::PH, PW = G::PW, NPIX = G::NPIX, NPH = G::NPH;
  constexpr int WN = BN / 16, WM = 4 / WN;          // consumer waves along N (one 16-col n-tile each) / along M
  constexpr int MT = (P / 16) / WM;                 // m-tiles (16 positions) per consumer wave
  constexpr int NPA = (NPIX * 4 + 255) / 256;       // 16-byte patch pieces per producer thread
  constexpr int TS = BN + 8;
  constexpr bool ROWSWEEP = (TW == 16 && NI == 1);
  constexpr int NC = CIN / 32, NB = COUT / BN;
  constexpr int PATCH = NI * PH * G::RS;            // bf16 elements of one patch buffer
  static_assert(NBL == 1 || (NBL == NB && NC == 2), "the channel-block loop needs every chunk resident in the two patch buffers");
  static_assert(TH * TW <= 16 || (TH % (MT * G::RPM) == 0) || (WM == 1 && (MT * G::RPM) % TH == 0), "m-tile rows must not straddle images");
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  __shared__ float coef_tab[(SRC == SRC_BNRELU) ? 4 * CIN : (SRC == SRC_BNBWD) ? 3 * CIN : 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0-3 consumers, 4-7 producers
  const bool first_wg = li_given >= 0 ? li_given == 0 : blockIdx.x == 0;
  eae_signal_first(a.sig, a.sig_val, first_wg);
  const int Hout = (KIND == KIND_CONV) ? a.Hin >> 1 : a.Hin * 2, Wout = (KIND == KIND_CONV) ? a.Win >> 1 : a.Win * 2;
  const int Hpos = (KIND == KIND_CONV) ? Hout : a.Hin, Wpos = (KIND == KIND_CONV) ? Wout : a.Win;   // position grid
  const int tiles_x = Wpos / TW, tiles_y = Hpos / TH;
  // Workgroup -> (tile, channel block): as in the one-role kernel (channel blocks of a tile adjacent on one XCD)
  int tile_id, nblk;
  {
    const int bid = blockIdx.x;
    if (li_given >= 0) { const bool one = NB == 1 || NBL > 1; nblk = one ? 0 : li_given % NB; tile_id = one ? li_given : li_given / NB; }
    else if (NB == 1 || NBL > 1) { tile_id = bid; nblk = 0; }
    else if ((a.ntiles * NB) % 8 == 0) { const int li = (bid & 7) * ((a.ntiles * NB) >> 3) + (bid >> 3); nblk = li % NB; tile_id = li / NB; }   // contiguous run of (tile, block) pairs per XCD
    else { nblk = bid % NB; tile_id = bid / NB; }
  }
  int t = tile_id;
  const int txb = t % tiles_x; t /= tiles_x;
  const int tyb = t % tiles_y; t /= tiles_y;
  const int img0 = t * NI;
  const int iy0 = (KIND == KIND_CONV) ? 2 * tyb * TH - 1 : tyb * TH, ix0 = (KIND == KIND_CONV) ? 2 * txb * TW - 1 : txb * TW;
  const bool folded = SRC == SRC_BNRELU && a.fold.acc != nullptr;
  const bool folded_b = SRC == SRC_BNBWD && a.bfold.acc != nullptr;
  bf16_t* const patch0 = smem;
  bf16_t* const patch1 = smem + PATCH;

  if (wave >= 4) {
    // ================================================================ producers
    const int ptid = tid - 256;
    const int kgs = ptid & 3;         // k-group staged by this thread (fixed per thread)
    SrcRsrc rs;
    rs.init<SRC>(a.src);
    uint32_t boff[NPA];
    int loff[NPA];
    bool val[NPA];
    {
      constexpr int DR = (64 / PW) % PH, DC = 64 % PW, DI = 64 / (PH * PW);
      int pix = ptid >> 2;
      int img = pix / (PH * PW), rem = pix % (PH * PW);
      int pr = rem / PW, pc = rem % PW;
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        int iy = iy0 + pr, ix = ix0 + pc, n = img0 + img;
        val[i] = (ptid + i * 256 < NPIX * 4) && (n < a.B) && ((unsigned)iy < (unsigned)a.Hin) && ((unsigned)ix < (unsigned)a.Win);
        boff[i] = val[i] ? ((uint32_t)((n * a.Hin + iy) * a.Win + ix) * CIN + kgs * 8) * 2u : OOB_OFF;
        loff[i] = (img * PH + pr) * G::RS + G::swz(pc, kgs, pr);
        pc += DC; pr += DR; img += DI;
        if (pc >= PW) { pc -= PW; pr += 1; }
        if (pr >= PH) { pr -= PH; img += 1; }
      }
    }
    RawPiece<SRC> ra[NPA], rb[NPA];
    auto issue = [&](int chunk, RawPiece<SRC>* r) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NPA; ++i) load_piece_b<SRC>(rs, val[i] ? boff[i] + chunk * 64 : OOB_OFF, r[i]);
    };
    const float* coefp = a.src.coef;
    auto stage = [&](int chunk, const RawPiece<SRC>* r, bf16_t* patch) __attribute__((always_inline)) {
      ChanCoef<SRC> cc;
      cc.load(coefp, CIN, chunk * 32 + kgs * 8);
#pragma unroll
      for (int i = 0; i < NPA; ++i)
        if (ptid + i * 256 < NPIX * 4)
          *reinterpret_cast<uint4*>(patch + loff[i]) = transform_piece<SRC>(r[i], val[i], cc);
    };
    BnFoldRegs fr;
    BnFoldRegsB frb;
    if (folded) bn_fold_load<CIN>(a.fold, fr, ptid);      // accumulator loads first: vector-memory results return in issue order
    if (folded_b) bn_fold_bwd_load<CIN>(a.bfold, frb, ptid);
    EAE_STAMP_T(100, 256);
    issue(0, ra);
    issue(1, rb);
    EAE_STAMP_T(101, 256);
    if (folded) {
      bn_fold_fwd_finish<CIN>(a.fold, fr, coef_tab, reinterpret_cast<long long*>(smem), first_wg, ptid);   // two barriers
      coefp = coef_tab;
    }
    if (folded_b) {
      bn_fold_bwd_finish<CIN>(a.bfold, frb, coef_tab, reinterpret_cast<long long*>(smem), first_wg, ptid);  // two barriers
      coefp = coef_tab;
    }
    EAE_STAMP_T(102, 256);
    stage(0, ra, patch0);
    if (NC > 2) issue(2, ra);
    EAE_STAMP_T(103, 256);
    __syncthreads();                                       // chunk 0 staged
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      EAE_STAMP_T(104 + 3 * c, 256);
      if (c + 1 < NC) {
        stage(c + 1, ((c + 1) & 1) ? rb : ra, ((c + 1) & 1) ? patch1 : patch0);
        if (c + 3 < NC) issue(c + 3, ((c + 1) & 1) ? rb : ra);
      }
      EAE_STAMP_T(105 + 3 * c, 256);
      __syncthreads();                                     // chunk c multiplied, chunk c+1 staged
      EAE_STAMP_T(106 + 3 * c, 256);
    }
    return;                                                // the epilogue is the consumers' (s_barrier only counts live waves)
  }

  // ================================================================== consumers
  const int wn = wave % WN, wm = wave / WN;
  const int kgl = lane >> 4;        // k-group of the lane inside an MFMA (8 channels)
  int lbase[G::NKX][G::NVAR];       // swizzled fragment base of this lane per column offset (and row residue); rows are immediate offsets
#pragma unroll
  for (int kx = 0; kx < G::NKX; ++kx)
#pragma unroll
    for (int v = 0; v < G::NVAR; ++v) lbase[kx][v] = G::frag_lane(lane & 15, wm * MT, kx, kgl, v);
  f32x4 acc[NPH][MT];
  bf16x8 wfa[9], wfb[9];
  const bf16_t* wrow = nullptr;
  auto load_w = [&](int chunk, bf16x8* wf) __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wf[tap] = *reinterpret_cast<const bf16x8*>(wrow + tap * CIN + chunk * 32);
  };
  // MFMAs of one chunk: for every distinct patch offset, read the pixel fragments once and feed all taps that use it.
  // The fragments of offset o+1 are REQUESTED before the MFMAs of offset o (second register set) and the order is pinned with
  // sched_barrier: left alone, hipcc sinks every ds_read_b128 next to its MFMA (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, 72
  // times per chunk: ~68 cycles per MFMA instead of 16 -- that, not the matrix pipe, was the "MFMA phase" of the stamps).
  auto mfma_chunk = [&](const bf16_t* patch, const bf16x8* wf) __attribute__((always_inline)) {
    if constexpr (ROWSWEEP) {
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
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cx = 0; cx < G::NKX; ++cx)
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            constexpr int dummy = 0; (void)dummy;
            const int o = G::tap_off(tap), ry = G::off_row(o);
            if (G::off_col(o) != cx || R < ry || (R - ry) % G::MUL != 0 || (R - ry) / G::MUL >= MT) continue;
            const int mi = (R - ry) / G::MUL;
            acc[G::tap_phase(tap)][mi] = mfma16(wf[tap], rf[R & 1][cx], acc[G::tap_phase(tap)][mi]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      bf16x8 pf[2][MT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        pf[0][mi] = *reinterpret_cast<const bf16x8*>(patch + lbase[G::off_col(0)][G::var_of(G::frag_radd(mi, 0))] + G::frag_const(mi, 0));
#pragma unroll
      for (int o = 0; o < G::NOFF; ++o) {
        if (o + 1 < G::NOFF) {
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
            pf[(o + 1) & 1][mi] = *reinterpret_cast<const bf16x8*>(patch + lbase[G::off_col(o + 1)][G::var_of(G::frag_radd(mi, o + 1))] + G::frag_const(mi, o + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (G::tap_off(tap) != o) continue;
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) acc[G::tap_phase(tap)][mi] = mfma16(wf[tap], pf[o & 1][mi], acc[G::tap_phase(tap)][mi]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  bf16_t* const tile = (NBL > 1) ? smem + 2 * PATCH : smem;
  float* const red = reinterpret_cast<float*>(tile);     // aliases the tile: TileEpilogue::end() starts with a barrier after the last rows() pass
  const int B = a.B;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;

#pragma unroll 1
  for (int nbi = 0; nbi < NBL; ++nbi) {
    const int n0 = (NBL > 1 ? nbi : nblk) * BN;
#pragma unroll
    for (int i = 0; i < NPH; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's weight-fragment row: output channel n0 + wn*16 + (lane&15), 8 input channels kgl*8..
    wrow = a.wpack + (size_t)(n0 + wn * 16 + (lane & 15)) * 9 * CIN + kgl * 8;
    EAE_STAMP_T(0, 0);
    load_w(0, wfa);
    load_w(1, wfb);
    if (nbi == 0) {
      if (folded || folded_b) { __syncthreads(); __syncthreads(); }    // the producers' coefficient-table barriers
      __syncthreads();                                     // chunk 0 staged
    }
    EAE_STAMP_T(1, 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      EAE_STAMP_T(8 + 3 * c, 0);
      mfma_chunk((c & 1) ? patch1 : patch0, (c & 1) ? wfb : wfa);
      if (c + 2 < NC) load_w(c + 2, (c & 1) ? wfb : wfa);
      EAE_STAMP_T(9 + 3 * c, 0);
      if (nbi == 0) __syncthreads();                       // chunk c multiplied, chunk c+1 staged (the producers retire after the last one)
      EAE_STAMP_T(10 + 3 * c, 0);
    }
    // ---- epilogue (the one-role kernel's): per phase, accumulators -> LDS tile -> global
    constexpr int PHG = (KIND == KIND_CONV) ? 1 : 2, R2 = P * PHG;
    if (NBL > 1 && nbi > 0) __syncthreads();               // the previous block's tile / reduction scratch has been consumed
    TileEpilogue<COUT, BN, EPI, false> epi;
    epi.begin(a, n0);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_FWD) bv = *reinterpret_cast<const float4*>(a.bias + n0 + wn * 16 + kgl * 4);
    const bool do_stats = (EPI == EPI_FWD) && (a.stat_part != nullptr || a.bacc.acc != nullptr) && wave < BN / 16;
    f32x4 st1 = (f32x4){0.f, 0.f, 0.f, 0.f}, st2 = (f32x4){0.f, 0.f, 0.f, 0.f};
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
          w2.x = pk2((f32x2){acc[ph][mi][0] + bv.x, acc[ph][mi][1] + bv.y});
          w2.y = pk2((f32x2){acc[ph][mi][2] + bv.z, acc[ph][mi][3] + bv.w});
          if (NI > 1 && img0 + row / (TH * TW) >= B) w2 = make_uint2(0, 0);   // images past the batch must not enter the statistics
          *reinterpret_cast<uint2*>(tile + (row * PHG + px) * TS + wn * 16 + kgl * 4) = w2;
        }
      }
      __syncthreads();
      if (do_stats) {
#pragma unroll
        for (int ks = 0; ks < R2 / 32; ++ks) {
          const bf16_t* lo = tile + (ks * 32 + 8 * tg + tq) * TS + wave * 16 + 4 * tp;
          bf16x8 fr = tr_frag(lo, lo + 4 * TS);
          st1 = mfma16(ones, fr, st1);
          st2 = mfma16(fr, fr, st2);
        }
      }
      auto rowmap = [=](int row2) -> long {
        const int row = row2 / PHG, px = row2 % PHG;
        int img = row / (TH * TW), ty = (row / TW) % TH, tx = row % TW;
        int n = img0 + img;
        if (n >= B) return -1;
        int oy = (KIND == KIND_CONV) ? tyb * TH + ty : 2 * (tyb * TH + ty) + pass;
        int ox = (KIND == KIND_CONV) ? txb * TW + tx : 2 * (txb * TW + tx) + px;
        return (((long)n * Hout + oy) * Wout + ox) * COUT;
      };
      epi.rows(a, tile, n0, R2, rowmap);
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
    EAE_STAMP_T(2, 0);
  }
}

template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI, int NBL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm2_s2_kernel(ConvArgs a) {
  igemm2_body<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI, NBL>(a);
}
template <int KIND, int CIN, int COUT, int BN, int TW, int TH, int NI, int SRC, int EPI, int NBL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm2_s2_kernel_g(GroupPack<ConvArgs> p, int gz) {
  unsigned member; int li;
  group_xcd_map(member, li);
  igemm2_body<KIND, CIN, COUT, BN, TW, TH, NI, SRC, EPI, NBL>(group_args_of<ConvArgs>(member), li);
}
