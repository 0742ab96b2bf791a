// Weight-gradient kernel of the 3x3 stride-2 conv / transposed-conv layers (aten::convolution_backward's wgrad,
// 39 % of the reference's CPU step time; autograd of R.md:292-304, 370-378).
//
//   R[cs][cb][tap] = sum over positions (n,oy,ox) of  S[n,oy,ox,cs] * Bg[n,2oy-1+ky,2ox-1+kx,cb]
//
//   conv   layer: S = dy (gradient of the conv output, small map), Bg = input activation (big map)  -> dW[co=cs][ci=cb][ky][kx]
//   deconv layer: S = input activation (small map), Bg = gradient of the deconv output (big map)    -> dW[ci=cs][co=cb][ky][kx]
// i.e. both land directly in the reference's parameter layout [cs][cb][3][3].
//
// The reduction runs over pixels, which are the *strided* dimension of NHWC tensors, so both MFMA operands are read
// with the hardware transposing LDS read (ds_read_b64_tr_b16): the S tile and the Bg patch sit in LDS in their natural
// [pixel][channel] order (staged with 16-byte loads, load transforms applied once per element) and each lane supplies
// the addresses of the gathered patch rows of its tap.
//
// Round-2 structure.  The round-1 kernel ran ONE 4-wave workgroup per CU through load -> wait -> stage -> barrier -> MFMA per
// tile, so a CU had either bytes in flight or matrix work, never both (8 % of the HBM roofline in situ).  Now a workgroup is
// 12 waves, three per SIMD, with two roles:
//   * waves 0-3, the CONSUMERS, hold nothing but the 64 cs x 32 cb x 9 accumulator block (72 registers) and feed the matrix
//     cores out of the current LDS tile buffer (wave w: cs-tiles {2*(w&1), +1} x cb-tile (w>>1) x 9 taps, 22 transposed reads
//     per 18 MFMAs);
//   * waves 4-11, the PRODUCERS, hold nothing but raw 16-byte pieces: they keep the NEXT TWO tiles in flight in two register
//     sets (a piece has two whole steps to arrive), apply the load transforms of tile t+1 on the VALU while the consumers
//     multiply tile t, write it to the idle LDS buffer and re-issue that register set for tile t+3.
//   So every SIMD runs one matrix wave beside two VALU / memory waves, every CU always has about two tiles (100-180 KB) in
//   flight, and neither role needs more than a third of the register file.  One barrier per tile.
//   * the block is written ONCE per workgroup as an fp32 partial already in the reference layout [cs][cb][3][3] (through an LDS
//     image: whole 1152-byte rows, 16-byte stores); the partials of the position slices are summed in slice order by
//     reduce_slices_kernel (deterministic, no atomics, no permutation pass).
#pragma once
#include "eae_common.hip.h"
#include "eae_igemm.hip.h"

struct WgradArgs {
  SrcDesc small, big;
  float* part;            // [nslices][CS][CB][9]
  int B, Hs, Ws;          // small-map spatial size (big map = 2Hs x 2Ws)
  int tiles_per_block, ntiles, nslices;
  BnBwdFold bfold;        // the SRC_BNBWD operand's coefficient table from the layer's backward accumulators (eae_common.hip.h)
  const float* qs;        // fp8 variant (wgrad8_s2_kernel): 1/scale of the small operand, 1/scale of the big operand, 1/(product)
};

constexpr int S_STRIDE = 72;   // bf16 elements per staged S row: 64 channels + 8 pad (144 B)
constexpr int WG_EP_STRIDE = 292;   // floats per cs row of the epilogue image [64][32*9 (+4)]: 16-byte rows, conflict-free 4-byte writes
constexpr int WG_THREADS = 768, WG_PRODUCERS = 512;

template <int TW, int TH, int NI>
struct WgGeo {
  static constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1, NPIX = NI * PH * PW;
  static constexpr int NPA = (NPIX * 4 + WG_PRODUCERS - 1) / WG_PRODUCERS;        // patch pieces per producer thread
  static constexpr int BUF_ELEMS = NPIX * PIX_STRIDE + 128 * S_STRIDE;            // bf16 elements of one tile buffer (patch + S)
  static constexpr size_t smem() {
    size_t tiles = (size_t)2 * BUF_ELEMS * 2, ep = (size_t)64 * WG_EP_STRIDE * 4;
    return tiles > ep ? tiles : ep;
  }
};

// raw 16-byte pieces of one tile as they come back from memory, plus their validity (zero padding / images past the batch)
template <int NPA, int SMODE, int BMODE>
struct WgRaw {
  RawPiece<BMODE> b[NPA];
  RawPiece<SMODE> s[2];
  uint32_t vm;            // bit i: patch piece i is inside the image; bit 16 + i: S piece i is inside the batch
};

#ifdef EAE_WG_EXP          // ablation builds: -DEAE_WG_EXP=<mask>, 1 = producers skip the stage, 2 = skip the re-issue, 4 = consumers skip the MFMAs
#define WG_EXP(bit) (EAE_WG_EXP & (bit))
#else
#define WG_EXP(bit) 0
#endif

// Q = 1: both fragment operands converted in registers to fp8 right before the MFMA (e5m2 for the gradient operand -- SRC_BNBWD or
// SRC_RAWG --, e4m3 for the activation operand), accumulators rescaled in the epilogue (see igemm_body, eae_igemm.hip.h)
template <int CS, int CB, int TW, int TH, int NI, int SMODE, int BMODE, int Q>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int li_given = -1) {     // li_given >= 0: the workgroup's logical id (grouped twin)
  static_assert(NI * TH * TW == 128, "tile must hold 128 positions");
  static_assert(NI <= 8, "image index of a piece is kept in 4 bits");
  constexpr bool SG = (SMODE == SRC_BNBWD || SMODE == SRC_RAWG), BG = (BMODE == SRC_BNBWD || BMODE == SRC_RAWG);
  if (Q) fp8_saturate_mode();
  const float q_s = Q ? a.qs[0] : 1.f, q_b = Q ? a.qs[1] : 1.f, q_out = Q ? a.qs[2] : 1.f;
  static_assert(CS % 64 == 0 && CB % 32 == 0, "shape");
  using G = WgGeo<TW, TH, NI>;
  constexpr int PH = G::PH, PW = G::PW, NPIX = G::NPIX, NPA = G::NPA;
  static_assert(NPA <= 6, "piece masks: 6 patch pieces + 2 S pieces per producer thread");
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);      // 0..11, wave-uniform by construction
  // workgroup -> (position slice, output block): the (CS/64)*(CB/32) output blocks of one slice read the same tiles, so they
  // sit on the same XCD, adjacent in dispatch order (ids go round-robin over the 8 XCDs), and share them through that L2
  constexpr int NBLK = (CS / 64) * (CB / 32);
  int slice, oblk;
  {
    const int bid = blockIdx.x;
    const int total = a.nslices * NBLK;
    if (li_given >= 0) { slice = li_given / NBLK; oblk = li_given % NBLK; }
    else if (NBLK == 1) { slice = bid; oblk = 0; }
    else if (total % 8 == 0) {
      // every XCD takes a CONTIGUOUS run of (slice, block) pairs: the blocks of a slice land on as few XCDs as possible (one when
      // nslices % 8 == 0; two at 4 slices x 16 blocks, where the round-1 fallback `bid % NBLK` spread a slice's blocks over all eight
      // and every operand tile was fetched 4.25x: profiles/r02_per_kernel_roofline.md)
      const int li = (bid & 7) * (total >> 3) + (bid >> 3);
      slice = li / NBLK; oblk = li % NBLK;
    }
    else { oblk = bid % NBLK; slice = bid / NBLK; }
  }
  // a slice whose blocks are split over two XCDs: adjacent block ids share the channel block of the operand with MORE bytes per tile
  // (two source tensors: g and y), so that each XCD fetches half of it and all of the lighter one (dec.deconv1: 2.05x -> ~1.4x)
  constexpr bool BHEAVY = (BMODE == SRC_BNBWD);
  const int cs0 = (BHEAVY ? oblk % (CS / 64) : oblk / (CB / 32)) * 64, cb0 = (BHEAVY ? oblk / (CS / 64) : oblk % (CB / 32)) * 32;
  const int t_first = slice * a.tiles_per_block;
  int t_end = t_first + a.tiles_per_block;
  if (t_end > a.ntiles) t_end = a.ntiles;
  const int n = t_end > t_first ? t_end - t_first : 0;          // tiles of this workgroup (same for all its waves)
  bf16_t* buf0 = smem;
  bf16_t* buf1 = smem + G::BUF_ELEMS;
  float* ep = reinterpret_cast<float*>(smem);                   // epilogue image, aliases the tile buffers
  // folded BatchNorm-backward finalize: the consumer waves (idle until the first tile is staged) build the table of the
  // SRC_BNBWD operand while the producers' first loads are in flight
  constexpr bool HASB = SMODE == SRC_BNBWD || BMODE == SRC_BNBWD;
  constexpr int CX = (SMODE == SRC_BNBWD) ? CS : CB;
  __shared__ float coef_tab[HASB ? 3 * CX : 4];
  const bool folded_b = HASB && a.bfold.acc != nullptr;

  if (wave >= 4) {
    // ================================================================ producers (512 threads)
    const int tid = threadIdx.x - 256;
    const int Hb = a.Hs * 2, Wb = a.Ws * 2;
    const int tiles_x = a.Ws / TW, tiles_y = a.Hs / TH;
    const int kgs4 = tid & 3, kgs8 = tid & 7;
    ChanCoef<BMODE> ccb;
    ChanCoef<SMODE> ccs;
    if (!(folded_b && BMODE == SRC_BNBWD)) ccb.load(a.big.coef, CB, cb0 + kgs4 * 8);
    if (!(folded_b && SMODE == SRC_BNBWD)) ccs.load(a.small.coef, CS, cs0 + kgs8 * 8);
    SrcRsrc rsb, rss;
    rsb.init<BMODE>(a.big);
    rss.init<SMODE>(a.small);
    // Piece -> (image, row, column) is a property of the THREAD, not of the tile: byte offsets relative to the tile's first pixel and
    // the "top halo row / left halo column / beyond the patch" flags are worked out once; per tile a piece costs an add, a bit test
    // and a select.  (Round 3 redid the divisions for every piece of every tile: ~170 of the 414 vector instructions a producer wave
    // spent per tile, on SIMDs it shares with the matrix waves -- by the ablation builds the producers alone needed 1.6 us per tile.)
    uint32_t relb[NPA], rels[2];
    uint32_t m_in = 0, m_top = 0, m_left = 0, imgpk = 0;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int pix = (tid + i * WG_PRODUCERS) >> 2;
      const int img = pix / (PH * PW), rem = pix % (PH * PW);
      const int pr = rem / PW, pc = rem % PW;
      relb[i] = (uint32_t)(((img * Hb + pr) * Wb + pc) * CB + kgs4 * 8) * 2u;
      if (pix < NPIX) m_in |= 1u << i;
      if (pr == 0) m_top |= 1u << i;
      if (pc == 0) m_left |= 1u << i;
      imgpk |= (uint32_t)(img & 15) << (4 * i);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = (tid + i * WG_PRODUCERS) >> 3;
      const int img = m / (TH * TW), ty = (m / TW) % TH, tx = m % TW;
      rels[i] = (uint32_t)(((img * a.Hs + ty) * a.Ws + tx) * CS + kgs8 * 8) * 2u;
      imgpk |= (uint32_t)img << (24 + 4 * i);
    }
    using Raw = WgRaw<NPA, SMODE, BMODE>;
    // request the raw pieces of tile t (live = false: a step past the end of the slice, every piece out of range -- the hardware
    // returns zeros without a memory access)
    auto issue = [&](Raw& r, int t, bool live) __attribute__((always_inline)) {
      const int txb = t % tiles_x; t /= tiles_x;
      const int tyb = t % tiles_y; t /= tiles_y;
      const int img0 = t * NI;
      const int nleft = live ? a.B - img0 : 0;          // images of this tile that exist
      // (wrapping arithmetic: the halo row / column of the first tile puts the base one row and one pixel in front of the tensor)
      const uint32_t base_b = (uint32_t)(((img0 * Hb + 2 * tyb * TH - 1) * Wb + 2 * txb * TW - 1) * CB + cb0) * 2u;
      const uint32_t base_s = (uint32_t)(((img0 * a.Hs + tyb * TH) * a.Ws + txb * TW) * CS + cs0) * 2u;
      uint32_t vm = m_in | (3u << 16);
      if (tyb == 0) vm &= ~m_top;
      if (txb == 0) vm &= ~m_left;
      if (nleft < NI) {                                  // wave-uniform and rare: the last tile of a batch that is no multiple of NI, dead steps
#pragma unroll
        for (int i = 0; i < NPA; ++i) if ((int)((imgpk >> (4 * i)) & 15u) >= nleft) vm &= ~(1u << i);
#pragma unroll
        for (int i = 0; i < 2; ++i) if ((int)((imgpk >> (24 + 4 * i)) & 15u) >= nleft) vm &= ~(1u << (16 + i));
      }
      r.vm = vm;
#pragma unroll
      for (int i = 0; i < NPA; ++i) load_piece_b<BMODE>(rsb, ((vm >> i) & 1u) ? base_b + relb[i] : OOB_OFF, r.b[i]);
#pragma unroll
      for (int i = 0; i < 2; ++i) load_piece_b<SMODE>(rss, ((vm >> (16 + i)) & 1u) ? base_s + rels[i] : OOB_OFF, r.s[i]);
    };
    auto stage = [&](const Raw& r, bf16_t* buf) __attribute__((always_inline)) {   // load transforms, once per element, into a tile buffer
      bf16_t* patch = buf;
      bf16_t* sl = buf + NPIX * PIX_STRIDE;
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int qq = tid + i * WG_PRODUCERS;
        if (qq < NPIX * 4)
          *reinterpret_cast<uint4*>(patch + (qq >> 2) * PIX_STRIDE + kgs4 * 8) = transform_piece<BMODE>(r.b[i], (r.vm >> i) & 1u, ccb);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int qq = tid + i * WG_PRODUCERS;
        *reinterpret_cast<uint4*>(sl + (qq >> 3) * S_STRIDE + kgs8 * 8) = transform_piece<SMODE>(r.s[i], (r.vm >> (16 + i)) & 1u, ccs);
      }
    };
    // EVERY step issues and stages, past the end of the slice with out-of-range offsets: with a fixed number of loads per step
    // hipcc's s_waitcnt bookkeeping sees the same two register sets in flight on every path into the loop and waits
    // vmcnt(loads of one tile) before a stage.  With the issues under `if (t + 3 < n)` (round 3) it merged the paths pessimistically
    // and drained vmcnt(0) in every stage: one tile in flight, not two (tools/waitcnt_audit.py lists such loops).
    Raw ra, rb;
    issue(ra, t_first, n > 0);
    issue(rb, t_first + 1, n > 1);
    if (folded_b) {
      __syncthreads(); __syncthreads();               // the consumers' coefficient-table barriers
      if (BMODE == SRC_BNBWD) ccb.load(coef_tab, CB, cb0 + kgs4 * 8);
      if (SMODE == SRC_BNBWD) ccs.load(coef_tab, CS, cs0 + kgs8 * 8);
    }
    stage(ra, buf0);
    issue(ra, t_first + 2, n > 2);
    __syncthreads();                                  // tile 0 staged
    // step t (consumers multiply tile t): stage tile t+1 into the idle buffer, then re-issue its register set for tile t+3
    for (int t = 0; t < n; t += 2) {
      if (!WG_EXP(1)) stage(rb, buf1);
      if (!WG_EXP(2)) issue(rb, t_first + t + 3, t + 3 < n);
      __syncthreads();
      if (!WG_EXP(1)) stage(ra, buf0);
      if (!WG_EXP(2)) issue(ra, t_first + t + 4, t + 4 < n);
      __syncthreads();
    }
    __syncthreads();                                  // consumers have written the epilogue image
  } else {
    // ================================================================ consumers (256 threads)
    if (folded_b) {
      BnFoldRegsB fr;
      bn_fold_bwd_load<CX>(a.bfold, fr);
      bn_fold_bwd_finish<CX>(a.bfold, fr, coef_tab, reinterpret_cast<long long*>(smem), false);      // two barriers
    }
    const int it0 = 2 * (wave & 1), jt = wave >> 1;
    f32x4 acc[9][2];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) acc[t9][0] = acc[t9][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // Fragment addresses = (lane part, computed once) + (compile-time part of the K-step and the tap, folded into the ds_read
    // immediate): a K-step is 32 consecutive positions, so position m = 32*ks + ml with ml = 8g+q (+4) < 32 never carries
    // into the K-step part of (image, row, column).
    auto pix_of = [](int m) { const int img = m / (TH * TW), ty = (m / TW) % TH, tx = m % TW; return (img * PH + 2 * ty) * PW + 2 * tx; };
    const int ml = 8 * g + q;
    const int s_lo = ml * S_STRIDE + it0 * 16 + 4 * p, s_hi = s_lo + 4 * S_STRIDE;
    const int p_lo = pix_of(ml) * PIX_STRIDE + jt * 16 + 4 * p, p_hi = pix_of(ml + 4) * PIX_STRIDE + jt * 16 + 4 * p;
    // One tile = 36 steps (4 K-steps of 32 positions x 9 taps), each one patch fragment feeding two MFMAs.  The fragment of step
    // s + LA is requested before the MFMAs of step s, and the order is pinned: left alone hipcc issued the reads of four steps, waited for
    // the first and multiplied -- one exposed LDS round trip per eight MFMAs (round-3 ISA; the consumers alone needed 1760 cycles per
    // tile for 1152 cycles of matrix issue).
    auto mfma_tile = [&](const bf16_t* buf) __attribute__((always_inline)) {
      const bf16_t* patch = buf;
      const bf16_t* sl = buf + NPIX * PIX_STRIDE;
      constexpr int LA = 3;
      auto rd_b = [&](int s) __attribute__((always_inline)) {
        const int ks = s / 9, tap = s % 9;
        const int pk = ((ks * 32) / (TH * TW) * PH + 2 * (((ks * 32) / TW) % TH)) * PW + 2 * ((ks * 32) % TW);   // pix_of(32*ks), compile time
        const int toff = (tap / 3) * PW + (tap % 3);
        return tr_frag(patch + p_lo + (pk + toff) * PIX_STRIDE, patch + p_hi + (pk + toff) * PIX_STRIDE);
      };
      auto rd_s = [&](int ks, int ii) __attribute__((always_inline)) {
        return tr_frag(sl + s_lo + ks * 32 * S_STRIDE + ii * 16, sl + s_hi + ks * 32 * S_STRIDE + ii * 16);
      };
      bf16x8 sa[2][2], bb[LA + 1];
      long sq[2] = {0, 0};
      sa[0][0] = rd_s(0, 0); sa[0][1] = rd_s(0, 1);
#pragma unroll
      for (int s = 0; s < LA; ++s) bb[s] = rd_b(s);
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        const int ks = s / 9, tap = s % 9;
        if (s + LA < 36) bb[(s + LA) % (LA + 1)] = rd_b(s + LA);
        if (tap == 5 && ks < 3) { sa[(ks + 1) & 1][0] = rd_s(ks + 1, 0); sa[(ks + 1) & 1][1] = rd_s(ks + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        if (Q) {
          if (tap == 0) { sq[0] = cvt8<SG>(sa[ks & 1][0], q_s); sq[1] = cvt8<SG>(sa[ks & 1][1], q_s); }
          const long bq = cvt8<BG>(bb[s % (LA + 1)], q_b);
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
            acc[tap][ii] = SG ? (BG ? __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(sq[ii], bq, acc[tap][ii], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(sq[ii], bq, acc[tap][ii], 0, 0, 0))
                              : (BG ? __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(sq[ii], bq, acc[tap][ii], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(sq[ii], bq, acc[tap][ii], 0, 0, 0));
        } else {
          acc[tap][0] = mfma16(sa[ks & 1][0], bb[s % (LA + 1)], acc[tap][0]);
          acc[tap][1] = mfma16(sa[ks & 1][1], bb[s % (LA + 1)], acc[tap][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __syncthreads();                                  // tile 0 staged
    for (int t = 0; t < n; t += 2) {
      if (!WG_EXP(4)) mfma_tile(buf0);
      __syncthreads();
      if (t + 1 < n && !WG_EXP(4)) mfma_tile(buf1);
      __syncthreads();
    }
    // epilogue image [64 cs][32 cb * 9 taps] (the loop ended with a barrier: every wave is done with the tile buffers)
    const int ecb = (jt * 16 + (lane & 15)) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ep[((it0 + ii) * 16 + (lane >> 4) * 4 + r) * WG_EP_STRIDE + ecb + tap] = Q ? acc[tap][ii][r] * q_out : acc[tap][ii][r];
    __syncthreads();
  }
  // ---- whole 1152-byte rows of the partial go out with 16-byte stores, already in the reference layout
  float* out = a.part + (size_t)slice * ((size_t)CS * CB * 9) + (size_t)cs0 * (CB * 9) + cb0 * 9;
#pragma unroll
  for (int i = 0; i < 6; ++i) {          // 64 rows x 72 float4 = 4608 = 6 x 768
    const int e = (int)threadIdx.x + i * WG_THREADS;
    const int row = e / 72, c4 = e % 72;
    *reinterpret_cast<float4*>(out + (size_t)row * (CB * 9) + c4 * 4) = *reinterpret_cast<const float4*>(ep + row * WG_EP_STRIDE + c4 * 4);
  }
}

template <int CS, int CB, int TW, int TH, int NI, int SMODE, int BMODE>
__global__ __launch_bounds__(WG_THREADS, 3) void wgrad_s2_kernel(WgradArgs a) {
  wgrad_body<CS, CB, TW, TH, NI, SMODE, BMODE, 0>(a);
}
template <int CS, int CB, int TW, int TH, int NI, int SMODE, int BMODE>
__global__ __launch_bounds__(WG_THREADS, 3) void wgrad_s2_kernel_g(GroupPack<WgradArgs> p, int gz) {      // grouped twin (eae_group.h)
  unsigned member; int li;
  group_xcd_map(member, li);
  wgrad_body<CS, CB, TW, TH, NI, SMODE, BMODE, 0>(group_args_of<WgradArgs>(member), li);
}
template <int CS, int CB, int TW, int TH, int NI, int SMODE, int BMODE>
__global__ __launch_bounds__(WG_THREADS, 3) void wgrad8_s2_kernel(WgradArgs a) {
  wgrad_body<CS, CB, TW, TH, NI, SMODE, BMODE, 1>(a);
}

// Deterministic slice reductions: block = 16 float4 lanes x 16 slice lanes; every thread sums its slices in order, then
// the 16 slice lanes are combined in a fixed order through LDS.
template <class Store>
__device__ __forceinline__ void reduce_slices_body(const float* __restrict__ part, int nslices, long n4, Store store) {
  __shared__ float4 red[16][16];
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + lx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4)
    for (int s0 = ly; s0 < nslices; s0 += 16 * 4) {     // 4 loads in flight, summed in slice order (one load per iteration exposed a
      float4 v[4];                                       // memory round trip per slice: these kernels were ~6 us of pure latency)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sidx = s0 + 16 * q;
        v[q] = sidx < nslices ? reinterpret_cast<const float4*>(part)[(long)sidx * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
  red[ly][lx] = s;
  __syncthreads();
  if (ly == 0 && i < n4) {
    float4 r = red[0][lx];
#pragma unroll
    for (int k = 1; k < 16; ++k) { r.x += red[k][lx].x; r.y += red[k][lx].y; r.z += red[k][lx].z; r.w += red[k][lx].w; }
    store(i, r);
  }
}

// out[i] = sum_s part[s][i]
struct ReduceArgs { const float* part; int nslices; long n4; float* out; float scale; };
static __device__ __forceinline__ EAE_NO_PK void reduce_slices_run(const ReduceArgs& a) {
  float* __restrict__ out = a.out;
  const float scale = a.scale;
  reduce_slices_body(a.part, a.nslices, a.n4, [=](long i, float4 r) {
    r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
    reinterpret_cast<float4*>(out)[i] = r;
  });
}
static __global__ EAE_NO_PK __launch_bounds__(256) void reduce_slices_kernel(ReduceArgs a) { reduce_slices_run(a); }
static __global__ EAE_NO_PK __launch_bounds__(256) void reduce_slices_kernel_g(GroupPack<ReduceArgs> p, int gz) { reduce_slices_run(group_args<ReduceArgs>(gz)); }

static inline unsigned reduce_slices_grid(long n4) { return (unsigned)((n4 + 15) / 16); }
static inline void launch_reduce_slices(hipStream_t st, const float* part, int nslices, long n4, float* out, float scale) {
  const ReduceArgs ra = {part, nslices, n4, out, scale};
  eae_launch(reduce_slices_kernel, reduce_slices_kernel_g, dim3(reduce_slices_grid(n4)), dim3(256), 0, st, ra);
}

// Tall variant for few outputs and many slices (conv1 / deconv4 weight gradients: 216 float4, 512 slices): 4 float4 columns
// x 64 slice lanes per block -> 4x the blocks and a quarter of the serial loads per thread; fixed summation order.
// tail: an optional gate (eae_misc.h GateArgs, same protocol as gate_kernel) that block 0 waits for AFTER its share of the reduction,
// so the kernels behind this one on the stream also follow the signalling streams' work.
struct ReduceTallArgs { const float* part; int nslices; long n4; float* out; GateArgs tail; };
static __device__ __forceinline__ EAE_NO_PK void reduce_slices_tall_body(const ReduceTallArgs& a) {
  const float* __restrict__ part = a.part; const int nslices = a.nslices; const long n4 = a.n4; float* __restrict__ out = a.out;
  __shared__ float4 red[64][4];
  const int lx = threadIdx.x & 3, ly = threadIdx.x >> 2;
  const long i = (long)blockIdx.x * 4 + lx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4)
    for (int s0 = ly; s0 < nslices; s0 += 64 * 8) {     // 8 loads in flight, summed in slice order
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int sidx = s0 + 64 * q;
        v[q] = sidx < nslices ? reinterpret_cast<const float4*>(part)[(long)sidx * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
  red[ly][lx] = s;
  __syncthreads();
  if (ly == 0 && i < n4) {
    float4 r = red[0][lx];
    for (int k = 1; k < 64; ++k) { r.x += red[k][lx].x; r.y += red[k][lx].y; r.z += red[k][lx].z; r.w += red[k][lx].w; }
    reinterpret_cast<float4*>(out)[i] = r;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.tail.n) gate_wait(a.tail);
}
static __global__ EAE_NO_PK __launch_bounds__(256) void reduce_slices_tall_kernel(ReduceTallArgs a) { reduce_slices_tall_body(a); }
static __global__ EAE_NO_PK __launch_bounds__(256) void reduce_slices_tall_kernel_g(GroupPack<ReduceTallArgs> p, int gz) { reduce_slices_tall_body(group_args<ReduceTallArgs>(gz)); }
