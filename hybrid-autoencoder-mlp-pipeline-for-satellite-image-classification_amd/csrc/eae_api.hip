// C ABI of libeae.so (include/eae.h): context, arena layout, the fused forward / backward / Adam step of the
// supervised autoencoder, and thin per-op wrappers used by the kernel-level parity tests.
#include "eae_internal.h"
#include "eae_igemm.hip.h"
#include "eae_edge.hip.h"
#include "eae_wgrad.hip.h"
#include "eae_fc.hip.h"
#include <functional>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <future>
#include <memory>
#include <thread>
#include <mutex>
#include <atomic>

static thread_local std::string g_err;
int eae_set_error(int code, const char* msg) { g_err = msg ? msg : "unknown error"; return code; }
extern "C" const char* eae_last_error(void) { return g_err.c_str(); }
extern "C" int eae_version(void) { return 100; }

// fork/join events only order kernels of THIS device: no timing, and no system-scope fence when they complete (the agent-scope
// release at the end of every kernel is what makes its results visible to the other streams' kernels)
static const unsigned EV_FLAGS = hipEventDisableTiming | (getenv("EAE_EVENT_SYSTEM_FENCE") ? 0u : hipEventDisableSystemFence);

thread_local GroupRec* eae_rec = nullptr;        // eae_group.h
thread_local int eae_geo_mult = 1;
int eae_rec_fail(const char* what) {
  if (eae_rec && !eae_rec->error) eae_rec->error = EAE_ERR_STATE;
  return eae_set_error(EAE_ERR_STATE, what);
}

namespace {

constexpr float BN_EPS = 1e-5f, BN_MOM = 0.1f;
const int ENC_C[5] = {3, 32, 64, 128, 256};
const int BN_C[7] = {32, 64, 128, 256, 128, 64, 32};
const int BN_GAMMA_IDX[7] = {2, 6, 10, 14, 22, 26, 30};
const int W3_PARAM[6] = {4, 8, 12, 20, 24, 28};      // conv2, conv3, conv4, deconv1, deconv2, deconv3
const int W3_A[6] = {64, 128, 256, 256, 128, 64};
const int W3_B[6] = {32, 64, 128, 128, 64, 32};
const int PREBN_BIAS[7] = {1, 5, 9, 13, 21, 25, 29};  // biases in front of a BatchNorm: gradient is identically zero

long long r4(long long n) { return (n + 3) & ~3LL; }

void param_sizes(const eae_config& c, long long* sz) {
  const long long P = (long long)(c.image_h / 16) * (c.image_w / 16), K = 256 * P, L = c.latent_dim, C = c.num_classes;
  const long long s[38] = {32 * 27, 32, 32, 32, 64 * 32 * 9, 64, 64, 64, 128 * 64 * 9, 128, 128, 128, 256 * 128 * 9, 256, 256, 256,
                           L * K, L, K * L, K, 256 * 128 * 9, 128, 128, 128, 128 * 64 * 9, 64, 64, 64, 64 * 32 * 9, 32, 32, 32,
                           32 * 27, 3, 128 * L, 128, C * 128, C};
  for (int i = 0; i < 38; ++i) sz[i] = s[i];
}

int check_cfg(const eae_config* c) {
  if (!c) return eae_set_error(EAE_ERR_ARG, "config is NULL");
  if (c->image_h <= 0 || c->image_w <= 0 || c->image_h % 64 || c->image_w % 64) return eae_set_error(EAE_ERR_ARG, "image size must be a positive multiple of 64");
  if (c->latent_dim <= 0 || c->latent_dim > 256) return eae_set_error(EAE_ERR_ARG, "latent_dim must be in 1..256");
  if (c->num_classes <= 0 || c->num_classes > 64) return eae_set_error(EAE_ERR_ARG, "num_classes must be in 1..64");
  if (c->max_batch <= 0) return eae_set_error(EAE_ERR_ARG, "max_batch must be positive");
  if (c->quant != 0 && c->quant != 1) return eae_set_error(EAE_ERR_ARG, "quant must be 0 (bf16) or 1 (fp8 conv GEMMs)");
  if (c->quant == 1 && (c->image_h % 128 || c->image_w % 256))
    return eae_set_error(EAE_ERR_ARG, "quant=1: the fp8 kernels are built for 16 x 8 tiles on every map (image height % 128 == 0, width % 256 == 0)");
  return 0;
}

}  // namespace

struct eae_ctx {
  eae_config cfg;
  int H, W, L, C, Bm;
  // The latent-projection kernels work on a latent width padded to a multiple of 64 (Lp): the padded weight rows / columns are
  // zero in the packs, so the padded latent columns are exactly zero.  When Lp != L (`lpad`) the kernels that produce gradients in
  // parameter layout write padded shadows (gs_*), which compact_* copies into the gradient arena; with Lp == L they write the arena.
  int Lp = 0;
  bool lpad = false;
  float *gs_encw = nullptr, *gs_encb = nullptr, *gs_decw = nullptr, *gs_head = nullptr, *zstage = nullptr;
  size_t pk_w1p = 0, pk_bep = 0;     // lpad: fp32 copies of classifier.0.weight [128][Lp] and enc.fc.bias [Lp]
  // fp8 variant of the six 3x3 layers' GEMMs (eae_config::quant = 1, BASELINE config 5): e4m3 weight packs + delayed-scaling state
  bool fp8 = false;
  size_t pk8_p1[6] = {}, pk8_p2[6] = {};
  Fp8State* q = nullptr;
  float* bn_save = nullptr;          // eae_fp8_calibrate: copy of the running statistics + num_batches_tracked
  long long Pn, K;                 // pixels of the 256-channel map, flattened features
  long long poff[39], bnoff[15];
  float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr, *bnrun = nullptr;
  long long* nbt = nullptr;
  long long adam_step = 0;
  bool packed = false;
  bool fwd_ready = false;          // a train-mode forward with gradient staging is resident in the workspace
  bool fwd_eval_ready = false;     // ... or an eval-mode one (BatchNorm with running statistics): eae_ae_backward differentiates that too
  int enc_ready = 0, dec_ready = 0; // stand-alone eae_encoder_forward / eae_decoder_forward resident: 0 no, 1 train mode, 2 eval mode
  bool bwd_eval = false;           // the running backward differentiates an eval-mode forward: BatchNorm is a per-channel affine map
  bool prebn_dirty = false;        // an eval-mode backward wrote the gradients of the biases in front of the BatchNorms (train mode: zero)
  int fwd_B = 0, fwd_head = 0;
  const float* fwd_x = nullptr;    // only used between eae_ae_grad_step_begin / _end (the caller keeps the batch alive in between)
  long long fwd_gen = 0;           // bumped by every forward: eae_ae_backward refuses to differentiate a forward that is no longer resident
  // workspace
  void* ws = nullptr;
  bf16_t *y[4], *u[3], *d0, *gy[4], *gu[3], *gd0, *g4;
  // dy = BatchNorm-backward-applied gradients, written by the backward-data kernels while they stage them (ConvArgs::dy_out) and
  // read by the weight-gradient kernels: dyy[i] has the shape of gy[i] (i = 1..3), dyu[i] that of gu[i]
  bf16_t *dyy[4], *dyu[3];
  float *z, *dz, *dzc, *coef_f[7], *coef_b[7], *stat, *wscratch, *fcpart, *msepart, *cepart, *headpart, *lossbuf;
  long long wscratch_floats, head_stride;
  uint8_t* pack = nullptr;
  PackDesc* descs_dev = nullptr;
  int ndesc = 0;
  unsigned short* blkmap = nullptr;   // flattened pack launch (eae_pack_assign_blocks): workgroup -> descriptor; blk_tot workgroups, the first blk_late belong to descriptors 0..ndesc_late-1
  int blk_tot = 0, blk_late = 0;
  size_t pk_c1, pk_p1[6], pk_p2[6], pk_d4j, pk_d4k, pk_we1, pk_we2, pk_wd1, pk_wd2, pk_bd;
  // second stream: weight-gradient kernels, the classifier head and the slice reductions do not sit on the
  // forward / backward-data dependency chain, so they run concurrently with it (fork/join through events)
  hipStream_t side = nullptr;
  // extra side streams: weight-gradient groups go round-robin over side + these, each with its own split-K scratch, so a
  // layer's slice reduction overlaps the next layers' wgrad kernels
  static constexpr int MAXX = 3;
  int nx = 0;
  hipStream_t sidex[MAXX] = {};
  hipEvent_t ev_joinx[MAXX] = {}, ev_sx[MAXX] = {};
  float* wscratchx[MAXX] = {};
  float* wscratch_main = nullptr;
  // data-parallel hand-off streams: after a gradient step #0 is ordered after gradient tensors 18..37 (classifier, decoder,
  // dec.fc) and #1 after tensors 8..17 (enc.fc, conv4, conv3); tensors 0..7 are complete when the step's join is reached
  hipStream_t dp_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_part[2] = {nullptr, nullptr};
  // RCCL communicator owned by the engine (eae_dp_init): the gradient all-reduce is enqueued by the engine itself, no host
  // code between the backward and the collective
  void* dp_comm = nullptr;
  int dp_rank = 0, dp_world = 0;
  hipEvent_t ev_dp_done = nullptr;
  // folded BatchNorm finalize (forward): fixed-point statistics accumulators per BN layer
  unsigned long long* accf[7] = {};
  int acc_copies[7] = {};
  unsigned long long* accb[7] = {};      // BatchNorm-backward accumulators (same region and layout, behind the forward ones)
  bool fold_bwd = true;
  uint8_t* acc_base = nullptr;
  size_t poison_off = 0;       // byte offset of the step-wide non-finite word inside the accumulator region (cleared with it)
  size_t acc_bytes = 0;
  bool acc_clean = false;          // all zero (cleared by the engine's own Adam launch or at creation)
  bool bwd_dirty = false;          // the BACKWARD half of the accumulators holds the sums of an earlier backward (no clear since)
  size_t acc_half = 0;             // byte offset of the backward half inside the accumulator region
  // synchronized BatchNorm across data-parallel replicas (eae_set_sync_bn): the batch statistics of every BN layer are summed
  // over the replicas through the caller's hook (forward: the fixed-point accumulators; backward: the fp64 sums) before the
  // consumers turn them into coefficients with the GLOBAL element count
  int sync_world = 1;
  eae_sync_fn sync_fn = nullptr;
  void* sync_user = nullptr;
  double* sync_sums = nullptr;     // caller-owned device buffer [7][2][256] fp64
  bool fold_fwd = true;
  int side_rr = 0;
  // Hand-overs to the side streams (sq_* below): queued launches, the progress value their group waits for, the value the next
  // kernel of the caller's stream has to publish, and the device words: [0] progress of the caller's stream, [1 + k] work done
  // by side stream k, [8] gate time-out report.
  struct SideItem { std::function<int(hipStream_t, float*)> fn; int pin; };
  std::vector<SideItem> sq_items;
  unsigned sig_seq = 0, sq_wait = 0, pending_sig = 0;
  bool sq_forked = false;          // the queued group has been released (sq_fork): commit it behind the next kernel of the caller's stream
  unsigned* sigwords = nullptr;
  unsigned side_done_seq[1 + MAXX] = {};
  unsigned side_used = 0;          // bit k: side stream k received work since the last join
  // Per layer: does the backward-data kernel store dy (ConvArgs::dy_out) for the weight gradient, which then runs BEHIND it on one
  // plain tensor, or does the weight gradient transform g and y itself and run BESIDE the backward-data kernel?  bit i (1..3) =
  // enc.conv(i+1), bit 4 + i (0..2) = dec.deconv(i+1).  EAE_DY_MASK overrides (diagnostic A/B).
  unsigned dy_mask = 0;
  // Split optimizer of the fused eager step (round 4): Adam + pack of gradient tensors 8..37 run on a side stream as soon as those
  // tensors are complete, beside conv2's backward-data and conv1's weight gradient; only tensors 0..7 (conv1, conv2 and their
  // BatchNorms) are updated and packed behind the join.  ndesc_late = leading pack descriptors that read tensors 0..7.
  int ndesc_late = 0;
  // Default OFF (measured, B=512, same box): one optimizer launch behind the join 0.4743 ms; split with full-width bulk kernels
  // 0.4837; bulk narrowed to 256 / 128 / 64 workgroups 0.4833 / 0.4905 / 0.5128.  The timeline (gpurun_out/r4tl3) shows why: both side
  // streams are busy with the last weight gradients until the main chain ends (that is what the 64-workgroup grids balance), so the
  // bulk starts behind conv3's weight gradient and the join waits for it -- the work is conserved, it only moves.
  bool split_opt = false;          // EAE_SPLIT_OPT=1 switches it on
  bool bulk_done = false;          // backward_impl has enqueued the bulk part of this step's optimizer
  unsigned bulk_seq = 0;
  int nan_exact = 0;               // EAE_NAN_EXACT=1: a diverged step writes NaN into every parameter and moment like the reference's does
  bool skip_wgrad = false;         // EAE_SKIP_WGRAD=1 (diagnostic): the six 3x3 weight gradients are not launched (main chain alone)
  bool use_gates = true;           // device-side gates instead of event records on the caller's stream (EAE_FORK_EVENTS=1: events)
  unsigned long long gate_limit = 3000000000ULL;     // gate spin bound in 100 MHz ticks (30 s; EAE_GATE_TIMEOUT_MS, 0 = unbounded)
  float* last_loss = nullptr;      // the caller's loss_last buffer of the most recent step: poisoned with NaN when a gate has timed out
  std::atomic<long long> last_step_ns{0};   // steady-clock time of the last step that went through streams_distinct (idle contexts' claims are ignored)
  bool streams_exposed = false;     // eae_side_stream() handed a side stream to the caller: it is never replaced afterwards
  hipStream_t probed_user = nullptr; bool probed = false;   // streams_distinct(): the caller's stream the side streams were checked against
  int side_prio = 0;
  hipStream_t own_main = nullptr;  // capture is not permitted on the legacy default stream: graphs run here, bracketed by events
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  static constexpr int NEV = 16;
  hipEvent_t ev_fork[NEV] = {};
  hipEvent_t ev_join = nullptr;
  hipEvent_t ev_head = nullptr;    // classification head finished on the side stream
  bool head_side = false, head_pending = false;
  int ev_i = 0;
  bool use_side = true;
  // hipGraph replay of the whole train step: ~80 launches + fork/join events per step make the eager path host-bound
  struct GraphKey {
    const void *x, *labels, *x_hat, *accum, *last;
    int B, head; float alpha;
    bool operator==(const GraphKey& o) const {
      return x == o.x && labels == o.labels && x_hat == o.x_hat && accum == o.accum && last == o.last && B == o.B && head == o.head && alpha == o.alpha;
    }
  };
  struct GraphEntry { GraphKey key; int seen = 0; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; };
  static constexpr int NGRAPH = 8;
  GraphEntry graphs[NGRAPH];
  int ngraphs = 0;
  bool use_graph = true;
  bool capturing = false;
  float* dyn = nullptr;            // device: lr/bc1, sqrt(bc2), weight decay of the current Adam step
  // optional in-situ timing of ONE launch site (eae_profile_enable(ctx, site); sites: include/eae.h) with HIP events on the
  // stream that launch goes to
  static constexpr int PROF_RING = 64;
  bool prof_on = false;
  int prof_site = 0;
  int prof_n = 0;
  hipEvent_t prof_ev[3 * PROF_RING] = {};     // per sample: before, after, after an EMPTY bracket (calibration)
  EaeProfHook prof_hook = {};
  long long act_elems(int lvl) const {   // per-image elements of the map after `lvl` stride-2 stages (1..4)
    return (long long)(H >> lvl) * (W >> lvl) * ENC_C[lvl];
  }
};

extern "C" int eae_ae_layout(const eae_config* cfg, long long* param_off, long long* bn_off) {
  if (int rc = check_cfg(cfg)) return rc;
  long long sz[38];
  param_sizes(*cfg, sz);
  long long o = 0;
  for (int i = 0; i < 38; ++i) { if (param_off) param_off[i] = o; o += r4(sz[i]); }
  if (param_off) param_off[38] = o;
  o = 0;
  for (int l = 0; l < 7; ++l) {
    if (bn_off) { bn_off[2 * l] = o; bn_off[2 * l + 1] = o + BN_C[l]; }
    o += 2 * BN_C[l];
  }
  if (bn_off) bn_off[14] = o;
  return 0;
}

extern "C" int eae_create(const eae_config* cfg, eae_ctx** out) {
  if (!out) return eae_set_error(EAE_ERR_ARG, "out is NULL");
  if (int rc = check_cfg(cfg)) return rc;
  eae_ctx* c = new eae_ctx();
  c->cfg = *cfg; c->H = cfg->image_h; c->W = cfg->image_w; c->L = cfg->latent_dim; c->C = cfg->num_classes; c->Bm = cfg->max_batch;
  c->Lp = (c->L + 63) / 64 * 64; c->lpad = c->Lp != c->L;
  c->Pn = (long long)(c->H / 16) * (c->W / 16); c->K = 256 * c->Pn;
  eae_ae_layout(cfg, c->poff, c->bnoff);
  const long long Bm = c->Bm;
  // ---- carve one allocation
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  size_t o_y[4], o_u[3], o_gy[4], o_gu[3], o_dyy[4] = {0, 0, 0, 0}, o_dyu[3];
  for (int i = 0; i < 4; ++i) { o_y[i] = carve(Bm * c->act_elems(i + 1) * 2); o_gy[i] = carve(Bm * c->act_elems(i + 1) * 2); }
  for (int i = 0; i < 3; ++i) { o_u[i] = carve(Bm * c->act_elems(3 - i) * 2); o_gu[i] = carve(Bm * c->act_elems(3 - i) * 2); }
  const unsigned dy_mask_env = (getenv("EAE_DY_MASK") ? (unsigned)strtoul(getenv("EAE_DY_MASK"), nullptr, 0) : 0u) & 0x3cu;
  for (int i = 1; i < 4; ++i) o_dyy[i] = ((dy_mask_env >> i) & 1u) ? carve(Bm * c->act_elems(i + 1) * 2) : 0;
  for (int i = 0; i < 3; ++i) o_dyu[i] = ((dy_mask_env >> (4 + i)) & 1u) ? carve(Bm * c->act_elems(3 - i) * 2) : 0;
  size_t o_d0 = carve(Bm * c->K * 2), o_gd0 = carve(Bm * c->K * 2), o_g4 = carve(Bm * (size_t)c->H * c->W * 4 * 2);
  size_t o_z = carve(Bm * c->Lp * 4), o_dz = carve(Bm * c->Lp * 4), o_dzc = carve(Bm * c->Lp * 4);
  size_t o_cf[7], o_cb[7];
  for (int l = 0; l < 7; ++l) { o_cf[l] = carve(4 * BN_C[l] * 4); o_cb[l] = carve(3 * BN_C[l] * 4); }
  // statistics partials: the largest producer is conv1 / deconv4-backward (tiles x 2 x 32) or enc.fc backward (mtiles*P x 2 x 256)
  long long stat_floats = 0;
  {
    long long t1 = (long long)eae_edge_tiles((int)Bm, c->H, c->W) * 2 * 32;
    long long t2 = (long long)eae_conv_s2_ntiles(0, (int)Bm, c->H / 2, c->W / 2) * 2 * 64;
    long long t3 = ((Bm + 127) / 128) * c->Pn * 2 * 256;
    long long t4 = (long long)eae_conv_s2_ntiles(1, (int)Bm, c->H / 4, c->W / 4, 64) * 2 * 32;
    long long t5 = (long long)eae_conv_s2_ntiles(1, (int)Bm, c->H / 8, c->W / 8) * 2 * 64 + (long long)Bm * 2 * 256;
    stat_floats = std::max(std::max(t1, t2), std::max(t3, std::max(t4, t5))) + 1024;
  }
  size_t o_stat = carve(stat_floats * 4);
  c->wscratch_floats = 6LL * 1024 * 1024;    // 24 MB of fp32 split-K partials
  size_t o_wscr = carve(c->wscratch_floats * 4), o_wscrx[eae_ctx::MAXX];
  {
    // Three side streams + the caller's stream = the GPU's four hardware queues (round 4: 0.4655 vs 0.4731 ms per B=512 step with two,
    // 0.505 with four -- five streams on four queues; c2 0.322 vs 0.326, config-5 shape unchanged).  Rounds 1-3 measured no gain from a
    // third one: its queue was whichever the runtime handed out, often the caller's (streams_distinct below now checks and repairs).
    // Grouped steps keep two (train.py: 0.596 vs 0.604 ms per group step), concurrent groups one each.
    const char* e = getenv("EAE_SIDE_STREAMS");
    int ns = e ? atoi(e) : 3;
    if (cfg->side_streams > 0) ns = cfg->side_streams;
    c->nx = ns < 1 ? 0 : (ns - 1 > eae_ctx::MAXX ? eae_ctx::MAXX : ns - 1);
    if (getenv("EAE_ONE_SIDE_STREAM")) c->nx = 0;
  }
  for (int i = 0; i < c->nx; ++i) o_wscrx[i] = carve(c->wscratch_floats * 4);
  size_t o_wscrm = carve((size_t)2048 * 864 * 4);
  size_t o_acc[7], acc_total = 0;
  {
    const int Bi = (int)Bm;
    const int nt[7] = {eae_edge_tiles(Bi, c->H, c->W), eae_conv_s2_ntiles(0, Bi, c->H / 2, c->W / 2), eae_conv_s2_ntiles(0, Bi, c->H / 4, c->W / 4, 64),
                       eae_conv_s2_ntiles(0, Bi, c->H / 8, c->W / 8, 128), eae_conv_s2_ntiles(1, Bi, c->H / 16, c->W / 16),
                       eae_conv_s2_ntiles(1, Bi, c->H / 8, c->W / 8), eae_conv_s2_ntiles(1, Bi, c->H / 4, c->W / 4, 64)};
    for (int l = 0; l < 7; ++l) {
      int cp = 8;
      while (cp < 64 && cp * 2 * 16 <= nt[l]) cp *= 2;        // about one accumulator set per 16 producer workgroups ...
      while (cp > BN_FOLD_K * (256 / BN_C[l])) cp /= 2;       // ... but at most BN_FOLD_K sets per consumer thread
      if (const char* e = getenv("EAE_ACC_COPIES_MAX")) { int mx = atoi(e); while (mx >= 1 && cp > mx) cp /= 2; }
      c->acc_copies[l] = cp;
      o_acc[l] = acc_total;
      acc_total += (size_t)cp * 2 * BN_C[l] * 8 + (size_t)BN_C[l] * 8;        // + the layer's [C] sticky non-finite flag words (BnAcc::flag)
    }
  }
  size_t o_accb = carve(2 * acc_total + 32);  // forward accumulators, then the backward ones (cleared together)   // conv1 weight gradient (last kernel of the backward, runs on the main stream)
  const int ksplit = (int)(c->K / 128);
  size_t o_fcp = carve((size_t)ksplit * Bm * c->Lp * 4);
  size_t o_mse = carve(std::max((size_t)eae_edge_tiles((int)Bm, c->H, c->W), (size_t)((Bm * c->H * c->W + 255) / 256)) * 4 * 4);
  const long long hb = eae_head_blocks((int)Bm, 256);     // (the narrow-row variant of wide latents has the most blocks)
  c->head_stride = r4(128LL * c->Lp) + 128 + r4(128LL * c->C) + r4(c->C);
  size_t o_gsew = 0, o_gseb = 0, o_gsdw = 0, o_gsh = 0, o_zst = 0;
  if (c->lpad) {
    o_gsew = carve((size_t)c->Lp * c->K * 4); o_gseb = carve((size_t)c->Lp * 4); o_gsdw = carve((size_t)c->K * c->Lp * 4);
    o_gsh = carve((size_t)c->head_stride * 4); o_zst = carve(Bm * c->Lp * 4);
  }
  size_t o_ce = carve(hb * 2 * 4), o_head = carve(hb * c->head_stride * 4), o_loss = carve(64 * 4), o_dyn = carve(64), o_sig = carve(64);
  // ---- pack arena
  size_t poffb = 0;
  auto pcarve = [&](size_t bytes) { size_t o = poffb; poffb += (bytes + 255) & ~(size_t)255; return o; };
  std::vector<PackDesc> descs;
  auto add = [&](long long src, size_t dst, long long cnt, int mode, int d0, int d1, int d2, int f32) {
    PackDesc d; d.src_off = src; d.dst_off = (long long)dst; d.count = cnt; d.mode = mode; d.d0 = d0; d.d1 = d1; d.d2 = d2; d.out_f32 = f32;
    d.lv = d0; d.q_layer = -1;
    descs.push_back(d);
  };
  c->pk_c1 = pcarve(32 * 64 * 2); add(c->poff[0], c->pk_c1, 32 * 64, PACK_K36, 32, 3, 0, 0);
  for (int i = 0; i < 6; ++i) {
    long long n = (long long)W3_A[i] * W3_B[i] * 9;
    c->pk_p1[i] = pcarve(n * 2); add(c->poff[W3_PARAM[i]], c->pk_p1[i], n, PACK_3x3_P1, W3_A[i], W3_B[i], 0, 0);
    c->pk_p2[i] = pcarve(n * 2); add(c->poff[W3_PARAM[i]], c->pk_p2[i], n, PACK_3x3_P2, W3_A[i], W3_B[i], 0, 0);
    if (cfg->quant == 1) {
      c->pk8_p1[i] = pcarve(n); add(c->poff[W3_PARAM[i]], c->pk8_p1[i], n, PACK_3x3_P1, W3_A[i], W3_B[i], 0, 0); descs.back().q_layer = i;
      c->pk8_p2[i] = pcarve(n); add(c->poff[W3_PARAM[i]], c->pk8_p2[i], n, PACK_3x3_P2, W3_A[i], W3_B[i], 0, 0); descs.back().q_layer = i;
    }
  }
  c->fp8 = cfg->quant == 1;
  c->pk_d4j = pcarve(16 * 128 * 2); add(c->poff[32], c->pk_d4j, 16 * 128, PACK_DECONV4_JOINT, 0, 0, 0, 0);
  c->pk_d4k = pcarve(32 * 64 * 2); add(c->poff[32], c->pk_d4k, 32 * 64, PACK_K36, 32, 3, 0, 0);
  const long long LK = c->Lp * c->K;      // d0 = padded latent width, lv = the real one (rows / columns beyond it are zero)
  c->pk_we1 = pcarve(LK * 2); add(c->poff[16], c->pk_we1, LK, PACK_FC_ROWMAJOR_KPERM, c->Lp, 256, (int)c->Pn, 0);
  c->pk_we2 = pcarve(LK * 2); add(c->poff[16], c->pk_we2, LK, PACK_FC_TRANS_KPERM, c->Lp, 256, (int)c->Pn, 0);
  c->pk_wd1 = pcarve(LK * 2); add(c->poff[18], c->pk_wd1, LK, PACK_FC_ROWPERM, c->Lp, 256, (int)c->Pn, 0);
  c->pk_wd2 = pcarve(LK * 2); add(c->poff[18], c->pk_wd2, LK, PACK_FC_ROWPERM_TRANS, c->Lp, 256, (int)c->Pn, 0);
  for (int k = 0; k < 4; ++k) descs[descs.size() - 1 - k].lv = c->L;
  if (c->lpad) {
    c->pk_w1p = pcarve(128LL * c->Lp * 4); add(c->poff[34], c->pk_w1p, 128LL * c->Lp, PACK_PAD_COLS, c->Lp, 0, 0, 1); descs.back().lv = c->L;
    c->pk_bep = pcarve(c->Lp * 4); add(c->poff[17], c->pk_bep, c->Lp, PACK_PAD_COLS, c->Lp, 0, 0, 1); descs.back().lv = c->L;
  }
  c->pk_bd = pcarve(c->K * 4); add(c->poff[19], c->pk_bd, c->K, PACK_FC_ROWPERM, 1, 256, (int)c->Pn, 1);
  c->ndesc = (int)descs.size();
  c->ndesc_late = 0;
  while (c->ndesc_late < c->ndesc && descs[c->ndesc_late].src_off < c->poff[8]) c->ndesc_late++;
  for (int k = c->ndesc_late; k < c->ndesc; ++k) if (descs[k].src_off < c->poff[8]) c->ndesc_late = -1;      // (not a prefix: no split)
  std::vector<unsigned short> blkmap(descs.size() * 1024);
  c->blk_tot = eae_pack_assign_blocks(descs.data(), (int)descs.size(), blkmap.data(), (int)blkmap.size());
  c->blk_late = c->ndesc_late > 0 ? descs[c->ndesc_late - 1].blk0 + descs[c->ndesc_late - 1].nblk : 0;
  size_t o_pack = carve(poffb), o_desc = carve(descs.size() * sizeof(PackDesc)), o_q = carve(sizeof(Fp8State)), o_bns = carve(2048 * 4 + 64);
  size_t o_bmap = carve((size_t)(c->blk_tot > 0 ? c->blk_tot : 1) * sizeof(unsigned short));
  hipError_t e = hipMalloc(&c->ws, off);
  if (e != hipSuccess) { delete c; return eae_set_error(EAE_ERR_HIP, hipGetErrorString(e)); }
  uint8_t* b = static_cast<uint8_t*>(c->ws);
  for (int i = 0; i < 4; ++i) { c->y[i] = (bf16_t*)(b + o_y[i]); c->gy[i] = (bf16_t*)(b + o_gy[i]); }
  for (int i = 0; i < 3; ++i) {
    c->u[i] = (bf16_t*)(b + o_u[i]); c->gu[i] = (bf16_t*)(b + o_gu[i]);
    c->dyu[i] = ((dy_mask_env >> (4 + i)) & 1u) ? (bf16_t*)(b + o_dyu[i]) : nullptr;
  }
  c->dyy[0] = nullptr;
  for (int i = 1; i < 4; ++i) c->dyy[i] = ((dy_mask_env >> i) & 1u) ? (bf16_t*)(b + o_dyy[i]) : nullptr;
  c->d0 = (bf16_t*)(b + o_d0); c->gd0 = (bf16_t*)(b + o_gd0); c->g4 = (bf16_t*)(b + o_g4);
  c->z = (float*)(b + o_z); c->dz = (float*)(b + o_dz); c->dzc = (float*)(b + o_dzc);
  if (c->lpad) {
    c->gs_encw = (float*)(b + o_gsew); c->gs_encb = (float*)(b + o_gseb); c->gs_decw = (float*)(b + o_gsdw);
    c->gs_head = (float*)(b + o_gsh); c->zstage = (float*)(b + o_zst);
  }
  for (int l = 0; l < 7; ++l) { c->coef_f[l] = (float*)(b + o_cf[l]); c->coef_b[l] = (float*)(b + o_cb[l]); }
  c->stat = (float*)(b + o_stat); c->wscratch = (float*)(b + o_wscr); c->wscratch_main = (float*)(b + o_wscrm);
  c->acc_base = b + o_accb; c->poison_off = (2 * acc_total + 15) & ~(size_t)15; c->acc_bytes = c->poison_off + 16; c->acc_half = acc_total;      // + the step-wide poison word
  for (int l = 0; l < 7; ++l) c->accf[l] = (unsigned long long*)(b + o_accb + o_acc[l]);
  for (int l = 0; l < 7; ++l) c->accb[l] = (unsigned long long*)(b + o_accb + acc_total + o_acc[l]);
  for (int i = 0; i < c->nx; ++i) c->wscratchx[i] = (float*)(b + o_wscrx[i]); c->fcpart = (float*)(b + o_fcp);
  c->msepart = (float*)(b + o_mse); c->cepart = (float*)(b + o_ce); c->headpart = (float*)(b + o_head); c->lossbuf = (float*)(b + o_loss);
  c->pack = b + o_pack; c->descs_dev = (PackDesc*)(b + o_desc); c->blkmap = (unsigned short*)(b + o_bmap);
  c->dyn = (float*)(b + o_dyn);
  c->sigwords = (unsigned*)(b + o_sig);
  c->q = (Fp8State*)(b + o_q);
  c->bn_save = (float*)(b + o_bns);
  {
    Fp8State h;
    eae_fp8_state_init(&h);
    hipError_t eq = hipMemcpy(c->q, &h, sizeof(h), hipMemcpyHostToDevice);
    if (eq != hipSuccess) { hipFree(c->ws); delete c; return eae_set_error(EAE_ERR_HIP, hipGetErrorString(eq)); }
  }
  // Gate kernels need the kernel they wait for to be able to start while they spin.  rocprofv3's counter collection (--pmc) runs one
  // kernel at a time on the device: under it (ROCPROF_COUNTER_COLLECTION=1 in the environment) the hand-overs fall back to events.
  const char* rcc = getenv("ROCPROF_COUNTER_COLLECTION");
  c->use_gates = getenv("EAE_FORK_EVENTS") == nullptr && !(rcc && atoi(rcc) != 0);
  if (const char* gt = getenv("EAE_GATE_TIMEOUT_MS")) c->gate_limit = (unsigned long long)(atof(gt) * 1e5);
  // hipGraph replay is opt-in (EAE_GRAPH=1): on ROCm 7.2 the replayed graph ran its two branches one after the other
  // (0.80 ms/step) while the eager two-stream launch sequence overlaps them (0.71 ms/step)
  c->use_graph = getenv("EAE_GRAPH") != nullptr && getenv("EAE_NO_GRAPH") == nullptr;
  e = hipMemcpy(c->descs_dev, descs.data(), descs.size() * sizeof(PackDesc), hipMemcpyHostToDevice);
  if (e == hipSuccess && c->blk_tot > 0) e = hipMemcpy(c->blkmap, blkmap.data(), (size_t)c->blk_tot * sizeof(unsigned short), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(c->g4, 0, Bm * (size_t)c->H * c->W * 4 * 2);
  if (e == hipSuccess && c->lpad) e = hipMemset(c->zstage, 0, Bm * (size_t)c->Lp * 4);
  if (e == hipSuccess) e = hipMemset(c->z, 0, Bm * (size_t)c->Lp * 4);
  if (e == hipSuccess) e = hipMemset(c->dz, 0, Bm * (size_t)c->Lp * 4);
  if (e == hipSuccess) e = hipMemset(c->acc_base, 0, c->acc_bytes);
  if (e == hipSuccess) e = hipMemset(c->sigwords, 0, 64);
  c->acc_clean = true; c->bwd_dirty = false;
  // Default OFF (round 4, measured): with dy the weight-gradient kernels lose a third of their stand-alone time (31-35 -> 27-32 us),
  // but they start one kernel later and the backward-data kernels -- the critical chain -- carry the extra stores: ms per step at
  // B=512 with dy for no layer / conv3+conv4 / deconv1+deconv2 / all four: 0.4840 / 0.4833 / 0.4861 / 0.4933.  The 32 <-> 64-channel
  // layers' backward-data kernels are built without the store (eae_igemm.hip.h: DY), so bits 1 and 6 are never honoured.
  c->dy_mask = (getenv("EAE_DY_MASK") ? (unsigned)strtoul(getenv("EAE_DY_MASK"), nullptr, 0) : 0u) & 0x3cu;
  c->skip_wgrad = getenv("EAE_SKIP_WGRAD") != nullptr;
  c->split_opt = getenv("EAE_SPLIT_OPT") && atoi(getenv("EAE_SPLIT_OPT")) != 0;
  c->nan_exact = (getenv("EAE_NAN_EXACT") && atoi(getenv("EAE_NAN_EXACT")) != 0) ? 1 : 0;
  c->fold_fwd = getenv("EAE_NO_FOLD_FWD") == nullptr;
  c->fold_bwd = getenv("EAE_NO_FOLD_BWD") == nullptr;
  if (e != hipSuccess) { hipFree(c->ws); delete c; return eae_set_error(EAE_ERR_HIP, hipGetErrorString(e)); }
  c->use_side = getenv("EAE_NO_SIDE_STREAM") == nullptr && cfg->side_streams >= 0;
  if (c->use_side) {
    // Side work only feeds the optimizer.  Lowest stream priority (EAE_SIDE_PRIO_LOW=1) lets the dependency chain on the
    // caller's stream win the CUs (-7 us/step at B=512), but it is not the default: whenever only low-priority queues had
    // work while the caller's stream waited on them through a third stream (the data-parallel step: all-reduce -> Adam) the
    // step took 1.2-2.3 ms instead of 0.63 (queue scheduling quanta), DESIGN.md section 6.
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const int side_prio = getenv("EAE_SIDE_PRIO_LOW") ? prio_lo : 0;
    c->side_prio = side_prio;
    e = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, side_prio);
    for (int i = 0; i < c->nx && e == hipSuccess; ++i) {
      e = hipStreamCreateWithPriority(&c->sidex[i], hipStreamNonBlocking, side_prio);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_joinx[i], EV_FLAGS);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_sx[i], EV_FLAGS);
    }
    for (int i = 0; i < eae_ctx::NEV && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_fork[i], EV_FLAGS);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, EV_FLAGS);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_head, EV_FLAGS);
    c->head_side = getenv("EAE_HEAD_MAIN") == nullptr;     // the head only needs z: it runs beside the decoder
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_main, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, EV_FLAGS);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, EV_FLAGS);
    if (e != hipSuccess) { hipFree(c->ws); delete c; return eae_set_error(EAE_ERR_HIP, hipGetErrorString(e)); }
  }
  *out = c;
  return 0;
}

// diagnostic: copy an internal fp32 workspace buffer to `dst` (0 = z, 1 = dz, 2 = dzc, 3 = headpart, 4 = cepart)
extern "C" int eae_debug_copy(eae_ctx* c, int which, float* dst, long long n) {
  if (!c || !dst) return eae_set_error(EAE_ERR_ARG, "debug_copy: NULL");
  const float* src = which == 0 ? c->z : which == 1 ? c->dz : which == 2 ? c->dzc : which == 3 ? c->headpart : c->cepart;
  EAE_HIP(hipDeviceSynchronize());
  EAE_HIP(hipMemcpy(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice));
  return 0;
}

extern "C" int eae_profile_enable(eae_ctx* c, int site) {
  if (!c) return eae_set_error(EAE_ERR_ARG, "ctx is NULL");
  if (site < 0 || site >= EAE_PROF_NSITES) return eae_set_error(EAE_ERR_ARG, "profile: unknown launch site");
  if (site && !c->prof_ev[0])
    for (int i = 0; i < 3 * eae_ctx::PROF_RING; ++i) EAE_HIP(hipEventCreate(&c->prof_ev[i]));
  c->prof_on = site != 0;
  c->prof_site = site;
  c->prof_n = 0;
  return 0;
}

extern "C" int eae_profile_read2(eae_ctx* c, double* total_ms, double* empty_ms, long long* count) {
  if (!c || !total_ms || !count) return eae_set_error(EAE_ERR_ARG, "profile_read: NULL argument");
  double tot = 0.0, emp = 0.0;
  for (int i = 0; i < c->prof_n; ++i) {
    float ms = 0.f;
    EAE_HIP(hipEventSynchronize(c->prof_ev[3 * i + 2]));
    EAE_HIP(hipEventElapsedTime(&ms, c->prof_ev[3 * i], c->prof_ev[3 * i + 1]));
    tot += ms;
    EAE_HIP(hipEventElapsedTime(&ms, c->prof_ev[3 * i + 1], c->prof_ev[3 * i + 2]));
    emp += ms;
  }
  *total_ms = tot; *count = c->prof_n;
  if (empty_ms) *empty_ms = emp;
  c->prof_n = 0;
  return 0;
}
extern "C" int eae_profile_read(eae_ctx* c, double* total_ms, long long* count) { return eae_profile_read2(c, total_ms, nullptr, count); }

extern "C" int eae_dp_destroy(eae_ctx* c);
namespace { void streams_forget(const eae_ctx* c); }
extern "C" int eae_destroy(eae_ctx* c) {
  if (!c) return 0;
  hipDeviceSynchronize();
  streams_forget(c);
  eae_dp_destroy(c);
  if (c->prof_ev[0]) for (int i = 0; i < 3 * eae_ctx::PROF_RING; ++i) hipEventDestroy(c->prof_ev[i]);
  for (int i = 0; i < c->ngraphs; ++i) {
    if (c->graphs[i].exec) hipGraphExecDestroy(c->graphs[i].exec);
    if (c->graphs[i].graph) hipGraphDestroy(c->graphs[i].graph);
  }
  if (c->side) {
    for (int i = 0; i < eae_ctx::NEV; ++i) hipEventDestroy(c->ev_fork[i]);
    hipEventDestroy(c->ev_join);
    if (c->ev_head) hipEventDestroy(c->ev_head);
    hipStreamDestroy(c->side);
    for (int i = 0; i < 2; ++i) if (c->dp_stream[i]) { hipStreamDestroy(c->dp_stream[i]); hipEventDestroy(c->ev_part[i]); }
    for (int i = 0; i < c->nx; ++i)
      if (c->sidex[i]) { hipStreamDestroy(c->sidex[i]); hipEventDestroy(c->ev_joinx[i]); hipEventDestroy(c->ev_sx[i]); }
    if (c->own_main) { hipStreamDestroy(c->own_main); hipEventDestroy(c->ev_in); hipEventDestroy(c->ev_out); }
  }
  if (c->ws) hipFree(c->ws);
  delete c;
  return 0;
}

extern "C" int eae_bind(eae_ctx* c, float* params, float* grads, float* adam_m, float* adam_v, float* bn_running, long long* bn_nbt) {
  if (!c || !params || !bn_running) return eae_set_error(EAE_ERR_ARG, "bind: ctx, params and bn_running are required");
  c->P = params; c->G = grads; c->M = adam_m; c->V = adam_v; c->bnrun = bn_running; c->nbt = bn_nbt;
  c->packed = false; c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0;
  if (grads)
    for (int k = 0; k < 7; ++k)
      EAE_HIP(hipMemset(grads + c->poff[PREBN_BIAS[k]], 0, (size_t)(c->poff[PREBN_BIAS[k] + 1] - c->poff[PREBN_BIAS[k]]) * 4));
  return 0;
}
// Synchronized BatchNorm (new work, SURVEY.md 8e "Equivalence to test": R ranks x B/R with SyncBN == 1 rank x B).
extern "C" long long eae_sync_bn_acc_elems(eae_ctx* c) { return c ? (long long)(c->acc_bytes / 8) : -1; }
extern "C" int eae_set_sync_bn(eae_ctx* c, int world, eae_sync_fn fn, void* user, void* acc_i64, void* sums_f64) {
  if (!c) return eae_set_error(EAE_ERR_ARG, "ctx is NULL");
  if (world <= 1 || !fn) { c->sync_world = 1; c->sync_fn = nullptr; return 0; }
  if (!acc_i64 || !sums_f64) return eae_set_error(EAE_ERR_ARG, "sync_bn: the accumulator and sums buffers are required");
  if (!c->fold_fwd) return eae_set_error(EAE_ERR_STATE, "SyncBN needs the folded forward finalize (unset EAE_NO_FOLD_FWD)");
  // the forward accumulators move into the caller's buffer (same layout), so that the hook can hand tensor views of it to the collective
  uint8_t* nb = static_cast<uint8_t*>(acc_i64);
  for (int l = 0; l < 7; ++l) {
    c->accf[l] = reinterpret_cast<unsigned long long*>(nb + (reinterpret_cast<uint8_t*>(c->accf[l]) - c->acc_base));
    c->accb[l] = reinterpret_cast<unsigned long long*>(nb + (reinterpret_cast<uint8_t*>(c->accb[l]) - c->acc_base));
  }
  c->acc_base = nb;
  c->acc_clean = false;
  c->sync_world = world; c->sync_fn = fn; c->sync_user = user; c->sync_sums = static_cast<double*>(sums_f64);
  return 0;
}
// Diagnostic (synchronises the device): 0, or the progress value a gate kernel gave up waiting for after its bounded spin
// (include/eae.h); the step in which that happened produced wrong gradients.
// Clear the sticky time-out word (after the caller has dealt with the failed step); synchronises the device.
extern "C" int eae_gate_timeouts_clear(eae_ctx* c) {
  if (!c) return eae_set_error(EAE_ERR_ARG, "ctx is NULL");
  EAE_HIP(hipDeviceSynchronize());
  EAE_HIP(hipMemset(c->sigwords + 8, 0, 4));
  return 0;
}
// The same word WITHOUT synchronising the device: for callers that have just synchronised the stream they step on (every gate of
// a completed step has run by then) and share the device with other contexts -- the concurrent grid driver (train.py) must not
// stall every configuration at each epoch end of one of them.
extern "C" long long eae_gate_timeouts_nosync(eae_ctx* c) {
  if (!c) return -1;
  unsigned v = 0;
  if (hipMemcpy(&v, c->sigwords + 8, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long long)v;
}
extern "C" long long eae_gate_timeouts(eae_ctx* c) {
  if (!c) return -1;
  unsigned v = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(&v, c->sigwords + 8, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (long long)v;
}
// hipGraph replay of eae_ae_train_step on/off for this context (default: the EAE_GRAPH environment switch at creation).  One replay per
// step instead of ~70 launches: the single-configuration step at B=512 is faster eager, but K small configurations
// stepped concurrently from K host threads are bound by the host's launch rate -- there the replay wins (train.run_concurrent).
extern "C" int eae_set_graph(eae_ctx* c, int on) {
  if (!c) return eae_set_error(EAE_ERR_ARG, "ctx is NULL");
  c->use_graph = on != 0;
  return 0;
}
extern "C" int eae_params_changed(eae_ctx* c) { if (!c) return eae_set_error(EAE_ERR_ARG, "ctx is NULL"); c->packed = false; c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0; return 0; }
extern "C" int eae_set_adam_step(eae_ctx* c, long long s) { if (!c) return eae_set_error(EAE_ERR_ARG, "ctx is NULL"); c->adam_step = s; return 0; }
extern "C" long long eae_get_adam_step(eae_ctx* c) { return c ? c->adam_step : -1; }

namespace {

#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

// HIP-event bracket around ONE launch on the stream it goes to (bench.py's `roofline`): before / after, then an EMPTY bracket
// recorded right behind it, which measures what the two event records cost by themselves on that stream.
static void prof_hook_begin(void* u, hipStream_t st) {
  eae_ctx* c = static_cast<eae_ctx*>(u);
  hipEventRecord(c->prof_ev[3 * c->prof_n], st);
}
static void prof_hook_end(void* u, hipStream_t st) {
  eae_ctx* c = static_cast<eae_ctx*>(u);
  hipEventRecord(c->prof_ev[3 * c->prof_n + 1], st);
  hipEventRecord(c->prof_ev[3 * c->prof_n + 2], st);
  c->prof_n++;
}
// hook for launchers that enqueue a second kernel behind the timed one; nullptr when `site` is not the one being profiled
static const EaeProfHook* prof_hook_for(eae_ctx* c, int site) {
  if (!(c->prof_on && c->prof_site == site && c->prof_n < eae_ctx::PROF_RING && !c->capturing)) return nullptr;
  c->prof_hook = EaeProfHook{prof_hook_begin, prof_hook_end, c};
  return &c->prof_hook;
}
struct ProfBracket {
  eae_ctx* c; hipStream_t st; bool on;
  ProfBracket(eae_ctx* c_, int site, hipStream_t st_) : c(c_), st(st_) {
    on = c->prof_on && c->prof_site == site && c->prof_n < eae_ctx::PROF_RING && !c->capturing;
    if (on) hipEventRecord(c->prof_ev[3 * c->prof_n], st);
  }
  ~ProfBracket() {
    if (!on) return;
    hipEventRecord(c->prof_ev[3 * c->prof_n + 1], st);
    hipEventRecord(c->prof_ev[3 * c->prof_n + 2], st);
    c->prof_n++;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Hand-overs to the side streams.  Side work (the classification head, the loss bookkeeping, weight gradients and their slice
// reductions) only feeds the optimizer, so it runs on engine-owned streams beside the dependency chain of the caller's stream.
// Round 1 ordered every hand-over with an event record on the caller's stream: ~5 us of bubble each, 9-10 per step
// (tools/timeline.py).  Now the order is kept on the DEVICE:
//   sq_push   queue a launch (optionally pinned to side stream 0, whose order the head -> loss bookkeeping chain needs);
//   sq_fork   the queued group may start once the caller's stream has completed everything enqueued so far: it gets the next
//             progress value, which the NEXT kernel enqueued on the caller's stream publishes when it starts (take_sig);
//   sq_commit called right after that kernel has been enqueued: a one-wave gate kernel that polls the progress word goes to
//             every side stream that receives a member, then the members (round robin; each stream has its own split-K scratch).
// The gate is always enqueued AFTER the kernel that releases it, so streams that share a hardware queue cannot dead-lock, and its
// spin is bounded (eae_gate_timeouts).  EAE_FORK_EVENTS=1, and hipGraph capture, use event records instead.
// ---------------------------------------------------------------------------------------------------------------------
bool gates_now(const eae_ctx* c) { return c->use_gates && !c->capturing; }
hipStream_t side_stream(eae_ctx* c, int k) { return k == 0 ? c->side : c->sidex[k - 1]; }
void sq_push(eae_ctx* c, std::function<int(hipStream_t, float*)> f, int pin = -1) { c->sq_items.push_back({std::move(f), pin}); }
void sq_fork(eae_ctx* c) {
  if (c->sq_items.empty() || c->sq_forked) return;
  c->sq_forked = true;
  if (!c->use_side || !gates_now(c)) return;
  c->sq_wait = ++c->sig_seq;
  c->pending_sig = c->sq_wait;
}
// the kernel about to be enqueued on the caller's stream publishes the pending progress value
void take_sig(eae_ctx* c, ConvArgs& a) {
  if (!c->pending_sig) return;
  a.sig = c->sigwords; a.sig_val = c->pending_sig;
  c->pending_sig = 0;
}
int sq_commit(eae_ctx* c, hipStream_t st) {
  c->sq_forked = false;
  if (c->sq_items.empty()) return 0;
  int rc = 0;
  if (!c->use_side) {
    for (auto& it : c->sq_items) if (!rc) rc = it.fn(st, c->wscratch);
    c->sq_items.clear();
    return rc;
  }
  const int ns = 1 + c->nx;
  // members -> streams
  std::vector<int> where(c->sq_items.size());
  unsigned used = 0;
  for (size_t i = 0; i < c->sq_items.size(); ++i) {
    where[i] = c->sq_items[i].pin >= 0 ? c->sq_items[i].pin : (c->side_rr++ % ns);
    used |= 1u << where[i];
  }
  if (gates_now(c)) {
    if (!c->sq_wait) { c->sq_wait = ++c->sig_seq; c->pending_sig = c->sq_wait; }      // commit without a fork: order after `st` as it stands
    if (c->pending_sig) {                      // no kernel of the caller's stream carried the value: publish it with a kernel of its own
      RC(eae_launch_signal(st, c->sigwords, c->pending_sig));
      c->pending_sig = 0;
    }
    GateArgs g = GateArgs();
    g.word[0] = c->sigwords; g.want[0] = c->sq_wait; g.n = 1; g.timeout = c->sigwords + 8; g.limit_ticks = c->gate_limit;
    for (int k = 0; k < ns; ++k) if (used & (1u << k)) RC(eae_launch_gate(side_stream(c, k), g));
  } else {
    hipEvent_t ev = c->ev_fork[c->ev_i];
    c->ev_i = (c->ev_i + 1) % eae_ctx::NEV;
    EAE_HIP(eae_event_record(ev, st));
    for (int k = 0; k < ns; ++k) if (used & (1u << k)) EAE_HIP(eae_stream_wait_event(side_stream(c, k), ev));
  }
  c->sq_wait = 0;
  c->side_used |= used;
  for (size_t i = 0; i < c->sq_items.size(); ++i)
    if (!rc) rc = c->sq_items[i].fn(side_stream(c, where[i]), where[i] == 0 ? c->wscratch : c->wscratchx[where[i] - 1]);
  c->sq_items.clear();
  return rc;
}
// everything enqueued so far on the extra side streams completes before later work on the first one (the DP path hands
// `side` to the all-reduce)
int fold_side2(eae_ctx* c) {
  if (!c->use_side) return 0;
  for (int i = 0; i < c->nx; ++i) {
    EAE_HIP(eae_event_record(c->ev_sx[i], c->sidex[i]));
    EAE_HIP(eae_stream_wait_event(c->side, c->ev_sx[i]));
  }
  return 0;
}
// join: work enqueued on `st` from now on starts after everything enqueued so far on the side streams.  With gates: every side
// stream that received work publishes a done-counter with a one-thread kernel, ONE gate on `st` waits for all of them.
// join_side_begin: first half of the gated join -- commits the pending side groups, publishes the side streams' done-counters and
// returns the gate that waits for them in *g (g->n == 0: nothing to wait for, or the event path is in use and join_side must follow).
// A caller that has one more kernel to enqueue on `st` hands the gate to that kernel's tail instead of paying a launch for it.
int join_side_begin(eae_ctx* c, hipStream_t st, GateArgs* g) {
  *g = GateArgs();
  if (!c->use_side || !gates_now(c)) return 0;
  RC(sq_commit(c, st));
  const int ns = 1 + c->nx;
  g->timeout = c->sigwords + 8; g->limit_ticks = c->gate_limit;
  for (int k = 0; k < ns; ++k) {
    if (!(c->side_used & (1u << k))) continue;
    c->side_done_seq[k] += 1;
    RC(eae_launch_signal(side_stream(c, k), c->sigwords + 1 + k, c->side_done_seq[k]));
    g->word[g->n] = c->sigwords + 1 + k; g->want[g->n] = c->side_done_seq[k]; g->n++;
  }
  c->side_used = 0;
  return 0;
}
int join_side(eae_ctx* c, hipStream_t st) {
  if (!c->use_side) return 0;
  if (gates_now(c)) {
    GateArgs g;
    RC(join_side_begin(c, st, &g));
    if (g.n) RC(eae_launch_gate(st, g));
    return 0;
  }
  RC(sq_commit(c, st));
  EAE_HIP(eae_event_record(c->ev_join, c->side));
  EAE_HIP(eae_stream_wait_event(st, c->ev_join));
  for (int i = 0; i < c->nx; ++i) {
    EAE_HIP(eae_event_record(c->ev_joinx[i], c->sidex[i]));
    EAE_HIP(eae_stream_wait_event(st, c->ev_joinx[i]));
  }
  c->side_used = 0;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The caller's stream and the side streams must reach the GPU through DIFFERENT hardware queues: ROCm multiplexes a process's streams
// onto 4 hardware queues, which one a stream gets depends on the streams alive when it was created, and two streams that share a
// queue run one after the other (measured: the grouped B=64 step 0.73 instead of 0.62 ms in a process that had trained other
// contexts from worker threads before; bench.py's grid leg).  Checked once per (context, caller's stream) before the first step:
// a gate on stream A waits (bounded, 0.3 ms) for a word that a kernel enqueued AFTERWARDS on stream B publishes -- it times out exactly
// when B's kernel cannot start beside it.  A side stream that collides is replaced by a fresh one (created while the colliding one
// is still alive, so it lands elsewhere), up to 16 candidates.  EAE_STREAM_PROBE=0 switches the check off, =2 reports what it found.
// ---------------------------------------------------------------------------------------------------------------------
// (two device words: [0] the word the gate waits for, [1] its time-out flag -- not the sticky word the optimizer looks at)
bool streams_share_queue(unsigned* w, hipStream_t a, hipStream_t b) {
  if (a == b) return true;
  if (hipMemsetAsync(w, 0, 8, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
  GateArgs g = GateArgs();
  g.word[0] = w; g.want[0] = 1; g.n = 1; g.timeout = w + 1; g.limit_ticks = 30000ULL;      // 0.3 ms of the 100 MHz clock (a kernel that CAN start beside the gate does so within microseconds)
  if (eae_launch_gate(a, g) || eae_launch_signal(b, w, 1)) return false;
  hipStreamSynchronize(a); hipStreamSynchronize(b);
  unsigned to = 0;
  if (hipMemcpy(&to, w + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
  return to != 0;
}
bool streams_clash(unsigned* w, hipStream_t a, hipStream_t b) { return streams_share_queue(w, a, b) || streams_share_queue(w, b, a); }
// Streams that steps are running on, process-wide: the caller's streams of probed contexts, their side streams, and streams a
// driver has reserved for its worker threads (eae_reserve_stream: train.run_concurrent).  A context's side streams also keep clear of
// these -- two groups stepped from two threads use four streams, and the GPU has four hardware queues.  One probe at a time.
struct StreamRegistry {
  std::mutex mu;
  std::vector<std::pair<hipStream_t, const void*>> used;       // (stream, owner: a context, or nullptr for a reserved stream)
};
StreamRegistry& stream_registry() { static StreamRegistry r; return r; }
int streams_distinct(eae_ctx* c, hipStream_t user) {
  static const int mode = getenv("EAE_STREAM_PROBE") ? atoi(getenv("EAE_STREAM_PROBE")) : 1;
  if (mode == 0 || !c->use_side || !c->use_gates || c->capturing || c->streams_exposed || eae_rec) return 0;
  const long long now_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  c->last_step_ns.store(now_ns, std::memory_order_relaxed);
  if (c->probed && c->probed_user == user) return 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(user, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;     // (not inside somebody's capture)
  StreamRegistry& reg = stream_registry();
  std::lock_guard<std::mutex> lock(reg.mu);
  c->probed = true; c->probed_user = user;
  EAE_HIP(hipDeviceSynchronize());
  unsigned* w = c->sigwords + 14;
  // forget this context's earlier claims (a new caller's stream), collect the others'
  for (size_t i = reg.used.size(); i-- > 0;) if (reg.used[i].second == c) reg.used.erase(reg.used.begin() + i);
  std::vector<hipStream_t> others;
  for (const auto& u : reg.used) {
    if (u.first == user) continue;
    const eae_ctx* o = static_cast<const eae_ctx*>(u.second);        // (a context that has not stepped for half a second is not in anybody's way)
    if (o && now_ns - o->last_step_ns.load(std::memory_order_relaxed) > 500000000LL) continue;
    others.push_back(u.first);
  }
  const int ns = 1 + c->nx;
  int replaced = 0, left = 0, left_other = 0;
  std::vector<hipStream_t> drop;
  for (int k = 0; k < ns; ++k) {
    hipStream_t* slot = k == 0 ? &c->side : &c->sidex[k - 1];
    for (int attempt = 0; attempt < 16; ++attempt) {
      bool clash = streams_clash(w, *slot, user);
      for (int j = 0; j < k && !clash; ++j) clash = streams_clash(w, *slot, j == 0 ? c->side : c->sidex[j - 1]);
      bool clash_other = false;
      for (size_t j = 0; j < others.size() && !clash && !clash_other; ++j) clash_other = streams_clash(w, *slot, others[j]);
      if (!clash && !clash_other) break;
      // (with the others' streams the four queues may simply be taken: after 12 candidates only collisions inside the context count.
      //  Four were not enough: two groups stepped from two threads need the ONE queue the other three streams leave free, and the
      //  runs in which the second group gave up early measured 0.72 instead of 0.94 M images/s)
      if (!clash && attempt >= 11) { left_other++; break; }
      if (attempt == 15) { left++; break; }
      hipStream_t fresh = nullptr;
      EAE_HIP(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, c->side_prio));
      drop.push_back(*slot);            // destroyed at the end: while it lives, the next candidate goes to another queue
      *slot = fresh;
      replaced++;
    }
  }
  for (hipStream_t st : drop) hipStreamDestroy(st);
  reg.used.emplace_back(user, c);
  for (int k = 0; k < ns; ++k) reg.used.emplace_back(k == 0 ? c->side : c->sidex[k - 1], c);
  if (mode >= 2) fprintf(stderr, "[eae] stream probe: %d side stream(s) replaced, %d still share a hardware queue inside the context, %d with another context's (%zu other streams in use)\n", replaced, left, left_other, others.size());
  return 0;
}
void streams_forget(const eae_ctx* c) {
  StreamRegistry& reg = stream_registry();
  std::lock_guard<std::mutex> lock(reg.mu);
  for (size_t i = reg.used.size(); i-- > 0;) if (reg.used[i].second == c) reg.used.erase(reg.used.begin() + i);
}

unsigned* poison_word(const eae_ctx* c) { return reinterpret_cast<unsigned*>(c->acc_base + c->poison_off); }
int ensure_packed(eae_ctx* c, hipStream_t st) {
  if (c->packed) return 0;
  // (the pack kernel also clears the step-wide non-finite word: the optimizer kernel that clears the accumulators READS that word)
  // (the four latent-projection packs are 96 % of the elements at 256x256 inputs -- 16.7 M each: four times the workgroups there)
  static const int pack_blocks_big = getenv("EAE_PACK_BLOCKS") ? atoi(getenv("EAE_PACK_BLOCKS")) : 1024;
  // EAE_PACK_FLAT=0: the 2-D launch of rounds 1-4 (the same number of workgroups for every descriptor; A/B switch)
  static const bool flat = !(getenv("EAE_PACK_FLAT") && atoi(getenv("EAE_PACK_FLAT")) == 0);
  if (flat && c->blk_tot > 0) RC(eae_launch_pack_flat(st, c->descs_dev, c->blkmap, 0, c->blk_tot, c->P, c->pack, c->fp8 ? c->q : nullptr, poison_word(c)));
  else
  RC(eae_launch_pack_all(st, c->descs_dev, c->ndesc, c->P, c->pack, c->fp8 ? c->q : nullptr, poison_word(c),
                         (long long)c->K * c->Lp >= (1LL << 22) ? pack_blocks_big : 256));
  c->packed = true;
  return 0;
}

// split-K of the latent projections: K-range per slice.  128 (32 slices) at the reference's 64x64 inputs; wider inputs keep the number
// of slices at EAE_FC_SLICES (default 64): K / 128 = 512 slices at 256x256 wrote and re-read 67 MB of partials per projection (ms per
// config-5 step with 512 / 128 / 64 / 32 / 16 slices: 2.066 / 2.050 / 2.040 / 2.058 / 2.107, with the prefetching K loop of fc_nt_kernel)
int fc_klen(const eae_ctx* c) {
  static const int slices = getenv("EAE_FC_SLICES") ? atoi(getenv("EAE_FC_SLICES")) : 64;
  long long klen = 128;
  while (c->K / klen > slices && c->K % (klen * 2) == 0) klen *= 2;
  return (int)klen;
}
SrcDesc src_raw(const bf16_t* p) { SrcDesc s; s.p0 = p; s.p1 = nullptr; s.coef = nullptr; return s; }
SrcDesc src_bnrelu(const bf16_t* y, const float* coef) { SrcDesc s; s.p0 = y; s.p1 = nullptr; s.coef = coef; return s; }
SrcDesc src_bnbwd(const bf16_t* g, const bf16_t* y, const float* coef) { SrcDesc s; s.p0 = g; s.p1 = y; s.coef = coef; return s; }
SrcDesc src_f32(const float* p) { SrcDesc s; s.p0 = reinterpret_cast<const bf16_t*>(p); s.p1 = nullptr; s.coef = nullptr; return s; }

// the [C] sticky non-finite flag words of BN layer l sit behind the layer's [acc_copies][2][C] accumulators (forward or backward half)
unsigned long long* acc_flag(const eae_ctx* c, unsigned long long* acc, int l) { return acc + (size_t)c->acc_copies[l] * 2 * BN_C[l]; }
constexpr float ACC_SCALE_FWD = 16777216.f;      // 2^24: sums of y and y^2 over <= 2^21 elements of |y| <~ 1e3 stay far below 2^63

// fp8 variant: weight pack, scales and the amax word of 3x3 layer j (W3 order) for a forward / backward-data launch
void fp8_conv_args(eae_ctx* c, ConvArgs& a, int j, bool forward, bool p1) {
  if (!c->fp8) return;
  a.wpack = (const bf16_t*)(c->pack + (p1 ? c->pk8_p1[j] : c->pk8_p2[j]));
  a.qs = forward ? c->q->qs_fwd[j] : c->q->qs_bwd[j];
  a.amax = forward ? c->q->amax_act[j] : c->q->amax_grad[j];
  a.amax_mask = FP8_AMAX_SLOTS - 1; a.amax_stride = FP8_AMAX_STRIDE;
}

// producer side of the folded forward finalize of BN layer l
void fold_producer(eae_ctx* c, ConvArgs& a, int l, bool train) {
  if (!train || !c->fold_fwd) return;
  a.bacc.acc = c->accf[l]; a.bacc.copies = c->acc_copies[l]; a.bacc.scale = ACC_SCALE_FWD; a.bacc.flag = acc_flag(c, c->accf[l], l);
  a.stat_part = nullptr;
}
// consumer side: coefficient table of BN layer l from its accumulators
// SyncBN: sum layer l's forward accumulators over the replicas (order-independent integer sums: every replica ends with the
// same bits) between its producer and its first consumer
int sync_fwd(eae_ctx* c, hipStream_t st, int l, bool train) {
  if (!train || c->sync_world <= 1) return 0;
  if (!c->fold_fwd) return eae_set_error(EAE_ERR_STATE, "SyncBN needs the folded forward finalize (unset EAE_NO_FOLD_FWD)");
  const long long off = (long long)(c->accf[l] - reinterpret_cast<unsigned long long*>(c->acc_base));
  if (c->sync_fn(c->sync_user, 0, off, (long long)c->acc_copies[l] * 2 * BN_C[l] + BN_C[l], (void*)st) != 0)      // (+C: the non-finite flag words)
    return eae_set_error(EAE_ERR_STATE, "SyncBN: the exchange hook failed (forward statistics)");
  return 0;
}
void fold_consumer(eae_ctx* c, BnFold& f, int l, long long count, bool train) {
  f = BnFold();
  if (!train || !c->fold_fwd) return;
  count *= c->sync_world;
  f.acc = c->accf[l]; f.copies = c->acc_copies[l]; f.inv_scale = 1.0f / ACC_SCALE_FWD; f.count = (float)count; f.flag = acc_flag(c, c->accf[l], l);
  f.poison = poison_word(c);
  f.momentum = BN_MOM; f.eps = BN_EPS;
  f.gamma = c->P + c->poff[BN_GAMMA_IDX[l]]; f.beta = c->P + c->poff[BN_GAMMA_IDX[l] + 1];
  f.rm = c->bnrun + c->bnoff[2 * l]; f.rv = c->bnrun + c->bnoff[2 * l + 1]; f.nbt = c->nbt ? c->nbt + l : nullptr;
  f.coef_out = c->coef_f[l];
}

// the statistics accumulators of a train-mode forward must be zero when its producers start: the engine's own Adam clears them
// as a side job, any other sequence (forward only, external optimizer, encoder / decoder alone) pays one memset here
int prep_accumulators(eae_ctx* c, hipStream_t st, bool train) {
  if (!train || !c->fold_fwd) return 0;
  if (!c->acc_clean || c->capturing) { EAE_HIP(eae_memset_async(c->acc_base, 0, c->acc_bytes, st)); c->bwd_dirty = false; }   // a captured step always carries it
  c->acc_clean = false;
  return 0;
}
// ... and the BACKWARD accumulators when its producers start.  A train-mode forward with the folded finalize has just cleared the
// whole region (or the optimizer kernel did); what is left are the sequences that reach a backward without either: eval-mode
// backward after eval-mode backward (autograd with frozen statistics and an external optimizer), EAE_NO_FOLD_FWD -- their sums
// used to pile up (found by the EAE_NO_FOLD_FWD x fp8-calibration sweep: 8 gradient steps, gradients 92x too large)
int prep_bwd_accumulators(eae_ctx* c, hipStream_t st) {
  if (!c->fold_bwd) return 0;
  if (c->bwd_dirty) EAE_HIP(eae_memset_async(c->acc_base + c->acc_half, 0, c->poison_off - c->acc_half, st));
  c->bwd_dirty = true; c->acc_clean = false;
  return 0;
}

int bn_fwd_finalize(eae_ctx* c, hipStream_t st, int l, int ntiles, long long count, bool train) {
  if (train && c->fold_fwd) return 0;       // folded into the producer (accumulators) and the next kernel (prologue)
  const float* gamma = c->P + c->poff[BN_GAMMA_IDX[l]];
  const float* beta = c->P + c->poff[BN_GAMMA_IDX[l] + 1];
  float* rm = c->bnrun + c->bnoff[2 * l];
  float* rv = c->bnrun + c->bnoff[2 * l + 1];
  if (train) return eae_launch_bn_finalize(st, c->stat, ntiles, BN_C[l], count, gamma, beta, rm, rv, c->nbt ? c->nbt + l : nullptr, BN_MOM, BN_EPS, c->coef_f[l]);
  return eae_launch_bn_eval_coef(st, BN_C[l], gamma, beta, rm, rv, BN_EPS, c->coef_f[l]);
}

constexpr float ACC_SCALE_BWD = 4398046511104.f;   // 2^42 (eae_common.hip.h, BnBwdFold)
bool bwd_folded(const eae_ctx* c, int l) { (void)l; return c->fold_bwd && c->sync_world <= 1; }
int bwd_copies(const eae_ctx* c, int l) { return std::min(c->acc_copies[l], BN_FOLD_KB * (256 / BN_C[l])); }
// producer side of the folded BatchNorm-backward finalize of layer l: call before launching the kernel whose epilogue takes the sums
void fold_bwd_producer(eae_ctx* c, ConvArgs& a, int l) {
  if (!bwd_folded(c, l)) return;
  a.stat_part = nullptr;
  a.bacc.acc = c->accb[l]; a.bacc.copies = bwd_copies(c, l); a.bacc.scale = ACC_SCALE_BWD; a.bacc.flag = acc_flag(c, c->accb[l], l);
}
// consumer side: the kernels that read layer l's (g, y) pair with SRC_BNBWD build A, B, Cc from the accumulators; `writer`: the
// main-stream consumer, whose workgroup 0 also stores dgamma / dbeta and the table
void fold_bwd_consumer(eae_ctx* c, BnBwdFold& f, int l, long long count, bool writer) {
  f = BnBwdFold();
  if (!bwd_folded(c, l)) return;
  f.acc = c->accb[l]; f.copies = bwd_copies(c, l); f.inv_scale = 1.0f / ACC_SCALE_BWD; f.flag = acc_flag(c, c->accb[l], l);
  f.poison = poison_word(c);
  f.count = c->bwd_eval ? __builtin_inff() : (float)count;
  f.gamma = c->P + c->poff[BN_GAMMA_IDX[l]]; f.coef_fwd = c->coef_f[l];
  if (writer) {
    f.dgamma = c->G + c->poff[BN_GAMMA_IDX[l]]; f.dbeta = c->G + c->poff[BN_GAMMA_IDX[l] + 1]; f.coef_out = c->coef_b[l];
    if (c->bwd_eval) f.dbias = c->G + c->poff[PREBN_BIAS[l]];
  }
}

int bn_bwd_fin(eae_ctx* c, hipStream_t st, int l, int ntiles, long long count) {
  if (bwd_folded(c, l)) return 0;           // folded into the producer (accumulators) and its consumers (prologue)
  if (c->bwd_eval) count = 1LL << 40;       // eval-mode BatchNorm: no batch-size terms (see eae_ae_backward)
  if (c->sync_world > 1) {
    double* sums = c->sync_sums + (size_t)l * 512;
    RC(eae_launch_bn_bwd_reduce(st, c->stat, ntiles, BN_C[l], sums, c->G + c->poff[BN_GAMMA_IDX[l]], c->G + c->poff[BN_GAMMA_IDX[l] + 1]));
    if (c->sync_fn(c->sync_user, 1, (long long)l * 512, 2LL * BN_C[l], (void*)st) != 0)
      return eae_set_error(EAE_ERR_STATE, "SyncBN: the exchange hook failed (backward sums)");
    return eae_launch_bn_bwd_coef(st, sums, BN_C[l], count * c->sync_world, c->P + c->poff[BN_GAMMA_IDX[l]], c->coef_f[l], c->coef_b[l]);
  }
  return eae_launch_bn_bwd_finalize(st, c->stat, ntiles, BN_C[l], count, c->P + c->poff[BN_GAMMA_IDX[l]], c->coef_f[l],
                                    c->G + c->poff[BN_GAMMA_IDX[l]], c->G + c->poff[BN_GAMMA_IDX[l] + 1], c->coef_b[l]);
}

// latent-width padding (eae_ctx::Lp): copies between the caller's [B][L] tensors and the padded [B][Lp] workspace rows
int copy_latent_out(eae_ctx* c, hipStream_t st, float* dst, const float* src_padded, int B) {
  if (!c->lpad) { EAE_HIP(hipMemcpyAsync(dst, src_padded, (size_t)B * c->L * 4, hipMemcpyDeviceToDevice, st)); return 0; }
  EAE_HIP(hipMemcpy2DAsync(dst, (size_t)c->L * 4, src_padded, (size_t)c->Lp * 4, (size_t)c->L * 4, B, hipMemcpyDeviceToDevice, st));
  return 0;
}
int copy_latent_in(eae_ctx* c, hipStream_t st, float* dst_padded, const float* src, int B) {      // padding columns of dst stay as they are (zero)
  if (!c->lpad) { EAE_HIP(hipMemcpyAsync(dst_padded, src, (size_t)B * c->L * 4, hipMemcpyDeviceToDevice, st)); return 0; }
  EAE_HIP(hipMemcpy2DAsync(dst_padded, (size_t)c->Lp * 4, src, (size_t)c->L * 4, (size_t)c->L * 4, B, hipMemcpyDeviceToDevice, st));
  return 0;
}
const float* stage_latent_in(eae_ctx* c, hipStream_t st, const float* src, int B, int* rc) {
  *rc = 0;
  if (!c->lpad || !src) return src;
  hipError_t e = hipMemcpy2DAsync(c->zstage, (size_t)c->Lp * 4, src, (size_t)c->L * 4, (size_t)c->L * 4, B, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { *rc = eae_set_error(EAE_ERR_HIP, hipGetErrorString(e)); return nullptr; }
  return c->zstage;          // padding columns stay zero (cleared once at creation, never written)
}

// ---- encoder: x -> y[0..3] (raw, bf16) + BN coefficients -> z (fp32)
int run_encoder(eae_ctx* c, hipStream_t st, const float* x, int B, bool train) {
  const int H = c->H, W = c->W;
  {
    EdgeArgs a;
    a.src3 = x; a.B = B; a.H = H; a.W = W;
    a.c = ConvArgs();
    a.c.wpack = (const bf16_t*)(c->pack + c->pk_c1); a.c.bias = c->P + c->poff[1]; a.c.out = c->y[0];
    a.c.stat_part = train ? c->stat : nullptr; a.c.B = B;
    fold_producer(c, a.c, 0, train);
    {
      ProfBracket pb(c, EAE_PROF_SITE(0, 0), st);
      RC(eae_launch_edge_conv(st, SRC3_NCHW_F32, EPI_FWD, a));
    }
    RC(sync_fwd(c, st, 0, train));
    RC(bn_fwd_finalize(c, st, 0, eae_edge_tiles(B, H, W), (long long)B * (H / 2) * (W / 2), train));
  }
  for (int i = 1; i < 4; ++i) {
    ConvArgs a = ConvArgs();
    a.src = src_bnrelu(c->y[i - 1], c->coef_f[i - 1]);
    a.wpack = (const bf16_t*)(c->pack + c->pk_p1[i - 1]); a.bias = c->P + c->poff[4 * i + 1]; a.out = c->y[i];
    a.stat_part = train ? c->stat : nullptr;
    a.B = B; a.Hin = H >> i; a.Win = W >> i;
    fold_producer(c, a, i, train);
    fold_consumer(c, a.fold, i - 1, (long long)B * a.Hin * a.Win, train);
    fp8_conv_args(c, a, i - 1, true, true);
    {
      ProfBracket pb(c, EAE_PROF_SITE(i, 0), st);
      RC(eae_launch_conv_s2(a, ENC_C[i], ENC_C[i + 1], SRC_BNRELU, EPI_FWD, st));
    }
    RC(sync_fwd(c, st, i, train));
    RC(bn_fwd_finalize(c, st, i, eae_conv_s2_ntiles(0, B, a.Hin, a.Win, ENC_C[i]), (long long)B * (a.Hin / 2) * (a.Win / 2), train));
  }
  FcNtArgs f = FcNtArgs();
  f.a = src_bnrelu(c->y[3], c->coef_f[3]);
  f.w = (const bf16_t*)(c->pack + c->pk_we1);
  f.M = B; f.N = c->Lp; f.K = (int)c->K; f.klen = fc_klen(c); f.part = c->fcpart;
  fold_consumer(c, f.c.fold, 3, (long long)B * c->Pn, train);
  const int ksplit = (int)(c->K / f.klen);
  RC(eae_launch_fc_nt(st, f, SRC_BNRELU, FCE_PARTIAL, ksplit));
  RC(eae_launch_fc_reduce(st, c->fcpart, ksplit, B, c->Lp, c->lpad ? (const float*)(c->pack + c->pk_bep) : c->P + c->poff[17], nullptr,
                          nullptr, c->z));
  return 0;
}

// ---- decoder: z (fp32 [B][L]) -> d0, u[0..2] -> deconv4 + sigmoid (+ MSE and its gradient)

int run_decoder(eae_ctx* c, hipStream_t st, const float* z, int B, bool train, const float* target, float gscale, float* x_hat,
                bool want_grad, bool want_loss) {
  const int H = c->H, W = c->W;

  {
    FcNtArgs f = FcNtArgs();
    f.a = src_f32(z);
    f.w = (const bf16_t*)(c->pack + c->pk_wd1);
    f.M = B; f.N = (int)c->K; f.K = c->Lp; f.klen = c->Lp;
    f.c = ConvArgs();
    f.c.out = c->d0; f.c.bias = (const float*)(c->pack + c->pk_bd);
    take_sig(c, f.c);
    RC(eae_launch_fc_nt(st, f, SRC_F32, FCE_BIAS_BF16, 1));
    RC(sq_commit(c, st));          // the classification head (queued by forward_impl) starts beside the decoder
  }
  const int cin[3] = {256, 128, 64};
  for (int i = 0; i < 3; ++i) {
    ConvArgs a = ConvArgs();
    a.src = (i == 0) ? src_raw(c->d0) : src_bnrelu(c->u[i - 1], c->coef_f[3 + i]);
    a.wpack = (const bf16_t*)(c->pack + c->pk_p2[3 + i]); a.bias = c->P + c->poff[21 + 4 * i]; a.out = c->u[i];
    a.stat_part = train ? c->stat : nullptr;
    a.B = B; a.Hin = H >> (4 - i); a.Win = W >> (4 - i);
    fold_producer(c, a, 4 + i, train);
    if (i > 0) fold_consumer(c, a.fold, 3 + i, (long long)B * a.Hin * a.Win, train);
    fp8_conv_args(c, a, 3 + i, true, false);
    {
      ProfBracket pb(c, EAE_PROF_SITE(4 + i, 0), st);
      RC(eae_launch_deconv_s2(a, cin[i], cin[i] / 2, i == 0 ? SRC_RAW : SRC_BNRELU, EPI_FWD, st));
    }
    RC(sync_fwd(c, st, 4 + i, train));
    RC(bn_fwd_finalize(c, st, 4 + i, eae_conv_s2_ntiles(1, B, a.Hin, a.Win, cin[i]), (long long)B * (a.Hin * 2) * (a.Win * 2), train));
  }
  Deconv4Args d = Deconv4Args();
  d.src = src_bnrelu(c->u[2], c->coef_f[6]);
  d.wjoint = (const bf16_t*)(c->pack + c->pk_d4j); d.bias = c->P + c->poff[33];
  d.x = target; d.x_hat = x_hat; d.g4 = want_grad ? c->g4 : nullptr; d.loss_part = (want_loss || want_grad) ? c->msepart : nullptr;
  d.gscale = gscale; d.B = B; d.Hin = H / 2; d.Win = W / 2;
  fold_consumer(c, d.fold, 6, (long long)B * (H / 2) * (W / 2), train);
  {
    ProfBracket pb(c, EAE_PROF_DECONV4_LOSS, st);
    RC(eae_launch_deconv4_loss(st, SRC_BNRELU, d));
  }
  return 0;
}

int run_head(eae_ctx* c, hipStream_t st, int B, const long long* labels, float* logits, bool want_grad, const float* dlogits_in) {
  HeadArgs h = HeadArgs();
  h.dlogits_in = dlogits_in;
  h.z = c->z; h.w1 = c->lpad ? (const float*)(c->pack + c->pk_w1p) : c->P + c->poff[34]; h.b1 = c->P + c->poff[35]; h.w2 = c->P + c->poff[36]; h.b2 = c->P + c->poff[37];
  h.labels = labels; h.B = B; h.L = c->Lp; h.C = c->C; h.inv_batch = 1.0f / (float)B;
  h.logits = logits; h.dz = c->dzc; h.grad_part = want_grad ? c->headpart : nullptr; h.grad_stride = c->head_stride;
  h.loss_part = c->cepart;
  return eae_launch_head(st, h);
}

int check_io(eae_ctx* c, const eae_step_io* io, bool need_grad) {
  if (!c || !io) return eae_set_error(EAE_ERR_ARG, "ctx / io is NULL");
  if (!c->P || !c->bnrun) return eae_set_error(EAE_ERR_STATE, "eae_bind has not been called");
  if (!io->x) return eae_set_error(EAE_ERR_ARG, "io->x is NULL");
  if (io->B <= 0 || io->B > c->Bm) return eae_set_error(EAE_ERR_ARG, "batch size outside 1..max_batch");
  if (need_grad) {
    if (!c->G) return eae_set_error(EAE_ERR_STATE, "no gradient arena bound");
    if (io->head && !io->labels) return eae_set_error(EAE_ERR_ARG, "labels required when head=1");
    if (!io->train) return eae_set_error(EAE_ERR_ARG, "gradient step requires train=1 (BatchNorm batch statistics)");
  }
  return 0;
}

int forward_impl(eae_ctx* c, hipStream_t st, const eae_step_io* io, bool want_grad) {
  const int B = io->B;
  const bool train = io->train != 0;
  c->fwd_ready = train; c->fwd_eval_ready = !train; c->enc_ready = 0; c->dec_ready = 0; c->fwd_B = B; c->fwd_head = io->head; c->fwd_x = io->x; c->fwd_gen += 1;
  if (want_grad) c->last_loss = io->loss_last;
  RC(ensure_packed(c, st));
  RC(prep_accumulators(c, st, train));
  RC(run_encoder(c, st, io->x, B, train));
  const double numel = (double)B * 3.0 * c->H * c->W;
  const float gscale = (float)(2.0 * io->alpha / numel);
  const bool want_loss = io->loss_accum || io->loss_last;
  const bool head = io->head != 0;
  // The head only needs z: in a gradient step it runs on the side stream beside the decoder; the backward waits for it
  // (ev_head) right before dz_cls is added to dz.  (head_kernel is built without packed-FP32 instructions, see EAE_NO_PK.)
  if (head) {
    if (want_grad && c->head_side && c->use_side) {
      const long long* labels = io->labels;
      float* logits = io->logits;
      sq_push(c, [=](hipStream_t hs, float*) {
        RC(run_head(c, hs, B, labels, logits, true, nullptr));
        EAE_HIP(eae_event_record(c->ev_head, hs));
        return 0;
      }, 0);
      sq_fork(c);                  // released by the decoder's first kernel (run_decoder commits behind it)
      c->head_pending = true;
    } else {
      RC(run_head(c, st, B, io->labels, io->logits, want_grad, nullptr));
    }
  }
  RC(run_decoder(c, st, c->z, B, train, (want_loss || want_grad) ? io->x : nullptr, gscale, io->x_hat, want_grad, want_loss));
  if (io->z) RC(copy_latent_out(c, st, io->z, c->z, B));
  if (want_loss || want_grad) {
    const int n_ce = (head && io->labels) ? eae_head_blocks(B, c->Lp) : 0;
    // in a gradient step nothing on the main stream reads what this kernel writes (deconv4 bias gradient, loss scalars):
    // it goes to the side stream, which backward_impl joins before the optimizer
    if (want_grad && c->use_side) {
      const float alpha = io->alpha;
      float *accum = io->loss_accum, *last = io->loss_last;
      const int ntile = eae_edge_tiles(B, c->H, c->W);
      sq_push(c, [=](hipStream_t ls, float*) {
        return eae_launch_loss_finalize(ls, c->msepart, ntile, c->cepart, n_ce, alpha, numel, B, c->G + c->poff[33], accum, last, poison_word(c));
      }, 0);
      sq_fork(c);                  // released by the first kernel of the backward-data chain (backward_impl commits behind it)
    } else {
      RC(eae_launch_loss_finalize(st, c->msepart, eae_edge_tiles(B, c->H, c->W), c->cepart, n_ce, io->alpha, numel, B,
                                  want_grad ? c->G + c->poff[33] : nullptr, io->loss_accum, io->loss_last, poison_word(c)));
    }
  }
  return 0;
}

// part 0 = everything, 1 = classifier + decoder + dec.fc (gradient tensors 18..37), 2 = enc.fc + encoder (tensors 0..17)
// fused_lr != nullptr: the caller is the fused eager train step and will run the optimizer for tensors 0..7 behind this call; the bulk
// of the optimizer (tensors 8..37) is enqueued HERE on a side stream when the context allows it (c->bulk_done tells the caller)
int backward_impl(eae_ctx* c, hipStream_t st, const eae_step_io* io, const float* dz_ext = nullptr, int part = 0, const float* fused_lr = nullptr) {
  const int B = io->B, H = c->H, W = c->W;
  c->bulk_done = false;
  // (part 2 = the encoder half: behind part 1 of a split backward -- whose side-stream consumers may still be reading the decoder
  //  layers' sums -- nothing is cleared; the stand-alone encoder backward clears before it calls)
  if (part != 2) RC(prep_bwd_accumulators(c, st));
  if (c->prebn_dirty && !c->bwd_eval) {      // train mode again: those biases have an identically zero gradient, never written
    for (int k = 0; k < 7; ++k)
      EAE_HIP(eae_memset_async(c->G + c->poff[PREBN_BIAS[k]], 0, (size_t)(c->poff[PREBN_BIAS[k] + 1] - c->poff[PREBN_BIAS[k]]) * 4, st));
    c->prebn_dirty = false;
  }
  const bool head = io->head != 0;
  // Default: every weight gradient is handed over as soon as its inputs exist (9 records).  EAE_FORK_GROUPS=1 hands them over in
  // 5 groups instead: measured SLOWER (0.596 vs 0.570 ms at B=512) -- the records saved (~5 us each) cost less than what two
  // 12-wave weight-gradient kernels running back to back on the side streams take away from the backward-data chain.
  static const bool fork_every = getenv("EAE_FORK_GROUPS") == nullptr;
  const bool dp = part != 0 || c->dp_stream[0] != nullptr || c->dp_stream[1] != nullptr;
  // fork() = "this group may start now" when every weight gradient is handed over by itself
  auto fork_if_every = [&]() { if (fork_every) sq_fork(c); };
  if (part != 2) {
    // ---- first group (side stream 0, behind the loss bookkeeping forward_impl queued): classifier weight gradients from the
    //      head kernel's partials, deconv4's weight gradient.  Released by the first kernel of the backward-data chain.
    c->side_rr = 1;                // the round robin of the later groups starts at side stream #1
    if (head) {
      sq_push(c, [=](hipStream_t ss, float*) {
        const int nb = eae_head_blocks(B, c->Lp);
        launch_reduce_slices(ss, c->headpart, nb, (long)(c->head_stride / 4), c->lpad ? c->gs_head : c->G + c->poff[34], 1.0f);
        EAE_LAUNCH_CHECK();
        if (c->lpad) {     // padded shadow -> arena: classifier.0.weight [128][L], then bias / classifier.2 (contiguous)
          EAE_NO_GROUP("a latent width that needs the padded classifier shadow");
          EAE_HIP(hipMemcpy2DAsync(c->G + c->poff[34], (size_t)c->L * 4, c->gs_head, (size_t)c->Lp * 4, (size_t)c->L * 4, 128,
                                   hipMemcpyDeviceToDevice, ss));
          EAE_HIP(hipMemcpyAsync(c->G + c->poff[35], c->gs_head + 128LL * c->Lp, (size_t)(c->poff[38] - c->poff[35]) * 4,
                                 hipMemcpyDeviceToDevice, ss));
        }
        return 0;
      }, 0);
    } else {
      sq_push(c, [=](hipStream_t ss, float*) {
        EAE_HIP(eae_memset_async(c->G + c->poff[34], 0, (size_t)(c->poff[38] - c->poff[34]) * 4, ss));
        return 0;
      }, 0);
    }
    sq_push(c, [=](hipStream_t ss, float* scr) {
      return eae_launch_edge_wgrad(ss, SRC3_NHWC4_BF16, c->g4, B, H, W, src_bnrelu(c->u[2], c->coef_f[6]), SRC_BNRELU, scr,
                                   c->wscratch_floats, c->G + c->poff[32], prof_hook_for(c, EAE_PROF_SITE(7, 2)));
    }, 0);
    sq_fork(c);
    // ---- deconv4: backward-data into u[2]'s BN+ReLU
    {
      EdgeArgs a;
      a.src3 = c->g4; a.B = B; a.H = H; a.W = W;
      a.c = ConvArgs();
      a.c.wpack = (const bf16_t*)(c->pack + c->pk_d4k); a.c.out = c->gu[2]; a.c.stat_part = c->stat;
      a.c.yprev = c->u[2]; a.c.prev_coef = c->coef_f[6]; a.c.B = B;
      fold_bwd_producer(c, a.c, 6);
      take_sig(c, a.c);
      {
        ProfBracket pb(c, EAE_PROF_DECONV4_BWD, st);
        RC(eae_launch_edge_conv(st, SRC3_NHWC4_BF16, EPI_MASK, a));
      }
      RC(sq_commit(c, st));
      RC(bn_bwd_fin(c, st, 6, eae_edge_tiles(B, H, W), (long long)B * (H / 2) * (W / 2)));
    }
    // ---- deconv3, deconv2, deconv1 (i = 2, 1, 0): weight gradients queued, handed over together before the last dgrad
    const int dcin[3] = {256, 128, 64};
    for (int i = 2; i >= 0; --i) {
      const int cs = dcin[i], cb = dcin[i] / 2;         // deconv weight [cs][cb][3][3]
      const int Hs = H >> (4 - i), Ws = W >> (4 - i);   // input (small) map of the deconv
      // dy mode: backward-data first -- while it stages dy = BatchNorm-backward(g, y) of this layer's output it also stores it
      // (dy_out); the weight gradient queued behind it reads that one tensor and is released when the NEXT kernel of the chain
      // starts.  Otherwise the weight gradient transforms g and y itself and is released together with the backward-data kernel.
      const bool dym = (c->dy_mask >> (4 + i)) & 1u;
      auto push_wgrad = [&]() {
        if (c->skip_wgrad) return;
        sq_push(c, [=](hipStream_t s2, float* scr) {
          WgradArgs w = WgradArgs();
          w.small = (i == 0) ? src_raw(c->d0) : src_bnrelu(c->u[i - 1], c->coef_f[3 + i]);
          w.big = dym ? src_raw(c->dyu[i]) : src_bnbwd(c->gu[i], c->u[i], c->coef_b[4 + i]);
          w.B = B; w.Hs = Hs; w.Ws = Ws;
          if (!dym) fold_bwd_consumer(c, w.bfold, 4 + i, (long long)B * (Hs * 2) * (Ws * 2), false);
          if (c->fp8) w.qs = c->q->qs_wg[3 + i];
          return eae_launch_wgrad_s2(s2, w, cs, cb, i == 0 ? SRC_RAW : SRC_BNRELU, dym ? SRC_RAWG : SRC_BNBWD, scr, c->wscratch_floats,
                                     c->G + c->poff[20 + 4 * i], prof_hook_for(c, EAE_PROF_SITE(4 + i, 2)));
        });
        if (i == 0) sq_fork(c); else fork_if_every();
      };
      if (!dym) push_wgrad();
      ConvArgs a = ConvArgs();
      a.src = src_bnbwd(c->gu[i], c->u[i], c->coef_b[4 + i]);
      a.dy_out = dym ? c->dyu[i] : nullptr;
      a.wpack = (const bf16_t*)(c->pack + c->pk_p1[3 + i]);
      a.B = B; a.Hin = Hs * 2; a.Win = Ws * 2;
      fold_bwd_consumer(c, a.bfold, 4 + i, (long long)B * a.Hin * a.Win, true);
      fp8_conv_args(c, a, 3 + i, false, true);
      take_sig(c, a);
      if (i > 0) {
        a.out = c->gu[i - 1]; a.stat_part = c->stat; a.yprev = c->u[i - 1]; a.prev_coef = c->coef_f[3 + i];
        fold_bwd_producer(c, a, 3 + i);
        {
          ProfBracket pb(c, EAE_PROF_SITE(4 + i, 1), st);
          RC(eae_launch_conv_s2(a, cb, cs, SRC_BNBWD, EPI_MASK, st));
        }
        if (c->sq_forked) RC(sq_commit(c, st));
        RC(bn_bwd_fin(c, st, 3 + i, eae_conv_s2_ntiles(0, B, a.Hin, a.Win, cb), (long long)B * Hs * Ws));
      } else {
        a.out = c->gd0;
        {
          ProfBracket pb(c, EAE_PROF_SITE(4, 1), st);
          RC(eae_launch_conv_s2(a, cb, cs, SRC_BNBWD, EPI_PLAIN, st));
        }
        if (c->sq_forked) RC(sq_commit(c, st));
      }
      if (dym) push_wgrad();
    }
    // ---- dec.fc: weight/bias gradient (queued: needs gd0) and dz
    sq_push(c, [=](hipStream_t s2, float*) {
      FcTnArgs t = FcTnArgs();
      t.p = src_raw(c->gd0); t.q = src_f32(c->z); t.Bt = B; t.I = (int)c->K; t.J = c->Lp;
      t.out = c->lpad ? c->gs_decw : c->G + c->poff[18]; t.colsum = c->G + c->poff[19]; t.out_mode = 0; t.Pn = (int)c->Pn;
      RC(eae_launch_fc_tn(s2, t, SRC_RAW, SRC_F32));
      if (c->lpad) EAE_HIP(hipMemcpy2DAsync(c->G + c->poff[18], (size_t)c->L * 4, c->gs_decw, (size_t)c->Lp * 4, (size_t)c->L * 4,
                                            (size_t)c->K, hipMemcpyDeviceToDevice, s2));
      return 0;
    });
    fork_if_every();
    {
      FcNtArgs f = FcNtArgs();
      f.a = src_raw(c->gd0); f.w = (const bf16_t*)(c->pack + c->pk_wd2);
      f.M = B; f.N = c->Lp; f.K = (int)c->K; f.klen = fc_klen(c); f.part = c->fcpart;
      f.c = ConvArgs();
      take_sig(c, f.c);
      const int ksplit = (int)(c->K / f.klen);
      RC(eae_launch_fc_nt(st, f, SRC_RAW, FCE_PARTIAL, ksplit));
      if (c->sq_forked) RC(sq_commit(c, st));
      if (c->head_pending) { EAE_HIP(eae_stream_wait_event(st, c->ev_head)); c->head_pending = false; }
      int src_rc = 0;
      const float* dze = stage_latent_in(c, st, dz_ext, B, &src_rc);      // caller's [B][L] gradient -> padded rows
      RC(src_rc);
      RC(eae_launch_fc_reduce(st, c->fcpart, ksplit, B, c->Lp, nullptr, head ? c->dzc : nullptr, dze, c->dz));
    }
    if (dp) RC(sq_commit(c, st));  // the hand-off below covers gradient tensors 18..37 only (ordered after `st` as it stands)
  }   // part != 2
  if (part == 1) return fold_side2(c);     // the caller may now all-reduce gradient tensors 18..37 behind the side stream
  if (part == 0 && c->dp_stream[0]) {      // same hand-off without splitting the call: see eae_dp_stream()
    RC(fold_side2(c));
    EAE_HIP(hipEventRecord(c->ev_part[0], c->side));
    EAE_HIP(hipStreamWaitEvent(c->dp_stream[0], c->ev_part[0], 0));
  }
  // ---- enc.fc: weight/bias gradient (queued with the dec.fc one: needs dz) and backward-data into y[3]'s BN+ReLU
  sq_push(c, [=](hipStream_t s2, float*) {
    FcTnArgs t = FcTnArgs();
    t.p = src_f32(c->dz); t.q = src_bnrelu(c->y[3], c->coef_f[3]); t.Bt = B; t.I = c->Lp; t.J = (int)c->K;
    t.out = c->lpad ? c->gs_encw : c->G + c->poff[16]; t.colsum = c->lpad ? c->gs_encb : c->G + c->poff[17]; t.out_mode = 1; t.Pn = (int)c->Pn;
    RC(eae_launch_fc_tn(s2, t, SRC_F32, SRC_BNRELU));
    if (c->lpad) {       // the first L rows of the padded shadows are the arena tensors
      EAE_HIP(eae_memcpy_d2d_async(c->G + c->poff[16], c->gs_encw, (size_t)c->L * c->K * 4, s2));
      EAE_HIP(eae_memcpy_d2d_async(c->G + c->poff[17], c->gs_encb, (size_t)c->L * 4, s2));
    }
    return 0;
  });
  sq_fork(c);
  {
    FcNtArgs f = FcNtArgs();
    f.a = src_f32(c->dz); f.w = (const bf16_t*)(c->pack + c->pk_we2);
    f.M = B; f.N = (int)c->K; f.K = c->Lp; f.klen = c->Lp;
    f.c = ConvArgs();
    f.c.out = c->gy[3]; f.c.stat_part = c->stat; f.c.yprev = c->y[3]; f.c.prev_coef = c->coef_f[3];
    fold_bwd_producer(c, f.c, 3);
    take_sig(c, f.c);
    RC(eae_launch_fc_nt(st, f, SRC_F32, FCE_MASK, 1));
    RC(sq_commit(c, st));
    RC(bn_bwd_fin(c, st, 3, ((B + 127) / 128) * (int)c->Pn, (long long)B * c->Pn));
  }
  // ---- conv4, conv3, conv2 (i = 3, 2, 1): weight gradient (queued; conv4 + conv3 go together, conv2 before the last dgrad so
  //      that it runs beside it and beside conv1's weight gradient) + backward-data
  for (int i = 3; i >= 1; --i) {
    const int cs = ENC_C[i + 1], cb = ENC_C[i];       // conv weight [cs][cb][3][3]
    const int Hs = H >> (i + 1), Ws = W >> (i + 1);   // output (small) map of the conv
    const bool dym = (c->dy_mask >> i) & 1u;         // (see the transposed layers above)
    // conv2 (the last hand-over of the step) with a split optimizer: its weight gradient is pinned to side stream 0, the bulk of the
    // optimizer to side stream 1, both released when conv2's backward-data kernel starts (= conv3's has finished: every gradient
    // tensor >= 8 the main chain writes is complete); a signal in FRONT of the weight gradient tells the bulk that side stream 0's
    // earlier members are done too.  Everything is joined behind conv1's weight gradient as before.
    const bool split = i == 1 && fused_lr != nullptr && part == 0 && !dp && c->split_opt && c->use_side && gates_now(c) && c->nx >= 1 &&
                       !c->fp8 && !c->skip_wgrad && !dym && c->ndesc_late > 0 && c->M && c->V;
    if (split) {
      c->bulk_seq += 1;
      RC(eae_launch_signal(side_stream(c, 0), c->sigwords + 10, c->bulk_seq));
      c->side_used |= 1u;
    }
    auto push_wgrad = [&]() {
      if (c->skip_wgrad) return;
      sq_push(c, [=](hipStream_t s2, float* scr) {
        WgradArgs w = WgradArgs();
        w.small = dym ? src_raw(c->dyy[i]) : src_bnbwd(c->gy[i], c->y[i], c->coef_b[i]);
        w.big = src_bnrelu(c->y[i - 1], c->coef_f[i - 1]);
        w.B = B; w.Hs = Hs; w.Ws = Ws;
        if (!dym) fold_bwd_consumer(c, w.bfold, i, (long long)B * Hs * Ws, false);
        if (c->fp8) w.qs = c->q->qs_wg[i - 1];
        return eae_launch_wgrad_s2(s2, w, cs, cb, dym ? SRC_RAWG : SRC_BNBWD, SRC_BNRELU, scr, c->wscratch_floats, c->G + c->poff[4 * i],
                                   prof_hook_for(c, EAE_PROF_SITE(i, 2)));
      }, split ? 0 : -1);
      sq_fork(c);              // released by the next kernel of the chain (dy mode, conv2: by conv1's weight gradient)
      if (split) {
        const float lr = *fused_lr;
        const unsigned seq = c->bulk_seq;
        sq_push(c, [=](hipStream_t s1, float*) {
          GateArgs g = GateArgs();
          g.word[0] = c->sigwords + 10; g.want[0] = seq; g.n = 1; g.timeout = c->sigwords + 8; g.limit_ticks = c->gate_limit;
          RC(eae_launch_gate(s1, g));
          const long long o8 = c->poff[8];
          // (few workgroups: the two kernels have ~60 us of slack beside conv2's backward-data and conv1's weight gradient, which are the
          //  critical chain -- at full width they took bandwidth and CUs from it and the step got 1 % SLOWER than with one optimizer launch)
          static const int ab = getenv("EAE_BULK_ADAM_BLOCKS") ? atoi(getenv("EAE_BULK_ADAM_BLOCKS")) : 128;
          static const int pb = getenv("EAE_BULK_PACK_BLOCKS") ? atoi(getenv("EAE_BULK_PACK_BLOCKS")) : 32;
          const bool big = (long long)c->K * c->Lp >= (1LL << 22);
          RC(eae_launch_adam_scaled(s1, c->P + o8, c->G + o8, c->M + o8, c->V + o8, c->poff[38] - o8, lr, 0.9, 0.999, 1e-8, 0.0, c->adam_step, 1.0f,
                                    nullptr, 0, c->sigwords + 8, poison_word(c), c->last_loss, c->nan_exact, big ? 2048 : ab));
          return eae_launch_pack_all(s1, c->descs_dev + c->ndesc_late, c->ndesc - c->ndesc_late, c->P, c->pack, nullptr, nullptr, big ? 1024 : pb);
        }, 1);
        c->bulk_done = true;
      }
    };
    if (!dym) push_wgrad();
    ConvArgs a = ConvArgs();
    a.src = src_bnbwd(c->gy[i], c->y[i], c->coef_b[i]);
    a.dy_out = dym ? c->dyy[i] : nullptr;
    a.wpack = (const bf16_t*)(c->pack + c->pk_p2[i - 1]);
    a.out = c->gy[i - 1]; a.stat_part = c->stat; a.yprev = c->y[i - 1]; a.prev_coef = c->coef_f[i - 1];
    a.B = B; a.Hin = Hs; a.Win = Ws;
    fold_bwd_producer(c, a, i - 1);
    fold_bwd_consumer(c, a.bfold, i, (long long)B * Hs * Ws, true);
    fp8_conv_args(c, a, i - 1, false, false);
    take_sig(c, a);
    {
      ProfBracket pb(c, EAE_PROF_SITE(i, 1), st);
      RC(eae_launch_deconv_s2(a, cs, cb, SRC_BNBWD, EPI_MASK, st));
    }
    if (c->sq_forked) RC(sq_commit(c, st));
    RC(bn_bwd_fin(c, st, i - 1, eae_conv_s2_ntiles(1, B, Hs, Ws, cs), (long long)B * (Hs * 2) * (Ws * 2)));
    if (dym) push_wgrad();
    if (i == 2 && part == 0 && c->dp_stream[1]) {     // enc.fc, conv4 and conv3 weight gradients: enqueued by the commit below
      RC(sq_commit(c, st));
      RC(fold_side2(c));
      EAE_HIP(hipEventRecord(c->ev_part[1], c->side));
      EAE_HIP(hipStreamWaitEvent(c->dp_stream[1], c->ev_part[1], 0));
    }
  }
  // ---- conv1 weight gradient: nothing is left for the main stream to do, so the last weight gradient runs there (no fork
  //      latency in the tail of the step) while the side streams drain
  BnBwdFold bf0;
  fold_bwd_consumer(c, bf0, 0, (long long)B * (H / 2) * (W / 2), true);
  {
    ConvArgs sg = ConvArgs();                          // carries the progress value that releases conv2's weight gradient
    take_sig(c, sg);
    // the join with the side streams rides in the tail of this weight gradient's slice reduction (one launch less in the step's
    // tail).  The side groups are committed BETWEEN the two launches: gates go behind the kernel that releases them (see sq_commit).
    static const bool tail_gate = !getenv("EAE_TAIL_GATE") || atoi(getenv("EAE_TAIL_GATE")) != 0;
    struct Mid { eae_ctx* c; hipStream_t st; } mid = {c, st};
    auto mid_fn = [](void* u, GateArgs* g) { Mid* m = static_cast<Mid*>(u); return join_side_begin(m->c, m->st, g); };
    RC(eae_launch_edge_wgrad(st, SRC3_NCHW_F32, io->x, B, H, W, src_bnbwd(c->gy[0], c->y[0], c->coef_b[0]), SRC_BNBWD, c->wscratch_main,
                             2048LL * 864, c->G + c->poff[0], prof_hook_for(c, EAE_PROF_CONV1_WGRAD), &bf0, sg.sig, sg.sig_val,
                             tail_gate ? +mid_fn : nullptr, &mid));
  }
  RC(join_side(c, st));                                 // (nothing left to wait for when the gate went with the reduction)
  if (c->fp8) RC(eae_launch_fp8_scales(st, c->q));      // every reader of this step's scales has finished: derive the next step's
  // Biases in front of a BatchNorm have an identically zero gradient (the reference computes ~1e-9 rounding noise);
  // their slots in the gradient arena are zeroed once in eae_bind and never written.
  return 0;
}

}  // namespace

extern "C" int eae_ae_forward(eae_ctx* c, void* stream, const eae_step_io* io) {
  RC(check_io(c, io, false));
  return forward_impl(c, (hipStream_t)stream, io, false);
}

// Backward of the most recent TRAIN-mode eae_ae_forward for externally supplied output gradients (the autograd path:
// `loss.backward()` on a torch loss built from x_hat / logits / z, R.md:649-653).  Activations of that forward are still
// resident in the workspace; the INPUT batch is not (conv1's weight gradient reads it): the caller passes it again, together
// with the generation id of the forward it differentiates.  Any of dlogits / dz may be NULL (= zero).
extern "C" long long eae_forward_generation(eae_ctx* c) { return c ? c->fwd_gen : -1; }
extern "C" int eae_ae_backward(eae_ctx* c, void* stream, long long generation, const float* x, const float* x_hat, const float* dx_hat,
                               const float* dlogits, const float* dz) {
  if (!c || !c->G) return eae_set_error(EAE_ERR_STATE, "backward: no gradient arena bound");
  if (!c->fwd_ready && !c->fwd_eval_ready) return eae_set_error(EAE_ERR_STATE, "backward: no forward is resident (call eae_ae_forward first)");
  if (generation != c->fwd_gen)
    return eae_set_error(EAE_ERR_STATE, "backward: a later forward has replaced the activations of the forward being differentiated "
                                        "(one backward per forward, in order)");
  if (!x || !x_hat || !dx_hat) return eae_set_error(EAE_ERR_ARG, "backward: x, x_hat and dx_hat are required");
  hipStream_t st = (hipStream_t)stream;
  const int B = c->fwd_B;
  eae_step_io io = eae_step_io();
  io.x = x; io.B = B; io.train = 1; io.head = (dlogits != nullptr) ? 1 : 0;
  RC(eae_launch_sigmoid_bwd(st, x_hat, dx_hat, c->g4, c->msepart, B, c->H, c->W));
  const int nblk = (int)(((long long)B * c->H * c->W + 255) / 256);
  RC(eae_launch_loss_finalize(st, c->msepart, nblk, nullptr, 0, 0.f, 1.0, B, c->G + c->poff[33], nullptr, nullptr));
  if (dlogits) RC(run_head(c, st, B, nullptr, nullptr, true, dlogits));
  // An eval-mode forward normalises with the running statistics: y -> gamma*(y - rm)*invstd_r + beta is affine per channel, so its
  // backward is dy = gamma*invstd_r * g with dgamma = sum g*xhat_r, dbeta = sum g -- the same kernels with the batch-size terms
  // (B and C of the BatchNorm-backward transform, both ~ 1/count) switched off by an infinite count.
  c->bwd_eval = c->fwd_eval_ready;
  if (c->bwd_eval && !(c->fold_bwd && c->sync_world <= 1))
    return eae_set_error(EAE_ERR_STATE, "backward of an eval-mode forward needs the folded BatchNorm-backward finalize (no EAE_NO_FOLD_BWD, no SyncBN)");
  if (c->bwd_eval) c->prebn_dirty = true;
  const int rc = backward_impl(c, st, &io, dz);
  c->bwd_eval = false;
  c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0;
  return rc;
}

extern "C" int eae_ae_grad_step(eae_ctx* c, void* stream, const eae_step_io* io) {
  RC(check_io(c, io, true));
  hipStream_t st = (hipStream_t)stream;
  RC(streams_distinct(c, st));
  RC(forward_impl(c, st, io, true));
  return backward_impl(c, st, io);
}

// fp8 variant: settle the delayed scales before the first real step.  Every iteration is a gradient step without the optimizer (the
// fp8 packs are rebuilt with the current weight scale each time); a scale is right once the tensors upstream of it were computed with
// right scales, so the backward chain of 6 layers needs 7 iterations.  BatchNorm running statistics and num_batches_tracked are
// restored afterwards; the gradient arena holds the last iteration's gradients.
extern "C" int eae_fp8_calibrate(eae_ctx* c, void* stream, const eae_step_io* io, int iters) {
  RC(check_io(c, io, true));
  if (!c->fp8) return eae_set_error(EAE_ERR_STATE, "fp8_calibrate: the context was not created with quant = 1");
  if (iters <= 0) iters = 7;
  hipStream_t st = (hipStream_t)stream;
  eae_step_io t = *io;
  t.x_hat = nullptr; t.logits = nullptr; t.z = nullptr; t.loss_accum = nullptr; t.loss_last = nullptr;
  const size_t bn_bytes = (size_t)c->bnoff[14] * 4;
  EAE_HIP(hipMemcpyAsync(c->bn_save, c->bnrun, bn_bytes, hipMemcpyDeviceToDevice, st));
  if (c->nbt) EAE_HIP(hipMemcpyAsync(c->bn_save + c->bnoff[14], c->nbt, 7 * 8, hipMemcpyDeviceToDevice, st));
  for (int it = 0; it < iters; ++it) {
    c->packed = false;
    RC(forward_impl(c, st, &t, true));
    RC(backward_impl(c, st, &t));
  }
  EAE_HIP(hipMemcpyAsync(c->bnrun, c->bn_save, bn_bytes, hipMemcpyDeviceToDevice, st));
  if (c->nbt) EAE_HIP(hipMemcpyAsync(c->nbt, c->bn_save + c->bnoff[14], 7 * 8, hipMemcpyDeviceToDevice, st));
  c->packed = false; c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0;
  return 0;
}
// current scales: s_act[6], s_grad[6], s_w[6] (3x3 layers in the order conv2, conv3, conv4, deconv1, deconv2, deconv3); synchronises
extern "C" int eae_fp8_scales(eae_ctx* c, float* out18) {
  if (!c || !out18) return eae_set_error(EAE_ERR_ARG, "fp8_scales: NULL argument");
  if (!c->fp8) return eae_set_error(EAE_ERR_STATE, "fp8_scales: the context was not created with quant = 1");
  EAE_HIP(hipDeviceSynchronize());
  Fp8State h;
  EAE_HIP(hipMemcpy(&h, c->q, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 6; ++i) { out18[i] = h.s_act[i]; out18[6 + i] = h.s_grad[i]; out18[12 + i] = h.s_w[i]; }
  return 0;
}

extern "C" int eae_adam_step(eae_ctx* c, void* stream, float lr, float weight_decay) {
  if (!c || !c->P || !c->G || !c->M || !c->V) return eae_set_error(EAE_ERR_STATE, "adam: parameter, gradient and moment arenas must be bound");
  c->adam_step += 1;
  float* const ll = c->last_loss;
  c->last_loss = nullptr;            // the step's loss_last buffer is the caller's: it is written (NaN, when the update is refused) by THIS launch only
  RC(eae_launch_adam_scaled((hipStream_t)stream, c->P, c->G, c->M, c->V, c->poff[38], lr, 0.9, 0.999, 1e-8, weight_decay, c->adam_step, 1.0f,
                            c->acc_base, (long long)c->poison_off, c->sigwords + 8, poison_word(c), ll, c->nan_exact));
  c->packed = false; c->acc_clean = true; c->bwd_dirty = false;
  return 0;
}

// Data-parallel pieces: the gradient step in two halves so that the all-reduce of the first half's gradients (tensors
// 18..37: dec.fc, decoder, classifier) can run behind the side stream while the encoder half is still being computed.
extern "C" int eae_ae_grad_step_begin(eae_ctx* c, void* stream, const eae_step_io* io) {
  RC(check_io(c, io, true));
  hipStream_t st = (hipStream_t)stream;
  RC(forward_impl(c, st, io, true));
  return backward_impl(c, st, io, nullptr, 1);
}
extern "C" int eae_ae_grad_step_end(eae_ctx* c, void* stream) {
  if (!c || !c->fwd_ready) return eae_set_error(EAE_ERR_STATE, "grad_step_end without grad_step_begin");
  eae_step_io io = eae_step_io();
  io.x = c->fwd_x; io.B = c->fwd_B; io.train = 1; io.head = c->fwd_head;
  int rc = backward_impl(c, (hipStream_t)stream, &io, nullptr, 2);
  c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0;
  return rc;
}
extern "C" void* eae_side_stream(eae_ctx* c) { if (c) c->streams_exposed = true; return c ? (void*)c->side : nullptr; }
// Test / diagnostic access to the engine's workspace tensors of the most recent step (device synchronised first; bf16 NHWC, sized
// for the context's max_batch): kind 0 = y[idx] (idx 0..3), 1 = gy[idx], 2 = u[idx] (0..2), 3 = gu[idx], 4 = dyy[idx] (1..3), 5 = dyu[idx].
// Copies up to `bytes` to `host_dst`, returns the number of bytes copied or a negative status.
extern "C" long long eae_debug_read(eae_ctx* c, int kind, int idx, void* host_dst, long long bytes) {
  if (!c || !host_dst) return eae_set_error(EAE_ERR_ARG, "debug_read: null argument");
  const bool enc = kind == 0 || kind == 1 || kind == 4;
  if (kind < 0 || kind > 5 || idx < 0 || idx > (enc ? 3 : 2) || (kind == 4 && idx == 0)) return eae_set_error(EAE_ERR_ARG, "debug_read: no such tensor");
  const bf16_t* p = kind == 0 ? c->y[idx] : kind == 1 ? c->gy[idx] : kind == 2 ? c->u[idx] : kind == 3 ? c->gu[idx] : kind == 4 ? c->dyy[idx] : c->dyu[idx];
  if (!p) return eae_set_error(EAE_ERR_STATE, "debug_read: this context keeps no dy tensor for that layer (EAE_DY_MASK)");
  const long long have = (long long)c->Bm * c->act_elems(enc ? idx + 1 : 3 - idx) * 2;
  if (bytes > have) bytes = have;
  EAE_HIP(hipDeviceSynchronize());
  EAE_HIP(hipMemcpy(host_dst, p, (size_t)bytes, hipMemcpyDeviceToHost));
  return bytes;
}
extern "C" void* eae_dp_stream(eae_ctx* c, int which) {
  if (!c || !c->use_side || which < 0 || which > 1) return nullptr;
  if (!c->dp_stream[which]) {
    if (hipStreamCreateWithFlags(&c->dp_stream[which], hipStreamNonBlocking) != hipSuccess) { c->dp_stream[which] = nullptr; return nullptr; }
    if (hipEventCreateWithFlags(&c->ev_part[which], EV_FLAGS) != hipSuccess) {
      hipStreamDestroy(c->dp_stream[which]); c->dp_stream[which] = nullptr; return nullptr;
    }
  }
  return (void*)c->dp_stream[which];
}
// optimizer.step() on gradients that are SUMS over `1/grad_scale` replicas (grad_scale = 1/world_size)
extern "C" int eae_adam_step_scaled(eae_ctx* c, void* stream, float lr, float weight_decay, float grad_scale) {
  if (!c || !c->P || !c->G || !c->M || !c->V) return eae_set_error(EAE_ERR_STATE, "adam: parameter, gradient and moment arenas must be bound");
  c->adam_step += 1;
  float* const ll = c->last_loss;
  c->last_loss = nullptr;            // (see eae_adam_step)
  RC(eae_launch_adam_scaled((hipStream_t)stream, c->P, c->G, c->M, c->V, c->poff[38], lr, 0.9, 0.999, 1e-8, weight_decay, c->adam_step, grad_scale,
                            c->acc_base, (long long)c->poison_off, c->sigwords + 8, poison_word(c), ll, c->nan_exact));
  c->packed = false; c->acc_clean = true; c->bwd_dirty = false;
  return 0;
}

// The plain eager step: Adam takes its bias-correction scalars by value (one launch less on the critical path).
static int train_step_eager(eae_ctx* c, hipStream_t st, const eae_step_io* io, float lr) {
  RC(streams_distinct(c, st));
  int rc = forward_impl(c, st, io, true);
  c->adam_step += 1;               // (the bulk optimizer inside backward_impl needs this step's count; taken back when the step fails)
  if (!rc) rc = backward_impl(c, st, io, nullptr, 0, &lr);
  bool packed = false;
  if (!rc && c->bulk_done) {
    // tensors 8..37 were updated and packed on a side stream beside the end of the backward (joined by now): the rest behind the join
    const long long n8 = c->poff[8];
    rc = eae_launch_adam_scaled(st, c->P, c->G, c->M, c->V, n8, lr, 0.9, 0.999, 1e-8, 0.0, c->adam_step, 1.0f,
                                c->acc_base, (long long)c->poison_off, c->sigwords + 8, poison_word(c), c->last_loss, c->nan_exact);
    if (!rc) rc = eae_launch_pack_all(st, c->descs_dev, c->ndesc_late, c->P, c->pack, nullptr, poison_word(c), 64);
    packed = rc == 0;
  } else if (!rc) {
    rc = eae_launch_adam_scaled(st, c->P, c->G, c->M, c->V, c->poff[38], lr, 0.9, 0.999, 1e-8, 0.0, c->adam_step, 1.0f,
                                c->acc_base, (long long)c->poison_off, c->sigwords + 8, poison_word(c), c->last_loss, c->nan_exact);
  }
  if (rc) c->adam_step -= 1;
  c->bulk_done = false;
  c->last_loss = nullptr;
  c->packed = packed; c->acc_clean = (rc == 0); c->bwd_dirty = !c->acc_clean;
  return rc;
}

// One iteration of the batch loop.  Steady state (same buffers, batch size and alpha as the previous calls, parameters
// last touched by this engine's own Adam): the whole step -- pack, forward, loss, backward on two streams, Adam -- is replayed
// from a captured hipGraph; the only per-step host work is one tiny launch that refreshes Adam's bias-correction scalars.
extern "C" int eae_ae_train_step(eae_ctx* c, void* stream, const eae_step_io* io, float lr) {
  RC(check_io(c, io, true));
  if (!c->M || !c->V) return eae_set_error(EAE_ERR_STATE, "adam: moment arenas must be bound");
  hipStream_t user = (hipStream_t)stream, st = user;
  const bool graph_ok = c->use_graph && (c->use_side || user != nullptr) && c->fold_fwd && c->fold_bwd && !c->prof_on && !c->packed && io->logits == nullptr && io->z == nullptr;
  if (graph_ok && user == nullptr) {      // legacy default stream: run on the engine's own stream, ordered by events
    st = c->own_main;
    EAE_HIP(hipEventRecord(c->ev_in, user));
    EAE_HIP(hipStreamWaitEvent(st, c->ev_in, 0));
  }
  struct Rejoin {                          // order the caller's stream after the step on every exit path
    eae_ctx* c; hipStream_t user, st;
    ~Rejoin() { if (st != user) { hipEventRecord(c->ev_out, st); hipStreamWaitEvent(user, c->ev_out, 0); } }
  } rejoin{c, user, st};
  eae_ctx::GraphEntry* ent = nullptr;
  if (graph_ok) {
    eae_ctx::GraphKey key{io->x, io->labels, io->x_hat, io->loss_accum, io->loss_last, io->B, io->head, io->alpha};
    for (int i = 0; i < c->ngraphs; ++i) if (c->graphs[i].key == key) ent = &c->graphs[i];
    if (!ent && c->ngraphs < eae_ctx::NGRAPH) { ent = &c->graphs[c->ngraphs++]; ent->key = key; }
    if (ent) ent->seen++;
  }
  if (!ent) return train_step_eager(c, st, io, lr);
  c->adam_step += 1;
  RC(eae_launch_set_dyn(st, c->dyn, lr, 0.9, 0.999, 0.0, c->adam_step));
  if (ent && ent->exec) {
    EAE_HIP(hipGraphLaunch(ent->exec, st));
    c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0; c->packed = false; c->acc_clean = false;
    return 0;
  }
  const bool capture = ent && ent->seen >= 3;      // two eager warm-up steps with this key first (lazy kernel attributes etc.)
  if (capture) EAE_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  c->capturing = capture;
  int rc = forward_impl(c, st, io, true);
  if (!rc) rc = backward_impl(c, st, io);
  if (!rc) rc = eae_launch_adam_dyn(st, c->P, c->G, c->M, c->V, c->poff[38], 0.9, 0.999, 1e-8, c->dyn, c->sigwords + 8, poison_word(c), c->last_loss);
  c->capturing = false;
  c->packed = false;
  if (capture) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return eae_set_error(EAE_ERR_HIP, hipGetErrorString(e));
    e = hipGraphInstantiate(&ent->exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(g); ent->exec = nullptr; return eae_set_error(EAE_ERR_HIP, hipGetErrorString(e)); }
    ent->graph = g;
    EAE_HIP(hipGraphLaunch(ent->exec, st));
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped train step (eae_group.h): K contexts of one shape -- the grid of (alpha, lr) configurations the reference trains one
// after the other at batch 64 (R.md:599-711) -- stepped by ONE sequence of launches.  Every member's step logic runs with the recorder
// installed (its own state advances exactly as in eae_ae_train_step's eager path), the K recordings are zipped and enqueued on the
// FIRST member's streams.  The members must agree in everything that shapes the launches (configuration, batch size, which
// outputs are requested); what they need not share: parameters, statistics, inputs, labels, alpha, lr.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
int group_slot_of(void* ctx, hipStream_t user, hipStream_t st) {
  eae_ctx* c = static_cast<eae_ctx*>(ctx);
  if (st == user) return 0;
  if (st == c->side) return 1;
  for (int i = 0; i < c->nx; ++i) if (st == c->sidex[i]) return 2 + i;
  return -1;
}
hipStream_t group_stream(eae_ctx* c, hipStream_t user, int slot) { return slot == 0 ? user : slot == 1 ? c->side : c->sidex[slot - 2]; }

}  // namespace

// Do two streams reach the GPU through the same hardware queue (1), through different ones (0)?  Synchronises the device; < 0: error.
extern "C" int eae_streams_share_queue(void* a, void* b) {
  static thread_local unsigned* w = nullptr;
  static thread_local int w_dev = -1;
  int dev = 0;
  EAE_HIP(hipGetDevice(&dev));
  if (!w || w_dev != dev) { EAE_HIP(hipMalloc(reinterpret_cast<void**>(&w), 64)); w_dev = dev; }      // (a few bytes per thread and device, kept)
  StreamRegistry& reg = stream_registry();
  std::lock_guard<std::mutex> lock(reg.mu);
  EAE_HIP(hipDeviceSynchronize());
  return streams_clash(w, (hipStream_t)a, (hipStream_t)b) ? 1 : 0;
}
// A driver that steps contexts from several threads reserves its worker streams (on = 1) before the first step: the contexts' side
// streams then keep clear of them too.  on = 0 releases.
extern "C" int eae_reserve_stream(void* stream, int on) {
  StreamRegistry& reg = stream_registry();
  std::lock_guard<std::mutex> lock(reg.mu);
  for (size_t i = reg.used.size(); i-- > 0;)
    if (reg.used[i].first == (hipStream_t)stream && reg.used[i].second == nullptr) reg.used.erase(reg.used.begin() + i);
  if (on) reg.used.emplace_back((hipStream_t)stream, nullptr);
  return 0;
}

extern "C" int eae_set_geometry_mult(int mult) {
  if (mult < 1 || mult > 64) return eae_set_error(EAE_ERR_ARG, "geometry_mult: 1..64");
  eae_geo_mult = mult;
  return 0;
}

namespace {
// what = 0: eae_ae_train_step's eager path (lrs required); 1: eae_ae_forward
int group_run(int what, eae_ctx* const* ctxs, int n, int mult, void* stream, const eae_step_io* ios, const float* lrs) {
  if (!ctxs || !ios || (what == 0 && !lrs) || n < 1 || n > 64) return eae_set_error(EAE_ERR_ARG, "group call: 1..64 contexts, one io block (and one lr) each");
  if (mult == 0) mult = n;
  if (mult < n || mult > 64) return eae_set_error(EAE_ERR_ARG, "group call: geometry_mult must be 0 (= n) or n..64");
  if (eae_rec) return eae_set_error(EAE_ERR_STATE, "group call: already recording on this thread");
  hipStream_t user = (hipStream_t)stream;
  eae_ctx* c0 = ctxs[0];
  for (int k = 0; k < n; ++k) {
    eae_ctx* c = ctxs[k];
    RC(check_io(c, &ios[k], what == 0));
    if (what == 0 && (!c->M || !c->V)) return eae_set_error(EAE_ERR_STATE, "adam: moment arenas must be bound");
    if (c->prof_on || c->fp8 || c->dp_comm || c->use_gates != c0->use_gates || c->use_side != c0->use_side || c->nx != c0->nx)
      return eae_set_error(EAE_ERR_STATE, "group call: members must share the stream layout, with profiling, fp8 and data parallel off");
    for (int j = 0; j < k; ++j) if (ctxs[j] == c) return eae_set_error(EAE_ERR_ARG, "group call: a context appears twice");
  }
  RC(streams_distinct(c0, user));
  static thread_local std::vector<GroupRec> recs;
  if ((int)recs.size() < n) recs.resize(n);
  static const bool timing = getenv("EAE_GROUP_TIMING") != nullptr;      // diagnostic: host microseconds of the phases, every 100th call
  static thread_local double t_acc[3] = {0, 0, 0};
  static thread_local int t_n = 0;
  const auto tp0 = std::chrono::steady_clock::now();
  // (recording a member costs ~3 us of host time -- the step logic without its launches; a thread pool that recorded the members side
  //  by side was measured slower than this loop: waking a worker costs more than the work it takes)
  const int mult0 = eae_geo_mult;
  eae_geo_mult = mult;
  for (int k = 0; k < n; ++k) {
    GroupRec& r = recs[k];
    r.clear();
    r.ctx = ctxs[k]; r.user = user; r.slot_of = &group_slot_of;
    eae_rec = &r;
    const int rc = what == 0 ? train_step_eager(ctxs[k], user, &ios[k], lrs[k]) : forward_impl(ctxs[k], user, &ios[k], false);
    eae_rec = nullptr;
    if (rc && !r.error) r.error = rc;
    if (r.error) r.msg = eae_last_error();
  }
  eae_geo_mult = mult0;
  // (on a failure the members have advanced their host-side state all the same: the group is unusable, as a context is after a failed step)
  for (int k = 0; k < n; ++k) if (recs[k].error) return eae_set_error(recs[k].error, recs[k].msg.c_str());
  const auto tp1 = std::chrono::steady_clock::now();
  // zip
  const size_t len = recs[0].items.size();
  for (int k = 1; k < n; ++k) {
    if (recs[k].items.size() != len) return eae_set_error(EAE_ERR_STATE, "group call: the members' launch sequences differ in length (different shapes or state)");
    for (size_t i = 0; i < len; ++i) {
      const GroupItem& a = recs[0].items[i];
      const GroupItem& b = recs[k].items[i];
      if (a.kind != b.kind || a.slot != b.slot || a.kg != b.kg || (a.kind == GroupItem::LAUNCH &&
          (a.grid.x != b.grid.x || a.grid.y != b.grid.y || a.grid.z != b.grid.z || a.block.x != b.block.x || a.smem != b.smem || a.arg_size != b.arg_size)))
        return eae_set_error(EAE_ERR_STATE, "group call: the members' launch sequences differ (different shapes or state)");
    }
  }
  const auto tp2 = std::chrono::steady_clock::now();
  const unsigned char* argv[64];
  for (size_t i = 0; i < len; ++i) {
    const GroupItem& a = recs[0].items[i];
    hipStream_t st = group_stream(c0, user, a.slot);
    switch (a.kind) {
      case GroupItem::LAUNCH:
        for (int k = 0; k < n; ++k) argv[k] = recs[k].argbuf.data() + recs[k].items[i].arg_off;
        if (a.fn(a.kg, a.grid, a.block, a.smem, st, argv, n)) return eae_set_error(EAE_ERR_HIP, "group call: a grouped launch failed");
        break;
      case GroupItem::EV_RECORD: EAE_HIP(hipEventRecord(a.ev, st)); break;
      case GroupItem::EV_WAIT: EAE_HIP(hipStreamWaitEvent(st, a.ev, 0)); break;
      case GroupItem::OP:
        for (int k = 0; k < n; ++k) if (int e = recs[k].items[i].op(st)) return eae_set_error(EAE_ERR_HIP, "group call: a copy / memset failed"), e;
        break;
    }
  }
  if (timing) {
    const auto tp3 = std::chrono::steady_clock::now();
    t_acc[0] += std::chrono::duration<double, std::micro>(tp1 - tp0).count();
    t_acc[1] += std::chrono::duration<double, std::micro>(tp2 - tp1).count();
    t_acc[2] += std::chrono::duration<double, std::micro>(tp3 - tp2).count();
    if (++t_n == 100) {
      fprintf(stderr, "[eae group] n=%d items=%zu  record %.1f us  zip %.1f us  enqueue %.1f us\n", n, len, t_acc[0] / 100, t_acc[1] / 100, t_acc[2] / 100);
      t_acc[0] = t_acc[1] = t_acc[2] = 0; t_n = 0;
    }
  }
  return 0;
}
}  // namespace

extern "C" int eae_group_train_step(eae_ctx* const* ctxs, int n, int geometry_mult, void* stream, const eae_step_io* ios, const float* lrs) {
  return group_run(0, ctxs, n, geometry_mult, stream, ios, lrs);
}
extern "C" int eae_group_forward(eae_ctx* const* ctxs, int n, int geometry_mult, void* stream, const eae_step_io* ios) {
  return group_run(1, ctxs, n, geometry_mult, stream, ios, nullptr);
}

extern "C" int eae_encoder_forward(eae_ctx* c, void* stream, const float* x, int B, int train, float* z) {
  if (!c || !x || !z) return eae_set_error(EAE_ERR_ARG, "encoder_forward: NULL argument");
  if (!c->P || !c->bnrun) return eae_set_error(EAE_ERR_STATE, "eae_bind has not been called");
  if (B <= 0 || B > c->Bm) return eae_set_error(EAE_ERR_ARG, "batch size outside 1..max_batch");
  hipStream_t st = (hipStream_t)stream;
  c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0;
  RC(ensure_packed(c, st));
  RC(prep_accumulators(c, st, train != 0));
  RC(run_encoder(c, st, x, B, train != 0));
  c->enc_ready = train ? 1 : 2; c->fwd_B = B; c->fwd_gen += 1;
  return copy_latent_out(c, st, z, c->z, B);
}

namespace {
int half_backward_checks(eae_ctx* c, int ready, long long generation) {
  if (!c || !c->G) return eae_set_error(EAE_ERR_STATE, "backward: no gradient arena bound");
  if (!ready) return eae_set_error(EAE_ERR_STATE, "backward: the matching stand-alone forward is not resident");
  if (generation != c->fwd_gen)
    return eae_set_error(EAE_ERR_STATE, "backward: a later forward has replaced the activations of the forward being differentiated");
  if (ready == 2 && !(c->fold_bwd && c->sync_world <= 1))
    return eae_set_error(EAE_ERR_STATE, "backward of an eval-mode forward needs the folded BatchNorm-backward finalize (no EAE_NO_FOLD_BWD, no SyncBN)");
  return 0;
}
}  // namespace

// Backward of a stand-alone Encoder (z = enc(x); ... ; z.backward(dz)): the encoder half of the gradient step for an externally supplied
// dL/dz [B][L].  x = the forward's input batch (conv1's weight gradient reads it), generation as for eae_ae_backward.
extern "C" int eae_encoder_backward(eae_ctx* c, void* stream, long long generation, const float* x, const float* dz) {
  RC(half_backward_checks(c, c ? c->enc_ready : 0, generation));
  if (!x || !dz) return eae_set_error(EAE_ERR_ARG, "encoder_backward: x and dz are required");
  hipStream_t st = (hipStream_t)stream;
  RC(copy_latent_in(c, st, c->dz, dz, c->fwd_B));
  eae_step_io io = eae_step_io();
  io.x = x; io.B = c->fwd_B; io.train = 1; io.head = 0;
  c->bwd_eval = c->enc_ready == 2;
  if (c->bwd_eval) c->prebn_dirty = true;
  RC(prep_bwd_accumulators(c, st));            // (a stand-alone backward: backward_impl leaves part 2 alone)
  const int rc = backward_impl(c, st, &io, nullptr, 2);
  c->bwd_eval = false;
  c->enc_ready = 0;
  return rc;
}

// Backward of a stand-alone Decoder (x_hat = dec(z); ... ; x_hat.backward(dx_hat)): decoder + dec.fc gradients and dL/dz -> dz_out [B][L].
extern "C" int eae_decoder_backward(eae_ctx* c, void* stream, long long generation, const float* x_hat, const float* dx_hat, float* dz_out) {
  RC(half_backward_checks(c, c ? c->dec_ready : 0, generation));
  if (!x_hat || !dx_hat) return eae_set_error(EAE_ERR_ARG, "decoder_backward: x_hat and dx_hat are required");
  hipStream_t st = (hipStream_t)stream;
  const int B = c->fwd_B;
  eae_step_io io = eae_step_io();
  io.B = B; io.train = 1; io.head = 0;
  RC(eae_launch_sigmoid_bwd(st, x_hat, dx_hat, c->g4, c->msepart, B, c->H, c->W));
  const int nblk = (int)(((long long)B * c->H * c->W + 255) / 256);
  RC(eae_launch_loss_finalize(st, c->msepart, nblk, nullptr, 0, 0.f, 1.0, B, c->G + c->poff[33], nullptr, nullptr));
  c->bwd_eval = c->dec_ready == 2;
  if (c->bwd_eval) c->prebn_dirty = true;
  int rc = backward_impl(c, st, &io, nullptr, 1);
  c->bwd_eval = false;
  c->dec_ready = 0;
  if (!rc) rc = join_side(c, st);
  if (!rc && dz_out) rc = copy_latent_out(c, st, dz_out, c->dz, B);
  return rc;
}

extern "C" int eae_decoder_forward(eae_ctx* c, void* stream, const float* z, int B, int train, float* x_hat) {
  if (!c || !x_hat || !z) return eae_set_error(EAE_ERR_ARG, "decoder_forward: NULL argument");
  if (!c->P || !c->bnrun) return eae_set_error(EAE_ERR_STATE, "eae_bind has not been called");
  if (B <= 0 || B > c->Bm) return eae_set_error(EAE_ERR_ARG, "batch size outside 1..max_batch");
  hipStream_t st = (hipStream_t)stream;
  c->fwd_ready = false; c->fwd_eval_ready = false; c->enc_ready = 0; c->dec_ready = 0;
  RC(ensure_packed(c, st));
  RC(prep_accumulators(c, st, train != 0));
  RC(copy_latent_in(c, st, c->z, z, B));       // resident for the backward (dec.fc's weight gradient reads z again)
  RC(run_decoder(c, st, c->z, B, train != 0, nullptr, 0.f, x_hat, false, false));
  c->dec_ready = train ? 1 : 2; c->fwd_B = B; c->fwd_gen += 1;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Data parallel over RCCL, driven by the engine (eae_dp_init / eae_dp_allreduce_bucket; 8e).  librccl is bound at
// run time with dlopen -- the copy torch has already loaded if there is one, so the process holds ONE RCCL --, libeae.so itself
// keeps no link-time dependency on it.  One communicator per context, one process per GPU.
//   eae_dp_unique_id   rank 0 draws the 128-byte id (the caller ships it to the other ranks by whatever means it has)
//   eae_dp_init        every rank: ncclCommInitRank on the context's device
//   eae_dp_allreduce_bucket   sum over the ranks of gradient-arena elements [off, off + count), in place, on `stream`
//   eae_ae_dp_train_step      forward + loss + backward + all-reduce + Adam(grad_scale = 1/world) enqueued by ONE call.  overlap=1:
//                      the decoder-side bucket (gradient tensors 18..37, 3.3 MB at latent 64) is all-reduced on the engine's
//                      hand-off stream as soon as those tensors are complete, while the encoder half of the backward still runs;
//                      the encoder-side bucket follows on the caller's stream after the join.  overlap=0: one all-reduce of the
//                      whole arena after the backward.
// ---------------------------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only; every symbol is resolved with dlsym
namespace {
struct RcclApi {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclBroadcast) broadcast = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
  // optional (older libraries lack them): grouped launches
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
  std::string err;
};
RcclApi* rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) if (!a.h) a.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process (torch's)
    for (const char* n : names) if (!a.h) a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!a.h) a.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!a.h) { a.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return a; }
    a.getUniqueId = reinterpret_cast<decltype(a.getUniqueId)>(dlsym(a.h, "ncclGetUniqueId"));
    a.commInitRank = reinterpret_cast<decltype(a.commInitRank)>(dlsym(a.h, "ncclCommInitRank"));
    a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(dlsym(a.h, "ncclCommDestroy"));
    a.allReduce = reinterpret_cast<decltype(a.allReduce)>(dlsym(a.h, "ncclAllReduce"));
    a.broadcast = reinterpret_cast<decltype(a.broadcast)>(dlsym(a.h, "ncclBroadcast"));
    a.getErrorString = reinterpret_cast<decltype(a.getErrorString)>(dlsym(a.h, "ncclGetErrorString"));
    if (!a.getUniqueId || !a.commInitRank || !a.commDestroy || !a.allReduce || !a.broadcast || !a.getErrorString) a.err = "librccl lacks a required symbol";
    a.groupStart = reinterpret_cast<decltype(a.groupStart)>(dlsym(a.h, "ncclGroupStart"));
    a.groupEnd = reinterpret_cast<decltype(a.groupEnd)>(dlsym(a.h, "ncclGroupEnd"));
    return a;
  }();
  return &api;
}
int rccl_fail(RcclApi* r, ncclResult_t e, const char* what) {
  std::string m = std::string(what) + ": " + (r->getErrorString ? r->getErrorString(e) : "rccl error");
  return eae_set_error(EAE_ERR_HIP, m.c_str());
}
#define RCCL(x, what) do { ncclResult_t e__ = (x); if (e__ != ncclSuccess) return rccl_fail(r, e__, what); } while (0)
}  // namespace

extern "C" int eae_dp_unique_id(void* id128) {
  if (!id128) return eae_set_error(EAE_ERR_ARG, "dp_unique_id: NULL");
  RcclApi* r = rccl_api();
  if (!r->err.empty()) return eae_set_error(EAE_ERR_STATE, r->err.c_str());
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  RCCL(r->getUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, 128);
  return 0;
}
extern "C" int eae_dp_init(eae_ctx* c, int rank, int world, const void* id128) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return eae_set_error(EAE_ERR_ARG, "dp_init: bad argument");
  if (c->dp_comm) return eae_set_error(EAE_ERR_STATE, "dp_init: this context already owns a communicator");
  RcclApi* r = rccl_api();
  if (!r->err.empty()) return eae_set_error(EAE_ERR_STATE, r->err.c_str());
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t comm = nullptr;
  // Time-bounded bring-up (EAE_DP_INIT_TIMEOUT_S, default 120; 0 = plain blocking call): ncclCommInitRank is a rendezvous of all ranks
  // and blocks for ever when a peer never arrives.  It runs on a helper thread and the caller waits for it with a deadline; on a
  // time-out the call returns an error (dp.py: every rank falls back together) and the helper is left behind, still blocked.
  // (Not ncclCommInitRankConfig(blocking = 0): that makes EVERY later call on the communicator non-blocking -- an all-reduce may then
  //  return ncclInProgress and has to be polled -- which the enqueue-only train step does not want.)
  static const double limit_s = getenv("EAE_DP_INIT_TIMEOUT_S") ? atof(getenv("EAE_DP_INIT_TIMEOUT_S")) : 120.0;
  if (limit_s > 0) {
    int dev = 0;
    EAE_HIP(hipGetDevice(&dev));
    struct Box { std::promise<ncclResult_t> done; ncclComm_t comm = nullptr; };
    auto box = std::make_shared<Box>();
    std::future<ncclResult_t> fut = box->done.get_future();
    auto fn = r->commInitRank;
    std::thread([box, fn, world, id, rank, dev]() {
      ncclResult_t e = ncclSystemError;
      if (hipSetDevice(dev) == hipSuccess) e = fn(&box->comm, world, id, rank);
      box->done.set_value(e);
    }).detach();
    if (fut.wait_for(std::chrono::duration<double>(limit_s)) != std::future_status::ready)
      return eae_set_error(EAE_ERR_STATE, "dp_init: the communicator did not come up within EAE_DP_INIT_TIMEOUT_S (a peer never joined?)");
    const ncclResult_t e = fut.get();
    if (e != ncclSuccess) return rccl_fail(r, e, "ncclCommInitRank");
    comm = box->comm;
  } else {
    RCCL(r->commInitRank(&comm, world, id, rank), "ncclCommInitRank");
  }
  c->dp_comm = comm; c->dp_rank = rank; c->dp_world = world;
  if (!c->ev_dp_done) EAE_HIP(hipEventCreateWithFlags(&c->ev_dp_done, EV_FLAGS));
  return 0;
}
extern "C" int eae_dp_world(eae_ctx* c) { return c ? c->dp_world : 0; }
extern "C" int eae_dp_destroy(eae_ctx* c) {
  if (!c || !c->dp_comm) return 0;
  RcclApi* r = rccl_api();
  hipDeviceSynchronize();
  if (r->commDestroy) r->commDestroy(static_cast<ncclComm_t>(c->dp_comm));
  c->dp_comm = nullptr; c->dp_world = 0;
  if (c->ev_dp_done) { hipEventDestroy(c->ev_dp_done); c->ev_dp_done = nullptr; }
  return 0;
}
extern "C" int eae_dp_allreduce_bucket(eae_ctx* c, void* stream, long long elem_off, long long count) {
  if (!c || !c->dp_comm) return eae_set_error(EAE_ERR_STATE, "dp_allreduce_bucket: eae_dp_init has not been called");
  if (!c->G || elem_off < 0 || count <= 0 || elem_off + count > c->poff[38]) return eae_set_error(EAE_ERR_ARG, "dp_allreduce_bucket: range outside the gradient arena");
  RcclApi* r = rccl_api();
  RCCL(r->allReduce(c->G + elem_off, c->G + elem_off, (size_t)count, ncclFloat, ncclSum, static_cast<ncclComm_t>(c->dp_comm), (hipStream_t)stream),
       "ncclAllReduce");
  return 0;
}
// identical replicas: `bytes` of any device buffer from rank `root` (parameters, running statistics, Adam moments at start-up)
extern "C" int eae_dp_broadcast(eae_ctx* c, void* stream, void* buf, long long bytes, int root) {
  if (!c || !c->dp_comm) return eae_set_error(EAE_ERR_STATE, "dp_broadcast: eae_dp_init has not been called");
  if (!buf || bytes <= 0 || root < 0 || root >= c->dp_world) return eae_set_error(EAE_ERR_ARG, "dp_broadcast: bad argument");
  RcclApi* r = rccl_api();
  RCCL(r->broadcast(buf, buf, (size_t)bytes, ncclChar, root, static_cast<ncclComm_t>(c->dp_comm), (hipStream_t)stream), "ncclBroadcast");
  c->packed = false;
  return 0;
}
// The "optimizer refuses to update" switches (a gate that timed out: sticky; a non-finite BatchNorm statistic: this step) are local to a
// rank, but a synchronous data-parallel step must take the decision for ALL ranks or the replicas diverge silently: every
// rank publishes (timeout | poison) != 0 into one device word, the word is max-reduced over the ranks next to the last gradient bucket
// (one grouped launch where the library has ncclGroupStart), and the optimizer kernel reads the REDUCED word.
static __global__ EAE_NO_PK void dp_flag_kernel(const unsigned* a, const unsigned* b, unsigned* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {      // out[0]: stale (gate time-out), out[1]: diverged (non-finite statistics) -- kept apart:
    out[0] = (a != nullptr && __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1u : 0u;      // the optimizer
    out[1] = (b != nullptr && __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1u : 0u;      // treats them differently
  }
}
// sigwords[12..13]: this rank's two flags, then the reduced ones (same words, all-reduced in place)
static unsigned* dp_flag_word(eae_ctx* c) { return c->sigwords + 12; }
extern "C" int eae_dp_local_bad(eae_ctx* c, void* stream, unsigned* out_dev) {
  if (!c || !out_dev) return eae_set_error(EAE_ERR_ARG, "dp_local_bad: NULL argument");
  hipLaunchKernelGGL(dp_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c->sigwords + 8, poison_word(c), out_dev);
  EAE_LAUNCH_CHECK();
  return 0;
}
// optimizer.step() of a data-parallel replica whose peers' flags were reduced by the caller (torch.distributed path): `peer_bad` is a
// pair of device words (stale, diverged), non-zero = some rank must not update -> no rank updates
extern "C" int eae_adam_step_dp(eae_ctx* c, void* stream, float lr, float weight_decay, float grad_scale, const unsigned* peer_bad) {
  if (!c || !c->P || !c->G || !c->M || !c->V) return eae_set_error(EAE_ERR_STATE, "adam: parameter, gradient and moment arenas must be bound");
  c->adam_step += 1;
  int rc = eae_launch_adam_scaled((hipStream_t)stream, c->P, c->G, c->M, c->V, c->poff[38], lr, 0.9, 0.999, 1e-8, weight_decay, c->adam_step,
                                  grad_scale, c->acc_base, (long long)c->poison_off, peer_bad ? peer_bad : c->sigwords + 8,
                                  peer_bad ? peer_bad + 1 : poison_word(c), c->last_loss, c->nan_exact);
  c->last_loss = nullptr;            // (see eae_adam_step)
  c->packed = false; c->acc_clean = (rc == 0); c->bwd_dirty = !c->acc_clean;
  return rc;
}

extern "C" int eae_ae_dp_train_step(eae_ctx* c, void* stream, const eae_step_io* io, float lr, int overlap) {
  RC(check_io(c, io, true));
  if (!c->dp_comm) return eae_set_error(EAE_ERR_STATE, "dp_train_step: eae_dp_init has not been called");
  if (!c->M || !c->V) return eae_set_error(EAE_ERR_STATE, "adam: moment arenas must be bound");
  hipStream_t st = (hipStream_t)stream;
  const bool ov = overlap != 0 && c->use_side;
  RC(streams_distinct(c, st));
  if (ov && !eae_dp_stream(c, 0)) return eae_set_error(EAE_ERR_HIP, "dp_train_step: cannot create the hand-off stream");
  RC(forward_impl(c, st, io, true));
  RC(backward_impl(c, st, io));             // with a hand-off stream: dp_stream[0] is now ordered after gradient tensors 18..37
  c->adam_step += 1;                        // (behind the launches that can fail: a failed step must not advance the bias correction)
  const long long cut = c->poff[18], total = c->poff[38];
  RcclApi* r = rccl_api();
  unsigned* flag = dp_flag_word(c);
  hipLaunchKernelGGL(dp_flag_kernel, dim3(1), dim3(64), 0, st, c->sigwords + 8, poison_word(c), flag);
  EAE_LAUNCH_CHECK();
  const bool grp = r->groupStart && r->groupEnd;
  if (ov) {
    RC(eae_dp_allreduce_bucket(c, c->dp_stream[0], cut, total - cut));
    EAE_HIP(hipEventRecord(c->ev_dp_done, c->dp_stream[0]));
  }
  if (grp) RCCL(r->groupStart(), "ncclGroupStart");
  {
    int rc = eae_dp_allreduce_bucket(c, st, 0, ov ? cut : total);
    if (!rc) {
      ncclResult_t e = r->allReduce(flag, flag, 2, ncclUint32, ncclMax, static_cast<ncclComm_t>(c->dp_comm), st);
      if (e != ncclSuccess) rc = rccl_fail(r, e, "ncclAllReduce (flag)");
    }
    if (grp) { ncclResult_t e = r->groupEnd(); if (!rc && e != ncclSuccess) rc = rccl_fail(r, e, "ncclGroupEnd"); }
    RC(rc);
  }
  if (ov) EAE_HIP(hipStreamWaitEvent(st, c->ev_dp_done, 0));
  RC(eae_launch_adam_scaled(st, c->P, c->G, c->M, c->V, c->poff[38], lr, 0.9, 0.999, 1e-8, 0.0, c->adam_step, 1.0f / (float)c->dp_world,
                            c->acc_base, (long long)c->poison_off, flag, flag + 1, c->last_loss, c->nan_exact));
  c->last_loss = nullptr;
  c->packed = false; c->acc_clean = true; c->bwd_dirty = false;
  return 0;
}

// ------------------------------------------------------------------------------------------- per-op wrappers
namespace {
SrcDesc to_src(const eae_src& s) {
  SrcDesc d; d.p0 = (const bf16_t*)s.p0; d.p1 = (const bf16_t*)s.p1; d.coef = s.coef; return d;
}
}  // namespace

#ifdef EAE_STAMPS
static unsigned long long* g_dbg = nullptr; static int g_dbg_block = 0;
extern "C" int eae_debug_set(void* p, int block) { g_dbg = (unsigned long long*)p; g_dbg_block = block; return 0; }
#endif
extern "C" int eae_op_conv_s2(void* stream, int kind, eae_src src, int cin, int cout, int B, int Hin, int Win, const void* wpack,
                              const float* bias, void* out, float* stat_part, int epilogue, const void* yprev, const float* prev_coef) {
  ConvArgs a = ConvArgs();
  a.src = to_src(src); a.wpack = (const bf16_t*)wpack; a.bias = bias; a.out = (bf16_t*)out; a.stat_part = stat_part;
  a.yprev = (const bf16_t*)yprev; a.prev_coef = prev_coef; a.B = B; a.Hin = Hin; a.Win = Win;
#ifdef EAE_STAMPS
  a.dbg = g_dbg; a.dbg_block = g_dbg_block;
#endif
  if (kind == 0) return eae_launch_conv_s2(a, cin, cout, src.mode, epilogue, (hipStream_t)stream);
  return eae_launch_deconv_s2(a, cin, cout, src.mode, epilogue, (hipStream_t)stream);
}
extern "C" int eae_op_conv_s2_ntiles(int kind, int cin, int B, int Hin, int Win) { return eae_conv_s2_ntiles(kind, B, Hin, Win, cin); }
// fp8 variant of the same op (16 x 8-tileable maps only): wpack = e4m3 bytes [cout][9][cin] of w * s_w;  qs (device) = {1/s_pixel,
// 1/(s_pixel * s_w)};  amax (device, may be NULL) receives max |staged pixel operand| as float bits (atomicMax)
extern "C" int eae_op_conv_s2_fp8(void* stream, int kind, eae_src src, int cin, int cout, int B, int Hin, int Win, const void* wpack_e4m3,
                                  const float* bias, void* out, float* stat_part, int epilogue, const void* yprev, const float* prev_coef,
                                  const float* qs, unsigned* amax) {
  if (!qs) return eae_set_error(EAE_ERR_ARG, "conv_s2_fp8: qs is NULL");
  ConvArgs a = ConvArgs();
  a.src = to_src(src); a.wpack = (const bf16_t*)wpack_e4m3; a.bias = bias; a.out = (bf16_t*)out; a.stat_part = stat_part;
  a.yprev = (const bf16_t*)yprev; a.prev_coef = prev_coef; a.B = B; a.Hin = Hin; a.Win = Win;
  a.qs = qs; a.amax = amax;
  if (kind == 0) return eae_launch_conv_s2(a, cin, cout, src.mode, epilogue, (hipStream_t)stream);
  return eae_launch_deconv_s2(a, cin, cout, src.mode, epilogue, (hipStream_t)stream);
}

extern "C" int eae_op_edge_conv(void* stream, int src3_kind, const void* src3, int B, int H, int W, const void* wpack, const float* bias,
                                void* out, float* stat_part, int epilogue, const void* yprev, const float* prev_coef) {
  EdgeArgs a;
  a.src3 = src3; a.B = B; a.H = H; a.W = W;
  a.c = ConvArgs();
  a.c.wpack = (const bf16_t*)wpack; a.c.bias = bias; a.c.out = (bf16_t*)out; a.c.stat_part = stat_part;
  a.c.yprev = (const bf16_t*)yprev; a.c.prev_coef = prev_coef; a.c.B = B;
  return eae_launch_edge_conv((hipStream_t)stream, src3_kind, epilogue, a);
}
extern "C" int eae_op_edge_wgrad(void* stream, int src3_kind, const void* src3, int B, int H, int W, eae_src side, float* scratch,
                                 long long scratch_floats, float* dw) {
  return eae_launch_edge_wgrad((hipStream_t)stream, src3_kind, src3, B, H, W, to_src(side), side.mode, scratch, scratch_floats, dw);
}
extern "C" int eae_op_deconv4_loss(void* stream, eae_src a3, int B, int Hin, int Win, const void* wjoint, const float* bias,
                                   const float* x, float gscale, float* x_hat, void* g4, float* loss_part) {
  Deconv4Args d = Deconv4Args();
  d.src = to_src(a3); d.wjoint = (const bf16_t*)wjoint; d.bias = bias; d.x = x; d.x_hat = x_hat; d.g4 = (bf16_t*)g4;
  d.loss_part = loss_part; d.gscale = gscale; d.B = B; d.Hin = Hin; d.Win = Win;
  return eae_launch_deconv4_loss((hipStream_t)stream, a3.mode, d);
}
extern "C" int eae_op_wgrad_s2(void* stream, eae_src small_src, eae_src big_src, int cs, int cb, int B, int Hs, int Ws, float* scratch,
                               long long scratch_floats, float* dw) {
  WgradArgs w = WgradArgs();
  w.small = to_src(small_src); w.big = to_src(big_src); w.B = B; w.Hs = Hs; w.Ws = Ws;
  return eae_launch_wgrad_s2((hipStream_t)stream, w, cs, cb, small_src.mode, big_src.mode, scratch, scratch_floats, dw);
}
// fp8 variant: qs (device) = {1/s_small, 1/s_big, 1/(s_small * s_big)}; the SRC_BNBWD operand is converted to e5m2, the other to e4m3
extern "C" int eae_op_wgrad_s2_fp8(void* stream, eae_src small_src, eae_src big_src, int cs, int cb, int B, int Hs, int Ws, float* scratch,
                                   long long scratch_floats, float* dw, const float* qs) {
  if (!qs) return eae_set_error(EAE_ERR_ARG, "wgrad_s2_fp8: qs is NULL");
  WgradArgs w = WgradArgs();
  w.small = to_src(small_src); w.big = to_src(big_src); w.B = B; w.Hs = Hs; w.Ws = Ws; w.qs = qs;
  return eae_launch_wgrad_s2((hipStream_t)stream, w, cs, cb, small_src.mode, big_src.mode, scratch, scratch_floats, dw);
}
extern "C" int eae_op_bn_finalize(void* stream, const float* stat_part, int ntiles, int C, long long count, const float* gamma,
                                  const float* beta, float* rm, float* rv, long long* nbt, float momentum, float eps, float* coef) {
  return eae_launch_bn_finalize((hipStream_t)stream, stat_part, ntiles, C, count, gamma, beta, rm, rv, nbt, momentum, eps, coef);
}
extern "C" int eae_op_bn_eval_coef(void* stream, int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                   float eps, float* coef) {
  return eae_launch_bn_eval_coef((hipStream_t)stream, C, gamma, beta, rm, rv, eps, coef);
}
extern "C" int eae_op_bn_bwd_finalize(void* stream, const float* stat_part, int ntiles, int C, long long count, const float* gamma,
                                      const float* coef_fwd, float* dgamma, float* dbeta, float* coef_bwd) {
  return eae_launch_bn_bwd_finalize((hipStream_t)stream, stat_part, ntiles, C, count, gamma, coef_fwd, dgamma, dbeta, coef_bwd);
}
extern "C" int eae_op_fc_splitk(void* stream, eae_src a, const void* w, int M, int N, int K, const float* bias, const float* addend,
                                float* scratch, long long scratch_floats, float* out) {
  if (a.mode != SRC_RAW && a.mode != SRC_BNRELU) return eae_set_error(EAE_ERR_ARG, "fc_splitk: source mode must be 0 or 1");
  if (K % 128 || (long long)(K / 128) * M * N > scratch_floats) return eae_set_error(EAE_ERR_ARG, "fc_splitk: K % 128 != 0 or scratch too small");
  FcNtArgs f = FcNtArgs();
  f.a = to_src(a); f.w = (const bf16_t*)w; f.M = M; f.N = N; f.K = K; f.klen = 128; f.part = scratch;
  RC(eae_launch_fc_nt((hipStream_t)stream, f, a.mode, FCE_PARTIAL, K / 128));
  return eae_launch_fc_reduce((hipStream_t)stream, scratch, K / 128, M, N, bias, addend, nullptr, out);
}
extern "C" int eae_op_fc_bias_bf16(void* stream, const float* a_f32, const void* w, int M, int N, int K, const float* bias, void* out) {
  FcNtArgs f = FcNtArgs();
  f.a = src_f32(a_f32); f.w = (const bf16_t*)w; f.M = M; f.N = N; f.K = K; f.klen = K;
  f.c = ConvArgs(); f.c.out = (bf16_t*)out; f.c.bias = bias;
  return eae_launch_fc_nt((hipStream_t)stream, f, SRC_F32, FCE_BIAS_BF16, 1);
}
extern "C" int eae_op_fc_wgrad(void* stream, int mode, eae_src p, eae_src q, int Bt, int I, int J, int Pn, float* dw, float* colsum) {
  FcTnArgs t = FcTnArgs();
  t.p = to_src(p); t.q = to_src(q); t.Bt = Bt; t.I = I; t.J = J; t.out = dw; t.colsum = colsum; t.out_mode = mode; t.Pn = Pn;
  if (mode == 0) return eae_launch_fc_tn((hipStream_t)stream, t, SRC_RAW, SRC_F32);
  if (mode == 1) return eae_launch_fc_tn((hipStream_t)stream, t, SRC_F32, SRC_BNRELU);
  return eae_set_error(EAE_ERR_ARG, "fc_wgrad: mode must be 0 (dec.fc) or 1 (enc.fc)");
}
static long long head_stride_of(int L, int C) { return r4(128LL * L) + 128 + r4(128LL * C) + r4(C); }
extern "C" long long eae_op_head_scratch_floats(int B, int L, int C) {
  return (long long)eae_head_blocks(B, L) * (head_stride_of(L, C) + 2) + 64;
}
extern "C" int eae_op_head_ce(void* stream, const float* z, const float* w1, const float* b1, const float* w2, const float* b2,
                              const long long* labels, int B, int L, int C, float* logits, float* dz, float* grads, float* loss2,
                              float* scratch, long long scratch_floats) {
  if (!z || !w1 || !b1 || !w2 || !b2 || !scratch) return eae_set_error(EAE_ERR_ARG, "head_ce: NULL argument");
  if (scratch_floats < eae_op_head_scratch_floats(B, L, C)) return eae_set_error(EAE_ERR_ARG, "head_ce: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const long long stride = head_stride_of(L, C);
  const int nb = eae_head_blocks(B, L);
  float* ce_part = scratch;                         // [nb][2]
  float* gpart = scratch + (((long long)nb * 2 + 3) & ~3LL);
  HeadArgs h = HeadArgs();
  h.z = z; h.w1 = w1; h.b1 = b1; h.w2 = w2; h.b2 = b2; h.labels = labels; h.B = B; h.L = L; h.C = C; h.inv_batch = 1.0f / (float)B;
  h.logits = logits; h.dz = dz; h.grad_part = (labels && grads) ? gpart : nullptr; h.grad_stride = stride; h.loss_part = ce_part;
  RC(eae_launch_head(st, h));
  if (labels && grads) {
    launch_reduce_slices(st, gpart, nb, (long)(stride / 4), grads, 1.0f);
    EAE_LAUNCH_CHECK();
  }
  if (labels && loss2) RC(eae_launch_ce_mean(st, ce_part, nb, B, loss2));
  return 0;
}
extern "C" int eae_op_sigmoid_bwd(void* stream, const float* x_hat, const float* dx_hat, int B, int H, int W, void* g4, float* db,
                                  float* scratch) {
  hipStream_t st = (hipStream_t)stream;
  RC(eae_launch_sigmoid_bwd(st, x_hat, dx_hat, g4, scratch, B, H, W));
  const int nblk = (int)(((long long)B * H * W + 255) / 256);
  return eae_launch_loss_finalize(st, scratch, nblk, nullptr, 0, 0.f, 1.0, B, db, nullptr, nullptr);
}
extern "C" int eae_op_pack3x3(void* stream, const float* w, int A, int B, void* p1, void* p2) {
  // one-off helper for tests: builds a 2-entry descriptor table on the fly (synchronous upload)
  PackDesc d[2];
  long long n = (long long)A * B * 9;
  d[0] = PackDesc{0, 0, n, PACK_3x3_P1, A, B, 0, 0, 0, -1};
  d[1] = PackDesc{0, (long long)((char*)p2 - (char*)p1), n, PACK_3x3_P2, A, B, 0, 0, 0, -1};
  PackDesc* dev = nullptr;
  EAE_HIP(hipMalloc(&dev, sizeof(d)));
  EAE_HIP(hipMemcpy(dev, d, sizeof(d), hipMemcpyHostToDevice));
  int rc = eae_launch_pack_all((hipStream_t)stream, dev, 2, w, p1);
  hipStreamSynchronize((hipStream_t)stream);
  hipFree(dev);
  return rc;
}
extern "C" int eae_augment(void* stream, const void* in_u8, float* out, int B, int H, int W, int train, float noise_std,
                           unsigned long long seed, unsigned long long step, const int* params, const float* noise) {
  return eae_launch_augment((hipStream_t)stream, in_u8, out, B, H, W, train, noise_std, seed, step, params, noise);
}
extern "C" int eae_op_adam(void* stream, float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2,
                           double eps, double wd, long long step) {
  return eae_launch_adam((hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, wd, step);
}
