/* eae.h -- C ABI of libeae.so: the MI355X (gfx950) engine behind the notebook's Encoder / Decoder /
 * SupervisedAutoencoder / MLP modules and its fit / evaluate loops.
 *
 * The reference has no FFI for this path: its boundary is the Python class surface of the notebook
 * (R.md = /root/reference/Report/Hybrid_autoencoder–MLP_pipeline_for_satellite_image_classification.md):
 *   Encoder.forward  R.md:312        Decoder.forward R.md:386-389      SupervisedAutoencoder.forward R.md:429-433
 *   AE train step    R.md:646-654    AE validation step R.md:673-677   MLP.forward R.md:2565
 *   MLP train step   R.md:2641-2646  extract_features R.md:2498-2510   final evaluate R.md:3171-3187
 * Each entry point below names the reference lines it replaces.  The Python shells in
 * hybrid-autoencoder-mlp-pipeline-for-satellite-image-classification_amd/ bind these with ctypes.
 *
 * Conventions: every function returns 0 on success and a negative code on error (message: eae_last_error());
 * no C++ exception crosses the boundary.  All tensor arguments are raw DEVICE pointers owned by the caller with the
 * layout stated per argument; `stream` is a hipStream_t passed as void*; all work is enqueued on it and nothing
 * synchronises except eae_destroy / eae_mlp_destroy.  A ctx is bound to the current device and used from one thread.
 */
#ifndef EAE_H
#define EAE_H
#ifdef __cplusplus
extern "C" {
#endif

#define EAE_OK 0
#define EAE_ERR_ARG (-2)     /* bad shape / null pointer / unsupported configuration */
#define EAE_ERR_HIP (-3)     /* a HIP runtime call failed */
#define EAE_ERR_STATE (-4)   /* call order (e.g. backward without forward, ctx not bound) */

const char* eae_last_error(void);
int eae_version(void);

/* ------------------------------------------------------------------ autoencoder engine ---------------------- */
typedef struct eae_ctx eae_ctx;

typedef struct eae_config {
  int latent_dim;    /* Encoder/Decoder/SupervisedAutoencoder(latent_dim)  R.md:288, 362, 417; any width in 1..256 (padded to a
                        multiple of 64 inside the packs and workspaces; the parameter / gradient arenas keep the reference shapes) */
  int num_classes;   /* SupervisedAutoencoder(num_classes=10)             R.md:417; 1..64 */
  int image_h;       /* 64 for EuroSAT; must be a multiple of 64 */
  int image_w;
  int max_batch;     /* workspaces are sized for this many images per call */
  int quant;         /* 0: bf16 operands.  1: BASELINE config 5's stress variant -- the GEMMs of the six 3x3 layers (forward,
                        backward-data, weight gradient) take fp8 operands on v_mfma_f32_16x16x32_{fp8,bf8}_{fp8,bf8}: weights and
                        activations OCP e4m3, gradients e5m2, per-tensor power-of-two scales with delayed scaling (eae_fp8_calibrate),
                        fp32 accumulation, everything else as with 0.  Needs image_h % 128 == 0 and image_w % 256 == 0. */
  int side_streams;  /* 0: default (two engine-owned side streams for the work that only feeds the optimizer: weight gradients, head,
                        loss bookkeeping); 1 or 2: that many; -1: none, every kernel goes to the caller's stream in dependency order.
                        A process reaches the GPU through 4 hardware queues: when SEVERAL contexts are stepped concurrently (a grid of
                        small configurations, train.run_concurrent) one stream per context lets four of them run side by side, three
                        streams per context share the four queues (measured at batch 64: 4 contexts 487 K vs 352 K images/s). */
} eae_config;

/* fp8 variant only.  eae_fp8_calibrate: `iters` (<= 0: 7) gradient steps WITHOUT optimizer on the given batch to settle the delayed
 * scales before the first real step (running statistics are restored, the gradient arena is overwritten).
 * eae_fp8_scales: the current scales, s_act[6], s_grad[6], s_w[6] for conv2, conv3, conv4, deconv1, deconv2, deconv3 (synchronises);
 * a scale multiplies a value before its conversion. */
struct eae_step_io;
int eae_fp8_calibrate(eae_ctx* ctx, void* stream, const struct eae_step_io* io, int iters);
int eae_fp8_scales(eae_ctx* ctx, float* out18);

#define EAE_AE_NPARAMS 38   /* model.parameters() order of SupervisedAutoencoder */
#define EAE_AE_NBN 7        /* BatchNorm2d layers: enc.encoder.{1,4,7,10}, dec.decoder.{2,5,8} */

/* Flat arena layout.  param_off[i] (i < 38) = element offset of the i-th tensor of named_parameters() in the fp32
 * parameter / gradient / Adam arenas, param_off[38] = arena length (multiple of 4; every tensor 16-byte aligned).
 * bn_off[2*l] / bn_off[2*l+1] = offsets of running_mean / running_var of BN layer l in the running-stat arena,
 * bn_off[14] = its length. */
int eae_ae_layout(const eae_config* cfg, long long* param_off, long long* bn_off);

/* Environment switches read by eae_create are listed in INTEGRATION.md.  One of them changes results: EAE_NAN_EXACT=1 -- a DIVERGED
 * train step (a non-finite BatchNorm statistic was consumed: the loss scalars read NaN) writes NaN into every parameter and both Adam
 * moments, which is what the reference's loop does to its model (loss.backward() on a non-finite loss gives every parameter a NaN
 * gradient, torch.optim.Adam propagates it: R.md:653-654; pinned by tests/golden/ae_nan_step_b8.npz).  Default: the optimizer refuses
 * the update and the last finite parameters stay (the run still reads as diverged through its NaN losses). */
int eae_create(const eae_config* cfg, eae_ctx** out);
int eae_destroy(eae_ctx* ctx);

/* Bind the caller-owned arenas (fp32 unless noted).  grads/adam_m/adam_v may be NULL for inference-only use.
 * bn_nbt: int64[7] num_batches_tracked. */
int eae_bind(eae_ctx* ctx, float* params, float* grads, float* adam_m, float* adam_v, float* bn_running,
             long long* bn_nbt);
/* The engine orders its side streams behind the caller's stream with one-wave GATE kernels that poll a device progress word
 * (no event record on the caller's stream; EAE_FORK_EVENTS=1 restores events).  Their spin is bounded by wall-clock time (30 s;
 * EAE_GATE_TIMEOUT_MS=<ms>, 0 = unbounded), so a caller's stream stalled for seconds in front of a step is simply waited for.
 * A gate that gives up sets a STICKY device word: from then on every optimizer kernel of the context leaves the parameters
 * untouched and writes NaN into the step's loss_last (the gradients of that step may come from stale activations).
 * eae_gate_timeouts synchronises the device and returns 0, or the value a gate gave up waiting for; eae_gate_timeouts_clear
 * resets the word once the caller has dealt with the failed step.  Replaces nothing in the reference (R.md:642-658 runs on one
 * in-order stream); it guards the engine's own side-stream concurrency. */
long long eae_gate_timeouts(eae_ctx* ctx);
long long eae_gate_timeouts_nosync(eae_ctx* ctx);   /* same word, no device synchronisation (the caller has synchronised its own stream) */
int eae_gate_timeouts_clear(eae_ctx* ctx);
/* Replay eae_ae_train_step from a captured hipGraph (from the third call with the same buffers, batch size and alpha) or enqueue it
 * eagerly (default; EAE_GRAPH=1 in the environment turns replay on at creation).  Replay pays when the HOST is the limit: several small
 * configurations stepped concurrently (train.py run_concurrent, R.md:599-711 at batch 64). */
int eae_set_graph(eae_ctx* ctx, int on);
/* The host changed parameter values (load_state_dict, optimizer outside the engine): repack before next use. */
int eae_params_changed(eae_ctx* ctx);
int eae_set_adam_step(eae_ctx* ctx, long long step);
long long eae_get_adam_step(eae_ctx* ctx);

typedef struct eae_step_io {
  const float* x;            /* [B,3,H,W] fp32 NCHW, the loader contract (R.md:643) */
  const long long* labels;   /* [B] int64 (R.md:644) or NULL */
  int B;
  int train;                 /* 1: model.train() semantics (batch statistics, running-stat update); 0: model.eval() */
  int head;                  /* 1: classifier head + CrossEntropy; 0: encoder+decoder only (MSE) */
  float alpha;               /* loss = alpha * MSE(x_hat, x) + CE(logits, labels)   R.md:649-651 */
  float* x_hat;              /* optional [B,3,H,W] fp32 NCHW */
  float* logits;             /* optional [B,num_classes] fp32 */
  float* z;                  /* optional [B,latent_dim] fp32 */
  float* loss_accum;         /* optional float[8]: += loss*B, mse*B, ce*B, B, #correct  (R.md:656-657, 679-681) */
  float* loss_last;          /* optional float[4]: loss, mse, ce of this call.  With a SEPARATE optimizer call (eae_ae_grad_step /
                              * eae_ae_backward followed by eae_adam_step*) the pointer is kept until that ONE optimizer launch, which
                              * writes NaN there when it refuses the update; it is dropped afterwards: keep the buffer alive until the
                              * optimizer call of the step has been enqueued */
} eae_step_io;

/* x_hat, logits, z = model(x) (R.md:647 / 673), plus the loss terms when io->x target / labels are given. */
int eae_ae_forward(eae_ctx* ctx, void* stream, const eae_step_io* io);
/* loss.backward() for a torch-side loss (R.md:649-653): backward of the most recent eae_ae_forward -- train mode, or eval mode
 * (io->train = 0: BatchNorm with the running statistics is differentiated as the per-channel affine map it then is; the biases in
 * front of the BatchNorms then get their gradient A[c] * sum g instead of zero) -- given the gradients of its outputs (fp32; dx_hat [B,3,H,W], dlogits [B,C] or NULL, dz [B,L] or NULL).  x = that forward's input batch
 * (conv1's weight gradient reads it again: the engine keeps no pointer to caller memory across calls), x_hat = its output,
 * generation = eae_forward_generation() read right after that forward: EAE_ERR_STATE if any forward ran since.
 * Gradients of all 38 tensors land in the grad arena (train mode: biases in front of a BatchNorm are exact zeros). */
long long eae_forward_generation(eae_ctx* ctx);
int eae_ae_backward(eae_ctx* ctx, void* stream, long long generation, const float* x, const float* x_hat, const float* dx_hat,
                    const float* dlogits, const float* dz);
/* zero_grad + forward + loss + backward (R.md:646-653): gradients of all 38 tensors land in the grad arena. */
int eae_ae_grad_step(eae_ctx* ctx, void* stream, const eae_step_io* io);
/* optimizer.step() of torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) over the bound arenas (R.md:624, 654). */
int eae_adam_step(eae_ctx* ctx, void* stream, float lr, float weight_decay);
/* Data-parallel training (new work, no reference counterpart): the gradient step in two halves.  After _begin the gradient
 * tensors 18..37 (dec.fc, decoder, classifier) are complete once the engine's side stream (eae_side_stream) has drained, so
 * their all-reduce can be enqueued behind that stream and overlap with _end (enc.fc + encoder).  eae_adam_step_scaled
 * multiplies the (summed) gradients by grad_scale = 1/world_size inside the optimizer kernel. */
int eae_ae_grad_step_begin(eae_ctx* ctx, void* stream, const eae_step_io* io);
int eae_ae_grad_step_end(eae_ctx* ctx, void* stream);
void* eae_side_stream(eae_ctx* ctx);
/* Test / diagnostic access to the workspace tensors of the most recent step (bf16 NHWC; the device is synchronised first):
 * kind 0 = raw conv outputs y[idx] (idx 0..3), 1 = their masked gradients gy[idx], 2 = raw transposed-conv outputs u[idx] (0..2),
 * 3 = gu[idx], 4 / 5 = the BatchNorm-backward-applied gradients dy the backward-data kernels store for the weight-gradient kernels
 * (idx 1..3 / 0..2).  Copies up to `bytes` into host memory; returns the bytes copied (negative: error).  No reference counterpart:
 * torch keeps these as autograd-internal buffers of loss.backward() (R.md:653). */
long long eae_debug_read(eae_ctx* ctx, int kind, int idx, void* host_dst, long long bytes);
/* The same hand-off without splitting the call (no host gap between the halves): once requested, engine-owned stream `which`
 * is, after every eae_ae_grad_step, ordered after the completion of gradient tensors 18..37 (which = 0: classifier, decoder,
 * dec.fc) or 8..17 (which = 1: enc.fc, conv4, conv3); a collective enqueued behind it overlaps the rest of that step's
 * backward.  Tensors 0..7 are complete on the caller's stream when the call returns.  NULL when side-stream concurrency is
 * disabled. */
void* eae_dp_stream(eae_ctx* ctx, int which);
int eae_adam_step_scaled(eae_ctx* ctx, void* stream, float lr, float weight_decay, float grad_scale);
/* The collective owned by the engine (SURVEY.md 8b "DP eae_dp_init(ctx, rank, world, ncclUniqueId), eae_dp_allreduce_bucket"; new
 * work: the reference has no parallelism).  One process per GPU; librccl is bound at run time (the copy already in the
 * process if any).  eae_dp_unique_id: rank 0 draws the 128-byte ncclUniqueId, the caller ships it to the other ranks;
 * eae_dp_init: every rank joins (collective call).  eae_dp_allreduce_bucket: in-place sum over the ranks of gradient-arena elements
 * [elem_off, elem_off + count) on `stream`.  eae_dp_broadcast: `bytes` of a device buffer from `root` (identical replicas at start).
 * eae_ae_dp_train_step: eae_ae_train_step for a replica -- forward, loss, backward, gradient all-reduce, Adam with the 1/world scale
 * folded into the optimizer kernel -- enqueued by one call with no host code between backward and collective; overlap != 0 sends
 * the decoder-side bucket (tensors 18..37) over the engine's hand-off stream while the encoder half of the backward computes. */
int eae_dp_unique_id(void* id128);
int eae_dp_init(eae_ctx* ctx, int rank, int world, const void* id128);
int eae_dp_world(eae_ctx* ctx);
int eae_dp_destroy(eae_ctx* ctx);
int eae_dp_allreduce_bucket(eae_ctx* ctx, void* stream, long long elem_off, long long count);
int eae_dp_broadcast(eae_ctx* ctx, void* stream, void* buf, long long bytes, int root);
int eae_ae_dp_train_step(eae_ctx* ctx, void* stream, const eae_step_io* io, float lr, int overlap);
/* The decision "this step must not update the parameters" (a side-stream gate timed out: sticky; a non-finite BatchNorm statistic:
 * this step) is taken for ALL replicas together, or they would diverge silently: eae_ae_dp_train_step max-reduces one flag word over
 * the ranks next to the last gradient bucket.  For an exchange the caller runs itself (torch.distributed): eae_dp_local_bad writes this
 * rank's TWO flags (stale, diverged; 0 / 1 each) to a pair of device words on `stream`; the caller max-reduces the pair and hands it to
 * eae_adam_step_dp, which is eae_adam_step_scaled with those words as the refusal switches (NULL: the local switches).  eae_dp_init is time-bounded
 * (EAE_DP_INIT_TIMEOUT_S, default 120 s: the rendezvous runs on a helper thread the caller waits for with a deadline; the communicator
 * stays a BLOCKING one): a peer that never joins yields an error, not a hang. */
int eae_dp_local_bad(eae_ctx* ctx, void* stream, unsigned* out_dev);
int eae_adam_step_dp(eae_ctx* ctx, void* stream, float lr, float weight_decay, float grad_scale, const unsigned* peer_bad);
/* Synchronized BatchNorm across data-parallel replicas (new work; SURVEY.md 8e: R ranks x B/R with SyncBN == 1 rank x B).
 * In train mode the engine calls `fn` once per BatchNorm layer in the forward (kind 0: `count` int64 fixed-point accumulators
 * starting at element `elem_offset` of acc_i64) and once per layer in the backward (kind 1: `count` fp64 sums at element
 * `elem_offset` of sums_f64): the hook all-reduces (SUM) that range over the replicas, enqueued on `stream`, and returns 0.
 * acc_i64: caller-owned device buffer of eae_sync_bn_acc_elems() int64 (zero-initialised); sums_f64: 7*2*256 fp64.
 * world <= 1 or fn NULL switches it off (per-replica statistics, the default: DDP semantics). */
typedef int (*eae_sync_fn)(void* user, int kind, long long elem_offset, long long count, void* stream);
long long eae_sync_bn_acc_elems(eae_ctx* ctx);
int eae_set_sync_bn(eae_ctx* ctx, int world, eae_sync_fn fn, void* user, void* acc_i64, void* sums_f64);
/* eae_ae_grad_step + eae_adam_step: one iteration of the reference's batch loop (R.md:642-658). */
int eae_ae_train_step(eae_ctx* ctx, void* stream, const eae_step_io* io, float lr);
/* The grid of independent configurations the reference trains one after the other at batch 64 (R.md:246, 599-711), n of them per
 * call: eae_ae_train_step's eager path (eae_group_train_step) or eae_ae_forward (eae_group_forward: the validation pass, R.md:670-682)
 * for ctxs[0..n) -- each with its own io block (inputs, labels, alpha, outputs) and lr -- enqueued as ONE sequence of grouped launches
 * on `stream` and ctxs[0]'s side streams (workgroup z of every launch works for member z with that member's own arguments).
 * geometry_mult: the launchers choose tile geometries and grids as for a batch of B * geometry_mult (0 = n; a caller whose group
 * shrinks -- early stopping -- keeps passing the original size so that a member's arithmetic does not depend on who else is still
 * training).  Each member's results are bitwise what the single-context call gives it under eae_set_geometry_mult(geometry_mult).
 * The members must have the same configuration, batch size and requested outputs, live on the current device, and have profiling,
 * fp8 and data parallel off; a mismatch is an error and leaves the members' host-side state advanced (as a failed step does).
 * n <= 64; launches carry 8 members at a time. */
int eae_group_train_step(eae_ctx* const* ctxs, int n, int geometry_mult, void* stream, const eae_step_io* ios, const float* lrs);
int eae_group_forward(eae_ctx* const* ctxs, int n, int geometry_mult, void* stream, const eae_step_io* ios);
/* Hardware queues.  ROCm multiplexes a process's streams onto 4 hardware queues and two streams on one queue run one after the other.
 * Before its first step on a caller's stream a context checks its side streams against that stream, against each other and against
 * every stream other contexts are stepping on, and replaces a side stream that collides (EAE_STREAM_PROBE=0: off, =2: verbose).
 * eae_streams_share_queue: the same check for two arbitrary streams (1 = one queue, 0 = different ones; synchronises the device);
 * eae_reserve_stream(stream, 1): a driver that steps contexts from several host threads announces its worker streams, so that the
 * contexts' side streams keep clear of them too; (stream, 0) releases. */
int eae_streams_share_queue(void* stream_a, void* stream_b);
int eae_reserve_stream(void* stream, int on);
/* Thread-local: launchers that choose a tile geometry or a grid by the size of the batch see batch * mult (1 = default).  Used by the
 * group-vs-alone parity tests; the group calls set it to their geometry_mult for their own duration. */
int eae_set_geometry_mult(int mult);
/* Encoder alone in the current mode (extract_features, R.md:2504: z = encoder(imgs)). */
int eae_encoder_forward(eae_ctx* ctx, void* stream, const float* x, int B, int train, float* z);
/* Decoder alone (Decoder.forward, R.md:386-389). */
int eae_decoder_forward(eae_ctx* ctx, void* stream, const float* z, int B, int train, float* x_hat);
/* loss.backward() through a stand-alone Encoder / Decoder (the notebook defines them as separate modules, R.md:287, 361; e.g.
 * x_hat = dec(enc(x)) with an MSE loss): backward of the most recent eae_encoder_forward / eae_decoder_forward (train or eval mode)
 * for an externally supplied gradient of its output.  generation = eae_forward_generation() right after that forward.
 * encoder: dz [B][L], x = the forward's input;  decoder: dx_hat [B,3,H,W], x_hat = the forward's output, dz_out [B][L] (may be NULL). */
int eae_encoder_backward(eae_ctx* ctx, void* stream, long long generation, const float* x, const float* dz);
int eae_decoder_backward(eae_ctx* ctx, void* stream, long long generation, const float* x_hat, const float* dx_hat, float* dz_out);

/* In-situ timing of ONE launch site inside real train steps (bench.py's `roofline` object): HIP events are recorded around that
 * launch, on the stream it goes to, for up to 64 steps; eae_profile_read2 synchronises them and returns the summed bracket time,
 * the summed time of an EMPTY bracket recorded right behind each timed one (what the two event records cost by themselves) and
 * the number of launches measured.  site 0 switches the timing off.  Kernel behind each site (rocprofv3 name): */
#define EAE_PROF_OFF 0
/* site = EAE_PROF_SITE(layer, role): layer 0-3 = enc.conv1-4, 4-7 = dec.deconv1-4; role 0 forward, 1 backward-data, 2 weight gradient
 * (layer 0 has no backward-data; layer 7's forward is the fused deconv4 + sigmoid + MSE kernel) */
#define EAE_PROF_SITE(layer, role) (16 + 3 * (layer) + (role))
#define EAE_PROF_CONV2_FWD EAE_PROF_SITE(1, 0)      /* igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 1, 0>   enc.conv2 forward */
#define EAE_PROF_CONV2_BWD EAE_PROF_SITE(1, 1)      /* igemm_s2_kernel<1, 64, 32, 32, 16, 8, 1, 2, 1>   enc.conv2 backward-data */
#define EAE_PROF_DECONV3_BWD EAE_PROF_SITE(6, 1)    /* igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 2, 1>   dec.deconv3 backward-data */
#define EAE_PROF_CONV2_WGRAD EAE_PROF_SITE(1, 2)    /* wgrad_s2_kernel<64, 32, 16, 8, 1, 2, 1>          enc.conv2 weight gradient */
#define EAE_PROF_DECONV3_WGRAD EAE_PROF_SITE(6, 2)  /* wgrad_s2_kernel<64, 32, 16, 8, 1, 1, 2>          dec.deconv3 weight gradient */
#define EAE_PROF_DECONV4_LOSS EAE_PROF_SITE(7, 0)   /* deconv4_loss_kernel<1>                           dec.deconv4 + sigmoid + MSE + gradient */
#define EAE_PROF_CONV1_WGRAD EAE_PROF_SITE(0, 2)    /* edge_wgrad_kernel<0, 2>                          enc.conv1 weight gradient */
#define EAE_PROF_DECONV4_BWD EAE_PROF_SITE(7, 1)    /* edge_conv_kernel<1, 1>                           dec.deconv4 backward-data */
#define EAE_PROF_DECONV3_FWD EAE_PROF_SITE(6, 0)    /* igemm_s2_kernel<1, 64, 32, 32, 16, 8, 1, 1, 0>   dec.deconv3 forward */
#define EAE_PROF_NSITES 40
int eae_profile_enable(eae_ctx* ctx, int site);
/* diagnostic: copy an internal fp32 buffer (0 z, 1 dz, 2 dz_head, 3 head partials, 4 CE partials) to dst (device) */
int eae_debug_copy(eae_ctx* ctx, int which, float* dst, long long n);
int eae_profile_read(eae_ctx* ctx, double* total_ms, long long* count);
/* same, plus the summed duration of an EMPTY event bracket recorded right after each timed one (the cost of the two event
 * records themselves; subtracting it gives the kernel's own duration, which is what rocprofv3 reports) */
int eae_profile_read2(eae_ctx* ctx, double* total_ms, double* empty_ms, long long* count);

/* ------------------------------------------------------------------ per-op entry points ---------------------- */
/* Building blocks of the fused step, exported for kernel-level parity tests.  Activations are NHWC bf16. */
typedef struct eae_src {
  const void* p0;      /* mode 0: tensor; 1: raw pre-BN tensor y; 2: masked gradient g; 3: fp32 tensor */
  const void* p1;      /* mode 2: raw pre-BN tensor y */
  const float* coef;   /* mode 1: [4][C] s,t,mean,invstd; mode 2: [3][C] A,B,C */
  int mode;            /* 0 raw, 1 BN-apply+ReLU on load, 2 BN-backward-apply on load, 3 fp32 */
} eae_src;

/* kind 0: 3x3 stride-2 pad-1 conv (aten::convolution of nn.Conv2d, R.md:292-304);
 * kind 1: 3x3 stride-2 pad-1 output_padding-1 transposed conv (nn.ConvTranspose2d, R.md:370-378).
 * wpack: bf16 [cout][9][cin].  epilogue 0: +bias, raw bf16 out, statistics partials [2][cout][ntiles] (channel-major);
 * 1: ReLU mask of (yprev, prev_coef) + BN-backward partials; 2: plain store. */
int eae_op_conv_s2(void* stream, int kind, eae_src src, int cin, int cout, int B, int Hin, int Win, const void* wpack,
                   const float* bias, void* out, float* stat_part, int epilogue, const void* yprev, const float* prev_coef);
/* fp8 variants of the two ops above (eae_config::quant = 1 uses them; maps must be multiples of 8 x 16 positions).
 * conv: wpack_e4m3 = OCP e4m3 bytes [cout][9][cin] of w * s_w; qs (device) = {1/s_pixel, 1/(s_pixel*s_w)}; the pixel operand is
 * converted to e4m3 (activation sources) or e5m2 (source mode 2, a gradient) in registers; amax (device, may be NULL) receives the
 * largest |staged pixel operand| as float bits.  wgrad: qs (device) = {1/s_small, 1/s_big, 1/(s_small*s_big)}. */
int eae_op_conv_s2_fp8(void* stream, int kind, eae_src src, int cin, int cout, int B, int Hin, int Win, const void* wpack_e4m3,
                       const float* bias, void* out, float* stat_part, int epilogue, const void* yprev, const float* prev_coef,
                       const float* qs, unsigned* amax);
int eae_op_wgrad_s2_fp8(void* stream, eae_src small_src, eae_src big_src, int cs, int cb, int B, int Hs, int Ws, float* scratch,
                        long long scratch_floats, float* dw, const float* qs);
/* number of statistics partials per channel (= workgroups) eae_op_conv_s2 writes for this shape */
int eae_op_conv_s2_ntiles(int kind, int cin, int B, int Hin, int Win);
/* first / last layer kernels: src3_kind 0 = fp32 NCHW [B,3,H,W], 1 = bf16 NHWC4 [B,H,W,4]; out [B,H/2,W/2,32] */
int eae_op_edge_conv(void* stream, int src3_kind, const void* src3, int B, int H, int W, const void* wpack32x64 /* k = tap*4 + c */,
                     const float* bias, void* out, float* stat_part, int epilogue, const void* yprev, const float* prev_coef);
int eae_op_edge_wgrad(void* stream, int src3_kind, const void* src3, int B, int H, int W, eae_src side, float* scratch,
                      long long scratch_floats, float* dw /*[32][3][3][3]*/);
int eae_op_deconv4_loss(void* stream, eae_src a3, int B, int Hin, int Win, const void* wjoint, const float* bias,
                        const float* x, float gscale, float* x_hat, void* g4, float* loss_part /*[ntiles][4]*/);
/* weight gradient of a 3x3 s2 layer: dw [cs][cb][3][3] fp32 (reference layout) */
int eae_op_wgrad_s2(void* stream, eae_src small_src, eae_src big_src, int cs, int cb, int B, int Hs, int Ws, float* scratch,
                    long long scratch_floats, float* dw);
int eae_op_bn_finalize(void* stream, const float* stat_part, int ntiles, int C, long long count, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, long long* nbt, float momentum,
                       float eps, float* coef);
int eae_op_bn_eval_coef(void* stream, int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* coef);
int eae_op_bn_bwd_finalize(void* stream, const float* stat_part, int ntiles, int C, long long count, const float* gamma,
                           const float* coef_fwd, float* dgamma, float* dbeta, float* coef_bwd);
/* Latent projections (nn.Linear(256*h*w, L) R.md:309 and nn.Linear(L, 256*h*w) R.md:365) on NHWC-flattened activations
 * (feature index k' = pixel*256 + channel; the permutation to the reference's c*P+p order lives in the packed weights).
 *   eae_op_fc_splitk : out[M][N] fp32 = T(a)[M][K] . w[N][K]^T (+ bias[N]) (+ addend[M][N]); split-K in slices of 128 through
 *                      `scratch` (>= K/128 * M * N floats) and a fixed-order reduction.  a.mode 0 (bf16) or 1 (BN+ReLU on load,
 *                      coef [4][256], channel = k % 256).  N % 64 == 0, K % 128 == 0.      (enc.fc forward, dec.fc backward-data)
 *   eae_op_fc_bias_bf16 : out[M][N] bf16 = a_f32[M][K] . w[N][K]^T + bias[N]; N % 256 == 0, K % 64 == 0.   (dec.fc forward)
 *   eae_op_fc_wgrad : dw = P^T . Q over the batch, written in the REFERENCE weight layout; mode 0: P = bf16 [Bt][I], Q = fp32
 *                     [Bt][J], dw [I][J] with rows permuted back to c*Pn+p (dec.fc); mode 1: P = fp32 [Bt][I], Q = BN+ReLU source
 *                     [Bt][J], dw [I][J] with columns permuted back (enc.fc); colsum = sum_b P (bias gradient) or NULL. */
int eae_op_fc_splitk(void* stream, eae_src a, const void* w_bf16, int M, int N, int K, const float* bias, const float* addend,
                     float* scratch, long long scratch_floats, float* out);
int eae_op_fc_bias_bf16(void* stream, const float* a_f32, const void* w_bf16, int M, int N, int K, const float* bias, void* out_bf16);
int eae_op_fc_wgrad(void* stream, int mode, eae_src p, eae_src q, int Bt, int I, int J, int Pn, float* dw, float* colsum);
/* Classification head Linear(L,128)-ReLU-Linear(128,C) (R.md:423-427) fused with CrossEntropyLoss(mean) (R.md:623) and its
 * whole backward, fp32.  grads = the 4 head gradients in state-dict order (W1 [128][L], b1 [128], W2 [C][128], b2 [C], each
 * padded to a multiple of 4 floats); loss2[0] = mean CE, loss2[1] = number of correct argmax.  labels NULL: forward only.
 * scratch >= eae_op_head_scratch_floats(B, L, C). */
long long eae_op_head_scratch_floats(int B, int L, int C);
int eae_op_head_ce(void* stream, const float* z, const float* w1, const float* b1, const float* w2, const float* b2,
                   const long long* labels, int B, int L, int C, float* logits, float* dz, float* grads, float* loss2,
                   float* scratch, long long scratch_floats);
/* Sigmoid backward for an externally supplied dL/dx_hat (autograd path): g4 = bf16 NHWC4 of dx_hat*x_hat*(1-x_hat), plus the
 * deconv4 bias gradient db[3]; x_hat, dx_hat fp32 NCHW [B,3,H,W]; scratch >= ceil(B*H*W/256)*4 floats. */
int eae_op_sigmoid_bwd(void* stream, const float* x_hat, const float* dx_hat, int B, int H, int W, void* g4, float* db,
                       float* scratch);
/* pack a [A][B][3][3] fp32 weight into bf16 p1 [A][9][B] and p2 [B][9][A] */
int eae_op_pack3x3(void* stream, const float* w, int A, int B, void* p1, void* p2);
int eae_op_adam(void* stream, float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                double beta2, double eps, double weight_decay, long long step);

/* ------------------------------------------------------------------ input staging ------ */
/* The loader-side transforms of the reference fused on the device: train = RandomHorizontalFlip -> RandomCrop(H, padding=4)
 * -> ToTensor -> AddGaussianNoise(0, noise_std) (R.md:211-230), eval = ToTensor (R.md:232-234).  in_u8: uint8 HWC [B,H,W,3];
 * out: fp32 NCHW [B,3,H,W].  Randomness: Philox4x32-10 keyed by (seed, step) unless explicit per-image params int32 [B][3] =
 * (flip, top, left) with top,left in 0..8 and/or standard-normal noise [B,3,H,W] are supplied. */
int eae_augment(void* stream, const void* in_u8, float* out, int B, int H, int W, int train, float noise_std,
                unsigned long long seed, unsigned long long step, const int* params, const float* noise);

/* ------------------------------------------------------------------ external MLP (R.md:2549-2566) ---------- */
typedef struct eae_mlp eae_mlp;
#define EAE_MLP_NPARAMS 10
int eae_mlp_layout(int input_dim, int num_classes, long long* param_off /*[11]*/, long long* bn_off /*[5]*/);
int eae_mlp_create(int input_dim, int num_classes, int max_batch, eae_mlp** out);
int eae_mlp_destroy(eae_mlp* m);
int eae_mlp_bind(eae_mlp* m, float* params, float* grads, float* adam_m, float* adam_v, float* bn_running,
                 long long* bn_nbt);
int eae_mlp_set_adam_step(eae_mlp* m, long long step);
/* logits = clf(xb) (R.md:2643 train mode / 2663, 3182 eval mode); dropout mask = Philox(seed, step counter) or the
 * caller-supplied keep-mask [B][128] (fp32 0/1) when drop_mask != NULL. */
int eae_mlp_forward(eae_mlp* m, void* stream, const float* x, int B, int train, unsigned long long seed,
                    const float* drop_mask, float* logits);
/* loss.backward() for a torch-side loss on the logits of the preceding train-mode eae_mlp_forward of the same batch
 * (R.md:2644-2645); gradients land in the gradient arena. */
int eae_mlp_backward(eae_mlp* m, void* stream, const float* x, int B, unsigned long long seed, const float* drop_mask,
                     const float* dlogits);
/* one iteration of R.md:2641-2649: zero_grad, forward, CE, backward, Adam(lr, weight_decay);
 * stats: float[8] += loss*B, B, #correct. */
int eae_mlp_train_step(eae_mlp* m, void* stream, const float* x, const long long* labels, int B, float lr,
                       float weight_decay, unsigned long long seed, const float* drop_mask, float* logits, float* stats);
/* forward + CE + accuracy bookkeeping without update (validation / test loops, R.md:2660-2668, 2689-2695) */
int eae_mlp_eval_step(eae_mlp* m, void* stream, const float* x, const long long* labels, int B, float* logits,
                      float* stats);

#ifdef __cplusplus
}
#endif
#endif /* EAE_H */
