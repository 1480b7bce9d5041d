/*
 * inr_abi.h -- C-ABI of libinr_mi355x.so, the MI355X (gfx950) engine for the coordinate-MLP
 * fitting hot path of luisdavid64/MRI-Implicit-Neural-Representations.
 *
 * The reference has no FFI: the path sits behind torch.nn.Module + torch.optim.
 * Each entry point below names the reference code it replaces (paths under /root/reference/src).
 * Conventions:
 *   - every function returns 0 on success, <0 on error; inr_last_error() gives the message
 *     (thread-local); nothing throws or aborts across the ABI;
 *   - all tensor arguments are DEVICE pointers to contiguous row-major fp32 buffers owned by the
 *     caller; the library retains no pointer across calls and allocates no device memory -- with ONE exception:
 *     an INR_PRECISION_BF16 plan owns 32 bytes of device memory, its gradient-scale state (inr_plan_grad_scale_state),
 *     allocated with hipMalloc and initialised with a synchronous hipMemcpy by the first call that needs it on a device
 *     (a step, a backward, or inr_plan_grad_scale_state), freed by inr_plan_destroy.  That first call must therefore run
 *     OUTSIDE stream capture;
 *   - all work is ordered on the hipStream_t passed as `stream` (void*); no implicit sync.  One exception to
 *     "enqueued on": a fused step whose tiles leave the last round of the persistent grid partly empty runs part of
 *     its weight-gradient GEMM on a low-priority side stream the plan creates on first use, forked from and joined
 *     back into `stream` with events inside the call (csrc/inr_api.hip step_schedule; INR_OVERLAP=0 disables it) --
 *     to the caller the call still behaves as work on `stream`, including under stream capture;
 *   - a plan's description is immutable after creation (that side stream, and a bf16 plan's scale state, are its only state) and it may be shared
 *     between threads and streams as long as each in-flight call has its own workspace buffers.
 *
 * Parameter layout ("flat params", P floats): layer k's weight [M_k, K_k] (PyTorch [out,in]
 * layout) followed by its bias [M_k], k = 0..D-1 -- i.e. the reference's state_dict order
 * (SIREN: model.{k}.linear.weight/bias, networks.py:79,114-117; FFN: model.{2k}.weight/bias,
 * networks.py:57-62; WIRE: net.{k}.linear.weight/bias + net.{depth+1}.weight/bias with complex64
 * tensors stored as interleaved (re, im) float pairs = torch.view_as_real; the frozen
 * omega_0/scale_0 Parameters are NOT part of the flat buffer).  Gradients (torch's convention
 * dL/dRe + j dL/dIm for complex tensors) and Adam moments use the same layout.
 */
#ifndef INR_ABI_H
#define INR_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7: inr_adam_step_shard (data-parallel update on the entries a rank owns), inr_reg_grad (penalty gradients, complex64 tensors
 *    included; the Adam entry points refuse l1 / l2 != 0 on plans with complex tensors).
 * 6: inr_plan_step_info, inr_loss_tv_grad, 16-word gradient-scale state.  5: bf16 plans.  4: inr_adam_step_dev.  3: inr_workspace. */
#define INR_ABI_VERSION 7

/* error codes */
#define INR_OK 0
#define INR_ERR_INVALID (-1)     /* bad argument (null pointer, B <= 0, unsupported shape) */
#define INR_ERR_UNSUPPORTED (-2) /* valid request the engine has no kernel for */
#define INR_ERR_HIP (-3)         /* a HIP runtime call failed */

/* network family: which reference class the plan mirrors */
enum inr_kind {
  INR_KIND_SIREN = 0, /* models/networks.py:99-124  SIREN / SirenLayer :74-96 */
  INR_KIND_FFN = 1,   /* models/networks.py:48-69   FFN (ReLU hidden, Sigmoid output) */
  INR_KIND_WIRE = 2,  /* models/networks.py:206-260 WIRE / ComplexGaborLayer :160-204 (complex64 layers) */
  INR_KIND_FOURIER = 3,   /* models/mfn.py:61-94    FourierNet (one output_linear after the last stage) */
  INR_KIND_MSFOURIER = 4, /* models/mfn.py:206-267  MultiscaleKFourier: heads output_linear[i], i in [1,3,5,7];
                             the unused last stage / heads exist in flat params but are never evaluated or
                             stepped (the reference leaves their .grad = None) */
  INR_KIND_MSBOUNDED = 5, /* models/mfn.py:288-355  MultiscaleBoundedFourier: BoundedLinear (:269-286) zeroes the rows
                             of h whose dist lies outside [lo,hi] before each hidden Linear */
  INR_KIND_GABOR = 6,     /* models/mfn.py:133-162  GaborNet: GaborLayer filters (:96-131)
                             sin(F x + c) * exp(-0.5 * gamma_j * |x - mu_j|^2), mu and gamma trainable */
  INR_KIND_KGABOR = 7,    /* models/mfn.py:164-204  KGaborNet: same arithmetic (dist_to_center is passed to the
                             filters but with_dist_filtering is never enabled); the dist argument is ignored */
  INR_KIND_WIRE2D = 8     /* models/wire2d.py:62-117 WIRE2D / ComplexGaborLayer2D :3-60: two Linears per layer
                             (linear, scale_orth), width NOT reduced; flat params interleave them per layer.
                             inr_forward needs a save buffer (the orth terms travel through it). */
};

/* arithmetic of the three GEMM loops (accumulation, master weights and Adam are fp32 in both) */
enum inr_precision {
  INR_PRECISION_F32 = 0,  /* v_mfma_f32_32x32x2_f32: the parity path (1e-5 against the reference's fp32) */
  INR_PRECISION_BF16 = 1  /* v_mfma_f32_32x32x16_bf16 on bf16-rounded operands + hardware sin/cos: throughput path
                             for SIREN + fused gauss encoder, width 129..256; always needs the save buffer */
};

/* activation of the last layer */
enum inr_act {
  INR_ACT_ID = 0,      /* network_last_linear: True (default), networks.py:96 */
  INR_ACT_SIN = 1,     /* sin(w0 z): hidden SIREN layers; last layer if network_last_linear False */
  INR_ACT_TANH = 2,    /* last_tanh: True, networks.py:94-95 */
  INR_ACT_RELU = 3,    /* FFN hidden, networks.py:57-60 */
  INR_ACT_SIGMOID = 4, /* FFN output, networks.py:63 */
  INR_ACT_CTANH = 7    /* WIRE2D last_tanh: torch.nn.Tanh() on the complex output before .real (wire2d.py:106-107,
                          113-117); out_features <= 2 */
};

/* how the first layer's input is produced */
enum inr_input {
  INR_INPUT_X = 0,     /* x [B, in_features] already in memory (what model.forward receives,
                          train.py:163-169: coords = encoder.embedding(coords); model(coords)); every kind,
                          the filter networks of models/mfn.py included (mfn.py:34-43,85-94,255-267) */
  INR_INPUT_GAUSS = 1  /* fused Positional_Encoder 'gauss' (networks.py:30-33): coords [B,3] and
                          enc_B [E,3]; in_features must equal 2E; [B,2E] is never materialised */
};

/* pointwise losses the fused train step can evaluate in-kernel (others: inr_loss_grad + tier 1) */
enum inr_loss {
  INR_LOSS_L2_HALF = 0, /* 0.5 * torch.nn.MSELoss()   train.py:82,182 */
  INR_LOSS_L1_HALF = 1, /* 0.5 * torch.nn.L1Loss()    train.py:92,182 */
  INR_LOSS_TANH = 2,    /* TanhL2Loss                 metrics/losses.py:130-139 */
  INR_LOSS_LOGSPACE = 3,/* LogSpaceLoss               metrics/losses.py:214-223 */
  INR_LOSS_HDR = 4,     /* HDRLoss_FF (separable form) metrics/losses.py:236-264 */
  INR_LOSS_MSLE_HALF = 5, /* 0.5 * MSLELoss()          metrics/losses.py:18-27; train.py:84,182 */
  INR_LOSS_CENTER = 6    /* CenterLoss ('LSL' of train.py:87-88), pointwise part: metrics/losses.py:141-173,201;
                            its random-pair term is inr_center_pairs_grad */
};

typedef struct inr_net_desc {
  int32_t kind;         /* enum inr_kind */
  int32_t in_features;  /* net.network_input_size */
  int32_t width;        /* hidden features: net.network_width for SIREN / FFN / MFN / WIRE2D (1..512; WIRE2D counts
                           complex features, <= 256); for WIRE the number of COMPLEX hidden features,
                           int(network_width / sqrt(2)) (networks.py:228), <= 192.  Widths between the built
                           block counts run zero-padded. */
  int32_t depth;        /* net.network_depth as the reference counts it: all Linear layers for SIREN/FFN,
                           hidden complex layers only for WIRE (total Linear = depth + 2) */
  int32_t out_features; /* net.network_output_size, 1..4 */
  int32_t last_act;     /* enum inr_act */
  int32_t input;        /* enum inr_input */
  int32_t enc_size;     /* E (encoder.embedding_size) when input == INR_INPUT_GAUSS */
  float w0;             /* 30 for SIREN (networks.py:75); ignored otherwise */
  float first_omega_0;  /* WIRE net.first_omega_0 */
  float hidden_omega_0; /* WIRE net.hidden_omega_0 */
  float scale_0;        /* WIRE net.scale */
  int32_t precision;   /* enum inr_precision; 0 (the default of a zeroed struct) = exact fp32 */
  int32_t reserved[3];
} inr_net_desc;

typedef struct inr_loss_desc {
  int32_t kind;       /* enum inr_loss */
  float eps;          /* loss_opts.hdr_eps */
  float sigma;        /* loss_opts.hdr_ff_sigma */
  float factor;       /* loss_opts.hdr_ff_factor */
  float inv_count;    /* 1 / (number of rows entering the mean, summed over all ranks) */
  float hdr_A;        /* mean_i((1-f_i)^2) over the batch's kcoords (HDR only) */
  float scale;        /* multiplies the pointwise loss; 0 means 1.  The multiscale loop uses 0.5*loss_fn
                         (train_kspace_multiscale.py:188-190): 0.5 for LogSpace, 1 for the *_HALF kinds */
  /* ConsistencyLoss between consecutive heads (metrics/losses.py:315-324; multiscale kinds only) */
  float cons_w;       /* 0.1 (train_kspace_multiscale.py:179); 0 disables */
  int32_t cons_chan;  /* 2: dist is [B] (rows compared); 1: per-coil dist [B,1] -> channel 0 only */
  float cons_lo[4], cons_hi[4]; /* bounds[i] of pair i: rows with dist < lo or dist > hi are compared */
  float cons_inv[4];  /* 1 / (number of compared elements of pair i over all ranks); 0 when the pair is empty */
} inr_loss_desc;

typedef struct inr_plan inr_plan;

/* Caller-owned scratch of one call, with its extents so that a short buffer is INR_ERR_INVALID and never an
 * out-of-bounds GPU write.  `save` = the activation stash (slots of inr_sizes.save_bytes_per_tile), `slabs` = the
 * gradient slabs (of inr_sizes.slab_floats floats); how many of each a batch of B rows needs:
 * inr_plan_launch_dims / inr_plan_workspace.  Extents are in floats. */
typedef struct inr_workspace {
  float* save;
  int64_t save_floats;
  float* slabs;
  int64_t slab_floats;
} inr_workspace;

/* sizes (in bytes unless stated) a caller needs to allocate buffers for a plan */
typedef struct inr_sizes {
  int64_t n_params;        /* P: floats in flat params / grads / Adam moments */
  int64_t packed_floats;   /* floats in the MFMA-fragment-ordered weight image */
  int64_t tile_rows;       /* coordinates per workgroup tile (128) */
  int64_t save_bytes_per_tile; /* activation stash per tile (tier 1: n_tiles of them) */
  int64_t max_blocks;      /* upper bound on the persistent grid (slab count) */
  int64_t slab_floats;     /* floats per gradient slab (= P + 4 loss words, padded) */
  int64_t step_save_by_tile; /* inr_train_step_multi's `save`: 1 = n_tiles slots, 0 = n_blocks slots */
} inr_sizes;

int inr_abi_version(void);
/* copies the calling thread's last error message; returns its length */
int inr_last_error(char* buf, size_t cap);

/* Replaces: SIREN.__init__ / FFN.__init__ shape bookkeeping (networks.py:100-119, 49-65). */
int inr_plan_create(const inr_net_desc* desc, inr_plan** out);
int inr_plan_destroy(inr_plan* plan);
int inr_plan_sizes(const inr_plan* plan, inr_sizes* out);
/* number of tiles / persistent blocks / workspace bytes for a batch of B rows */
int inr_plan_launch_dims(const inr_plan* plan, int64_t B, int64_t* n_tiles, int64_t* n_blocks);
/* buffers of a training step on B rows: `save` slots (of save_bytes_per_tile) that inr_train_step /
 * inr_train_step_multi need -- n_tiles for plans whose weight gradients come from the batch-level GEMM
 * (inr_sizes.step_save_by_tile), n_blocks otherwise -- and the number of slabs (of slab_floats) behind `slabs` for
 * the step and backward entry points: n_blocks, plus one per K-chunk of that GEMM. */
int inr_plan_workspace(const inr_plan* plan, int64_t B, int64_t* step_save_slots, int64_t* n_slabs);
/* (v6) Which kernel runs the fused step (inr_train_step*) of a batch of B rows, for benchmarks and tests; no reference
 * counterpart.  row_split = 1: inr_mlp_rs_kernel<ncb, .> -- the four waves of a workgroup split the OUTPUT rows of every layer
 * and share the tile's coordinates, dealt in column blocks of 16 (SIREN / FFN behind the fused gauss encoder, hidden width
 * 129..256): `grid` workgroups run `rounds` tiles each, tile t = round * grid + workgroup holds `hi` blocks if t < n_hi, else
 * `lo`.  row_split = 0: the plan's per-coordinate-tile kernel (inr_mlp_kernel, inr_mfn_kernel, inr_siren_bf16_kernel, ...):
 * grid = n_blocks of inr_plan_launch_dims, rounds = ceil(n_tiles / grid), the other fields 0. */
typedef struct inr_step_info {
  int32_t row_split, ncb, grid, rounds, hi, lo, n_hi, reserved;
} inr_step_info;
int inr_plan_step_info(const inr_plan* plan, int64_t B, inr_step_info* out);
/* INR_PRECISION_BF16 plans (v5; 16 words since v6).  Their backward pass stashes dZ in 8 bits under a power-of-two scale that
 * follows the gradient's magnitude from step to step; the sixteen words of that state live on the device with the plan (layout:
 * csrc/inr_w2.h -- [0..3] fused steps, [4..7] split steps: next scale, bits of the last step's largest scaled |dZ|, the
 * factor the last step multiplied d(loss)/d(out) by, the scale inside it; [8], [9] fused steps so far whose gradients were
 * clipped at bf8's largest finite value / mostly flushed under its subnormals, [10], [11] the same for split steps -- the scale
 * lags the gradient by one step and has 2^10.8 of headroom; [12..15] zero).  This call waits for the work queued on
 * `stream` and copies them to host_out[16]: for tests and diagnostics (no reference counterpart).  Because of this state a
 * bf16 plan is to be stepped from one stream at a time; the plan's first step of a kind, and a step whose loss or batch
 * size makes the remembered scale meaningless, runs the kernel twice (once to find the scale). */
int inr_plan_grad_scale_state(const inr_plan* plan, float* host_out, void* stream);

/* Re-orders flat params into the MFMA A-fragment image the kernels stream (no reference
 * counterpart: it is what `model.to(device)` + ATen's GEMM packing do implicitly). */
int inr_pack_params(const inr_plan* plan, const float* params, float* packed, void* stream);

/* Replaces Positional_Encoder.embedding, 'gauss' (networks.py:30-33): out [B, 2E]. */
int inr_encode_gauss(const float* coords, const float* enc_B, int64_t B, int32_t E, float* out,
                     void* stream);

/* Replaces Positional_Encoder.embedding, 'LogF' (networks.py:16,24-29): bands [n_bands] =
 * 2^linspace(0, scale, n_bands); out [B, 6*n_bands] = per axis [sin(2 pi x_a b) | cos(2 pi x_a b)]. */
int inr_encode_logf(const float* coords, const float* bands, int64_t B, int32_t n_bands, float* out,
                    void* stream);

/* Replaces model.forward (networks.py:121-124 / 67-69).  `x` is [B,in_features] (INR_INPUT_X) or
 * coords [B,3] (INR_INPUT_GAUSS, with enc_B [E,3]).  out [B,out_features].  `ws` may be NULL (or ws->save NULL)
 * for a no_grad forward (train.py:203-220); otherwise ws->save receives n_tiles stash slots. */
int inr_forward(const inr_plan* plan, const float* params, const float* packed, const float* x,
                const float* enc_B, int64_t B, float* out, const inr_workspace* ws, void* stream);

/* Replaces loss.backward() through the model (train.py:189): given d(loss)/d(out) [B,out_features]
 * and the forward's stash (ws->save, n_tiles slots), writes d(loss)/d(params) into grads [P].  ws->slabs holds
 * n_slabs slabs (inr_plan_workspace).  The stash is CONSUMED: plans with inr_sizes.step_save_by_tile overwrite
 * act'(z_l) with dZ_l, the operand of their weight-gradient GEMM. */
int inr_backward(const inr_plan* plan, const float* params, const float* packed, const float* x,
                 const float* enc_B, int64_t B, const float* dout, const inr_workspace* ws,
                 float* grads, void* stream);

/* Replaces the loss modules + their autograd (train.py:178-182; metrics/losses.py): writes
 * dout [B,2] = d(loss)/d(out) and accumulates the scalar loss into loss_out[0] (device).
 * `mask` (may be NULL) is one byte per row: rows with 0 do not enter the loss (train.py:172-177).
 * `kcoords` [B,3] is only read for INR_LOSS_HDR. */
int inr_loss_grad(const inr_loss_desc* loss, const float* out, const float* gt, const float* kcoords,
                  const uint8_t* mask, int64_t B, float* loss_out, float* dout, void* stream);

/* Multi-head form for the multiscale loop (train_kspace_multiscale.py:176-195): outs / douts are [n_heads][B][2]
 * (what inr_forward_multi writes / inr_backward_multi reads); the pointwise terms (scale * loss_fn, summed over
 * heads) see the rows with mask != 0, ConsistencyLoss (cons_* fields, metrics/losses.py:315-324) every row. */
int inr_loss_grad_multi(const inr_loss_desc* loss, const float* outs, const float* gt, const float* dist,
                        const uint8_t* mask, int32_t n_heads, int64_t B, float* loss_out, float* douts, void* stream);

/* Replaces tv_loss (metrics/losses.py:326-343) + its autograd on one coil's predicted k-space
 * (train.py:173-175, train_kspace_multiscale.py:173-175; per-coil batches from
 * MRICoilWrapperDataset, data/nerp_datasets.py:397-441):
 *   weight * ( mean|img[:, :-1] - img[:, 1:]| + mean|img[:-1] - img[1:]| ),  img [H][W][2].
 * `out` [R][W][2] holds R consecutive image rows of which the first R_own belong to the caller;
 * a data-parallel rank passes its rows plus one halo row (R = R_own + 1) so that every vertical
 * pair is counted by exactly one rank, and the single-GPU call passes R = R_own = H.  The means
 * are over the whole image (H), so per-rank losses and gradients sum to the reference's.
 * ADDS the gradient into dout [R][W][2] and the loss into loss_out[0]; loss_out[1..64] is scratch
 * (same layout as inr_loss_grad, which is normally called first on the same buffers). */
int inr_tv_grad(const float* out, int64_t R, int64_t R_own, int64_t W, int64_t H, float weight,
                float* loss_out, float* dout, void* stream);

/* (v6) inr_loss_grad + inr_tv_grad of one coil's rows in one pass -- the per-coil step of train.py:163-189 with use_tv:
 * the masked pointwise loss (train.py:176-182) on the rows the caller owns (r < R_own; a data-parallel rank's halo row stays
 * with its owner: no zeroed copy of the mask is needed) plus tv_loss (metrics/losses.py:326-343, arguments as inr_tv_grad).
 * out / gt / dout [R][W][2], mask [R][W] bytes or NULL.  WRITES dout (does not add) and loss_out[0] = pointwise + TV loss;
 * loss_out[1..256] is scratch: a loss_out buffer holds INR_LOSS_WORDS floats. */
#define INR_LOSS_WORDS 512
int inr_loss_tv_grad(const inr_loss_desc* loss, const float* out, const float* gt, const uint8_t* mask, int64_t R,
                     int64_t R_own, int64_t W, int64_t H, float tv_weight, float* loss_out, float* dout, void* stream);

/* Replaces the random-pair ("centre") term of CenterLoss.forward (metrics/losses.py:175-199; 'LSL' of train.py:87-88,
 * called at train.py:178-180) for ONE radial band: the caller draws the n row pairs (idx_a[p] from the inner mask, idx_b[p]
 * from the ring outside it) with torch.randperm exactly as the reference does (losses.py:193-194) and passes global row
 * indices into out / gt [B,2];  r_p = (|gt_a| - |gt_b|) - (|out_a| - |out_b|).  ADDS  weight * mean_p r_p^2  to loss_out[0]
 * (loss_out[1..64] is scratch, same layout as inr_loss_grad) and its gradient to dout [B,2].  Call after inr_loss_grad with
 * INR_LOSS_CENTER on the same buffers (weight = 0.1 per band, losses.py:201), before inr_backward. */
int inr_center_pairs_grad(const float* out, const float* gt, const int64_t* idx_a, const int64_t* idx_b, int64_t n,
                          int64_t B, float weight, float* loss_out, float* dout, void* stream);

/* Fused tier-2 step, stages 1-3 of train.py:163-189 in one launch: encode -> forward -> pointwise
 * loss -> backward, then the fixed-order slab reduction.  Leaves grads [P] and loss_out[0];
 * the caller all-reduces grads across ranks (if any) and calls inr_adam_step.  `grads` may be
 * NULL to launch the fused kernel alone and leave the slabs unreduced (used to time it). */
int inr_train_step(const inr_plan* plan, const inr_loss_desc* loss, const float* params,
                   const float* packed, const float* x, const float* enc_B, const float* gt,
                   const uint8_t* mask, int64_t B, const inr_workspace* ws, float* grads,
                   float* loss_out, void* stream);

/* Multi-head networks (MultiscaleKFourier.forward returns a list, mfn.py:255-267): `out` / `dout` are
 * [n_heads][B][out_features]; `coords` is coords [B,3] with enc_B [E,3] (INR_INPUT_GAUSS plans) or the encoded
 * x [B,in_features] with enc_B NULL (INR_INPUT_X plans: what the reference's forward receives, mfn.py:34-43);
 * `dist` [B] = dist_to_center (nerp_datasets.py:385), read by the consistency term and the bounded linears;
 * ws->save is always required (it also carries the input features between stages;
 * pass n_blocks slots and by_block = 1 for a no_grad sweep, n_tiles slots and 0 before inr_backward_multi).
 * inr_backward_multi CONSUMES the stash: plans with inr_sizes.step_save_by_tile overwrite stashed factors with
 * the gradient operands of their batch-level weight-gradient GEMM, so one forward serves one backward.
 * inr_train_step_multi takes n_tiles slots when step_save_by_tile is set, n_blocks slots otherwise. */
int inr_forward_multi(const inr_plan* plan, const float* params, const float* packed, const float* coords,
                      const float* enc_B, const float* dist, int64_t B, float* out, const inr_workspace* ws,
                      int32_t by_block, void* stream);
int inr_backward_multi(const inr_plan* plan, const float* params, const float* packed, const float* coords,
                       const float* enc_B, const float* dist, int64_t B, const float* dout,
                       const inr_workspace* ws, float* grads, void* stream);
/* MultiscaleBoundedFourier(boundaries=pairs_model) (train_kspace_multiscale.py:85,95): one (lo,hi) per hidden
 * Linear; must be called before the plan is used (dist may be NULL for the other kinds above). */
int inr_plan_set_bounds(inr_plan* plan, const float* lo, const float* hi, int32_t n);
int inr_train_step_multi(const inr_plan* plan, const inr_loss_desc* loss, const float* params,
                         const float* packed, const float* coords, const float* enc_B, const float* gt,
                         const float* dist, const uint8_t* mask, int64_t B, const inr_workspace* ws,
                         float* grads, float* loss_out, void* stream);
int inr_plan_heads(const inr_plan* plan, int32_t* n_heads);

/* Replaces torch.optim.Adam.step (train.py:76,190; eps 1e-8, amsgrad False, L2-style
 * weight_decay) plus the L1/L2 "regularization" gradient terms (models/regularization.py:21-36),
 * and refreshes `packed` for the next forward.  `step` counts from 1.  Hyper-parameters are
 * doubles because torch derives step_size = lr / (1 - beta1^t) etc. in Python doubles. */
int inr_adam_step(const inr_plan* plan, float* params, const float* grads, float* exp_avg,
                  float* exp_avg_sq, float* packed, double lr, double beta1, double beta2, double eps,
                  double weight_decay, double l1, double l2, int32_t step, void* stream);

/* Replaces the backward of `train_loss += regularization(model.parameters())` (train.py:185-187;
 * models/regularization.py:21-36) for ANY plan, complex64 tensors included (WIRE / WIRE2D: interleaved (re, im) pairs in
 * the flat vector): adds to grads[i - lo], lo <= i < hi, the gradient of
 *   l1 * sum |p|         -- real entries sign(p); a complex entry z contributes |z|: (re, im) / |z|, 0 at z = 0
 *   l2 * |S|, S = sum p^2 over every Parameter -- complex when the model has complex tensors (z^2 = a^2 - b^2 + 2iab):
 *                           with u = conj(S) / |S|: real entry 2 p Re(u); d/d re = 2 Re(u z), d/d im = -2 Im(u z).
 * `l2_dir` = device pointer to (Re u, Im u); NULL means (1, 0) and is required to be non-NULL only for l2 != 0 on a plan
 * with complex tensors.  The caller forms S -- it also holds the squares of the frozen omega_0 / scale_0 Parameters
 * (networks.py:191-192), which the flat vector does not carry -- and the penalty VALUE it logs.  The Adam entry points
 * form the real-entry terms themselves (l1, l2 arguments) and refuse l1 / l2 != 0 on plans with complex tensors: there,
 * call this first and pass them 0.  [lo, hi) = [0, P) for a whole gradient, or a rank's chunk before inr_adam_step_shard. */
int inr_reg_grad(const inr_plan* plan, const float* params, float* grads, int64_t lo, int64_t hi, double l1,
                 double l2, const float* l2_dir, void* stream);

/* The update of a data-parallel job whose ranks each own 1/N of the flat parameter vector (no reference counterpart:
 * train_kspace_multiscale.py:161-201 is single-process; this is torch.optim.Adam on a slice): the same arithmetic as
 * inr_adam_step on the entries [lo, hi) only.  `grads_shard[i - lo]` is the summed gradient of entry i -- the rank's
 * chunk of a reduce-scatter of the flat gradient; params / exp_avg / exp_avg_sq are the full-length vectors (only
 * [lo, hi) is read and written).  No image is refreshed: after the all-gather of the updated parameters every rank calls
 * inr_pack_params.  lo == hi is a no-op (a rank whose chunk lies in the padding). */
int inr_adam_step_shard(const inr_plan* plan, float* params, const float* grads_shard, float* exp_avg,
                        float* exp_avg_sq, int64_t lo, int64_t hi, double lr, double beta1, double beta2,
                        double eps, double weight_decay, double l1, double l2, int32_t step, void* stream);

/* inr_train_step and inr_adam_step as ONE call for single-rank steps (nothing sits between the reduction and the
 * update): the Adam update and the re-pack ride in the slab reduction's launch (flat-layout plans: SIREN / FFN, fp32 and
 * bf16; two launches otherwise) -- one launch and one pass over the gradient less per step.  Bit-identical to the two
 * calls; `grads` still receives the gradient. */
int inr_train_adam_step(const inr_plan* plan, const inr_loss_desc* loss, float* params, float* packed,
                        const float* x, const float* enc_B, const float* gt, const uint8_t* mask, int64_t B,
                        const inr_workspace* ws, float* grads, float* loss_out, float* exp_avg,
                        float* exp_avg_sq, double lr, double beta1, double beta2, double eps,
                        double weight_decay, double l1, double l2, int32_t step, void* stream);

/* The same update with the step count in DEVICE memory, so that the launch carries no argument that changes from
 * one step to the next and a whole step (inr_train_step + this) can be captured once in a HIP graph and replayed:
 * `step_dev` holds the number of steps taken so far (t); the kernel reads (step_size, bc2_sqrt) from
 * `sched[2*min(t, n_sched-1)]` — device copy of a table made by inr_adam_schedule — and a one-thread kernel behind it
 * stores t + 1.  Results are bit-identical to inr_adam_step(step = t + 1) while t < n_sched; with beta1 = 0.9,
 * beta2 = 0.999 both fp32 terms have converged after 16 600 steps, so a table of 32 768 entries is exact for any run
 * length.  A new learning rate (the per-epoch LambdaLR of train.py:153,251) means a new table, not a new graph. */
int inr_adam_step_dev(const inr_plan* plan, float* params, const float* grads, float* exp_avg,
                      float* exp_avg_sq, float* packed, const float* sched, int32_t n_sched,
                      int32_t* step_dev, double beta1, double beta2, double eps, double weight_decay,
                      double l1, double l2, void* stream);
/* Host helper: host_out[2*t] = float(lr / (1 - beta1^(t+1))), host_out[2*t+1] = float(sqrt(1 - beta2^(t+1))),
 * t = 0..n-1, in the doubles torch.optim.Adam uses (the same code path as inr_adam_step).  No GPU work. */
int inr_adam_schedule(double lr, double beta1, double beta2, int32_t n, float* host_out);

#ifdef __cplusplus
}
#endif
#endif /* INR_ABI_H */
