/* iwae_amd -- C ABI of the MI355X-native IWAE train / eval step.
 *
 * Drop-in boundary for the hot path of nbip/IWAE (a pure-Python TF2 repo with no FFI of its
 * own): each entry point replaces the Python-level call named next to it (file:line under
 * /root/reference).  Plain C types only; every function returns IWAE_OK (0) or a negative
 * iwae_status, the message is available from iwae_last_error().  No C++ exceptions cross
 * this boundary.  A handle is bound to one GPU and one HIP stream; calls on one handle are
 * stream-ordered and not thread-safe, different handles are independent.
 *
 * Pointers named `x`, `eps`, `flat`, `out` may be host OR device pointers (the copy uses
 * hipMemcpyDefault); host results are valid when the call returns.
 * Tensor order at this boundary is the reference's: sample axis first, [k, B, ...]
 * (src/iwae1.py:59,107-125), C-contiguous float32.
 */
#ifndef IWAE_AMD_H
#define IWAE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iwae_model* iwae_handle;

typedef enum {
    IWAE_OK = 0,
    IWAE_ERR_ARG = -1,      /* bad argument (the Python shim maps it to ValueError / KeyError) */
    IWAE_ERR_HIP = -2,      /* HIP runtime error */
    IWAE_ERR_NOMEM = -3,    /* device allocation failed */
    IWAE_ERR_STATE = -4     /* call order (e.g. backward without forward) */
} iwae_status;

/* --objective choices of main.py:23, plus the DReG estimator of tasks/task02.py:87-101 */
typedef enum {
    IWAE_OBJ_VAE_ELBO = 0,     /* src/iwae1.py:120 */
    IWAE_OBJ_IWAE_ELBO = 1,    /* src/iwae1.py:125 */
    IWAE_OBJ_IWAE_EQ14 = 2,    /* src/iwae1.py:128-134 */
    IWAE_OBJ_VAE_ELBO_KL = 3,  /* src/iwae1.py:121 (1-layer only; a KeyError for the 2-layer model, src/iwae2.py:154-167) */
    IWAE_OBJ_DREG = 4          /* tasks/task02.py:61-101 (1-layer only) */
} iwae_objective;

/* arithmetic of the GEMMs.  The reference computes everything in float32 (Keras Dense defaults, src/iwae1.py:31-34,72-75);
 * BASELINE.json configs[1] asks for bf16 operands. */
typedef enum {
    IWAE_PREC_BF16 = 0,        /* bf16 GEMM operands, fp32 accumulation (v_mfma_f32_16x16x32_bf16): the fast training path */
    IWAE_PREC_FP32 = 1         /* exact float32 GEMMs (v_mfma_f32_16x16x4_f32): every product as the reference's float32 graph */
} iwae_precision;

/* iwae1.IWAE(n_hidden, n_latent) src/iwae1.py:89-96 / iwae2.IWAE(n_hidden[2], n_latent[2]) src/iwae2.py:100-107.
 * struct_size MUST be set to sizeof(iwae_config) by the caller: iwae_create rejects any other value, so a binding
 * compiled against an older or newer layout fails loudly instead of reading past its struct. */
typedef struct {
    uint32_t struct_size;      /* = sizeof(iwae_config) (64); ABI guard */
    int32_t n_layers;          /* 1 or 2 stochastic layers (main.py:17) */
    int32_t n_hidden[2];       /* main.py:86,90 : {200} / {200,100}; each <= 256 */
    int32_t n_latent[2];       /* main.py:85,89 : {100} / {100,50};  each <= 128 */
    int32_t x_dim;             /* 784 */
    int32_t device;            /* HIP device ordinal */
    uint64_t seed;             /* Philox key for the reparameterisation noise (main.py:40-41 seeds TF) */
    int32_t world_size;        /* data-parallel ranks (1 = single GPU); iwae_comm_init must be given the same values */
    int32_t rank;              /* this process's rank (checked against iwae_comm_init).  The library does NOT derive noise keys from it:
                                  the caller passes the global index of the shard's first image, rank*B, through iwae_set_step */
    int32_t cond_dim;          /* 0, or C > 0: the conditional model of tasks/task05.py:101-168 (1-layer only): the encoder
                                  sees concat(x, y), the decoder concat(z, y), y [B, C] set with iwae_set_condition
                                  (one-hot labels there, C = 10); needs n_latent + C <= round_up(n_latent, 32) */
    int32_t cond_prior;        /* with cond_dim > 0: 1 = the learned conditional prior p(z|y) of tasks/task04.py:101-173 (a BasicBlock on y,
                                  created after the decoder) replaces N(0,1) in lpz; sample(z, y) maps z through it (:190-196) */
    int32_t precision;         /* iwae_precision of forward / train calls (iwae_eval_llh: iwae_set_eval_precision) */
    int32_t reserved;          /* 0 */
} iwae_config;

/* scalar entries of the result dict (src/iwae1.py:141-144, tasks/task02.py:78-79) and the
 * means that IWAE.write_to_tensorboard logs (src/iwae1.py:228-232) */
typedef struct {
    float vae_elbo;
    float vae_elbo_kl;
    float iwae_elbo;
    float iwae_eq14;
    float inference_loss;      /* DReG only */
    float mean_lpxz;           /* mean over [k,B] of lpxz (lpxz1) */
    float mean_lpz;            /* 1-layer: lpz ; 2-layer: lpz1z2 */
    float mean_lqzx;           /* 1-layer: lqzx ; 2-layer: lpz2 */
    float mean_kl;
    float reserved[7];
} iwae_scalars;

/* optional tensor outputs (NULL = not wanted, then never materialised).  1-layer names;
 * for the 2-layer model z=z1, z2=z2, lpz=lpz1z2, lpz2=lpz2, lqzx=lqz1x, lqzx2=lqz2z1 */
typedef struct {
    float* z;        /* [k,B,D1]  src/iwae1.py:145 */
    float* z2;       /* [k,B,D2]  src/iwae2.py:158 */
    float* snis_z;   /* [B,D1]    src/iwae1.py:146 */
    float* snis_z2;  /* [B,D2]    src/iwae2.py:160 */
    float* al;       /* [k,B]     src/iwae1.py:147 */
    float* logits;   /* [k,B,X]   src/iwae1.py:148 */
    float* lpxz;     /* [k,B]     src/iwae1.py:149 */
    float* lpz;      /* [k,B]     src/iwae1.py:150 */
    float* lqzx;     /* [k,B]     src/iwae1.py:151 */
    float* lpz2;     /* [k,B]     src/iwae2.py:165 */
    float* lqzx2;    /* [k,B]     src/iwae2.py:167 */
    float* log_w;    /* [k,B]     src/iwae1.py:113 */
} iwae_tensors;

const char* iwae_last_error(void);
int iwae_version(void);
/* 16 hex digits of the sha256 over the sources this binary was built from (iwae_amd/csrc/build.sh; "-diag" appended for diagnostic
 * builds): ties a shipped .so to a source tree -- the test suite rebuilds on mismatch, bench.py stamps its line with it and drops
 * profile artefacts (profiles/ *_kernel_traffic.json) taken on another build.  No reference counterpart. */
const char* iwae_build_id(void);

/* model construction: iwae1.IWAE(...) / iwae2.IWAE(...) ; weights glorot-uniform / zero-bias
 * (Keras Dense defaults, src/iwae1.py:31-34,72-75) drawn from `seed`; set the data-mean output
 * bias (src/utils.py:11-23) with iwae_set_params or iwae_set_output_bias. */
int iwae_create(const iwae_config* cfg, iwae_handle* out);
void iwae_destroy(iwae_handle h);
int iwae_set_stream(iwae_handle h, void* hip_stream);        /* run on the caller's stream (e.g. torch's) */
int iwae_sync(iwae_handle h);

/* model.trainable_weights (Keras creation order, kernel [in,out] then bias [out] per Dense) */
int iwae_param_count(iwae_handle h, size_t* n);
int iwae_num_tensors(iwae_handle h, int32_t* n);
int iwae_tensor_info(iwae_handle h, int32_t idx, char* name, size_t name_cap, int32_t* rows, int32_t* cols, size_t* offset);
int iwae_set_params(iwae_handle h, const float* flat, size_t n);   /* model.load_weights / set_weights */
int iwae_get_params(iwae_handle h, float* flat, size_t n);         /* model.save_weights (main.py:165) */
int iwae_set_output_bias(iwae_handle h, const float* bias, size_t n);  /* utils.get_bias(), src/utils.py:11-23 */
int iwae_get_grads(iwae_handle h, float* flat, size_t n);          /* tape.gradient result, src/iwae1.py:159 */
int iwae_get_adam_state(iwae_handle h, float* m, float* v, size_t n, int64_t* step);
int iwae_set_adam_state(iwae_handle h, const float* m, const float* v, size_t n, int64_t step);

/* model(x, n_samples, beta) / val_step : src/iwae1.py:98-151,164-166 (main.py:152,176).
 * x [B, x_dim] in {0,1}; eps NULL (device Philox) or the N(0,1) draws of qzx.sample:
 * 1-layer [k,B,D1]; 2-layer eps = [k,B,D1] followed by [k,B,D2]. */
int iwae_forward(iwae_handle h, const float* x, int32_t B, int32_t k, float beta, const float* eps,
                 iwae_scalars* scalars, const iwae_tensors* want);

/* model.train_step(x, n_samples, beta, optimizer, objective) : src/iwae1.py:153-162 (main.py:143),
 * tasks/task02.py:87-101 for IWAE_OBJ_DREG.  lr = optimizer.learning_rate (main.py:93,128-133). */
int iwae_train_step(iwae_handle h, const float* x, int32_t B, int32_t k, float beta, float lr, int32_t objective,
                    const float* eps, iwae_scalars* scalars, const iwae_tensors* want);

/* the two halves of train_step, for data-parallel training: forward+backward leaves the flat fp32
 * gradient of THIS rank's shard (mean over its B images) on the device; the caller all-reduces
 * iwae_grad_devptr() (RCCL) and applies Adam with grad_scale = 1/world_size. */
int iwae_forward_backward(iwae_handle h, const float* x, int32_t B, int32_t k, float beta, int32_t objective,
                          const float* eps, iwae_scalars* scalars, const iwae_tensors* want);
int iwae_grad_devptr(iwae_handle h, void** dev_ptr, size_t* n);
/* iwae_forward_backward for a data-parallel step that overlaps its exchange with the backward pass: the gradient of the
 * decoder's layers -- floats [*side_offset, n) of the flat buffer, final long before the encoder's -- is completed on the
 * library's side stream (*side_stream, a hipStream_t) and NOT joined into the main stream: the caller orders its
 * all-reduce of that segment behind *side_stream and of [0, *side_offset) behind the main stream, makes the main stream
 * wait for both and calls iwae_adam_step.  Models without such a segment return *side_offset = n (nothing left on the
 * side stream: float32 mode, and steps on <= 2 048 data rows, whose weight gradients all run on the main stream).  No reference counterpart (the reference is
 * single-device, main.py:32). */
int iwae_forward_backward_split(iwae_handle h, const float* x, int32_t B, int32_t k, float beta, int32_t objective,
                                const float* eps, void** side_stream, size_t* side_offset);
int iwae_adam_step(iwae_handle h, float lr, float grad_scale);      /* keras Adam(lr, epsilon=1e-4), main.py:93 */
/* keras.optimizers.Adam(learning_rate, beta_1, beta_2, epsilon) hyper-parameters of this handle's optimizer; the default
 * is what the reference trains with: Adam(lr, epsilon=1e-4) = (0.9, 0.999, 1e-4), main.py:93.  Keras form: epsilon is
 * added to sqrt(v) outside the bias correction. */
int iwae_set_adam(iwae_handle h, float beta_1, float beta_2, float epsilon);
/* conditional model (cond_dim > 0): y [n, cond_dim] (host or device) for the NEXT forward / train step / eval_llh / decode
 * of n images -- tasks/task05.py:108-118 (y_onehot), :185-190 (sample(z, y)).  Stays set until replaced. */
int iwae_set_condition(iwae_handle h, const float* y, int32_t n);
int iwae_set_step(iwae_handle h, uint32_t noise_step, uint32_t batch_offset); /* Philox counter words */

/* Data-parallel training INSIDE the library (BASELINE configs[4]; no reference counterpart, main.py:32 is single-device): one
 * process per GPU, every rank holds a handle created with the same seed / parameters and iwae_config.world_size / rank.
 * Rank 0 obtains an opaque id blob (iwae_comm_unique_id: RCCL ncclGetUniqueId, one per internal communicator), the caller
 * ships it to the other ranks by any means (MPI, a file, torch.distributed's store), and EVERY rank calls iwae_comm_init
 * with it (collective: ncclCommInitRank).  From then on iwae_train_step / iwae_train_step_dataset take the rank's shard
 * of the global batch (B = global batch / world_size images) and all-reduce the flat fp32 gradient with ncclAllReduce on the
 * library's own streams before Adam.  NOISE KEYS ARE THE CALLER'S JOB: the draws of an image are keyed by its global index,
 * batch_offset + row, and the library never adds rank * B itself -- every rank must call
 * iwae_set_step(step, global_batch_offset + rank * B) before each step (main.py and iwae_amd/parallel.py do), otherwise all ranks
 * draw the same noise and N ranks no longer reproduce what one rank would compute on the whole batch.  Adam runs with
 * grad_scale 1/world_size, identical on every rank (replicas stay bit-identical): the decoder's segment (done early, on the
 * side stream) is exchanged and applied there, beside the encoder's backward pass and the next encoder forward, exactly as
 * the single-GPU step defers it; the encoder's segment follows on the main stream.  RCCL is loaded at run time (dlopen):
 * the library itself does not link against it.  iwae_comm_destroy (or iwae_destroy) releases the communicators. */
int iwae_comm_unique_id(void* id_out, size_t cap, size_t* id_bytes);
int iwae_comm_init(iwae_handle h, const void* unique_id, size_t id_bytes, int32_t world_size, int32_t rank);
/* The checks of iwae_comm_init that need no other rank (arguments, handle state, RCCL loadable), without the rendezvous: ncclCommInitRank
 * blocks until every rank has entered it, so a multi-process caller runs this first, agrees on the outcome over its own channel, and only
 * then lets every rank call iwae_comm_init (no reference counterpart: the reference is single-device, main.py:24,32). */
int iwae_comm_preflight(iwae_handle h, const void* unique_id, size_t id_bytes, int32_t world_size, int32_t rank);
int iwae_comm_destroy(iwae_handle h);
/* what RCCL itself reports for the handle's communicators (ncclCommCount / ncclCommUserRank): *world_size = 0, *rank = -1 when
 * the handle has none.  bench.py records it so a multi-GPU line shows which exchange path ran and over how many ranks. */
int iwae_comm_info(iwae_handle h, int32_t* world_size, int32_t* rank);

/* test-set LLH loop of main.py:170-184: mean over N images of iwae_elbo(k samples, B=1), images
 * batched `chunk` at a time on the device (chunk <= 0: as many as the row cap per launch allows -- option eval_rows; by default 2^21 rows
 * where the whole decoder forward is ONE launch for the evaluator's precision (1-layer model, hidden width 200, unconditional, the fused
 * decoder kernels not switched off), 2^19 otherwise; beyond it an image's k samples are walked in chunks and merged with a running
 * log-sum-exp).
 * x: host or device pointer, [N, x_dim]; a host batch is uploaded once.  The launches' per-image estimates stay on the device until one
 * copy at the end: the call returns after one synchronisation.  An image's estimate does not depend on the launch it rode in (the draws are
 * keyed by the global image index: iwae_set_step).  llh_per_image may be NULL. */
int iwae_eval_llh(iwae_handle h, const float* x, int32_t N, int32_t k, int32_t chunk, double* llh, float* llh_per_image);
/* arithmetic of iwae_eval_llh, independent of iwae_config.precision: IWAE_PREC_FP32 by default (the reference evaluates in
 * float32, main.py:176; 10 000 images x k = 5000 take well under a second either way), IWAE_PREC_BF16 for the fast path. */
int iwae_set_eval_precision(iwae_handle h, int32_t precision);

/* IWAE.sample(z): decoder only -> probs [n, x_dim].  1-layer: src/iwae1.py:168-178, z [n,D1].  2-layer:
 * src/iwae2.py:184-196, z = z2 [n,D2]: z1 ~ p(z1|z2) is drawn on the device (Philox), then decoded. */
int iwae_decode(iwae_handle h, const float* z, int32_t n, float* probs);

/* Data pipeline on the device (main.py:59-65,117-120 + src/utils.py:26-27): the grey-level training set
 * stays resident in HBM as uint8 [n, x_dim]; every epoch gets a visiting order (tf.data shuffle) and a
 * fresh dynamic binarisation, x = 1 iff (philox(seed, epoch, image, pixel/4) >> 8) < floor(g*2^24/255 + 0.5),
 * i.e. Bernoulli(g/255), one draw per image per epoch like the reference's per-epoch bernoullisample.
 * iwae_train_step_dataset takes rows [start, start+B) of the current order; gather + binarise are fused
 * into the input kernel, no host traffic.  iwae_dataset_get_batch returns the same batch for inspection. */
int iwae_dataset_upload(iwae_handle h, const uint8_t* gray, int32_t n);
int iwae_dataset_begin_epoch(iwae_handle h, uint32_t epoch, const int32_t* order /* NULL keeps the order */, int32_t n);
int iwae_dataset_get_batch(iwae_handle h, int32_t start, int32_t B, float* x_out);
/* Conditional models (cond_dim > 0; tasks/task05.py:296-322 trains on (x, y) batches of a labelled set): one class id per image of the
 * uploaded set, each < cond_dim; kept resident next to the images.  iwae_train_step_dataset then feeds onehot(y) of the batch's images
 * wherever a host-fed step takes iwae_set_condition's rows (the encoder's input concat(x, onehot(y)), the decoder's concat(z, onehot(y)),
 * the prior network of tasks/task04.py) -- gathered by the same input kernel, no host traffic.  A new iwae_dataset_upload drops the labels.
 * iwae_dataset_get_labels returns the batch's one-hot rows [B, cond_dim] for inspection. */
int iwae_dataset_set_labels(iwae_handle h, const uint8_t* labels, int32_t n);
int iwae_dataset_get_labels(iwae_handle h, int32_t start, int32_t B, float* y_out);
int iwae_train_step_dataset(iwae_handle h, int32_t start, int32_t B, int32_t k, float beta, float lr, int32_t objective,
                            iwae_scalars* scalars);

/* HIP-event timing of the step's heavy kernels, each on the stream it is launched on (used by bench.py's roofline
 * object): enable, run steps, then read the average launch duration.  enable = n > 0 brackets the kernels of every n-th
 * step (an event record costs a few us of stream bubble, so bench.py samples rather than timing every launch); 0 switches
 * it off.  name: "decoder_fwd" (whole decoder forward + log-likelihood), "out_bwd" (output-layer backward), "decoder_bwd" (the decoder's whole dX chain where it is one launch), "wgrad_out",
 * "dx_hidden", "dx_latent", "wgrad_hidden", "wgrad_latent" (the decoder's other backward kernels), "latent_bwd",
 * "encoder_fwd", "reduce_adam" (main-stream slab reduction + Adam).  A kernel a configuration does not launch reports 0 launches. */
int iwae_enable_timing(iwae_handle h, int32_t enable);
int iwae_kernel_time(iwae_handle h, const char* name, double* avg_us, int64_t* launches);

/* Kernel-selection switches of a handle, for A/B measurements and for the parity tests that compare kernel variants of the same
 * mathematics (names and meanings: tools/README.md; the defaults are the measured best).  This call is the ONLY way to steer the
 * library: it never reads the environment.  Synchronises the handle's streams.  Unknown names fail with IWAE_ERR_ARG. */
int iwae_set_option(iwae_handle h, const char* name, int64_t value);

/* debugging: fetch an internal activation / gradient as float32 [rows, feat] (names in DESIGN.md) */
int iwae_debug_tensor(iwae_handle h, const char* name, float* out, size_t cap, int32_t* rows, int32_t* cols);
/* the N(0,1) draws the device generator produces for (B,k): [k,B,D] */
int iwae_debug_eps(iwae_handle h, int32_t B, int32_t k, int32_t layer, float* out);

#ifdef __cplusplus
}
#endif
#endif
