// Host side of libiwae_amd.so: device memory, launch sequencing and the C ABI of include/iwae_amd.h.
// Step structure follows the reference's train_step (src/iwae1.py:153-162): forward (IWAE.call,
// :98-151), backward (closed form of tape.gradient), Adam (main.py:93).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stddef.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>
#include <random>
#include <memory>
#include <dlfcn.h>
#include <rccl/rccl.h>       // types only: RCCL is loaded with dlopen at iwae_comm_init, the library does not link against it
#include "../../include/iwae_amd.h"
#include "kernels.h"
#include "layout.h"

using namespace iwae;

static_assert(sizeof(iwae_config) == 64 && offsetof(iwae_config, struct_size) == 0 && offsetof(iwae_config, seed) == 32 && offsetof(iwae_config, cond_dim) == 48 &&
              offsetof(iwae_config, cond_prior) == 52 && offsetof(iwae_config, precision) == 56, "iwae_config layout is part of the ABI (iwae_amd/_capi.py)");
static_assert(sizeof(iwae_scalars) == 64 && sizeof(iwae_tensors) == 12 * sizeof(void*), "ABI struct layout");

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? IWAE_ERR_NOMEM : IWAE_ERR_HIP,                     \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                            \
    } while (0)
#define CHK(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_ != IWAE_OK) return rc_; \
    } while (0)

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct KerasLayer {
    std::string name;
    int Kin, Nout;
    size_t offW, offb;
};

// one GEMM-able linear map; the mu|sigma head merges two Keras layers into one (joff = 0 / Dp)
struct Linear {
    int Kin = 0, Nspace = 0;          // in-features, out-feature space (heads: 2*Dp)
    int Kp32 = 0, Np32 = 0, KT = 0, MG = 0;
    int nsub = 0, sub[2] = {0, 0}, joff[2] = {0, 0};
    char* imgF = nullptr; size_t imgF_bytes = 0;
    char* imgB = nullptr; size_t imgB_bytes = 0; int KT_B = 0, MG_B = 0, MT_B = 0, kmajor = 0;
    DevBuf slabW, slabB;
    int IT = 0, JT = 0, nsplit = 1;
};

struct BlockWs {   // activations / gradients of one BasicBlock applied to R rows
    DevBuf h1P, h2P, head, dheadP, d2P, d1P, dx;
};
struct MlpWs {     // decode_z_to_x applied to M rows
    DevBuf g1P, g2P, dlP, d2P, d1P, dz;
    DevBuf g2wP;      // g2 times the row weight (bf16, P-layout; pad feature H = the row weight): the pre-weighted operand of the output layer's weight gradient
};

}  // namespace

// RCCL entry points, resolved at run time (the process may already hold torch's copy of librccl: that one is reused)
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int load_rccl() {
    if (g_rccl.lib) return IWAE_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;       // already in the process (e.g. torch's)
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!lib) return fail(IWAE_ERR_STATE, std::string("RCCL not found (dlopen librccl.so): ") + (dlerror() ? dlerror() : ""));
    RcclApi r;
    r.lib = lib;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(lib, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(lib, "ncclCommUserRank");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.CommCount || !r.CommUserRank || !r.GetErrorString) return fail(IWAE_ERR_STATE, "librccl.so lacks an expected symbol");
    g_rccl = r;
    return IWAE_OK;
}
#define NCCLCHK(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fail(IWAE_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));    \
    } while (0)

// kernels iwae_enable_timing brackets with HIP events (on the stream each is launched on); names: iwae_kernel_time
enum TimedKernel { T_OUT_BWD = 0, T_DEC_FWD, T_WGRAD_OUT, T_DX_HID, T_DX_LAT, T_WGRAD_HID, T_WGRAD_LAT, T_LATENT_BWD, T_ENC_FWD, T_REDUCE, T_DEC_BWD, T_AR_ENC, T_AR_DEC, T_COUNT };
static const char* const kTimedNames[T_COUNT] = {"out_bwd", "decoder_fwd", "wgrad_out", "dx_hidden", "dx_latent", "wgrad_hidden", "wgrad_latent",
                                                 "latent_bwd", "encoder_fwd", "reduce_adam", "decoder_bwd",
                                                 "allreduce_enc", "allreduce_dec"};      // (the data-parallel step's two ncclAllReduce calls, each on its own stream)

struct iwae_model {
    iwae_config cfg;
    int X, Xp32;
    int C = 0, Xinp = 0;       // conditional model: condition width; row width of the encoder input concat(x, y) (= Xp32 without)
    DevBuf cond; int cond_n = 0;   // y [cond_n][C] fp32 for the next call (iwae_set_condition)
    int cond_row0 = 0;         // first row of `cond` the current forward uses (eval_llh walks chunks)
    int H[2], D[2], Hp[2], Dp[2];
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<KerasLayer> klayers;
    size_t nparam = 0;
    // Linear maps.  1-layer: enc{l1,l2,head}, dec{d1,d2,out}.  2-layer adds enc2, dec2 blocks.
    Linear enc1[3], enc2[3], dec2[3], dec1[3];
    Linear prior[3];           // conditional prior network p(z|y) (cfg.cond_prior, tasks/task04.py:108): BasicBlock on y
    bool has_prior = false;
    DevBuf condP;              // y as bf16 P-layout [Bp][32*ceil(C/32)] (the prior block's input)
    float *param = nullptr, *grad = nullptr, *mom = nullptr, *vel = nullptr;
    int64_t adam_t = 0;
    // float32 mode (iwae_config.precision / iwae_set_eval_precision): row-major float32 activations, GEMMs on v_mfma_f32_16x16x4_f32
    struct F32Block { DevBuf h1, h2, dhead, d2, d1, dx; };
    struct F32State { F32Block enc1, enc2, dec2, prior; DevBuf z[2], g1, g2, logits, d2, d1, slab, bpart, xcat, kslab; } f32;
    // float32 weight gradients of a step keep their row-split slabs (each in its own region of f32.slab) and are summed by ONE launch at the end of
    // backward_f32 (reduce_slabs_multi_f32_kernel): jobs queued by f32_dw, slab offsets in floats (the buffer may still grow while they queue)
    struct F32Pending { size_t off; size_t stride; size_t n; float* out; int nsplit; int seg; };
    std::vector<F32Pending> f32_pending; size_t f32_slab_used = 0; bool allow_f32_multi_reduce = true;
    bool allow_f32_side = true, f32_side_active = false, f32_wout_first = true;      // float32 step: the decoder's weight gradients + update on the side stream (options no_f32_side, f32_wout_first)
    bool f32_z_pending = false;
    int f32_dw_last = 0;
    bool bf16_side_used = false;      // a bf16 call may have left a speculative draw on a side stream (forward_f32 waits for it on the host)
    size_t f32_slab_want = 0, f32_slab_want_step = 0;      // floats of slabs the last whole step asked for (the buffer's target size) / this step so far
    int eval_tag_kill = -1;
    int eval_k_total = 0, eval_s_off = 0;     // > 0 while iwae_eval_llh walks an image's samples in chunks (eps_src)
    bool in_eval_llh = false;                 // iwae_eval_llh's launches need log_w only: no second (DReG) density per sample (round 5: ~14 % of the sampling pass)
    DevBuf eval_x, eval_lme;                  // iwae_eval_llh: the images (uploaded once) and the per-image log-mean-exps of every launch
    int eval_rows = 0;                        // data rows per evaluator launch (option eval_rows): images x samples, k chunked beyond it; 0 = eval_rows_auto()
    int eval_precision = IWAE_PREC_FP32;      // arithmetic of iwae_eval_llh (iwae_set_eval_precision)
    bool fwd_was_f32 = false;                 // the last forward ran in float32 mode (its backward must too)
    const float* f32_x = nullptr;             // device x [B][X] of the last float32 forward
    // data-parallel training inside the library (iwae_comm_init): one communicator per stream that carries a collective
    ncclComm_t comm_main = nullptr, comm_side = nullptr;
    int comm_world = 1, comm_rank = 0;
    float adam_b1 = 0.9f, adam_b2 = 0.999f, adam_eps = 1e-4f;   // keras Adam(lr, epsilon=1e-4) of main.py:93 unless iwae_set_adam says otherwise
    uint32_t noise_step = 0, batch_offset = 0;
    // layer descriptor table
    std::vector<LayerDesc> descs;
    LayerDesc* d_descs = nullptr;
    int elem_blocks = 0, reduce_blocks = 0;
    bool descs_dirty = true;
    // per-call state
    int B = 0, k = 0, M = 0, Mp = 0, Bp = 0;
    float beta = 1.0f;
    bool have_forward = false, user_eps = false;
    unsigned dense_g1_mask = IWAE_DENSE_G1_DEFAULT;   // IWAE_DENSE_G1=<mask> (tuning aid, kernels.h)
    bool allow_s_mode = true;   // IWAE_OUT_RECOMPUTE=1 switches back to recomputing the logits in out_bwd (A/B measurements)
    DevBuf dg2_part;            // small row counts: out_bwd_s_kernel's per-pixel-group partial sums
    int px_parts = 1;           // > 1: log p(x|z) of this forward arrives in px_part as that many partial sums per row
    DevBuf px_part;
    bool s_mode = false;        // this step's forward kept s = x - sigmoid(l) in wdec1.dlP
    DevBuf xin, xP, epsbuf, zP[2];
    DevBuf rows[6];            // lpxz, t1, t2, t3, t4, lq_dreg   (per data row)
    DevBuf logw, wn, gx, cf, per_b, dzdir;
    // lse_kernel's outputs once more, written by the copy of it that runs on the side stream (see forward_impl): the output layer's
    // weight gradient takes its row weights from there
    DevBuf logw2, wn2, gx2, cf2, per_b2;
    int f32_dw_min_rows = 32;   // float32 weight gradients: a row split covers at least this many rows (option f32_dw_min_rows; 64 until round 5)
    int f32_dw_tiles = 1024;    // float32 weight gradients: workgroups aimed at per launch (row splits = this / output tiles; option f32_dw_tiles)
    bool f32_dec_fused_train = false;
    bool allow_f32_dec_fused = true;                           // float32 mode: the decoder forward as one launch (dec_fwd_f32_kernel; option no_f32_dec_fused)
    bool allow_f32_bern_fused = true, f32_keeps_s = false;      // float32 mode: log p(x|z) (and, in a training step, s) in the output layer's GEMM epilogue (option no_f32_bern_fused)
    bool allow_wg3 = true;                           // few rows: the decoder's three weight gradients as one grouped launch (option no_wg3)
    bool allow_dec_rows = true;                      // ... and, with <= 2 048 DATA rows, the decoder's in the same launch (dec_rows_step; option no_dec_rows)
    bool allow_wgrad_rows = true;                    // few rows (<= 2 048): the image encoder's weight gradients + Adam in ONE launch, whole row reduction per workgroup (wgrad_rows_kernel; option no_wgrad_rows)
    bool lse_fused = false, allow_lse_fused = true;  // the decoder kernel does lse_kernel's work for its rows (option no_lse_fused)
    // Option g2w (round 4, measured and NOT the default): the decoder kernel leaves g2w = bf16(g_r g2) and the output layer's weight gradient runs
    // unweighted on it (no 870 cycles of row weighting per loader stage).  That kernel got faster (107 -> 97 us in the step) and the step SLOWER
    // (0.2044 -> 0.2154 ms, interleaved A/B): the decoder kernel pays 4 us for 23 MB more writes and the backward phase is bound by its bytes, not
    // by that kernel's instruction stream.
    bool allow_g2w = false, g2w = false, g2w_descs = false;
    bool allow_lat_rows4 = false;                    // option lat_rows4 (round 5, measured and NOT the default): beyond 16 samples per image the sums inside block_bwd_kernel<4> (4 images per
                                                     // workgroup, an image's samples over four waves, 256 workgroups).  In the step it takes 32.6 us where latent_bwd_kernel + block_bwd_kernel
                                                     // take 18.7 + 10.1: its 1024-thread / 101-register workgroups need a whole CU each and only ~96 CUs are free beside the weight
                                                     // gradients (three rounds), where latent_bwd_kernel's small workgroups fit anywhere: c1 0.1965 vs 0.1962 ms, c2 0.3856 vs 0.3802
    bool allow_lat_in_block = true;                  // few images: latent_bwd_kernel's sums inside the encoder's block_bwd_kernel (option no_lat_in_block)
    bool lse_pending = false, allow_lse_in_bwd = true;      // few rows: this step's lse_kernel work was left to dec_bwd_rows_kernel (lse_saved; option no_lse_in_bwd)
    LseArgs lse_saved;
    bool lse_dup = false, allow_lse_dup = true;      // IWAE_NO_LSE_DUP=1: one lse_kernel, the side stream forks behind it (A/B measurements)
    BlockWs wenc1, wenc2, wdec2, wprior;
    MlpWs wdec1;
    DevBuf scratch;            // exports
    // resident dataset (iwae_dataset_*): uint8 grey levels [N][X] + the epoch's visiting order
    DevBuf ds_data, ds_order;
    DevBuf ds_labels; bool ds_has_labels = false;   // class id per image of the resident set (iwae_dataset_set_labels; conditional models)
    int ds_N = 0;
    int wg_target16_1 = 64;    // same, for layers that are a single block wide (IWAE_WG16_1): the hidden layers' gradients -- with the specialised-wave kernel 64 row splits (12.8 MB of slabs each) beat 128 (0.259 -> 0.249-0.254 ms/step); 48 and 32 are slower again
    int eps_blocks = 512;      // blocks of the ahead-of-time noise draw (IWAE_EPS_BLOCKS; 0 = one block per 256 threads of work)
    int wg_target8 = 128;      // same for the 8-wave launches on many rows (narrow layers of the 2-layer model; option wg8): 128 row splits halve the 109 MB of fp32 slabs 256 wrote per step (c2: 0.4193 -> 0.4176 ms; 64: 0.462)
    int wg_target8_few = 32;   // 8-wave launches on < 8 192 rows (the encoder's layers on the batch's images; IWAE_WG8_FEW): the 784-wide first layer in 4 row
                               // splits instead of 16 (10.6 -> 2.7 MB of slabs each way): 0.2439 -> 0.2351 ms/step at B = 1 024; 8 / 16 / 48: 0.2374 / 0.2374 / 0.2360
    int wg_target16 = 0;       // workgroups aimed at per 16-wave weight-gradient launch (option wg16; 0 = the model's default: 96 for the 1-layer model, 64 (round 5; 128 before) for
                               // the 2-layer one -- round 3, with the output layer's gradient starting right behind the decoder kernel: 80 / 88 / 96 / 104 / 112 / 128
                               // -> 0.2192 / 0.2168 / 0.2132 / 0.2164 / 0.2206 / 0.2175 ms, 24 row splits write 17 MB of slabs instead of 22.5; the 2-layer
                               // step: 0.3932 vs 0.3916): these are one-per-CU
                               // workgroups (128 KB of LDS); 256 of them lock every CU against the kernels running beside them on the main
                               // stream (256 -> 0.294, 192 -> 0.280, 160 -> 0.279 ms/step while the gradient forked behind out_bwd; forked
                               // behind lse_kernel, beside out_bwd: 96 -> 0.268, 112 -> 0.262, 128 -> 0.258, 144 -> 0.261, 160 -> 0.265)
    // N(0,1) draws of a step, fp32 [Mp][Dp] per latent layer, made by eps_gen_kernel and read by the sampling / decoder and
    // backward kernels.  A training step draws the NEXT step's noise during its forward pass on the side stream, idle then
    // (speculating step+1, same batch shape); it is ordered by the join the main stream performs anyway, and a forward
    // whose counters do not match the speculation draws on its own stream first.  Three ring slots: this step's draws, the
    // previous step's (its backward pass may still read them) and the next step's.
    DevBuf epsc[3][2];          // [ring slot][layer]: the step's draws, the previous step's (its backward may still read them
                                // when the next step's are requested) and the next step's (drawn during this step's forward)
    struct EpsTag { bool valid = false; uint32_t step = 0; uint64_t row_offset = 0; int M = 0; } eps_tag[3];
    int epsc_par = 0;
    // Few data rows (the single-stream regime of dec_rows_step, round 5): the draws of EPSM_STEPS consecutive steps in ONE launch, two buffers taking turns
    // (the next group is drawn during the forward pass of the current group's last step: the buffer it overwrites was last read a whole group ago, in stream order)
    DevBuf epsm[2][2];          // [buffer][layer]: [EPSM_STEPS][Mp][eps_ld]
    struct EpsMTag { bool valid = false; uint32_t step0 = 0; uint64_t row_offset = 0; int M = 0; } epsm_tag[2];
    bool allow_eps_multi = true;   // option no_eps_multi: one draw launch per step there too
    const float* epsc_ptr[2] = {nullptr, nullptr};
    char* d_zero = nullptr;    // 1 KiB of zeros (wgradp_kernel's source for rows >= M)
    uint32_t ds_epoch = 0;
    int ds_start = -1;         // >= 0: the next forward gathers + binarises rows ds_start.. from the dataset instead of reading x
    DevBuf stamps;             // diagnostic (IWAE_STAMPS=1)
    DevBuf dstamps; int dstamp_epi = -1, dstamp_kt = -1, dstamp_waves = 0;   // diagnostic (IWAE_DENSE_STAMPS)
    // optional HIP-event timing of the dominant kernels (iwae_enable_timing): pairs recorded on m->stream
    // fork/join of the decoder weight-gradient GEMMs (independent of the dz -> encoder chain) onto a side stream
    hipStream_t side = nullptr;
    hipStream_t tail = nullptr;        // this step's side stream that finishes last (carries the decoder's reduction / exchange / update)
    bool allow_wg_group = false;       // IWAE_WG_GROUP=1: the hidden layers' gradients as ONE grouped launch (measured: 0.2450 vs 0.2384 ms/step as two launches --
                                       // both at once take more of the machine from the output layer's gradient, which is what the step waits for)
    hipStream_t side2 = nullptr;       // the hidden layers' weight gradients beside the output layer's (IWAE_NO_SIDE2=1: behind it on `side`)
    hipEvent_t ev_s2 = nullptr;
    hipEvent_t ev_ar = nullptr;        // data-parallel step: recorded behind the encoder segment's all-reduce (dp_finish)
    bool early_held = false;           // in-library data-parallel step: backward_impl left the decoder's slab reduction to dp_finish
    bool dp_concurrent = false;        // option dp_concurrent: the two all-reduces of a step may run at the same time (see dp_finish)
    bool use_side2 = true;
    int dec_bwd_nw = 8;         // option dec_bwd_nw: dec_bwd_kernel's shape (8 waves x 16 rows, round 4 | 4 waves x 32 rows)
    hipEvent_t ev_lse = nullptr;
    bool early_wout = false, allow_early_wout = true;    // IWAE_NO_EARLY_WOUT=1: the output layer's weight gradient forks behind out_bwd with the others (A/B measurements)
    hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_blk = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_dec = nullptr;
    // Single-GPU train step: the decoder's slab reduction + Adam (90 % of the slab bytes) stays on the side stream and is
    // NOT joined at the end of the step -- nothing needs the decoder's new weights before the next step's d1 layer, so it
    // runs beside the next encoder forward.  dec_pending: ev_dec (recorded behind it) has not been waited for yet;
    // join_side() does that, and every entry point that touches parameters, gradients or the decoder calls it.
    bool dec_pending = false;
    size_t split_offset = 0;    // iwae_forward_backward_split: first float of the flat gradient that was left on the side stream
    int wg_shape9 = 0;          // IWAE_WG9 (bit mask, see wgradp_plan): layers that take the 8 + 8-wave / 128-feature shape of wgradws_kernel
    int fake_s = 0;             // DIAG builds: byte ablations of s (option fake_s)
    int abl_skip = 0;           // DIAG builds: launch ablations of the full-size step (option abl_skip; timing only, results wrong): 1 no output-layer weight gradient,
                                // 2 no hidden-layer weight gradients, 4 no deferred decoder reduction + update, 8 no latent_bwd_kernel, 16 no noise draw ahead
    int wg_debug = 0;           // IWAE_WG_DEBUG: diagnostic ablations of wgradp_kernel (kernels.h)
    bool allow_wg7 = true;      // IWAE_NO_WG7=1: the 16-wave weight-gradient shapes also where the 8-wave 7 x 4 shape exists (A/B measurements)
    int dec_rows_max = 1024;    // dec_bwd_rows_kernel up to this many rows (IWAE_DEC_ROWS), dec_bwd_kernel beyond
    bool small_dec_bwd = true; int small_rows = 8191;   // the one-launch dX chain also below 8 192 rows (IWAE_NO_SMALL_DEC_BWD=1: the per-pixel-group out_bwd + finish + two dX launches
                                                        // there).  Measured: B=20,k=1 0.1417 -> 0.1383 ms/step, B=100,k=5 150.7 -> 144.6 us, B=160,k=50 189.1 -> 165.7 us
    bool allow_dz_half = true;  // IWAE_DZ_F32=1: dec_bwd_kernel leaves dz as float32 (A/B measurements)
    bool allow_chain2_bwd = true, chain2_bwd = false;      // option no_chain2_bwd: the per-sample blocks' backward as gauss_bwd_kernel + dense_kernel launches; chain2_bwd: this step takes gblock_bwd_kernel
    bool allow_chain2 = true;   // option no_chain2: the 2-layer model's per-sample blocks as dense_kernel launches + sample_kernel + gauss_lp_kernel (A/B measurements, variant tests)
    bool allow_dec_bwd = true;  // IWAE_NO_DEC_BWD=1: out_bwd_s + the two dX kernels stay three launches (A/B measurements)
    bool allow_zin = true;      // IWAE_NO_ZIN=1: always the separate sampling kernel (A/B measurements)
    bool allow_zin_eval = false; // option zin_eval (round 4, measured and NOT the default): forward-only calls on many rows take their draws from eps_gen_kernel and let the decoder
                                 // kernel make z in its prologue instead of sample_kernel (inline Philox) in front of it -- bf16 evaluator 146 k vs 158 k images/s: the prologue's 20 MB of
                                 // float32 draws cost the vector-issue-bound kernel more than the separate pass
    bool allow_out_in_block = true;  // IWAE_NO_OUT_IN_BLOCK=1: the output layer of a few-row decoder stays a dense_kernel<EPI_BERN> launch (A/B measurements)
    bool allow_block_fused = true;   // IWAE_NO_BLOCK_FUSED=1: a BasicBlock on few rows stays three dense_kernel launches (A/B measurements)
    int num_cus = 256;               // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    bool bern_qw_force = false;      // IWAE_BERN_QW_FORCE=1: that shape at every row count it exists for (tests)
    bool bern_qw = true;             // IWAE_NO_BERN_QW=1: the decoder kernel's 8-wave / 128-row shape instead of 16 waves / 200 rows (A/B measurements)
    bool allow_dec_fused = true;     // IWAE_NO_DEC_FUSED=1: the two tanh layers of the decoder stay dense_kernel launches (A/B measurements)
    bool allow_bern_pipe = true;   // IWAE_NO_BERN_PIPE=1: the Bernoulli forward stays on dense_kernel<EPI_BERN> (A/B measurements)
    bool allow_defer = true;    // IWAE_NO_DEFER=1: always join at the end of the step (A/B measurements)
    int wout_split = 0, wout_wg1 = 56, wout_wg2 = 128;      // option wout_split (percent of the rows, 0 = off; round 5): the output layer's weight gradient as an EARLY launch on few
                                // workgroups beside dec_bwd_kernel (rows [0, R1)) and a LATE one behind it (the rest, beside the hidden layers' gradients)
    bool defer_split = false;   // option defer_split (round 5): 1-layer step, each side stream sums + updates the decoder layers whose gradients IT carried
    int early_first2 = -1;      // 2-layer model: first reduce block behind the image encoder's layers (everything whose weight gradients run on the side streams)
    bool allow_defer2 = true;   // option no_defer2
    bool allow_defer2_split = true, dec2_pending = false;      // ... one deferred update per side stream (option no_defer2_split: one, on `tail`)
    hipEvent_t ev_dec2 = nullptr;
    int early_first = -1;       // first reduce block of the decoder's layers when they are the tail of the table, else -1
    int timing = 0;            // 0 off, n > 0: time every n-th forward (event records cost a few us of stream bubble each)
    int64_t timing_calls = 0;
    bool time_this = false;
    std::vector<hipEvent_t> ev_start[T_COUNT], ev_stop[T_COUNT];   // per timed kernel (enum TimedKernel)
    size_t ev_used[T_COUNT] = {};
    bool want_stamps = false;
    float* d_scalars = nullptr;
    float* h_scalars = nullptr;   // pinned
};

namespace {

int ensure(DevBuf& b, size_t bytes, hipStream_t st) {
    if (bytes <= b.cap) return IWAE_OK;
    if (b.p) {
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    return IWAE_OK;
}
template <class T>
T* ptr(const DevBuf& b) { return (T*)b.p; }

void free_buf(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

void init_linear(Linear& L, int Kin, int Nspace, bool need_B, bool kmajor) {
    L.Kin = Kin;
    L.Nspace = Nspace;
    L.Kp32 = round_up(Kin, 32);
    L.Np32 = round_up(Nspace, 32);
    L.KT = L.Kp32 / 32;
    L.MG = (L.Np32 + 63) / 64;
    L.imgF_bytes = (size_t)L.MG * img_mg_group_bytes(L.KT);
    L.kmajor = kmajor ? 1 : 0;
    if (need_B) {
        if (kmajor) {
            L.MT_B = L.Kp32 / 16;                      // rows = in-features (hidden)
            L.imgB_bytes = (size_t)L.MG * 2 * L.MT_B * 1024;   // one k-group per 64 out-features
        } else {
            L.KT_B = L.Np32 / 32;
            L.MG_B = (L.Kp32 + 63) / 64;
            L.imgB_bytes = (size_t)L.MG_B * img_mg_group_bytes(L.KT_B);
        }
    }
    L.IT = L.Kp32 / 16;
    L.JT = L.Np32 / 16;
}

int alloc_linear(Linear& L) {
    HIPCHK(hipMalloc((void**)&L.imgF, L.imgF_bytes));
    HIPCHK(hipMemset(L.imgF, 0, L.imgF_bytes));
    if (L.imgB_bytes) {
        HIPCHK(hipMalloc((void**)&L.imgB, L.imgB_bytes));
        HIPCHK(hipMemset(L.imgB, 0, L.imgB_bytes));
    }
    return IWAE_OK;
}

void free_linear(Linear& L) {
    if (L.imgF) (void)hipFree(L.imgF);
    if (L.imgB) (void)hipFree(L.imgB);
    free_buf(L.slabW);
    free_buf(L.slabB);
}

// Keras creation order: a BasicBlock is l1, l2, lmu, lstd
void add_block(iwae_model* m, Linear* blk, const char* prefix, int Kin, int H, int D, bool need_dx_first) {
    const int base = (int)m->klayers.size();
    const char* nm[4] = {"l1", "l2", "lmu", "lstd"};
    const int kin[4] = {Kin, H, H, H}, nout[4] = {H, H, D, D};
    for (int i = 0; i < 4; ++i) {
        KerasLayer kl;
        kl.name = std::string(prefix) + "." + nm[i];
        kl.Kin = kin[i];
        kl.Nout = nout[i];
        kl.offW = m->nparam;
        m->nparam += (size_t)kin[i] * nout[i];
        kl.offb = m->nparam;
        m->nparam += nout[i];
        m->klayers.push_back(kl);
    }
    const int Dp = round_up(D, 32);
    init_linear(blk[0], Kin, H, need_dx_first, false);
    blk[0].nsub = 1; blk[0].sub[0] = base;
    init_linear(blk[1], H, H, true, false);
    blk[1].nsub = 1; blk[1].sub[0] = base + 1;
    init_linear(blk[2], H, 2 * Dp, true, false);
    blk[2].nsub = 2; blk[2].sub[0] = base + 2; blk[2].sub[1] = base + 3; blk[2].joff[1] = Dp;
}

void add_mlp3(iwae_model* m, Linear* mlp, const char* prefix, int D, int H, int X) {
    const int base = (int)m->klayers.size();
    const char* nm[3] = {"d1", "d2", "out"};
    const int kin[3] = {D, H, H}, nout[3] = {H, H, X};
    for (int i = 0; i < 3; ++i) {
        KerasLayer kl;
        kl.name = std::string(prefix) + "." + nm[i];
        kl.Kin = kin[i];
        kl.Nout = nout[i];
        kl.offW = m->nparam;
        m->nparam += (size_t)kin[i] * nout[i];
        kl.offb = m->nparam;
        m->nparam += nout[i];
        m->klayers.push_back(kl);
    }
    init_linear(mlp[0], D, H, true, false);
    mlp[0].nsub = 1; mlp[0].sub[0] = base;
    init_linear(mlp[1], H, H, true, false);
    mlp[1].nsub = 1; mlp[1].sub[0] = base + 1;
    init_linear(mlp[2], H, X, true, true);
    mlp[2].nsub = 1; mlp[2].sub[0] = base + 2;
}

std::vector<Linear*> all_linears(iwae_model* m) {
    std::vector<Linear*> v;
    for (int i = 0; i < 3; ++i) v.push_back(&m->enc1[i]);
    if (m->cfg.n_layers == 2) {
        for (int i = 0; i < 3; ++i) v.push_back(&m->enc2[i]);
        for (int i = 0; i < 3; ++i) v.push_back(&m->dec2[i]);
    }
    for (int i = 0; i < 3; ++i) v.push_back(&m->dec1[i]);
    if (m->has_prior) for (int i = 0; i < 3; ++i) v.push_back(&m->prior[i]);
    return v;
}

int build_descs(iwae_model* m) {
    m->descs.assign(m->klayers.size(), LayerDesc());
    for (Linear* L : all_linears(m)) {
        for (int s = 0; s < L->nsub; ++s) {
            const KerasLayer& kl = m->klayers[L->sub[s]];
            LayerDesc& d = m->descs[L->sub[s]];
            d.Kin = kl.Kin; d.Nout = kl.Nout; d.joff = L->joff[s];
            d.offW = kl.offW; d.offb = kl.offb;
            d.imgF = L->imgF; d.KT_F = L->KT;
            d.imgB = L->imgB; d.KT_B = L->KT_B; d.MT_B = L->MT_B; d.imgB_kmajor = L->kmajor;
            d.slabW = ptr<float>(L->slabW); d.slabB = ptr<float>(L->slabB);
            d.nsplit = L->nsplit; d.slab_ld = L->JT * 16; d.slab_stride = (size_t)L->IT * 16 * L->JT * 16;
            d.slabB_stride = 0;
            if (m->g2w && L == &m->dec1[2]) {      // pre-weighted output layer: the bias gradient is product row H (the pad feature that carries g_r) of every slab
                d.slabB = ptr<float>(L->slabW) + (size_t)kl.Kin * d.slab_ld;
                d.slabB_stride = d.slab_stride;
            }
        }
    }
    int blocks = 0, rblocks = 0;
    for (auto& d : m->descs) {
        d.block_begin = blocks;
        d.rblock_begin = rblocks;
        blocks += (d.Kin * d.Nout + d.Nout + 255) / 256;
        rblocks += ((d.Kin + 1) * ((d.Nout + 3) / 4) + 63) / 64;     // float4 groups: (Kin weight rows + bias row) x ceil(Nout/4)
    }
    m->elem_blocks = blocks;
    m->reduce_blocks = rblocks;
    m->early_first = -1;
    m->early_first2 = (m->cfg.n_layers == 2 && m->enc2[0].nsub >= 1) ? m->descs[m->enc2[0].sub[0]].rblock_begin : -1;      // (table order: enc1, enc2, dec2, dec1)
    const int d0 = m->dec1[0].sub[0];
    if (m->dec1[0].nsub == 1 && m->dec1[1].nsub == 1 && m->dec1[2].nsub == 1 && m->dec1[1].sub[0] == d0 + 1 &&
        m->dec1[2].sub[0] == d0 + 2 && d0 + 3 == (int)m->descs.size())
        m->early_first = m->descs[d0].rblock_begin;
    if (!m->d_descs) HIPCHK(hipMalloc((void**)&m->d_descs, sizeof(LayerDesc) * m->descs.size()));
    HIPCHK(hipMemcpyAsync(m->d_descs, m->descs.data(), sizeof(LayerDesc) * m->descs.size(), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->side) HIPCHK(hipStreamSynchronize(m->side));      // a deferred decoder update may still be reading the old table
    if (m->side2) HIPCHK(hipStreamSynchronize(m->side2));
    m->descs_dirty = false;
    m->g2w_descs = m->g2w;
    return IWAE_OK;
}

int refresh_images(iwae_model* m) {   // rebuild bf16 A-images from the fp32 master weights
    if (m->descs_dirty) CHK(build_descs(m));
    launch_adam(m->d_descs, (int)m->descs.size(), m->elem_blocks, m->param, m->grad, m->mom, m->vel, 0.f, 1.f, m->adam_b1, m->adam_b2, m->adam_eps, 0, m->stream);
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

struct ScopedTimer {     // records a start/stop event pair around a launch when timing is enabled
    iwae_model* m; int id; bool on;
    hipStream_t ts;
    ScopedTimer(iwae_model* m_, int id_, hipStream_t s_ = nullptr) : m(m_), id(id_), on(m_->time_this), ts(s_ ? s_ : m_->stream) {
        if (!on) return;
        if (m->ev_used[id] == m->ev_start[id].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            m->ev_start[id].push_back(a); m->ev_stop[id].push_back(b);
        }
        (void)hipEventRecord(m->ev_start[id][m->ev_used[id]], ts);
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(m->ev_stop[id][m->ev_used[id]], ts);
        m->ev_used[id] += 1;
    }
};

// row stride of the cached draws: the latent width rounded to 4, NOT the 32-padded operand width -- D = 100: 400 B instead of 512 B per row,
// 5.7 MB less per pass over the k = 50, B = 1 024 step's draws (written once, read by the decoder kernel and by latent_bwd_kernel)
static inline int eps_ld(const iwae_model* m, int layer) { return 4 * ((m->D[layer] + 3) / 4); }
EpsSrc eps_src(iwae_model* m, int layer) {
    EpsSrc e;
    e.user = nullptr;
    e.cache = m->epsc_ptr[layer];
    e.ldC = eps_ld(m, layer);
    if (m->user_eps) e.user = ptr<float>(m->epsbuf) + (layer == 0 ? 0 : (size_t)m->k * m->B * m->D[0]);
    e.B = m->B;
    e.seed = m->cfg.seed;
    e.row_offset = (uint64_t)m->batch_offset * (uint64_t)m->k;
    e.step = m->noise_step;
    e.stream = (uint32_t)layer;
    if (m->eval_k_total > 0) {      // k-chunked evaluation: the unchunked call's Philox rows
        e.k_total = m->eval_k_total; e.s_off = m->eval_s_off; e.kc = m->k;
        e.row_offset = (uint64_t)m->batch_offset * (uint64_t)m->eval_k_total;
    }
    return e;
}

// diagnostic (STAMPS=1 build + IWAE_DENSE_STAMPS="<epi>:<KT>"): record the phase stamps of the matching dense launch
int attach_dense_stamps(iwae_model* m, int epi, DenseArgs& a) {
    if (m->dstamp_epi != epi || m->dstamp_kt != a.KT || (a.M < 4096 && a.KT <= 8)) return IWAE_OK;
    m->dstamp_waves = std::max(((a.M + 127) / 128) * ((a.MG + a.mg_per_block - 1) / a.mg_per_block) * 4, epi == EPI_BERN ? ((a.M + 127) / 128) * 16 : 0);      // (bern_pipe_kernel: <= 16 waves per 128+ rows)
    CHK(ensure(m->dstamps, (size_t)m->dstamp_waves * 64, m->stream));
    a.stamps = ptr<unsigned long long>(m->dstamps);
    return IWAE_OK;
}

// orders the main stream behind a deferred decoder update (and the noise prefetch in front of it) still on the side stream
int join_side(iwae_model* m) {
    if (m->dec2_pending) {      // (2-layer step: the first side stream's own deferred update)
        HIPCHK(hipStreamWaitEvent(m->stream, m->ev_dec2, 0));
        m->dec2_pending = false;
    }
    if (!m->dec_pending) return IWAE_OK;
    HIPCHK(hipStreamWaitEvent(m->stream, m->ev_dec, 0));
    m->dec_pending = false;
    return IWAE_OK;
}

// ---------------------------------------------------------------- forward pieces
int dense_fwd(iwae_model* m, Linear& L, int epi, const uint16_t* XP, int rows, uint16_t* YP, float* YF, int ldYF, const SampleArgs* zin = nullptr) {
    DenseArgs a;
    memset(&a, 0, sizeof(a));
    a.X = XP; a.ldX = L.Kp32; a.img = L.imgF;
    a.split = (L.nsub == 2) ? L.joff[1] : (1 << 30);
    a.M = rows; a.KT = L.KT; a.MG = L.MG; a.mg_per_block = (rows <= 8192) ? 1 : L.MG; a.Np32 = L.Np32; a.g1_mask = m->dense_g1_mask;
    a.stage_all = (a.mg_per_block == 1 && L.KT > 8 && (L.KT + 7) / 8 <= 4) ? 1 : 0;
    if (zin) {      // sampled-input mode: the layer makes its own input rows z = mu + sigma*eps (and keeps them in zin->ZP)
        a.zhead = zin->head; a.ldZH = zin->ldH; a.zeps = zin->eps.cache; a.zldE = zin->eps.ldC; a.zD = zin->D; a.zDp = zin->Dp;
        a.ZPout = zin->ZP; a.zlp = zin->lp_prior; a.zlq = zin->lq; a.k = zin->k;
        a.mg_per_block = L.MG;          // one block owns all out-feature groups of its rows (the z rows are made once)
    }
    a.YP = YP; a.ldYP = L.Np32; a.YF = YF; a.ldYF = ldYF;
    CHK(attach_dense_stamps(m, epi, a));
    launch_dense(epi, a, m->stream);
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

int block_alloc(iwae_model* m, Linear* blk, BlockWs& w, int R, int Rp, bool bwd, bool need_dx) {
    const int Hp = blk[0].Np32, N2 = blk[2].Np32;
    CHK(ensure(w.h1P, (size_t)Rp * Hp * 2, m->stream));
    CHK(ensure(w.h2P, (size_t)Rp * Hp * 2, m->stream));
    CHK(ensure(w.head, (size_t)Rp * N2 * 4, m->stream));
    if (bwd) {
        CHK(ensure(w.dheadP, (size_t)Rp * N2 * 2, m->stream));
        CHK(ensure(w.d2P, (size_t)Rp * Hp * 2, m->stream));
        CHK(ensure(w.d1P, (size_t)Rp * Hp * 2, m->stream));
        if (need_dx) CHK(ensure(w.dx, (size_t)Rp * blk[0].Kp32 * 4, m->stream));
    }
    (void)R;
    return IWAE_OK;
}

// xf != null: the input rows are still fp32 [R][xdim]; the fused kernel converts them into XP itself (returns *took = true),
// otherwise the caller runs prep_rows first
int block_fwd(iwae_model* m, Linear* blk, BlockWs& w, const uint16_t* XP, int R, const float* xf = nullptr, int xdim = 0, bool* took = nullptr) {
    if (took) *took = false;
    if (m->allow_block_fused) {       // few rows (the encoder on the batch's images): the whole block in one launch
        BlockFwdArgs a;
        memset(&a, 0, sizeof(a));
        a.X = XP; a.ldX = blk[0].Kp32; a.img0 = blk[0].imgF; a.img1 = blk[1].imgF; a.img2 = blk[2].imgF;
        a.KT0 = blk[0].KT; a.KT1 = blk[1].KT; a.NT1 = blk[0].Np32 / 16; a.NT2 = blk[2].Np32 / 16; a.R = R;
        a.H1 = ptr<uint16_t>(w.h1P); a.H2 = ptr<uint16_t>(w.h2P); a.ldH = blk[0].Np32;
        a.YF = ptr<float>(w.head); a.ldYF = blk[2].Np32; a.split = (blk[2].nsub == 2) ? blk[2].joff[1] : (1 << 30);
        if (blk[1].Np32 == blk[0].Np32 && blk[2].KT == blk[1].KT && blk[1].Kp32 == blk[0].Np32 && block_fwd_ok(a)) {
            if (xf && took) { a.Xf = xf; a.Xdim = xdim; a.XPout = const_cast<uint16_t*>(XP); *took = true; }
            launch_block_fwd(a, m->stream);
            HIPCHK(hipGetLastError());
            return IWAE_OK;
        }
    }
    if (xf) return IWAE_OK;      // not taken (*took = false): the caller converts the rows and calls again
    CHK(dense_fwd(m, blk[0], EPI_TANH, XP, R, ptr<uint16_t>(w.h1P), nullptr, 0));
    CHK(dense_fwd(m, blk[1], EPI_TANH, ptr<uint16_t>(w.h1P), R, ptr<uint16_t>(w.h2P), nullptr, 0));
    CHK(dense_fwd(m, blk[2], EPI_HEAD, ptr<uint16_t>(w.h2P), R, nullptr, ptr<float>(w.head), blk[2].Np32));
    return IWAE_OK;
}

// ---------------------------------------------------------------- backward pieces
// weight gradient from the P-layout operands (wgradp_kernel); XP/GP row-major bf16, `rows` valid rows.
// wgradp_plan sizes the row splits and the slabs and fills the argument block; nw = the kernel shape (launch_wgradp:
// 8 = small, 16 = 16 waves, 7 = 8 waves with 7 x 4 accumulator tiles each, for inputs <= 224 features wide).
int wgradp_plan(iwae_model* m, Linear& L, const uint16_t* XP, const uint16_t* GP, int rows, WgradPArgs& a, int& nsplit, int& nw) {
    const int chunks = (rows + 63) / 64;
    nw = (L.JT > 8 && chunks >= 128) ? 16 : 8;
    if (nw == 16 && L.IT <= 14 && m->allow_wg7) nw = (m->wg_shape9 & (L.JT > 16 ? 1 : 2)) ? 9 : 7;      // IWAE_WG9: bit 0 the output layer, bit 1 the hidden layers
    const int blocks = ((L.JT + wgradp_strip(nw) - 1) / wgradp_strip(nw)) * ((L.IT + 15) / 16);
    // Workgroup targets (measured at k=50, B=1024).  Early builds, the weight gradients alone on the machine: 64 -> 0.501,
    // 128 -> 0.425, 256 -> 0.406, 384 -> 0.443 ms/step (fewer leaves CUs idle, more pays a full fp32 slab per extra split).
    // Since they run beside the dX chain and with the register-blocked kernel: 160 (see wg_target16); the
    // single-block-wide hidden layers prefer 128.
    // (round 5: the 2-layer step's MAIN stream is its long chain and its side streams have slack: 64 workgroups for the output layer's gradient leave the
    // main chain's kernels more CUs -- 128 / 96 / 72 / 64 / 56 / 48 -> 0.3915 / 0.3834 / 0.3896 / 0.3769 / 0.3805 / 0.3852 ms, interleaved)
    const int target16 = m->wg_target16 > 0 ? m->wg_target16 : (m->cfg.n_layers == 2 ? 64 : 96);
    const int target = (nw != 8) ? (blocks == 1 ? m->wg_target16_1 : target16) : (chunks < 128 ? m->wg_target8_few : m->wg_target8);
    nsplit = std::max(1, std::min(chunks, target / std::max(1, blocks)));
    const int cps = (chunks + nsplit - 1) / nsplit;
    nsplit = (chunks + cps - 1) / cps;
    const size_t needW = (size_t)nsplit * L.IT * 16 * L.JT * 16 * 4, needB = (size_t)nsplit * L.JT * 16 * 4;
    void* oldW = L.slabW.p; void* oldB = L.slabB.p;
    CHK(ensure(L.slabW, needW, m->stream));
    CHK(ensure(L.slabB, needB, m->stream));
    if (oldW != L.slabW.p || oldB != L.slabB.p || nsplit != L.nsplit) { L.nsplit = nsplit; m->descs_dirty = true; }
    a.X = XP; a.ldX = L.Kp32; a.IT = L.IT; a.G = GP; a.ldG = L.Np32; a.JT = L.JT; a.M = rows; a.rows_per_split = cps * 64;
    a.slabW = ptr<float>(L.slabW); a.slabB = ptr<float>(L.slabB); a.zero = m->d_zero; a.rowscale = nullptr;
    a.dbg = m->wg_debug;
    a.stamps = nullptr;
    return IWAE_OK;
}

int wgradp(iwae_model* m, Linear& L, const uint16_t* XP, const uint16_t* GP, int rows, hipStream_t st = nullptr,
           const float* rowscale = nullptr) {
    WgradPArgs a;
    int nsplit = 1, nw = 8;
    CHK(wgradp_plan(m, L, XP, GP, rows, a, nsplit, nw));
    a.rowscale = rowscale;
    if (rowscale && (m->fake_s & 2)) a.dbg |= 32;
#ifdef IWAE_DENSE_STAMPS
    if (m->dstamp_epi == 10 && rowscale && nw == 7) {      // diagnostic (STAMPS=1 build, option dense_stamps_epi = 10): phase stamps of the output layer's weight gradient
        m->dstamp_waves = ((L.JT + 15) / 16) * nsplit * 12;
        CHK(ensure(m->dstamps, (size_t)m->dstamp_waves * 64, m->stream));
        a.stamps = ptr<unsigned long long>(m->dstamps);
    }
#endif
    launch_wgradp(a, nsplit, nw, st ? st : m->stream);
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

// dX (times tanh' of the stored activation, or raw fp32) of a layer: X = dpre of the layer's outputs
// The output layer's gradient in two launches over disjoint row ranges (option wout_split): slabs [0, n1) come from rows [0, R1), slabs
// [n1, n1 + n2) from the rest -- the reduction sums them in that fixed order whichever launch ends first.
int wgradp_two_plan(iwae_model* m, Linear& L, const uint16_t* XP, const uint16_t* GP, int rows, const float* rowscale, WgradPArgs& a1, int& n1, WgradPArgs& a2, int& n2) {
    const int blocks = (L.JT + 15) / 16;
    const int chunks = (rows + 63) / 64;
    const int c1 = std::min(chunks - 1, std::max(1, (int)((long)chunks * m->wout_split / 100)));
    const int c2 = chunks - c1;
    auto split = [&](int ch, int target, int& n, int& cps) { n = std::max(1, std::min(ch, target / std::max(1, blocks))); cps = (ch + n - 1) / n; n = (ch + cps - 1) / cps; };
    int cps1, cps2;
    split(c1, m->wout_wg1, n1, cps1);
    split(c2, m->wout_wg2, n2, cps2);
    const int ns = n1 + n2;
    const size_t stride = (size_t)L.IT * 16 * L.JT * 16;
    void* oldW = L.slabW.p; void* oldB = L.slabB.p;
    CHK(ensure(L.slabW, (size_t)ns * stride * 4, m->stream));
    CHK(ensure(L.slabB, (size_t)ns * L.JT * 16 * 4, m->stream));
    if (oldW != L.slabW.p || oldB != L.slabB.p || ns != L.nsplit) { L.nsplit = ns; m->descs_dirty = true; }
    const int R1 = c1 * 64;
    memset(&a1, 0, sizeof(a1));
    a1.X = XP; a1.ldX = L.Kp32; a1.IT = L.IT; a1.G = GP; a1.ldG = L.Np32; a1.JT = L.JT; a1.M = R1; a1.rows_per_split = cps1 * 64;
    a1.slabW = ptr<float>(L.slabW); a1.slabB = ptr<float>(L.slabB); a1.zero = m->d_zero; a1.rowscale = rowscale;
    a2 = a1;
    a2.X = XP + (size_t)R1 * L.Kp32; a2.G = GP + (size_t)R1 * L.Np32; a2.M = rows - R1; a2.rows_per_split = cps2 * 64;
    a2.slabW = a1.slabW + (size_t)n1 * stride; a2.slabB = a1.slabB + (size_t)n1 * L.JT * 16; a2.rowscale = rowscale + R1;
    return IWAE_OK;
}

int dense_dx(iwae_model* m, Linear& L, const uint16_t* GP, int rows, const uint16_t* ACT, uint16_t* YP, float* YF) {
    DenseArgs a;
    memset(&a, 0, sizeof(a));
    a.X = GP; a.ldX = L.Np32; a.img = L.imgB;
    a.split = 1 << 30;
    a.M = rows; a.KT = L.KT_B; a.MG = L.MG_B; a.mg_per_block = (rows <= 8192) ? 1 : L.MG_B; a.Np32 = L.Kp32; a.g1_mask = m->dense_g1_mask;
    a.YP = YP; a.ldYP = L.Kp32; a.YF = YF; a.ldYF = L.Kp32;
    a.ACT = ACT; a.ldACT = L.Kp32;
    CHK(attach_dense_stamps(m, ACT ? EPI_DX : EPI_F32, a));
    launch_dense(ACT ? EPI_DX : EPI_F32, a, m->stream);
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

// backward of one BasicBlock over R rows: the dX chain first, then the three weight gradients -- they only feed the
// slab reduction, so for small R (latency-bound 8-wave kernels) they go out as ONE grouped launch
// dx_done: the dX chain (dhead -> d2 -> d1 -> dx) has been computed already (gblock_bwd_kernel): only the weight gradients are left
// lat != null: the block's dhead rows are made inside block_bwd_kernel (latent_bwd_kernel's sums, a wave per image) -- *lat_taken says whether
// that happened (else the caller launches latent_bwd_kernel first and calls again without it)
int block_bwd(iwae_model* m, Linear* blk, BlockWs& w, const uint16_t* inP, int R, bool need_dx, bool wgrad_on_side, bool dx_done = false, hipStream_t side_st = nullptr,
              bool skip_wgrad = false, const LatentBwdArgs* lat = nullptr, bool* lat_taken = nullptr) {
    if (lat_taken) *lat_taken = false;
    bool chain_fused = dx_done;
    if (dx_done) need_dx = false;
    if (!dx_done && m->allow_block_fused && !blk[2].kmajor && !blk[1].kmajor && blk[1].Np32 == blk[0].Np32 && blk[1].Kp32 == blk[0].Np32 && blk[2].Kp32 == blk[0].Np32) {
        BlockBwdArgs b;      // few rows (the encoder on the batch's images): both dX products in one launch
        memset(&b, 0, sizeof(b));
        b.DH = ptr<uint16_t>(w.dheadP); b.ldDH = blk[2].Np32; b.imgH = blk[2].imgB; b.imgL2 = blk[1].imgB;
        b.KTH = blk[2].KT_B; b.KT1 = blk[1].KT_B; b.NT1 = blk[0].Np32 / 16; b.R = R;
        b.H2 = ptr<uint16_t>(w.h2P); b.H1 = ptr<uint16_t>(w.h1P); b.ldH = blk[0].Np32;
        b.D2 = ptr<uint16_t>(w.d2P); b.D1 = ptr<uint16_t>(w.d1P);
        if (block_bwd_ok(b)) {
            if (lat && lat->Dp <= 128 && lat->Dp == blk[2].Np32 / 2 && !lat->prior_head && !lat->DHF) {
                b.lat_on = 1; b.lat = *lat; if (lat_taken) *lat_taken = true;
                b.rows_per_wg = lat->k > 16 ? 4 : 16;      // (many samples per image: four waves per image, 4 images per workgroup -- block_bwd_kernel<4>)
            }
            else if (lat) return IWAE_OK;      // (not taken: nothing launched)
            launch_block_bwd(b, m->stream); chain_fused = true;
        } else if (lat) return IWAE_OK;
    } else if (lat) return IWAE_OK;
    if (!chain_fused) {
        CHK(dense_dx(m, blk[2], ptr<uint16_t>(w.dheadP), R, ptr<uint16_t>(w.h2P), ptr<uint16_t>(w.d2P), nullptr));
        CHK(dense_dx(m, blk[1], ptr<uint16_t>(w.d2P), R, ptr<uint16_t>(w.h1P), ptr<uint16_t>(w.d1P), nullptr));
    }
    if (need_dx) CHK(dense_dx(m, blk[0], ptr<uint16_t>(w.d1P), R, nullptr, nullptr, ptr<float>(w.dx)));
    if (skip_wgrad) { HIPCHK(hipGetLastError()); return IWAE_OK; }      // (the caller takes the weight gradients: block_wgrad_rows)
    const uint16_t* xs[3] = {ptr<uint16_t>(w.h2P), ptr<uint16_t>(w.h1P), inP};
    const uint16_t* gs[3] = {ptr<uint16_t>(w.dheadP), ptr<uint16_t>(w.d2P), ptr<uint16_t>(w.d1P)};
    Linear* ls[3] = {&blk[2], &blk[1], &blk[0]};
    // The weight gradients only feed the slab reduction.  For the per-sample blocks of the 2-layer model (R = B*k rows)
    // they go to the side stream, ordered behind this block's dX chain, so the main stream's dependency chain does not
    // wait for them; the small encoder block (R = B) stays on the main stream (it is the tail of the step anyway).
    hipStream_t ws = m->stream;
    if (wgrad_on_side) {
        ws = side_st ? side_st : m->side;
        if (!dx_done) HIPCHK(hipEventRecord(m->ev_blk, m->stream));      // (dx_done: the event rode on gblock_bwd_kernel's dispatch packet)
        HIPCHK(hipStreamWaitEvent(ws, m->ev_blk, 0));
    }
    WgradPGroup g;
    memset(&g, 0, sizeof(g));
    int nsplit[3], nw[3];
    bool small = true;
    for (int i = 0; i < 3; ++i) {
        CHK(wgradp_plan(m, *ls[i], xs[i], gs[i], R, g.a[i], nsplit[i], nw[i]));
        small = small && nw[i] == 8;
    }
    if (small) {
        g.n = 3;
        for (int i = 0; i < 3; ++i) {
            g.gx[i] = (ls[i]->JT + 7) / 8; g.gy[i] = (ls[i]->IT + 15) / 16;
            g.zbeg[i + 1] = g.zbeg[i] + nsplit[i];
        }
        launch_wgradp_group(g, ws);
    } else {
        for (int i = 0; i < 3; ++i) launch_wgradp(g.a[i], nsplit[i], nw[i], ws);
    }
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

// Few rows (the image encoder: R = batch size): the block's three weight gradients with the whole row reduction inside each workgroup and
// (fuse) the Adam update in the epilogue -- one launch instead of the grouped weight gradient + slabs + reduce_grads_kernel (round 4).
// with_means: one extra block turns the step's per-image values into its batch means (what reduce_grads_kernel's extra block did).
bool wgrad_rows_ok(const iwae_model* m, const Linear* blk, int R) {
    return m->allow_wgrad_rows && R <= 2048 && blk[0].nsub == 1 && blk[1].nsub == 1 && blk[2].nsub <= 2 && !blk[0].kmajor && !blk[1].kmajor && !blk[2].kmajor;      // (its epilogue writes MG-major images)
}
// Few DATA rows too (M = B * k <= 2 048: the reference's default regime, B = 20): the decoder's three weight gradients join the encoder's in the
// SAME launch (six jobs; the output layer's G = the stored s takes its row weights on the way in) -- no side-stream launches, no slabs, no
// deferred reduction, no cross-stream events in the whole backward pass.  1-layer model only (the 2-layer model's per-sample blocks keep their path).
bool dec_rows_step(const iwae_model* m, int M, int B) {
    return m->allow_dec_rows && m->cfg.n_layers == 1 && M <= 2048 && wgrad_rows_ok(m, m->enc1, B) && m->dec1[0].nsub == 1 && m->dec1[1].nsub == 1 &&
           m->dec1[2].nsub == 1 && !m->dec1[0].kmajor && !m->dec1[1].kmajor;
}
int block_wgrad_rows(iwae_model* m, Linear* blk, BlockWs& w, const uint16_t* inP, int R, float alpha, bool fuse, bool with_means, bool with_decoder = false) {
    if (m->descs_dirty) CHK(build_descs(m));
    const uint16_t* xs[6] = {ptr<uint16_t>(w.h2P), ptr<uint16_t>(w.h1P), inP, ptr<uint16_t>(m->wdec1.g2P), ptr<uint16_t>(m->wdec1.g1P), ptr<uint16_t>(m->zP[0])};
    const uint16_t* gs[6] = {ptr<uint16_t>(w.dheadP), ptr<uint16_t>(w.d2P), ptr<uint16_t>(w.d1P), ptr<uint16_t>(m->wdec1.dlP), ptr<uint16_t>(m->wdec1.d2P), ptr<uint16_t>(m->wdec1.d1P)};
    Linear* ls[6] = {&blk[2], &blk[1], &blk[0], &m->dec1[2], &m->dec1[1], &m->dec1[0]};
    const int njobs = with_decoder ? 6 : 3;
    WgradRowsJob jobs[6];
    memset(jobs, 0, sizeof(jobs));
    for (int i = 0; i < njobs; ++i) {
        jobs[i].X = xs[i]; jobs[i].ldX = ls[i]->Kp32; jobs[i].G = gs[i]; jobs[i].ldG = ls[i]->Np32; jobs[i].R = i < 3 ? R : m->M;
        jobs[i].sub0 = ls[i]->sub[0];
        jobs[i].sub1 = ls[i]->nsub == 2 ? ls[i]->sub[1] : -1;
        jobs[i].split = ls[i]->nsub == 2 ? ls[i]->joff[1] : (1 << 30);
    }
    if (with_decoder && m->s_mode) jobs[3].rowscale = ptr<float>(m->gx);      // the forward pass kept s: dl = g_r s is made on the way in (else dlP already holds dl)
    const bool two = m->cfg.n_layers == 2;
    launch_wgrad_rows(jobs, njobs, m->d_descs, m->grad, m->param, m->mom, m->vel, alpha, m->adam_b1, m->adam_b2, m->adam_eps, fuse ? 1 : 0,
                      with_means ? ptr<float>(m->per_b) : nullptr, m->B, two ? 1.f : m->beta, m->d_scalars, m->d_zero, m->stream);
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

bool is_device_ptr(const void* p, int device) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain host memory
    return at.type == hipMemoryTypeDevice && at.device == device;
}

int copy_in(iwae_model* m, DevBuf& dst, const void* src, size_t bytes) {
    CHK(ensure(dst, bytes, m->stream));
    HIPCHK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyDefault, m->stream));
    return IWAE_OK;
}

int copy_out(iwae_model* m, void* dst, const void* src, size_t bytes) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, m->stream));
    return IWAE_OK;
}

// ---------------------------------------------------------------- the forward pass
// fills eps buffer `par` with the draws of (step, current batch offset) for M data rows on stream gs
int draw_eps(iwae_model* m, int par, uint32_t step, int M, hipStream_t gs, int max_blocks = 0) {
    const int Mp = round_up(M, 128);
    iwae_model::EpsTag& tg = m->eps_tag[par];
    tg.valid = false;
    for (int l = 0; l < m->cfg.n_layers; ++l) {
        CHK(ensure(m->epsc[par][l], (size_t)Mp * m->Dp[l] * 4, m->stream));
        EpsSrc e = eps_src(m, l);
        e.user = nullptr; e.cache = nullptr; e.step = step;
        if (!(m->abl_skip & 16)) launch_eps_gen(e, M, m->D[l], eps_ld(m, l), ptr<float>(m->epsc[par][l]), gs, max_blocks);
    }
    HIPCHK(hipGetLastError());
    tg.valid = true; tg.step = step; tg.row_offset = (uint64_t)m->batch_offset * (uint64_t)m->k; tg.M = M;
    return IWAE_OK;
}

#define EPSM_STEPS 8
// few rows: draws of steps [step0, step0 + EPSM_STEPS) into multi-step buffer `buf` (stream order on gs protects the buffer: see iwae_model::epsm)
int draw_eps_multi(iwae_model* m, int buf, uint32_t step0, int M, hipStream_t gs) {
    const int Mp = round_up(M, 128);
    iwae_model::EpsMTag& tg = m->epsm_tag[buf];
    tg.valid = false;
    for (int l = 0; l < m->cfg.n_layers; ++l) {
        const size_t stride = (size_t)Mp * eps_ld(m, l);
        CHK(ensure(m->epsm[buf][l], (size_t)EPSM_STEPS * stride * 4, m->stream));
        EpsSrc e = eps_src(m, l);
        e.user = nullptr; e.cache = nullptr; e.step = step0;
        launch_eps_gen_multi(e, M, m->D[l], eps_ld(m, l), ptr<float>(m->epsm[buf][l]), EPSM_STEPS, stride, gs);
    }
    HIPCHK(hipGetLastError());
    tg.valid = true; tg.step0 = step0; tg.row_offset = (uint64_t)m->batch_offset * (uint64_t)m->k; tg.M = M;
    return IWAE_OK;
}
// the multi-step buffer that holds the draws of `step` for this batch shape, or -1
int epsm_find(const iwae_model* m, uint32_t step, int M) {
    const uint64_t ro = (uint64_t)m->batch_offset * (uint64_t)m->k;
    for (int i = 0; i < 2; ++i) {
        const iwae_model::EpsMTag& t = m->epsm_tag[i];
        if (t.valid && t.M == M && t.row_offset == ro && step - t.step0 < (uint32_t)EPSM_STEPS) return i;      // (unsigned: step >= step0)
    }
    return -1;
}

// the stream that carries the speculative draw of the NEXT step's noise: one that this step's backward pass orders behind the main stream
// and whose last event the next forward joins (see the call in forward_impl; m->early_wout must be decided)
hipStream_t eps_draw_stream(const iwae_model* m, int M) {
    if (dec_rows_step(m, M, m->B)) return m->stream;      // (that backward pass touches no side stream at all: the draw stays in stream order)
    return (m->allow_wg3 && m->use_side2 && m->side2 && M <= 4096 && m->early_wout) ? m->side2 : m->side;
}

// will this step's backward pass run the decoder's dX chain as dec_bwd_rows_kernel (few rows)?  Mirrors backward_impl's choice (which
// copes with a wrong answer: a pending log-mean-exp is then launched as lse_kernel after all); m->s_mode must be decided.
bool dec_bwd_rows_planned(const iwae_model* m, int M) {
    const Linear& L = m->dec1[2];
    if (!(m->s_mode && m->allow_dec_bwd && m->small_dec_bwd && M <= m->small_rows && out_bwd_has_s_mode(L.KT) && !m->want_stamps)) return false;
    if (!(L.kmajor && L.imgB && m->dec1[1].KT_B == L.KT && m->dec1[1].MG_B == (L.KT + 1) / 2 && m->dec1[0].KT_B == L.KT && m->dec1[1].Kp32 == L.Kp32 &&
          m->dec1[0].Np32 == L.Kp32)) return false;
    return M <= m->dec_rows_max && m->allow_block_fused && m->cfg.n_layers == 1;
}

int forward_impl(iwae_model* m, const float* x, int B, int k, float beta, const float* eps, int objective, bool bwd,
                 const iwae_tensors* want) {
    const bool from_ds = m->ds_start >= 0;
    if ((!x && !from_ds) || B <= 0 || k <= 0) return fail(IWAE_ERR_ARG, "forward: need x, B > 0, k > 0");
    if ((int64_t)B * k > (int64_t)1 << 30) return fail(IWAE_ERR_ARG, "forward: B*k too large");
    const bool two = m->cfg.n_layers == 2;
    m->B = B; m->k = k; m->M = B * k; m->beta = beta;
    m->bf16_side_used = true;
    if (m->f32_z_pending) { HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0)); m->f32_z_pending = false; }      // (ev_join is this path's too)
    m->lse_pending = false;      // (set again below if this step leaves its log-mean-exp to the backward pass)
    m->time_this = m->timing > 0 && (m->timing_calls++ % m->timing) == 0;
    m->Mp = round_up(m->M, 128); m->Bp = round_up(B, 128);
    const int M = m->M, Mp = m->Mp, Bp = m->Bp, X = m->X, Xp = m->Xp32, Xinp = m->Xinp;
    hipStream_t st = m->stream;
    const float* cond = nullptr;
    if (m->C > 0) {
        if (from_ds) {      // (x, y) from the resident set: the input kernel writes onehot(y) of the batch's images into m->cond (tasks/task05.py:296-322)
            if (!m->ds_has_labels) return fail(IWAE_ERR_STATE, "conditional model on the resident dataset: call iwae_dataset_set_labels first");
            CHK(ensure(m->cond, (size_t)B * m->C * 4, m->stream));
            m->cond_n = B; m->cond_row0 = 0;
        }
        if (m->cond_row0 + B > m->cond_n) return fail(IWAE_ERR_STATE, "conditional model: call iwae_set_condition with y for these images first");
        cond = ptr<float>(m->cond) + (size_t)m->cond_row0 * m->C;
    }
    m->user_eps = eps != nullptr;
    // ---- the step's N(0,1) draws: normally already there (prefetched by the previous training step), else drawn now
    // (round 4: a forward-only call on many rows -- the k = 5000 evaluator on bf16 operands -- also takes its draws from eps_gen_kernel, so that the
    // decoder kernel makes z itself in its prologue instead of sample_kernel drawing inline and writing z out: option zin_eval, off by default -- measured slower)
    const bool zin_eval = !bwd && !two && m->C == 0 && m->allow_zin && m->allow_zin_eval && (int64_t)B * k >= 8192 && m->allow_dec_fused && m->allow_bern_pipe;
    const bool keep_eps = !eps && (bwd || two || zin_eval);
    m->epsc_ptr[0] = m->epsc_ptr[1] = nullptr;
    // few data rows, training step, single-stream backward (dec_rows_step): the draws come from the multi-step buffers (one launch per EPSM_STEPS steps)
    const bool eps_multi = keep_eps && bwd && !two && m->allow_eps_multi && m->eval_k_total == 0 && dec_rows_step(m, M, B);
    if (eps_multi) {
        int bi = epsm_find(m, m->noise_step, M);
        if (bi < 0) {      // (first step, another batch shape, a jump of iwae_set_step: drawn now, in stream order)
            bi = (m->epsm_tag[0].valid && !m->epsm_tag[1].valid) ? 1 : 0;
            CHK(draw_eps_multi(m, bi, m->noise_step, M, st));
        }
        const size_t soff = (size_t)(m->noise_step - m->epsm_tag[bi].step0) * (size_t)round_up(M, 128);
        for (int l = 0; l < m->cfg.n_layers; ++l) m->epsc_ptr[l] = ptr<float>(m->epsm[bi][l]) + soff * eps_ld(m, l);
    } else
    if (keep_eps) {
        const int np = (m->epsc_par + 1) % 3;
        const uint64_t ro = (uint64_t)m->batch_offset * (uint64_t)k;
        iwae_model::EpsTag& tg = m->eps_tag[np];
        if (m->eval_k_total > 0 || !(tg.valid && tg.step == m->noise_step && tg.row_offset == ro && tg.M == M)) {
            CHK(join_side(m));          // a speculative draw into this slot may still be on a side stream
            if (m->side) HIPCHK(hipStreamSynchronize(m->side));
            if (m->side2) HIPCHK(hipStreamSynchronize(m->side2));
            CHK(draw_eps(m, np, m->noise_step, M, st));
            if (m->eval_k_total > 0) m->eval_tag_kill = np;      // (a k-chunk's draws: the tag does not describe them)
        }
        if (m->eval_tag_kill >= 0) { m->eps_tag[m->eval_tag_kill].valid = false; m->eval_tag_kill = -1; }
        if (bwd) m->epsc_par = np;      // (forward-only calls -- the evaluator's launches -- reuse ONE slot: stream order protects it, and three slots of 2^21 rows are 2.5 GB grown inside the first calls)
        for (int l = 0; l < m->cfg.n_layers; ++l) m->epsc_ptr[l] = ptr<float>(m->epsc[np][l]);
    }
    if (eps) CHK(copy_in(m, m->epsbuf, eps, (size_t)M * (m->D[0] + (two ? m->D[1] : 0)) * 4));
    CHK(ensure(m->xP, (size_t)Bp * Xinp * 2, st));
    const float* xf_pending = nullptr;
    if (from_ds) {
        // main.py:117-120 on the device: gather the batch by the epoch's order and binarise it on the fly
        launch_gather_binarize(ptr<uint8_t>(m->ds_data), ptr<int32_t>(m->ds_order), m->ds_start, m->ds_N, B, X, Xinp, Bp, m->cfg.seed,
                               m->ds_epoch, ptr<uint16_t>(m->xP), nullptr, st, m->C > 0 ? ptr<uint8_t>(m->ds_labels) : nullptr, m->C, m->C > 0 ? ptr<float>(m->cond) : nullptr);
        m->ds_start = -1;
    } else {
        const float* xd = x;
        if (!is_device_ptr(x, m->cfg.device)) {       // host batches are staged; device batches are read in place
            CHK(copy_in(m, m->xin, x, (size_t)B * X * 4));
            xd = ptr<float>(m->xin);
        }
        if (m->C == 0 && m->allow_block_fused) xf_pending = xd;      // the fused encoder kernel converts the rows itself
        else launch_prep_rows(xd, cond, B, X, m->C, Xinp, Bp, ptr<uint16_t>(m->xP), st);
    }

    // ---- encoder over images (iwae1.py:57 / iwae2.py:59)
    CHK(block_alloc(m, m->enc1, m->wenc1, B, Bp, bwd, false));
    {
    ScopedTimer tm_enc(m, T_ENC_FWD);
    if (xf_pending) {
        bool took = false;
        CHK(block_fwd(m, m->enc1, m->wenc1, ptr<uint16_t>(m->xP), B, xf_pending, X, &took));
        if (!took) {        // shapes the fused kernel does not cover: convert, then the three-launch path
            launch_prep_rows(xf_pending, nullptr, B, X, 0, Xinp, Bp, ptr<uint16_t>(m->xP), st);
            CHK(block_fwd(m, m->enc1, m->wenc1, ptr<uint16_t>(m->xP), B));
        }
    } else {
        CHK(block_fwd(m, m->enc1, m->wenc1, ptr<uint16_t>(m->xP), B));
    }
    }

    for (int i = 0; i < 6; ++i) CHK(ensure(m->rows[i], (size_t)Mp * 4, st));
    float* lpxz = ptr<float>(m->rows[0]);
    float* t1 = ptr<float>(m->rows[1]);   // 1-layer lpz   | 2-layer lpz1z2
    float* t2 = ptr<float>(m->rows[2]);   // 1-layer lqzx  | 2-layer lpz2
    float* t3 = ptr<float>(m->rows[3]);   //               | 2-layer lqz1x
    float* t4 = ptr<float>(m->rows[4]);   //               | 2-layer lqz2z1
    float* lqd = ptr<float>(m->rows[5]);

    // ---- z (z1) = mu + sigma*eps and its densities (iwae1.py:59,107,109)
    // the draws are kept when later kernels of this call need them again (backward, 2-layer densities)
    CHK(ensure(m->zP[0], (size_t)Mp * m->Dp[0] * 2, st));
    CHK(join_side(m));      // from here on: the prefetched noise, then the decoder's weights
    bool fuse_z = false, sample_in_block = false, chain = false;
    SampleArgs zin;
    memset(&zin, 0, sizeof(zin));
    if (m->has_prior) {     // p(z|y) = N(mu_p(y), sigma_p(y)) (tasks/task04.py:124): the prior block on the B condition rows
        const int Cp = round_up(m->C, 32);
        CHK(ensure(m->condP, (size_t)Bp * Cp * 2, st));
        launch_prep_rows(cond, nullptr, B, m->C, 0, Cp, Bp, ptr<uint16_t>(m->condP), st);
        CHK(block_alloc(m, m->prior, m->wprior, B, Bp, bwd, false));
        CHK(block_fwd(m, m->prior, m->wprior, ptr<uint16_t>(m->condP), B));
    }
    {
        SampleArgs s;
        memset(&s, 0, sizeof(s));
        s.head = ptr<float>(m->wenc1.head); s.ldH = 2 * m->Dp[0]; s.Dp = m->Dp[0]; s.D = m->D[0]; s.head_per_row = 0;
        s.M = M; s.Mp = Mp; s.k = k; s.B = B; s.eps = eps_src(m, 0);
        s.ZP = ptr<uint16_t>(m->zP[0]);
        s.cond = cond; s.C = m->C;
        s.prior_head = m->has_prior ? ptr<float>(m->wprior.head) : nullptr;
        s.lp_prior = two ? nullptr : t1;
        s.lq = two ? t3 : t2;
        const bool want_dreg = !two && (objective == OBJ_DREG || (!bwd && !m->in_eval_llh));    // tasks/task02.py:63-65
        s.lq_dreg = want_dreg ? lqd : nullptr;
        // 1-layer training step on the device's own noise: the first decoder layer makes z itself (dense_kernel ZIN mode)
        // (the DReG step too: the decoder kernel's prologue also sums the second log q; if that kernel turns out not to apply, sample_kernel runs after all)
        fuse_z = m->allow_zin && !two && m->C == 0 && (bwd || zin_eval) && s.eps.cache != nullptr && M >= 8192 &&
                 (m->dec1[0].KT == 4 || m->dec1[0].KT == 2) && m->dec1[0].Kp32 == m->Dp[0];
        // few rows: block_fwd_kernel (the decoder's two tanh layers in one launch, below) makes z itself -- one latency-bound launch less
        sample_in_block = m->allow_block_fused && m->allow_zin && !two && !fuse_z && M <= 4096 && !s.ZF && m->dec1[0].Kp32 == m->Dp[0];
        // 2-layer model at large row counts, the reference's dims: chain2_fwd_kernel (below) makes z1 itself, as its first layer's operand
        if (two) {
            const Linear *e2 = m->enc2, *d2 = m->dec2;
            chain = m->allow_chain2 && chain2_fwd_ok(e2[0].KT, e2[1].KT, d2[0].KT, M) && e2[0].Kp32 == m->Dp[0] && e2[0].Np32 == 32 * e2[1].KT &&
                    e2[1].Np32 == e2[0].Np32 && e2[2].KT == e2[1].KT && e2[2].Np32 == 2 * m->Dp[1] && d2[0].Kp32 == m->Dp[1] && d2[0].Np32 == e2[0].Np32 &&
                    d2[1].KT == e2[1].KT && d2[1].Np32 == d2[0].Np32 && d2[2].KT == e2[1].KT && d2[2].Np32 == 2 * m->Dp[0];
        }
        if (fuse_z || sample_in_block) zin = s;
        else if (!chain) launch_sample(s, st);
    }
    if (two) {
        // ---- q(z2|z1), z2, p(z1|z2)  (iwae2.py:63-65, :90, :118-124)
        CHK(block_alloc(m, m->enc2, m->wenc2, M, Mp, bwd, true));
        CHK(ensure(m->zP[1], (size_t)Mp * m->Dp[1] * 2, st));
        CHK(block_alloc(m, m->dec2, m->wdec2, M, Mp, bwd, true));
        // large row counts, the reference's dims: the z1 sampling, both per-sample blocks, the z2 sampling and the four log-densities in ONE launch
        const Linear *e2 = m->enc2, *d2 = m->dec2;
        m->chain2_bwd = false;
        if (chain) {
            Chain2FwdArgs c;
            memset(&c, 0, sizeof(c));
            c.Z1P = ptr<uint16_t>(m->zP[0]); c.lqz1x = t3;
            c.e_img1 = e2[0].imgF; c.e_img2 = e2[1].imgF; c.e_imgh = e2[2].imgF;
            c.d_img1 = d2[0].imgF; c.d_img2 = d2[1].imgF; c.d_imgh = d2[2].imgF;
            c.M = M; c.k = k; c.B = B; c.D0 = m->D[0]; c.D1 = m->D[1];
            // the backward pass reads the blocks' tanh activations; the float32 heads only where something still reads THEM: the unfused
            // backward kernels (gauss_bwd_kernel) and the z2 / snis exports -- gblock_bwd_kernel recomputes them from h2
            m->chain2_bwd = bwd && m->allow_chain2_bwd && gblock_bwd_ok(e2[0].KT, e2[1].KT, d2[0].KT, M) && e2[0].imgB && d2[0].imgB;
            const bool heads = want != nullptr || (bwd && !m->chain2_bwd);
            c.EH1 = bwd ? ptr<uint16_t>(m->wenc2.h1P) : nullptr; c.EH2 = bwd ? ptr<uint16_t>(m->wenc2.h2P) : nullptr;
            c.EHEAD = heads ? ptr<float>(m->wenc2.head) : nullptr;
            c.Z2P = ptr<uint16_t>(m->zP[1]);
            c.DH1 = bwd ? ptr<uint16_t>(m->wdec2.h1P) : nullptr; c.DH2 = bwd ? ptr<uint16_t>(m->wdec2.h2P) : nullptr;
            c.DHEAD = heads ? ptr<float>(m->wdec2.head) : nullptr;
            c.head1 = ptr<float>(m->wenc1.head); c.ldH1 = 2 * m->Dp[0];
            c.eps1 = eps_src(m, 0); c.eps2 = eps_src(m, 1);
            c.lpz1z2 = t1; c.lpz2 = t2; c.lqz2z1 = t4;
            launch_chain2_fwd(c, st);
        } else {
        CHK(block_fwd(m, m->enc2, m->wenc2, ptr<uint16_t>(m->zP[0]), M));
        SampleArgs s;
        memset(&s, 0, sizeof(s));
        s.head = ptr<float>(m->wenc2.head); s.ldH = 2 * m->Dp[1]; s.Dp = m->Dp[1]; s.D = m->D[1]; s.head_per_row = 1;
        s.M = M; s.Mp = Mp; s.k = k; s.B = B; s.eps = eps_src(m, 1);
        s.ZP = ptr<uint16_t>(m->zP[1]);
        s.lp_prior = t2; s.lq = t4; s.lq_dreg = nullptr;
        launch_sample(s, st);
        CHK(block_fwd(m, m->dec2, m->wdec2, ptr<uint16_t>(m->zP[1]), M));
        GaussLpArgs g;
        memset(&g, 0, sizeof(g));
        g.zhead = ptr<float>(m->wenc1.head); g.ldZH = 2 * m->Dp[0]; g.Dzp = m->Dp[0];
        g.phead = ptr<float>(m->wdec2.head); g.ldPH = 2 * m->Dp[0]; g.Dpp = m->Dp[0];
        g.D = m->D[0]; g.M = M; g.k = k; g.eps = eps_src(m, 0); g.out = t1;
        launch_gauss_lp(g, st);
        }
    }

    // log_w / log-mean-exp arguments (lse_kernel, or the decoder kernel where it does that itself); allocates the outputs
    LseArgs la;
    auto lse_args = [&](LseArgs& a) -> int {
        memset(&a, 0, sizeof(a));
        CHK(ensure(m->logw, (size_t)Mp * 4, st));
        CHK(ensure(m->wn, (size_t)Mp * 4, st));
        {   // the row weights are also read 64 at a time by the output layer's weight gradient: pad rows must stay finite
            const void* before = m->gx.p;
            CHK(ensure(m->gx, (size_t)Mp * 4, st));
            if (m->gx.p != before) HIPCHK(hipMemsetAsync(m->gx.p, 0, m->gx.cap, st));
        }
        CHK(ensure(m->cf, (size_t)Mp * 16, st));
        CHK(ensure(m->per_b, (size_t)PB_COUNT * B * 4, st));
        if (!two) {
            a.term[0] = lpxz; a.coef[0] = 1.f;
            a.term[1] = t1; a.coef[1] = beta;
            a.term[2] = t2; a.coef[2] = -beta;
            a.head = ptr<float>(m->wenc1.head); a.ldH = 2 * m->Dp[0]; a.D = m->D[0]; a.Dp = m->Dp[0];
            a.cz_on = 1.f;
        } else {
            a.term[0] = lpxz; a.coef[0] = 1.f;    // iwae2.py:128 (beta unused there)
            a.term[1] = t1; a.coef[1] = 1.f;
            a.term[2] = t2; a.coef[2] = 1.f;
            a.term[3] = t3; a.coef[3] = -1.f;
            a.term[4] = t4; a.coef[4] = -1.f;
            a.head = nullptr;
            a.cz_on = 0.f;
        }
        a.lq_dreg = (!two && (objective == OBJ_DREG || (!bwd && !m->in_eval_llh))) ? lqd : nullptr;
        a.B = B; a.k = k; a.beta = two ? 1.f : beta; a.objective = objective;
        a.lme_only = (!bwd && m->in_eval_llh && !want) ? 1 : 0;
        a.logw = ptr<float>(m->logw); a.wn = ptr<float>(m->wn); a.gx = ptr<float>(m->gx);
        a.cf = ptr<float4>(m->cf); a.per_b = ptr<float>(m->per_b);
        a.n_px_part = 1; a.px_stride = (size_t)Mp; a.term0_out = lpxz;
        return IWAE_OK;
    };

    // ---- decoder + Bernoulli log-likelihood (iwae1.py:81-83,111)
    MlpWs& w = m->wdec1;
    const int Hp = m->dec1[0].Np32;
    CHK(ensure(w.g1P, (size_t)Mp * Hp * 2, st));
    CHK(ensure(w.g2P, (size_t)Mp * Hp * 2, st));
    {
        Linear& L = m->dec1[2];
        DenseArgs a;
        memset(&a, 0, sizeof(a));
        a.X = ptr<uint16_t>(w.g2P); a.ldX = L.Kp32; a.img = L.imgF;
        a.split = 1 << 30;
        a.M = M; a.KT = L.KT; a.MG = L.MG; a.mg_per_block = L.MG; a.Np32 = L.Np32; a.g1_mask = m->dense_g1_mask;
        // small row counts (the reference's default B = 20, k = 5): a handful of workgroups walking all 13 pixel groups
        // in turn is 40 us of latency -- one pixel group per block instead, log p(x|z) as per-group partial sums that
        // lse_kernel adds up in fixed order
        m->px_parts = 1;
        // few rows: the output layer runs inside block_fwd_kernel, behind the two tanh layers of the same 16-row tile (one launch for the
        // whole decoder; log p(x|z) per row comes out whole, not as per-group partial sums)
        const bool out_in_block = m->allow_block_fused && m->allow_out_in_block && !fuse_z && M <= 4096 && !(want && want->logits) && !m->want_stamps &&
                                  m->dec1[1].Np32 == m->dec1[0].Np32 && m->dec1[1].Kp32 == m->dec1[0].Np32 && L.Kp32 == m->dec1[1].Np32 && L.KT == m->dec1[1].KT && L.KT <= 8;
        if (M < 8192 && L.MG > 1 && !out_in_block) {
            a.mg_per_block = 1;
            m->px_parts = L.MG;
            CHK(ensure(m->px_part, (size_t)L.MG * Mp * 4, st));
        }
        a.XB = ptr<uint16_t>(m->xP); a.ldXB = Xinp; a.k = k; a.B = B; a.Xdim = X;
        a.lpxz = m->px_parts > 1 ? ptr<float>(m->px_part) : lpxz; a.lpxz_stride = m->px_parts > 1 ? (size_t)Mp : 0;
        // training step: keep s = x - sigmoid(l) for the backward pass (out_bwd_s_kernel, output-layer weight gradient)
        m->s_mode = bwd && m->allow_s_mode && out_bwd_has_s_mode(L.KT);
        if (m->s_mode) {
            CHK(ensure(m->wdec1.dlP, (size_t)Mp * Xp * 2, st));
            a.YP = ptr<uint16_t>(m->wdec1.dlP); a.ldYP = Xp;
            if (m->fake_s & 4) a.dbg = 32;
            if (m->fake_s & 16) a.dbg |= 64;      // (the decoder kernel's tanh layers without their weight DMA)
            if (m->fake_s & 32) a.dbg |= 128;     // (phase exits of the decoder kernel, for instruction counters: behind the prologue,
            if (m->fake_s & 64) a.dbg |= 256;     //  behind both tanh layers,
            if (m->fake_s & 128) a.dbg |= 512;    //  behind the first)
        }
        a.logits_out = nullptr;
        a.pipe = m->allow_bern_pipe ? 1 : 0;
        if (a.pipe && m->bern_qw) {      // the 16-wave / 200-row shape is one workgroup per CU: only where its last round is nearly full
            const int nwg = (M + 199) / 200, ncu = std::max(1, m->num_cus);
            const int rounds = (nwg + ncu - 1) / ncu;
            if ((double)nwg >= 0.9 * (double)rounds * ncu || m->bern_qw_force) a.pipe = 2;
        }
        if (want && want->logits) {
            CHK(ensure(m->scratch, (size_t)M * X * 4, st));
            a.logits_out = ptr<float>(m->scratch);
        }
        // The whole decoder in one launch (bern_pipe_kernel<.., PRE>) where that kernel exists: the two tanh layers' activations
        // stay in registers from layer to layer (z made in the kernel when the step runs on the device's noise).
        bool fuse_dec = false;
        if (m->allow_dec_fused && a.pipe && m->C == 0 && m->dec1[0].KT <= 4 && m->dec1[0].Kp32 == m->Dp[0] &&
            m->dec1[0].Np32 == L.Kp32 && m->dec1[1].Kp32 == L.Kp32 && m->dec1[1].Np32 == L.Kp32) {
            a.pre_img1 = m->dec1[0].imgF; a.pre_KT1 = m->dec1[0].KT; a.pre_img2 = m->dec1[1].imgF;
            a.pre_Z = ptr<uint16_t>(m->zP[0]);
            // g1, g2 are kept for the backward pass only: a forward-only call (val_step, the k = 5000 evaluator) never reads them back
            a.pre_G1 = (bwd && !(m->fake_s & 8)) ? ptr<uint16_t>(w.g1P) : nullptr; a.pre_G2 = (bwd && !(m->fake_s & 8)) ? ptr<uint16_t>(w.g2P) : nullptr;      // (fake_s & 8, DIAG builds: timing without the activation stores)
            if (fuse_z) {
                a.zhead = zin.head; a.ldZH = zin.ldH; a.zeps = zin.eps.cache; a.zldE = zin.eps.ldC; a.zD = zin.D; a.zDp = zin.Dp;
                a.ZPout = (bwd && !(m->fake_s & 8)) ? zin.ZP : nullptr; a.zlp = zin.lp_prior; a.zlq = zin.lq; a.zlq_dreg = zin.lq_dreg;      // (a forward-only call never reads z back)
            }
            fuse_dec = bern_pipe_ok(a);
            if (!fuse_dec) {
                a.pre_img1 = a.pre_img2 = nullptr; a.pre_Z = nullptr; a.pre_G1 = a.pre_G2 = nullptr; a.pre_KT1 = 0;
                a.zhead = nullptr; a.zeps = nullptr; a.ZPout = nullptr; a.zlp = a.zlq = nullptr; a.zlq_dreg = nullptr;
            }
        }
        if (fuse_z && zin.lq_dreg && !fuse_dec) { launch_sample(zin, st); fuse_z = false; }      // (dense_kernel's sampled-input mode has no DReG sum)
        if (fuse_dec && sample_in_block) { launch_sample(zin, st); sample_in_block = false; }
        bool out_done = false;
        if (!fuse_dec) {
            // few rows: the two tanh layers as ONE launch of block_fwd_kernel (a BasicBlock without its head: 16-row workgroups,
            // weights straight from the L2-resident images) instead of two latency-bound dense_kernel launches
            bool two_in_one = false;
            if (m->allow_block_fused && !fuse_z && M <= 4096 && m->dec1[1].Np32 == m->dec1[0].Np32 && m->dec1[1].Kp32 == m->dec1[0].Np32) {
                BlockFwdArgs bf;
                memset(&bf, 0, sizeof(bf));
                bf.X = ptr<uint16_t>(m->zP[0]); bf.ldX = m->dec1[0].Kp32; bf.img0 = m->dec1[0].imgF; bf.img1 = m->dec1[1].imgF; bf.img2 = m->dec1[1].imgF;
                bf.KT0 = m->dec1[0].KT; bf.KT1 = m->dec1[1].KT; bf.NT1 = m->dec1[0].Np32 / 16; bf.NT2 = 0; bf.R = M;
                bf.H1 = ptr<uint16_t>(w.g1P); bf.H2 = ptr<uint16_t>(w.g2P); bf.ldH = m->dec1[0].Np32;
                bf.YF = nullptr; bf.ldYF = 0; bf.split = 1 << 30;
                if (sample_in_block) { bf.sample = 1; bf.S = zin; }
                if (out_in_block) {
                    bf.oimg = L.imgF; bf.oXdim = X; bf.oH = L.Np32 >> 5; bf.oXB = ptr<uint16_t>(m->xP); bf.oldXB = Xinp; bf.ok = k;
                    bf.oSP = m->s_mode ? ptr<uint16_t>(m->wdec1.dlP) : nullptr; bf.oldS = Xp; bf.olpxz = lpxz;
                }
                if (block_fwd_ok(bf)) {
                    ScopedTimer tm(m, T_DEC_FWD);
                    launch_block_fwd(bf, st); two_in_one = true; sample_in_block = false; out_done = out_in_block;
                }
            }
            if (sample_in_block) { launch_sample(zin, st); sample_in_block = false; }      // (shapes the fused kernel does not cover)
            if (!two_in_one) {
                CHK(dense_fwd(m, m->dec1[0], EPI_TANH, ptr<uint16_t>(m->zP[0]), M, ptr<uint16_t>(w.g1P), nullptr, 0, fuse_z ? &zin : nullptr));
                CHK(dense_fwd(m, m->dec1[1], EPI_TANH, ptr<uint16_t>(w.g1P), M, ptr<uint16_t>(w.g2P), nullptr, 0));
            }
        }
            CHK(attach_dense_stamps(m, EPI_BERN, a));
            m->early_wout = bwd && m->s_mode && m->allow_early_wout && !dec_rows_step(m, M, B);      // (few data rows: no side-stream work in the backward pass at all)      // (round 3: the 2-layer model too -- its weight gradients are 220 us of kernels, on ONE side stream behind dec_bwd they ended 100 us after the main stream)
            // Round 3: where the decoder kernel's workgroups own whole images (16-wave / 200-row shape, k a divisor of 200) it also does
            // lse_kernel's work for them -- the backward pass starts right behind it: one launch (7 us) and one dispatch gap (6 us) less
            // on the loop that sets the step, and no second lse_kernel on the side stream.
            CHK(lse_args(la));
            m->lse_fused = false;
            if (!out_done && fuse_dec && m->allow_lse_fused && !m->want_stamps) {
                a.lse = la; a.lse_on = 1;
                if (bern_lse_ok(a)) m->lse_fused = true;
                else { a.lse_on = 0; memset(&a.lse, 0, sizeof(a.lse)); }
            }
            // round 4: with the row weights made inside the decoder kernel, it also leaves g2w = bf16(g_r g2) -- the output layer's weight gradient
            // (forked right behind this kernel) then needs no row weighting.  Needs a pad column in the hidden width for g_r itself (the bias gradient).
            const bool g2w_now = m->lse_fused && m->early_wout && m->allow_g2w && m->s_mode && m->dec1[2].Kin < m->dec1[2].Kp32 && !two;
            if (g2w_now) {
                CHK(ensure(w.g2wP, (size_t)Mp * Hp * 2, st));
                a.G2W = ptr<uint16_t>(w.g2wP); a.g2w_feat = m->dec1[2].Kin;
            }
            if (bwd && g2w_now != m->g2w) { m->g2w = g2w_now; m->descs_dirty = true; }      // (the layer table says where the output layer's bias sums are)
            // s-mode training step: the output layer's weight gradient needs s, g2 and the row weights -- not out_bwd -- so the
            // side stream forks early.  Round 2: it forks behind THIS kernel (event on its dispatch packet) and runs its own copy of
            // lse_kernel (7 us, a few waves) for the row weights, instead of forking behind the main stream's lse_kernel: the ~12 us
            // a cross-stream hand-off takes now pass beside the main stream's lse_kernel, not behind it.
            m->lse_dup = m->early_wout && m->allow_lse_dup && m->px_parts == 1 && !out_done && !m->lse_fused;
            const bool fork_here = m->lse_dup || (m->lse_fused && m->early_wout);
            if (!out_done) { ScopedTimer tm(m, T_DEC_FWD); if (fork_here && !m->time_this) set_launch_stop_event(m->ev_lse); launch_dense(EPI_BERN, a, st); }
            if (fork_here && m->time_this) HIPCHK(hipEventRecord(m->ev_lse, st));      // (a timed step: the timer's stop event sits behind the kernel)
        HIPCHK(hipGetLastError());
        // The NEXT step's noise (speculating step + 1 with the same batch shape; the tag is checked on use): drawn now, on the side
        // stream, idle until the backward pass forks -- enqueued behind the decoder kernel so that its dispatch does not delay that one
        // (few rows, where the backward pass launches the decoder's weight gradients as one group on the SECOND side stream and `side` carries nothing
        // that a later event would cover: the draw goes to that second stream, in front of the group and the decoder update whose event the next step waits for.
        // Round 4 (advisor finding): the choice must follow what the backward pass will really use.  It touches `side2` only when the output layer's
        // gradient forks early (early_wout, known here); without that -- hidden widths without a stored-s instantiation, options out_recompute /
        // no_early_wout -- everything runs on `side`, the stream whose event the next step joins and which is re-ordered behind the main stream every
        // step (the ring slot written here was last read by step t - 2's backward pass on the main stream).)
        if (eps_multi) {      // the next GROUP of steps, once per group: into the buffer the current step does not read (main stream: this regime touches no other)
            if (epsm_find(m, m->noise_step + 1, M) < 0) CHK(draw_eps_multi(m, 1 - epsm_find(m, m->noise_step, M), m->noise_step + 1, M, st));
        } else
        if (bwd && keep_eps && m->side)
            CHK(draw_eps(m, (m->epsc_par + 1) % 3, m->noise_step + 1, M, eps_draw_stream(m, M), m->eps_blocks));
        if (want && want->logits) CHK(copy_out(m, want->logits, m->scratch.p, (size_t)M * X * 4));
    }

    // ---- log_w, log-mean-exp over k, objectives (iwae1.py:113-139)
    {
        LseArgs a = la;
        if (m->px_parts > 1) a.term[0] = ptr<float>(m->px_part);
        a.n_px_part = m->px_parts;
        if (m->lse_fused) {      // the decoder kernel did it; the side stream (output layer's weight gradient) forks behind that kernel
            if (m->early_wout) HIPCHK(hipStreamWaitEvent(m->side, m->ev_lse, 0));
        } else {
        // s-mode training step: the output layer's weight gradient needs s, g2 and the row weights lse_kernel leaves -- not
        // out_bwd -- so the side stream forks here (ev_lse on this kernel's dispatch packet), one kernel earlier, and the
        // gradient runs beside out_bwd (both read s)
        // Few rows (round 4): the training step's log-mean-exp is done by the backward pass's first kernel (dec_bwd_rows_kernel, a wave per
        // image in front of its own work): one dependent launch less on a chain of ~10 us launches
        m->lse_pending = bwd && m->allow_lse_in_bwd && !m->lse_dup && !m->early_wout && dec_bwd_rows_planned(m, M);
        if (m->lse_pending) m->lse_saved = a;
        if (m->early_wout && !m->lse_dup) set_launch_stop_event(m->ev_lse);
        if (!m->lse_pending) launch_lse(a, st);
        if (m->lse_dup) {       // the side stream's copy: same inputs, its own outputs
            CHK(ensure(m->logw2, (size_t)Mp * 4, st));
            CHK(ensure(m->wn2, (size_t)Mp * 4, st));
            {
                const void* before = m->gx2.p;
                CHK(ensure(m->gx2, (size_t)Mp * 4, st));
                if (m->gx2.p != before) { HIPCHK(hipMemsetAsync(m->gx2.p, 0, m->gx2.cap, st)); HIPCHK(hipStreamSynchronize(st)); }
            }
            CHK(ensure(m->cf2, (size_t)Mp * 16, st));
            CHK(ensure(m->per_b2, (size_t)PB_COUNT * B * 4, st));
            LseArgs a2 = a;
            a2.logw = ptr<float>(m->logw2); a2.wn = ptr<float>(m->wn2); a2.gx = ptr<float>(m->gx2);
            a2.cf = ptr<float4>(m->cf2); a2.per_b = ptr<float>(m->per_b2);
            HIPCHK(hipStreamWaitEvent(m->side, m->ev_lse, 0));
            launch_lse(a2, m->side);
        }
        }
        // batch means: a training step folds them into its last kernel (backward_impl), a forward-only call takes them here
        if (!bwd) launch_scalars(ptr<float>(m->per_b), B, two ? 1.f : beta, m->d_scalars, st);
    }
    HIPCHK(hipGetLastError());
    m->have_forward = true;
    m->fwd_was_f32 = false;
    return IWAE_OK;
}

// ---------------------------------------------------------------- the backward pass
float adam_alpha(iwae_model* m, float lr) {      // keras Adam: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); advances t
    m->adam_t += 1;
    const double t = (double)m->adam_t;
    return (float)((double)lr * sqrt(1.0 - pow((double)m->adam_b2, t)) / (1.0 - pow((double)m->adam_b1, t)));
}

// fused_lr >= 0: the optimizer update runs inside the slab reduction (single-GPU train step); < 0: gradient only
// hold_early (the in-library data-parallel step): the decoder's slab reduction is NOT launched here but by dp_finish, behind the wait that
// orders the step's two all-reduces (m->early_held)
int backward_impl(iwae_model* m, int objective, float fused_lr = -1.0f, bool split = false, bool hold_early = false) {
    if (!m->have_forward) return fail(IWAE_ERR_STATE, "backward without forward");
    if (m->fwd_was_f32) return fail(IWAE_ERR_STATE, "the last forward ran in float32 mode");
    const bool two = m->cfg.n_layers == 2;
    const int B = m->B, k = m->k, M = m->M, Mp = m->Mp, Bp = m->Bp, X = m->X, Xp = m->Xp32;
    hipStream_t st = m->stream;
    MlpWs& w = m->wdec1;
    const int Hp = m->dec1[0].Np32;
    CHK(ensure(w.dlP, (size_t)Mp * Xp * 2, st));
    CHK(ensure(w.d2P, (size_t)Mp * Hp * 2, st));
    CHK(ensure(w.d1P, (size_t)Mp * Hp * 2, st));
    CHK(ensure(w.dz, (size_t)Mp * m->Dp[0] * 4, st));
    bool fused_dx = false, dz_half = false;
    const bool dec_rows = dec_rows_step(m, M, B);      // the decoder's weight gradients ride in the encoder's wgrad_rows_kernel launch (few data rows)
    {
        Linear& L = m->dec1[2];
        OutBwdArgs a;
        memset(&a, 0, sizeof(a));
        a.G2 = ptr<uint16_t>(w.g2P); a.ldG = L.Kp32; a.img1 = L.imgF; a.img2 = L.imgB;
        a.Xdim = X; a.Xp32 = Xp;
        a.gx = ptr<float>(m->gx); a.XB = ptr<uint16_t>(m->xP); a.ldXB = m->Xinp; a.k = k;
        a.M = M; a.KT = L.KT; a.NG = L.MG;
        a.DPP = ptr<uint16_t>(w.d2P);
        if (m->s_mode) a.SP = ptr<uint16_t>(w.dlP);     // dlP holds s: one product, no recompute
        if (m->fake_s & 1) a.dbg = 32;
        const bool small_fused = m->small_dec_bwd && m->allow_dec_bwd && m->s_mode && M <= m->small_rows && out_bwd_has_s_mode(L.KT) && !m->want_stamps;      // dec_bwd_kernel at small row counts too
        if (m->s_mode && M < 8192 && L.MG > 1 && !small_fused) {         // small row counts: one pixel group per block, partial sums + finish kernel
            a.gpb = 1;
            CHK(ensure(m->dg2_part, (size_t)L.MG * M * L.Kp32 * 4, st));
            a.part = ptr<float>(m->dg2_part);
        }
        else a.DLP = ptr<uint16_t>(w.dlP);                // recompute mode: out_bwd writes dl = gx * s there
        if (m->want_stamps && L.KT == 7) {
            CHK(ensure(m->stamps, (size_t)(Mp / 64) * 4 * 8 * 8, st));
            a.stamps = ptr<unsigned long long>(m->stamps);
        }
            // One launch for out_bwd + dX of d2 + dX of d1 (dec_bwd_kernel) where it exists: large row counts, s kept by the forward
            // pass, hidden width with an instantiation; dpre2 / dpre1 stay in registers from product to product.
            fused_dx = m->allow_dec_bwd && m->s_mode && !a.part && (M >= 8192 || small_fused) && !a.stamps && L.kmajor && L.imgB && m->dec1[1].KT_B == L.KT && m->dec1[1].MG_B == (L.KT + 1) / 2 &&
                       m->dec1[0].KT_B == L.KT && m->dec1[1].Kp32 == L.Kp32 && m->dec1[0].Np32 == L.Kp32;
            bool rows_kernel = false;
            if (fused_dx && M <= m->dec_rows_max && m->allow_block_fused && L.kmajor && L.imgB) {      // few rows: 16-row workgroups, weights straight from L2
                DecBwdRowsArgs r;
                memset(&r, 0, sizeof(r));
                r.SP = ptr<uint16_t>(w.dlP); r.ldS = Xp; r.KTX = Xp / 32; r.imgK3 = L.imgB; r.MT3 = L.MT_B;
                r.imgB2 = m->dec1[1].imgB; r.imgB1 = m->dec1[0].imgB; r.KT = L.KT; r.NT1 = L.Kp32 / 16; r.NT3 = m->dec1[0].Kp32 / 16; r.M = M;
                r.G2 = ptr<uint16_t>(w.g2P); r.G1 = ptr<uint16_t>(w.g1P); r.ldH = L.Kp32; r.gx = ptr<float>(m->gx);
                r.D2P = ptr<uint16_t>(w.d2P); r.D1P = ptr<uint16_t>(w.d1P); r.ldDZ = m->dec1[0].Kp32;
                dz_half = !two && m->allow_dz_half;
                r.DZ = ptr<float>(w.dz); r.DZH = dz_half ? (uint16_t*)w.dz.p : nullptr;
                if (dec_bwd_rows_ok(r)) {
                    if (m->lse_pending) { r.lse_on = 1; r.lse = m->lse_saved; m->lse_pending = false; }      // (forward_impl left the log-mean-exp to this kernel)
                    ScopedTimer tm(m, T_DEC_BWD);
                    if (!dec_rows) set_launch_stop_event(m->ev_fork2);
                    launch_dec_bwd_rows(r, st);
                    rows_kernel = true;
                }
            }
            if (m->lse_pending) { launch_lse(m->lse_saved, st); m->lse_pending = false; }      // (the plan did not hold: lse_kernel after all, in front of everything that reads the row weights)
            if (rows_kernel) {
            } else if (fused_dx) {
                DecBwdArgs d;
                memset(&d, 0, sizeof(d));
                d.o = a;
                d.imgB2 = m->dec1[1].imgB; d.G1 = ptr<uint16_t>(w.g1P); d.D1P = ptr<uint16_t>(w.d1P);
                d.imgB1 = m->dec1[0].imgB; d.MG1 = m->dec1[0].MG_B; d.DZ = ptr<float>(w.dz); d.ldDZ = m->dec1[0].Kp32;
                // 1-layer model: dz has one reader (latent_bwd_kernel): bf16 halves its 26 MB each way (the 2-layer model adds two more
                // float32 terms to it there and keeps float32)
                dz_half = !two && m->allow_dz_half;
                if (dz_half) d.DZH = (uint16_t*)w.dz.p;
                d.nw = m->dec_bwd_nw;
                if (m->dstamp_epi == 9) {      // diagnostic (STAMPS=1 build, IWAE_DENSE_STAMPS=9:0): phase stamps of dec_bwd_kernel
                    m->dstamp_waves = ((M + 127) / 128) * (L.KT == 7 ? m->dec_bwd_nw : 4);
                    CHK(ensure(m->dstamps, (size_t)m->dstamp_waves * 64, st));
                    d.o.stamps = ptr<unsigned long long>(m->dstamps);
                }
                ScopedTimer tm(m, T_DEC_BWD);
                if (!dec_rows) set_launch_stop_event(m->ev_fork2);          // dpre2, dpre1 and the last read of the decoder's weight images: one event
                launch_dec_bwd(d, st);
            } else {
            // (forked behind lse_kernel already: the side stream then needs nothing from the main stream until dX of d1 is done)
            { ScopedTimer tm(m, T_OUT_BWD); if (!m->early_wout) set_launch_stop_event(m->ev_fork); launch_out_bwd(a, st); }
            }
        HIPCHK(hipGetLastError());
    }
    // fork: the decoder weight gradients only need what out_bwd produced (dl, dpre2) plus forward activations, so
    // they start on the side stream right behind it and fill the machine next to the dz -> encoder chain; the
    // first decoder layer's gradient additionally waits for dpre1 (second event).
    hipStream_t sd = m->side;
    // Few rows (round 3): the decoder's three weight gradients as ONE grouped launch behind the dX chain (the B = 20 step is bound by the host's
    // launches and the streams' hand-offs, not by these kernels: 13 -> 11 launches, two events less)
    bool group3 = false, split_upd = false;
    WgradPGroup g3;
    if (!dec_rows && m->allow_wg3 && M <= 4096 && fused_dx && m->early_wout && m->use_side2 && m->s_mode) {
        memset(&g3, 0, sizeof(g3));
        Linear* ls[3] = {&m->dec1[2], &m->dec1[1], &m->dec1[0]};
        const uint16_t* xs[3] = {ptr<uint16_t>(w.g2P), ptr<uint16_t>(w.g1P), ptr<uint16_t>(m->zP[0])};
        const uint16_t* gs[3] = {ptr<uint16_t>(w.dlP), ptr<uint16_t>(w.d2P), ptr<uint16_t>(w.d1P)};
        int ns[3], nw[3];
        group3 = true;
        for (int i = 0; i < 3; ++i) {
            CHK(wgradp_plan(m, *ls[i], xs[i], gs[i], M, g3.a[i], ns[i], nw[i]));
            group3 = group3 && nw[i] == 8;
            g3.gx[i] = (ls[i]->JT + 7) / 8; g3.gy[i] = (ls[i]->IT + 15) / 16;
            g3.zbeg[i + 1] = g3.zbeg[i] + ns[i];
        }
        g3.n = 3;
        g3.a[0].rowscale = ptr<float>(m->gx);      // (the main stream's row weights: this group waits for ev_fork2, i.e. for the main stream -- never the side stream's copy)
    }
    if (dec_rows) {
        if (!fused_dx) {      // (the dX chain as three launches; their weight gradients follow in wgrad_rows_kernel, further down the main stream)
            { ScopedTimer tm(m, T_DX_HID); CHK(dense_dx(m, m->dec1[1], ptr<uint16_t>(w.d2P), M, ptr<uint16_t>(w.g1P), ptr<uint16_t>(w.d1P), nullptr)); }
            { ScopedTimer tm(m, T_DX_LAT); CHK(dense_dx(m, m->dec1[0], ptr<uint16_t>(w.d1P), M, nullptr, nullptr, ptr<float>(w.dz))); }
        }
        m->tail = m->side;
    } else if (group3) {
        HIPCHK(hipStreamWaitEvent(m->side2, m->ev_fork2, 0));
        { ScopedTimer tm(m, T_WGRAD_OUT, m->side2); launch_wgradp_group(g3, m->side2); }
        HIPCHK(hipGetLastError());
        m->tail = m->side2;
    } else {
    if (m->early_wout && (m->lse_dup || m->lse_fused)) {}                            // forked behind the decoder kernel already (forward_impl)
    else if (m->early_wout) HIPCHK(hipStreamWaitEvent(m->side, m->ev_lse, 0));            // forked behind lse_kernel (forward_impl)
    else HIPCHK(hipStreamWaitEvent(m->side, fused_dx ? m->ev_fork2 : m->ev_fork, 0));  // the event rode on out_bwd's / dec_bwd's dispatch packet
    {   // (its completion event ev_s2 rides on the dispatch packet: the stream that later picks `side` up waits ~8 us less than behind a record)
        ScopedTimer tm(m, T_WGRAD_OUT, sd);
        const bool two_part = m->wout_split > 0 && m->s_mode && !m->g2w && m->early_wout && m->use_side2 && fused_dx && M >= 8192 && m->dec1[2].IT <= 14 && m->allow_wg7 && !(m->abl_skip & 1);
        if (two_part) {
            WgradPArgs a1, a2;
            int n1 = 1, n2 = 1;
            CHK(wgradp_two_plan(m, m->dec1[2], ptr<uint16_t>(w.g2P), ptr<uint16_t>(w.dlP), M, ptr<float>(m->lse_dup ? m->gx2 : m->gx), a1, n1, a2, n2));
            launch_wgradp(a1, n1, 7, sd);
            HIPCHK(hipStreamWaitEvent(sd, m->ev_fork2, 0));      // (the late part starts behind dec_bwd_kernel, beside the hidden layers' gradients)
            set_launch_stop_event(m->ev_s2);
            launch_wgradp(a2, n2, 7, sd);
            HIPCHK(hipGetLastError());
        } else
        if (m->abl_skip & 1) { if (m->early_wout && m->use_side2) HIPCHK(hipEventRecord(m->ev_s2, sd)); }
        else if (m->early_wout && m->use_side2) set_launch_stop_event(m->ev_s2);
        if (two_part) {} else
        if (m->abl_skip & 1) { WgradPArgs a0; int n0 = 1, w0 = 8; CHK(wgradp_plan(m, m->dec1[2], ptr<uint16_t>(w.g2P), ptr<uint16_t>(w.dlP), M, a0, n0, w0)); }      // (slabs allocated: the reduction still reads them)
        else if (m->g2w) CHK(wgradp(m, m->dec1[2], ptr<uint16_t>(w.g2wP), ptr<uint16_t>(w.dlP), M, sd, nullptr));      // (pre-weighted X operand: the unweighted kernel)
        else CHK(wgradp(m, m->dec1[2], ptr<uint16_t>(w.g2P), ptr<uint16_t>(w.dlP), M, sd, m->s_mode ? ptr<float>(m->lse_dup ? m->gx2 : m->gx) : nullptr));
    }
    if (!fused_dx) {
        { ScopedTimer tm(m, T_DX_HID); CHK(dense_dx(m, m->dec1[1], ptr<uint16_t>(w.d2P), M, ptr<uint16_t>(w.g1P), ptr<uint16_t>(w.d1P), nullptr)); }
        { ScopedTimer tm(m, T_DX_LAT); set_launch_stop_event(m->ev_fork2); CHK(dense_dx(m, m->dec1[0], ptr<uint16_t>(w.d1P), M, nullptr, nullptr, ptr<float>(w.dz))); }
    }
    // ONE event (ev_fork2) behind the whole dX chain -- the last of its kernels carries it on its dispatch packet (every separate
    // record costs the main stream a ~6 us bubble): the hidden layers' weight gradients need dpre2 and dpre1, and the deferred
    // decoder update further down the side stream must come after dX of d1, the last reader of the decoder's weight images.
    // Forked early, the side stream is busy with the output layer's gradient until after that: ONE wait then covers everything,
    // and out_bwd carries no event at all -- one bubble less on the main stream, one wait less on the side stream.
    // The hidden layers' weight gradients need dpre2 / dpre1 (ev_fork2), not the output layer's gradient: forked early, that one
    // keeps `side` busy well past the end of the dX chain, so they go to a second side stream and run beside it -- as ONE grouped
    // launch where both take the specialised-wave shape.  They finish last, so that stream (`tail`) also carries what follows the
    // weight gradients (the decoder's slab reduction [+ exchange] + Adam): it picks up `side` (ev_s2, recorded behind the output
    // layer's gradient, long complete by then) instead of `side` picking up the later of the two.
    hipStream_t ws = sd;
    m->tail = sd;
    if (m->early_wout && m->use_side2) {
        HIPCHK(hipStreamWaitEvent(m->side2, m->ev_fork2, 0));
        ws = m->side2;
        m->tail = m->side2;
    } else if (m->early_wout) HIPCHK(hipStreamWaitEvent(m->side, m->ev_fork2, 0));
    {
        WgradPArgs ah, al;
        int nsh = 1, nsl = 1, shh = 8, shl = 8;
        CHK(wgradp_plan(m, m->dec1[1], ptr<uint16_t>(w.g1P), ptr<uint16_t>(w.d2P), M, ah, nsh, shh));
        CHK(wgradp_plan(m, m->dec1[0], ptr<uint16_t>(m->zP[0]), ptr<uint16_t>(w.d1P), M, al, nsl, shl));
        if (shh == 7 && shl == 7 && m->allow_wg_group && (m->early_wout || fused_dx)) {      // (both inputs ready: one launch)
            WgradPGroup g;
            memset(&g, 0, sizeof(g));
            g.n = 2; g.a[0] = ah; g.a[1] = al;
            g.gx[0] = (m->dec1[1].JT + 15) / 16; g.gx[1] = (m->dec1[0].JT + 15) / 16; g.gy[0] = g.gy[1] = 1;
            g.zbeg[0] = 0; g.zbeg[1] = nsh; g.zbeg[2] = nsh + nsl;
            ScopedTimer tm(m, T_WGRAD_HID, ws);
            launch_wgradws_group(g, ws);
        } else {
            if (!(m->abl_skip & 2)) { ScopedTimer tm(m, T_WGRAD_HID, ws); launch_wgradp(ah, nsh, shh, ws); }
            if (!m->early_wout && !fused_dx) HIPCHK(hipStreamWaitEvent(m->side, m->ev_fork2, 0));
            if (!(m->abl_skip & 2)) { ScopedTimer tm(m, T_WGRAD_LAT, ws); launch_wgradp(al, nsl, shl, ws); }
        }
        HIPCHK(hipGetLastError());
    }
    if (m->defer_split && m->descs_dirty) CHK(build_descs(m));      // (early_first comes from the table)
    split_upd = m->defer_split && fused_lr >= 0.0f && m->allow_defer && m->early_first > 0 && !two && ws == m->side2 && m->early_wout;
    if (ws == m->side2 && !split_upd) HIPCHK(hipStreamWaitEvent(m->side2, m->ev_s2, 0));
    }      // (!group3)
    const bool fuse = fused_lr >= 0.0f;
    const float alpha = fuse ? adam_alpha(m, fused_lr) : 0.0f;
    if (m->descs_dirty) CHK(build_descs(m));
    const bool defer = fuse && m->allow_defer && m->early_first > 0 && !two && !dec_rows;      // (2-layer: the main stream needs the side-stream block gradients anyway)

    const float* dz1 = ptr<float>(w.dz);
    const float *dz1_b = nullptr, *dz1_c = nullptr;
    bool dz_sum_half = false;      // 2-layer model, fused block backward: the three terms of dz1 arrive summed, as bf16, in dzdir
    if (two) {
        // ---- p(z1|z2) head, dec2, q(z2|z1) head, enc2
        CHK(ensure(m->dzdir, (size_t)Mp * m->Dp[0] * 4, st));
        if (m->chain2_bwd) {
            // one launch per block: the head recomputed from h2, its gradient, the block's dX chain (gblock_bwd_kernel); the second one
            // also sums the three terms of dz1 (bf16, in dzdir) -- latent_bwd_kernel reads that like the 1-layer step's dz
            GBlockBwdArgs gb;
            memset(&gb, 0, sizeof(gb));
            gb.imgH = m->dec2[2].imgF; gb.imgBh = m->dec2[2].imgB; gb.imgB2 = m->dec2[1].imgB; gb.imgB1 = m->dec2[0].imgB;
            gb.H2 = ptr<uint16_t>(m->wdec2.h2P); gb.H1 = ptr<uint16_t>(m->wdec2.h1P); gb.gx = ptr<float>(m->gx);
            gb.M = M; gb.k = k; gb.B = B; gb.D = m->D[0]; gb.eps = eps_src(m, 0);
            gb.head1 = ptr<float>(m->wenc1.head); gb.ldH1 = 2 * m->Dp[0];
            gb.DHP = ptr<uint16_t>(m->wdec2.dheadP); gb.D2P = ptr<uint16_t>(m->wdec2.d2P); gb.D1P = ptr<uint16_t>(m->wdec2.d1P);
            gb.DZ2 = ptr<float>(m->wdec2.dx); gb.DZD = (uint16_t*)m->wenc2.dx.p;
            set_launch_stop_event(m->ev_blk);
            launch_gblock_bwd(0, gb, st);
            CHK(block_bwd(m, m->dec2, m->wdec2, ptr<uint16_t>(m->zP[1]), M, false, true, true));
            memset(&gb, 0, sizeof(gb));
            gb.imgH = m->enc2[2].imgF; gb.imgBh = m->enc2[2].imgB; gb.imgB2 = m->enc2[1].imgB; gb.imgB1 = m->enc2[0].imgB;
            gb.H2 = ptr<uint16_t>(m->wenc2.h2P); gb.H1 = ptr<uint16_t>(m->wenc2.h1P); gb.gx = ptr<float>(m->gx);
            gb.M = M; gb.k = k; gb.B = B; gb.D = m->D[1]; gb.eps = eps_src(m, 1);
            gb.DZIN = ptr<float>(m->wdec2.dx);
            gb.DHP = ptr<uint16_t>(m->wenc2.dheadP); gb.D2P = ptr<uint16_t>(m->wenc2.d2P); gb.D1P = ptr<uint16_t>(m->wenc2.d1P);
            gb.DZD = (uint16_t*)m->wenc2.dx.p; gb.DZDEC = ptr<float>(w.dz); gb.ldDZDEC = m->dec1[0].Kp32; gb.DZOUT = (uint16_t*)m->dzdir.p;
            set_launch_stop_event(m->ev_blk);
            launch_gblock_bwd(1, gb, st);
            // (the encode block's weight gradients on the second side stream, free by now: beside the decode block's, not behind them)
            CHK(block_bwd(m, m->enc2, m->wenc2, ptr<uint16_t>(m->zP[0]), M, false, true, true, (m->early_wout && m->use_side2) ? m->side2 : nullptr));
            HIPCHK(hipGetLastError());
            dz1 = nullptr; dz_sum_half = true;
        } else {
        GaussBwdArgs g;
        memset(&g, 0, sizeof(g));
        g.mode = 0; g.G = ptr<float>(m->gx);
        g.head = ptr<float>(m->wdec2.head); g.ldH = 2 * m->Dp[0]; g.D = m->D[0]; g.Dp = m->Dp[0];
        g.zhead = ptr<float>(m->wenc1.head); g.ldZH = 2 * m->Dp[0]; g.Dzp = m->Dp[0];
        g.dz_direct = ptr<float>(m->dzdir); g.ldDZ = m->Dp[0];
        g.eps = eps_src(m, 0); g.M = M; g.Mp = Mp; g.k = k;
        g.DHP = ptr<uint16_t>(m->wdec2.dheadP);
        launch_gauss_bwd(g, st);
        CHK(block_bwd(m, m->dec2, m->wdec2, ptr<uint16_t>(m->zP[1]), M, true, true));
        memset(&g, 0, sizeof(g));
        g.mode = 1; g.G = ptr<float>(m->gx);
        g.head = ptr<float>(m->wenc2.head); g.ldH = 2 * m->Dp[1]; g.D = m->D[1]; g.Dp = m->Dp[1];
        g.dz_in = ptr<float>(m->wdec2.dx); g.ldDZ = m->Dp[1];
        g.eps = eps_src(m, 1); g.M = M; g.Mp = Mp; g.k = k;
        g.DHP = ptr<uint16_t>(m->wenc2.dheadP);
        launch_gauss_bwd(g, st);
        CHK(block_bwd(m, m->enc2, m->wenc2, ptr<uint16_t>(m->zP[0]), M, true, true));
        dz1_b = ptr<float>(m->dzdir); dz1_c = ptr<float>(m->wenc2.dx);     // summed inside latent_bwd_kernel
        }
    }
    LatentBwdArgs lat_args;
    bool lat_fuse = false;
    {
        LatentBwdArgs a;
        memset(&a, 0, sizeof(a));
        a.dz = dz1; a.dz2 = dz1_b; a.dz3 = dz1_c; a.ldDZ = m->Dp[0];
        if (dz_half) a.dzh = (const uint16_t*)w.dz.p;
        if (dz_sum_half) a.dzh = (const uint16_t*)m->dzdir.p;
        a.head = ptr<float>(m->wenc1.head); a.ldH = 2 * m->Dp[0]; a.D = m->D[0]; a.Dp = m->Dp[0];
        a.cf = ptr<float4>(m->cf); a.eps = eps_src(m, 0);
        a.B = B; a.Bp = Bp; a.k = k;
        a.kmu = a.ksig = (objective == OBJ_VAE_ELBO_KL) ? m->beta / (float)B : 0.f;
        a.DHP = ptr<uint16_t>(m->wenc1.dheadP);
        if (m->has_prior) { a.prior_head = ptr<float>(m->wprior.head); a.DHP2 = ptr<uint16_t>(m->wprior.dheadP); }
        // Few images and samples (round 4): these per-image sums are made inside the encoder's block_bwd_kernel (a wave per image in front of
        // its dX chain) -- one dependent launch less; the separate kernel (256 threads per image) stays for many samples per image
        lat_args = a;
        // (measured, end-to-end us per step with / without: B = 20, k = 1: 67.7 / 70.0; B = 20, k = 5: 67.5 / 69.3; B = 100, k = 5: 72.1 / 73.4; B = 20, k = 50: 88.2 / 78.5 --
        // one wave walking 50 samples is slower than latent_bwd_kernel's 256 threads: up to 16 samples per image)
        // (round 5, option lat_rows4: beyond 16 samples per image block_bwd_kernel<4> -- 4 images per workgroup, an image's samples over four waves --
        // can take the sums on 4 x the workgroups; measured no faster than the two launches in the step, see allow_lat_rows4)
        lat_fuse = m->allow_lat_in_block && !m->has_prior && B <= 1024 && (k <= 16 || (m->allow_lat_rows4 && B >= 64)) && m->allow_block_fused;
        if (!lat_fuse && !(m->abl_skip & 8)) { ScopedTimer tm(m, T_LATENT_BWD); launch_latent_bwd(a, st); }
    }
    if (m->has_prior) CHK(block_bwd(m, m->prior, m->wprior, ptr<uint16_t>(m->condP), B, false, false));
    // Round 4: on few rows the image encoder's weight gradients, their sum over ALL rows and (fused step) the Adam update are one launch
    // (wgrad_rows_kernel) -- the encoder's layers (the head of the table) then need no slabs and no share of reduce_grads_kernel
    const bool rows_enc = wgrad_rows_ok(m, m->enc1, B);
    if (lat_fuse) {
        bool taken = false;
        CHK(block_bwd(m, m->enc1, m->wenc1, ptr<uint16_t>(m->xP), B, false, false, false, nullptr, rows_enc, &lat_args, &taken));
        if (!taken) {      // (shapes block_bwd_kernel does not cover: the separate kernel after all, then the block's backward pass)
            { ScopedTimer tm(m, T_LATENT_BWD); launch_latent_bwd(lat_args, st); }
            CHK(block_bwd(m, m->enc1, m->wenc1, ptr<uint16_t>(m->xP), B, false, false, false, nullptr, rows_enc));
        }
    } else
    CHK(block_bwd(m, m->enc1, m->wenc1, ptr<uint16_t>(m->xP), B, false, false, false, nullptr, rows_enc));
    if (m->descs_dirty) CHK(build_descs(m));      // (the encoder's splits were planned after the first build)
    const int enc_end = m->enc1[2].sub[m->enc1[2].nsub - 1] + 1;      // first table entry behind the image encoder's layers
    const int rb_lo = !rows_enc ? 0 : enc_end < (int)m->descs.size() ? m->descs[enc_end].rblock_begin : m->reduce_blocks;
    // split (data-parallel step, iwae_forward_backward_split): the decoder's layers are summed into the flat gradient on the
    // side stream, right behind their weight gradients, and NOT joined here -- the caller's all-reduce of that segment is
    // ordered behind the side stream and runs beside the encoder's backward pass; join_side() (every later entry point) joins.
    // Without split (iwae_forward_backward: gradient only, e.g. the one-message data-parallel step) the same early decoder
    // reduction runs on the side stream and the main stream joins it behind its own, shorter, encoder reduction.
    const bool early = !fuse && m->early_first > 0 && !two && !dec_rows;
    // 2-layer training step at large row counts (round 3): every layer behind the image encoder has its weight gradients on the side
    // streams; their slab sums + Adam follow there (one launch on `tail`, which picks `side` up), instead of the main stream waiting for
    // both side streams and then summing all 94 MB itself.  The next forward joins in front of z1 (join_side).  The image rewrite is safe
    // for the same reason as in the 1-layer step: the side streams' weight gradients wait for the events behind the dX chains
    // (ev_fork2, ev_blk), the last readers of those images.
    const bool defer2 = fuse && two && m->allow_defer && m->allow_defer2 && m->early_first2 > 0 && m->chain2_bwd && m->early_wout && m->use_side2 && !split;
    m->split_offset = m->nparam;
    m->early_held = false;
    if (early) {
        if (hold_early && split) m->early_held = true;
        else {
        set_launch_stop_event(m->ev_dec);
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), m->early_first, m->reduce_blocks - m->early_first, m->grad, m->param, m->mom,
                            m->vel, 0.0f, m->adam_b1, m->adam_b2, m->adam_eps, 0, nullptr, 0, 0.f, nullptr, m->tail);
        }
        m->dec_pending = true;
        if (split) m->split_offset = m->descs[m->dec1[0].sub[0]].offW;
    } else if (defer2) {       // (the side streams' layers are summed and updated there, further down: nothing to join)
    } else if (dec_rows) {     // (nothing ran on a side stream)
    } else if (!defer) {       // join: every weight gradient launched on the side stream is in its slabs
        HIPCHK(hipEventRecord(m->ev_join, m->tail));
        HIPCHK(hipStreamWaitEvent(st, m->ev_join, 0));
        if (two && m->tail != m->side) {      // the per-sample blocks' weight gradients went to `side` behind the output layer's: both side streams join
            HIPCHK(hipEventRecord(m->ev_join2, m->side));
            HIPCHK(hipStreamWaitEvent(st, m->ev_join2, 0));
        }
    }
    {
        // the main stream's share of the table: [rb_lo, rb_hi) -- empty when wgrad_rows_kernel took the encoder and everything else is
        // deferred to the side streams (the full-size 1- and 2-layer steps) or rode in the same launch (few data rows: dec_rows); that
        // kernel's extra block makes the batch means whenever it runs
        const int rb_hi = (defer || early) ? m->early_first : defer2 ? m->early_first2 : m->reduce_blocks;
        const int rb_d0 = dec_rows ? m->descs[m->dec1[0].sub[0]].rblock_begin : rb_hi;        // (dec_rows: the decoder's three layers drop out of the range)
        const int d_end = m->dec1[2].sub[0] + 1;
        const int rb_d1 = !dec_rows ? rb_hi : d_end < (int)m->descs.size() ? m->descs[d_end].rblock_begin : m->reduce_blocks;
        ScopedTimer tm_red(m, T_REDUCE);
        if (rows_enc) CHK(block_wgrad_rows(m, m->enc1, m->wenc1, ptr<uint16_t>(m->xP), B, alpha, fuse, true, dec_rows));
        const int n1 = std::max(0, std::min(rb_d0, rb_hi) - rb_lo), n2 = std::max(0, rb_hi - rb_d1);
        if (!rows_enc || n1 + n2 > 0)
            launch_reduce_grads(m->d_descs, (int)m->descs.size(), rb_lo, n1, m->grad, m->param, m->mom, m->vel,
                                alpha, m->adam_b1, m->adam_b2, m->adam_eps, fuse ? 1 : 0, rows_enc ? nullptr : ptr<float>(m->per_b), B, two ? 1.f : m->beta, m->d_scalars, st,
                                rb_d1, n2);
    }
    if (early && !split) CHK(join_side(m));
    if (defer2 && m->tail == m->side2 && m->tail != m->side && m->allow_defer2_split && m->dec2[0].nsub == 1 && m->dec1[0].nsub == 1 && m->dec1[2].nsub == 1) {
        // each side stream sums and updates the layers whose weight gradients IT carried, as soon as its own chain ends: the second one the
        // encode block q(z2|z1) and the decoder's two tanh layers, the first one the decode block p(z1|z2) and the output layer (two block
        // ranges per launch: table order enc2 | dec2 | dec1).  The next forward waits for both events (join_side).
        const int b_dec2 = m->descs[m->dec2[0].sub[0]].rblock_begin, b_dec1 = m->descs[m->dec1[0].sub[0]].rblock_begin, b_out = m->descs[m->dec1[2].sub[0]].rblock_begin;
        set_launch_stop_event(m->ev_dec);
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), m->early_first2, b_dec2 - m->early_first2, m->grad, m->param, m->mom, m->vel, alpha, m->adam_b1,
                            m->adam_b2, m->adam_eps, 1, nullptr, 0, 0.f, nullptr, m->side2, b_dec1, b_out - b_dec1);
        set_launch_stop_event(m->ev_dec2);
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), b_dec2, b_dec1 - b_dec2, m->grad, m->param, m->mom, m->vel, alpha, m->adam_b1, m->adam_b2,
                            m->adam_eps, 1, nullptr, 0, 0.f, nullptr, m->side, b_out, m->reduce_blocks - b_out);
        m->dec_pending = true; m->dec2_pending = true;
    } else if (defer2) {
        if (m->tail != m->side) {
            HIPCHK(hipEventRecord(m->ev_join2, m->side));
            HIPCHK(hipStreamWaitEvent(m->tail, m->ev_join2, 0));
        }
        set_launch_stop_event(m->ev_dec);
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), m->early_first2, m->reduce_blocks - m->early_first2, m->grad, m->param, m->mom,
                            m->vel, alpha, m->adam_b1, m->adam_b2, m->adam_eps, 1, nullptr, 0, 0.f, nullptr, m->tail);
        m->dec_pending = true;
    }
    if (defer && split_upd && m->early_first > 0) {
        // Round 5: one deferred update per side stream, each behind the weight gradients it carried -- no hand-off between the two side
        // streams in front of the update, and the output layer's share (54 % of the decoder's slabs) is done ~20 us before the hidden layers'
        // gradients end.  The output layer's gradient forked behind the decoder FORWARD: its update rewrites the W3 image dec_bwd_kernel
        // reads, so `side` waits for that kernel's event first (long complete by then).
        const int b_out = m->descs[m->dec1[2].sub[0]].rblock_begin;
        HIPCHK(hipStreamWaitEvent(m->side, m->ev_fork2, 0));
        set_launch_stop_event(m->ev_dec2);
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), b_out, m->reduce_blocks - b_out, m->grad, m->param, m->mom, m->vel, alpha, m->adam_b1,
                            m->adam_b2, m->adam_eps, 1, nullptr, 0, 0.f, nullptr, m->side);
        set_launch_stop_event(m->ev_dec);
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), m->early_first, b_out - m->early_first, m->grad, m->param, m->mom, m->vel, alpha, m->adam_b1,
                            m->adam_b2, m->adam_eps, 1, nullptr, 0, 0.f, nullptr, m->side2);
        m->dec_pending = true; m->dec2_pending = true;
    } else if (defer) {
        // The decoder's layers (90 % of the slab bytes): slab sums + Adam on the side stream, behind its weight gradients
        // (which wait for ev_fork2, i.e. for dX of d1, the last reader of the decoder's weight images -- without that order
        // the trajectory test caught a stale-image race), joined by the next user of the decoder (join_side): it runs beside
        // the encoder's backward pass / update and the next step's encoder forward.
        if (m->abl_skip & 4) HIPCHK(hipEventRecord(m->ev_dec, m->tail));
        else {
        set_launch_stop_event(m->ev_dec);
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), m->early_first, m->reduce_blocks - m->early_first, m->grad, m->param, m->mom,
                            m->vel, alpha, m->adam_b1, m->adam_b2, m->adam_eps, 1, nullptr, 0, 0.f, nullptr, m->tail);
        }
        m->dec_pending = true;
    }
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

int fetch_outputs(iwae_model* m, iwae_scalars* scalars, const iwae_tensors* want) {
    hipStream_t st = m->stream;
    const int B = m->B, k = m->k, M = m->M;
    const bool two = m->cfg.n_layers == 2;
    if (want) {
        const float* rowsrc[6] = {ptr<float>(m->rows[0]), ptr<float>(m->rows[1]), ptr<float>(m->rows[2]),
                                  ptr<float>(m->rows[3]), ptr<float>(m->rows[4]), ptr<float>(m->logw)};
        float* dsts[7];
        const float* srcs[7];
        int n = 0;
        if (want->lpxz) { dsts[n] = want->lpxz; srcs[n++] = rowsrc[0]; }
        if (want->lpz) { dsts[n] = want->lpz; srcs[n++] = rowsrc[1]; }
        if (want->lqzx) { dsts[n] = want->lqzx; srcs[n++] = two ? rowsrc[3] : rowsrc[2]; }
        if (want->lpz2 && two) { dsts[n] = want->lpz2; srcs[n++] = rowsrc[2]; }
        if (want->lqzx2 && two) { dsts[n] = want->lqzx2; srcs[n++] = rowsrc[4]; }
        if (want->log_w) { dsts[n] = want->log_w; srcs[n++] = rowsrc[5]; }
        if (want->al) { dsts[n] = want->al; srcs[n++] = ptr<float>(m->wn); }
        const size_t zmax = (size_t)M * std::max(m->D[0], two ? m->D[1] : 0);
        CHK(ensure(m->scratch, std::max((size_t)M * 4, zmax * 4 + (size_t)B * 256 * 4), st));
        for (int i = 0; i < n; ++i) {
            launch_export_rows(srcs[i], B, k, ptr<float>(m->scratch), st);
            CHK(copy_out(m, dsts[i], m->scratch.p, (size_t)M * 4));
        }
        for (int layer = 0; layer < (two ? 2 : 1); ++layer) {
            float* zdst = layer == 0 ? want->z : want->z2;
            float* sdst = layer == 0 ? want->snis_z : want->snis_z2;
            if (!zdst && !sdst) continue;
            SampleArgs s;
            memset(&s, 0, sizeof(s));
            BlockWs& hw = layer == 0 ? m->wenc1 : m->wenc2;
            s.head = ptr<float>(hw.head); s.ldH = 2 * m->Dp[layer]; s.Dp = m->Dp[layer]; s.D = m->D[layer];
            s.head_per_row = layer; s.M = M; s.Mp = m->Mp; s.k = k; s.B = B; s.eps = eps_src(m, layer);
            float* zdev = ptr<float>(m->scratch);
            launch_export_z(s, zdev, st);
            if (zdst) CHK(copy_out(m, zdst, zdev, (size_t)M * m->D[layer] * 4));
            if (sdst) {
                float* sdev = zdev + zmax;
                launch_snis(zdev, ptr<float>(m->wn), B, k, m->D[layer], sdev, st);
                CHK(copy_out(m, sdst, sdev, (size_t)B * m->D[layer] * 4));
            }
        }
        HIPCHK(hipGetLastError());
    }
    if (scalars) {
        HIPCHK(hipMemcpyAsync(m->h_scalars, m->d_scalars, SC_COUNT * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const float* s = m->h_scalars;
        memset(scalars, 0, sizeof(*scalars));
        scalars->vae_elbo = s[SC_VAE_ELBO];
        scalars->vae_elbo_kl = s[SC_VAE_ELBO_KL];
        scalars->iwae_elbo = s[SC_IWAE_ELBO];
        scalars->iwae_eq14 = s[SC_IWAE_EQ14];
        scalars->inference_loss = s[SC_INFERENCE_LOSS];
        scalars->mean_lpxz = s[SC_MEAN_LPXZ];
        scalars->mean_lpz = s[SC_MEAN_T1];
        scalars->mean_lqzx = s[SC_MEAN_T2];
        scalars->mean_kl = s[SC_KL];
    } else if (want) {
        HIPCHK(hipStreamSynchronize(st));
    }
    return IWAE_OK;
}

int check_objective(iwae_model* m, int objective) {
    if (objective < 0 || objective > 4) return fail(IWAE_ERR_ARG, "unknown objective");
    if (m->cfg.n_layers == 2 && (objective == OBJ_VAE_ELBO_KL || objective == OBJ_DREG))
        return fail(IWAE_ERR_ARG, "objective not defined for the 2-layer model (KeyError in src/iwae2.py:154-173)");
    if (m->C > 0 && objective == OBJ_DREG) return fail(IWAE_ERR_ARG, "the DReG estimator is defined for the unconditional 1-layer model (tasks/task02.py)");
    return IWAE_OK;
}

int adam_impl(iwae_model* m, float lr, float gscale) {
    if (m->descs_dirty) CHK(build_descs(m));
    const float alpha = adam_alpha(m, lr);
    launch_adam(m->d_descs, (int)m->descs.size(), m->elem_blocks, m->param, m->grad, m->mom, m->vel, alpha, gscale, m->adam_b1, m->adam_b2, m->adam_eps, 1, m->stream);
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

// =================================================================== float32 mode
// The reference's own arithmetic: Keras Dense layers in float32 (src/iwae1.py:31-34,72-75).  Same step structure as the bf16
// path with plain row-major float32 tensors and one generic MFMA GEMM (fp32_kernels.hip); the per-sample kernels that already
// work in float32 (sampling + densities, lse_kernel, latent_bwd_kernel, gauss_*_kernel, Adam) are shared.
int f32_gemm(iwae_model* m, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M, int N, int K,
             const float* bias, int epi, const float* ACT, long ldact, bool accumulate, const float* brow_scale = nullptr, const float* orow_scale = nullptr,
             hipStream_t st = nullptr) {
    GemmF32Args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.sam = sam; a.sak = sak; a.B = B; a.sbk = sbk; a.sbn = sbn; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.bias = bias; a.epi = epi; a.ACT = ACT; a.ldact = ldact; a.accumulate = accumulate ? 1 : 0; a.kchunk = K; a.slab_stride = 0;
    a.brow_scale = brow_scale; a.orow_scale = orow_scale;
    {   // few rows (the encoder on the batch's images): K split + one reduction pass that carries the epilogue
        hipStream_t s_ = st ? st : m->stream;
        // (not inside iwae_eval_llh: the split depends on how many images a launch holds, and an image's estimate must not -- the evaluator's
        // encoder is 419 rows beside 2^21 decoder rows, nothing to gain there: test_eval_llh_images_per_launch_are_invisible)
        const int ns = m->in_eval_llh ? 1 : gemm_f32_fewrows_split(M, N, K);
        if (ns > 1 && !brow_scale) {
            a.avec = a.bvec = 0;
            CHK(ensure(m->f32.kslab, (size_t)ns * M * N * 4, s_));
            launch_gemm_f32_fewrows(a, ptr<float>(m->f32.kslab), s_);
            HIPCHK(hipGetLastError());
            return IWAE_OK;
        }
    }
#ifdef IWAE_DENSE_STAMPS
    if (m->dstamp_epi == 12 && orow_scale && M >= 4096) {      // diagnostic (STAMPS=1 build, option dense_stamps_epi = 12): phase stamps of the output layer's dX product
        m->dstamp_waves = ((M + 63) / 64) * ((N + 223) / 224) * 4;
        CHK(ensure(m->dstamps, (size_t)m->dstamp_waves * 64, st ? st : m->stream));
        a.stamps = ptr<unsigned long long>(m->dstamps);
    }
#endif
    launch_gemm_f32(a, 1, st ? st : m->stream);
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}
// Y = epi(X W + b), W = the Keras kernel [in, out] of layer kl inside the flat float32 parameters
int f32_fwd(iwae_model* m, const KerasLayer& kl, const float* X, long ldx, int rows, float* Y, long ldy, int epi) {
    return f32_gemm(m, X, ldx, 1, m->param + kl.offW, kl.Nout, 1, Y, ldy, rows, kl.Nout, kl.Kin, m->param + kl.offb, epi, nullptr, 0, false);
}
// DX (+)= (G W^T) * (1 - ACT^2)   (ACT = the stored tanh output of the layer below, or null)
// (rowscale: row r of G counts with weight rowscale[r] -- applied to the product's rows, in front of the tanh' factor)
int f32_dx(iwae_model* m, const KerasLayer& kl, const float* G, long ldg, int rows, float* DX, long lddx, const float* ACT, long ldact, bool accumulate,
           const float* rowscale = nullptr) {
    return f32_gemm(m, G, ldg, 1, m->param + kl.offW, 1, kl.Nout, DX, lddx, rows, kl.Kin, kl.Nout, nullptr, ACT ? GEMM_EPI_DTANH : GEMM_EPI_NONE, ACT, ldact, accumulate,
                    nullptr, rowscale);
}
// the queued slab sums of this step's float32 weight gradients, one launch per segment (f32_dw): seg 0 = the gradients made on the main stream,
// seg 1 = the decoder's, made on the side stream (backward_f32); seg < 0: whatever is queued, each segment on its own stream
int f32_flush_reductions(iwae_model* m, int seg = -1) {
    for (int sg = 0; sg < 2; ++sg) {
        if (seg >= 0 && sg != seg) continue;
        ReduceSlabsJobs jobs;
        memset(&jobs, 0, sizeof(jobs));
        for (const auto& p : m->f32_pending) {
            if (p.seg != sg) continue;
            ReduceSlabsJob& j = jobs.job[jobs.n++];
            j.slabs = ptr<float>(m->f32.slab) + p.off; j.stride = p.stride; j.n = p.n; j.out = p.out; j.nsplit = p.nsplit;
        }
        if (jobs.n > 0) {
            launch_reduce_slabs_multi_f32(jobs, sg == 1 ? m->side : m->stream);
            HIPCHK(hipGetLastError());
        }
    }
    if (seg < 0) m->f32_pending.clear();
    else m->f32_pending.erase(std::remove_if(m->f32_pending.begin(), m->f32_pending.end(), [seg](const iwae_model::F32Pending& p) { return p.seg == seg; }), m->f32_pending.end());
    if (m->f32_pending.empty()) m->f32_slab_used = 0;
    return IWAE_OK;
}
// grad W = X^T G, grad b = column sums of G: the row axis is split into fp32 slabs summed in a fixed order (deterministic)
// (rowscale: G's row r is multiplied by rowscale[r] as it is fetched -- the values the separate g_r s pass used to store)
int f32_dw(iwae_model* m, const KerasLayer& kl, const float* X, long ldx, const float* G, long ldg, int rows, const float* rowscale = nullptr, int seg = 0, int tile_mode = 0) {
    hipStream_t st = seg == 1 ? m->side : m->stream;
    // row splits: enough workgroups to fill the machine (~1 000 tiles of 64 x 64 or 128 x 128), at least 64 rows per split
    const int tiles = (int)gemm_f32_tiles(kl.Kin + 1, kl.Nout, tile_mode);      // (+ 1: the row of ones whose product row is the bias gradient)
    const int slots = std::min(m->f32_dw_tiles, gemm_f32_slots(kl.Kin + 1, kl.Nout, tile_mode));
    int nsplit = std::max(1, std::min(std::min(256, rows / m->f32_dw_min_rows), slots / tiles));      // (rounded DOWN: 1 027 workgroups on 1 024 slots are a second round of 3)
    while (nsplit > 8 && (tiles * nsplit) % 8 != 0) --nsplit;      // (a multiple of 8 workgroups: gemm_f32_v2_kernel then keeps a split's tiles on one XCD)
    const size_t nW = (size_t)kl.Kin * kl.Nout;
    // (round 5: the slabs of every gradient of the step stay until ONE reduction launch at the end of the backward pass; the buffer is sized for a
    // whole step -- a step that outgrows it falls back to the reduction per tensor, and the buffer grows for the next step)
    const size_t need = ((size_t)nsplit * (nW + kl.Nout) + 3) & ~(size_t)3;      // (a multiple of 4 floats: the next gradient's slabs stay 16-byte aligned)
    const bool queue = m->allow_f32_multi_reduce && nsplit > 1 && (m->f32_slab_used + need) * 4 <= m->f32.slab.cap && m->f32_pending.size() + 2 <= REDUCE_SLABS_MAX_JOBS;
    if (!queue) {
        if (!m->f32_pending.empty()) CHK(f32_flush_reductions(m));      // (queued jobs still read the buffer ensure() may replace)
        if (m->f32_side_active) { HIPCHK(hipStreamSynchronize(m->side)); HIPCHK(hipStreamSynchronize(m->stream)); }      // (first steps only: the buffer is still growing)
        CHK(ensure(m->f32.slab, std::max(need, m->f32_slab_want) * 4, st));
    }
    m->f32_slab_want_step += need;
    float* slabW = ptr<float>(m->f32.slab) + (queue ? m->f32_slab_used : 0);
    float* slabB = slabW + (size_t)nsplit * nW;
    GemmF32Args a;
    memset(&a, 0, sizeof(a));
    a.A = X; a.sam = 1; a.sak = ldx; a.B = G; a.sbk = ldg; a.sbn = 1; a.M = kl.Kin; a.N = kl.Nout; a.K = rows;
    a.brow_scale = rowscale; a.tile_mode = tile_mode;
    a.kchunk = (rows + nsplit - 1) / nsplit; a.kchunk = (a.kchunk + 15) / 16 * 16;
    const int ns = (rows + a.kchunk - 1) / a.kchunk;
    // the bias gradient = the column sums of (weighted) G = the product row of a row of ONES appended to X^T (GemmF32Args.Cones): no pass of its own
    if (ns == 1) {
        a.C = m->grad + kl.offW; a.ldc = kl.Nout; a.slab_stride = 0; a.Cones = m->grad + kl.offb; a.cones_stride = 0;
        launch_gemm_f32(a, 1, st);
    } else {
        a.C = slabW; a.ldc = kl.Nout; a.slab_stride = nW; a.Cones = slabB; a.cones_stride = (size_t)kl.Nout;
#ifdef IWAE_DENSE_STAMPS
        if (m->dstamp_epi == 13 && rowscale) {      // diagnostic (STAMPS=1 build, option dense_stamps_epi = 13): phase stamps of the output layer's weight gradient
            m->dstamp_waves = ((kl.Kin + 1 + 223) / 224) * ((kl.Nout + 63) / 64) * ns * 4;
            CHK(ensure(m->dstamps, (size_t)m->dstamp_waves * 64, st));
            a.stamps = ptr<unsigned long long>(m->dstamps);
        }
#endif
        launch_gemm_f32(a, ns, st);
        if (queue) {
            m->f32_pending.push_back({(size_t)(slabW - ptr<float>(m->f32.slab)), nW, nW, m->grad + kl.offW, ns, seg});
            m->f32_pending.push_back({(size_t)(slabB - ptr<float>(m->f32.slab)), (size_t)kl.Nout, (size_t)kl.Nout, m->grad + kl.offb, ns, seg});
            m->f32_slab_used += need;
        } else {
            launch_reduce_slabs_f32(slabW, nW, ns, nW, m->grad + kl.offW, st);
            launch_reduce_slabs_f32(slabB, kl.Nout, ns, kl.Nout, m->grad + kl.offb, st);
            if (m->f32_side_active) { HIPCHK(hipStreamSynchronize(st)); }      // (the next unqueued gradient reuses the buffer's front from another stream)
        }
    }
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}
// BasicBlock (iwae1.py:36-44) on R rows: X [R][ldx] -> h1, h2 [R][H], head [R][2Dp] (mu at 0.., sigma = exp(.)+1e-6 at Dp..)
int f32_block_fwd(iwae_model* m, int base, iwae_model::F32Block& w, const float* X, long ldx, int R, float* head, int Dp, bool bwd) {
    const KerasLayer *l1 = &m->klayers[base], *l2 = l1 + 1, *lmu = l1 + 2, *lsd = l1 + 3;
    const int H = l1->Nout, D = lmu->Nout;
    CHK(ensure(w.h1, (size_t)R * H * 4, m->stream));
    CHK(ensure(w.h2, (size_t)R * H * 4, m->stream));
    CHK(f32_fwd(m, *l1, X, ldx, R, ptr<float>(w.h1), H, GEMM_EPI_TANH));
    CHK(f32_fwd(m, *l2, ptr<float>(w.h1), H, R, ptr<float>(w.h2), H, GEMM_EPI_TANH));
    CHK(f32_fwd(m, *lmu, ptr<float>(w.h2), H, R, head, 2 * Dp, GEMM_EPI_NONE));
    CHK(f32_fwd(m, *lsd, ptr<float>(w.h2), H, R, head + Dp, 2 * Dp, GEMM_EPI_EXP));
    (void)D; (void)bwd;
    return IWAE_OK;
}
// backward of a BasicBlock from dhead [R][2Dp] (d mu | d pre-exp): all four weight gradients, optionally dX [R][lddx]
int f32_block_bwd(iwae_model* m, int base, iwae_model::F32Block& w, const float* X, long ldx, int R, int Dp, float* dX, long lddx) {
    const KerasLayer *l1 = &m->klayers[base], *l2 = l1 + 1, *lmu = l1 + 2, *lsd = l1 + 3;
    const int H = l1->Nout;
    const float* dh = ptr<float>(w.dhead);
    CHK(ensure(w.d2, (size_t)R * H * 4, m->stream));
    CHK(ensure(w.d1, (size_t)R * H * 4, m->stream));
    CHK(f32_dw(m, *lmu, ptr<float>(w.h2), H, dh, 2 * Dp, R));
    CHK(f32_dw(m, *lsd, ptr<float>(w.h2), H, dh + Dp, 2 * Dp, R));
    CHK(f32_dx(m, *lmu, dh, 2 * Dp, R, ptr<float>(w.d2), H, ptr<float>(w.h2), H, false));
    CHK(f32_dx(m, *lsd, dh + Dp, 2 * Dp, R, ptr<float>(w.d2), H, ptr<float>(w.h2), H, true));
    CHK(f32_dw(m, *l2, ptr<float>(w.h1), H, ptr<float>(w.d2), H, R));
    CHK(f32_dx(m, *l2, ptr<float>(w.d2), H, R, ptr<float>(w.d1), H, ptr<float>(w.h1), H, false));
    CHK(f32_dw(m, *l1, X, ldx, ptr<float>(w.d1), H, R));
    if (dX) CHK(f32_dx(m, *l1, ptr<float>(w.d1), H, R, dX, lddx, nullptr, 0, false));
    return IWAE_OK;
}

int forward_f32(iwae_model* m, const float* x, int B, int k, float beta, const float* eps, int objective, bool bwd, const iwae_tensors* want) {
    const bool from_ds = m->ds_start >= 0;
    if ((!x && !from_ds) || B <= 0 || k <= 0) return fail(IWAE_ERR_ARG, "forward: need x, B > 0, k > 0");
    if ((int64_t)B * k > (int64_t)1 << 30) return fail(IWAE_ERR_ARG, "forward: B*k too large");
    const float* cond = nullptr;      // conditional models (tasks/task05.py, tasks/task04.py): y of these images
    if (m->C > 0) {
        if (from_ds) {      // (x, y) from the resident set: the input kernel writes onehot(y) of the batch's images into m->cond (tasks/task05.py:296-322)
            if (!m->ds_has_labels) return fail(IWAE_ERR_STATE, "conditional model on the resident dataset: call iwae_dataset_set_labels first");
            CHK(ensure(m->cond, (size_t)B * m->C * 4, m->stream));
            m->cond_n = B; m->cond_row0 = 0;
        }
        if (m->cond_row0 + B > m->cond_n) return fail(IWAE_ERR_STATE, "conditional model: call iwae_set_condition with y for these images first");
        cond = ptr<float>(m->cond) + (size_t)m->cond_row0 * m->C;
    }
    const bool two = m->cfg.n_layers == 2;
    m->B = B; m->k = k; m->M = B * k; m->beta = beta;
    m->Mp = round_up(m->M, 128); m->Bp = round_up(B, 128);
    m->time_this = false;
    const int M = m->M, Mp = m->Mp, Bp = m->Bp, X = m->X;
    hipStream_t st = m->stream;
    if (m->bf16_side_used) {      // a bf16 call's deferred update / speculative draw may sit on either side stream
        CHK(join_side(m));
        if (m->side) HIPCHK(hipStreamSynchronize(m->side));
        if (m->side2) HIPCHK(hipStreamSynchronize(m->side2));
        m->bf16_side_used = false;
    }
    // (a float32 step's own deferred decoder update is joined in front of the decoder forward: the encoder and the sampling run beside it)
    m->user_eps = eps != nullptr;
    m->epsc_ptr[0] = m->epsc_ptr[1] = nullptr;
    if (!eps) {       // the step's draws, kept for the backward pass and the 2-layer densities (same generator as the bf16 path)
        const int np = (m->epsc_par + 1) % 3;
        CHK(draw_eps(m, np, m->noise_step, M, st));
        if (m->eval_k_total > 0) m->eps_tag[np].valid = false;      // a k-chunk's draws: the tag (step, offset, rows) does not describe them
        if (bwd) m->epsc_par = np;      // (forward-only calls reuse one slot, as in forward_impl)
        for (int l = 0; l < m->cfg.n_layers; ++l) m->epsc_ptr[l] = ptr<float>(m->epsc[np][l]);
    } else {
        CHK(copy_in(m, m->epsbuf, eps, (size_t)M * (m->D[0] + (two ? m->D[1] : 0)) * 4));
    }
    const float* xd = x;
    if (from_ds) {      // main.py:117-120 on the device: gather + binarise, keeping the float32 copy of the batch
        CHK(ensure(m->xP, (size_t)Bp * m->Xinp * 2, st));
        CHK(ensure(m->xin, (size_t)B * X * 4, st));
        launch_gather_binarize(ptr<uint8_t>(m->ds_data), ptr<int32_t>(m->ds_order), m->ds_start, m->ds_N, B, X, m->Xinp, Bp, m->cfg.seed,
                               m->ds_epoch, ptr<uint16_t>(m->xP), ptr<float>(m->xin), st, m->C > 0 ? ptr<uint8_t>(m->ds_labels) : nullptr, m->C, m->C > 0 ? ptr<float>(m->cond) : nullptr);
        m->ds_start = -1;
        xd = ptr<float>(m->xin);
    } else if (!is_device_ptr(x, m->cfg.device)) { CHK(copy_in(m, m->xin, x, (size_t)B * X * 4)); xd = ptr<float>(m->xin); }
    m->f32_x = xd;
    // ---- encoder on the images (conditional models: on concat(x, y), tasks/task05.py:113-118)
    const int b_enc1 = m->enc1[0].sub[0];
    CHK(ensure(m->wenc1.head, (size_t)Bp * 2 * m->Dp[0] * 4, st));
    const float* xenc = xd;
    if (m->C > 0) {
        CHK(ensure(m->f32.xcat, (size_t)B * (X + m->C) * 4, st));
        launch_concat_f32(xd, X, cond, m->C, B, ptr<float>(m->f32.xcat), st);
        xenc = ptr<float>(m->f32.xcat);
    }
    CHK(f32_block_fwd(m, b_enc1, m->f32.enc1, xenc, X + m->C, B, ptr<float>(m->wenc1.head), m->Dp[0], bwd));
    if (m->has_prior) {     // p(z|y) = N(mu_p(y), sigma_p(y)): the prior block on the B condition rows (tasks/task04.py:108,124)
        CHK(ensure(m->wprior.head, (size_t)Bp * 2 * m->Dp[0] * 4, st));
        CHK(f32_block_fwd(m, m->prior[0].sub[0], m->f32.prior, cond, m->C, B, ptr<float>(m->wprior.head), m->Dp[0], bwd));
    }
    for (int i = 0; i < 6; ++i) CHK(ensure(m->rows[i], (size_t)Mp * 4, st));
    float* lpxz = ptr<float>(m->rows[0]);
    float* t1 = ptr<float>(m->rows[1]);
    float* t2 = ptr<float>(m->rows[2]);
    float* t3 = ptr<float>(m->rows[3]);
    float* t4 = ptr<float>(m->rows[4]);
    float* lqd = ptr<float>(m->rows[5]);
    // ---- z (z1) = mu + sigma*eps and its densities (iwae1.py:59,107,109)
    const int Dz = m->D[0] + m->C;      // row width of the decoder's input: z, or concat(z, y) (tasks/task05.py:185)
    if (m->f32_z_pending) {      // the previous step's gradient of the decoder's first layer reads z on the side stream
        HIPCHK(hipStreamWaitEvent(st, m->ev_join, 0));
        m->f32_z_pending = false;
    }
    CHK(ensure(m->f32.z[0], (size_t)Mp * Dz * 4, st));
    {
        SampleArgs s;
        memset(&s, 0, sizeof(s));
        s.head = ptr<float>(m->wenc1.head); s.ldH = 2 * m->Dp[0]; s.Dp = m->Dp[0]; s.D = m->D[0]; s.head_per_row = 0;
        s.M = M; s.Mp = Mp; s.k = k; s.B = B; s.eps = eps_src(m, 0);
        s.ZP = nullptr; s.ZF = ptr<float>(m->f32.z[0]); s.ldZF = Dz;
        s.cond = cond; s.C = m->C;      // (the sampling kernel writes y into features D .. D + C - 1 of every row)
        s.prior_head = m->has_prior ? ptr<float>(m->wprior.head) : nullptr;
        s.lp_prior = two ? nullptr : t1;
        s.lq = two ? t3 : t2;
        const bool want_dreg = !two && (objective == OBJ_DREG || (!bwd && !m->in_eval_llh));
        s.lq_dreg = want_dreg ? lqd : nullptr;
        launch_sample(s, st);
    }
    if (two) {       // q(z2|z1), z2, p(z1|z2)  (iwae2.py:63-65, :90, :118-124)
        const int b_enc2 = m->enc2[0].sub[0], b_dec2 = m->dec2[0].sub[0];
        CHK(ensure(m->wenc2.head, (size_t)Mp * 2 * m->Dp[1] * 4, st));
        CHK(f32_block_fwd(m, b_enc2, m->f32.enc2, ptr<float>(m->f32.z[0]), m->D[0], M, ptr<float>(m->wenc2.head), m->Dp[1], bwd));
        CHK(ensure(m->f32.z[1], (size_t)Mp * m->D[1] * 4, st));
        SampleArgs s;
        memset(&s, 0, sizeof(s));
        s.head = ptr<float>(m->wenc2.head); s.ldH = 2 * m->Dp[1]; s.Dp = m->Dp[1]; s.D = m->D[1]; s.head_per_row = 1;
        s.M = M; s.Mp = Mp; s.k = k; s.B = B; s.eps = eps_src(m, 1);
        s.ZP = nullptr; s.ZF = ptr<float>(m->f32.z[1]); s.ldZF = m->D[1];
        s.lp_prior = t2; s.lq = t4; s.lq_dreg = nullptr;
        launch_sample(s, st);
        CHK(ensure(m->wdec2.head, (size_t)Mp * 2 * m->Dp[0] * 4, st));
        CHK(f32_block_fwd(m, b_dec2, m->f32.dec2, ptr<float>(m->f32.z[1]), m->D[1], M, ptr<float>(m->wdec2.head), m->Dp[0], bwd));
        GaussLpArgs g;
        memset(&g, 0, sizeof(g));
        g.zhead = ptr<float>(m->wenc1.head); g.ldZH = 2 * m->Dp[0]; g.Dzp = m->Dp[0];
        g.phead = ptr<float>(m->wdec2.head); g.ldPH = 2 * m->Dp[0]; g.Dpp = m->Dp[0];
        g.D = m->D[0]; g.M = M; g.k = k; g.eps = eps_src(m, 0); g.out = t1;
        launch_gauss_lp(g, st);
    }
    // ---- decoder + Bernoulli log-likelihood (iwae1.py:81-83,111)
    const int b_dec1 = m->dec1[0].sub[0];
    const KerasLayer *d1 = &m->klayers[b_dec1], *d2 = d1 + 1, *d3 = d1 + 2;
    const int H = d1->Nout;
    CHK(join_side(m));      // the decoder's parameters (and g1, g2, s, which the previous step's weight gradients still read)
    // Round 4: the whole decoder forward in ONE launch where its shapes fit (dec_fwd_f32_kernel: rows stationary, activations in LDS, the
    // weights streamed from the float32 master parameters; log p(x|z) per row comes out whole) -- the k = 5000 evaluator's three GEMM launches
    // ran at 0.35 of the f32 MFMA peak between them.  A training step also keeps g1, g2 and s = x - sigmoid(l) for the backward pass.
    DecFwdF32Args df;
    memset(&df, 0, sizeof(df));
    df.Z = ptr<float>(m->f32.z[0]); df.ldz = Dz; df.Din = Dz; df.M = M; df.H = H; df.X = X;
    df.W1 = m->param + d1->offW; df.b1 = m->param + d1->offb; df.W2 = m->param + d2->offW; df.b2 = m->param + d2->offb;
    df.W3 = m->param + d3->offW; df.b3 = m->param + d3->offb;
    df.XB = xd; df.k = k; df.lpxz = lpxz; df.zero = m->d_zero; df.ldg = H; df.ldS = X;
    // (round 5: a TRAINING step takes the three GEMM launches again -- with gemm_f32_v2_kernel they are faster than the fused kernel once g1, g2 and s
    // have to be stored anyway: 1.280 -> 1.248 ms; option f32_dec_fused_train = 1 for the fused kernel)
    const bool fused_dec = m->allow_f32_dec_fused && (!bwd || m->f32_dec_fused_train) && !(want && want->logits) && M >= 4096 && dec_fwd_f32_ok(df);
    if (bwd || !fused_dec) {
        CHK(ensure(m->f32.g1, (size_t)M * H * 4, st));
        CHK(ensure(m->f32.g2, (size_t)M * H * 4, st));
    }
    if (fused_dec) {
        m->px_parts = 1;
        m->f32_keeps_s = false;
        if (bwd) {
            CHK(ensure(m->f32.logits, (size_t)M * X * 4, st));
            df.G1 = ptr<float>(m->f32.g1); df.G2 = ptr<float>(m->f32.g2); df.S = ptr<float>(m->f32.logits);
            m->f32_keeps_s = true;
        }
#ifdef IWAE_DENSE_STAMPS
        if (m->dstamp_epi == 11) {      // diagnostic (STAMPS=1 build, option dense_stamps_epi = 11): phase stamps of dec_fwd_f32_kernel
            m->dstamp_waves = ((M + 63) / 64) * 4;
            CHK(ensure(m->dstamps, (size_t)m->dstamp_waves * 64, st));
            df.stamps = ptr<unsigned long long>(m->dstamps);
        }
#endif
        launch_dec_fwd_f32(df, st);
    } else {
    CHK(f32_fwd(m, *d1, ptr<float>(m->f32.z[0]), Dz, M, ptr<float>(m->f32.g1), H, GEMM_EPI_TANH));
    CHK(f32_fwd(m, *d2, ptr<float>(m->f32.g1), H, M, ptr<float>(m->f32.g2), H, GEMM_EPI_TANH));
    // forward-only calls at large row counts (the k = 5000 evaluator): log p(x|z) in the epilogue of the output layer's GEMM -- the float32
    // logits (1.6 GB per launch of 2^19 rows) are neither written nor read back; per 64-column half tile a partial sum that lse_kernel adds
    // Round 3, training step: the same epilogue also leaves s = x - sigmoid(l) where the logits would have gone -- the backward pass reads s and
    // takes the row weight g_r inside its two consumers (f32_dw / f32_dx with rowscale) instead of a pass that rewrites 160 MB into dl = g_r s.
    const bool fuse_bern = m->allow_f32_bern_fused && !(want && want->logits) && gemm_f32_takes_big(M, X, 1);
    m->px_parts = 1;
    m->f32_keeps_s = false;
    if (fuse_bern) {
        m->px_parts = 2 * ((X + 127) / 128);
        CHK(ensure(m->px_part, (size_t)m->px_parts * Mp * 4, st));
        GemmF32Args ga;
        memset(&ga, 0, sizeof(ga));
        ga.A = ptr<float>(m->f32.g2); ga.sam = H; ga.sak = 1; ga.B = m->param + d3->offW; ga.sbk = d3->Nout; ga.sbn = 1; ga.M = M; ga.N = X; ga.K = H;
        ga.bias = m->param + d3->offb; ga.epi = GEMM_EPI_BERN; ga.kchunk = H;
        if (bwd) {
            CHK(ensure(m->f32.logits, (size_t)M * X * 4, st));
            ga.C = ptr<float>(m->f32.logits); ga.ldc = X;
            m->f32_keeps_s = true;
        }
        ga.XB = xd; ga.bern_k = k; ga.bern_X = X; ga.part = ptr<float>(m->px_part); ga.part_stride = (size_t)Mp;
        launch_gemm_f32(ga, 1, st);
    } else {
    CHK(ensure(m->f32.logits, (size_t)M * X * 4, st));
    CHK(f32_fwd(m, *d3, ptr<float>(m->f32.g2), H, M, ptr<float>(m->f32.logits), X, GEMM_EPI_NONE));
    launch_bern_f32(ptr<float>(m->f32.logits), X, xd, X, M, k, lpxz, st);
    }
    }      // (!fused_dec)
    if (want && want->logits) {      // reference [k,B,X] order
        CHK(ensure(m->scratch, (size_t)M * X * 4, st));
        launch_export_mat(ptr<float>(m->f32.logits), B, k, X, ptr<float>(m->scratch), st);
        CHK(copy_out(m, want->logits, m->scratch.p, (size_t)M * X * 4));
    }
    // ---- log_w, log-mean-exp over k, objectives (iwae1.py:113-139): the shared kernel
    CHK(ensure(m->logw, (size_t)Mp * 4, st));
    CHK(ensure(m->wn, (size_t)Mp * 4, st));
    {   // (as forward_impl: wgrad_rows_kernel's row-weighted path reads gx in whole 32-row stages -- 0 x a non-finite pad would be NaN)
        const void* before = m->gx.p;
        CHK(ensure(m->gx, (size_t)Mp * 4, st));
        if (m->gx.p != before) HIPCHK(hipMemsetAsync(m->gx.p, 0, m->gx.cap, st));
    }
    CHK(ensure(m->cf, (size_t)Mp * 16, st));
    CHK(ensure(m->per_b, (size_t)PB_COUNT * B * 4, st));
    {
        LseArgs a;
        memset(&a, 0, sizeof(a));
        if (!two) {
            a.term[0] = lpxz; a.coef[0] = 1.f; a.term[1] = t1; a.coef[1] = beta; a.term[2] = t2; a.coef[2] = -beta;
            a.head = ptr<float>(m->wenc1.head); a.ldH = 2 * m->Dp[0]; a.D = m->D[0]; a.Dp = m->Dp[0]; a.cz_on = 1.f;
        } else {
            a.term[0] = lpxz; a.coef[0] = 1.f; a.term[1] = t1; a.coef[1] = 1.f; a.term[2] = t2; a.coef[2] = 1.f;
            a.term[3] = t3; a.coef[3] = -1.f; a.term[4] = t4; a.coef[4] = -1.f; a.head = nullptr; a.cz_on = 0.f;
        }
        a.lq_dreg = (!two && (objective == OBJ_DREG || (!bwd && !m->in_eval_llh))) ? lqd : nullptr;
        a.B = B; a.k = k; a.beta = two ? 1.f : beta; a.objective = objective;
        a.lme_only = (!bwd && m->in_eval_llh && !want) ? 1 : 0;
        a.logw = ptr<float>(m->logw); a.wn = ptr<float>(m->wn); a.gx = ptr<float>(m->gx);
        a.cf = ptr<float4>(m->cf); a.per_b = ptr<float>(m->per_b);
        a.n_px_part = m->px_parts; a.px_stride = (size_t)Mp; a.term0_out = lpxz;
        if (m->px_parts > 1) a.term[0] = ptr<float>(m->px_part);      // (the fused Bernoulli epilogue's per-half-tile partial sums)
        launch_lse(a, st);
        launch_scalars(ptr<float>(m->per_b), B, two ? 1.f : beta, m->d_scalars, st);
    }
    HIPCHK(hipGetLastError());
    m->have_forward = true;
    m->fwd_was_f32 = true;
    return IWAE_OK;
}

// closed-form backward in float32: leaves the flat gradient in m->grad
// Round 5: two streams.  The decoder's three weight gradients (58 % of the backward pass's FLOPs, needed by nobody until the update) go to the
// side stream: the hidden layers' behind the dX product that makes their operand, the output layer's LAST -- it needs only s, g2 and the
// row weights, so it is what runs beside the main stream's few-row tail (dz, the latent sums, the encoder's seven launches on the batch's
// images: 64 workgroups each on 256 CUs).  fused_lr >= 0 (the single-GPU train step): the update is part of it -- the encoder's
// layers on the main stream, the decoder's on the side stream behind its own slab reduction, DEFERRED: the next step's encoder forward
// and sampling run beside the output layer's gradient, and forward_f32 joins (ev_dec) in front of the decoder forward.
int backward_f32(iwae_model* m, int objective, float fused_lr = -1.0f) {
    m->f32_slab_want_step = 0;
    if (!m->have_forward || !m->fwd_was_f32) return fail(IWAE_ERR_STATE, "backward without a float32 forward");
    const bool two = m->cfg.n_layers == 2;
    const int B = m->B, k = m->k, M = m->M, Mp = m->Mp, X = m->X;
    hipStream_t st = m->stream;
    const int b_dec1 = m->dec1[0].sub[0];
    const KerasLayer *d1 = &m->klayers[b_dec1], *d2 = d1 + 1, *d3 = d1 + 2;
    const int H = d1->Nout, D0 = m->D[0], Dp0 = m->Dp[0];
    float* dl = ptr<float>(m->f32.logits);
    const float* rw = nullptr;      // the row weight g_r, where the forward pass kept s instead of the logits (forward_f32): taken by the two consumers
    if (m->f32_keeps_s) rw = ptr<float>(m->gx);
    else launch_dl_f32(dl, X, m->f32_x, X, M, k, ptr<float>(m->gx), st);        // dl = g_r (x - sigmoid(l)), in place
    CHK(ensure(m->f32.d2, (size_t)M * H * 4, st));
    CHK(ensure(m->f32.d1, (size_t)M * H * 4, st));
    CHK(ensure(m->wdec1.dz, (size_t)Mp * Dp0 * 4, st));
    if (m->descs_dirty) CHK(build_descs(m));
    // (the conditional prior's block sits BEHIND the decoder in the flat parameters: its gradient is made on the main stream -- one stream for that model)
    const bool use_side = m->allow_f32_side && m->side && !m->has_prior && M >= 4096 && b_dec1 + 3 == (int)m->klayers.size();
    m->f32_side_active = use_side;
    const int sg = use_side ? 1 : 0;
    const bool dw_last = use_side && m->f32_dw_last > 0;      // option: every decoder weight gradient behind the dX chain, beside the main stream's few-row tail
    if (use_side) {
        HIPCHK(hipEventRecord(m->ev_fork, st));      // s, g1, g2, z, the row weights
        HIPCHK(hipStreamWaitEvent(m->side, m->ev_fork, 0));
        if (m->f32_wout_first && !dw_last) CHK(f32_dw(m, *d3, ptr<float>(m->f32.g2), H, dl, X, M, rw, 1));
    } else {
        CHK(f32_dw(m, *d3, ptr<float>(m->f32.g2), H, dl, X, M, rw));
    }
    CHK(f32_dx(m, *d3, dl, X, M, ptr<float>(m->f32.d2), H, ptr<float>(m->f32.g2), H, false, rw));
    if (use_side && !dw_last) { HIPCHK(hipEventRecord(m->ev_fork2, st)); HIPCHK(hipStreamWaitEvent(m->side, m->ev_fork2, 0)); }
    if (!dw_last) CHK(f32_dw(m, *d2, ptr<float>(m->f32.g1), H, ptr<float>(m->f32.d2), H, M, nullptr, sg));
    CHK(f32_dx(m, *d2, ptr<float>(m->f32.d2), H, M, ptr<float>(m->f32.d1), H, ptr<float>(m->f32.g1), H, false));
    if (use_side && !dw_last) { HIPCHK(hipEventRecord(m->ev_blk, st)); HIPCHK(hipStreamWaitEvent(m->side, m->ev_blk, 0)); }
    if (!dw_last) {
        CHK(f32_dw(m, *d1, ptr<float>(m->f32.z[0]), D0 + m->C, ptr<float>(m->f32.d1), H, M, nullptr, sg));
        if (use_side) { HIPCHK(hipEventRecord(m->ev_join, m->side)); m->f32_z_pending = true; }      // (z is free for the next step's sampling)
        if (use_side && !m->f32_wout_first) CHK(f32_dw(m, *d3, ptr<float>(m->f32.g2), H, dl, X, M, rw, 1));
    }
    CHK(f32_dx(m, *d1, ptr<float>(m->f32.d1), H, M, ptr<float>(m->wdec1.dz), Dp0, nullptr, 0, false));
    if (dw_last) {
        const int tmode = m->f32_dw_last - 1;      // (1: tiles as picked, 2: 4-wave tiles, 3: 4-wave tiles at 3 waves per SIMD)
        HIPCHK(hipEventRecord(m->ev_blk, st));
        HIPCHK(hipStreamWaitEvent(m->side, m->ev_blk, 0));
        CHK(f32_dw(m, *d1, ptr<float>(m->f32.z[0]), D0 + m->C, ptr<float>(m->f32.d1), H, M, nullptr, 1, tmode));
        HIPCHK(hipEventRecord(m->ev_join, m->side));
        m->f32_z_pending = true;
        CHK(f32_dw(m, *d2, ptr<float>(m->f32.g1), H, ptr<float>(m->f32.d2), H, M, nullptr, 1, tmode));
        CHK(f32_dw(m, *d3, ptr<float>(m->f32.g2), H, dl, X, M, rw, 1, tmode));
    }
    const float *dz1_b = nullptr, *dz1_c = nullptr;
    if (two) {
        const int b_enc2 = m->enc2[0].sub[0], b_dec2 = m->dec2[0].sub[0];
        const int Dp1 = m->Dp[1];
        CHK(ensure(m->dzdir, (size_t)Mp * Dp0 * 4, st));
        CHK(ensure(m->f32.dec2.dhead, (size_t)Mp * 2 * Dp0 * 4, st));
        CHK(ensure(m->f32.dec2.dx, (size_t)Mp * Dp1 * 4, st));
        CHK(ensure(m->f32.enc2.dhead, (size_t)Mp * 2 * Dp1 * 4, st));
        CHK(ensure(m->f32.enc2.dx, (size_t)Mp * Dp0 * 4, st));
        HIPCHK(hipMemsetAsync(m->f32.dec2.dhead.p, 0, (size_t)Mp * 2 * Dp0 * 4, st));      // (pad columns are read by the weight-gradient GEMMs' strided views: keep them zero)
        HIPCHK(hipMemsetAsync(m->f32.enc2.dhead.p, 0, (size_t)Mp * 2 * Dp1 * 4, st));
        GaussBwdArgs g;
        memset(&g, 0, sizeof(g));
        g.mode = 0; g.G = ptr<float>(m->gx);
        g.head = ptr<float>(m->wdec2.head); g.ldH = 2 * Dp0; g.D = D0; g.Dp = Dp0;
        g.zhead = ptr<float>(m->wenc1.head); g.ldZH = 2 * Dp0; g.Dzp = Dp0;
        g.dz_direct = ptr<float>(m->dzdir); g.ldDZ = Dp0;
        g.eps = eps_src(m, 0); g.M = M; g.Mp = Mp; g.k = k;
        g.DHP = nullptr; g.DHF = ptr<float>(m->f32.dec2.dhead);
        launch_gauss_bwd(g, st);
        CHK(f32_block_bwd(m, b_dec2, m->f32.dec2, ptr<float>(m->f32.z[1]), m->D[1], M, Dp0, ptr<float>(m->f32.dec2.dx), Dp1));
        memset(&g, 0, sizeof(g));
        g.mode = 1; g.G = ptr<float>(m->gx);
        g.head = ptr<float>(m->wenc2.head); g.ldH = 2 * Dp1; g.D = m->D[1]; g.Dp = Dp1;
        g.dz_in = ptr<float>(m->f32.dec2.dx); g.ldDZ = Dp1;
        g.eps = eps_src(m, 1); g.M = M; g.Mp = Mp; g.k = k;
        g.DHP = nullptr; g.DHF = ptr<float>(m->f32.enc2.dhead);
        launch_gauss_bwd(g, st);
        CHK(f32_block_bwd(m, b_enc2, m->f32.enc2, ptr<float>(m->f32.z[0]), D0, M, Dp1, ptr<float>(m->f32.enc2.dx), Dp0));
        dz1_b = ptr<float>(m->dzdir); dz1_c = ptr<float>(m->f32.enc2.dx);
    }
    {
        CHK(ensure(m->f32.enc1.dhead, (size_t)m->Bp * 2 * Dp0 * 4, st));
        HIPCHK(hipMemsetAsync(m->f32.enc1.dhead.p, 0, (size_t)m->Bp * 2 * Dp0 * 4, st));
        LatentBwdArgs a;
        memset(&a, 0, sizeof(a));
        a.dz = ptr<float>(m->wdec1.dz); a.dz2 = dz1_b; a.dz3 = dz1_c; a.ldDZ = Dp0;
        a.head = ptr<float>(m->wenc1.head); a.ldH = 2 * Dp0; a.D = D0; a.Dp = Dp0;
        a.cf = ptr<float4>(m->cf); a.eps = eps_src(m, 0);
        a.B = B; a.Bp = m->Bp; a.k = k;
        a.kmu = a.ksig = (objective == OBJ_VAE_ELBO_KL) ? m->beta / (float)B : 0.f;
        a.DHP = nullptr; a.DHF = ptr<float>(m->f32.enc1.dhead);
        if (m->has_prior) {      // gradient of the conditional prior's head, summed over the image's samples (tasks/task04.py:124-130)
            CHK(ensure(m->f32.prior.dhead, (size_t)m->Bp * 2 * Dp0 * 4, st));
            HIPCHK(hipMemsetAsync(m->f32.prior.dhead.p, 0, (size_t)m->Bp * 2 * Dp0 * 4, st));
            a.prior_head = ptr<float>(m->wprior.head); a.DHF2 = ptr<float>(m->f32.prior.dhead);
        }
        launch_latent_bwd(a, st);
    }
    if (m->has_prior)
        CHK(f32_block_bwd(m, m->prior[0].sub[0], m->f32.prior, ptr<float>(m->cond) + (size_t)m->cond_row0 * m->C, m->C, B, Dp0, nullptr, 0));
    CHK(f32_block_bwd(m, m->enc1[0].sub[0], m->f32.enc1, m->C > 0 ? ptr<float>(m->f32.xcat) : m->f32_x, X + m->C, B, Dp0, nullptr, 0));
    m->f32_slab_want = std::max(m->f32_slab_want, m->f32_slab_want_step);
    m->split_offset = m->nparam;       // (data-parallel step: one all-reduce of the whole gradient)
    if (!use_side) {
        CHK(f32_flush_reductions(m));      // every row-split gradient's slabs -> the flat gradient, one launch
        if (fused_lr >= 0.0f) CHK(adam_impl(m, fused_lr, 1.0f));
        HIPCHK(hipGetLastError());
        return IWAE_OK;
    }
    const int b0 = m->descs[b_dec1].block_begin;
    const float alpha = fused_lr >= 0.0f ? adam_alpha(m, fused_lr) : 0.0f;
    CHK(f32_flush_reductions(m, 0));       // the slabs of the main stream's gradients (every block but the decoder)
    if (fused_lr >= 0.0f)
        launch_adam(m->d_descs, (int)m->descs.size(), b0, m->param, m->grad, m->mom, m->vel, alpha, 1.0f, m->adam_b1, m->adam_b2, m->adam_eps, 1, st, 0);
    CHK(f32_flush_reductions(m, 1));       // the decoder's, on the side stream
    if (fused_lr >= 0.0f)
        launch_adam(m->d_descs, (int)m->descs.size(), m->elem_blocks - b0, m->param, m->grad, m->mom, m->vel, alpha, 1.0f, m->adam_b1, m->adam_b2, m->adam_eps, 1, m->side, b0);
    HIPCHK(hipEventRecord(m->ev_dec, m->side));
    m->dec_pending = true;                 // (join_side: whoever reads the decoder's gradient or parameters next)
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

// Data-parallel step, second half (the gradient of this rank's shard is in m->grad; backward_impl(split) left the decoder's
// segment on the side stream, unjoined): all-reduce + Adam(grad_scale 1/N) of the decoder's layers on the SIDE stream -- they run
// beside the encoder's backward pass and the next encoder forward, as the single-GPU step's deferred update does -- and of the
// encoder's layers on the main stream.  Models without such a segment: one all-reduce + Adam on the main stream.
int dp_finish(iwae_model* m, float lr) {
    const float alpha = adam_alpha(m, lr);
    const float gs = 1.0f / (float)m->comm_world;
    if (m->descs_dirty) CHK(build_descs(m));
    const size_t n = m->nparam, off = m->split_offset;
    if (m->early_held && !(off < n && m->dec_pending)) {      // (cannot happen on today's call paths -- nothing joins between backward_impl and here --; kept correct anyway: the held reduction runs now, joined)
        launch_reduce_grads(m->d_descs, (int)m->descs.size(), m->early_first, m->reduce_blocks - m->early_first, m->grad, m->param, m->mom,
                            m->vel, 0.0f, m->adam_b1, m->adam_b2, m->adam_eps, 0, nullptr, 0, 0.f, nullptr, m->tail);
        HIPCHK(hipEventRecord(m->ev_join, m->tail));
        HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0));
        m->early_held = false;
    }
    if (off < n && m->dec_pending) {
        const int b0 = m->descs[m->dec1[0].sub[0]].block_begin;
        // Two communicators, one per stream.  Until an N > 1 run has shown that the two collectives may be co-resident, they are ORDERED
        // on the device, and in the order in which their inputs become ready: the encoder's segment first (main stream: its gradient is
        // complete ~30 us before the decoder's, whose reduction waits for the hidden layers' weight gradients), the decoder's behind an
        // event recorded after it -- a wait that is normally already satisfied.  (Round 3 first had them the other way round: the main
        // stream's update and the next encoder forward then waited for the decoder's reduction, +22 us per step in the one-rank
        // rehearsal.)  Every rank enqueues them in this order.  Option dp_concurrent = 1 drops the wait.
        { ScopedTimer tm(m, T_AR_ENC); NCCLCHK(g_rccl.AllReduce(m->grad, m->grad, off, ncclFloat32, ncclSum, m->comm_main, m->stream)); }
        // Round 5: the order costs (almost) nothing.  The event rides on the dispatch packet of the encoder's update (the kernel right behind
        // the all-reduce: no record bubble on the main stream), and the tail stream waits for it IN FRONT of the decoder's slab reduction --
        // which backward_impl left to this function (early_held) -- i.e. right behind the wait for the output layer's gradient it performs
        // there anyway, ~15 us before the decoder's exchange instead of directly in front of it (a barrier packet costs its 6-10 us wherever
        // its event stands; here it falls into the shadow of the hidden layers' gradients).  Measured in the one-rank rehearsal: see DESIGN.md 8.
        if (!m->dp_concurrent) set_launch_stop_event(m->ev_ar);
        launch_adam(m->d_descs, (int)m->descs.size(), b0, m->param, m->grad, m->mom, m->vel, alpha, gs, m->adam_b1, m->adam_b2, m->adam_eps, 1, m->stream, 0);
        if (!m->dp_concurrent) HIPCHK(hipStreamWaitEvent(m->tail, m->ev_ar, 0));
        if (m->early_held) {
            launch_reduce_grads(m->d_descs, (int)m->descs.size(), m->early_first, m->reduce_blocks - m->early_first, m->grad, m->param, m->mom,
                                m->vel, 0.0f, m->adam_b1, m->adam_b2, m->adam_eps, 0, nullptr, 0, 0.f, nullptr, m->tail);
            m->early_held = false;
        }
        { ScopedTimer tm(m, T_AR_DEC, m->tail); NCCLCHK(g_rccl.AllReduce(m->grad + off, m->grad + off, n - off, ncclFloat32, ncclSum, m->comm_side, m->tail)); }
        set_launch_stop_event(m->ev_dec);           // join_side() now waits for the decoder's UPDATE, not just its gradient
        launch_adam(m->d_descs, (int)m->descs.size(), m->elem_blocks - b0, m->param, m->grad, m->mom, m->vel, alpha, gs, m->adam_b1, m->adam_b2, m->adam_eps, 1,
                    m->tail, b0);
    } else {
        CHK(join_side(m));
        { ScopedTimer tm(m, T_AR_ENC); NCCLCHK(g_rccl.AllReduce(m->grad, m->grad, n, ncclFloat32, ncclSum, m->comm_main, m->stream)); }
        launch_adam(m->d_descs, (int)m->descs.size(), m->elem_blocks, m->param, m->grad, m->mom, m->vel, alpha, gs, m->adam_b1, m->adam_b2, m->adam_eps, 1, m->stream, 0);
    }
    HIPCHK(hipGetLastError());
    return IWAE_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

const char* iwae_last_error(void) { return g_err.c_str(); }
int iwae_version(void) { return 1; }
#ifndef IWAE_BUILD_ID
#define IWAE_BUILD_ID "unknown"
#endif
static const char kBuildIdMarker[] = "IWAE_BUILD_ID=" IWAE_BUILD_ID;      // (the marker lets a checker read the id from the file without loading it)
const char* iwae_build_id(void) { return kBuildIdMarker + 14; }

int iwae_create(const iwae_config* cfg, iwae_handle* out) {
    if (!cfg || !out) return fail(IWAE_ERR_ARG, "iwae_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(iwae_config))
        return fail(IWAE_ERR_ARG, "iwae_create: iwae_config.struct_size is " + std::to_string(cfg->struct_size) + ", this library's iwae_config has " +
                                      std::to_string(sizeof(iwae_config)) + " bytes (binding built against another include/iwae_amd.h?)");
    if (cfg->reserved != 0) return fail(IWAE_ERR_ARG, "iwae_config.reserved must be 0");
    if (cfg->precision != IWAE_PREC_BF16 && cfg->precision != IWAE_PREC_FP32) return fail(IWAE_ERR_ARG, "precision must be IWAE_PREC_BF16 or IWAE_PREC_FP32");
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) return fail(IWAE_ERR_ARG, "need world_size >= 1 and 0 <= rank < world_size");
    if (cfg->n_layers != 1 && cfg->n_layers != 2) return fail(IWAE_ERR_ARG, "n_layers must be 1 or 2 (main.py:17)");
    for (int i = 0; i < cfg->n_layers; ++i) {
        if (cfg->n_hidden[i] < 1 || cfg->n_hidden[i] > 256) return fail(IWAE_ERR_ARG, "n_hidden must be in [1,256]");
        if (cfg->n_latent[i] < 1 || cfg->n_latent[i] > 128) return fail(IWAE_ERR_ARG, "n_latent must be in [1,128]");
    }
    if (cfg->x_dim < 1 || cfg->x_dim > 4096) return fail(IWAE_ERR_ARG, "x_dim must be in [1,4096]");
    if (cfg->cond_dim < 0 || cfg->cond_dim > 64) return fail(IWAE_ERR_ARG, "cond_dim must be in [0,64]");
    if (cfg->cond_dim > 0 && cfg->n_layers != 1) return fail(IWAE_ERR_ARG, "the conditional model is 1-layer (tasks/task05.py:101)");
    if (cfg->cond_prior != 0 && cfg->cond_dim <= 0) return fail(IWAE_ERR_ARG, "cond_prior needs cond_dim > 0 (tasks/task04.py)");
    if (cfg->cond_dim > 0 && cfg->n_latent[0] + cfg->cond_dim > round_up(cfg->n_latent[0], 32))
        return fail(IWAE_ERR_ARG, "conditional model: n_latent + cond_dim must fit the 32-feature padding of z");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(IWAE_ERR_HIP, "no HIP device: the IWAE hot path needs an AMD GPU (there is no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(IWAE_ERR_ARG, "device " + std::to_string(cfg->device) + " out of range (" + std::to_string(ndev) + " HIP devices)");
    HIPCHK(hipSetDevice(cfg->device));
    // the half-built model is owned by `guard` until the very end: every failing path below (HIPCHK / CHK return) frees
    // its streams, events and device memory through iwae_destroy
    std::unique_ptr<iwae_model, void (*)(iwae_model*)> guard(new iwae_model(), iwae_destroy);
    iwae_model* m = guard.get();
    m->cfg = *cfg;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) m->num_cus = ncu;
    }

    m->X = cfg->x_dim;
    m->Xp32 = round_up(cfg->x_dim, 32);
    m->C = cfg->cond_dim;
    m->Xinp = round_up(cfg->x_dim + cfg->cond_dim, 32);
    for (int i = 0; i < 2; ++i) {
        m->H[i] = cfg->n_hidden[i]; m->D[i] = cfg->n_latent[i];
        m->Hp[i] = round_up(std::max(1, m->H[i]), 32); m->Dp[i] = round_up(std::max(1, m->D[i]), 32);
    }
    HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    m->own_stream = true;
    {   // the side stream carries work with slack (weight gradients, next step's noise, the deferred decoder update): lowest
        // priority, so the main stream's dependency chain gets the CUs first whenever both have workgroups ready
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        int prio = least;
        HIPCHK(hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, prio));
        prio = least;
        HIPCHK(hipStreamCreateWithPriority(&m->side2, hipStreamNonBlocking, prio));
        HIPCHK(hipEventCreateWithFlags(&m->ev_s2, hipEventDisableTiming));
        m->tail = m->side;
    }
    HIPCHK(hipEventCreateWithFlags(&m->ev_lse, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_fork2, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_join2, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_dec, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_blk, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_dec2, hipEventDisableTiming));
    if (cfg->n_layers == 1) {
        add_block(m, m->enc1, "enc", m->X + m->C, m->H[0], m->D[0], false);      // tasks/task05.py:113-118 when C > 0
        add_mlp3(m, m->dec1, "dec", m->D[0] + m->C, m->H[0], m->X);
        m->has_prior = cfg->cond_prior != 0;
        if (m->has_prior) add_block(m, m->prior, "prior", m->C, m->H[0], m->D[0], false);      // tasks/task04.py:108 (after the decoder)
    } else {
        add_block(m, m->enc1, "enc1", m->X, m->H[0], m->D[0], false);
        add_block(m, m->enc2, "enc2", m->D[0], m->H[1], m->D[1], true);
        add_block(m, m->dec2, "dec2", m->D[1], m->H[1], m->D[0], true);
        add_mlp3(m, m->dec1, "dec1", m->D[0], m->H[0], m->X);
    }
    for (Linear* L : all_linears(m)) CHK(alloc_linear(*L));
    const size_t nb = m->nparam * 4;
    HIPCHK(hipMalloc((void**)&m->param, nb));
    HIPCHK(hipMalloc((void**)&m->grad, nb));
    HIPCHK(hipMalloc((void**)&m->mom, nb));
    HIPCHK(hipMalloc((void**)&m->vel, nb));
    HIPCHK(hipMemset(m->grad, 0, nb));
    HIPCHK(hipMemset(m->mom, 0, nb));
    HIPCHK(hipMemset(m->vel, 0, nb));
    HIPCHK(hipMalloc((void**)&m->d_zero, 1024));
    HIPCHK(hipMemset(m->d_zero, 0, 1024));
    HIPCHK(hipMalloc((void**)&m->d_scalars, SC_COUNT * 4));
    HIPCHK(hipMemset(m->d_scalars, 0, SC_COUNT * 4));
    HIPCHK(hipHostMalloc((void**)&m->h_scalars, SC_COUNT * 4));
    // Keras Dense defaults: glorot-uniform kernels, zero biases (src/iwae1.py:31-34,72-75)
    std::vector<float> init(m->nparam, 0.f);
    std::mt19937_64 rng(cfg->seed ^ 0x9E3779B97F4A7C15ull);
    for (const KerasLayer& kl : m->klayers) {
        const double lim = sqrt(6.0 / (double)(kl.Kin + kl.Nout));
        std::uniform_real_distribution<double> U(-lim, lim);
        for (size_t i = 0; i < (size_t)kl.Kin * kl.Nout; ++i) init[kl.offW + i] = (float)U(rng);
    }
    HIPCHK(hipMemcpy(m->param, init.data(), nb, hipMemcpyHostToDevice));
    CHK(refresh_images(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    *out = guard.release();
    return IWAE_OK;
}

void iwae_destroy(iwae_handle m) {
    if (!m) return;
    (void)hipSetDevice(m->cfg.device);
    if (m->side) (void)hipStreamSynchronize(m->side);
    if (m->side2) (void)hipStreamSynchronize(m->side2);      // the decoder's all-reduce + Adam run on `tail`, which may be side2
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm_main && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm_main);
    if (m->comm_side && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm_side);
    for (Linear* L : all_linears(m)) free_linear(*L);
    DevBuf* bufs[] = {&m->xin, &m->xP, &m->epsbuf, &m->zP[0], &m->zP[1], &m->rows[0], &m->rows[1],
                      &m->rows[2], &m->rows[3], &m->rows[4], &m->rows[5], &m->logw, &m->wn, &m->gx, &m->cf, &m->per_b, &m->logw2, &m->wn2, &m->gx2, &m->cf2, &m->per_b2,
                      &m->dzdir, &m->scratch, &m->ds_data, &m->ds_order, &m->dstamps, &m->px_part, &m->dg2_part, &m->cond, &m->condP, &m->epsc[0][0], &m->epsc[0][1], &m->epsc[1][0], &m->epsc[1][1], &m->epsc[2][0], &m->epsc[2][1], &m->eval_x, &m->eval_lme,
                      &m->ds_labels, &m->epsm[0][0], &m->epsm[0][1], &m->epsm[1][0], &m->epsm[1][1]};
    for (DevBuf* b : bufs) free_buf(*b);
    BlockWs* bw[] = {&m->wenc1, &m->wenc2, &m->wdec2, &m->wprior};
    for (BlockWs* w : bw) {
        DevBuf* bb[] = {&w->h1P, &w->h2P, &w->head, &w->dheadP, &w->d2P, &w->d1P, &w->dx};
        for (DevBuf* b : bb) free_buf(*b);
    }
    {
        MlpWs* w = &m->wdec1;
        DevBuf* bb[] = {&w->g1P, &w->g2P, &w->dlP, &w->d2P, &w->d1P, &w->dz};
        for (DevBuf* b : bb) free_buf(*b);
    }
    {
        iwae_model::F32Block* fb[] = {&m->f32.enc1, &m->f32.enc2, &m->f32.dec2, &m->f32.prior};
        for (auto* w : fb) { DevBuf* bb[] = {&w->h1, &w->h2, &w->dhead, &w->d2, &w->d1, &w->dx}; for (DevBuf* b : bb) free_buf(*b); }
        DevBuf* bb[] = {&m->f32.z[0], &m->f32.z[1], &m->f32.g1, &m->f32.g2, &m->f32.logits, &m->f32.d2, &m->f32.d1, &m->f32.slab, &m->f32.bpart, &m->f32.xcat, &m->f32.kslab};
        for (DevBuf* b : bb) free_buf(*b);
    }
    if (m->param) (void)hipFree(m->param);
    if (m->grad) (void)hipFree(m->grad);
    if (m->mom) (void)hipFree(m->mom);
    if (m->vel) (void)hipFree(m->vel);
    if (m->d_descs) (void)hipFree(m->d_descs);
    if (m->d_zero) (void)hipFree(m->d_zero);
    if (m->d_scalars) (void)hipFree(m->d_scalars);
    if (m->h_scalars) (void)hipHostFree(m->h_scalars);
    for (int i = 0; i < T_COUNT; ++i) {
        for (hipEvent_t e : m->ev_start[i]) (void)hipEventDestroy(e);
        for (hipEvent_t e : m->ev_stop[i]) (void)hipEventDestroy(e);
    }
    if (m->side2) { (void)hipStreamSynchronize(m->side2); (void)hipStreamDestroy(m->side2); }
    if (m->ev_s2) (void)hipEventDestroy(m->ev_s2);
    if (m->ev_ar) (void)hipEventDestroy(m->ev_ar);
    if (m->side) { (void)hipStreamSynchronize(m->side); (void)hipStreamDestroy(m->side); }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_fork2) (void)hipEventDestroy(m->ev_fork2);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->ev_join2) (void)hipEventDestroy(m->ev_join2);
    if (m->ev_dec) (void)hipEventDestroy(m->ev_dec);
    if (m->ev_lse) (void)hipEventDestroy(m->ev_lse);
    if (m->ev_dec2) (void)hipEventDestroy(m->ev_dec2);
    if (m->ev_blk) (void)hipEventDestroy(m->ev_blk);
    if (m->own_stream && m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int iwae_set_stream(iwae_handle m, void* s) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    CHK(join_side(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->own_stream) { HIPCHK(hipStreamDestroy(m->stream)); m->own_stream = false; }
    if (s) {
        m->stream = (hipStream_t)s;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        m->own_stream = true;
    }
    return IWAE_OK;
}

int iwae_sync(iwae_handle m) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    CHK(join_side(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    return IWAE_OK;
}

int iwae_param_count(iwae_handle m, size_t* n) {
    if (!m || !n) return fail(IWAE_ERR_ARG, "null argument");
    *n = m->nparam;
    return IWAE_OK;
}
int iwae_num_tensors(iwae_handle m, int32_t* n) {
    if (!m || !n) return fail(IWAE_ERR_ARG, "null argument");
    *n = (int32_t)m->klayers.size() * 2;
    return IWAE_OK;
}
int iwae_tensor_info(iwae_handle m, int32_t idx, char* name, size_t cap, int32_t* rows, int32_t* cols, size_t* offset) {
    if (!m || idx < 0 || idx >= (int32_t)m->klayers.size() * 2) return fail(IWAE_ERR_ARG, "tensor index out of range");
    const KerasLayer& kl = m->klayers[idx / 2];
    const bool bias = idx & 1;
    if (name && cap) snprintf(name, cap, "%s/%s", kl.name.c_str(), bias ? "bias" : "kernel");
    if (rows) *rows = bias ? kl.Nout : kl.Kin;
    if (cols) *cols = bias ? 1 : kl.Nout;
    if (offset) *offset = bias ? kl.offb : kl.offW;
    return IWAE_OK;
}

int iwae_set_params(iwae_handle m, const float* flat, size_t n) {
    if (!m || !flat || n != m->nparam) return fail(IWAE_ERR_ARG, "set_params: size mismatch");
    CHK(join_side(m));
    HIPCHK(hipMemcpyAsync(m->param, flat, n * 4, hipMemcpyDefault, m->stream));
    CHK(refresh_images(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    return IWAE_OK;
}
int iwae_get_params(iwae_handle m, float* flat, size_t n) {
    if (!m || !flat || n != m->nparam) return fail(IWAE_ERR_ARG, "get_params: size mismatch");
    CHK(join_side(m));
    HIPCHK(hipMemcpyAsync(flat, m->param, n * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return IWAE_OK;
}
int iwae_set_output_bias(iwae_handle m, const float* bias, size_t n) {
    if (!m || !bias || n != (size_t)m->X) return fail(IWAE_ERR_ARG, "set_output_bias: need x_dim values");
    CHK(join_side(m));
    HIPCHK(hipMemcpyAsync(m->param + m->klayers[m->dec1[2].sub[0]].offb, bias, n * 4, hipMemcpyDefault, m->stream));
    CHK(refresh_images(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    return IWAE_OK;
}
int iwae_get_grads(iwae_handle m, float* flat, size_t n) {
    if (!m || !flat || n != m->nparam) return fail(IWAE_ERR_ARG, "get_grads: size mismatch");
    CHK(join_side(m));
    HIPCHK(hipMemcpyAsync(flat, m->grad, n * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return IWAE_OK;
}
int iwae_get_adam_state(iwae_handle m, float* mo, float* ve, size_t n, int64_t* step) {
    if (!m || n != m->nparam) return fail(IWAE_ERR_ARG, "get_adam_state: size mismatch");
    CHK(join_side(m));
    if (mo) HIPCHK(hipMemcpyAsync(mo, m->mom, n * 4, hipMemcpyDefault, m->stream));
    if (ve) HIPCHK(hipMemcpyAsync(ve, m->vel, n * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (step) *step = m->adam_t;
    return IWAE_OK;
}
int iwae_set_adam_state(iwae_handle m, const float* mo, const float* ve, size_t n, int64_t step) {
    if (!m || !mo || !ve || n != m->nparam || step < 0) return fail(IWAE_ERR_ARG, "set_adam_state: bad argument");
    CHK(join_side(m));
    HIPCHK(hipMemcpyAsync(m->mom, mo, n * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipMemcpyAsync(m->vel, ve, n * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->adam_t = step;
    return IWAE_OK;
}

int iwae_forward(iwae_handle m, const float* x, int32_t B, int32_t k, float beta, const float* eps, iwae_scalars* scalars,
                 const iwae_tensors* want) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(m->cfg.device));
    if (m->cfg.precision == IWAE_PREC_FP32) CHK(forward_f32(m, x, B, k, beta, eps, OBJ_IWAE_ELBO, false, want));
    else CHK(forward_impl(m, x, B, k, beta, eps, OBJ_IWAE_ELBO, false, want));
    CHK(fetch_outputs(m, scalars, want));
    m->noise_step += 1;
    return IWAE_OK;
}

int iwae_forward_backward(iwae_handle m, const float* x, int32_t B, int32_t k, float beta, int32_t objective, const float* eps,
                          iwae_scalars* scalars, const iwae_tensors* want) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(check_objective(m, objective));
    if (m->cfg.precision == IWAE_PREC_FP32) {
        CHK(forward_f32(m, x, B, k, beta, eps, objective, true, want));
        CHK(backward_f32(m, objective));
    } else {
        CHK(forward_impl(m, x, B, k, beta, eps, objective, true, want));
        CHK(backward_impl(m, objective));
    }
    CHK(fetch_outputs(m, scalars, want));
    m->noise_step += 1;
    return IWAE_OK;
}

int iwae_forward_backward_split(iwae_handle m, const float* x, int32_t B, int32_t k, float beta, int32_t objective, const float* eps,
                                void** side_stream, size_t* side_offset) {
    if (!m || !side_stream || !side_offset) return fail(IWAE_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(check_objective(m, objective));
    if (m->cfg.precision == IWAE_PREC_FP32) {       // float32 mode: nothing is left on the side stream (*side_offset = n)
        CHK(forward_f32(m, x, B, k, beta, eps, objective, true, nullptr));
        CHK(backward_f32(m, objective));
        CHK(join_side(m));
    } else {
        CHK(forward_impl(m, x, B, k, beta, eps, objective, true, nullptr));
        CHK(backward_impl(m, objective, -1.0f, true));
    }
    *side_stream = (void*)m->tail;
    *side_offset = m->split_offset;
    m->noise_step += 1;
    return IWAE_OK;
}

int iwae_grad_devptr(iwae_handle m, void** p, size_t* n) {
    if (!m || !p || !n) return fail(IWAE_ERR_ARG, "null argument");
    CHK(join_side(m));
    *p = m->grad;
    *n = m->nparam;
    return IWAE_OK;
}

int iwae_adam_step(iwae_handle m, float lr, float grad_scale) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(join_side(m));
    return adam_impl(m, lr, grad_scale);
}

// Kernel-selection switches of a handle (A/B measurements and the parity tests that compare kernel variants; the defaults are the
// measured best).  The library never reads the environment: this call is the only way to steer it.  Names: tools/README.md.
int iwae_set_option(iwae_handle m, const char* name, int64_t value) {
    if (!m || !name) return fail(IWAE_ERR_ARG, "set_option: null argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    // a switch changes which kernels the next call launches: nothing of the previous calls may still be in flight
    CHK(join_side(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->side) HIPCHK(hipStreamSynchronize(m->side));
    if (m->side2) HIPCHK(hipStreamSynchronize(m->side2));
    m->have_forward = false;
    const bool on = value != 0;
    const int iv = (int)value;
    const std::string n(name);
    if (n == "out_recompute") m->allow_s_mode = !on;                  // recompute the logits in out_bwd instead of keeping s
    else if (n == "no_defer") m->allow_defer = !on;                   // join the decoder update at the end of every step
    else if (n == "wout_split") m->wout_split = std::max(0, std::min(95, iv));      // percent of the rows in the output layer's EARLY gradient launch (0: one launch)
    else if (n == "wout_wg1") m->wout_wg1 = std::max(1, iv);          // ... its workgroups / those of the late launch
    else if (n == "wout_wg2") m->wout_wg2 = std::max(1, iv);
    else if (n == "defer_split") m->defer_split = on;                 // 1-layer step: one deferred decoder update per side stream
    else if (n == "no_eps_multi") m->allow_eps_multi = !on;           // few rows: one noise-draw launch per step instead of one per 8 steps
    else if (n == "no_zin") m->allow_zin = !on;                       // always the separate sampling kernel
    else if (n == "zin_eval") m->allow_zin_eval = on;                 // forward-only calls: z made in the decoder kernel's prologue (measured slower)
    else if (n == "no_chain2_bwd") m->allow_chain2_bwd = !on;         // ... only their backward unfused
    else if (n == "no_chain2") m->allow_chain2 = !on;                 // 2-layer model: the per-sample blocks unfused
    else if (n == "no_dec_bwd") m->allow_dec_bwd = !on;               // the decoder's dX chain as three launches
    else if (n == "no_defer2_split") m->allow_defer2_split = !on;     // ... one deferred update on `tail` instead of one per side stream
    else if (n == "no_defer2") m->allow_defer2 = !on;                 // 2-layer step: one reduction + update of all layers on the main stream
    else if (n == "f32_dw_tiles") m->f32_dw_tiles = std::max(1, iv);
    else if (n == "f32_dw_min_rows") m->f32_dw_min_rows = std::max(16, iv);
    else if (n == "f32_gemm_dbg") g_gemm_f32_dbg = (int)value;         // DIAG builds: timing ablations of gemm_f32_v2_kernel (1 no fetch, 2 no stash, 4 no MFMAs, 16 no barrier)
    else if (n == "f32_gemm_small_min") g_gemm_f32_v2_small_min = std::max(1, iv);
    else if (n == "f32_ksplit_min_tiles") g_gemm_f32_ksplit_min_tiles = std::max(1, iv);      // ... only from that many 64 x 64 output tiles on
    else if (n == "f32_no_ksplit") g_gemm_f32_ksplit = !on;            // ... few-row products as one 64-tile launch
    else if (n == "f32_gemm_small_v1") g_gemm_f32_v2_small = !on;      // ... the round-3 64-tile kernel for every 64 x 64-tiled product
    else if (n == "f32_gemm_w4") g_gemm_f32_w8 = !on;                  // ... without the 8-wave tiles (process-wide, A/B only)
    else if (n == "f32_gemm_v1") g_gemm_f32_v2 = !on;                  // float32 GEMMs with the round-3 k loop (process-wide switch, A/B only)
    else if (n == "no_f32_side") m->allow_f32_side = !on;             // float32 step on one stream (no side-stream weight gradients, no deferred decoder update)
    else if (n == "f32_dw_last") m->f32_dw_last = iv;                 // float32 step: all decoder weight gradients behind the dX chain (1: tiles as picked, 2: 4-wave tiles, 3: ... at 3 waves per SIMD)
    else if (n == "f32_wout_last") m->f32_wout_first = !on;           // ... with the output layer's gradient last on the side stream (beside the encoder's few-row kernels) instead of first (beside the dX chain)
    else if (n == "no_f32_multi_reduce") m->allow_f32_multi_reduce = !on;      // float32 mode: a slab reduction launch per gradient tensor instead of one per step
    else if (n == "f32_dec_fused_train") m->f32_dec_fused_train = on;   // float32 training step: the decoder forward as dec_fwd_f32_kernel (round 4) instead of three GEMM launches
    else if (n == "no_f32_dec_fused") m->allow_f32_dec_fused = !on;   // float32 mode: the decoder forward as three GEMM launches
    else if (n == "no_f32_bern_fused") m->allow_f32_bern_fused = !on; // float32 mode: logits to memory, bern_f32_kernel / dl_f32_kernel as their own passes
    else if (n == "no_dec_rows") m->allow_dec_rows = !on;             // few data rows: the decoder's weight gradients as the grouped launch on the side stream + deferred reduction
    else if (n == "no_wgrad_rows") m->allow_wgrad_rows = !on;         // few rows: the encoder's weight gradients as the grouped launch + slabs + reduce_grads_kernel
    else if (n == "no_wg3") m->allow_wg3 = !on;                       // few rows: the decoder's weight gradients as three launches on two streams
    else if (n == "g2w") m->allow_g2w = on;                           // the decoder kernel leaves bf16(g_r g2); the output layer's weight gradient runs unweighted on it (measured slower)
    else if (n == "lat_rows4") m->allow_lat_rows4 = on;               // many samples per image: latent_bwd_kernel's sums inside block_bwd_kernel<4> (measured no faster)
    else if (n == "no_lat_in_block") m->allow_lat_in_block = !on;     // few images: latent_bwd_kernel as its own launch in front of the encoder's backward pass
    else if (n == "no_lse_in_bwd") m->allow_lse_in_bwd = !on;         // few rows: lse_kernel as its own launch between decoder forward and backward
    else if (n == "no_lse_fused") m->allow_lse_fused = !on;           // lse_kernel as its own launch behind the decoder kernel
    else if (n == "no_lse_dup") m->allow_lse_dup = !on;               // one lse_kernel, the side stream forks behind it
    else if (n == "dz_f32") m->allow_dz_half = !on;                   // dec_bwd_kernel leaves dz as float32
    else if (n == "no_small_dec_bwd") m->small_dec_bwd = !on;         // per-pixel-group out_bwd + finish + two dX launches below 8 192 rows
    else if (n == "small_rows") m->small_rows = iv;
    else if (n == "dec_rows") m->dec_rows_max = iv;                   // dec_bwd_rows_kernel up to this many rows
    else if (n == "no_wg7") m->allow_wg7 = !on;                       // the 16-wave weight-gradient shapes everywhere
    else if (n == "wg9") m->wg_shape9 = iv;                           // bit mask: layers that take the 8 + 8-wave / 128-feature wgradws shape
    else if (n == "eval_rows") m->eval_rows = iv > 0 ? std::max(64, iv) : 0;       // data rows per evaluator launch
    else if (n == "no_bern_pipe") m->allow_bern_pipe = !on;           // the Bernoulli forward on dense_kernel<EPI_BERN>
    else if (n == "no_block_fused") m->allow_block_fused = !on;       // a BasicBlock on few rows as three dense_kernel launches
    else if (n == "no_out_in_block") m->allow_out_in_block = !on;     // the few-row decoder's output layer as its own launch
    else if (n == "no_dec_fused") m->allow_dec_fused = !on;           // the decoder's tanh layers as dense_kernel launches
    else if (n == "no_bern_qw") m->bern_qw = !on;                     // the decoder kernel's 8-wave / 128-row shape
    else if (n == "bern_qw_force") m->bern_qw_force = on;             // the 16-wave / 200-row shape at every row count
    else if (n == "dense_g1") m->dense_g1_mask = (unsigned)iv;        // EPI bit mask of the 8-wave x 16-row dense shape
    else if (n == "wg8") m->wg_target8 = std::max(1, iv);             // workgroup targets of the weight-gradient launches
    else if (n == "wg8_few") m->wg_target8_few = std::max(1, iv);
    else if (n == "wg16") m->wg_target16 = std::max(1, iv);
    else if (n == "wg16_1") m->wg_target16_1 = std::max(1, iv);
    else if (n == "eps_blocks") m->eps_blocks = std::max(0, iv);      // blocks of the ahead-of-time noise draw
    else if (n == "dec_bwd_nw") m->dec_bwd_nw = iv == 8 ? 8 : 4;
    else if (n == "no_side2") m->use_side2 = !on;                     // the hidden layers' weight gradients behind the output layer's
    else if (n == "wg_group") m->allow_wg_group = on;                 // ... as one grouped launch
    else if (n == "no_early_wout") m->allow_early_wout = !on;         // the output layer's weight gradient forks behind out_bwd
    else if (n == "dp_concurrent") m->dp_concurrent = on;             // data-parallel step: no device-side order between its two all-reduces
#ifdef IWAE_DIAG
    // diagnostic builds only (DIAG=1 ./build.sh): in-kernel phase stamps and the weight-gradient ablations -- results are wrong or slower
    else if (n == "stamps") { m->want_stamps = on; if (on) { m->allow_s_mode = false; m->allow_bern_pipe = false; m->allow_block_fused = false; m->allow_dec_fused = false; } }
    else if (n == "dense_stamps_epi") m->dstamp_epi = iv;
    else if (n == "dense_stamps_kt") m->dstamp_kt = iv;
    else if (n == "wg_debug") m->wg_debug = iv;
    else if (n == "abl_skip") m->abl_skip = iv;      // launch ablations of the full-size step (timing only)
    else if (n == "fake_s") m->fake_s = iv;          // byte ablations (timing only): 1 dec_bwd_kernel reads s from 32 rows, 2 the output layer's gradient likewise, 4 the decoder kernel does not store s, 8 ... nor z / g1 / g2
#endif
    else return fail(IWAE_ERR_ARG, "set_option: unknown option '" + n + "'");
    return IWAE_OK;
}

int iwae_set_eval_precision(iwae_handle m, int32_t precision) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    if (precision != IWAE_PREC_BF16 && precision != IWAE_PREC_FP32) return fail(IWAE_ERR_ARG, "precision must be IWAE_PREC_BF16 or IWAE_PREC_FP32");
    m->eval_precision = precision;
    return IWAE_OK;
}

int iwae_set_adam(iwae_handle m, float beta1, float beta2, float epsilon) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(epsilon > 0.f)) return fail(IWAE_ERR_ARG, "set_adam: need 0 <= beta < 1, epsilon > 0");
    m->adam_b1 = beta1; m->adam_b2 = beta2; m->adam_eps = epsilon;
    return IWAE_OK;
}

int iwae_train_step(iwae_handle m, const float* x, int32_t B, int32_t k, float beta, float lr, int32_t objective, const float* eps,
                    iwae_scalars* scalars, const iwae_tensors* want) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(check_objective(m, objective));
    if (m->cfg.precision == IWAE_PREC_FP32) {   // float32 mode: forward, closed-form backward, [exchange,] Adam -- all in float32
        CHK(forward_f32(m, x, B, k, beta, eps, objective, true, want));
        if (m->comm_main || want) {
            CHK(backward_f32(m, objective));
            if (want) CHK(fetch_outputs(m, nullptr, want));      // tensors refer to the pre-update forward (src/iwae1.py:162)
            if (m->comm_main) CHK(dp_finish(m, lr));
            else { CHK(join_side(m)); CHK(adam_impl(m, lr, 1.0f)); }
        } else {
            CHK(backward_f32(m, objective, lr));   // the update rides behind the gradients, the decoder's on the side stream (deferred)
        }
        CHK(fetch_outputs(m, scalars, nullptr));
        m->noise_step += 1;
        return IWAE_OK;
    }
    CHK(forward_impl(m, x, B, k, beta, eps, objective, true, want));
    if (m->comm_main) {                         // data-parallel step: exchange between gradient and update (iwae_comm_init)
        CHK(backward_impl(m, objective, -1.0f, true, true));
        if (want) CHK(fetch_outputs(m, nullptr, want));
        CHK(dp_finish(m, lr));
    } else if (want) {
        CHK(backward_impl(m, objective));
        CHK(fetch_outputs(m, nullptr, want));   // tensors refer to the pre-update forward (src/iwae1.py:162)
        CHK(adam_impl(m, lr, 1.0f));
    } else {
        CHK(backward_impl(m, objective, lr));   // Adam fused into the gradient reduction
    }
    CHK(fetch_outputs(m, scalars, nullptr));
    m->noise_step += 1;
    return IWAE_OK;
}

int iwae_comm_unique_id(void* id_out, size_t cap, size_t* id_bytes) {
    if (!id_out || !id_bytes || cap < 2 * sizeof(ncclUniqueId)) return fail(IWAE_ERR_ARG, "comm_unique_id: need a buffer of >= 256 bytes");
    CHK(load_rccl());
    ncclUniqueId ids[2];
    NCCLCHK(g_rccl.GetUniqueId(&ids[0]));
    NCCLCHK(g_rccl.GetUniqueId(&ids[1]));
    memcpy(id_out, ids, sizeof(ids));
    *id_bytes = sizeof(ids);
    return IWAE_OK;
}

// Everything iwae_comm_init can refuse WITHOUT talking to another rank: arguments, handle state, RCCL loadable.  ncclCommInitRank is itself
// a blocking rendezvous: a rank that fails one of these checks must not leave the others inside it, so callers agree on the preflight's
// outcome first (iwae_amd/parallel.py::init_in_library_exchange) and only then enter iwae_comm_init together.
int iwae_comm_preflight(iwae_handle m, const void* unique_id, size_t id_bytes, int32_t world_size, int32_t rank) {
    if (!m || !unique_id || id_bytes != 2 * sizeof(ncclUniqueId)) return fail(IWAE_ERR_ARG, "comm_init: bad id blob (iwae_comm_unique_id makes it)");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(IWAE_ERR_ARG, "comm_init: need 0 <= rank < world_size");
    if (world_size != m->cfg.world_size || rank != m->cfg.rank)
        return fail(IWAE_ERR_ARG, "comm_init: world_size / rank differ from the iwae_config this handle was created with");
    if (m->comm_main) return fail(IWAE_ERR_STATE, "comm_init: communicators already exist (iwae_comm_destroy first)");
    CHK(load_rccl());
    return IWAE_OK;
}

int iwae_comm_init(iwae_handle m, const void* unique_id, size_t id_bytes, int32_t world_size, int32_t rank) {
    CHK(iwae_comm_preflight(m, unique_id, id_bytes, world_size, rank));
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(join_side(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    ncclUniqueId ids[2];
    memcpy(ids, unique_id, sizeof(ids));
    // both communicators or neither: a half-initialised pair would send the next train step into ncclAllReduce with a null
    // communicator, and could not be retried ("communicators already exist")
    if (!m->ev_ar) HIPCHK(hipEventCreateWithFlags(&m->ev_ar, hipEventDisableTiming));
    ncclComm_t cm = nullptr, cs = nullptr;
    ncclResult_t r1 = g_rccl.CommInitRank(&cm, world_size, ids[0], rank);
    ncclResult_t r2 = r1 == ncclSuccess ? g_rccl.CommInitRank(&cs, world_size, ids[1], rank) : r1;
    if (r1 != ncclSuccess || r2 != ncclSuccess) {
        if (r1 == ncclSuccess && cm) (void)g_rccl.CommDestroy(cm);
        return fail(IWAE_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r1 != ncclSuccess ? r1 : r2));
    }
    m->comm_main = cm; m->comm_side = cs;
    m->comm_world = world_size; m->comm_rank = rank;
    return IWAE_OK;
}

int iwae_comm_destroy(iwae_handle m) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    CHK(join_side(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->side) HIPCHK(hipStreamSynchronize(m->side));
    if (m->side2) HIPCHK(hipStreamSynchronize(m->side2));      // `tail` (the decoder's exchange + update) may be either side stream
    if (m->comm_main) { NCCLCHK(g_rccl.CommDestroy(m->comm_main)); m->comm_main = nullptr; }
    if (m->comm_side) { NCCLCHK(g_rccl.CommDestroy(m->comm_side)); m->comm_side = nullptr; }
    m->comm_world = 1; m->comm_rank = 0;
    return IWAE_OK;
}

int iwae_comm_info(iwae_handle m, int32_t* world_size, int32_t* rank) {
    if (!m || !world_size || !rank) return fail(IWAE_ERR_ARG, "comm_info: null argument");
    *world_size = 0; *rank = -1;
    if (!m->comm_main) return IWAE_OK;            // no communicator: the handle trains alone
    int n = 0, r = -1, n2 = 0;
    NCCLCHK(g_rccl.CommCount(m->comm_main, &n));
    NCCLCHK(g_rccl.CommUserRank(m->comm_main, &r));
    NCCLCHK(g_rccl.CommCount(m->comm_side, &n2));
    if (n2 != n) return fail(IWAE_ERR_STATE, "comm_info: the two communicators disagree about the world size");
    *world_size = n; *rank = r;
    return IWAE_OK;
}

int iwae_set_condition(iwae_handle m, const float* y, int32_t n) {
    if (!m || !y || n <= 0) return fail(IWAE_ERR_ARG, "set_condition: bad argument");
    if (m->C <= 0) return fail(IWAE_ERR_STATE, "set_condition: the model was created with cond_dim = 0");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(copy_in(m, m->cond, y, (size_t)n * m->C * 4));
    HIPCHK(hipStreamSynchronize(m->stream));       // y may be a temporary of the caller
    m->cond_n = n;
    return IWAE_OK;
}

int iwae_set_step(iwae_handle m, uint32_t noise_step, uint32_t batch_offset) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    m->noise_step = noise_step;
    m->batch_offset = batch_offset;
    return IWAE_OK;
}

int iwae_eval_llh(iwae_handle m, const float* x, int32_t N, int32_t k, int32_t chunk, double* llh, float* per_image) {
    if (!m || !x || N <= 0 || k <= 0 || !llh) return fail(IWAE_ERR_ARG, "eval_llh: bad argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    // Launches of at most eval_rows data rows: `chunk` images x kc samples.  k <= eval_rows: whole images (kc = k).  Larger k: the
    // samples of an image are walked in chunks of kc and the per-chunk log-mean-exps are merged with a running log-sum-exp
    // (src/utils.py:6-8 is associative in that form) -- the activations of ALL k samples never exist at once.
    // (2^21 rows per launch at the reference's hidden width -- the single-launch decoder kernels keep nothing per pixel, ~0.4 KiB of HBM per row, and
    // the per-launch costs (image encoder on ~100 images, log-mean-exp, launch boundaries) are a quarter of what they are at 2^19: bf16 186 -> 205 k
    // images/s, float32 35.0 -> 36.7 k; other shapes, whose fallback paths may keep float32 logits, stay at 2^19)
    // (advisor, round 4: the large cap only where the single-launch decoder really runs for the evaluator's precision -- 1-layer model, the reference's
    // hidden width, and neither of the fused paths switched off; the 2-layer model and the unfused float32 path keep per-row tensors: 2^19)
    const bool eval_f32 = m->eval_precision == IWAE_PREC_FP32;
    const bool one_launch_dec = m->cfg.n_layers == 1 && m->dec1[2].KT == 7 && m->C == 0 && !m->has_prior &&
                                (eval_f32 ? (m->allow_f32_dec_fused && m->allow_f32_bern_fused) : (m->allow_bern_pipe && m->allow_dec_fused));
    const int eval_rows = m->eval_rows > 0 ? m->eval_rows : (one_launch_dec ? 1 << 21 : 1 << 19);
    const int kc = std::min(k, eval_rows);
    if (chunk <= 0) chunk = std::max(1, eval_rows / kc);
    chunk = std::min(chunk, N);
    // Round 4: no host round trip per launch.  The images go to the device once, every launch leaves its per-image log-mean-exps in one device
    // array [k-chunks][N], and ONE copy + synchronisation at the end feeds the same host arithmetic in the same order (the results are bitwise
    // what the per-launch copies gave); the k = 5000 evaluator's launches used to sit ~50 us apart (10 % of the bf16 evaluator's time).
    const int ns = (k + kc - 1) / kc;
    const float* xd = x;
    const size_t xbytes = (size_t)N * m->X * 4;
    if (!is_device_ptr(x, m->cfg.device) && xbytes <= ((size_t)1 << 31)) { CHK(copy_in(m, m->eval_x, x, xbytes)); xd = ptr<float>(m->eval_x); }
    CHK(ensure(m->eval_lme, (size_t)ns * N * 4, m->stream));
    const uint32_t saved_off = m->batch_offset;
    int rc = IWAE_OK;
    for (int i0 = 0; i0 < N && rc == IWAE_OK; i0 += chunk) {
        const int nb = std::min(chunk, N - i0);
        m->batch_offset = saved_off + (uint32_t)i0;
        m->cond_row0 = i0;
        for (int si = 0; si < ns && rc == IWAE_OK; ++si) {
            const int s0 = si * kc, kn = std::min(kc, k - s0);
            m->eval_k_total = (kc < k) ? k : 0;
            m->eval_s_off = s0;
            const bool f32 = m->eval_precision == IWAE_PREC_FP32;
            m->in_eval_llh = true;
            rc = f32 ? forward_f32(m, xd + (size_t)i0 * m->X, nb, kn, 1.0f, nullptr, OBJ_IWAE_ELBO, false, nullptr)
                     : forward_impl(m, xd + (size_t)i0 * m->X, nb, kn, 1.0f, nullptr, OBJ_IWAE_ELBO, false, nullptr);
            m->eval_k_total = 0; m->eval_s_off = 0; m->in_eval_llh = false;
            if (rc != IWAE_OK) break;
            if (hipMemcpyAsync(ptr<float>(m->eval_lme) + (size_t)si * N + i0, ptr<float>(m->per_b) + (size_t)PB_LME * nb, (size_t)nb * 4, hipMemcpyDeviceToDevice,
                               m->stream) != hipSuccess) { rc = fail(IWAE_ERR_HIP, "eval_llh: keeping the per-image estimates failed"); break; }
        }
        m->cond_row0 = 0;
    }
    m->batch_offset = saved_off;
    if (rc != IWAE_OK) { (void)hipStreamSynchronize(m->stream); return rc; }
    std::vector<float> lme((size_t)ns * N);
    if (hipMemcpyAsync(lme.data(), m->eval_lme.p, lme.size() * 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess)
        return fail(IWAE_ERR_HIP, "eval_llh: copying the per-image estimates failed");
    double total = 0.0;
    for (int i = 0; i < N; ++i) {
        double run = 0.0;
        for (int si = 0; si < ns; ++si) {
            const int kn = std::min(kc, k - si * kc);
            const double part = (double)lme[(size_t)si * N + i] + log((double)kn);       // log sum_s exp(log_w) over this chunk of samples
            if (si == 0) run = part;
            else { const double hi = std::max(run, part), lo = std::min(run, part); run = hi + log1p(exp(lo - hi)); }
        }
        const double v = run - log((double)k);
        total += v;                    // MyMetric: sum / count (src/utils.py:39-41)
        if (per_image) per_image[i] = (float)v;
    }
    m->noise_step += 1;
    *llh = total / (double)N;
    return IWAE_OK;
}

int iwae_decode(iwae_handle m, const float* z, int32_t n, float* probs) {
    if (!m || !z || !probs || n <= 0) return fail(IWAE_ERR_ARG, "decode: bad argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(join_side(m));
    hipStream_t st = m->stream;
    const bool two = m->cfg.n_layers == 2;
    const int np = round_up(n, 128), D0 = m->D[0], Dp0 = m->Dp[0], Hp = m->dec1[0].Np32, Xp = m->Xp32;
    MlpWs& w = m->wdec1;
    const int Din = two ? m->D[1] : D0, Dinp = two ? m->Dp[1] : Dp0;       // the caller's z is the LAST latent (z for 1 layer, z2 for 2)
    CHK(copy_in(m, m->xin, z, (size_t)n * Din * 4));
    CHK(ensure(m->zP[0], (size_t)np * Dp0 * 2, st));
    CHK(ensure(w.g1P, (size_t)np * Hp * 2, st));
    CHK(ensure(w.g2P, (size_t)np * Hp * 2, st));
    CHK(ensure(m->scratch, (size_t)np * Xp * 4, st));
    if (two) {
        // src/iwae2.py:184-196: pz1z2 = decode_z2_to_z1(z2); z1 = pz1z2.sample(); logits = decode_z1_to_x(z1)
        CHK(ensure(m->zP[1], (size_t)np * Dinp * 2, st));
        launch_prep_rows(ptr<float>(m->xin), nullptr, n, Din, 0, Dinp, np, ptr<uint16_t>(m->zP[1]), st);
        CHK(block_alloc(m, m->dec2, m->wdec2, n, np, false, false));
        CHK(block_fwd(m, m->dec2, m->wdec2, ptr<uint16_t>(m->zP[1]), n));
        for (int i = 0; i < 2; ++i) CHK(ensure(m->rows[i], (size_t)np * 4, st));
        SampleArgs s;
        memset(&s, 0, sizeof(s));
        s.head = ptr<float>(m->wdec2.head); s.ldH = 2 * Dp0; s.Dp = Dp0; s.D = D0; s.head_per_row = 1;
        s.M = n; s.Mp = np; s.k = 1; s.B = n;
        s.eps.user = nullptr; s.eps.B = n; s.eps.seed = m->cfg.seed; s.eps.row_offset = 0; s.eps.step = m->noise_step; s.eps.stream = 2;
        s.ZP = ptr<uint16_t>(m->zP[0]);
        s.lp_prior = ptr<float>(m->rows[0]); s.lq = ptr<float>(m->rows[1]); s.lq_dreg = nullptr;
        launch_sample(s, st);
        m->noise_step += 1;
    } else {
        if (m->C > 0 && n > m->cond_n) return fail(IWAE_ERR_STATE, "conditional model: call iwae_set_condition with y for these rows first");
        if (m->has_prior) {
            // tasks/task04.py:190-196: z_new = pzy.loc + pzy.scale * z with pzy the conditional prior of the label rows
            const int Cp = round_up(m->C, 32);
            CHK(ensure(m->condP, (size_t)np * Cp * 2, st));
            launch_prep_rows(ptr<float>(m->cond), nullptr, n, m->C, 0, Cp, np, ptr<uint16_t>(m->condP), st);
            CHK(block_alloc(m, m->prior, m->wprior, n, np, false, false));
            CHK(block_fwd(m, m->prior, m->wprior, ptr<uint16_t>(m->condP), n));
            for (int i = 0; i < 2; ++i) CHK(ensure(m->rows[i], (size_t)np * 4, st));
            SampleArgs s;
            memset(&s, 0, sizeof(s));
            s.head = ptr<float>(m->wprior.head); s.ldH = 2 * Dp0; s.Dp = Dp0; s.D = D0; s.head_per_row = 1;
            s.M = n; s.Mp = np; s.k = 1; s.B = n;
            s.eps.user = ptr<float>(m->xin); s.eps.B = n;          // the caller's z plays the role of the N(0,1) draw
            s.ZP = ptr<uint16_t>(m->zP[0]);
            s.cond = ptr<float>(m->cond); s.C = m->C;
            s.lp_prior = ptr<float>(m->rows[0]); s.lq = ptr<float>(m->rows[1]); s.lq_dreg = nullptr;
            launch_sample(s, st);
        } else {
            launch_prep_rows(ptr<float>(m->xin), m->C > 0 ? ptr<float>(m->cond) : nullptr, n, D0, m->C, Dp0, np, ptr<uint16_t>(m->zP[0]), st);
        }
    }
    CHK(dense_fwd(m, m->dec1[0], EPI_TANH, ptr<uint16_t>(m->zP[0]), n, ptr<uint16_t>(w.g1P), nullptr, 0));
    CHK(dense_fwd(m, m->dec1[1], EPI_TANH, ptr<uint16_t>(w.g1P), n, ptr<uint16_t>(w.g2P), nullptr, 0));
    CHK(dense_fwd(m, m->dec1[2], EPI_SIGMOID, ptr<uint16_t>(w.g2P), n, nullptr, ptr<float>(m->scratch), Xp));
    HIPCHK(hipMemcpy2DAsync(probs, (size_t)m->X * 4, m->scratch.p, (size_t)Xp * 4, (size_t)m->X * 4, n, hipMemcpyDefault, st));
    HIPCHK(hipStreamSynchronize(st));
    m->have_forward = false;
    return IWAE_OK;
}

int iwae_dataset_upload(iwae_handle m, const uint8_t* gray, int32_t n) {
    if (!m || !gray || n <= 0) return fail(IWAE_ERR_ARG, "dataset_upload: bad argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(ensure(m->ds_data, (size_t)n * m->X, m->stream));
    CHK(ensure(m->ds_order, (size_t)n * 4, m->stream));
    HIPCHK(hipMemcpyAsync(m->ds_data.p, gray, (size_t)n * m->X, hipMemcpyDefault, m->stream));
    std::vector<int32_t> ident(n);
    for (int i = 0; i < n; ++i) ident[i] = i;
    HIPCHK(hipMemcpyAsync(m->ds_order.p, ident.data(), (size_t)n * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->ds_N = n;
    m->ds_epoch = 0;
    m->ds_has_labels = false;      // (a new set: its labels, if any, follow)
    return IWAE_OK;
}

int iwae_dataset_set_labels(iwae_handle m, const uint8_t* labels, int32_t n) {
    if (!m || !labels) return fail(IWAE_ERR_ARG, "dataset_set_labels: null argument");
    if (m->C <= 0) return fail(IWAE_ERR_STATE, "dataset_set_labels: the model was created with cond_dim = 0");
    if (m->ds_N <= 0) return fail(IWAE_ERR_STATE, "dataset_set_labels: no dataset uploaded");
    if (n != m->ds_N) return fail(IWAE_ERR_ARG, "dataset_set_labels: one label per image of the uploaded set");
    HIPCHK(hipSetDevice(m->cfg.device));
    if (!is_device_ptr(labels, m->cfg.device))
        for (int i = 0; i < n; ++i)
            if ((int)labels[i] >= m->C) return fail(IWAE_ERR_ARG, "dataset_set_labels: label " + std::to_string((int)labels[i]) + " at " + std::to_string(i) + " is not below cond_dim");
    CHK(ensure(m->ds_labels, (size_t)n, m->stream));
    HIPCHK(hipMemcpyAsync(m->ds_labels.p, labels, (size_t)n, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->ds_has_labels = true;
    return IWAE_OK;
}

int iwae_dataset_begin_epoch(iwae_handle m, uint32_t epoch, const int32_t* order, int32_t n) {
    if (!m || m->ds_N <= 0) return fail(IWAE_ERR_STATE, "dataset_begin_epoch: no dataset uploaded");
    if (order) {
        if (n != m->ds_N) return fail(IWAE_ERR_ARG, "dataset_begin_epoch: order must have one entry per image");
        for (int i = 0; i < n; ++i)
            if (order[i] < 0 || order[i] >= m->ds_N) return fail(IWAE_ERR_ARG, "dataset_begin_epoch: index out of range");
        HIPCHK(hipMemcpyAsync(m->ds_order.p, order, (size_t)n * 4, hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    m->ds_epoch = epoch;
    return IWAE_OK;
}

int iwae_dataset_get_batch(iwae_handle m, int32_t start, int32_t B, float* x_out) {
    if (!m || !x_out || m->ds_N <= 0 || start < 0 || B <= 0 || start + B > m->ds_N) return fail(IWAE_ERR_ARG, "dataset_get_batch: bad range");
    HIPCHK(hipSetDevice(m->cfg.device));
    const int Bp = round_up(B, 128);
    CHK(ensure(m->xP, (size_t)Bp * m->Xinp * 2, m->stream));
    CHK(ensure(m->scratch, (size_t)B * m->X * 4, m->stream));
    launch_gather_binarize(ptr<uint8_t>(m->ds_data), ptr<int32_t>(m->ds_order), start, m->ds_N, B, m->X, m->Xinp, Bp, m->cfg.seed, m->ds_epoch,
                           ptr<uint16_t>(m->xP), ptr<float>(m->scratch), m->stream);
    HIPCHK(hipMemcpyAsync(x_out, m->scratch.p, (size_t)B * m->X * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->have_forward = false;
    return IWAE_OK;
}

int iwae_dataset_get_labels(iwae_handle m, int32_t start, int32_t B, float* y_out) {
    if (!m || !y_out || m->ds_N <= 0 || start < 0 || B <= 0 || start + B > m->ds_N) return fail(IWAE_ERR_ARG, "dataset_get_labels: bad range");
    if (m->C <= 0 || !m->ds_has_labels) return fail(IWAE_ERR_STATE, "dataset_get_labels: no labels (iwae_dataset_set_labels on a conditional model)");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(join_side(m));
    const int Bp = round_up(B, 128);
    CHK(ensure(m->xP, (size_t)Bp * m->Xinp * 2, m->stream));
    CHK(ensure(m->cond, (size_t)B * m->C * 4, m->stream));
    launch_gather_binarize(ptr<uint8_t>(m->ds_data), ptr<int32_t>(m->ds_order), start, m->ds_N, B, m->X, m->Xinp, Bp, m->cfg.seed, m->ds_epoch,
                           ptr<uint16_t>(m->xP), nullptr, m->stream, ptr<uint8_t>(m->ds_labels), m->C, ptr<float>(m->cond));
    HIPCHK(hipMemcpyAsync(y_out, m->cond.p, (size_t)B * m->C * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    m->cond_n = 0;                 // (the buffer no longer holds what iwae_set_condition put there)
    m->have_forward = false;
    return IWAE_OK;
}

int iwae_train_step_dataset(iwae_handle m, int32_t start, int32_t B, int32_t k, float beta, float lr, int32_t objective, iwae_scalars* scalars) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    if (m->ds_N <= 0) return fail(IWAE_ERR_STATE, "train_step_dataset: no dataset uploaded");
    if (start < 0 || B <= 0 || start + B > m->ds_N) return fail(IWAE_ERR_ARG, "train_step_dataset: batch range outside the dataset");
    m->ds_start = start;
    int rc = iwae_train_step(m, nullptr, B, k, beta, lr, objective, nullptr, scalars, nullptr);
    m->ds_start = -1;
    return rc;
}

int iwae_enable_timing(iwae_handle m, int32_t enable) {
    if (!m) return fail(IWAE_ERR_ARG, "null handle");
    HIPCHK(hipStreamSynchronize(m->stream));
    m->timing = enable > 0 ? enable : 0;
    m->timing_calls = 0;
    m->time_this = false;
    for (int i = 0; i < T_COUNT; ++i) m->ev_used[i] = 0;
    return IWAE_OK;
}

int iwae_kernel_time(iwae_handle m, const char* name, double* avg_us, int64_t* launches) {
    if (!m || !name || !avg_us) return fail(IWAE_ERR_ARG, "kernel_time: null argument");
    int id = -1;
    for (int i = 0; i < T_COUNT; ++i)
        if (!strcmp(name, kTimedNames[i])) id = i;
    if (!strcmp(name, "bernoulli_fwd")) id = T_DEC_FWD;      // round-1 name of the decoder forward kernel
    if (id < 0) {
        std::string all;
        for (int i = 0; i < T_COUNT; ++i) all += std::string(i ? " | " : "") + kTimedNames[i];
        return fail(IWAE_ERR_ARG, "kernel_time: unknown kernel (" + all + ")");
    }
    CHK(join_side(m));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (m->side) HIPCHK(hipStreamSynchronize(m->side));      // the weight gradients are timed on the side streams
    if (m->side2) HIPCHK(hipStreamSynchronize(m->side2));
    double tot = 0.0;
    for (size_t i = 0; i < m->ev_used[id]; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, m->ev_start[id][i], m->ev_stop[id][i]));
        tot += ms;
    }
    *avg_us = m->ev_used[id] ? tot * 1e3 / (double)m->ev_used[id] : 0.0;
    if (launches) *launches = (int64_t)m->ev_used[id];
    return IWAE_OK;
}

int iwae_debug_eps(iwae_handle m, int32_t B, int32_t k, int32_t layer, float* out) {
    if (!m || !out || B <= 0 || k <= 0 || layer < 0 || layer >= m->cfg.n_layers) return fail(IWAE_ERR_ARG, "debug_eps: bad argument");
    HIPCHK(hipSetDevice(m->cfg.device));
    const int D = m->D[layer];
    CHK(ensure(m->scratch, (size_t)B * k * D * 4, m->stream));
    EpsSrc e;
    e.user = nullptr; e.B = B; e.seed = m->cfg.seed; e.row_offset = (uint64_t)m->batch_offset * k; e.step = m->noise_step; e.stream = layer;
    launch_eps_dump(e, B, k, D, ptr<float>(m->scratch), m->stream);
    HIPCHK(hipMemcpyAsync(out, m->scratch.p, (size_t)B * k * D * 4, hipMemcpyDefault, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return IWAE_OK;
}

int iwae_debug_tensor(iwae_handle m, const char* name, float* out, size_t cap, int32_t* rows, int32_t* cols) {
    if (!m || !name) return fail(IWAE_ERR_ARG, "debug_tensor: null argument");
    if (!m->have_forward) return fail(IWAE_ERR_STATE, "debug_tensor: no forward pass yet");
    HIPCHK(hipSetDevice(m->cfg.device));
    CHK(join_side(m));
    struct Ent { const char* nm; int kind; const DevBuf* buf; int R; int F; int Fp; };   // kind 0: bf16 P-layout, 2: fp32
    const int B = m->B, M = m->M, Mp = m->Mp;
    const int H0 = m->H[0], Hp0 = m->Hp[0], D0 = m->D[0], Dp0 = m->Dp[0];
    std::vector<Ent> ents = {
        {"x", 0, &m->xP, B, m->X, m->Xinp},
        {"enc.h1", 0, &m->wenc1.h1P, B, H0, Hp0},
        {"enc.h2", 0, &m->wenc1.h2P, B, H0, Hp0},
        {"enc.head", 2, &m->wenc1.head, B, 2 * Dp0, 2 * Dp0},
        {"enc.dhead", 0, &m->wenc1.dheadP, B, 2 * Dp0, 2 * Dp0},
        {"enc.d2", 0, &m->wenc1.d2P, B, H0, Hp0}, {"enc.d1", 0, &m->wenc1.d1P, B, H0, Hp0},
        {"z", 0, &m->zP[0], M, D0, Dp0},
        {"dec.g1", 0, &m->wdec1.g1P, M, H0, Hp0},
        {"dec.g2", 0, &m->wdec1.g2P, M, H0, Hp0},
        {"dec.dl", 0, &m->wdec1.dlP, M, m->X, m->Xp32},
        {"dec.d2", 0, &m->wdec1.d2P, M, H0, Hp0},
        {"dec.d1", 0, &m->wdec1.d1P, M, H0, Hp0},
        {"dec.dz", 2, &m->wdec1.dz, M, Dp0, Dp0},
        {"gx", 2, &m->gx, M, 1, 1}, {"wn", 2, &m->wn, M, 1, 1}, {"log_w", 2, &m->logw, M, 1, 1},
    };
    if (m->cfg.n_layers == 2) {
        const int H1 = m->H[1], Hp1 = m->Hp[1], D1 = m->D[1], Dp1 = m->Dp[1];
        std::vector<Ent> e2 = {
            {"enc2.h1", 0, &m->wenc2.h1P, M, H1, Hp1}, {"enc2.h2", 0, &m->wenc2.h2P, M, H1, Hp1},
            {"enc2.head", 2, &m->wenc2.head, M, 2 * Dp1, 2 * Dp1}, {"enc2.dhead", 0, &m->wenc2.dheadP, M, 2 * Dp1, 2 * Dp1},
            {"enc2.dx", 2, &m->wenc2.dx, M, Dp0, Dp0},
            {"z2", 0, &m->zP[1], M, D1, Dp1},
            {"dec2.h1", 0, &m->wdec2.h1P, M, H1, Hp1}, {"dec2.h2", 0, &m->wdec2.h2P, M, H1, Hp1},
            {"dec2.head", 2, &m->wdec2.head, M, 2 * Dp0, 2 * Dp0}, {"dec2.dhead", 0, &m->wdec2.dheadP, M, 2 * Dp0, 2 * Dp0},
            {"dec2.dx", 2, &m->wdec2.dx, M, Dp1, Dp1},
            {"dz1_direct", 2, &m->dzdir, M, Dp0, Dp0},
        };
        ents.insert(ents.end(), e2.begin(), e2.end());
    }
    if (strcmp(name, "dense_stamps") == 0) {   // diagnostic: [waves][8] phase cycle sums of the selected dense launch
        if (rows) *rows = m->dstamp_waves;
        if (cols) *cols = 8;
        if (!out) return IWAE_OK;
        if (!m->dstamps.p) return fail(IWAE_ERR_STATE, "dense stamps not enabled (STAMPS=1 build + IWAE_DENSE_STAMPS=epi:KT)");
        std::vector<unsigned long long> h((size_t)m->dstamp_waves * 8);
        HIPCHK(hipStreamSynchronize(m->stream));
        HIPCHK(hipMemcpy(h.data(), m->dstamps.p, h.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) out[i] = (float)h[i];
        return IWAE_OK;
    }
    if (strcmp(name, "stamps") == 0) {   // diagnostic: [waves][8] phase cycle sums of out_bwd, as float
        const int nw = (Mp / 64) * 4;
        if (rows) *rows = nw;
        if (cols) *cols = 8;
        if (!out) return IWAE_OK;
        if (!m->stamps.p) return fail(IWAE_ERR_STATE, "stamps not enabled (IWAE_STAMPS=1)");
        std::vector<unsigned long long> h((size_t)nw * 8);
        HIPCHK(hipStreamSynchronize(m->stream));
        HIPCHK(hipMemcpy(h.data(), m->stamps.p, h.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); ++i) out[i] = (float)h[i];
        return IWAE_OK;
    }
    for (const Ent& e : ents) {
        if (strcmp(e.nm, name) != 0) continue;
        if (rows) *rows = e.R;
        if (cols) *cols = e.F;
        if (!out) return IWAE_OK;
        if (!e.buf->p) return fail(IWAE_ERR_STATE, std::string("debug_tensor: buffer not populated: ") + name);
        const size_t n = (size_t)e.R * e.F;
        if (cap < n) return fail(IWAE_ERR_ARG, "debug_tensor: output too small");
        if (e.kind == 2) {
            HIPCHK(hipMemcpyAsync(out, e.buf->p, n * 4, hipMemcpyDefault, m->stream));
        } else {
            CHK(ensure(m->scratch, n * 4, m->stream));
            launch_unpack_p(ptr<uint16_t>(*e.buf), e.R, e.F, e.Fp, ptr<float>(m->scratch), m->stream);
            HIPCHK(hipMemcpyAsync(out, m->scratch.p, n * 4, hipMemcpyDefault, m->stream));
        }
        HIPCHK(hipStreamSynchronize(m->stream));
        return IWAE_OK;
    }
    return fail(IWAE_ERR_ARG, std::string("debug_tensor: unknown name ") + name);
}

}  // extern "C"
