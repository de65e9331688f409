// Argument blocks and launch wrappers for kernels.hip (internal; the public ABI is include/iwae_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iwae {

#define IWAE_DENSE_G1_DEFAULT 29u     // TANH | DX | F32 | BERN (measured: 0.339 -> 0.328 ms/step at B=1024, k=50)
enum { EPI_TANH = 0, EPI_HEAD = 1, EPI_DX = 2, EPI_F32 = 3, EPI_BERN = 4, EPI_SIGMOID = 5 };
enum { OBJ_VAE_ELBO = 0, OBJ_IWAE_ELBO = 1, OBJ_IWAE_EQ14 = 2, OBJ_VAE_ELBO_KL = 3, OBJ_DREG = 4 };
// per-image reductions written by lse_kernel
enum { PB_LME = 0, PB_MEAN, PB_EQ14, PB_KL, PB_PX, PB_T1, PB_T2, PB_DREG, PB_COUNT };
// scalar outputs (device float[16]); must match iwae_scalars in include/iwae_amd.h
enum { SC_VAE_ELBO = 0, SC_VAE_ELBO_KL, SC_IWAE_ELBO, SC_IWAE_EQ14, SC_INFERENCE_LOSS, SC_MEAN_LPXZ, SC_MEAN_T1, SC_MEAN_T2, SC_KL, SC_COUNT = 16 };

struct EpsSrc {
    const float* user = nullptr;     // [k][B][D] host-supplied draws (reference order) or null -> Philox
    const float* cache = nullptr;    // [rows][ldC] draws eps_gen_kernel made ahead of the step (Philox costs ~40 quarter-rate
    int ldC = 0;                     //   integer multiplies per 4 draws; re-reading 16 B is cheaper than redrawing) or null
    int B = 0;
    uint64_t seed = 0;
    uint64_t row_offset = 0;   // global index of this rank's first data row (batch_offset * k)
    uint32_t step = 0;
    uint32_t stream = 0;       // 0: first latent layer, 1: second
    // k-chunked calls (iwae_eval_llh at large k): this call holds samples [s_off, s_off + kc) of k_total per image; the Philox row
    // index is that of the unchunked call, (image)*k_total + sample, so chunking does not change the draws.  k_total = 0: off.
    int k_total = 0, s_off = 0, kc = 0;
};

struct LseArgs {
    const float* term[5]; float coef[5];
    const float* lq_dreg;
    int B, k; float beta; int objective;
    float cz_on;                         // 1: prior term -z reaches dz (1-layer); 0: 2-layer (handled per row)
    const float* head; int ldH, D, Dp;   // for KL (1-layer) or null
    float* logw; float* wn; float* gx; float4* cf; float* per_b;
    int n_px_part; size_t px_stride;     // term[0] arrives as n partial sums [n][stride] (n <= 1: a plain array)
    float* term0_out;                    // [M] total of term[0] (== term[0] when n_px_part <= 1)
    int lme_only;                        // iwae_eval_llh's launches: only the per-image log-mean-exp and the means are wanted -- no softmax weights / gradient coefficients pass (28 B of stores per row)
    float* gx_local; int gx_r0, gx_n;    // optional: the row weights of rows [gx_r0, gx_r0 + gx_n) also go to gx_local[row - gx_r0] (kernels that do lse_image's work for their own rows: dec_bwd_rows_kernel, bern_pipe_kernel)
};

struct DenseArgs {
    const uint16_t* X; int ldX;       // P-layout rows [M][ldX], ldX = 32*KT
    const char* img;                  // MG-major A-image of the weight
    int split;                        // EPI_HEAD: out-features >= split are the sigma head (exp + 1e-6)
    int M, KT, MG, mg_per_block;
    int stage_all;                    // run-time K path: stage every window's weights up front (needs mg_per_block * windows <= 4)
    unsigned g1_mask;                 // EPI bit mask: launches (M >= 8192) that take the 8-wave x 16-row shape
    int Np32;                         // out-features that are stored (multiple of 32)
    uint16_t* YP; int ldYP;           // bf16 P-layout out
    float* YF; int ldYF;              // fp32 natural out
    const uint16_t* ACT; int ldACT;   // EPI_DX: stored activation of the out-features (P-layout)
    // EPI_BERN
    const uint16_t* XB; int ldXB; int k; int B; int Xdim;
    // sampled-input mode (dense_kernel<..., ZIN>): z = mu + sigma*eps made in place of reading X (X's layout: ldX = 32*KT)
    const float* zhead; int ldZH;     // encoder head per image [B][ldZH] (mu | sigma at zDp)
    const float* zeps; int zldE; int zD, zDp;   // the step's draws fp32 [rows][zldE]; latent width and its 32-padding (head layout, z rows)
    uint16_t* ZPout;                  // z as bf16 P-layout [M][ldX] (kept for the weight gradient)
    float* zlp; float* zlq;           // per-row log p(z), log q(z|x)
    float* zlq_dreg;                  // DReG (tasks/task02.py:63-65): log q(z|x) with sigma + 1e-6 as the scale, or null (bern_pipe_kernel's prologue only)
    float* lpxz; size_t lpxz_stride;  // log p(x|z) per row; stride > 0: block row y of the grid writes its partial sum to lpxz[y*stride + row]
    float* logits_out;
    int pipe;                         // EPI_BERN: take bern_pipe_kernel where it exists (IWAE_NO_BERN_PIPE=1 clears it)
    // bern_pipe_kernel<.., PRE>: the two tanh layers in front of the output layer run inside the same launch (pre_img1 != null)
    const char* pre_img1; int pre_KT1;   // first decoder layer: forward image, k-steps of its input (latent, <= 4)
    const char* pre_img2;                // second decoder layer (KT k-steps in and out)
    const uint16_t* pre_Z;               // z rows, P-layout [M][32*pre_KT1] (when zhead == null; else z is made in the kernel, see ZIN fields)
    uint16_t *pre_G1, *pre_G2;           // the layers' activations, P-layout [M][32*KT], kept for the backward pass (null: forward only, not stored)
    unsigned long long* stamps;       // diagnostic build (IWAE_DENSE_STAMPS) only: [blocks*4 waves][8] phase cycle sums, else null
    // bern_pipe_kernel<.., PRE, QW>, k a divisor of its 200 rows: the workgroup's rows are whole images, so what lse_kernel would do for
    // them (log_w, log-mean-exp, row weights, per-image values: `lse`) runs at the end of the workgroup -- no launch between the decoder
    // forward and the backward pass (lse.term[0] is ignored: log p(x|z) comes from the kernel's own sums)
    int lse_on;
    LseArgs lse;
    // ... and with the row weights in hand it leaves g2w = bf16(g_r * g2) [M][32*KT] (P-layout) for the output layer's weight gradient, which then
    // needs NO row weighting (dW3 = g2w^T s; feature g2w_feat -- a pad column of the hidden width -- carries g_r itself: its product row is the
    // bias gradient).  null: not wanted.
    uint16_t* G2W; int g2w_feat;
    int dbg;                          // DIAG builds only (option fake_s): 32 = the Bernoulli epilogue does not store s
};

struct SampleArgs {
    const float* head; int ldH; int Dp; int D; int head_per_row;
    int M, Mp, k, B;
    EpsSrc eps;
    uint16_t* ZP;                     // z as bf16 P-layout [M][Dp], or null
    float* ZF; int ldZF;              // z as float32 rows [M][ldZF] (float32 mode), or null
    const float* prior_head;          // conditional prior: per-image head [B][ldH] (mu_p | sigma_p) scoring z, or null = N(0,1)
    const float* cond; int C;         // conditional model: y [B][C] goes into features D..D+C-1 of the z rows (decoder input concat(z, y))
    float* lp_prior; float* lq; float* lq_dreg;
};

struct BlockFwdArgs {                 // block_fwd_kernel: one BasicBlock on R <= 4096 rows
    const uint16_t* X; int ldX;       // input rows, P-layout [R][32*KT0]
    const float* Xf; int Xdim; uint16_t* XPout;   // or (Xf != null) fp32 rows [R][Xdim], converted on the way in and kept in XPout (P-layout, ldX)
    const char *img0, *img1, *img2;   // MG-major forward images of l1 (KT0 k-steps), l2 and the head (KT1 k-steps each)
    int KT0, KT1;
    int NT1, NT2;                     // stored 16-feature tiles of the hidden layers (= 2*KT1) and of the head (<= 16 each)
    int R;
    uint16_t *H1, *H2; int ldH;       // hidden activations, P-layout [R][32*KT1]
    float* YF; int ldYF; int split;   // head, fp32 [R][ldYF]; out-features >= split are the sigma head
    int sample; SampleArgs S;         // sample != 0: the input rows are z = mu + sigma*eps made HERE (sample_kernel's job, without its launch): S.Dp = 32*KT0
    // oimg != null (decoder on few rows, NT2 = 0): the Bernoulli output layer follows in the same launch -- logits = h2 W3 + c3 from the MG-major
    // image oimg (KT1 k-steps), log p(x|z) per row -> olpxz, s = x - sigmoid(l) -> oSP (P-layout [R][oldS], or null: forward-only call)
    const char* oimg; int oXdim, oH;  // pixels, 32-pixel halves that hold real pixels
    const uint16_t* oXB; int oldXB; int ok;   // x as bf16 P-layout [B][oldXB]; samples per image (row r belongs to image r / ok)
    uint16_t* oSP; int oldS; float* olpxz;
};
struct DecBwdRowsArgs {               // dec_bwd_rows_kernel: the decoder's dX chain on few rows (16-row workgroups, weights straight from L2)
    const uint16_t* SP; int ldS; int KTX;     // s = x - sigmoid(l), P-layout [M][ldS], KTX = ldS/32 pixel k-steps
    const char* imgK3; int MT3;               // K-major image of the output layer's W (block (pixel k-step, hidden tile)), MT3 hidden tiles per k-step
    const char *imgB2, *imgB1;                // backward images of the second / first tanh layer (MG-major, KT k-steps each)
    int KT, NT1, NT3;                         // hidden k-steps, hidden tiles (= 2*KT), latent tiles of dz
    int M;
    const uint16_t *G2, *G1; int ldH;         // stored tanh activations, P-layout [M][32*KT]
    const float* gx;                          // [M] row weights
    uint16_t *D2P, *D1P;                      // dpre2, dpre1 out, P-layout [M][32*KT]
    float* DZ; uint16_t* DZH; int ldDZ;       // dz out: float32 or (DZH != null) bf16, [M][ldDZ]
    // lse_on: the workgroup first does lse_kernel's work (log_w, log-mean-exp over k, row weights, per-image values) for the images its 16 rows
    // belong to -- a wave per image, beside the first weight fragments' round trip -- and takes its rows' weights from LDS: no lse launch
    // between the decoder forward and this kernel (few rows: the step is a chain of dependent launches).  Images that straddle workgroups are
    // done by each of them (identical values).
    int lse_on;
    LseArgs lse;
};
bool dec_bwd_rows_ok(const DecBwdRowsArgs& a);
void launch_dec_bwd_rows(const DecBwdRowsArgs& a, hipStream_t st);

struct LatentBwdArgs {
    const float* dz; int ldDZ;
    const uint16_t* dzh;              // dz as bf16 [rows][ldDZ] instead of `dz` (dec_bwd_kernel's output), or null
    const float* dz2; const float* dz3;   // optional further terms of dz (same layout), added on load; both or neither
    const float* head; int ldH; int D, Dp;
    const float4* cf;
    EpsSrc eps;
    int B, Bp, k;
    float kmu, ksig;
    uint16_t* DHP;                    // dhead bf16, P-layout [B][2Dp] (or null)
    float* DHF;                       // float32 mode: dhead as float32 [B][2Dp] (d mu at f, d pre-exp at Dp + f), or null
    const float* prior_head;          // conditional prior p(z|y) (tasks/task04.py:124-130): per-image head [B][ldH] like `head`, or null = N(0,1)
    uint16_t* DHP2;                   // its dhead, P-layout [B][2Dp] (written when prior_head != null)
    float* DHF2;                      // float32 mode: the same as float32 [B][2Dp]
};
struct BlockBwdArgs {                 // block_bwd_kernel: the dX chain of a BasicBlock on R <= 4096 rows
    const uint16_t* DH; int ldDH;     // dhead, bf16 P-layout [R][32*KTH]
    const char *imgH, *imgL2;         // backward images (MG-major: out-feature groups over hidden) of the head (KTH k-steps) and of l2 (KT1)
    int KTH, KT1, NT1;                // NT1 = hidden tiles = 2*KT1
    int R;
    const uint16_t *H2, *H1; int ldH; // stored tanh activations, P-layout [R][32*KT1]
    uint16_t *D2, *D1;                // outputs: dpre of l2 and of l1, P-layout [R][32*KT1]
    // lat_on (the image encoder on few rows): the block's dhead rows are MADE here -- latent_bwd_kernel's per-image sums over the k samples
    // (d mu, d sigma: `lat`), a wave per image in front of the dX chain -- instead of being read from DH; they are also stored to lat.DHP
    // for the head's weight gradient.  One dependent launch less per step.
    int lat_on;
    LatentBwdArgs lat;
    int rows_per_wg;                  // 16, or 4 with lat_on (many samples per image: four waves per image, four times the workgroups)
};
bool block_bwd_ok(const BlockBwdArgs& a);
void launch_block_bwd(const BlockBwdArgs& a, hipStream_t st);
bool block_fwd_ok(const BlockFwdArgs& a);
void launch_block_fwd(const BlockFwdArgs& a, hipStream_t st);

struct OutBwdArgs {
    const uint16_t* G2; int ldG;      // last hidden activation, P-layout [M][32*KT]
    const char* img1;                 // MG-major image of W^T (out = pixels, k = hidden)
    const char* img2;                 // K-major image of W   (out = hidden, k = pixels)
    int Xdim; int Xp32;
    const float* gx;                  // [M] dLoss/dlpxz
    const uint16_t* XB; int ldXB; int k;
    int M, KT, NG;
    uint16_t* DLP;                    // dlogits, P-layout [M][Xp32] or null
    const uint16_t* SP;               // s = x - sigmoid(l) kept by the forward pass, P-layout [M][Xp32]: no recompute (else null)
    float* part; int gpb;             // SP mode, small row counts: block row y handles gpb pixel groups, fp32 partial dg2 in part[y][M][ldG]
    uint16_t* DPP;                    // dpre of the last hidden layer, P-layout [M][32*KT]
    unsigned long long* stamps;       // diagnostic build only: [blocks*4 waves][8] phase cycle sums, else null
    int dbg;                          // DIAG builds only (option fake_s; timing only -- results are wrong): 32 = s rows read from the first 32 rows (no HBM stream)
};

struct DecBwdArgs {                  // dec_bwd_kernel: out_bwd_s + dX of the two tanh layers in one launch
    OutBwdArgs o;                     // SP (s), G2, gx, img1 (W3^T image), M, KT, NG, Xp32, ldG, DPP (dpre2 out); part = null
    const char* imgB2;                // backward image of the second tanh layer ((KT+1)/2 64-out-feature groups over hidden, KT k-steps each)
    const uint16_t* G1;               // stored first-layer activation, P-layout [M][32*KT]
    uint16_t* D1P;                    // dpre1 out, P-layout [M][32*KT]
    const char* imgB1; int MG1;       // backward image of the first decoder layer (out = latent groups, k = hidden)
    float* DZ; int ldDZ;              // dz fp32 [M][ldDZ] (or null)
    uint16_t* DZH;                    // dz as bf16 [M][ldDZ], natural feature order (1-layer model: latent_bwd_kernel is its only reader), or null
    int nw;                           // 8: the 8-wave x 16-row shape (KT = 7 only), else 4 waves x 32 rows
};

struct WgradPArgs {
    const uint16_t* X; int ldX; int IT;     // layer input, P-layout [rows][ldX]; IT = ldX/16 i-tiles
    const uint16_t* G; int ldG; int JT;     // dpre of layer output, P-layout [rows][ldG]
    int M, rows_per_split;                  // valid rows; rows per split (multiple of 64)
    float* slabW;                           // [nsplit][IT*16][JT*16]
    float* slabB;                           // [nsplit][JT*16]
    const char* zero;                       // >= 512 B of zeros (source of rows >= M and of unused slots)
    const float* rowscale;                  // optional [M]: G rows are multiplied by it (rounded to bf16) on the way in, else null
    int dbg;                                // diagnostic ablations (DIAG build, option wg_debug; timing only -- results are wrong): 1 no DMA in the loop, 2 no A reads / MFMAs, 4 no bias sums, 8 no row scaling
    unsigned long long* stamps;             // diagnostic build (STAMPS=1) only: wgradws_kernel's per-wave phase cycle sums [workgroups * waves][8], else null
};

// wgrad_rows_kernel (round 4): weight gradient of one linear map on FEW rows (<= 2 048) with the whole row reduction inside the workgroup
// and the optimizer update in its epilogue -- no slabs, no reduction launch
#define WGR_MAX_JOBS 6
struct WgradRowsJob {
    const uint16_t* X; int ldX;             // layer input, P-layout [R][ldX]
    const uint16_t* G; int ldG;             // dpre of the layer's outputs, P-layout [R][ldG]
    int R;                                  // valid rows
    const float* rowscale;                  // optional [>= R rounded up to 32]: G rows are multiplied by it on the way in (the output layer: G = the stored s), else null
    int sub0, sub1, split;                  // layer-table index of out-features < split and (merged mu | sigma head) >= split; sub1 = -1: none
    int ib, jb, wg_begin;                   // filled by launch_wgrad_rows: 64-feature blocks of the in / out space, first workgroup
};
struct WgradPGroup {                        // up to 3 independent 8-wave weight gradients in one launch
    WgradPArgs a[3];
    int n;
    int gx[3], gy[3];                       // j-blocks / i-blocks of each
    int zbeg[4];                            // first blockIdx.z of each (zbeg[n] = total)
};

struct Chain2FwdArgs {                // chain2_fwd_kernel: the 2-layer model's per-sample blocks (q(z2|z1) and p(z1|z2)) on M rows, one launch
    uint16_t* Z1P;                    // z1 rows, bf16 P-layout [M][32*KT0]: MADE here (z1 = mu1 + sigma1*eps1, iwae2.py:61) and kept for the decoder and the weight gradient
    float* lqz1x;                     // log q(z1|x) per row (iwae2.py:123)
    const char *e_img1, *e_img2, *e_imgh;     // MG-major forward images of encode_z1_to_z2: l1 (KT0 k-steps), l2, head (KTH k-steps each)
    const char *d_img1, *d_img2, *d_imgh;     // ... of decode_z2_to_z1: l1 (KT1 k-steps), l2, head (KTH k-steps each)
    int M, k, B, D0, D1;              // data rows, samples per image, images; latent widths of z1 and z2
    uint16_t *EH1, *EH2;              // encode block's tanh activations, P-layout [M][32*KTH], kept for the backward pass (null: forward only)
    float* EHEAD;                     // its head, fp32 [M][64*KT1] (mu2 | sigma2), or null
    uint16_t* Z2P;                    // z2 rows, P-layout [M][32*KT1], or null
    uint16_t *DH1, *DH2;              // decode block's tanh activations, or null
    float* DHEAD;                     // its head, fp32 [M][64*KT0] (mu_p | sigma_p), or null
    const float* head1; int ldH1;     // head of q(z1|x) per image [B][ldH1] (mu1 | sigma1 at 32*KT0): z1 in float32 for log p(z1|z2)
    EpsSrc eps1, eps2;                // the draws of z1 and z2
    float *lpz1z2, *lpz2, *lqz2z1;    // per-row log-densities (iwae2.py:122-124)
};
bool chain2_fwd_ok(int KT0, int KTH, int KT1, int M);
void launch_chain2_fwd(const Chain2FwdArgs& a, hipStream_t st);

struct GBlockBwdArgs {                // gblock_bwd_kernel: backward of a per-sample BasicBlock of the 2-layer model from its Gaussian head, one launch
    const char* imgH;                 // FORWARD image of the head (KTH k-steps, KTL groups: mu groups then sigma groups): the head is recomputed
    const char *imgBh, *imgB2, *imgB1;   // backward images (MG-major) of the head (2*KTL k-steps), l2 and l1 (KTH k-steps each)
    const uint16_t *H2, *H1;          // the block's stored tanh activations, P-layout [M][32*KTH]
    const float* gx;                  // [M] dLoss/dlog_w of the row
    int M, k, B, D;                   // D: real width of the head's latent
    EpsSrc eps;                       // MODE 0: the draws of z1; MODE 1: the draws of z2
    const float* head1; int ldH1;     // MODE 0: head of q(z1|x) per image (z1 = mu1 + sigma1*eps1 in float32)
    const float* DZIN;                // MODE 1: dz2 from the decode block, float32 [M][32*KTL]
    uint16_t* DHP;                    // dhead, P-layout [M][64*KTL] (for the head's weight gradient)
    uint16_t *D2P, *D1P;              // dpre of l2 / l1, P-layout [M][32*KTH]
    float* DZ2;                       // MODE 0: dz2 out, float32 [M][32*KTIN]
    uint16_t* DZD;                    // MODE 0: out, MODE 1: in -- the direct term of dz1 (-dm), bf16 [M][128] natural feature order
    const float* DZDEC;               // MODE 1: dz1 from the z1 -> x decoder, float32 [M][ldDZDEC]
    int ldDZDEC;
    uint16_t* DZOUT;                  // MODE 1: the three terms of dz1 summed, bf16 [M][32*KTIN] natural feature order (latent_bwd_kernel's input)
};
bool gblock_bwd_ok(int KT0, int KTH, int KT1, int M);
void launch_gblock_bwd(int mode, const GBlockBwdArgs& a, hipStream_t st);

struct GaussLpArgs {
    const float* zhead; int ldZH; int Dzp;   // head that generated z (per image)
    const float* phead; int ldPH; int Dpp;   // head that scores z (per row)
    int D, M, k;
    EpsSrc eps;
    float* out;
};



struct GaussBwdArgs {
    int mode;
    const float* G;
    const float* head; int ldH; int D, Dp;       // per-row head
    const float* zhead; int ldZH; int Dzp;       // mode 0: head generating z1 (per image)
    const float* dz_in; float* dz_direct; int ldDZ;
    EpsSrc eps;
    int M, Mp, k;
    uint16_t* DHP;                    // dhead bf16 P-layout [M][2Dp], or null
    float* DHF;                       // float32 mode: dhead float32 [M][2Dp], or null
};

struct LayerDesc {
    int Kin, Nout, joff;
    size_t offW, offb;                // offsets into the flat parameter / gradient buffers
    char* imgF; int KT_F;             // forward image (rows = out-features)
    char* imgB; int KT_B; int MT_B; int imgB_kmajor;   // backward image (rows = in-features) or null
    const float* slabW; const float* slabB; int nsplit; int slab_ld; size_t slab_stride;
    size_t slabB_stride;              // distance between the row splits' bias sums (0: slab_ld, the usual [nsplit][slab_ld] array; the pre-weighted output layer keeps them as a row of slabW)
    int block_begin;                  // first block of this layer in the flat elementwise grid (256 elements per block)
    int rblock_begin;                 // same for the slab-reduce grid (64 elements per block)
};

// arms a completion event for the NEXT launch_dense / launch_out_bwd / launch_reduce_grads of this thread: it rides on that
// kernel's dispatch packet (cheaper on both streams than hipEventRecord behind the kernel); consumed by that launch
void set_launch_stop_event(hipEvent_t e);
void launch_dense(int epi, const DenseArgs& a, hipStream_t st);
void launch_out_bwd(const OutBwdArgs& a, hipStream_t st);
bool bern_lse_ok(const DenseArgs& a);   // ... and can take lse_kernel's work for its rows (DenseArgs.lse_on)
bool bern_pipe_ok(const DenseArgs& a);  // shapes bern_pipe_kernel covers (launch_dense falls back to dense_kernel<EPI_BERN> otherwise)
bool out_bwd_has_s_mode(int KT);      // hidden widths with a compiled out_bwd_s_kernel / dec_bwd_kernel
void launch_dec_bwd(const DecBwdArgs& d, hipStream_t st);
void launch_wgradp(const WgradPArgs& a, int nsplit, int shape, hipStream_t st);      // shape: see wgradp_strip
int wgradp_strip(int shape);          // j-tiles (16 out-features each) per block of a shape: 8 -> 8, 16 / 7 -> 16
void launch_wgradp_group(const WgradPGroup& g, hipStream_t st);
void launch_wgradws_group(const WgradPGroup& g, hipStream_t st);     // shape-7 (specialised waves) gradients, non-row-weighted, in one launch
void launch_prep_rows(const float* x, const float* cond, int B, int X, int C, int Xp, int Bp, uint16_t* XP, hipStream_t st);
void launch_gather_binarize(const uint8_t* data, const int32_t* order, int start, int N, int B, int X, int Xp, int Bp, uint64_t seed,
                            uint32_t epoch, uint16_t* XP, float* xf, hipStream_t st,
                            const uint8_t* labels = nullptr, int C = 0, float* cond_out = nullptr);      // labels: class ids of the resident set (conditional models)
void launch_eps_gen(const EpsSrc& e, int M, int D, int ld, float* out, hipStream_t st, int max_blocks = 0);   // max_blocks > 0: grid-stride over at most that many blocks
void launch_eps_gen_multi(const EpsSrc& e, int M, int D, int ld, float* out, int nsteps, size_t step_stride, hipStream_t st);   // steps e.step .. e.step + nsteps - 1, step s at out + s * step_stride
void launch_sample(const SampleArgs& a, hipStream_t st);
void launch_gauss_lp(const GaussLpArgs& a, hipStream_t st);
void launch_lse(const LseArgs& a, hipStream_t st);
void launch_latent_bwd(const LatentBwdArgs& a, hipStream_t st);
void launch_gauss_bwd(const GaussBwdArgs& a, hipStream_t st);
// Sums the slabs of reduce blocks [first_block, first_block + nblocks) of the layer table into the flat gradient.
// per_b != null: one extra block turns the per-image values into the batch means (scalars) of the step.
// fuse_adam: the Adam update (grad_scale 1) + weight-image refresh of each element follows its slab sum in the same thread.
void launch_reduce_grads(const LayerDesc* layers, int nlayers, int first_block, int nblocks, float* grad, float* param, float* mom, float* vel,
                         float alpha, float beta1, float beta2, float eps, int fuse_adam, const float* per_b, int B, float beta, float* scalars, hipStream_t st,
                         int first_block2 = 0, int nblocks2 = 0);      // (optional second block range of the same launch)
// fuse_adam: Keras Adam (grad_scale 1) + image refresh in the epilogue; per_b != null: one extra block makes the step's batch means
void launch_wgrad_rows(const WgradRowsJob* jobs, int njobs, const LayerDesc* layers, float* grad, float* param, float* mom, float* vel, float alpha,
                       float beta1, float beta2, float eps, int fuse_adam, const float* per_b, int B, float beta, float* scalars, const char* zero, hipStream_t st);
void launch_scalars(const float* per_b, int B, float beta, float* out, hipStream_t st);
void launch_adam(const LayerDesc* layers, int nlayers, int nblocks, float* param, const float* grad, float* mom, float* vel,
                 float alpha, float gscale, float beta1, float beta2, float eps, int do_update, hipStream_t st, int first_block = 0);   // blocks [first_block, first_block + nblocks) of the elementwise grid
void launch_export_rows(const float* in, int B, int k, float* out, hipStream_t st);
void launch_export_z(const SampleArgs& a, float* zout, hipStream_t st);
void launch_snis(const float* z, const float* wn, int B, int k, int D, float* out, hipStream_t st);
void launch_unpack_p(const uint16_t* P, int rows, int F, int Fp, float* out, hipStream_t st);
void launch_eps_dump(const EpsSrc& e, int B, int k, int D, float* out, hipStream_t st);

// ---- float32 mode (fp32_kernels.hip)
enum { GEMM_EPI_NONE = 0, GEMM_EPI_TANH = 1, GEMM_EPI_EXP = 2, GEMM_EPI_DTANH = 3, GEMM_EPI_BERN = 4 };
struct GemmF32Args {
    const float* A; long sam, sak;        // element (m,k) of op(A) at A[m*sam + k*sak]
    const float* B; long sbk, sbn;        // element (k,n) of op(B) at B[k*sbk + n*sbn]
    float* C; long ldc;                   // C[m*ldc + n]
    int M, N, K;
    const float* bias;                    // [N] or null
    int epi;                              // GEMM_EPI_*: none | tanh | exp(.)+1e-6 | times (1 - ACT^2)
    const float* ACT; long ldact;         // GEMM_EPI_DTANH: stored tanh activation of the outputs
    int accumulate;                       // C += instead of C =
    int kchunk; size_t slab_stride;       // K split over blockIdx.z: split z covers kchunk k's and writes C + z*slab_stride
    int avec, bvec;                       // set by launch_gemm_f32: the operand's quads may be fetched as float4
    int cvec;                             // ... and C (bias, ACT, Cones) may be accessed as float4 (gemm_f32_v2_kernel's transposed epilogue)
    // GEMM_EPI_BERN (128-tile kernel only, forward-only calls): the logits are not stored; every 64-column half tile leaves its partial
    // log p(x|z) = sum_n x_n l_n - softplus(l_n) per row in part[(2 * column tile + half) * part_stride + m]; lse_kernel adds them in fixed order
    const float* XB; int bern_k, bern_X;  // x [B][bern_X] float32 in {0,1}; row m belongs to image m / bern_k
    float* part; size_t part_stride;
    // training step (round 3): GEMM_EPI_BERN with C != null also keeps s = x - sigmoid(l) there (the gradient of that sum wrt l; what the
    // backward pass reads instead of logits), and its consumers take the row weight g_r on the fly instead of a pass that makes dl = g_r s:
    const float* brow_scale;              // element (k,n) of op(B) is multiplied by brow_scale[k] as it is fetched (the weight gradient X^T (g_r s)), or null
    const float* orow_scale;              // row m of the product is multiplied by orow_scale[m] in front of the epilogue (dX = g_r (s W^T) ...), or null
    // weight gradients: op(A) = X^T gets one more row m == M of ONES, whose product row -- the column sums of op(B), i.e. the bias gradient --
    // goes to Cones[z * cones_stride + n] instead of C (round 3: no separate pass over G for the bias gradients)
    float* Cones; size_t cones_stride;
    unsigned long long* stamps;           // diagnostic build (STAMPS=1) only: [workgroups * 4 waves][8] phase cycle sums of gemm_f32_v2_kernel, else null
    int tile_mode;                        // 0: launch_gemm_f32 picks the tile; 1: 4-wave tiles only; 2: 4-wave tiles at <= 3 waves per SIMD (a slot per SIMD stays free for few-row kernels beside it)
    int dbg;                              // diagnostic builds (DIAG=1) only: timing ablations of gemm_f32_v2_kernel (option f32_gemm_dbg; results wrong)
};
// dec_fwd_f32_kernel: the decoder forward in float32 in ONE launch (z -> tanh -> tanh -> logits -> log p(x|z)), rows stationary
struct DecFwdF32Args {
    const float* Z; int ldz; int Din;      // decoder input rows [M][ldz] (z, or concat(z, y)), Din features
    int M, H, X;                           // data rows, hidden width, pixels
    const float *W1, *b1, *W2, *b2, *W3, *b3;      // Keras kernels [in][out] and biases inside the float32 master parameters
    float *G1, *G2; int ldg;               // the tanh activations [M][ldg], kept for the backward pass (null: forward only)
    float* S; int ldS;                     // s = x - sigmoid(l) [M][ldS] for the backward pass, or null
    const float* XB; int k;                // x [B][X] float32 in {0,1}; row m belongs to image m / k
    float* lpxz;                           // [M] log p(x|z) per row
    const char* zero;                      // >= 1 KiB of zeros
    unsigned long long* stamps;            // diagnostic build (STAMPS=1) only: [workgroups * 4 waves][8] phase cycle sums, else null
};
bool dec_fwd_f32_ok(const DecFwdF32Args& a);
void launch_dec_fwd_f32(const DecFwdF32Args& a, hipStream_t st);
long gemm_f32_tiles(int M, int N, int tile_mode = 0);                       // output tiles of the kernel launch_gemm_f32 takes for an M x N product
bool gemm_f32_takes_big(int M, int N, int nsplit);      // launch_gemm_f32's kernel choice (GEMM_EPI_BERN needs the 128-tile kernel)
void launch_gemm_f32(const GemmF32Args& a, int nsplit, hipStream_t st);
extern int g_gemm_f32_dbg;
extern int g_gemm_f32_ksplit_min_tiles, g_gemm_f32_v2_small_min;
extern bool g_gemm_f32_w8, g_gemm_f32_v2_small, g_gemm_f32_ksplit;
int gemm_f32_fewrows_split(int M, int N, int K);        // > 1: launch_gemm_f32_fewrows splits K that many ways (scratch: split * M * N floats)
void launch_gemm_f32_fewrows(const GemmF32Args& a, float* slabs, hipStream_t st);
int gemm_f32_slots(int M, int N, int tile_mode = 0);                       // workgroups of launch_gemm_f32's kernel for an M x N product the chip holds at once
extern bool g_gemm_f32_v2;                               // false: gemm_f32_big_kernel (the round-3 k loop) instead of gemm_f32_v2_kernel
void launch_reduce_slabs_f32(const float* slabs, size_t stride, int nsplit, size_t n, float* out, hipStream_t st);
// every slab sum of a float32 step in one launch (round 5): out[i] = sum over the job's slabs, i < n; block_begin is filled by the launcher
#define REDUCE_SLABS_MAX_JOBS 40
struct ReduceSlabsJob { const float* slabs; size_t stride; size_t n; float* out; int nsplit; int block_begin; };
struct ReduceSlabsJobs { ReduceSlabsJob job[REDUCE_SLABS_MAX_JOBS]; int n; };
void launch_reduce_slabs_multi_f32(ReduceSlabsJobs& jobs, hipStream_t st);
void launch_bern_f32(const float* logits, size_t ld, const float* x, int X, int M, int k, float* lpxz, hipStream_t st);
void launch_dl_f32(float* logits, size_t ld, const float* x, int X, int M, int k, const float* gx, hipStream_t st);
void launch_sigmoid_f32(float* v, size_t n, hipStream_t st);
void launch_export_mat(const float* in, int B, int k, int X, float* out, hipStream_t st);
void launch_concat_f32(const float* a, int na, const float* b, int nb, int rows, float* out, hipStream_t st);

}  // namespace iwae
