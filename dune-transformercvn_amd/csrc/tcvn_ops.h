// Internal host-side launch interface of the gfx950 kernels (the public C-ABI in include/tcvn_hip.h wraps these).
#pragma once
#include "tcvn_common.h"
#include "bn_lf.h"

namespace tcvn {

enum AMode { A_1X1 = 0, A_1X1_POOL = 1, A_3X3 = 2, A_STEM = 3 };

// ---------------------------------------------------------------------------------------------------------
// Forward convolution as an implicit GEMM:  Out[m][n_off+n] = sum_k a(m,k) * Wk[n][k] + bias[n]
//   a(m,k) is generated from the NHWC input with the producer-side BatchNorm + PReLU folded into the load:
//   A_1X1      a = f_k(X[m][k])                                   f_c(x) = prelu(x*sc[c] + sh[c], sl[c])
//   A_1X1_POOL a = 1/4 sum_{2x2} f_k(X[(img,2ho+dy,2wo+dx)][k])   (avg-pool commuted in front of the 1x1 conv)
//   A_3X3      k = tap*C + c ; a = inside ? f_c(X[m + (ky-1)*W + (kx-1)][c]) : 0
//   A_STEM     k = tap*3 + c ; 7x7 stride 2 pad 3 on the raw image, no transform
// Optional epilogue: dropout, per-channel (sum, sum of squares) partials for the consumer BatchNorm.
// ---------------------------------------------------------------------------------------------------------
struct ConvFwdArgs {
    int mode, amode;
    const void* A; long lda;
    int M, N, K, Kp;
    int C, H, W, Hin, Win;
    const float *sc, *sh, *sl;
    const void* Wk; const float* bias;
    void* Out; long ldo; int n_off;
    const void* Wfrag;               // optional: the same weights in MFMA fragment order (bf16 fast paths)
    const void* Aact;                // optional: bf16 copy of A with the BN+PReLU already applied (k_act_bf16)
    int dbg;                         // timing-ablation bits (TCVN_DBG env, validation builds only): 1 no DMA, 2 no MFMA, 4 no epilogue
    const void* zeros;               // >= 256 B of zeros in device memory (source of padding rows for LDS-DMA)
    double* part; int nblk;          // [nblk][N][2]; nblk = grid.x
    float drop_p; uint64_t seed; uint32_t stream_id;
    uint32_t* keep_out;              // optional [M]: the 3x3 pair kernel stores the keep flags of a pixel's N <= 32 channels as one word
    int act_fused;                   // bf16 3x3 tile kernels (forward pair kernel, weight gradient): Aact is the RAW [pixels][128] map; the wave that
                                     // fetched a row applies prelu(sc*x + sh, sl) to it in LDS, once, before any tap reads it -- no activated copy in HBM
    LfLink lf;                       // forward pair kernel with act_fused (round 5): lf.isum != nullptr -> the (sc, sh) table of the image's BatchNorm is
                                     // derived in the prologue from the producer's fixed-point sums (bn_lf.h); sc / sh are then not read
    long long* isum_out; long isum_stride;   // forward pair kernel: when set, the statistics of the N output channels are ADDED to replica
                                     // (blockIdx.x % LF_REP) of isum_out[n][2] (fixed point, bn_lf.h; replicas isum_stride long longs apart) instead of
                                     // leaving as a partial row in `part`
};
int conv_fwd(const ConvFwdArgs& a, hipStream_t st);
bool conv3x3_fwd_pair(const ConvFwdArgs& a);          // true when conv_fwd(a) runs the pair kernel (the one that honours ConvFwdArgs::lf / isum_out)
bool conv3x3_fwd_writes_keep(const ConvFwdArgs& a);   // true when conv_fwd(a) runs the kernel that fills keep_out
bool conv3x3_act_fusable(const ConvFwdArgs& a);       // true when both the forward kernel conv_fwd(a) would run and the weight-gradient tile kernel of the
                                                      // same layer can take the raw map (ConvFwdArgs::act_fused)
int conv_fwd_grid(int M);            // number of M-blocks of the generic kernels for M rows (<= 512)
int conv_fwd_nblk(const ConvFwdArgs& a);   // grid.x (== rows of `part`) conv_fwd will use for these arguments (<= 512)
// bf16 3x3 fast path on padded LDS tiles (conv3x3_tile.hip)
bool conv3x3_tile_ok(const ConvFwdArgs& a);
int conv3x3_tile_nblk(const ConvFwdArgs& a);
int conv3x3_fwd_tile(const ConvFwdArgs& a, hipStream_t st);
bool conv3x3_tile_enabled();                         // false when TCVN_DISABLE_TILE is set

// ---------------------------------------------------------------------------------------------------------
// BatchNorm plumbing
// ---------------------------------------------------------------------------------------------------------
// Reduce the producer's stat partials of channels [c_new0, c_new0+n_new) into bstat (mean, biased var) and build
// the consumer's (scale, shift) table for channels [0, C); update the consumer's running statistics.
struct BnLinkArgs {
    const double* part; int nblk; int part_ld;   // partial row length (channels of the producing launch)
    int c_new0, n_new;                           // channel window that `part` describes (may be empty)
    double* bstat;                               // [Cbuf][2] (mean, biased var) of the buffer
    long count;                                  // elements per channel
    int C;                                       // consumer channels
    const float *gamma, *beta;
    float *running_mean, *running_var;           // updated when `train`
    float *sc, *sh;                              // outputs [C]
    int train; float eps, momentum;
    const long long* isum; long isum_stride;     // optional: the window's sums come from fixed-point accumulators (bn_lf.h: LF_REP replicas of
                                                 // [n_new][2], isum_stride long longs apart, added by a producer with lf_add) instead of the partial rows
};
int bn_link(const BnLinkArgs& a, hipStream_t st);

// eval-mode tables for many BN layers in one launch
struct BnEvalDesc { const float *gamma, *beta, *rm, *rv; float *sc, *sh; int C; };
int bn_eval_tables(const BnEvalDesc* d_descs, int n, float eps, hipStream_t st);

// ---------------------------------------------------------------------------------------------------------
// Pixel-map scatter (COO -> dense NHWC), stem pooling, global pooling
// ---------------------------------------------------------------------------------------------------------
struct ScatterArgs {
    int mode; const int* coords; const float* values; long nnz; int n_img;
    void* img; int H, W, Cpix; int log_pixels; float noise_std; uint64_t seed;
};
int scatter_pixels(const ScatterArgs& a, hipStream_t st);

// D[img,ho,wo,0:C] = avgpool3x3s2( prelu(bn(C0)) ), with stat partials for those channels
struct Pool0Args {
    int mode; const void* X; int n_img, Hin, Win, C; const float *sc, *sh, *sl;
    void* Out; long ldo; int Ho, Wo; double* part; int nblk;
};
int pool0_fwd(const Pool0Args& a, hipStream_t st);
int pool0_grid(int n_img, int Ho, int Wo);

// F[img][c] = mean_hw prelu(bn(X[img,:,:,c]))   (fp32 output)
struct HeadPoolArgs { int mode; const void* X; long ldx; int n_img, HW, C; const float *sc, *sh, *sl; float* F; };
int head_pool_fwd(const HeadPoolArgs& a, hipStream_t st);

// Out[m][0:C] = bf16(prelu(X[m][0:C]*sc + sh, sl)) (bf16 in/out)
struct ActArgs { const void* X; long ldx; long M; int C; const float *sc, *sh, *sl; void* Out; long ldo; };
int act_bf16(const ActArgs& a, hipStream_t st);

// Weight re-layout: reference OIHW fp32 -> kernel layout [N][Kp] (k = tap*Cin + c), typed T, zero padded.
struct PackDesc { const float* src; void* dst; int N, Cin, taps, Kp; int transpose; int frag; };
int pack_weights(const PackDesc* d_descs, int n, int mode, hipStream_t st);

// ---------------------------------------------------------------------------------------------------------
// Backward convolutions
// ---------------------------------------------------------------------------------------------------------
// Effective output gradient of a produced tensor X (one dense-layer output slice, a bottleneck output, a transition
// output or conv0's output):   eff(m, n) = drop(m, n) * ( G[m][c_off+n] + P[n] * X[m][c_off+n] + Q[n] )
// G holds the sum over consumers of (gamma*rstd) * dU; P/Q carry the batch-mean terms of every consumer BatchNorm's
// backward (they are affine in X, so they are accumulated per channel instead of per element).
struct EffSrc {
    const void* G; long ldg;
    const void* X; long ldx;
    int c_off, N;
    const float *P, *Q;
    float drop_p; uint64_t seed; uint32_t stream_id;
    const uint32_t* keep;            // optional (bf16 tile kernels): bit n of keep[m] = dropout keep flag of channel n of pixel m, as the
                                     // forward kernel drew it (ConvFwdArgs::keep_out) -- a bit test instead of the hash per element
    const void* ey;                  // optional (3x3 weight gradient, N == 32): the eff rows themselves, [pixels][32] bf16, as the data-gradient
                                     // kernel of the same layer built and stored them (ConvDgradArgs::ey_out): fetched by LDS-DMA, no arithmetic
};

enum DMode { DG_1X1 = 0, DG_1X1_POOL = 1, DG_3X3 = 2 };
// dA[m][c] = sum_k eff(...) * Wt[c][k], followed by the PReLU + BatchNorm backward of the consumer norm:
//   u = sc*x + sh ; dU = dA * prelu'(u) ; Gout[m][c] (+)= sc * dU ; partial sums (sum dU, sum dU*x, sum dA*min(u,0))
struct ConvDgradArgs {
    int mode, dmode;
    EffSrc e;
    int M, N, Kp;
    int H, W, Hin, Win;
    const void* Wt;
    const void* Xin; long ldxin;
    const float *sc, *sh, *sl;
    void* Gout; long ldgo; int accumulate;
    double* part; int nblk;        // [nblk][N][3]
    const void* Wfrag;             // optional: Wt in MFMA fragment order (bf16 padded-tile 3x3 kernel)
    const void* zeros;             // optional: >= 64 B of zeros (LDS-DMA source of padding rows)
    void* ey_out;                  // optional [M][32] bf16: the consecutive-tile 3x3 kernel stores every eff row it builds (for the weight gradient)
};
int conv_dgrad(const ConvDgradArgs& a, hipStream_t st);
bool conv3x3_dgrad_writes_ey(const ConvDgradArgs& a);   // true when conv_dgrad(a) runs the kernel that fills ey_out
int conv_dgrad_nblk(const ConvDgradArgs& a);      // grid.x (rows of `part`) conv_dgrad will use (<= 512)
bool conv3x3_dgrad_tile_ok(const ConvDgradArgs& a);
int conv3x3_dgrad_tile_nblk(const ConvDgradArgs& a);
int conv3x3_dgrad_tile(const ConvDgradArgs& a, hipStream_t st);

// one reduction job: dst[i] += sum_{s < nslab} slab[s*stride + i], i < count
struct SlabJob { const float* slab; float* dst; int nslab; long count, stride; int ny, per_y; int v4; };   // v4: 16-B loads (count, stride multiples of 4, 16-B aligned)
// dWk[n][k] += sum_m eff(m, n) * a(m, k)   (a = the forward A operand, regenerated), dbias[n] += sum_m eff(m, n)
struct ConvWgradArgs {
    int mode;
    ConvFwdArgs fa;
    EffSrc e;
    float* dWk; float* dbias;
    float* slab; long slab_bytes;   // scratch for per-workgroup partial gradients (padded-tile kernel)
    int nfast;        // 1: dWk is laid out [k][32] (out-channel fastest) and the padded-tile kernel must be used (bf16 3x3)
    SlabJob* deferred;   // optional [2] (padded-tile kernel): its slab reductions (weights, bias) are returned as jobs instead of being launched --
                         // the caller folds them into a later reduction launch; the slab must stay untouched until then
};
int conv_wgrad(const ConvWgradArgs& a, hipStream_t st);
bool gemm_tn_f32_ok(const ConvWgradArgs& a);        // gemm_tn_f32.hip: fp32 1x1 weight gradient over an already activated operand (transitions, parity mode)
int gemm_tn_f32(const ConvWgradArgs& a, hipStream_t st);
bool conv3x3_wgrad_tile_ok(const ConvWgradArgs& a);
// fp32 padded-tile 3x3 kernels (conv3x3_f32.hip): parity mode, C = 128 -> N <= 32
bool conv3x3_fwd_f32_ok(const ConvFwdArgs& a);
int conv3x3_fwd_f32_nblk(const ConvFwdArgs& a);
int conv3x3_fwd_f32(const ConvFwdArgs& a, hipStream_t st);
bool conv3x3_dgrad_f32_ok(const ConvDgradArgs& a);
int conv3x3_dgrad_f32_nblk(const ConvDgradArgs& a);
int conv3x3_dgrad_f32(const ConvDgradArgs& a, hipStream_t st);
// fp32 1x1 backward kernels (conv1x1_f32.hip): parity mode, Cin <= 512 -> 128 channels
bool conv1x1_fwd_f32_ok(const ConvFwdArgs& a);
int conv1x1_fwd_f32_nblk(const ConvFwdArgs& a);
int conv1x1_fwd_f32(const ConvFwdArgs& a, hipStream_t st);
bool conv1x1_dgrad_f32_ok(const ConvDgradArgs& a);
int conv1x1_dgrad_f32_nblk(const ConvDgradArgs& a);
int conv1x1_dgrad_f32(const ConvDgradArgs& a, hipStream_t st);
bool conv1x1_wgrad_f32_ok(const ConvWgradArgs& a);
int conv1x1_wgrad_f32(const ConvWgradArgs& a, hipStream_t st);
bool conv3x3_wgrad_f32_ok(const ConvWgradArgs& a);
int conv3x3_wgrad_f32(const ConvWgradArgs& a, hipStream_t st);
int conv3x3_wgrad_tile(const ConvWgradArgs& a, hipStream_t st);

// Reduce the backward partials of one BatchNorm, emit parameter gradients and the (P, Q) coefficients of its input.
struct BnBwdLinkArgs {
    const double* part; int nblk; int C;
    const double* bstat;                 // (mean, biased var) of the BN input channels
    long count; float eps;
    const float* gamma;
    float *dgamma, *dbeta, *dslope;      // accumulated
    float *P, *Q; int accumulate_pq;
};
int bn_bwd_link(const BnBwdLinkArgs& a, hipStream_t st);
int slab_reduce4_link(const SlabJob* jobs, int n, const BnBwdLinkArgs& link, hipStream_t st);   // up to four slab reductions + one link in ONE launch

// head: dF[img][c] -> G of the last block (+ partials);  stem tail: eff of block-1's first channels -> DU0 (+ partials)
struct HeadPoolBwdArgs {
    int mode; const void* X; long ldx; int n_img, HW, C; const float *sc, *sh, *sl; const float* dF;
    void* Gout; long ldgo; double* part; int nblk;
};
int head_pool_bwd(const HeadPoolBwdArgs& a, hipStream_t st);
int head_pool_bwd_grid(int n_img);
struct Pool0BwdArgs {
    int mode; const void* X; int n_img, Hin, Win, C; const float *sc, *sh, *sl;   // X = conv0 output
    EffSrc e; int Ho, Wo;                                                          // gradient of the pooled map
    void* DU; double* part; int nblk;
    const uint32_t* act; const void* cline;      // optional (bf16 tile kernel): inactive positions read `cline` instead of X and their DU rows are not stored
};
int pool0_bwd(const Pool0BwdArgs& a, hipStream_t st);
int pool0_bwd_grid(int n_img, int Hin, int Win);

// bf16 TN GEMM over pixels (gemm_tn.hip): C[i][j] += sum_m L[m][i] * R[m][j]
SlabJob slab_job(const float* slab, int nslab, long count, float* dst, long stride);
int slab_reduce2(const SlabJob& a, const SlabJob& b, hipStream_t st);          // two independent jobs in one launch (b may be empty)
int slab_reduce4(const SlabJob* jobs, int n, hipStream_t st);                  // up to four; empty jobs are skipped
struct GemmTnArgs { const void* L; long ldl; int Li; const void* R; long ldr; int Rj; long M; float* C; long ldc; const void* zeros;
                    float* slab; long slab_bytes;        // slab: scratch for per-slice partial tiles (no contended atomics)
                    int Ci;                              // rows of C written (<= Li; L columns in [Ci, Li) are zero padding)
                    SlabJob extra;                       // optional second reduction folded into this GEMM's slab reduction
                    const float *rsc, *rsh, *rsl; int Rreal; };   // optional: R is raw, transform prelu(rsc*x + rsh, rsl) in LDS
// dst[i] += sum_s slab[s*count + i]   (deterministic reduction of per-workgroup partial results)
int slab_reduce(const float* slab, int nslab, long count, float* dst, hipStream_t st, long stride = 0);   // stride 0 = count
bool gemm_tn_ok(const GemmTnArgs& a);
int gemm_tn_bf16(const GemmTnArgs& a, const char* label, hipStream_t st);

// bf16 NT GEMM with fused 1x1-convolution epilogues (gemm_nt.hip)
enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_DGRAD_POOL = 2 };
struct GemmNtArgs {
    int epi;
    const void* A; long lda; int K;          // [M][lda] bf16, K columns used
    long M; int N;                           // output columns (any count; 8-column chunks, the tail chunk is masked)
    const void* Wfrag; int Kp;               // weights [N][Kp] in MFMA fragment order
    const void* zeros;
    const float* bias; void* Out; long ldo; int n_off;             // EPI_FWD
    const float *asc, *ash, *asl; int Kreal;                       // EPI_FWD, optional: A is raw, transform prelu(asc*x + ash, asl) in LDS;
                                                                   // channels >= Kreal are zeroed
    const float *osc, *osh, *osl;                                  // EPI_FWD, optional (eval mode, part == nullptr): Out = prelu(osc*(C + bias) + osh, osl)
                                                                   // -- the consumer's BatchNorm (running statistics) + PReLU applied in the epilogue
    const void* Xin; long ldxin; const float *sc, *sh, *sl;        // EPI_DGRAD*: BatchNorm input + its table
    void* Gout; long ldgo;
    int g_write;                             // EPI_DGRAD_POOL: this launch is the FIRST contribution to Gout: write, do not read-add
                                             // (pixels outside every 2x2 window are zeroed by zero_pool_remainder)
    int H, W, Hin, Win;                      // EPI_DGRAD_POOL geometry (rows = pooled pixels H x W of inputs Hin x Win)
    double* part; int nblk;                  // [nblk][N][2 (fwd) | 3 (dgrad)]
};
// rows >= 2*Ho and columns >= 2*Wo of an [n, Hin, Win, ld] bf16 map := 0 (the pixels a floor-mode 2x2 pooling never reads)
int zero_pool_remainder(void* G, long ld, int n_img, int Hin, int Win, int Ho, int Wo, hipStream_t st);
bool gemm_nt_ok(const GemmNtArgs& a);
int gemm_nt_nblk(const GemmNtArgs& a);
int gemm_nt_bf16(const GemmNtArgs& a, const char* label, hipStream_t st);

// Fused backward of a bottleneck 1x1 convolution (bwd1x1_fused.hip): effective gradient of the 128-channel bottleneck output formed in
// LDS from (DU, Y, PY, QY), bias gradient, data gradient with the norm1 / PReLU1 backward epilogue (G += sc*dU, statistics partials) and
// the weight gradient against the activated input, in one pass over the pixels -- replaces eff_materialize_bf16 + gemm_tn_bf16 +
// gemm_nt_bf16<EPI_DGRAD> of a dense layer.
struct Bwd1x1Args {
    const void* DU; const void* Y;          // [M][128] bf16 each: the EffSrc (G, X) of the bottleneck output
    const float *PY, *QY;                   // [128]
    long M;
    const void* Xin; long ldx; int cin;     // raw concat buffer (norm1 input), its row pitch, channels of this layer (any count: a partial last 8-channel chunk is masked)
    const float *sc, *sh, *sl;              // norm1 (scale, shift) table, PReLU1 slope
    void* Gout; long ldg;                   // gradient accumulator of the concat buffer (read-add-write on [0, cin))
    const void* Wfrag; int Kp;              // W1 transposed ([cin][128]) in MFMA fragment order; Kp == 128
    const void* zeros;
    double* part; int nblk;                 // [nblk][cin][3]; nblk = bwd1x1_fused_nblk()
    float* slab; long slab_bytes; long ldc; // per-workgroup weight-gradient tiles [nblk][128][ldc]; ldc = row pitch of the kernel-layout dW1
    float* tail;                            // [nblk][128] bias column-sum partials
};
bool bwd1x1_fused_ok(const Bwd1x1Args& a);
int bwd1x1_fused_nblk(const Bwd1x1Args& a);
int bwd1x1_fused_launch(const Bwd1x1Args& a, hipStream_t st);
bool bwd1x1_wide_ok(const Bwd1x1Args& a);       // bwd1x1_wide.hip: the same launch interface for 128 < cin <= 512 (called through bwd1x1_fused_*)
int bwd1x1_wide_nblk(const Bwd1x1Args& a);
int bwd1x1_wide_launch(const Bwd1x1Args& a, hipStream_t st);
int bwd1x1_fused_reduce(const Bwd1x1Args& a, float* dWk, float* dbias, const SlabJob* extra, hipStream_t st, const BnBwdLinkArgs* link = nullptr);   // slab reductions into dWk [128][ldc], dbias [128]
                                                                                                  // (+ up to two more jobs in the same launch)

// Forward of a bottleneck 1x1 convolution on the RAW concat buffer (fwd1x1_fused.hip): norm1 + PReLU1 applied to the landed LDS tiles,
// Y[m][0:128] = bf16(act(x) x W1^T + bias), statistics partials of Y -- replaces act_bf16 + gemm_nt_bf16<EPI_FWD> of a dense layer.
struct Fwd1x1Args {
    const void* Xin; long ldx; int cin;     // raw concat buffer, row pitch, input channels (<= 512)
    const float *sc, *sh, *sl;              // norm1 (scale, shift) table, PReLU1 slope
    long M;
    const void* Wfrag; int Kp;              // W1 [128][Kp] in MFMA fragment order
    const float* bias; void* Out;           // Y [M][128] bf16
    const float *osc, *osh, *osl;           // optional (eval mode, part == nullptr): Out = prelu(osc*(C + bias) + osh, osl) -- norm2 (running statistics) + PReLU2
    const void* zeros;
    double* part; int nblk;                 // [nblk][128][2] or null; nblk = fwd1x1_fused_nblk()
    LfLink lf;                              // round 5: lf.isum != nullptr -> norm1's (sc, sh) table is derived in the prologue from fixed-point sums
                                            // (bn_lf.h) and published by workgroup 0; sc / sh are then not read
    long long* isum_out; long isum_stride;  // when set (train mode): Y's statistics are ADDED to replica (blockIdx.x % LF_REP) of isum_out[128][2]
                                            // (fixed point, replicas isum_stride long longs apart) instead of `part`
};
bool fwd1x1_fused_ok(const Fwd1x1Args& a);
int fwd1x1_fused_nblk(const Fwd1x1Args& a);
int fwd1x1_fused(const Fwd1x1Args& a, hipStream_t st);

// Materialise an effective gradient: Out[m][n] = bf16(drop * (G[m][c_off+n] + P[n]*X[m][c_off+n] + Q[n])), n < e.N;
// optionally colsum[n] += sum_m Out[m][n] (bias gradient of the producing convolution).
struct EffMatArgs { EffSrc e; long M; void* Out; long ldo; float* colsum; float* slab;   // slab: >= 2048*N floats when colsum
                    SlabJob* deferred; };   // when set: the column-sum reduction is returned as a job instead of being launched
int eff_materialize_bf16(const EffMatArgs& a, hipStream_t st);

// XP[m'][c] = bf16( 1/4 sum_{2x2} prelu(D[pixel][c]*sc + sh, sl) )  (pooled activation in front of a transition's 1x1 conv)
struct ActPoolArgs { const void* X; long ldx; int n_img, Hin, Win, C; const float *sc, *sh, *sl; void* Out; long ldo; };
int act_pool_bf16(const ActPoolArgs& a, hipStream_t st);
int act_pool_f32(const ActPoolArgs& a, hipStream_t st);      // fp32 operands; C % 4 == 0; columns [C, ldo) := 0

// bf16 stem kernels (stem.hip)
// Activity of the conv0 output map: bit (img, oy, ox) is set when at least one hit of the COO list lies in the 7x7 / stride-2 window of output
// position (oy, ox).  Every other position holds exactly bf16(bias) (a sum of zeros plus the bias): the pooling backward kernel reads one
// shared 128-B row `cline` for them and does not store their gradient rows (nothing reads those: the conv0 weight gradient walks the hit
// list).  Words per map row: stem_act_words(Wc).
inline int stem_act_words(int Wc) { return (Wc + 31) >> 5; }
int stem_mark(const int* coords, long nnz, int n_img, int H, int W, int Hc, int Wc, uint32_t* act, const float* bias, void* cline, hipStream_t st);
int pool0_bwd_vec_grid(int n_img, int Hin, int Win);
bool pool0_bwd_vec_ok(const Pool0BwdArgs& a);
int pool0_bwd_vec(const Pool0BwdArgs& a, hipStream_t st);
struct StemWgradArgs {
    const int* coords; long nnz; const void* img; int n_img, H, W, Cpix;   // COO hit list + the dense map [n,H,W,Cpix]
    EffSrc e;                                                                // gradient of the conv0 output [n,Hc,Wc,N]
    int Hc, Wc, Kp; float* slab; long slab_bytes;
    int mode;                                                                // element type of img / G / X (MODE_F32 or MODE_BF16)
};
int stem_wgrad_sparse(const StemWgradArgs& a, float* dWk, hipStream_t st);
// conv0 forward on an NHWC4 LDS patch (bf16, 3 -> 64 channels, 7x7 / 2)
bool stem_fwd_ok(const ConvFwdArgs& a);
int stem_fwd_nblk(const ConvFwdArgs& a);
int stem_fwd_bf16(const ConvFwdArgs& a, hipStream_t st);

// Sparse-aware stem (stem_sparse.hip, bf16 mode): conv0 + BatchNorm0 + PReLU0 + AvgPool(3,2) and their backward from the COO hit
// list; neither the dense pixel map nor the conv0 output exists in HBM.  One argument block serves the index build and all passes.
struct StemSparseArgs {
    const int* coords; const float* values; long nnz;       // COO list [nnz][3] (image, y, x), values [nnz][Cpix]
    int n_img, H, W, Cpix, value_mode; float noise_std; uint64_t seed;
    int *row_start, *row_fill; uint4* rec;                   // index (stem_sparse_carve): per (map, pixel row) prefix and cursors; 16-byte records
                                                             // (y << 16 | x, v0 | v1 << 16 as bf16, v2 | flags, list index) bucketed by row
    const void* Wk; int Kp; const float* bias;               // conv0 weights [64][Kp] bf16 (k = tap*Cpix + c), bias fp32
    int Hc, Wc, Ho, Wo;                                      // conv0 output map, pooled map
    const float *sc, *sh, *sl;                               // BatchNorm0 (scale, shift) table, PReLU0 slope
    void* Out; long ldo;                                     // pooled map -> first 64 channels of dense block 1 (bf16)
    double* part;                                            // per-workgroup statistics partials ([grid][64][2] forward, [grid][64][3] backward)
    EffSrc e;                                                // backward: gradient of the pooled map (G, x = the pooled map, P, Q)
    const float *P0, *Q0;                                    // backward pass 1: BatchNorm0's (P, Q)
    float* slab; long slab_bytes; float* dWk;                // backward pass 1: per-workgroup partial weight gradients, result [64][Kp]
};
long stem_sparse_hit_capacity(int n_img);
long stem_sparse_index_bytes(int n_img, int H, int W);
void stem_sparse_carve(StemSparseArgs& a, char* base);      // points the index arrays into a buffer of stem_sparse_index_bytes()
bool stem_sparse_ok(int mode, int in_ch, int init_ch, int H, int W, int value_mode, long nnz, int n_img, long ldo);
int stem_sparse_stats_grid(const StemSparseArgs& a);
int stem_sparse_pool_grid(const StemSparseArgs& a);
int stem_sparse_bwd_grid(const StemSparseArgs& a);
int stem_sparse_index(const StemSparseArgs& a, hipStream_t st);
int stem_sparse_stats(const StemSparseArgs& a, hipStream_t st);
int stem_sparse_pool(const StemSparseArgs& a, hipStream_t st);
int stem_sparse_bwd(const StemSparseArgs& a, int pass, hipStream_t st);      // pass 0: backward sums; pass 1: conv0 weight gradient

// kernel-layout fp32 weight gradients -> reference OIHW gradients (accumulate)
struct UnpackDesc { const float* src; float* dst; int N, Cin, taps, Kp; int nfast; };   // nfast: src is [k][32]
int unpack_wgrads(const UnpackDesc* d_descs, int n, hipStream_t st);

}  // namespace tcvn
