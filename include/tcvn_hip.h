/* tcvn_hip.h -- C ABI of libtcvn_hip.so: the MI355X (gfx950) implementation of the TransformerCVN hot path.
 *
 * The reference (ayankele/dune-transformercvn) has no native interface of its own: its hot path is the Python call
 * chain NeutrinoFullDenseTrainer.forward -> NeutrinoDenseNetwork.forward -> torch ATen ops
 * (transformercvn/network/trainers/neutrino_full_base_trainer.py:90-116, networks/neutrino_full_base_network.py:87-125,
 * :166-188, layers/dense_net.py:8-167).  This header is the boundary a maintainer binds instead of those ATen calls
 * (see INTEGRATION.md for the ctypes stub).  Plain pointers and sizes only; every pointer is a DEVICE pointer unless
 * named h_*; every function enqueues work on `stream` (a hipStream_t passed as void*) and returns 0 on success or a
 * non-zero hipError_t / negative argument-error code.  No function allocates per call or synchronises the device;
 * workspaces are supplied by the caller.
 */
#ifndef TCVN_HIP_H
#define TCVN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCVN_MODE_F32 0  /* fp32 activations, v_mfma_f32_32x32x2_f32 : parity mode (1e-3 logit gate)      */
#define TCVN_MODE_BF16 1 /* bf16 activations/weights, v_mfma_f32_32x32x16_bf16, fp32 accumulate/statistics */

#define TCVN_SLOT_PARAM 0   /* float tensor with gradient            */
#define TCVN_SLOT_BUFFER 1  /* float tensor without gradient (BN running statistics) */
#define TCVN_SLOT_COUNTER 2 /* int64 scalar (num_batches_tracked); kept by the host, never read on the device */

int tcvn_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * DenseNet embedder (replaces transformercvn/network/layers/dense_net.py:97-167 DenseNet.forward and its autograd)
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct tcvn_densenet_cfg {
    int in_ch;          /* pixel channels (3)                                   */
    int out_dim;        /* embedding width (256 prong / 288 event)              */
    int init_ch;        /* options.initial_pixel_dim                            */
    int growth;         /* options.densenet_growth_rate                         */
    int bn_size;        /* options.densenet_batch_norm_size                     */
    int n_blocks;       /* len(options.densenet_structure) (<= 8)               */
    int layers[8];      /* options.densenet_structure                           */
    int H, W;           /* pixel map shape (400, 280)                           */
    float dropout;      /* options.dropout                                      */
    int mode;           /* TCVN_MODE_*                                          */
} tcvn_densenet_cfg;

typedef struct tcvn_densenet tcvn_densenet; /* opaque plan */

int tcvn_densenet_create(const tcvn_densenet_cfg* cfg, tcvn_densenet** out);
void tcvn_densenet_destroy(tcvn_densenet* p);

/* Parameter slots in reference state_dict order; names are relative to the DenseNet module
 * ("features.conv0.weight", ..., "output_block.relu.weight"). */
int tcvn_densenet_num_slots(const tcvn_densenet* p);
int tcvn_densenet_slot(const tcvn_densenet* p, int i, char* name, int name_cap, int64_t* numel, int* kind);
/* data[i]: device pointer of slot i (fp32, reference layout: conv OIHW); grad[i]: fp32 gradient of the same shape
 * (ignored for buffers/counters; may be NULL for inference-only use). */
int tcvn_densenet_bind(tcvn_densenet* p, void* const* data, void* const* grad);

int64_t tcvn_densenet_workspace_bytes(const tcvn_densenet* p, int n_img, int with_backward);

/* Forward over n_img sparse pixel maps.
 *   coords [nnz,3] int32 (image, y, x), values [nnz, in_ch] fp32 raw pixel values (reference:
 *   trainers/neutrino_full_dense_trainer.py:15-24,46-67: v/255 or log(v+1), optional multiplicative noise, COO->dense).
 *   out [n_img, out_dim] fp32 with row stride out_ld.
 *   train != 0: batch statistics, running-stat update, dropout (seed) -- and the workspace keeps what backward needs;
 *   `coords` must stay valid until tcvn_densenet_backward has run (the stem weight gradient walks the hit list). */
int tcvn_densenet_forward(tcvn_densenet* p, int n_img, const int32_t* coords, const float* values, int64_t nnz,
                          int log_pixels, float noise_std, float* out, int64_t out_ld, void* workspace,
                          int64_t workspace_bytes, int train, uint64_t seed, void* stream);
/* log_pixels: 0 = v/255, 1 = log(v+1), 2 = values are final, 3 = one_hot_pixels (reference :47-52): values [nnz, in_ch/256] hold
 * integers 0..255, pixel channel f*256 + v is set to 1 (no scaling, no noise); conv0 then has in_ch = 256 * value channels. */

/* Backward of the last train-mode forward on the same workspace: d_out [n_img, out_dim] fp32 -> parameter gradients
 * are ACCUMULATED into the bound grad pointers (zero them first). */
int tcvn_densenet_backward(tcvn_densenet* p, int n_img, const float* d_out, int64_t d_out_ld, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* The same backward in slices: dense blocks block_hi ... block_lo (0-based, walked downwards).  block_hi = n_blocks-1 also runs the
 * output block, block_lo = 0 also the stem; the parameter gradients of a slice are final when its call returns (in stream order), so
 * a data-parallel caller can start their all-reduce under the remaining slices.  Slices must be issued from the last block to the
 * first and cover every block exactly once. */
int tcvn_densenet_num_blocks(const tcvn_densenet* p);
int tcvn_densenet_backward_blocks(tcvn_densenet* p, int n_img, const float* d_out, int64_t d_out_ld, void* workspace,
                                  int64_t workspace_bytes, int block_hi, int block_lo, void* stream);

/* Debug/validation taps into the workspace of the last forward: name in {"img","conv0","dense<b>","bottleneck<b>.<l>","condense"},
 * in bf16 mode also the materialised operands "xa<b>.<l>" / "ya<b>.<l>" and the raw regions "raw:wk","raw:tabs","raw:bstat<b>";
 * returns the byte offset into the workspace and the logical NHWC shape + channel stride + element size. */
int tcvn_densenet_tap(const tcvn_densenet* p, int n_img, const char* name, int64_t* byte_off, int* n, int* h, int* w,
                      int* c, int* ld, int* elem_bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * SDXL-style embedder (replaces transformercvn/network/layers/sdxl_net.py:7-42 SDXLNet.forward and its autograd: the
 * diffusers VAE Encoder with block_out_channels [d,d,2d,2d,4d,4d,8d,8d,out], GroupNorm with one group, SiLU, stride-2
 * downsampling with (0,1,0,1) padding, one-token mid-block attention, then Flatten + Linear(out,out); selected by
 * networks/neutrino_full_sdxl_network.py:6-20).  Same calling convention as the DenseNet embedder.  Parity is UNPINNED:
 * diffusers is neither vendored nor pinned by the reference; oracle/sdxl_oracle.py restates the definitions.
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct tcvn_sdxl_cfg {
    int in_ch;          /* pixel channels (3)                                                   */
    int out_dim;        /* embedding width (256 prong / 288 event) = last block's channel count */
    int init_ch;        /* options.initial_pixel_dim (64)                                       */
    int repeat;         /* repeat_block_dim (2)                                                 */
    int num_blocks;     /* num_blocks (4): widths d, 2d, 4d, 8d, each `repeat` times            */
    int H, W;           /* pixel map shape (400, 280); must reduce to 1x1                       */
    int mode;           /* TCVN_MODE_*                                                          */
} tcvn_sdxl_cfg;
typedef struct tcvn_sdxl tcvn_sdxl;
int tcvn_sdxl_create(const tcvn_sdxl_cfg* cfg, tcvn_sdxl** out);
void tcvn_sdxl_destroy(tcvn_sdxl* p);
/* slots in state_dict order, names relative to the SDXLNet module ("encoder.conv_in.weight", ..., "output_layer.1.bias") */
int tcvn_sdxl_num_slots(const tcvn_sdxl* p);
int tcvn_sdxl_slot(const tcvn_sdxl* p, int i, char* name, int name_cap, int64_t* numel, int* kind);
int tcvn_sdxl_bind(tcvn_sdxl* p, void* const* data, void* const* grad);
int64_t tcvn_sdxl_workspace_bytes(const tcvn_sdxl* p, int n_img, int with_backward);
/* Same calling convention as tcvn_densenet_forward / _backward; `coords` must stay valid until tcvn_sdxl_backward has run (the
 * conv_in weight gradient walks the hit list). */
int tcvn_sdxl_forward(tcvn_sdxl* p, int n_img, const int32_t* coords, const float* values, int64_t nnz, int log_pixels,
                      float noise_std, float* out, int64_t out_ld, void* workspace, int64_t workspace_bytes, int train,
                      uint64_t seed, void* stream);
int tcvn_sdxl_backward(tcvn_sdxl* p, int n_img, const float* d_out, int64_t d_out_ld, void* workspace, int64_t workspace_bytes,
                       void* stream);
/* taps: "img", "conv_in", "block<i>" (output of down block i in front of its downsampler), "mid" */
int tcvn_sdxl_tap(const tcvn_sdxl* p, int n_img, const char* name, int64_t* byte_off, int* n, int* h, int* w, int* c, int* ld,
                  int* elem_bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * Token path: combined embedding, transformer encoder, decoders, focal loss
 * (networks/neutrino_full_base_network.py:99-125,184-188; layers/prong_custom_bert_encoder.py:57-75;
 *  layers/prong_decoder.py:15-16; layers/prong_target_decoder.py:34-41; trainers/neutrino_full_base_trainer.py:148-177)
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct tcvn_head_cfg {
    int hidden_dim, heads, n_layers;       /* encoder: d_model, heads, layers; FFN width == hidden_dim (reference quirk) */
    int in_dim;                            /* combined embedding input width (feat + pix + pos = 320)              */
    int event_classes, prong_classes;
    int n_dec; int dec_dims[8];            /* prong decoder widths after each hidden block (64,32,16,8)           */
    int dec_out_in;                        /* in_features of prong_decoder.output_layer (reference quirk value)   */
    int gelu;                              /* 1 = gelu, 0 = relu                                                   */
    int norm_first;
    int dropout_modules;                   /* 1 when options.dropout > 0 (shifts prong_decoder.hidden_layers idx)  */
    float dropout; float gamma; float event_weight;
    int no_linear_bn;                      /* 1 = options.linear_batch_norm False: LinearBlock = Linear(bias)-act-Dropout, decoder blocks without
                                              BatchNorm1d (prong_feature_embedding.py:11-16, encoder.py:13-14); 0 = the shipped option files   */
    int linear_relu;                       /* 1 = options.linear_prelu_activation False: ReLU instead of PReLU (:18-21, :16-19), no slope slots  */
} tcvn_head_cfg;

typedef struct tcvn_head tcvn_head;
int tcvn_head_create(const tcvn_head_cfg* cfg, tcvn_head** out);
void tcvn_head_destroy(tcvn_head* p);
int tcvn_head_num_slots(const tcvn_head* p);
int tcvn_head_slot(const tcvn_head* p, int i, char* name, int name_cap, int64_t* numel, int* kind);
int tcvn_head_bind(tcvn_head* p, void* const* data, void* const* grad);
int64_t tcvn_head_workspace_bytes(const tcvn_head* p, int batch, int max_prongs, int n_prongs);
/* The encoder stack runs as ONE launch forward (one workgroup per event, csrc/encoder_fused.hip) and TWO launches backward (the
 * data-gradient chain + a grouped weight-gradient GEMM) when hidden_dim == 128, heads in {4,8}, 1 + max_prongs <= 22 and <= 8
 * layers; otherwise, or after tcvn_head_set_fused_encoder(p, 0), layer by layer on the row kernels.  Both forwards fill the same
 * workspace buffers, so either backward follows either forward.  On by default. */
void tcvn_head_set_fused_encoder(tcvn_head* p, int on);

/* rows [batch + n_prongs, in_dim] fp32: event rows first, then packed prong rows (already concatenated with the
 * feature / position embeddings).  tok_row [batch, 1+max_prongs] int32: row index of each token or -1 for padding.
 * Outputs: event_logits [batch, event_classes], prong_logits [batch, max_prongs, prong_classes]. */
int tcvn_head_forward(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row,
                      float* event_logits, float* prong_logits, void* workspace, int64_t workspace_bytes, int train,
                      uint64_t seed, void* stream);
/* The three stages of tcvn_head_forward as separate forward-only entry points -- what the reference's sub-modules compute when
 * called on their own (CreateCompiled.ipynb cells 7-8 call network.prong_embedding / .encoder / .event_decoder / .prong_decoder):
 *   embed   rows, tok_row -> tokens [batch, 1+max_prongs, hidden] (batch-major, padding rows zero): the combined LinearBlock +
 *           masked pad of BaseProngEmbedding.forward (networks/neutrino_full_base_network.py:113-125);
 *   encode  tokens [batch, S, hidden], tok_row (only its sign is read: < 0 = padding) -> hidden [S, batch, hidden] (sequence-major,
 *           masked): ProngCustomBertEncoder.forward (layers/prong_custom_bert_encoder.py:57-75);
 *   decode  hidden [S, batch, hidden] -> event_logits [batch, Ce] from token 0 (layers/prong_decoder.py:15-16) and prong_logits
 *           [batch, max_prongs, Cp] from tokens 1.. (layers/prong_target_decoder.py:34-41); either output may be NULL.
 * Workspace: tcvn_head_workspace_bytes(batch, max_prongs, n_prongs) (n_prongs = 0 for encode / decode).  No backward: training
 * goes through tcvn_head_forward / tcvn_head_backward. */
int tcvn_head_embed(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row, float* tokens,
                    void* workspace, int64_t workspace_bytes, int train, uint64_t seed, void* stream);
int tcvn_head_encode(tcvn_head* p, int batch, int max_prongs, const float* tokens, const int32_t* tok_row, float* hidden,
                     void* workspace, int64_t workspace_bytes, int train, uint64_t seed, void* stream);
int tcvn_head_decode(tcvn_head* p, int batch, int max_prongs, const float* hidden, float* event_logits, float* prong_logits,
                     void* workspace, int64_t workspace_bytes, int train, uint64_t seed, void* stream);

/* Row operators behind the holder modules' own forward() (forward only, fp32):
 *   y = x W^T + b (torch.nn.Linear layout; bias may be NULL)                      -- layers/prong_decoder.py:15-16
 *   y = dropout(prelu(batchnorm1d(x)))  with batch statistics + running-stat update when train != 0, running statistics
 *   otherwise; save_mean_rstd: 2*channels floats of scratch                        -- layers/prong_feature_embedding.py:25-33
 *   Option variants of the block (:11-21): gamma == beta == running_mean == running_var == NULL -> no BatchNorm1d (norm = Identity,
 *   options.linear_batch_norm False); slope == NULL -> ReLU (options.linear_prelu_activation False).  Same rules in the backward. */
int tcvn_linear_forward(const float* x, int64_t ldx, const float* weight, const float* bias, float* y, int64_t ldy, int rows,
                        int n_out, int n_in, void* stream);
int tcvn_rows_bn_prelu_forward(const float* x, int64_t ldx, int rows, int channels, const float* gamma, const float* beta,
                               const float* slope, float* running_mean, float* running_var, float* y, int64_t ldy,
                               float* save_mean_rstd, int train, float drop_p, uint64_t seed, uint32_t stream_id, void* stream);

/* Backward of the two row operators (used by the smart-feature MLP, layers/prong_feature_embedding.py:36-78, the only LinearBlocks
 * outside the head plan): dx = dy W (NULL to skip), dweight += dy^T x, dbias += colsum(dy) (either may be NULL);
 * BatchNorm1d(train statistics kept in save_mean_rstd by the forward) + PReLU + dropout backward, parameter gradients accumulated. */
int tcvn_linear_backward(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* weight, float* dx, int64_t lddx,
                         float* dweight, float* dbias, int rows, int n_out, int n_in, void* stream);
int tcvn_rows_bn_prelu_backward(const float* x, int64_t ldx, const float* dy, int64_t lddy, int rows, int channels, const float* gamma,
                                const float* beta, const float* slope, const float* save_mean_rstd, float* dx, int64_t lddx,
                                float* dgamma, float* dbeta, float* dslope, float drop_p, uint64_t seed, uint32_t stream_id, void* stream);

/* Softmax focal loss and its gradient w.r.t. the logits (trainers/neutrino_full_base_trainer.py:148-177):
 * event_targets [batch] int64, prong_targets [batch, max_prongs] int8 (-1 = padding).
 * losses[3] (device, fp32) = {total, event, prong}; accs[6]: {event accuracy, prong accuracy} + 4 floats of scratch;
 * d_event_logits [batch, event_classes], d_prong_logits [batch, max_prongs, prong_classes] = d(total)/d(logits). */
int tcvn_head_loss(tcvn_head* p, int batch, int max_prongs, const float* event_logits, const float* prong_logits,
                   const int64_t* event_targets, const int8_t* prong_targets, float* losses, float* accs,
                   float* d_event_logits, float* d_prong_logits, void* stream);
/* Backward of the last train-mode forward on the same workspace: logit gradients -> d_rows [batch + n_prongs, in_dim];
 * parameter gradients are ACCUMULATED into the bound grad pointers. */
int tcvn_head_backward(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row,
                       const float* d_event_logits, const float* d_prong_logits, float* d_rows, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* Stand-alone softmax focal loss of one logit matrix [rows, classes] (targets int64, < 0 = ignore):
 * out2 = {mean loss, accuracy}; d_logits = weight * d(mean loss)/d(logits). Single workgroup; rows up to a few thousand. */
int tcvn_focal_loss(const float* logits, const int64_t* targets, int rows, int classes, float gamma, float weight,
                    float* d_logits, float* out2, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused optimizer step over flat arenas (replaces torch.optim.AdamW.step over 782 tensors + clip_grad_norm_:
 * trainers/neutrino_base.py:88-152, train.py:140).  All pointers are device pointers of `n` fp32 elements.
 * --------------------------------------------------------------------------------------------------------------- */
/* out[0] = sum of squares of x (fp64 accumulation; partials: scratch of n_partials doubles, n_partials <= 1024 used). */
int tcvn_grad_sumsq(const float* x, int64_t n, double* partials, int n_partials, float* out, void* stream);
/* One AdamW step (decoupled weight decay).  weight_decay[i] < 0 freezes element i (parameter the reference's optimizer never
 * touches).  grad_sumsq (device, may be NULL) and clip > 0 apply the global-norm clip coefficient min(1, clip/(norm+1e-6))
 * to the gradient on the fly (the gradient arena itself is left as is).  step is 1-based. */
int tcvn_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* weight_decay, int64_t n,
                    float lr, float beta1, float beta2, float eps, int64_t step, const float* grad_sumsq, float clip, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Validation aid: the dropout keep-scale (0 or 1/(1-p)) the kernels apply at one site, as a [rows, cols] fp32 tensor.
 * Masks are never stored: forward and backward recompute them from (seed, stream id, element).  kind 0: sites indexed by the
 * row-major element number (output block 0x4000, combined embedding 0x5000, encoder layer l 0x6000+8l+{0 attention
 * probabilities [B,H,S,S], 1 attention output, 2 FFN activation, 3 FFN output} [T,D], prong decoder 0x7000+i); kind 1: the
 * 3x3 convolution outputs of dense block b, layer l (stream id 64 b + l + 1; rows = pixels n*H*W, cols = growth).
 * The reference draws its masks from torch's global generator (layers/dense_net.py:29-40 via nn.Dropout); only the
 * distribution is comparable, so tests feed these masks to the CPU oracle.
 * --------------------------------------------------------------------------------------------------------------- */
int tcvn_dropout_keep(int kind, float p, uint64_t seed, uint32_t stream_id, int64_t rows, int cols, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline leg): when enabled, every convolution launch is bracketed by a HIP event pair on
 * its own stream.  tcvn_profile_get blocks on the record's end event; name is the kernel's label, flops the algorithmic
 * 2*M*N*K of that launch and bytes its algorithmic HBM traffic (operands once, results once).  Off by default; with a
 * filter set only the matching launches pay for their two events, which is cheap enough for the timed region.
 * --------------------------------------------------------------------------------------------------------------- */
/* tcvn_backward_overlap(1): backward runs the 3x3 weight-gradient kernels of the bf16 DenseNet (and the TN GEMMs of 1x1 layers the fused
 * backward kernel does not serve) on a plan-owned side stream beside the data-gradient chain (double-buffered EY, event-released).
 * OFF by default since round 4 (with the 1x1 backward fused into one kernel on the caller's stream the overlap no longer pays: same-box A/B
 * 19.55 against 19.55-19.60 ms/step); on by default in rounds 2-3 (-1.2 % then). */
void tcvn_backward_overlap(int on);
void tcvn_profile_enable(int on);
void tcvn_profile_filter(const char* label_substring); /* NULL or "" = every launch; else only matching kernel labels */
void tcvn_profile_reset(void);
int tcvn_profile_count(void);
int tcvn_profile_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* TCVN_HIP_H */
