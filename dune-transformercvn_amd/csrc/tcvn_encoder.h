// Launch interface of the token-path kernels (encoder.hip).
#pragma once
#include "tcvn_common.h"

namespace tcvn {

int gather_tokens(const float* C, const int* tok_row, float* X, int B, int S, int D, hipStream_t st);
int scatter_tokens_bwd(const float* dX, const int* tok_row, float* dC, int B, int S, int D, hipStream_t st);
int mask_rows(const float* X, const int* tok_row, float* Y, int B, int S, int D, hipStream_t st);
// out[(b*P+p)*C + c] = in[(p*B+b)*C + c]  (to_batch_major != 0) or the inverse
int permute_rows(const float* in, float* out, int B, int P, int C, int to_batch_major, hipStream_t st);

struct AttnArgs {
    const float* qkv; const int* tok_row; float* probs; float* ctx;
    int B, S, H, hd; float drop_p; uint64_t seed; uint32_t stream_id;
};
int attn_fwd(const AttnArgs& a, hipStream_t st);
struct AttnBwdArgs {
    const float* qkv; const float* probs; const float* dctx; float* dqkv;
    int B, S, H, hd; float drop_p; uint64_t seed; uint32_t stream_id;
};
int attn_bwd(const AttnBwdArgs& a, hipStream_t st);

struct AddLnArgs {
    const float* X; const float* R; const float *gamma, *beta; float* Y; float* XH; float* rstd;
    int T, D; float eps; float drop_p; uint64_t seed; uint32_t stream_id;
};
int add_ln_fwd(const AddLnArgs& a, hipStream_t st);
struct AddLnBwdArgs {
    const float* dY; const float* XH; const float* rstd; const float* gamma; float* dX; float* dR; float *dgamma, *dbeta;
    int T, D; float drop_p; uint64_t seed; uint32_t stream_id;
};
int add_ln_bwd(const AddLnBwdArgs& a, hipStream_t st);

int act_fwd(const float* X, float* Y, long n, int gelu, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st);
int act_bwd(const float* X, const float* dY, float* dX, long n, int gelu, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st);
int add_inplace(float* dst, const float* src, long n, hipStream_t st);
int add_drop(const float* x, const float* r, float* y, long n, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st);   // y = x + drop(r)
int mul_drop(const float* dy, float* dr, long n, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st);                  // dr = drop * dy

// Fused encoder forward (encoder_fused.hip): all layers in one launch, one workgroup per event.  Weight / buffer pointers per layer.
constexpr int ENC_MAX_LAYERS = 8;
struct EncLayerW { const float *win, *bin, *wo, *bo, *w1, *b1, *w2, *b2, *g1, *be1, *g2, *be2; };
struct EncLayerBuf { float *qkv, *probs, *ctx, *xh1, *rstd1, *x1, *hpre, *hact, *xh2, *rstd2, *xnext; };
struct EncFusedArgs {
    const float* X0; const int* tok_row; float* HID;
    int B, S, H, L, gelu, save; float eps, drop_p; uint64_t seed;
    EncLayerW w[ENC_MAX_LAYERS]; EncLayerBuf buf[ENC_MAX_LAYERS];
};
bool encoder_fused_ok(int S, int D, int H, int L, int norm_first);
int encoder_fused_fwd(const EncFusedArgs& a, hipStream_t st);
// Fused backward: the chain kernel's per-layer outputs (GEMM operands of the weight gradients) and the grouped weight-gradient launch
struct EncLayerGrad { float *dqkv, *dao, *dhp, *df; };      // [T][3D], [T][D], [T][D], [T][D]
struct EncFusedBwdArgs {
    const float* dY; float* dX; float* lnp;                  // dY / dX [T][D]; lnp [B][L][4][D] per-event LayerNorm sums (g1, b1, g2, b2)
    int B, S, H, L, gelu; float drop_p; uint64_t seed;
    EncLayerW w[ENC_MAX_LAYERS]; EncLayerBuf buf[ENC_MAX_LAYERS]; EncLayerGrad g[ENC_MAX_LAYERS];
};
struct EncWgradJob { const float* dY; int ldy; const float* X; float* dW; float* db; int tiles; };     // tiles = N / 32
struct EncWgradArgs {
    int T, B, L, n_jobs, n_tiles; const float* lnp;
    float* ln_dst[ENC_MAX_LAYERS][4];
    EncWgradJob job[4 * ENC_MAX_LAYERS];
};
int encoder_fused_bwd(const EncFusedBwdArgs& a, const EncWgradArgs& w, hipStream_t st);

int focal_i64(const float* logits, const int64_t* targets, int rows, int C, float gamma, float weight, float* dlogits, float* out,
              hipStream_t st);
int focal_i8(const float* logits, const int8_t* targets, int rows, int C, float gamma, float weight, float* dlogits, float* out,
             hipStream_t st);

}  // namespace tcvn
