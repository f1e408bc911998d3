// Launch interface of the SDXL-style embedder kernels (sdxl_kernels.hip): general NHWC convolutions as implicit GEMMs
// (any kernel size / stride / top-left padding, forward, data gradient, weight gradient) and GroupNorm(1 group) + SiLU.
// Reference call sites: transformercvn/network/layers/sdxl_net.py:27-34,41-42 (diffusers.models.vae.Encoder, see
// oracle/sdxl_oracle.py for the restated block definitions).
#pragma once
#include "tcvn_common.h"

namespace tcvn {

// One convolution: Out[(img,ho,wo)][n] = bias[n] + sum_{ky,kx,c} In[img][ho*stride + ky - pad][wo*stride + kx - pad][c] * W[n][c][ky][kx]
// (+ Res[(img,ho,wo)][n]).  Taps that fall outside [0,Hin) x [0,Win) read zero, which also realises diffusers' asymmetric
// F.pad(0,1,0,1) in front of its stride-2 downsampling convolution (pad = 0 here, the bottom/right row is the range check).
struct SConv {
    int mode;                               // MODE_F32 / MODE_BF16: element type of In, Out, Res, dOut, dIn and the packed weights
    int n, Hin, Win, Cin; long lda;         // input NHWC with row stride lda
    int Ho, Wo, Cout, ks, stride, pad;
    int Kp, Kpt;                            // padded K of the forward ([Cout][Kp], k = tap*Cin + c) and transposed ([Cin][Kpt], k = tap*Cout + n) packs
    float* slab; long slab_bytes;           // optional scratch for per-workgroup partial weight gradients (sconv_wgrad fast path)
    const int* hits; long nnz;              // optional: COO list (image, y, x) of the non-zero input pixels (conv_in weight gradient)
    double* stats;                          // optional (forward): [n][2] per-image (sum, sum of squares) of the stored output, pre-zeroed --
                                            // the GroupNorm statistics of the consumer, accumulated by the convolution's epilogue
};
constexpr long kSconvSlabBytes = 512L * (64 * 576 + 64) * 4;     // what the 64 -> 64 weight-gradient kernel asks for

int sconv_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, const void* Res, long ldres, void* Out, long ldo,
              int out_f32, hipStream_t st);
// dIn[(img,y,x)][c] (+)= sum_{ky,kx,n} dOut[img][(y+pad-ky)/stride][(x+pad-kx)/stride][n] * W[n][c][ky][kx]  (exact divisions only)
int sconv_dgrad(const SConv& g, const void* dOut, long lddo, const void* Wt, void* dIn, long lddi, int accumulate, hipStream_t st);
// dWk[n][k] += sum_m dOut[m][n] * a(m, k) (fp32, kernel layout [Cout][Kp]);  dbias[n] += sum_m dOut[m][n]
int sconv_wgrad(const SConv& g, const void* In, const void* dOut, long lddo, float* dWk, float* dbias, hipStream_t st);

// bf16 fast paths for 3x3 / stride 1 / pad 1, 64 -> 64 channels (sdxl_conv3x3.hip): halo patch in LDS, weights in registers
bool sconv3_c64_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, long ldres, const void* Out, long ldo, int out_f32);
int sconv3_c64_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, const void* Res, void* Out, double* stats, hipStream_t st);
bool sconv3_c64_dgrad_ok(const SConv& g, const void* dOut, long lddo, const void* Wt, const void* dIn, long lddi);
int sconv3_c64_dgrad(const SConv& g, const void* dOut, const void* Wt, void* dIn, int accumulate, hipStream_t st);
// general width (channels multiples of 64, <= 512): chunked patch, weight fragments streamed from L2
bool sconv3_g_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, long ldres, const void* Out, long ldo, int out_f32);
bool sconv3_g_fuses_stats(const SConv& g);
int sconv3_g_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, const void* Res, void* Out, double* stats, hipStream_t st);
bool sconv3_g_dgrad_ok(const SConv& g, const void* dOut, long lddo, const void* Wt, const void* dIn, long lddi);
int sconv3_g_dgrad(const SConv& g, const void* dOut, const void* Wt, void* dIn, int accumulate, hipStream_t st);
bool sconv3_g_wgrad_ok(const SConv& g, const void* In, const void* dOut, long lddo);
int sconv3_g_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st);
// 3x3 / stride 2 / pad 0 down-samplers (zeros beyond the map), channels multiples of 64
bool sconv3_s2_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, const void* Out, long ldo, int out_f32);
int sconv3_s2_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, void* Out, double* stats, hipStream_t st);
bool sconv3_s2_dgrad_ok(const SConv& g, const void* dOut, long lddo, const void* Wt, const void* dIn, long lddi);
int sconv3_s2_dgrad(const SConv& g, const void* dOut, const void* Wt, void* dIn, int accumulate, hipStream_t st);
bool sconv3_s2_wgrad_ok(const SConv& g, const void* In, const void* dOut, long lddo);
int sconv3_s2_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st);
// conv_in (3 -> 64, sdxl_stem.hip): gather-MFMA forward, weight gradient from the hit list
bool sconv_in_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, const void* Out, long ldo, int out_f32);
int sconv_in_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, void* Out, double* stats, hipStream_t st);
bool sconv_in_wgrad_ok(const SConv& g, const void* dOut, long lddo);
int sconv_in_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st);
bool sconv3_c64_wgrad_ok(const SConv& g, const void* In, const void* dOut, long lddo);
int sconv3_c64_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st);

// GroupNorm with ONE group (statistics over C*H*W of each image) followed by SiLU (act = 1) or nothing (act = 0).
// stats: [n][2] doubles (sum, sum of squares), zeroed by the caller before gn_stats.
struct GnArgs {
    int mode; const void* X; long ldx; int n, HW, C;
    const float *gamma, *beta; float eps; int act;
    double* stats;
};
int gn_stats(const GnArgs& a, hipStream_t st);
int gn_act(const GnArgs& a, void* Out, long ldo, hipStream_t st);
// Backward: dA = gradient w.r.t. the activated output.  bsum: [n][2] doubles (sum gamma*dU, sum gamma*dU*xhat), zeroed by the
// caller before gn_bwd_reduce, which also accumulates dgamma / dbeta.  gn_bwd_apply then writes / accumulates dX.
int gn_bwd_reduce(const GnArgs& a, const void* dA, long ldda, double* bsum, float* dgamma, float* dbeta, hipStream_t st);
int gn_bwd_apply(const GnArgs& a, const void* dA, long ldda, const double* bsum, void* dX, long lddx, int accumulate, hipStream_t st);

// elementwise helpers
int cast_f32_to(int mode, const float* src, long lds, void* dst, long ldd, long rows, int cols, hipStream_t st);
int add_into(int mode, void* dst, long ldd, const void* src, long lds, long rows, int cols, hipStream_t st);    // dst += src

}  // namespace tcvn
