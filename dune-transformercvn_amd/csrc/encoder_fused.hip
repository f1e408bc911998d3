// Fused transformer-encoder forward: ONE launch for all layers, one 256-thread workgroup per event.
// Reference: torch.nn.TransformerEncoderLayer (post-norm, math attention path) as instantiated at
// transformercvn/network/layers/prong_custom_bert_encoder.py:45-54,57-75.
//
// An event's (1 + P) <= 22 tokens x 128 features live in LDS for the whole stack; every 128 x 128 weight block is streamed
// from L2 through registers into a padded LDS image (prefetched one block ahead, under the previous block's arithmetic and
// the attention / LayerNorm phases) and consumed by exact fp32 FMA chains (k ascending, like the row GEMM it replaces).
// Everything the backward pass reads (qkv, attention probabilities, ctx, LayerNorm xhat / rstd, FFN pre-activation ...) is
// written to the same workspace buffers the unfused kernels of encoder.hip fill, so either forward can feed the backward.
// Dropout masks are the same stateless draws (stream ids 0x6000 + 8 l + {0,1,2,3}, element index of the sequence-major row).
#include "tcvn_encoder.h"

namespace tcvn {

namespace {

constexpr int D = 128, WLD = 132, QLD = 388, RMAX = 11, SMAX = 2 * RMAX;
constexpr float kInvSqrt2 = 0.70710678118654752440f;

__device__ __forceinline__ void w_issue(const float* __restrict__ W, float4 (&reg)[16]) {      // 128 rows x 128 floats, row stride 128
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + 256 * i;
        reg[i] = *reinterpret_cast<const float4*>(W + (idx >> 5) * D + (idx & 31) * 4);
    }
}
__device__ __forceinline__ void w_commit(float* Wl, const float4 (&reg)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = threadIdx.x + 256 * i;
        *reinterpret_cast<float4*>(Wl + (idx >> 5) * WLD + (idx & 31) * 4) = reg[i];
    }
}
// acc[i] = sum_k xin[g + 2 i][k] * Wl[c][k]   (thread: column c = tid & 127, row group g = tid >> 7)
__device__ __forceinline__ void gemm128(const float* xin, int ldx, int nrows, int g, const float* Wl, int c, float (&acc)[RMAX]) {
#pragma unroll
    for (int i = 0; i < RMAX; ++i) acc[i] = 0.f;
    const float* wrow = Wl + c * WLD;
    for (int k = 0; k < D; k += 4) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
        for (int i = 0; i < RMAX; ++i)
            if (i < nrows) {
                const float4 x = *reinterpret_cast<const float4*>(xin + (g + 2 * i) * ldx + k);
                acc[i] = fmaf(x.x, w.x, acc[i]); acc[i] = fmaf(x.y, w.y, acc[i]);
                acc[i] = fmaf(x.z, w.z, acc[i]); acc[i] = fmaf(x.w, w.w, acc[i]);
            }
    }
}

// rows of `y` (LDS, stride D) -> LayerNorm: xhat and rstd to global, gamma * xhat + beta to LDS `out` and global `outg`
__device__ __forceinline__ void ln_rows(const float* y, int S, int B, int b, const float* gamma, const float* beta, float eps, float* out,
                                        float* outg, float* xhg, float* rstdg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int s = wave; s < S; s += 4) {
        const float v0 = y[s * D + lane], v1 = y[s * D + lane + 64];
        const float mean = wave_sum(v0 + v1) / D;
        const float c0 = v0 - mean, c1 = v1 - mean;
        const float rstd = rsqrtf(wave_sum(c0 * c0 + c1 * c1) / D + eps);
        const long t = (long)s * B + b;
        const float h0 = c0 * rstd, h1 = c1 * rstd;
        const float o0 = h0 * gamma[lane] + beta[lane], o1 = h1 * gamma[lane + 64] + beta[lane + 64];
        out[s * D + lane] = o0; out[s * D + lane + 64] = o1;
        if (outg) { outg[t * D + lane] = o0; outg[t * D + lane + 64] = o1; }
        if (xhg) { xhg[t * D + lane] = h0; xhg[t * D + lane + 64] = h1; }
        if (rstdg && lane == 0) rstdg[t] = rstd;
    }
}

template <int HD>
__global__ __launch_bounds__(256, 1) void k_encoder_fwd(const EncFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = a.S, B = a.B, b = blockIdx.x, tid = threadIdx.x;
    float* Wl = lds;                         // [128][WLD]
    float* xin = Wl + D * WLD;               // [S][D]   layer input
    float* qkv = xin + S * D;                // [S][QLD] q | k | v, later the pre-LayerNorm sums
    float* ctx = qkv + S * QLD;              // [S][D]
    float* x1 = ctx + S * D;                 // [S][D]
    float* hb = x1 + S * D;                  // [S][D]   FFN activation
    int* valid = reinterpret_cast<int*>(hb + S * D);
    const int c = tid & 127, g = tid >> 7;
    const int nrows = (S - g + 1) / 2;       // rows g, g+2, ... < S
    constexpr int hd = HD;
    const int H = D / HD;
    const float dp = a.drop_p;
    for (int i = tid; i < S * D; i += 256) {
        const int s = i / D, d = i - s * D;
        xin[i] = a.X0[((long)s * B + b) * D + d];
    }
    if (tid < S) valid[tid] = a.tok_row[b * S + tid] >= 0;
    float4 wreg[16];
    float acc[RMAX];
    w_issue(a.w[0].win, wreg);
    for (int l = 0; l < a.L; ++l) {
        const EncLayerW& W = a.w[l];
        const EncLayerBuf& O = a.buf[l];
        const uint32_t sid = 0x6000u + l * 8;
        // ---- q, k, v projections -------------------------------------------------------------------------------------
#pragma unroll 1
        for (int ch = 0; ch < 3; ++ch) {
            __syncthreads();
            w_commit(Wl, wreg);
            __syncthreads();
            w_issue(ch < 2 ? W.win + (ch + 1) * D * D : W.wo, wreg);
            gemm128(xin, D, nrows, g, Wl, c, acc);
            const float bias = W.bin[ch * D + c];
#pragma unroll
            for (int i = 0; i < RMAX; ++i)
                if (i < nrows) {
                    const int s = g + 2 * i;
                    const float v = acc[i] + bias;
                    qkv[s * QLD + ch * D + c] = v;
                    if (a.save) O.qkv[((long)s * B + b) * 3 * D + ch * D + c] = v;
                }
        }
        __syncthreads();
        // ---- attention: thread (h, s) ----------------------------------------------------------------------------------
        if (tid < H * S) {
            const int s = tid % S, h = tid / S;
            const float scale = rsqrtf((float)hd);
            const float* q = qkv + s * QLD + h * hd;
            float qs[HD];
#pragma unroll
            for (int e = 0; e < HD; ++e) qs[e] = q[e] * scale;
            float sc[SMAX];                                   // fully unrolled below: stays in registers
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < SMAX; ++j) {
                sc[j] = -INFINITY;
                if (j < S) {
                    const float* k = qkv + j * QLD + D + h * hd;
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < HD; ++e) d = fmaf(qs[e], k[e], d);
                    sc[j] = valid[j] ? d : -INFINITY;
                    mx = fmaxf(mx, sc[j]);
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < SMAX; ++j)
                if (j < S) { sc[j] = valid[j] ? expf(sc[j] - mx) : 0.f; sum += sc[j]; }
            const float inv = 1.0f / sum;
            float cx[HD];
#pragma unroll
            for (int e = 0; e < HD; ++e) cx[e] = 0.f;
            float* P = a.save ? O.probs + (((long)b * H + h) * S + s) * S : nullptr;
#pragma unroll
            for (int j = 0; j < SMAX; ++j)
                if (j < S) {
                    float p = sc[j] * inv;
                    if (P) P[j] = p;
                    if (dp > 0.f) p *= drop_scale(dp, a.seed, sid, (((uint64_t)b * H + h) * S + s) * S + j);
                    const float* v = qkv + j * QLD + 2 * D + h * hd;
#pragma unroll
                    for (int e = 0; e < HD; ++e) cx[e] = fmaf(p, v[e], cx[e]);
                }
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                ctx[s * D + h * hd + e] = cx[e];
                if (a.save) O.ctx[((long)s * B + b) * D + h * hd + e] = cx[e];
            }
        }
        // ---- out_proj + residual + LayerNorm 1 ---------------------------------------------------------------------------
        __syncthreads();
        w_commit(Wl, wreg);
        __syncthreads();
        w_issue(W.w1, wreg);
        gemm128(ctx, D, nrows, g, Wl, c, acc);
        {
            const float bias = W.bo[c];
#pragma unroll
            for (int i = 0; i < RMAX; ++i)
                if (i < nrows) {
                    const int s = g + 2 * i;
                    float r = acc[i] + bias;
                    if (dp > 0.f) r *= drop_scale(dp, a.seed, sid + 1, ((uint64_t)s * B + b) * D + c);
                    qkv[s * D + c] = xin[s * D + c] + r;                 // q|k|v are dead: reuse as the [S][D] sum buffer
                }
        }
        __syncthreads();
        ln_rows(qkv, S, B, b, W.g1, W.be1, a.eps, x1, a.save ? O.x1 : nullptr, a.save ? O.xh1 : nullptr, a.save ? O.rstd1 : nullptr);
        // ---- FFN -----------------------------------------------------------------------------------------------------------
        __syncthreads();
        w_commit(Wl, wreg);
        __syncthreads();
        w_issue(W.w2, wreg);
        gemm128(x1, D, nrows, g, Wl, c, acc);
        {
            const float bias = W.b1[c];
#pragma unroll
            for (int i = 0; i < RMAX; ++i)
                if (i < nrows) {
                    const int s = g + 2 * i;
                    const float x = acc[i] + bias;
                    const long t = (long)s * B + b;
                    if (a.save) O.hpre[t * D + c] = x;
                    float y = a.gelu ? 0.5f * x * (1.f + erff(x * kInvSqrt2)) : fmaxf(x, 0.f);
                    if (dp > 0.f) y *= drop_scale(dp, a.seed, sid + 2, (uint64_t)t * D + c);
                    hb[s * D + c] = y;
                    if (a.save) O.hact[t * D + c] = y;
                }
        }
        __syncthreads();
        w_commit(Wl, wreg);
        __syncthreads();
        if (l + 1 < a.L) w_issue(a.w[l + 1].win, wreg);
        gemm128(hb, D, nrows, g, Wl, c, acc);
        {
            const float bias = W.b2[c];
#pragma unroll
            for (int i = 0; i < RMAX; ++i)
                if (i < nrows) {
                    const int s = g + 2 * i;
                    float r = acc[i] + bias;
                    if (dp > 0.f) r *= drop_scale(dp, a.seed, sid + 3, ((uint64_t)s * B + b) * D + c);
                    qkv[s * D + c] = x1[s * D + c] + r;
                }
        }
        __syncthreads();
        ln_rows(qkv, S, B, b, W.g2, W.be2, a.eps, xin, O.xnext, a.save ? O.xh2 : nullptr, a.save ? O.rstd2 : nullptr);
    }
    __syncthreads();
    for (int i = tid; i < S * D; i += 256) {                              // hidden * sequence_mask (:73)
        const int s = i / D, d = i - s * D;
        a.HID[((long)s * B + b) * D + d] = valid[s] ? xin[i] : 0.f;
    }
}

}  // namespace

bool encoder_fused_ok(int S, int Dm, int H, int L, int norm_first) {
    return Dm == D && (H == 4 || H == 8 || H == 16) && S >= 1 && S <= SMAX && L >= 1 && L <= ENC_MAX_LAYERS && !norm_first;
}

int encoder_fused_fwd(const EncFusedArgs& a, hipStream_t st) {
    if (!encoder_fused_ok(a.S, D, a.H, a.L, 0)) return -2;
    const size_t smem = ((size_t)D * WLD + (size_t)a.S * (4 * D + QLD)) * 4 + (size_t)a.S * 4 + 64;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_encoder_fwd<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_encoder_fwd<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_encoder_fwd<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    if (smem > 160 * 1024) return -2;
    if (a.H == 4) hipLaunchKernelGGL(k_encoder_fwd<32>, dim3(a.B), dim3(256), smem, st, a);
    else if (a.H == 8) hipLaunchKernelGGL(k_encoder_fwd<16>, dim3(a.B), dim3(256), smem, st, a);
    else hipLaunchKernelGGL(k_encoder_fwd<8>, dim3(a.B), dim3(256), smem, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
