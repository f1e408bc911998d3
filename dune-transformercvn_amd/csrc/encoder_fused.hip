// Fused transformer encoder: ONE launch for the forward of all layers and TWO for the backward (the
// data-gradient chain + a grouped weight-gradient GEMM), one 256-thread workgroup per event.
// Reference: torch.nn.TransformerEncoderLayer (post-norm, math attention path) as instantiated at
// transformercvn/network/layers/prong_custom_bert_encoder.py:45-54,57-75, and its autograd.
//
// An event's (1 + P) <= 22 tokens x 128 features live in LDS for the whole stack; every weight block is streamed from L2
// through registers into a padded LDS image (prefetched one block ahead, under the previous block's arithmetic and the
// attention / LayerNorm phases) and consumed by fp32 MFMAs (v_mfma_f32_16x16x4_f32: exact fp32 products and accumulation; round 5 --
// before that fp32 FMA chains, thread (c, g) = output column c, token rows g, g+2, ...).  The padded row count SP = 2 NR is a template
// parameter (rows beyond the sequence are zero padding in LDS); a wave owns 32 output columns x one or two 16-row tiles.  Dropout masks are the same stateless draws as in encoder.hip (stream ids 0x6000 + 8 l +
// {0,1,2,3}, element index of the sequence-major row), evaluated by rolled loops into an LDS mask buffer.
// Everything the backward reads is written to the workspace buffers the unfused kernels of encoder.hip fill, so either forward
// can feed either backward.
#include "tcvn_encoder.h"

namespace tcvn {

namespace {

constexpr int D = 128, WLD = 132, QLD = 388, SMAX = 22, HR = 64;
constexpr float kInvSqrt2 = 0.70710678118654752440f;

template <int NV>
__device__ __forceinline__ void w_issue(const float* __restrict__ W, f32x4 (&reg)[NV]) {      // NV*8 rows x 128 floats, row stride 128
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + 256 * i;
        reg[i] = *reinterpret_cast<const f32x4*>(W + (idx >> 5) * D + (idx & 31) * 4);
    }
}
template <int NV>
__device__ __forceinline__ void w_commit(float* Wl, const f32x4 (&reg)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + 256 * i;
        *reinterpret_cast<f32x4*>(Wl + (idx >> 5) * WLD + (idx & 31) * 4) = reg[i];
    }
}
// Forward product on the matrix pipe (round 5; rounds 2-4: fp32 FMA chains of a thread per (column, row pair)): OUT[r][c] = sum_k A[r][k] * Wl[c][k] for the wave's 32 output columns and MT 16-row tiles of
// token rows, `v_mfma_f32_16x16x4_f32` (exact fp32 products, fp32 accumulation: a k-ordered fmaf chain in another k order than gemm128's).
// Lane (r = lane & 15, q = lane >> 4) holds A[r][k] and B[k][r] for ONE k per instruction; it reads its operands as float4 at
// k = 16 kk + 4 q .. + 3 and feeds component j to the j-th instruction of the group -- any assignment of k values to (instruction, q) works
// as long as A and B agree.  A's pitch must not be a multiple of 64 floats (XLD / QLD: the 16 rows of a read then fall on distinct
// banks; at pitch 128 they are a 16-way conflict).  Rows beyond the padded sequence read whatever LDS holds (or zero past the
// allocation): their results are never used.
constexpr int XLD = 132;
template <int MT>
__device__ __forceinline__ void gemm128_mfma(const float* A, int lda, const float* Wl, f32x4 (&acc)[MT][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = A + r * lda + 4 * q;
    const float* bp = Wl + (wave * 32 + r) * WLD + 4 * q;
#pragma unroll 2
    for (int kk = 0; kk < D / 16; ++kk) {
        f32x4 av[MT], bv[2];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const f32x4*>(ap + m * 16 * lda + 16 * kk);
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bp + t * 16 * WLD + 16 * kk);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][j], bv[t][j], acc[m][t], 0, 0, 0);
    }
}
// visit the wave's results: f(row, tile-of-columns t, column, value) for every row < rows  (D layout: column = lane & 15, row = 4 (lane >> 4) + register)
template <int MT, typename F>
__device__ __forceinline__ void mfma_results(const f32x4 (&acc)[MT][2], int rows, F f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = m * 16 + 4 * (lane >> 4) + j;
                if (row < rows) f(row, t, wave * 32 + t * 16 + (lane & 15), acc[m][t][j]);
            }
}
// transposed product over one 64-row half block of the weight image, on the matrix pipe: acc[r][c] += sum_{n < 64} dy[r][n0 + n] * Wl[n][c] for the wave's 32 columns c (A = dy: float4 per lane along n,
// pitch not a multiple of 64 floats; B = four b32 reads down the rows of the weight image: bank (16 q + column) mod 64, conflict-free)
template <int MT>
__device__ __forceinline__ void gemm_t64_mfma(const float* dy, int ldy, int n0, const float* Wl, f32x4 (&acc)[MT][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const float* ap = dy + r * ldy + n0 + 4 * q;
    const float* bp = Wl + 4 * q * WLD + wave * 32 + r;
#pragma unroll
    for (int kk = 0; kk < HR / 16; ++kk) {
        f32x4 av[MT];
        float bv[2][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const f32x4*>(ap + m * 16 * ldy + 16 * kk);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[t][j] = bp[(16 * kk + j) * WLD + t * 16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][j], bv[t][j], acc[m][t], 0, 0, 0);
    }
}
template <int MT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][2]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// mask[s][c] = dropout keep-scale of element (t = s*B + b, c) of a [T][D] site
__device__ __forceinline__ void fill_mask(float* mask, int S, int B, int b, float dp, uint64_t seed, uint32_t sid) {
    // a thread draws four consecutive columns of a row = the four words of one Philox call (the row's first index (s*B + b)*128 is a multiple of 4)
#pragma unroll 1
    for (int i = threadIdx.x; i < S * (D / 4); i += 256) {
        const int s = i >> 5, c = (i & 31) * 4;
        DropCache dc{~0ull, {0, 0, 0, 0}};
        const uint64_t e0 = ((uint64_t)s * B + b) * D + c;
        f32x4 m;
        m.x = drop_scale_cached(dc, dp, seed, sid, e0); m.y = drop_scale_cached(dc, dp, seed, sid, e0 + 1);
        m.z = drop_scale_cached(dc, dp, seed, sid, e0 + 2); m.w = drop_scale_cached(dc, dp, seed, sid, e0 + 3);
        *reinterpret_cast<f32x4*>(mask + s * D + c) = m;
    }
}

// rows of `y` (LDS, stride D) -> LayerNorm: xhat and rstd to global, gamma * xhat + beta to LDS `out` (stride ldo) and global `outg`
__device__ __forceinline__ void ln_rows(const float* y, int S, int B, int b, const float* gamma, const float* beta, float eps, float* out, int ldo,
                                        float* outg, float* xhg, float* rstdg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float g0 = gamma[lane], g1 = gamma[lane + 64], b0 = beta[lane], b1 = beta[lane + 64];
#pragma unroll 1
    for (int s = wave; s < S; s += 4) {
        const float v0 = y[s * D + lane], v1 = y[s * D + lane + 64];
        const float mean = wave_sum(v0 + v1) / D;
        const float c0 = v0 - mean, c1 = v1 - mean;
        const float rstd = rsqrtf(wave_sum(c0 * c0 + c1 * c1) / D + eps);
        const long t = (long)s * B + b;
        const float h0 = c0 * rstd, h1 = c1 * rstd;
        const float o0 = h0 * g0 + b0, o1 = h1 * g1 + b1;
        out[s * ldo + lane] = o0; out[s * ldo + lane + 64] = o1;
        if (outg) { outg[t * D + lane] = o0; outg[t * D + lane + 64] = o1; }
        if (xhg) { xhg[t * D + lane] = h0; xhg[t * D + lane + 64] = h1; }
        if (rstdg && lane == 0) rstdg[t] = rstd;
    }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <int HD, int NR>
__global__ __launch_bounds__(256, 1) void k_encoder_fwd(const EncFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SP = 2 * NR, H = D / HD;          // padded rows
    const int S = a.S, B = a.B, b = blockIdx.x, tid = threadIdx.x;
    float* Wl = lds;                         // [128][WLD]
    float* xin = Wl + D * WLD;               // [SP][XLD] layer input               (xin, ctx, x1, hb are MFMA A operands: pitch XLD)
    float* qkv = xin + SP * XLD;             // [SP][QLD] q | k | v, later the pre-LayerNorm sums ([SP][D])
    float* ctx = qkv + SP * QLD;             // [SP][XLD]
    float* x1 = ctx + SP * XLD;              // [SP][XLD]
    float* hb = x1 + SP * XLD;               // [SP][XLD] FFN activation
    float* mask = hb + SP * XLD;             // [SP][D]   dropout keep-scales of the site being applied
    float* sc = x1;                          // [H][S][S + 1] attention probabilities: x1 | hb are dead while attention runs (their
                                             // padding rows may keep scores afterwards: rows >= S never reach a real row)
    int* valid = reinterpret_cast<int*>(mask + SP * D);
    static_assert(H * SMAX * (SMAX + 1) <= 2 * 2 * 11 * D, "score buffer must fit x1 | hb at the largest bucket");
    constexpr int MT = SP > 16 ? 2 : 1;      // 16-row MFMA tiles of token rows
    const int bcol = (tid >> 6) * 32 + (tid & 15);           // this lane's output column in column tile t is bcol + 16 t
    const float dp = a.drop_p;
    for (int i = tid; i < SP * D; i += 256) {
        const int s = i >> 7, d = i & 127;
        xin[s * XLD + d] = s < S ? a.X0[((long)s * B + b) * D + d] : 0.f;
        ctx[s * XLD + d] = 0.f; x1[s * XLD + d] = 0.f; hb[s * XLD + d] = 0.f; mask[i] = 1.f;
    }
    if (tid < SMAX) valid[tid] = tid < S ? a.tok_row[b * S + tid] >= 0 : 0;
    f32x4 wreg[16];
    f32x4 acc[MT][2];
    w_issue<16>(a.w[0].win, wreg);
#pragma unroll 1
    for (int l = 0; l < a.L; ++l) {
        const EncLayerW W = a.w[l];
        const EncLayerBuf O = a.buf[l];
        const uint32_t sid = 0x6000u + l * 8;
        // ---- q, k, v projections -------------------------------------------------------------------------------------
#pragma unroll 1
        for (int ch = 0; ch < 3; ++ch) {
            __syncthreads();
            w_commit<16>(Wl, wreg);
            __syncthreads();
            w_issue<16>(ch < 2 ? W.win + (ch + 1) * D * D : W.wo, wreg);
            gemm128_mfma<MT>(xin, XLD, Wl, acc);
            const float bias[2] = {W.bin[ch * D + bcol], W.bin[ch * D + bcol + 16]};
            mfma_results<MT>(acc, SP, [&](int s, int t, int c, float v0) {
                const float v = v0 + bias[t];
                qkv[s * QLD + ch * D + c] = v;
                if (s < S && a.save) O.qkv[((long)s * B + b) * 3 * D + ch * D + c] = v;
            });
        }
        __syncthreads();
        // ---- attention: thread (h, s); probabilities through LDS ----------------------------------------------------------
        if (tid < H * S) {
            const int s = tid % S, h = tid / S;
            const float scale = rsqrtf((float)HD);
            float qs[HD];
#pragma unroll
            for (int e = 0; e < HD; ++e) qs[e] = qkv[s * QLD + h * HD + e] * scale;
            float* p = sc + (h * S + s) * (S + 1);
            float mx = -INFINITY;
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                const float* kk = qkv + j * QLD + D + h * HD;
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < HD; ++e) d = fmaf(qs[e], kk[e], d);
                d = valid[j] ? d : -INFINITY;
                p[j] = d;
                mx = fmaxf(mx, d);
            }
            float sum = 0.f;
#pragma unroll 1
            for (int j = 0; j < S; ++j) { const float e = valid[j] ? expf(p[j] - mx) : 0.f; p[j] = e; sum += e; }
            const float inv = 1.0f / sum;
            float cx[HD];
#pragma unroll
            for (int e = 0; e < HD; ++e) cx[e] = 0.f;
            float* P = a.save ? O.probs + (((long)b * H + h) * S + s) * S : nullptr;
            DropCache dc{~0ull, {0, 0, 0, 0}};
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                float pr = p[j] * inv;
                if (P) P[j] = pr;
                if (dp > 0.f) pr *= drop_scale_cached(dc, dp, a.seed, sid, (((uint64_t)b * H + h) * S + s) * S + j);
                const float* v = qkv + j * QLD + 2 * D + h * HD;
#pragma unroll
                for (int e = 0; e < HD; ++e) cx[e] = fmaf(pr, v[e], cx[e]);
            }
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                ctx[s * XLD + h * HD + e] = cx[e];
                if (a.save) O.ctx[((long)s * B + b) * D + h * HD + e] = cx[e];
            }
        }
        if (dp > 0.f) fill_mask(mask, S, B, b, dp, a.seed, sid + 1);
        // ---- out_proj + residual + LayerNorm 1 ---------------------------------------------------------------------------
        __syncthreads();
        w_commit<16>(Wl, wreg);
        __syncthreads();
        w_issue<16>(W.w1, wreg);
        gemm128_mfma<MT>(ctx, XLD, Wl, acc);
        {
            const float bias[2] = {W.bo[bcol], W.bo[bcol + 16]};
            mfma_results<MT>(acc, SP, [&](int s, int t, int c, float v0) {
                const float r = (v0 + bias[t]) * mask[s * D + c];
                qkv[s * D + c] = xin[s * XLD + c] + r;               // q|k|v are dead: reuse as the [SP][D] sum buffer
            });
        }
        __syncthreads();
        ln_rows(qkv, S, B, b, W.g1, W.be1, a.eps, x1, XLD, a.save ? O.x1 : nullptr, a.save ? O.xh1 : nullptr, a.save ? O.rstd1 : nullptr);
        if (dp > 0.f) fill_mask(mask, S, B, b, dp, a.seed, sid + 2);
        // ---- FFN -----------------------------------------------------------------------------------------------------------
        __syncthreads();
        w_commit<16>(Wl, wreg);
        __syncthreads();
        w_issue<16>(W.w2, wreg);
        gemm128_mfma<MT>(x1, XLD, Wl, acc);
        {
            const float bias[2] = {W.b1[bcol], W.b1[bcol + 16]};
            mfma_results<MT>(acc, S, [&](int s, int t, int c, float v0) {
                const float x = v0 + bias[t];
                const float y = (a.gelu ? 0.5f * x * (1.f + erff(x * kInvSqrt2)) : fmaxf(x, 0.f)) * mask[s * D + c];
                hb[s * XLD + c] = y;
                if (a.save) { const long tt = (long)s * B + b; O.hpre[tt * D + c] = x; O.hact[tt * D + c] = y; }
            });
        }
        __syncthreads();
        if (dp > 0.f) fill_mask(mask, S, B, b, dp, a.seed, sid + 3);
        w_commit<16>(Wl, wreg);
        __syncthreads();
        if (l + 1 < a.L) w_issue<16>(a.w[l + 1].win, wreg);
        gemm128_mfma<MT>(hb, XLD, Wl, acc);
        {
            const float bias[2] = {W.b2[bcol], W.b2[bcol + 16]};
            mfma_results<MT>(acc, SP, [&](int s, int t, int c, float v0) {
                const float r = (v0 + bias[t]) * mask[s * D + c];
                qkv[s * D + c] = x1[s * XLD + c] + r;
            });
        }
        __syncthreads();
        ln_rows(qkv, S, B, b, W.g2, W.be2, a.eps, xin, XLD, O.xnext, a.save ? O.xh2 : nullptr, a.save ? O.rstd2 : nullptr);
    }
    __syncthreads();
    for (int i = tid; i < S * D; i += 256) {                              // hidden * sequence_mask (:73)
        const int s = i >> 7, d = i & 127;
        a.HID[((long)s * B + b) * D + d] = valid[s] ? xin[s * XLD + d] : 0.f;
    }
}

// =====================================================================================================================
// backward chain
// =====================================================================================================================
// LayerNorm backward of the rows in `dy` (LDS): ds -> `res` (LDS, the residual branch) and mask * ds -> `br` (LDS) + `brg` (global).
// Per-event column sums of dy*xhat and dy go to lnp[0][c], lnp[1][c] (global, this event's slice) via `red` (LDS, 4 x 256 floats).
__device__ __forceinline__ void ln_bwd_rows(const float* dy, int ldy, int S, int B, int b, const float* gamma, const float* xhg, const float* rstdg,
                                            float* res, int ldr, float* br, int ldb, float* brg, const float* mask, float* red, float* lnp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float ga0 = 0.f, ga1 = 0.f, be0 = 0.f, be1 = 0.f;
    const float g0 = gamma[lane], g1 = gamma[lane + 64];
#pragma unroll 1
    for (int s = wave; s < S; s += 4) {
        const long t = (long)s * B + b;
        const float y0 = dy[s * ldy + lane], y1 = dy[s * ldy + lane + 64];
        const float h0 = xhg[t * D + lane], h1 = xhg[t * D + lane + 64];
        ga0 = fmaf(y0, h0, ga0); ga1 = fmaf(y1, h1, ga1); be0 += y0; be1 += y1;
        const float a0 = y0 * g0, a1 = y1 * g1;
        const float s1 = wave_sum(a0 + a1) / D, s2 = wave_sum(a0 * h0 + a1 * h1) / D;
        const float rstd = rstdg[t];
        const float d0 = rstd * (a0 - s1 - h0 * s2), d1 = rstd * (a1 - s1 - h1 * s2);
        res[s * ldr + lane] = d0; res[s * ldr + lane + 64] = d1;
        const float r0 = d0 * mask[s * D + lane], r1 = d1 * mask[s * D + lane + 64];
        br[s * ldb + lane] = r0; br[s * ldb + lane + 64] = r1;
        brg[t * D + lane] = r0; brg[t * D + lane + 64] = r1;
    }
    red[wave * 256 + lane] = ga0; red[wave * 256 + 64 + lane] = ga1; red[wave * 256 + 128 + lane] = be0; red[wave * 256 + 192 + lane] = be1;
    __syncthreads();
    const int i = threadIdx.x;                           // 0..127: dgamma column, 128..255: dbeta column
    lnp[i] = red[i] + red[256 + i] + red[512 + i] + red[768 + i];
}

template <int HD, int NR>
__global__ __launch_bounds__(256, 1) void k_encoder_bwd(const EncFusedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SP = 2 * NR, H = D / HD;
    const int S = a.S, B = a.B, b = blockIdx.x, tid = threadIdx.x;
    float* Wl = lds;                         // [64][WLD]; doubles as the dS / Pd exchange and the LayerNorm reduction scratch
    float* bA = Wl + HR * WLD;               // [SP][XLD]   (bA, bC, bDQ are MFMA A operands: pitches XLD / QLD)
    float* bB = bA + SP * XLD;               // [SP][D]
    float* bC = bB + SP * D;                 // [SP][XLD]
    float* bQ = bC + SP * XLD;               // [SP][QLD] saved q | k | v
    float* bDQ = bQ + SP * QLD;              // [SP][QLD] d(q | k | v)
    float* mask = bDQ + SP * QLD;            // [SP][D]
    constexpr int MT = SP > 16 ? 2 : 1;
    const float dp = a.drop_p;
    for (int i = tid; i < SP * D; i += 256) {
        const int s = i >> 7, d = i & 127;
        bA[s * XLD + d] = s < S ? a.dY[((long)s * B + b) * D + d] : 0.f;
        bB[i] = 0.f; bC[s * XLD + d] = 0.f; mask[i] = 1.f;
    }
    for (int i = tid; i < SP * QLD; i += 256) bDQ[i] = 0.f;
    f32x4 wreg[8];
    f32x4 acc[MT][2];
    w_issue<8>(a.w[a.L - 1].w2, wreg);
    __syncthreads();
#pragma unroll 1
    for (int l = a.L - 1; l >= 0; --l) {
        const EncLayerW W = a.w[l];
        const EncLayerBuf O = a.buf[l];
        const EncLayerGrad G = a.g[l];
        const uint32_t sid = 0x6000u + l * 8;
        float* lnp = a.lnp + ((long)b * a.L + l) * 4 * D;
        // ---- LayerNorm 2 backward: bA = d x_{l+1} -> bB = d x1 (residual), bC = d f (dropped) -----------------------------------
        if (dp > 0.f) { fill_mask(mask, S, B, b, dp, a.seed, sid + 3); __syncthreads(); }
        ln_bwd_rows(bA, XLD, S, B, b, W.g2, O.xh2, O.rstd2, bB, D, bC, XLD, G.df, mask, Wl, lnp + 2 * D);
        // ---- d hact = d f W2 ; d hpre = d hact * drop * act'(hpre) -> bA -------------------------------------------------------
        __syncthreads();
        if (dp > 0.f) fill_mask(mask, S, B, b, dp, a.seed, sid + 2);
        zero_acc<MT>(acc);
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            __syncthreads();
            w_commit<8>(Wl, wreg);
            __syncthreads();
            w_issue<8>(hf == 0 ? W.w2 + HR * D : W.w1, wreg);
            gemm_t64_mfma<MT>(bC, XLD, hf * HR, Wl, acc);
        }
        mfma_results<MT>(acc, S, [&](int s, int, int k, float v0) {
            const long t = (long)s * B + b;
            const float x = O.hpre[t * D + k];
            const float gr = v0 * mask[s * D + k];
            const float dact = a.gelu ? 0.5f * (1.f + erff(x * kInvSqrt2)) + x * 0.3989422804014327f * expf(-0.5f * x * x) : (x > 0.f ? 1.f : 0.f);
            const float v = gr * dact;
            bA[s * XLD + k] = v;
            G.dhp[t * D + k] = v;
        });
        // ---- d x1 += d hpre W1 -------------------------------------------------------------------------------------------------
        zero_acc<MT>(acc);
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            __syncthreads();
            w_commit<8>(Wl, wreg);
            __syncthreads();
            w_issue<8>(hf == 0 ? W.w1 + HR * D : W.wo, wreg);
            gemm_t64_mfma<MT>(bA, XLD, hf * HR, Wl, acc);
        }
        mfma_results<MT>(acc, SP, [&](int s, int, int k, float v) { bB[s * D + k] += v; });
        if (dp > 0.f) fill_mask(mask, S, B, b, dp, a.seed, sid + 1);
        __syncthreads();
        // ---- LayerNorm 1 backward: bB = d x1 -> bA = d x_l (residual), bC = d attention-out (dropped) ---------------------------
        ln_bwd_rows(bB, D, S, B, b, W.g1, O.xh1, O.rstd1, bA, XLD, bC, XLD, G.dao, mask, Wl, lnp);
        // ---- d ctx = d ao Wo -> bB ; meanwhile the saved q | k | v come in ---------------------------------------------------
#pragma unroll 1
        for (int i = tid; i < S * 3 * D; i += 256) {
            const int s = i / (3 * D), d = i - s * 3 * D;
            bQ[s * QLD + d] = O.qkv[((long)s * B + b) * 3 * D + d];
        }
        zero_acc<MT>(acc);
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
            __syncthreads();
            w_commit<8>(Wl, wreg);
            __syncthreads();
            w_issue<8>(hf == 0 ? W.wo + HR * D : W.win, wreg);
            gemm_t64_mfma<MT>(bC, XLD, hf * HR, Wl, acc);
        }
        mfma_results<MT>(acc, SP, [&](int s, int, int k, float v) { bB[s * D + k] = v; });
        __syncthreads();
        // ---- attention backward: thread (h, s); dS and the dropped probabilities are exchanged through the W image ---------------
        float* xS = Wl;                                   // [H][SMAX][SMAX + 1]
        float* xP = Wl + H * SMAX * (SMAX + 1);           // [H][SMAX][SMAX + 1]   (2 * 8 * 22 * 23 floats <= 64 * 132)
        if (tid < H * S) {
            const int s = tid % S, h = tid / S;
            const float* P = O.probs + (((long)b * H + h) * S + s) * S;
            float dc[HD];
#pragma unroll
            for (int e = 0; e < HD; ++e) dc[e] = bB[s * D + h * HD + e];
            float* rS = xS + (h * SMAX + s) * (SMAX + 1);
            float* rP = xP + (h * SMAX + s) * (SMAX + 1);
            float dot = 0.f;
            DropCache dcc{~0ull, {0, 0, 0, 0}};
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                const float* v = bQ + j * QLD + 2 * D + h * HD;
                float gsum = 0.f;
#pragma unroll
                for (int e = 0; e < HD; ++e) gsum = fmaf(dc[e], v[e], gsum);
                float m = 1.f;
                if (dp > 0.f) m = drop_scale_cached(dcc, dp, a.seed, sid, (((uint64_t)b * H + h) * S + s) * S + j);
                const float pj = P[j];
                rP[j] = pj * m;
                const float dpj = gsum * m;
                rS[j] = dpj;                                     // dP for now
                dot = fmaf(dpj, pj, dot);
            }
#pragma unroll 1
            for (int j = 0; j < S; ++j) rS[j] = P[j] * (rS[j] - dot);
        }
        __syncthreads();
        if (tid < H * S) {
            const int s = tid % S, h = tid / S;
            const float scale = rsqrtf((float)HD);
            float dq[HD], dk[HD], dv[HD];
#pragma unroll
            for (int e = 0; e < HD; ++e) { dq[e] = 0.f; dk[e] = 0.f; dv[e] = 0.f; }
#pragma unroll 1
            for (int j = 0; j < S; ++j) {
                const float dsj = xS[(h * SMAX + s) * (SMAX + 1) + j], dst = xS[(h * SMAX + j) * (SMAX + 1) + s];
                const float pdt = xP[(h * SMAX + j) * (SMAX + 1) + s];
                const float* kj = bQ + j * QLD + D + h * HD;
                const float* qj = bQ + j * QLD + h * HD;
                const float* dcj = bB + j * D + h * HD;
#pragma unroll
                for (int e = 0; e < HD; ++e) {
                    dq[e] = fmaf(dsj, kj[e], dq[e]);
                    dk[e] = fmaf(dst, qj[e], dk[e]);
                    dv[e] = fmaf(pdt, dcj[e], dv[e]);
                }
            }
            float* og = G.dqkv + ((long)s * B + b) * 3 * D + h * HD;
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                const float q = dq[e] * scale, kk = dk[e] * scale;
                bDQ[s * QLD + h * HD + e] = q; bDQ[s * QLD + D + h * HD + e] = kk; bDQ[s * QLD + 2 * D + h * HD + e] = dv[e];
                og[e] = q; og[D + e] = kk; og[2 * D + e] = dv[e];
            }
        }
        // ---- d x_l = bA + d qkv Win (six half blocks) ------------------------------------------------------------------------------
        zero_acc<MT>(acc);
#pragma unroll 1
        for (int hf = 0; hf < 6; ++hf) {
            __syncthreads();
            w_commit<8>(Wl, wreg);
            __syncthreads();
            if (hf < 5) w_issue<8>(W.win + (hf + 1) * HR * D, wreg);
            else if (l > 0) w_issue<8>(a.w[l - 1].w2, wreg);
            gemm_t64_mfma<MT>(bDQ, QLD, hf * HR, Wl, acc);
        }
        mfma_results<MT>(acc, SP, [&](int s, int, int k, float v) { bA[s * XLD + k] += v; });
        __syncthreads();
    }
    for (int i = tid; i < S * D; i += 256) {
        const int s = i >> 7, d = i & 127;
        a.dX[((long)s * B + b) * D + d] = bA[s * XLD + d];
    }
}

// ---- grouped weight gradients: dW[n][k] += sum_t dY[t][n] X[t][k], db[n] += sum_t dY[t][n]; LayerNorm parameter sums ----------------
__global__ __launch_bounds__(256) void k_encoder_wgrad(const EncWgradArgs a) {
    __shared__ float Xs[32][D];
    __shared__ __attribute__((aligned(16))) float Ys[32][2][16];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk >= a.n_tiles) {                                        // LayerNorm parameter gradients: sum the per-event partials
        const int idx = (blk - a.n_tiles) * 256 + tid;              // (layer, which of 4, column)
        if (idx < a.L * 4 * D) {
            const int l = idx / (4 * D), r = idx - l * 4 * D;
            float s = 0.f;
            for (int b = 0; b < a.B; ++b) s += a.lnp[((long)b * a.L + l) * 4 * D + r];
            a.ln_dst[l][r / D][r % D] += s;
        }
        return;
    }
    int j = 0;
    while (blk >= a.job[j].tiles) { blk -= a.job[j].tiles; ++j; }
    const EncWgradJob J = a.job[j];
    const int n0 = blk * 32;
    const int k = tid & 127, g = tid >> 7;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    // the next 32 token rows travel in registers while the current ones multiply (round 5: one memory round trip per tile was the kernel:
    // nine dependent trips for T = 288, 72 us)
    float xr[16], yr[4];
    auto issue = [&](int t0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = tid + 256 * q, t = i >> 7, c = i & 127;
            xr[q] = t0 + t < a.T ? J.X[(long)(t0 + t) * D + c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q, t = i >> 5, n = i & 31;
            yr[q] = t0 + t < a.T ? J.dY[(long)(t0 + t) * J.ldy + n0 + n] : 0.f;
        }
    };
    issue(0);
    for (int t0 = 0; t0 < a.T; t0 += 32) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int i = tid + 256 * q; Xs[i >> 7][i & 127] = xr[q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = tid + 256 * q, n = i & 31; Ys[i >> 5][n & 1][n >> 1] = yr[q]; }
        __syncthreads();
        if (t0 + 32 < a.T) issue(t0 + 32);
#pragma unroll 4
        for (int t = 0; t < 32; ++t) {
            const float x = Xs[t][k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 y = *reinterpret_cast<const f32x4*>(&Ys[t][g][q * 4]);
                acc[q * 4 + 0] = fmaf(y.x, x, acc[q * 4 + 0]); acc[q * 4 + 1] = fmaf(y.y, x, acc[q * 4 + 1]);
                acc[q * 4 + 2] = fmaf(y.z, x, acc[q * 4 + 2]); acc[q * 4 + 3] = fmaf(y.w, x, acc[q * 4 + 3]);
            }
        }
        if (tid < 32)
            for (int t = 0; t < 32; ++t) bsum += Ys[t][tid & 1][tid >> 1];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) J.dW[(long)(n0 + g + 2 * i) * D + k] += acc[i];
    if (tid < 32) J.db[n0 + tid] += bsum;
}

// rows per thread for a sequence of S tokens: the smallest instantiated bucket
int rows_bucket(int S) { const int r = (S + 1) / 2; return r <= 3 ? 3 : r <= 5 ? 5 : r <= 8 ? 8 : 11; }
// dynamic LDS of the forward kernel.  The MFMA products read 16 (32 for the largest bucket) rows of their A operand whatever the padded
// sequence length: the last operand buffer (hb) is followed by the mask and the validity words; pad so that those reads stay inside.
size_t bwd_smem(int SP) {               // backward: the last A operand (bDQ) is followed by the mask
    const int MT = SP > 16 ? 2 : 1;
    long over = (long)16 * MT * QLD - ((long)SP * QLD + (long)SP * D);
    if (over < 0) over = 0;
    return ((size_t)HR * WLD + (size_t)SP * (2 * XLD + 2 * D + 2 * QLD) + (size_t)over) * 4 + 64;
}
size_t fwd_smem(int SP) {
    const int MT = SP > 16 ? 2 : 1;
    long over = (long)16 * MT * XLD - ((long)SP * XLD + (long)SP * D + SMAX);
    if (over < 0) over = 0;
    return ((size_t)D * WLD + (size_t)SP * (D + 4 * XLD + QLD) + (size_t)over) * 4 + SMAX * 4 + 64;
}

template <int HD, int NR>
int launch_fwd(const EncFusedArgs& a, hipStream_t st) {
    constexpr int SP = 2 * NR, H = D / HD;
    const size_t smem = fwd_smem(SP);
    if (H * a.S * (a.S + 1) > 2 * SP * D) return -2;
    static bool attr[16] = {};                  // per device: the attribute belongs to the device's copy of the function
    int dev = 0;
    TCVN_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_encoder_fwd<HD, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    if (smem > 160 * 1024) return -2;
    hipLaunchKernelGGL((k_encoder_fwd<HD, NR>), dim3(a.B), dim3(256), smem, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}
template <int HD, int NR>
int launch_bwd(const EncFusedBwdArgs& a, hipStream_t st) {
    constexpr int SP = 2 * NR;
    const size_t smem = bwd_smem(SP);
    static bool attr[16] = {};
    int dev = 0;
    TCVN_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr[dev]) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_encoder_bwd<HD, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 16) attr[dev] = true;
    }
    if (smem > 160 * 1024) return -2;
    hipLaunchKernelGGL((k_encoder_bwd<HD, NR>), dim3(a.B), dim3(256), smem, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Every condition the launchers below depend on: a shape this returns true for is never rejected later (the callers pick the
// layer-by-layer kernels on false; there is no fallback after a launch has been attempted).
bool encoder_fused_ok(int S, int Dm, int H, int L, int norm_first) {
    if (!(Dm == D && (H == 4 || H == 8) && S >= 1 && S <= SMAX && L >= 1 && L <= ENC_MAX_LAYERS && !norm_first)) return false;
    const int SP = 2 * rows_bucket(S);
    if ((long)H * S * (S + 1) > 2L * SP * D) return false;                          // score rows alias the token buffers (forward)
    if (2L * H * SMAX * (SMAX + 1) > (long)HR * WLD) return false;                  // dS / dropped-P exchange aliases the weight image (backward)
    const size_t smem_f = fwd_smem(SP);
    const size_t smem_b = bwd_smem(SP);
    return smem_f <= 160 * 1024 && smem_b <= 160 * 1024;
}

int encoder_fused_fwd(const EncFusedArgs& a, hipStream_t st) {
    if (!encoder_fused_ok(a.S, D, a.H, a.L, 0)) return -2;
    const int nr = rows_bucket(a.S);
    if (a.H == 8) return nr == 3 ? launch_fwd<16, 3>(a, st) : nr == 5 ? launch_fwd<16, 5>(a, st) : nr == 8 ? launch_fwd<16, 8>(a, st) : launch_fwd<16, 11>(a, st);
    return nr == 3 ? launch_fwd<32, 3>(a, st) : nr == 5 ? launch_fwd<32, 5>(a, st) : nr == 8 ? launch_fwd<32, 8>(a, st) : launch_fwd<32, 11>(a, st);
}

int encoder_fused_bwd(const EncFusedBwdArgs& a, const EncWgradArgs& w, hipStream_t st) {
    if (!encoder_fused_ok(a.S, D, a.H, a.L, 0)) return -2;
    if (2 * a.H * SMAX * (SMAX + 1) > HR * WLD) return -2;
    const int nr = rows_bucket(a.S);
    int rc;
    if (a.H == 8) rc = nr == 3 ? launch_bwd<16, 3>(a, st) : nr == 5 ? launch_bwd<16, 5>(a, st) : nr == 8 ? launch_bwd<16, 8>(a, st) : launch_bwd<16, 11>(a, st);
    else rc = nr == 3 ? launch_bwd<32, 3>(a, st) : nr == 5 ? launch_bwd<32, 5>(a, st) : nr == 8 ? launch_bwd<32, 8>(a, st) : launch_bwd<32, 11>(a, st);
    if (rc) return rc;
    const int ln_blocks = cdiv((long)a.L * 4 * D, 256);
    hipLaunchKernelGGL(k_encoder_wgrad, dim3(w.n_tiles + ln_blocks), dim3(256), 0, st, w);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
