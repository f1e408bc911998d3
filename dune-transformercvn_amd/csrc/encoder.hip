// Token-path kernels: token gather, multi-head self-attention with key-padding mask, residual + LayerNorm, GELU,
// masking and the softmax focal loss -- forward and backward, exact fp32.
// Reference: torch.nn.TransformerEncoderLayer (post-norm, seq-first, math attention path) as instantiated at
// transformercvn/network/layers/prong_custom_bert_encoder.py:45-54,57-75; loss at
// transformercvn/network/trainers/neutrino_full_base_trainer.py:148-177.
// Token rows are sequence-major: row t = s * B + b (so the prong decoder's reshape [P*B, D] is the row block [B, T)).
#include "tcvn_encoder.h"

namespace tcvn {

namespace {

constexpr float kInvSqrt2 = 0.70710678118654752440f;

__global__ void k_gather_tokens(const float* C, const int* tok_row, float* X, int B, int S, int D) {
    const int t = blockIdx.x;                       // t = s*B + b
    const int s = t / B, b = t - s * B;
    const int r = tok_row[b * S + s];
    for (int d = threadIdx.x; d < D; d += blockDim.x) X[(long)t * D + d] = r >= 0 ? C[(long)r * D + d] : 0.f;
}
__global__ void k_scatter_tokens_bwd(const float* dX, const int* tok_row, float* dC, int B, int S, int D) {
    const int t = blockIdx.x;
    const int s = t / B, b = t - s * B;
    const int r = tok_row[b * S + s];
    if (r < 0) return;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dC[(long)r * D + d] = dX[(long)t * D + d];   // each row used once
}
__global__ void k_mask_rows(const float* X, const int* tok_row, float* Y, int B, int S, int D) {
    const int t = blockIdx.x;
    const int s = t / B, b = t - s * B;
    const bool keep = tok_row[b * S + s] >= 0;
    for (int d = threadIdx.x; d < D; d += blockDim.x) Y[(long)t * D + d] = keep ? X[(long)t * D + d] : 0.f;
}

// ---- attention: one 64-thread block per (event b, head h); thread s owns query row s --------------------------
constexpr int MAXS = 64, MAXHD = 32;
__global__ __launch_bounds__(64) void k_attn_fwd(const AttnArgs a) {
    __shared__ float q[MAXS][MAXHD + 1], k[MAXS][MAXHD + 1], v[MAXS][MAXHD + 1];
    __shared__ int valid[MAXS];
    const int b = blockIdx.x, h = blockIdx.y, s = threadIdx.x;
    const int D = a.H * a.hd;
    for (int i = threadIdx.x; i < a.S * a.hd; i += 64) {
        const int ss = i / a.hd, d = i - ss * a.hd;
        const float* row = a.qkv + ((long)ss * a.B + b) * 3 * D + h * a.hd + d;
        q[ss][d] = row[0]; k[ss][d] = row[D]; v[ss][d] = row[2 * D];
    }
    if (s < a.S) valid[s] = a.tok_row[b * a.S + s] >= 0;
    __syncthreads();
    if (s >= a.S) return;
    const float scale = rsqrtf((float)a.hd);
    float sc[MAXS];
    float mx = -INFINITY;
    for (int j = 0; j < a.S; ++j) {
        float d = 0.f;
        for (int e = 0; e < a.hd; ++e) d = fmaf(q[s][e] * scale, k[j][e], d);
        sc[j] = valid[j] ? d : -INFINITY;
        mx = fmaxf(mx, sc[j]);
    }
    float sum = 0.f;
    for (int j = 0; j < a.S; ++j) { sc[j] = valid[j] ? expf(sc[j] - mx) : 0.f; sum += sc[j]; }
    const float inv = 1.0f / sum;
    float* P = a.probs + (((long)b * a.H + h) * a.S + s) * a.S;
    float ctx[MAXHD];
    for (int e = 0; e < a.hd; ++e) ctx[e] = 0.f;
    for (int j = 0; j < a.S; ++j) {
        float p = sc[j] * inv;
        P[j] = p;
        if (a.drop_p > 0.f) p *= drop_scale(a.drop_p, a.seed, a.stream_id, (((uint64_t)b * a.H + h) * a.S + s) * a.S + j);
        for (int e = 0; e < a.hd; ++e) ctx[e] = fmaf(p, v[j][e], ctx[e]);
    }
    float* o = a.ctx + ((long)s * a.B + b) * D + h * a.hd;
    for (int e = 0; e < a.hd; ++e) o[e] = ctx[e];
}

__global__ __launch_bounds__(64) void k_attn_bwd(const AttnBwdArgs a) {
    __shared__ float q[MAXS][MAXHD + 1], k[MAXS][MAXHD + 1], v[MAXS][MAXHD + 1], dc[MAXS][MAXHD + 1];
    __shared__ float dS[MAXS][MAXS + 1], Pd[MAXS][MAXS + 1];
    const int b = blockIdx.x, h = blockIdx.y, s = threadIdx.x;
    const int D = a.H * a.hd;
    for (int i = threadIdx.x; i < a.S * a.hd; i += 64) {
        const int ss = i / a.hd, d = i - ss * a.hd;
        const float* row = a.qkv + ((long)ss * a.B + b) * 3 * D + h * a.hd + d;
        q[ss][d] = row[0]; k[ss][d] = row[D]; v[ss][d] = row[2 * D];
        dc[ss][d] = a.dctx[((long)ss * a.B + b) * D + h * a.hd + d];
    }
    __syncthreads();
    const float scale = rsqrtf((float)a.hd);
    if (s < a.S) {
        const float* P = a.probs + (((long)b * a.H + h) * a.S + s) * a.S;
        float dot = 0.f;
        float dp[MAXS];
        for (int j = 0; j < a.S; ++j) {
            float g = 0.f;
            for (int e = 0; e < a.hd; ++e) g = fmaf(dc[s][e], v[j][e], g);        // d(P_drop)
            float m = 1.f;
            if (a.drop_p > 0.f) m = drop_scale(a.drop_p, a.seed, a.stream_id, (((uint64_t)b * a.H + h) * a.S + s) * a.S + j);
            Pd[s][j] = P[j] * m;
            dp[j] = g * m;                                                          // dP
            dot = fmaf(dp[j], P[j], dot);
        }
        for (int j = 0; j < a.S; ++j) dS[s][j] = P[j] * (dp[j] - dot);            // softmax backward (masked keys: P = 0)
    }
    __syncthreads();
    if (s >= a.S) return;
    float* o = a.dqkv + ((long)s * a.B + b) * 3 * D + h * a.hd;
    for (int e = 0; e < a.hd; ++e) {
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < a.S; ++j) {
            dq = fmaf(dS[s][j], k[j][e], dq);
            dk = fmaf(dS[j][s], q[j][e], dk);
            dv = fmaf(Pd[j][s], dc[j][e], dv);
        }
        o[e] = dq * scale; o[D + e] = dk * scale; o[2 * D + e] = dv;
    }
}

// ---- y = LayerNorm(x + drop(r)) : one wave per row ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_ln_fwd(const AddLnArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= a.T) return;
    float v[8];
    const int per = (a.D + 63) / 64;
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        const int d = lane + 64 * i;
        float x = 0.f;
        if (d < a.D) {
            float r = a.R[(long)t * a.D + d];
            if (a.drop_p > 0.f) r *= drop_scale(a.drop_p, a.seed, a.stream_id, (uint64_t)t * a.D + d);
            x = a.X[(long)t * a.D + d] + r;
        }
        v[i] = x; s += x;
    }
    const float mean = wave_sum(s) / a.D;
    float q = 0.f;
    for (int i = 0; i < per; ++i) { const int d = lane + 64 * i; if (d < a.D) { const float c = v[i] - mean; q += c * c; } }
    const float rstd = rsqrtf(wave_sum(q) / a.D + a.eps);
    for (int i = 0; i < per; ++i) {
        const int d = lane + 64 * i;
        if (d < a.D) {
            const float xh = (v[i] - mean) * rstd;
            a.XH[(long)t * a.D + d] = xh;
            a.Y[(long)t * a.D + d] = xh * a.gamma[d] + a.beta[d];
        }
    }
    if (lane == 0) a.rstd[t] = rstd;
}
// dS = LN backward; dX (+)= dS ; dR = drop * dS ; per-row contributions to dgamma/dbeta written as partials
__global__ __launch_bounds__(256) void k_add_ln_bwd(const AddLnBwdArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= a.T) return;
    const int per = (a.D + 63) / 64;
    float g[8], xh[8];
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < per; ++i) {
        const int d = lane + 64 * i;
        g[i] = 0.f; xh[i] = 0.f;
        if (d < a.D) {
            xh[i] = a.XH[(long)t * a.D + d];
            g[i] = a.dY[(long)t * a.D + d] * a.gamma[d];
            s1 += g[i]; s2 += g[i] * xh[i];
        }
    }
    s1 = wave_sum(s1) / a.D; s2 = wave_sum(s2) / a.D;
    const float rstd = a.rstd[t];
    for (int i = 0; i < per; ++i) {
        const int d = lane + 64 * i;
        if (d < a.D) {
            const float ds = rstd * (g[i] - s1 - xh[i] * s2);
            a.dX[(long)t * a.D + d] = ds;
            float dr = ds;
            if (a.drop_p > 0.f) dr *= drop_scale(a.drop_p, a.seed, a.stream_id, (uint64_t)t * a.D + d);
            a.dR[(long)t * a.D + d] = dr;
        }
    }
}
// column reductions for LayerNorm affine grads: dgamma[d] += sum_t dY*xhat ; dbeta[d] += sum_t dY
__global__ __launch_bounds__(256) void k_ln_param_grads(const float* dY, const float* XH, int T, int D, float* dgamma, float* dbeta) {
    __shared__ double sa[16][17], sb[16][17];                // 16 columns x 16 row lanes per workgroup
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, d = blockIdx.x * 16 + cl;
    double a = 0, b = 0;
    if (d < D)
        for (int t = rl; t < T; t += 16) { const float g = dY[(long)t * D + d]; a += (double)g * XH[(long)t * D + d]; b += g; }
    sa[rl][cl] = a; sb[rl][cl] = b;
    __syncthreads();
    if (threadIdx.x < 16 && d < D) {
        double x = 0, y = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { x += sa[i][cl]; y += sb[i][cl]; }
        dgamma[d] += (float)x; dbeta[d] += (float)y;
    }
}

__global__ void k_act_fwd(const float* X, float* Y, long n, int gelu, float drop_p, uint64_t seed, uint32_t stream_id) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = X[i];
    float y = gelu ? 0.5f * x * (1.f + erff(x * kInvSqrt2)) : fmaxf(x, 0.f);
    if (drop_p > 0.f) y *= drop_scale(drop_p, seed, stream_id, (uint64_t)i);
    Y[i] = y;
}
__global__ void k_act_bwd(const float* X, const float* dY, float* dX, long n, int gelu, float drop_p, uint64_t seed, uint32_t stream_id) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = X[i];
    float g = dY[i];
    if (drop_p > 0.f) g *= drop_scale(drop_p, seed, stream_id, (uint64_t)i);
    const float d = gelu ? 0.5f * (1.f + erff(x * kInvSqrt2)) + x * 0.3989422804014327f * expf(-0.5f * x * x) : (x > 0.f ? 1.f : 0.f);
    dX[i] = g * d;
}

// ---- softmax focal loss (single block): loss = mean_i( -log p_t * (1 - p_t)^gamma ) over rows with target >= 0 ----
// dlogits are written already scaled by `weight` / count; out[0] = loss, out[1] = accuracy
template <typename TT>
__global__ __launch_bounds__(256) void k_focal(const float* logits, const TT* targets, int rows, int C, float gamma, float weight,
                                               float* dlogits, float* out) {
    __shared__ double red[256][2];
    __shared__ int cnt_s[256];
    __shared__ int total;
    int cnt = 0;
    for (int r = threadIdx.x; r < rows; r += 256) cnt += targets[r] >= 0;
    cnt_s[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { int c = 0; for (int i = 0; i < 256; ++i) c += cnt_s[i]; total = c; }
    __syncthreads();
    const float invn = total > 0 ? 1.0f / total : 0.f;
    double loss = 0, acc = 0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const int t = (int)targets[r];
        float* dl = dlogits + (long)r * C;
        if (t < 0) { for (int c = 0; c < C; ++c) dl[c] = 0.f; continue; }
        const float* z = logits + (long)r * C;
        float mx = z[0]; int am = 0;
        for (int c = 1; c < C; ++c) if (z[c] > mx) { mx = z[c]; am = c; }
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(z[c] - mx);
        const float logpt = z[t] - mx - logf(sum);
        const float pt = expf(logpt);
        float dldlogpt;                                    // d loss_i / d log p_t
        if (gamma == 0.f) { loss += -logpt; dldlogpt = -1.f; }
        else {
            const float om = fmaxf(1.f - pt, 0.f);
            const float w = powf(om, gamma);
            loss += -logpt * w;
            // d/dlogpt [ -logpt * (1-pt)^g ] = -(1-pt)^g + logpt * g * (1-pt)^(g-1) * pt
            dldlogpt = -w + (om > 0.f ? logpt * gamma * powf(om, gamma - 1.f) * pt : 0.f);
        }
        acc += am == t;
        for (int c = 0; c < C; ++c) {
            const float p = expf(z[c] - mx) / sum;
            dl[c] = weight * invn * dldlogpt * ((c == t ? 1.f : 0.f) - p);
        }
    }
    red[threadIdx.x][0] = loss; red[threadIdx.x][1] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double l = 0, c = 0;
        for (int i = 0; i < 256; ++i) { l += red[i][0]; c += red[i][1]; }
        out[0] = (float)(l * invn); out[1] = (float)(c * invn);
    }
}

__global__ void k_permute_rows(const float* in, float* out, int B, int P, int C, int to_bm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P * C) return;
    const int c = i % C, r = i / C;
    if (to_bm) { const int b = r / P, p = r - b * P; out[i] = in[((long)p * B + b) * C + c]; }
    else { const int p = r / B, b = r - p * B; out[i] = in[((long)b * P + p) * C + c]; }
}
// pre-norm residual: y = x + drop(r)   /   its backward branch: dr = drop * dy
__global__ void k_add_drop(const float* x, const float* r, float* y, long n, float drop_p, uint64_t seed, uint32_t stream_id) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = r[i];
    if (drop_p > 0.f) v *= drop_scale(drop_p, seed, stream_id, (uint64_t)i);
    y[i] = x[i] + v;
}
__global__ void k_mul_drop(const float* dy, float* dr, long n, float drop_p, uint64_t seed, uint32_t stream_id) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = dy[i];
    if (drop_p > 0.f) v *= drop_scale(drop_p, seed, stream_id, (uint64_t)i);
    dr[i] = v;
}
__global__ void k_add_inplace(float* dst, const float* src, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

}  // namespace

int permute_rows(const float* in, float* out, int B, int P, int C, int to_bm, hipStream_t st) {
    if (B * P * C <= 0) return 0;
    hipLaunchKernelGGL(k_permute_rows, dim3(cdiv((long)B * P * C, 256)), dim3(256), 0, st, in, out, B, P, C, to_bm);
    TCVN_LAUNCH_CHECK(); return 0;
}
int add_drop(const float* x, const float* r, float* y, long n, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st) {
    hipLaunchKernelGGL(k_add_drop, dim3(cdiv(n, 256)), dim3(256), 0, st, x, r, y, n, drop_p, seed, sid);
    TCVN_LAUNCH_CHECK(); return 0;
}
int mul_drop(const float* dy, float* dr, long n, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st) {
    hipLaunchKernelGGL(k_mul_drop, dim3(cdiv(n, 256)), dim3(256), 0, st, dy, dr, n, drop_p, seed, sid);
    TCVN_LAUNCH_CHECK(); return 0;
}
int add_inplace(float* dst, const float* src, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_add_inplace, dim3(cdiv(n, 256)), dim3(256), 0, st, dst, src, n);
    TCVN_LAUNCH_CHECK(); return 0;
}
int gather_tokens(const float* C, const int* tok_row, float* X, int B, int S, int D, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_tokens, dim3(B * S), dim3(128), 0, st, C, tok_row, X, B, S, D);
    TCVN_LAUNCH_CHECK(); return 0;
}
int scatter_tokens_bwd(const float* dX, const int* tok_row, float* dC, int B, int S, int D, hipStream_t st) {
    hipLaunchKernelGGL(k_scatter_tokens_bwd, dim3(B * S), dim3(128), 0, st, dX, tok_row, dC, B, S, D);
    TCVN_LAUNCH_CHECK(); return 0;
}
int mask_rows(const float* X, const int* tok_row, float* Y, int B, int S, int D, hipStream_t st) {
    hipLaunchKernelGGL(k_mask_rows, dim3(B * S), dim3(128), 0, st, X, tok_row, Y, B, S, D);
    TCVN_LAUNCH_CHECK(); return 0;
}
int attn_fwd(const AttnArgs& a, hipStream_t st) {
    if (a.S > MAXS || a.hd > MAXHD) { fprintf(stderr, "tcvn: attention supports S<=%d, head_dim<=%d\n", MAXS, MAXHD); return -2; }
    hipLaunchKernelGGL(k_attn_fwd, dim3(a.B, a.H), dim3(64), 0, st, a);
    TCVN_LAUNCH_CHECK(); return 0;
}
int attn_bwd(const AttnBwdArgs& a, hipStream_t st) {
    if (a.S > MAXS || a.hd > MAXHD) return -2;
    hipLaunchKernelGGL(k_attn_bwd, dim3(a.B, a.H), dim3(64), 0, st, a);
    TCVN_LAUNCH_CHECK(); return 0;
}
int add_ln_fwd(const AddLnArgs& a, hipStream_t st) {
    if (a.D > 512) return -2;
    hipLaunchKernelGGL(k_add_ln_fwd, dim3(cdiv(a.T, 4)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK(); return 0;
}
int add_ln_bwd(const AddLnBwdArgs& a, hipStream_t st) {
    if (a.D > 512) return -2;
    hipLaunchKernelGGL(k_add_ln_bwd, dim3(cdiv(a.T, 4)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ln_param_grads, dim3(cdiv(a.D, 16)), dim3(256), 0, st, a.dY, a.XH, a.T, a.D, a.dgamma, a.dbeta);
    TCVN_LAUNCH_CHECK(); return 0;
}
int act_fwd(const float* X, float* Y, long n, int gelu, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st) {
    hipLaunchKernelGGL(k_act_fwd, dim3(cdiv(n, 256)), dim3(256), 0, st, X, Y, n, gelu, drop_p, seed, sid);
    TCVN_LAUNCH_CHECK(); return 0;
}
int act_bwd(const float* X, const float* dY, float* dX, long n, int gelu, float drop_p, uint64_t seed, uint32_t sid, hipStream_t st) {
    hipLaunchKernelGGL(k_act_bwd, dim3(cdiv(n, 256)), dim3(256), 0, st, X, dY, dX, n, gelu, drop_p, seed, sid);
    TCVN_LAUNCH_CHECK(); return 0;
}
int focal_i64(const float* logits, const int64_t* targets, int rows, int C, float gamma, float weight, float* dlogits, float* out,
              hipStream_t st) {
    hipLaunchKernelGGL(k_focal<int64_t>, dim3(1), dim3(256), 0, st, logits, targets, rows, C, gamma, weight, dlogits, out);
    TCVN_LAUNCH_CHECK(); return 0;
}
int focal_i8(const float* logits, const int8_t* targets, int rows, int C, float gamma, float weight, float* dlogits, float* out,
             hipStream_t st) {
    hipLaunchKernelGGL(k_focal<int8_t>, dim3(1), dim3(256), 0, st, logits, targets, rows, C, gamma, weight, dlogits, out);
    TCVN_LAUNCH_CHECK(); return 0;
}

}  // namespace tcvn
