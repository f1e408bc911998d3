// Small fp32 row kernels for everything behind the CNN embedders (a few hundred rows x <= 512 features):
// strided SGEMM (Linear forward / dX / dW), BatchNorm1d + PReLU + Dropout forward/backward.
// Reference call sites: layers/prong_feature_embedding.py:25-33 (LinearBlock), layers/dense_net.py:157-162
// (DenseNet.output_block), layers/encoder.py:10-24 + layers/prong_target_decoder.py:34-41, layers/prong_decoder.py:15-16.
// FLOPs here are < 0.03 % of a step; these kernels are written for exact fp32 and determinism.
#include "tcvn_rows.h"

namespace tcvn {

namespace {

// C[i][j] (+)= alpha * sum_k A[i*sai + k*sak] * B[j*sbj + k*sbk] + bias[j]
__global__ __launch_bounds__(256) void k_sgemm(const SgemmArgs a) {
    // 16 x 16 outputs per workgroup, K in panels of 128: all 16 loads of a thread are in flight together and a panel costs two
    // barriers (the token-path GEMMs are latency-bound: M = 288 rows, K <= 320 -- 16-wide panels spent 8 HBM round trips per launch)
    constexpr int KP = 128;
    __shared__ float As[16][KP + 1], Bs[16][KP + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    const int jb = blockIdx.x * 16 + ty;
    float acc = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += KP) {
        float va[KP / 16], vb[KP / 16];
#pragma unroll
        for (int q = 0; q < KP / 16; ++q) {
            const int k = k0 + tx + 16 * q;
            va[q] = (i < a.M && k < a.K) ? a.A[(long)i * a.sai + (long)k * a.sak] : 0.f;
            vb[q] = (jb < a.N && k < a.K) ? a.B[(long)jb * a.sbj + (long)k * a.sbk] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < KP / 16; ++q) { As[ty][tx + 16 * q] = va[q]; Bs[ty][tx + 16 * q] = vb[q]; }
        __syncthreads();
        const int kn = a.K - k0 < KP ? a.K - k0 : KP;
        if (kn == KP) {
#pragma unroll 16
            for (int k = 0; k < KP; ++k) acc = fmaf(As[ty][k], Bs[tx][k], acc);
        } else {
            for (int k = 0; k < kn; ++k) acc = fmaf(As[ty][k], Bs[tx][k], acc);
        }
        __syncthreads();
    }
    if (i < a.M && j < a.N) {
        float v = a.alpha * acc + (a.bias ? a.bias[j] : 0.f);
        float* c = a.C + (long)i * a.ldc + j;
        *c = a.accumulate ? *c + v : v;
    }
}

// ---- BatchNorm1d (+PReLU/ReLU, +Dropout) over R rows -------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_bn_fwd(const RowsBnArgs a) {
    __shared__ double red[16][16][2];
    __shared__ float s_mean[16], s_rstd[16];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;      // 16 columns x 16 row lanes per workgroup
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < a.C;
    if (a.no_norm) {                                          // LinearBlock without BatchNorm1d: activation + dropout only
        if (!ok) return;
        const float sl = a.slope ? a.slope[c] : 0.f;
        for (int r = rg; r < a.R; r += 16) {
            float z = prelu(a.X[(long)r * a.ldx + c], sl);
            if (a.train && a.drop_p > 0.f) z *= drop_scale(a.drop_p, a.seed, a.stream_id, (uint64_t)r * a.C + c);
            a.Y[(long)r * a.ldy + c] = z;
        }
        return;
    }
    if (a.train) {
        double s1 = 0, s2 = 0;
        if (ok)
            for (int r = rg; r < a.R; r += 16) { const double v = a.X[(long)r * a.ldx + c]; s1 += v; s2 += v * v; }
        red[rg][cl][0] = s1; red[rg][cl][1] = s2;
        __syncthreads();
        if (rg == 0 && ok) {
            double x = 0, y = 0;
            for (int g = 0; g < 16; ++g) { x += red[g][cl][0]; y += red[g][cl][1]; }
            const double mean = x / a.R;
            double var = y / a.R - mean * mean;
            if (var < 0) var = 0;
            s_mean[cl] = (float)mean;
            s_rstd[cl] = (float)(1.0 / sqrt(var + (double)a.eps));
            if (a.save_mean) { a.save_mean[c] = s_mean[cl]; a.save_rstd[c] = s_rstd[cl]; }
            if (a.running_mean) {
                const double unb = a.R > 1 ? var * a.R / (double)(a.R - 1) : var;
                a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
                a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
            }
        }
        __syncthreads();
    } else if (rg == 0 && ok) {
        s_mean[cl] = a.running_mean[c];
        s_rstd[cl] = 1.0f / sqrtf(a.running_var[c] + a.eps);
    }
    if (!a.train) __syncthreads();
    if (!ok) return;
    const float mean = s_mean[cl], rstd = s_rstd[cl];
    const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
    const float sl = a.slope ? a.slope[c] : 0.f;
    for (int r = rg; r < a.R; r += 16) {
        const float u = (a.X[(long)r * a.ldx + c] - mean) * rstd * g + b;
        float z = prelu(u, sl);
        if (a.train && a.drop_p > 0.f) z *= drop_scale(a.drop_p, a.seed, a.stream_id, (uint64_t)r * a.C + c);
        a.Y[(long)r * a.ldy + c] = z;
    }
}

__global__ __launch_bounds__(256) void k_rows_bn_bwd(const RowsBnBwdArgs a) {
    __shared__ double red[16][16][3];
    __shared__ float s_db[16], s_dg[16];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;      // 16 columns x 16 row lanes per workgroup
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < a.C;
    if (a.no_norm) {                                          // no BatchNorm1d in the block: element-wise backward + the slope's column sum
        const float sl = (ok && a.slope) ? a.slope[c] : 0.f;
        double s3 = 0;
        if (ok)
            for (int r = rg; r < a.R; r += 16) {
                const float u = a.X[(long)r * a.ldx + c];
                float dz = a.dY[(long)r * a.lddy + c];
                if (a.drop_p > 0.f) dz *= drop_scale(a.drop_p, a.seed, a.stream_id, (uint64_t)r * a.C + c);
                if (a.dX) a.dX[(long)r * a.lddx + c] = u > 0.f ? dz : sl * dz;
                s3 += u > 0.f ? 0.f : dz * u;
            }
        red[rg][cl][0] = s3;
        __syncthreads();
        if (rg == 0 && ok && a.dslope) {
            double z = 0;
            for (int q = 0; q < 16; ++q) z += red[q][cl][0];
            a.dslope[c] += (float)z;
        }
        return;
    }
    const float mean = ok ? a.save_mean[c] : 0.f, rstd = ok ? a.save_rstd[c] : 0.f;
    const float g = (ok && a.gamma) ? a.gamma[c] : 1.f, b = (ok && a.beta) ? a.beta[c] : 0.f;
    const float sl = (ok && a.slope) ? a.slope[c] : 0.f;
    double s1 = 0, s2 = 0, s3 = 0;
    if (ok)
        for (int r = rg; r < a.R; r += 16) {
            const float xh = (a.X[(long)r * a.ldx + c] - mean) * rstd;
            const float u = xh * g + b;
            float dz = a.dY[(long)r * a.lddy + c];
            if (a.drop_p > 0.f) dz *= drop_scale(a.drop_p, a.seed, a.stream_id, (uint64_t)r * a.C + c);
            const float du = u > 0.f ? dz : sl * dz;
            s1 += du; s2 += (double)du * xh; s3 += u > 0.f ? 0.f : dz * u;
        }
    red[rg][cl][0] = s1; red[rg][cl][1] = s2; red[rg][cl][2] = s3;
    __syncthreads();
    if (rg == 0 && ok) {
        double x = 0, y = 0, z = 0;
        for (int q = 0; q < 16; ++q) { x += red[q][cl][0]; y += red[q][cl][1]; z += red[q][cl][2]; }
        s_db[cl] = (float)x; s_dg[cl] = (float)y;
        if (a.dbeta) a.dbeta[c] += (float)x;
        if (a.dgamma) a.dgamma[c] += (float)y;
        if (a.dslope) a.dslope[c] += (float)z;
    }
    __syncthreads();
    if (!ok || a.dX == nullptr) return;
    const float db = s_db[cl] / a.R, dg = s_dg[cl] / a.R;
    for (int r = rg; r < a.R; r += 16) {
        const float xh = (a.X[(long)r * a.ldx + c] - mean) * rstd;
        const float u = xh * g + b;
        float dz = a.dY[(long)r * a.lddy + c];
        if (a.drop_p > 0.f) dz *= drop_scale(a.drop_p, a.seed, a.stream_id, (uint64_t)r * a.C + c);
        const float du = u > 0.f ? dz : sl * dz;
        a.dX[(long)r * a.lddx + c] = g * rstd * (du - db - xh * dg);
    }
}

__global__ __launch_bounds__(256) void k_colsum_acc(const float* dY, long lddy, int R, int N, float* db) {
    __shared__ double sa[16][17];                            // 16 columns x 16 row lanes per workgroup
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, n = blockIdx.x * 16 + cl;
    double s = 0;
    if (n < N)
        for (int r = rl; r < R; r += 16) s += dY[(long)r * lddy + n];
    sa[rl][cl] = s;
    __syncthreads();
    if (threadIdx.x < 16 && n < N) {
        double x = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) x += sa[i][cl];
        db[n] += (float)x;
    }
}

}  // namespace

int linear_bwd_dw(const float* dY, long lddy, const float* X, long ldx, float* dW, float* db, int R, int N, int K, hipStream_t st) {
    if (R <= 0) return 0;
    SgemmArgs a{dY, 1, lddy, X, 1, ldx, dW, (long)K, N, K, R, nullptr, 1.f, 1};
    int rc = sgemm(a, st);
    if (rc) return rc;
    if (db) {
        hipLaunchKernelGGL(k_colsum_acc, dim3(cdiv(N, 16)), dim3(256), 0, st, dY, lddy, R, N, db);
        TCVN_LAUNCH_CHECK();
    }
    return 0;
}

int sgemm(const SgemmArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0) return 0;
    hipLaunchKernelGGL(k_sgemm, dim3(cdiv(a.N, 16), cdiv(a.M, 16)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int rows_bn_fwd(const RowsBnArgs& a, hipStream_t st) {
    if (a.R <= 0) return 0;
    hipLaunchKernelGGL(k_rows_bn_fwd, dim3(cdiv(a.C, 16)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int rows_bn_bwd(const RowsBnBwdArgs& a, hipStream_t st) {
    if (a.R <= 0) return 0;
    hipLaunchKernelGGL(k_rows_bn_bwd, dim3(cdiv(a.C, 16)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
