// Fused optimizer step over the flat parameter / gradient arenas.
// Reference: NeutrinoBase.configure_optimizers (transformercvn/network/trainers/neutrino_base.py:88-152: torch.optim.AdamW with
// two parameter groups, weight decay l2_penalty on every parameter whose name lacks 'bias' / 'LayerNorm.weight') and Lightning's
// gradient_clip_val (train.py:140 -> torch.nn.utils.clip_grad_norm_, 2-norm over all gradients, coefficient
// min(1, clip / (norm + 1e-6))).  The reference runs ~782 small tensor updates per step; here every parameter is a view of one
// arena, so the step is two launches: a sum of squares and one element-wise AdamW pass that applies the clip coefficient on the fly.
// HBM bytes per element: read p, g, m, v, wd (20 B) + write p, m, v (12 B); 5.7 M elements -> 0.18 GB, ~45 us at 4 TB/s.
#include "../../include/tcvn_hip.h"
#include "tcvn_common.h"

namespace tcvn {
namespace {

__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ x, long n, double* __restrict__ partials) {
    __shared__ double red[4];
    double s = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const double v = x[i]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void k_sumsq_final(const double* __restrict__ partials, int nblk, float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.AdamW (decoupled weight decay, no amsgrad, no maximize):
//   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// wd[i] < 0 marks an element the reference's optimizer never touches (parameter without gradient).
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, const float* __restrict__ wd, long n, float lr, float b1,
                                               float b2, float eps, float bc1, float sqrt_bc2, const float* __restrict__ gss,
                                               float clip) {
    float coef = 1.f;
    if (gss != nullptr && clip > 0.f) coef = fminf(1.f, clip / (sqrtf(gss[0]) + 1e-6f));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float w = wd[i];
        if (w < 0.f) continue;
        const float gi = g[i] * coef;
        float pi = p[i] * (1.f - lr * w);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / sqrt_bc2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

}  // namespace
}  // namespace tcvn

using namespace tcvn;

extern "C" {

int tcvn_grad_sumsq(const float* x, int64_t n, double* partials, int n_partials, float* out, void* stream) {
    if (!x || !partials || !out || n < 0 || n_partials < 1) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nblk = n_partials < 1024 ? n_partials : 1024;
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nblk), dim3(256), 0, st, x, (long)n, partials);
    TCVN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, st, partials, nblk, out);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int tcvn_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* weight_decay, int64_t n,
                    float lr, float beta1, float beta2, float eps, int64_t step, const float* grad_sumsq, float clip, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !weight_decay || n < 0 || step < 1) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int nblk = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (nblk == 0) return 0;
    hipLaunchKernelGGL(k_adamw, dim3(nblk), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, weight_decay, (long)n, lr, beta1,
                       beta2, eps, (float)bc1, (float)sqrt(bc2), grad_sumsq, clip);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}
