// Launch interface of the small fp32 row kernels (rows.hip, encoder.hip).
#pragma once
#include "tcvn_common.h"

namespace tcvn {

struct SgemmArgs {
    const float* A; long sai, sak;
    const float* B; long sbj, sbk;
    float* C; long ldc;
    int M, N, K;
    const float* bias; float alpha; int accumulate;
};
int sgemm(const SgemmArgs& a, hipStream_t st);

// Y = X W^T + b      (W row-major [N][K] like torch.nn.Linear)
static inline int linear_fwd(const float* X, long ldx, const float* W, const float* b, float* Y, long ldy, int R, int N, int K,
                             hipStream_t st) {
    SgemmArgs a{X, ldx, 1, W, (long)K, 1, Y, ldy, R, N, K, b, 1.f, 0};
    return sgemm(a, st);
}
// dX (+)= dY W
static inline int linear_bwd_dx(const float* dY, long lddy, const float* W, float* dX, long lddx, int R, int N, int K,
                                int accumulate, hipStream_t st) {
    SgemmArgs a{dY, lddy, 1, W, 1, (long)K, dX, lddx, R, K, N, nullptr, 1.f, accumulate};
    return sgemm(a, st);
}
// dW += dY^T X ; db += colsum(dY)
int linear_bwd_dw(const float* dY, long lddy, const float* X, long ldx, float* dW, float* db, int R, int N, int K, hipStream_t st);

struct RowsBnArgs {
    const float* X; long ldx; int R, C;
    const float *gamma, *beta, *slope;
    float *running_mean, *running_var;
    float* Y; long ldy;
    float *save_mean, *save_rstd;
    int train; float eps, momentum;
    float drop_p; uint64_t seed; uint32_t stream_id;
    int no_norm;                 // options.linear_batch_norm == False: no BatchNorm1d in the block (norm = Identity): Y = drop(act(X));
                                 // slope == nullptr is ReLU (options.linear_prelu_activation == False)
};
int rows_bn_fwd(const RowsBnArgs& a, hipStream_t st);

struct RowsBnBwdArgs {
    const float* X; long ldx; const float* dY; long lddy; int R, C;
    const float *gamma, *beta, *slope;
    const float *save_mean, *save_rstd;
    float* dX; long lddx;
    float *dgamma, *dbeta, *dslope;
    float drop_p; uint64_t seed; uint32_t stream_id;
    int no_norm;                 // as in RowsBnArgs: dX = act'(X) * drop * dY, no batch terms (dslope may be null: ReLU)
};
int rows_bn_bwd(const RowsBnBwdArgs& a, hipStream_t st);

}  // namespace tcvn
