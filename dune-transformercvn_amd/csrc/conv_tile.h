// Shared building blocks of the implicit-GEMM convolution kernels (forward, dgrad, wgrad):
// LDS tile layouts per operand type, MFMA tile products, and the fused BN+PReLU A-operand loaders.
#pragma once
#include "tcvn_ops.h"

namespace tcvn {
namespace convk {

constexpr int BM = 128, BK = 32, NT = 256;

template <typename T, int ROWS> struct Tile;
template <int ROWS> struct Tile<float, ROWS> {               // [k][row], +1 pad
    float d[BK][ROWS + 1];
    __device__ __forceinline__ void store8(int row, int k8, const float v[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[k8 + j][row] = v[j];
    }
};
template <int ROWS> struct Tile<bf16, ROWS> {                // [row][k], +8 pad (16 B)
    __attribute__((aligned(16))) bf16 d[ROWS][BK + 8];
    __device__ __forceinline__ void store8(int row, int k8, const float v[8]) {
        u16x8 p;
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = f2bf(v[j]);
        *reinterpret_cast<u16x8*>(&d[row][k8]) = p;
    }
};

template <typename T, int RA, int RB, int TM, int TN>
__device__ __forceinline__ void mma_tile(const Tile<T, RA>& As, const Tile<T, RB>& Bs, int arow0, int brow0, int lane,
                                         f32x16 (&acc)[TM][TN]);

template <int RA, int RB, int TM, int TN>
__device__ __forceinline__ void mma_tile_f32(const Tile<float, RA>& As, const Tile<float, RB>& Bs, int arow0, int brow0,
                                             int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As.d[ks * 2 + h][arow0 + i * 32 + r];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs.d[ks * 2 + h][brow0 + j * 32 + r];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}
template <int RA, int RB, int TM, int TN>
__device__ __forceinline__ void mma_tile_bf16(const Tile<bf16, RA>& As, const Tile<bf16, RB>& Bs, int arow0, int brow0,
                                              int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8_t a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(&As.d[arow0 + i * 32 + r][ks * 16 + 8 * h]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(&Bs.d[brow0 + j * 32 + r][ks * 16 + 8 * h]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}
template <int RA, int RB, int TM, int TN>
__device__ __forceinline__ void mma(const Tile<float, RA>& As, const Tile<float, RB>& Bs, int a0, int b0, int lane,
                                    f32x16 (&acc)[TM][TN]) { mma_tile_f32(As, Bs, a0, b0, lane, acc); }
template <int RA, int RB, int TM, int TN>
__device__ __forceinline__ void mma(const Tile<bf16, RA>& As, const Tile<bf16, RB>& Bs, int a0, int b0, int lane,
                                    f32x16 (&acc)[TM][TN]) { mma_tile_bf16(As, Bs, a0, b0, lane, acc); }

struct RowInfo {
    long base;       // element offset of the row's first channel
    int h, w;        // output pixel (3x3) or top-left input pixel (stem)
    int img;
    bool valid;
};

template <typename T, int AMODE>
__device__ __forceinline__ RowInfo row_info(const ConvFwdArgs& g, int m) {
    RowInfo r;
    r.valid = m < g.M;
    r.img = 0; r.h = 0; r.w = 0; r.base = 0;
    if (!r.valid) return r;
    if (AMODE == A_1X1) {
        r.base = (long)m * g.lda;
    } else if (AMODE == A_3X3) {
        r.w = m % g.W;
        r.h = (m / g.W) % g.H;
        r.base = (long)m * g.lda;
    } else {                                   // pooled 1x1 and stem: decode (img, ho, wo)
        const int hw = g.H * g.W;
        r.img = m / hw;
        const int rem = m - r.img * hw;
        const int ho = rem / g.W, wo = rem - ho * g.W;
        if (AMODE == A_1X1_POOL) {
            r.base = (((long)r.img * g.Hin + 2 * ho) * g.Win + 2 * wo) * g.lda;
        } else {
            r.h = 2 * ho - 3; r.w = 2 * wo - 3;
        }
    }
    return r;
}

template <typename T>
__device__ __forceinline__ void act8(const ConvFwdArgs& g, int c, int n, float v[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < n) v[j] = prelu(fmaf(v[j], g.sc[c + j], g.sh[c + j]), g.sl[c + j]);
}

// 8 consecutive k of row `ri` starting at k (k % 8 == 0), transformed, zero where k >= K / row invalid / tap outside
template <typename T, int AMODE>
__device__ __forceinline__ void load_a8(const ConvFwdArgs& g, const T* __restrict__ A, const RowInfo& ri, int k, bool vec,
                                        float v[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (!ri.valid || k >= g.K) return;
    const int n = min(8, g.K - k);
    if (AMODE == A_1X1) {
        if (vec) load8<T>(A + ri.base + k, v); else load8_guard<T>(A + ri.base + k, n, v);
        if (g.sc != nullptr) act8<T>(g, k, n, v);              // sc == nullptr: A is a materialised operand (fp32 transitions: pooled + activated)
    } else if (AMODE == A_1X1_POOL) {
        float t[8];
        const long offs[4] = {0, g.lda, (long)g.Win * g.lda, (long)(g.Win + 1) * g.lda};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (vec) load8<T>(A + ri.base + offs[q] + k, t); else load8_guard<T>(A + ri.base + offs[q] + k, n, t);
            act8<T>(g, k, n, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += t[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= 0.25f;
    } else if (AMODE == A_3X3) {
        const int tap = k / g.C, c = k - tap * g.C;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int ih = ri.h + ky - 1, iw = ri.w + kx - 1;
        if (ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) return;
        const T* p = A + ri.base + (long)((ky - 1) * g.W + (kx - 1)) * g.lda + c;
        if (vec) load8<T>(p, v); else load8_guard<T>(p, min(n, g.C - c), v);
        act8<T>(g, c, min(n, g.C - c), v);
    } else {                                   // stem: per-element tap decode (C = 3 channels per tap)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = k + j;
            if (kk < g.K) {
                const int tap = kk / g.C, c = kk - tap * g.C;
                const int ky = tap / 7, kx = tap - ky * 7;
                const int ih = ri.h + ky, iw = ri.w + kx;
                if (ih >= 0 && ih < g.Hin && iw >= 0 && iw < g.Win)
                    v[j] = to_f<T>(A[(((long)ri.img * g.Hin + ih) * g.Win + iw) * g.lda + c]);
            }
        }
    }
}

template <typename T> struct StatAcc { typedef double type; };   // fp64 partials: order-independent to ~1e-16
template <> struct StatAcc<float> { typedef double type; };


// transposed stores: 8 consecutive ROWS at one k (wgrad operands arrive row-major in the contraction index)
template <int ROWS> __device__ __forceinline__ void store8_t(Tile<float, ROWS>& t, int k, int row8, const float v[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) t.d[k][row8 + j] = v[j];
}
template <int ROWS> __device__ __forceinline__ void store8_t(Tile<bf16, ROWS>& t, int k, int row8, const float v[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) t.d[row8 + j][k] = f2bf(v[j]);
}

}  // namespace convk
}  // namespace tcvn
