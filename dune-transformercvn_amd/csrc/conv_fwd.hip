// Forward convolutions of the DenseNet embedder as implicit GEMMs on the gfx950 matrix cores.
//
// Replaces the ATen call sites K2, K4, K5, K7 of SURVEY.md 2.3 (reference: transformercvn/network/layers/
// dense_net.py:18-27 bottleneck_block, :29-40 output_block, :84-94 Transition, :112-118 conv0): the producer-side
// BatchNorm + PReLU is applied while the A operand is staged into LDS, the bias / dropout / BatchNorm statistics of
// the *next* layer run in the epilogue, and the output goes straight into its channel slice of the block's
// concat buffer (no torch.cat).
//
// Tile: 128 rows (pixels) x {32,64,128} output channels per 256-thread workgroup, K staged 32 deep through LDS.
//   fp32 mode : v_mfma_f32_32x32x2_f32 (exact fp32, parity path), LDS tiles k-major so fragment reads are conflict free
//   bf16 mode : v_mfma_f32_32x32x16_bf16, LDS tiles row-major with 16 B row padding (ds_read_b128 conflict free)
// Workgroups walk M tiles with a grid stride and keep the per-channel statistics in registers across tiles, so one
// launch writes one partial row per workgroup (deterministic, no atomics).
#include "tcvn_ops.h"

namespace tcvn {

namespace {

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
        act8<T>(g, k, n, v);
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

template <typename T> struct StatAcc { typedef float type; };
template <> struct StatAcc<float> { typedef double type; };

template <typename T, int AMODE, int BN_>
__global__ __launch_bounds__(NT) void k_conv_fwd(const ConvFwdArgs g) {
    constexpr int WN = BN_ >= 64 ? 2 : 1, WM = 4 / WN, TM = BM / WM / 32, TN = BN_ / WN / 32;
    constexpr int A_OCT = BM * BK / 8 / NT;                       // octets of A per thread and k-tile (2)
    constexpr int B_OCT = (BN_ * BK / 8 + NT - 1) / NT;
    typedef typename StatAcc<T>::type stat_t;

    __shared__ Tile<T, BM> As;
    __shared__ Tile<T, BN_> Bs;
    __shared__ double red[WM][BN_][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN_;
    const T* __restrict__ A = reinterpret_cast<const T*>(g.A);
    const T* __restrict__ Wk = reinterpret_cast<const T*>(g.Wk);
    T* __restrict__ Out = reinterpret_cast<T*>(g.Out);
    const int mtiles = (g.M + BM - 1) / BM, ktiles = g.Kp / BK;
    constexpr unsigned ALIGN = sizeof(T) * 8 - 1;
    const bool vec = ((g.lda & 7) == 0) && ((reinterpret_cast<uintptr_t>(A) & ALIGN) == 0) &&
                     (AMODE == A_3X3 ? (g.C & 7) == 0 : (g.K & 7) == 0) && AMODE != A_STEM;
    const int oct = tid & 3, r0 = tid >> 2;

    stat_t s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0; s2[j] = 0; }

    for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
        const int m0 = mt * BM;
        RowInfo ri[A_OCT];
#pragma unroll
        for (int i = 0; i < A_OCT; ++i) ri[i] = row_info<T, AMODE>(g, m0 + r0 + i * 64);

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        float ra[A_OCT][8], rb[B_OCT][8];
        auto fetch = [&](int kt) {
            const int k = kt * BK + oct * 8;
#pragma unroll
            for (int i = 0; i < A_OCT; ++i) load_a8<T, AMODE>(g, A, ri[i], k, vec, ra[i]);
#pragma unroll
            for (int i = 0; i < B_OCT; ++i) {
                const int r = r0 + i * 64, n = n0 + r;
                if (r < BN_ && n < g.N) load8<T>(Wk + (long)n * g.Kp + k, rb[i]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) rb[i][j] = 0.f;
                }
            }
        };
        fetch(0);
        for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
            for (int i = 0; i < A_OCT; ++i) As.store8(r0 + i * 64, oct * 8, ra[i]);
#pragma unroll
            for (int i = 0; i < B_OCT; ++i)
                if (r0 + i * 64 < BN_) Bs.store8(r0 + i * 64, oct * 8, rb[i]);
            __syncthreads();
            if (kt + 1 < ktiles) fetch(kt + 1);
            mma(As, Bs, wm * (BM / WM), wn * (BN_ / WN), lane, acc);
            __syncthreads();
        }

        // epilogue: bias, dropout, store, statistics of the stored values
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN_ / WN) + j * 32 + (lane & 31);
            const bool nok = n < g.N;
            const float b = nok ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (nok && m < g.M) {
                        float v = acc[i][j][e] + b;
                        if (g.drop_p > 0.f) v *= drop_scale(g.drop_p, g.seed, g.stream_id, (uint64_t)m * g.N + n);
                        const T o = from_f<T>(v);
                        Out[(long)m * g.ldo + g.n_off + n] = o;
                        const stat_t x = (stat_t)to_f<T>(o);
                        s1[j] += x; s2[j] += x * x;
                    }
                }
            }
        }
    }

    if (g.part != nullptr) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            double a = (double)s1[j], b = (double)s2[j];
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (lane < 32) {
                red[wm][wn * (BN_ / WN) + j * 32 + lane][0] = a;
                red[wm][wn * (BN_ / WN) + j * 32 + lane][1] = b;
            }
        }
        __syncthreads();
        if (tid < BN_ && n0 + tid < g.N) {
            double a = 0, b = 0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += red[w][tid][0]; b += red[w][tid][1]; }
            double* p = g.part + ((long)blockIdx.x * g.N + n0 + tid) * 2;
            p[0] = a; p[1] = b;
        }
    }
}

template <typename T, int AMODE>
int launch_bn(const ConvFwdArgs& a, hipStream_t st) {
    const int gx = a.nblk > 0 ? a.nblk : conv_fwd_grid(a.M);
    if (a.N <= 32) {
        hipLaunchKernelGGL((k_conv_fwd<T, AMODE, 32>), dim3(gx, 1), dim3(NT), 0, st, a);
    } else if (a.N <= 64) {
        hipLaunchKernelGGL((k_conv_fwd<T, AMODE, 64>), dim3(gx, 1), dim3(NT), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_conv_fwd<T, AMODE, 128>), dim3(gx, cdiv(a.N, 128)), dim3(NT), 0, st, a);
    }
    TCVN_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int launch_mode(const ConvFwdArgs& a, hipStream_t st) {
    switch (a.amode) {
        case A_1X1: return launch_bn<T, A_1X1>(a, st);
        case A_1X1_POOL: return launch_bn<T, A_1X1_POOL>(a, st);
        case A_3X3: return launch_bn<T, A_3X3>(a, st);
        case A_STEM: return launch_bn<T, A_STEM>(a, st);
    }
    return -1;
}

}  // namespace

int conv_fwd_grid(int M) {
    const int mtiles = cdiv(M, BM);
    return mtiles < 512 ? mtiles : 512;
}

int conv_fwd(const ConvFwdArgs& a, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (a.Kp % BK != 0 || a.Kp < a.K) { fprintf(stderr, "tcvn: conv_fwd bad Kp=%d K=%d\n", a.Kp, a.K); return -2; }
    if (a.part != nullptr && a.nblk != conv_fwd_grid(a.M)) { fprintf(stderr, "tcvn: conv_fwd nblk mismatch\n"); return -3; }
    return a.mode == MODE_F32 ? launch_mode<float>(a, st) : launch_mode<bf16>(a, st);
}

}  // namespace tcvn
