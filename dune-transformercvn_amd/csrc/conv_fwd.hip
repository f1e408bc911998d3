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
#include "conv_tile.h"
#include "prof.h"

namespace tcvn {

using namespace convk;

namespace {

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
                        if (g.drop_p > 0.f) v *= drop_scale_mn(g.drop_p, g.seed, g.stream_id, m, n, g.N);
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

int conv_fwd_nblk(const ConvFwdArgs& a) {
    if (stem_fwd_ok(a)) return stem_fwd_nblk(a);
    if (conv3x3_fwd_f32_ok(a)) return conv3x3_fwd_f32_nblk(a);
    if (conv1x1_fwd_f32_ok(a)) return conv1x1_fwd_f32_nblk(a);
    return conv3x3_tile_ok(a) ? conv3x3_tile_nblk(a) : conv_fwd_grid(a.M);
}

int conv_fwd(const ConvFwdArgs& a, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (a.Kp % BK != 0 || a.Kp < a.K) { fprintf(stderr, "tcvn: conv_fwd bad Kp=%d K=%d\n", a.Kp, a.K); return -2; }
    if (a.part != nullptr && a.nblk != conv_fwd_nblk(a)) { fprintf(stderr, "tcvn: conv_fwd nblk mismatch\n"); return -3; }
    if (stem_fwd_ok(a)) return stem_fwd_bf16(a, st);
    if (conv3x3_tile_ok(a)) return conv3x3_fwd_tile(a, st);
    if (conv3x3_fwd_f32_ok(a)) return conv3x3_fwd_f32(a, st);
    if (conv1x1_fwd_f32_ok(a)) return conv1x1_fwd_f32(a, st);
    char nm[96];
    snprintf(nm, sizeof(nm), "k_conv_fwd<%s,%d,%d>", a.mode == MODE_F32 ? "float" : "bf16", a.amode, a.N <= 32 ? 32 : a.N <= 64 ? 64 : 128);
    ProfScope ps(nm, 2.0 * a.M * (double)a.N * a.K, (double)a.M * (a.mode == MODE_F32 ? 4.0 : 2.0) * (a.N + (a.amode == A_STEM ? 4.0 * a.C : a.amode == A_3X3 ? (double)a.C : (double)a.K)), st);
    return a.mode == MODE_F32 ? launch_mode<float>(a, st) : launch_mode<bf16>(a, st);
}

}  // namespace tcvn
