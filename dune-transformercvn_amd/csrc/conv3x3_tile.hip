// bf16 3x3 convolutions on padded LDS tiles (see tile3x3.h): forward, data gradient, weight gradient.
// Reference call site: Bottleneck.output_block (transformercvn/network/layers/dense_net.py:29-40) and its autograd.
//
// Forward: one workgroup = 128 padded output positions x 32 output channels; the BatchNorm+PReLU-transformed bf16 input
// image (128 + 2*(W+3) rows x 128 channels) is staged ONCE in LDS, then 9 taps x 8 k-steps of v_mfma_f32_32x32x16_bf16
// read it with row offsets (ds_read_b128, XOR-swizzled, conflict free); weights stream from L2 in fragment order.
// Algorithmic work per launch: 2 * pixels * 32 * 1152 FLOP; HBM: read 128 ch + write 32 ch per pixel.
#include <cstdlib>
#include "tile3x3.h"
#include "prof.h"

namespace tcvn {

using namespace t3;

namespace {

__device__ __forceinline__ int fdiv(int a, int d, float inv, int& rem) {     // a in [0, 2^24)
    int q = (int)((float)a * inv);
    rem = a - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
}
// pixel index of padded position g (or -1)
__device__ __forceinline__ int pix_of(const PadGeom& q, int g, float invWp, float invHp) {
    if (g < 0 || g >= (int)q.gtot) return -1;
    int wp, hp;
    const int row = fdiv(g, q.Wp, invWp, wp);
    const int img = fdiv(row, q.Hp, invHp, hp);
    if (hp < 1 || hp > q.H || wp < 1 || wp > q.W) return -1;
    return (img * q.H + (hp - 1)) * q.W + (wp - 1);
}

// Stage `nrows` padded positions starting at g_first of a [pixels,128] bf16 tensor into the swizzled 256-B-row LDS image,
// applying y = prelu(x*sc + sh, sl) per channel.  16 threads per row (one 16-B chunk each); rows advance by 16.
__device__ __forceinline__ void stage_act128(char* img, const bf16* __restrict__ X, const PadGeom& q, int g_first, int nrows,
                                             const float* __restrict__ sc, const float* __restrict__ sh,
                                             const float* __restrict__ sl, float invWp, float invHp, int tid) {
    const int chunk = tid & 15, r0 = tid >> 4;
    float csc[8], csh[8], csl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { csc[j] = sc[chunk * 8 + j]; csh[j] = sh[chunk * 8 + j]; csl[j] = sl[chunk * 8 + j]; }
#pragma unroll 4
    for (int row = r0; row < nrows; row += 16) {
        const int m = pix_of(q, g_first + row, invWp, invHp);
        u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m >= 0) {
            const u16x8 v = *reinterpret_cast<const u16x8*>(X + (long)m * 128 + chunk * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(prelu(fmaf(bf2f(v[j]), csc[j], csh[j]), csl[j]));
        }
        *reinterpret_cast<u16x8*>(img + off256(row, chunk)) = o;
    }
}

__global__ __launch_bounds__(256, 2) void k_conv3x3_fwd_bf16(const ConvFwdArgs g, int n_img, int ntiles, int swz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PadGeom q(n_img, g.H, g.W);
    const int nrows = q.rows();
    char* img = smem;
    int* pix = reinterpret_cast<int*>(smem + nrows * 256);
    double* red = reinterpret_cast<double*>(smem + nrows * 256 + TP * 4);      // [4][32][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const bf16* __restrict__ Y = reinterpret_cast<const bf16*>(g.A);
    const bf16* __restrict__ Wk = reinterpret_cast<const bf16*>(g.Wk);
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(g.Out);
    const int nb = gridDim.x;
    const int lb = swz ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;   // neighbours share an XCD's L2
    const bool nok = r < g.N;
    const bf16* wrow = Wk + (long)(nok ? r : 0) * g.Kp + 8 * h;
    const float bias = nok ? g.bias[r] : 0.f;
    const bool drop = g.drop_p > 0.f;

    double s1 = 0, s2 = 0;
    for (int t = lb; t < ntiles; t += nb) {
        const int g0 = t * TP;
        __syncthreads();
        stage_act128(img, Y, q, g0 - q.halo, nrows, g.sc, g.sh, g.sl, invWp, invHp, tid);
        if (tid < TP) pix[tid] = pix_of(q, g0 + tid, invWp, invHp);
        __syncthreads();

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int lrow0 = wave * 32 + r + q.halo;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int lr = lrow0 + (tap / 3 - 1) * q.Wp + (tap % 3 - 1);
            const char* arow = img + lr * 256;
            const int sw = lr & 15;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(arow + (((2 * ks + h) ^ sw) << 4));
                bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(wrow + tap * 128 + ks * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
        }
        // epilogue: bias, dropout (one Philox call per 4 consecutive pixels of a channel), store, statistics
        long cur_grp = -1;
        uint4 words = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lp = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int m = pix[lp];
            if (m >= 0 && nok) {
                float v = acc[e] + bias;
                if (drop) {
                    if ((m >> 2) != cur_grp) { cur_grp = m >> 2; words = drop_words(g.seed, g.stream_id, m, r, g.N); }
                    v *= drop_pick(words, m, g.drop_p);
                }
                const bf16 o = f2bf(v);
                Out[(long)m * g.ldo + g.n_off + r] = o;
                const double x = (double)bf2f(o);
                s1 += x; s2 += x * x;
            }
        }
    }
    if (g.part != nullptr) {
        double a = s1, b = s2;
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
        __syncthreads();
        if (lane < 32) { red[(wave * 32 + lane) * 2] = a; red[(wave * 32 + lane) * 2 + 1] = b; }
        __syncthreads();
        if (tid < g.N) {
            double x = 0, y = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { x += red[(w * 32 + tid) * 2]; y += red[(w * 32 + tid) * 2 + 1]; }
            g.part[((long)blockIdx.x * g.N + tid) * 2] = x;
            g.part[((long)blockIdx.x * g.N + tid) * 2 + 1] = y;
        }
    }
}

int tile_grid(long ntiles) {
    if (ntiles >= 512) return 512;
    if (ntiles >= 8) return (int)(ntiles / 8 * 8);
    return (int)ntiles;
}

}  // namespace

// n_img is recovered from M = n*H*W
static bool tile_disabled() {
    static const bool off = getenv("TCVN_DISABLE_TILE") != nullptr;      // validation switch: force the generic kernels
    return off;
}

bool conv3x3_tile_ok(const ConvFwdArgs& a) {
    if (tile_disabled()) return false;
    if (a.mode != MODE_BF16 || a.amode != A_3X3 || a.C != 128 || a.lda != 128 || a.N > 32 || a.Kp != 1152) return false;
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.Wk) & 15)) return false;
    if (a.M % (a.H * a.W) != 0) return false;
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return q.gtot < (1L << 24) && (long)q.rows() * 256 + TP * 4 + 4 * 32 * 16 <= 160 * 1024;
}
int conv3x3_tile_nblk(const ConvFwdArgs& a) {
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return tile_grid(q.tiles());
}
int conv3x3_fwd_tile(const ConvFwdArgs& a, hipStream_t st) {
    const int n_img = a.M / (a.H * a.W);
    const PadGeom q(n_img, a.H, a.W);
    const int ntiles = (int)q.tiles();
    const int nb = tile_grid(ntiles);
    const size_t smem = (size_t)q.rows() * 256 + TP * 4 + 4 * 32 * 16;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_fwd_bf16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024));
        attr = true;
    }
    ProfScope ps("k_conv3x3_fwd_bf16", 2.0 * a.M * (double)a.N * a.K, 0.0, st);
    hipLaunchKernelGGL(k_conv3x3_fwd_bf16, dim3(nb), dim3(256), smem, st, a, n_img, ntiles, (nb >= 8 && nb % 8 == 0) ? 1 : 0);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
