// conv_in of the SDXL-style embedder (diffusers Encoder.conv_in: Conv2d(3, 64, 3, padding=1) on the 400 x 280 pixel map;
// reference call site transformercvn/network/layers/sdxl_net.py:27-34), bf16:
//   forward          write-bound (128 B out per pixel, 6 B in): an 8 x 32 pixel tile, its 10 x 34 x 3 halo patch in LDS, K = 27
//                    padded to 32 as two bf16 MFMA k-steps whose A fragments are gathered from the patch (8 ds_read_u16 each),
//                    fp32 C tile in LDS, 16-B NHWC stores;
//   weight gradient  from the hit list: the pixel maps are > 99 % zeros, and a zero pixel contributes nothing, so one wave per
//                    non-zero pixel accumulates its 9 x 3 products with the 64-channel output-gradient rows it touches
//                    (lane = output channel) -- 60 k hits instead of 16 M positions;
//   bias gradient    column sums of the output gradient (one pass, 16 B per lane).
// The hit list is the COO list of the forward; like the DenseNet stem (stem.hip) this assumes coalesced coordinates (one entry
// per pixel), which is what the dataset produces.
#include "prof.h"
#include "sdxl_ops.h"
#include "tcvn_ops.h"

namespace tcvn {

namespace {

constexpr int TH = 8, TW = 32, PW = TW + 2, PH = TH + 2;
constexpr int CP = 68;

struct InFwdArgs {
    const bf16* img; const bf16* W; const float* bias; bf16* Out; double* stats;
    int n, H, W_, tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(256, 2) void k_sconv_in_fwd(const InFwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_in[];
    float* Cs = reinterpret_cast<float*>(smem_in);                             // [256][CP] fp32, 69 632 B
    bf16* patch = reinterpret_cast<bf16*>(smem_in + TH * TW * CP * 4);         // [PH*PW][3]
    float* lacc = reinterpret_cast<float*>(smem_in + TH * TW * CP * 4 + (PH * PW * 3 + 8) * 2);   // [2] tile statistics (GroupNorm of the consumer)
    if (threadIdx.x == 0) { lacc[0] = 0.f; lacc[1] = 0.f; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // B fragments: W[n][k], k = tap*3 + c (Kp = 32, zero padded): lane (n = l31 of n tile, lh) holds k = ks*16 + lh*8 .. +8
    bf16x8_t bw[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            bw[nt][ks] = *reinterpret_cast<const bf16x8_t*>(g.W + (long)(nt * 32 + l31) * 32 + ks * 16 + lh * 8);
    const float b0 = g.bias ? g.bias[l31] : 0.f, b1 = g.bias ? g.bias[32 + l31] : 0.f;
    // gather offsets of this lane's 16 contraction elements inside the patch (elements), relative to its pixel
    int off[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + lh * 8 + j, tap = (k * 11) >> 5, c = k - tap * 3;
            off[ks][j] = k < 27 ? ((tap / 3) * PW + tap % 3) * 3 + c : 0;
        }
    for (int t = blockIdx.x; t < g.ntiles; t += gridDim.x) {
        const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        __syncthreads();
        for (int i = tid; i < PH * PW; i += 256) {
            const int py = i / PW, px = i - py * PW, y = y0 + py, x = x0 + px;
            bf16 v0 = 0, v1 = 0, v2 = 0;
            if (y >= 0 && y < g.H && x >= 0 && x < g.W_) {
                const bf16* p = g.img + (((long)img * g.H + y) * g.W_ + x) * 3;
                v0 = p[0]; v1 = p[1]; v2 = p[2];
            }
            patch[i * 3] = v0; patch[i * 3 + 1] = v1; patch[i * 3 + 2] = v2;
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = wave * 2 + rt;
            const bf16* pp = patch + (row * PW + l31) * 3;
            f32x16 acc0, acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u16x8 av;
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = pp[off[ks][j]];
                const bf16x8_t a = __builtin_bit_cast(bf16x8_t, av);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[0][ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[1][ks], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pos = row * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                Cs[pos * CP + l31] = acc0[e] + b0;
                Cs[pos * CP + 32 + l31] = acc1[e] + b1;
            }
        }
        __syncthreads();
        float ts = 0.f, tss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * 256, pos = idx >> 3, ch = idx & 7;
            const int y = ty * TH + (pos >> 5), x = tx * TW + (pos & 31);
            if (y < g.H && x < g.W_) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8 + 4);
                u16x8 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ov[j] = f2bf(c0[j]); ov[4 + j] = f2bf(c1[j]); }
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = bf2f(ov[j]); ts += f; tss += f * f; }
                *reinterpret_cast<u16x8*>(g.Out + (((long)img * g.H + y) * g.W_ + x) * 64 + ch * 8) = ov;
            }
        }
        if (g.stats) {                                             // uniform
            ts = wave_sum(ts); tss = wave_sum(tss);
            if (lane == 0) { atomicAdd(&lacc[0], ts); atomicAdd(&lacc[1], tss); }
            __syncthreads();
            if (tid == 0) {
                atomicAdd(g.stats + 2 * img, (double)lacc[0]); atomicAdd(g.stats + 2 * img + 1, (double)lacc[1]);
                lacc[0] = 0.f; lacc[1] = 0.f;
            }
        }
    }
}

struct InWgradArgs {
    const int* coords; long nnz; const bf16* img; const bf16* dOut; float* dWk; int n, H, W_, Kp;
};

__global__ __launch_bounds__(256) void k_sconv_in_wgrad_sparse(const InWgradArgs a) {
    __shared__ float wacc[27][64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 27 * 64; i += 256) (&wacc[0][0])[i] = 0.f;
    __syncthreads();
    float acc[9][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; }
    const long gw = (long)blockIdx.x * 4 + (tid >> 6), nw = (long)gridDim.x * 4;
    for (long hit = gw; hit < a.nnz; hit += nw) {
        const int im = __builtin_amdgcn_readfirstlane(a.coords[hit * 3]);
        const int y = __builtin_amdgcn_readfirstlane(a.coords[hit * 3 + 1]);
        const int x = __builtin_amdgcn_readfirstlane(a.coords[hit * 3 + 2]);
        if (im < 0 || im >= a.n || y < 0 || y >= a.H || x < 0 || x >= a.W_) continue;
        const bf16* px = a.img + (((long)im * a.H + y) * a.W_ + x) * 3;
        const float v0 = bf2f(px[0]), v1 = bf2f(px[1]), v2 = bf2f(px[2]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int oy = y + 1 - ky;                             // out[oy][ox] reads in[oy + ky - 1][ox + kx - 1]
            if (oy < 0 || oy >= a.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ox = x + 1 - kx;
                if (ox < 0 || ox >= a.W_) continue;
                const float d = bf2f(a.dOut[(((long)im * a.H + oy) * a.W_ + ox) * 64 + lane]);
                acc[ky * 3 + kx][0] = fmaf(v0, d, acc[ky * 3 + kx][0]);
                acc[ky * 3 + kx][1] = fmaf(v1, d, acc[ky * 3 + kx][1]);
                acc[ky * 3 + kx][2] = fmaf(v2, d, acc[ky * 3 + kx][2]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&wacc[t * 3 + c][lane], acc[t][c]);
    __syncthreads();
    for (int i = tid; i < 27 * 64; i += 256) {
        const int k = i >> 6, n = i & 63;
        atomicAdd(a.dWk + (long)n * a.Kp + k, wacc[k][n]);
    }
}

// dbias[c] += sum over rows of X[row][c], C = 64 bf16, dense rows
__global__ __launch_bounds__(256) void k_colsum64_bf16(const bf16* __restrict__ X, long rows, float* __restrict__ dst) {
    __shared__ float red[32][64];
    const int tid = threadIdx.x, cg = tid & 7, rl = tid >> 3;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long r = (long)blockIdx.x * 32 + rl; r < rows; r += (long)gridDim.x * 32) {
        const u16x8 v = *reinterpret_cast<const u16x8*>(X + r * 64 + cg * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += bf2f(v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = s[j];
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += red[q][tid];
        atomicAdd(dst + tid, t);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool in_geom_ok(const SConv& g) {
    return conv3x3_tile_enabled() && g.mode == MODE_BF16 && g.ks == 3 && g.stride == 1 && g.pad == 1 && g.Cin == 3 && g.Cout == 64 && g.lda == 3 &&
           g.Ho == g.Hin && g.Wo == g.Win && g.Kp == 32;
}

}  // namespace

bool sconv_in_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, const void* Out, long ldo, int out_f32) {
    return in_geom_ok(g) && Res == nullptr && !out_f32 && ldo == 64 && al16(Wk) && al16(Out) && In != nullptr;
}
int sconv_in_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, void* Out, double* stats, hipStream_t st) {
    InFwdArgs a{};
    a.stats = stats;
    a.img = reinterpret_cast<const bf16*>(In); a.W = reinterpret_cast<const bf16*>(Wk); a.bias = bias; a.Out = reinterpret_cast<bf16*>(Out);
    a.n = g.n; a.H = g.Hin; a.W_ = g.Win;
    a.tiles_x = (g.Win + TW - 1) / TW; a.tiles_y = (g.Hin + TH - 1) / TH; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    constexpr size_t smem = TH * TW * CP * 4 + (PH * PW * 3 + 8) * 2 + 16;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv_in_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = true;
    }
    const int grid = a.ntiles < 512 ? a.ntiles : 512;
    hipLaunchKernelGGL(k_sconv_in_fwd, dim3(grid), dim3(256), smem, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

bool sconv_in_wgrad_ok(const SConv& g, const void* dOut, long lddo) {
    return in_geom_ok(g) && g.hits != nullptr && lddo == 64 && al16(dOut);
}
int sconv_in_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st) {
    InWgradArgs a{g.hits, g.nnz, reinterpret_cast<const bf16*>(In), reinterpret_cast<const bf16*>(dOut), dWk, g.n, g.Hin, g.Win, g.Kp};
    if (g.nnz > 0) {
        long nb = (g.nnz + 63) / 64;                              // about 16 hits per wave
        if (nb > 256) nb = 256;
        hipLaunchKernelGGL(k_sconv_in_wgrad_sparse, dim3((int)nb), dim3(256), 0, st, a);
        TCVN_LAUNCH_CHECK();
    }
    if (dbias != nullptr) {
        const long rows = (long)g.n * g.Hin * g.Win;
        long nb = (rows + 32 * 64 - 1) / (32 * 64);
        if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(k_colsum64_bf16, dim3((int)nb), dim3(256), 0, st, reinterpret_cast<const bf16*>(dOut), rows, dbias);
        TCVN_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace tcvn
