// Stem-specific bf16 kernels (conv0 7x7/2 + BN0 + PReLU0 + AvgPool 3/2; reference layers/dense_net.py:112-121):
//   * k_pool0_bwd_vec   : pooling + PReLU + BatchNorm backward over the conv0 output, 16 B per thread
//   * k_stem_wgrad_sparse: conv0 weight gradient from the COO hit list -- the pixel maps are mostly empty, so
//         dW0[n][ky][kx][c] = sum_hits v[hit][c] * eff0[(y+3-ky)/2, (x+3-kx)/2][n]
//     touches ~12 output pixels per hit instead of contracting 147 taps for every one of the 28 000 output pixels
//     (~1.7 GFLOP instead of 135 GFLOP per 288 maps).
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

__global__ __launch_bounds__(256) void k_pool0_bwd_vec(const Pool0BwdArgs a) {
    __shared__ double red[4][8][8][3];
    const bf16* X = reinterpret_cast<const bf16*>(a.X);
    const bf16* G = reinterpret_cast<const bf16*>(a.e.G);
    const bf16* D = reinterpret_cast<const bf16*>(a.e.X);
    bf16* DU = reinterpret_cast<bf16*>(a.DU);
    const int cpr = a.C >> 3;                              // chunks per pixel (<= 8)
    const int tid = threadIdx.x;
    const int c8 = tid % cpr;
    float sc[8], sh[8], sl[8], cP[8], cQ[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = a.sc[c8 * 8 + j]; sh[j] = a.sh[c8 * 8 + j]; sl[j] = a.sl[c8 * 8 + j];
        cP[j] = a.e.P[c8 * 8 + j]; cQ[j] = a.e.Q[c8 * 8 + j];
    }
    double s1[8], s2[8], s3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; s3[j] = 0; }
    const int ppb = 256 / cpr;                             // pixels per block iteration
    const long npix = (long)a.n_img * a.Hin * a.Win;
    for (long p = (long)blockIdx.x * ppb + tid / cpr; p < npix; p += (long)gridDim.x * ppb) {
        const int w = (int)(p % a.Win);
        const int h = (int)((p / a.Win) % a.Hin);
        const long img = p / ((long)a.Win * a.Hin);
        float dz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int ho_lo = max(0, (h - 1) / 2), ho_hi = min(a.Ho - 1, h / 2);
        const int wo_lo = max(0, (w - 1) / 2), wo_hi = min(a.Wo - 1, w / 2);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (2 * ho > h || 2 * ho + 2 < h || 2 * wo > w || 2 * wo + 2 < w) continue;
                const long mo = (img * a.Ho + ho) * a.Wo + wo;
                const u16x8 gv = *reinterpret_cast<const u16x8*>(G + mo * a.e.ldg + c8 * 8);
                const u16x8 dv = *reinterpret_cast<const u16x8*>(D + mo * a.e.ldx + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) dz[j] += bf2f(gv[j]) + cP[j] * bf2f(dv[j]) + cQ[j];
            }
        const u16x8 xv = *reinterpret_cast<const u16x8*>(X + p * a.C + c8 * 8);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = dz[j] * (1.0f / 9.0f);
            const float x = bf2f(xv[j]);
            const float u = fmaf(x, sc[j], sh[j]);
            const float du = u > 0.f ? z : sl[j] * z;
            s1[j] += du; s2[j] += (double)du * x; s3[j] += u > 0.f ? 0.f : z * u;
            o[j] = f2bf(sc[j] * du);
        }
        *reinterpret_cast<u16x8*>(DU + p * a.C + c8 * 8) = o;
    }
    // lanes with equal (tid % cpr): cpr divides 8 => lanes differing by multiples of 8 share the chunk
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); s3[j] += __shfl_xor(s3[j], o); }
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wave][lane][j][0] = s1[j]; red[wave][lane][j][1] = s2[j]; red[wave][lane][j][2] = s3[j]; }
    }
    __syncthreads();
    if (tid < a.C) {
        // channel c lives in chunk c/8; lanes 0..7 of a wave hold chunks (lane % cpr): sum the lanes of that chunk over the waves
        const int ch = tid >> 3, j = tid & 7;
        double x = 0, y = 0, z = 0;
        for (int w = 0; w < 4; ++w)
            for (int l = ch; l < 8; l += cpr) { x += red[w][l][j][0]; y += red[w][l][j][1]; z += red[w][l][j][2]; }
        // every lane l with l % cpr == ch carries the SAME fully reduced value after the xor-shuffles over 8,16,32 only if
        // cpr == 8; for cpr < 8 the lanes l, l+cpr, ... hold distinct partial sums -> they are all added above
        double* o = a.part + ((long)blockIdx.x * a.C + tid) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

constexpr int SW_MAXC = 4;
__global__ __launch_bounds__(256) void k_stem_wgrad_sparse(const StemWgradArgs a) {
    __shared__ float wacc[49 * SW_MAXC][64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 49 * SW_MAXC * 64; i += 256) (&wacc[0][0])[i] = 0.f;
    __syncthreads();
    const bf16* img = reinterpret_cast<const bf16*>(a.img);
    const bf16* G = reinterpret_cast<const bf16*>(a.e.G);
    const bf16* X = reinterpret_cast<const bf16*>(a.e.X);
    const int n = lane;
    const bool nok = n < a.e.N;
    const float pn = nok ? a.e.P[n] : 0.f, qn = nok ? a.e.Q[n] : 0.f;
    float acc[49][3];
#pragma unroll
    for (int t = 0; t < 49; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; }
    const long gw = (long)blockIdx.x * 4 + (tid >> 6), nw = (long)gridDim.x * 4;
    for (long hit = gw; hit < a.nnz; hit += nw) {
        const int im = __builtin_amdgcn_readfirstlane(a.coords[hit * 3]);
        const int y = __builtin_amdgcn_readfirstlane(a.coords[hit * 3 + 1]);
        const int x = __builtin_amdgcn_readfirstlane(a.coords[hit * 3 + 2]);
        if (im < 0 || im >= a.n_img || y < 0 || y >= a.H || x < 0 || x >= a.W) continue;
        const bf16* px = img + (((long)im * a.H + y) * a.W + x) * a.Cpix;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = c < a.Cpix ? bf2f(px[c]) : 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int ty = y + 3 - ky;
            if ((ty & 1) || ty < 0 || (ty >> 1) >= a.Hc) continue;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int tx = x + 3 - kx;
                if ((tx & 1) || tx < 0 || (tx >> 1) >= a.Wc) continue;
                const long p = ((long)im * a.Hc + (ty >> 1)) * a.Wc + (tx >> 1);
                const float eff = nok ? bf2f(G[p * a.e.ldg + n]) + pn * bf2f(X[p * a.e.ldx + n]) + qn : 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[ky * 7 + kx][c] = fmaf(v[c], eff, acc[ky * 7 + kx][c]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 49; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < a.Cpix) atomicAdd(&wacc[t * a.Cpix + c][n], acc[t][c]);      // LDS atomics: 4 waves per block
    __syncthreads();
    // slab[block][n][Kp] in the standard kernel layout (k = tap*Cpix + c)
    float* out = a.slab + (long)blockIdx.x * a.e.N * a.Kp;
    for (int i = tid; i < a.e.N * a.Kp; i += 256) {
        const int nn = i / a.Kp, k = i - nn * a.Kp;
        out[i] = k < 49 * a.Cpix ? wacc[k][nn] : 0.f;
    }
}

}  // namespace

int pool0_bwd_vec_grid(int n_img, int Hin, int Win) {
    const long g = ((long)n_img * Hin * Win + 31) / 32;
    return (int)(g < 2048 ? g : 2048);
}
bool pool0_bwd_vec_ok(const Pool0BwdArgs& a) {
    return a.mode == MODE_BF16 && (a.C == 8 || a.C == 16 || a.C == 32 || a.C == 64) && (a.e.ldg & 7) == 0 && (a.e.ldx & 7) == 0;
}
int pool0_bwd_vec(const Pool0BwdArgs& a, hipStream_t st) {
    if (!pool0_bwd_vec_ok(a)) return -2;
    if (a.nblk != pool0_bwd_vec_grid(a.n_img, a.Hin, a.Win)) return -3;
    hipLaunchKernelGGL(k_pool0_bwd_vec, dim3(a.nblk), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int stem_wgrad_sparse(const StemWgradArgs& a, float* dWk, hipStream_t st) {
    if (a.Cpix > 3 || a.e.N > 64 || a.Kp < 49 * a.Cpix) return -2;
    const int nb = 256;
    if ((long)nb * a.e.N * a.Kp * 4 > a.slab_bytes) return -3;
    {
        ProfScope ps("k_stem_wgrad_sparse", 2.0 * a.nnz * 12.25 * a.Cpix * a.e.N, 0.0, st);
        hipLaunchKernelGGL(k_stem_wgrad_sparse, dim3(nb), dim3(256), 0, st, a);
        TCVN_LAUNCH_CHECK();
    }
    return slab_reduce(a.slab, nb, (long)a.e.N * a.Kp, dWk, st);
}

}  // namespace tcvn
