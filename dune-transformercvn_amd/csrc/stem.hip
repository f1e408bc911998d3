// Stem-specific bf16 kernels (conv0 7x7/2 + BN0 + PReLU0 + AvgPool 3/2; reference layers/dense_net.py:112-121):
//   * k_pool0_bwd_vec   : pooling + PReLU + BatchNorm backward over the conv0 output, 16 B per thread
//   * k_stem_wgrad_sparse: conv0 weight gradient from the COO hit list -- the pixel maps are mostly empty, so
//         dW0[n][ky][kx][c] = sum_hits v[hit][c] * eff0[(y+3-ky)/2, (x+3-kx)/2][n]
//     touches ~12 output pixels per hit instead of contracting 147 taps for every one of the 28 000 output pixels
//     (~1.7 GFLOP instead of 135 GFLOP per 288 maps).
#include <cstdlib>
#include <type_traits>
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

// ---- activity bitmap of the conv0 output (see stem_mark in tcvn_ops.h): one thread per hit ------------------------------------------------
__global__ void k_stem_mark(const int* __restrict__ coords, long nnz, int n_img, int H, int W, int Hc, int Wc, int WW, uint32_t* __restrict__ act,
                            const float* __restrict__ bias, bf16* __restrict__ cline) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64) cline[threadIdx.x] = f2bf(bias[threadIdx.x]);     // the row of a position no hit reaches: bf16(0 + bias)
    if (i >= nnz) return;
    const int img = coords[i * 3], y = coords[i * 3 + 1], x = coords[i * 3 + 2];
    if (img < 0 || img >= n_img || y < 0 || y >= H || x < 0 || x >= W) return;                 // k_scatter drops the same hits
    // output (oy, ox) reads input rows 2*oy - 3 .. 2*oy + 3: ceil((y - 3) / 2) <= oy <= floor((y + 3) / 2)
    const int oy0 = max(0, (y - 2) >> 1), oy1 = min(Hc - 1, (y + 3) >> 1);
    const int ox0 = max(0, (x - 2) >> 1), ox1 = min(Wc - 1, (x + 3) >> 1);
    if (ox0 > ox1) return;
    const int w0 = ox0 >> 5, w1 = ox1 >> 5;
    const uint32_t m_lo = (w0 == w1 ? ((ox1 - ox0 + 1 >= 32) ? 0xffffffffu : ((1u << (ox1 - ox0 + 1)) - 1u)) : 0xffffffffu) << (ox0 & 31);
    const uint32_t m_hi = w0 == w1 ? 0u : (0xffffffffu >> (31 - (ox1 & 31)));
    for (int oy = oy0; oy <= oy1; ++oy) {
        uint32_t* row = act + ((long)img * Hc + oy) * WW;
        atomicOr(row + w0, m_lo);
        if (m_hi) atomicOr(row + w1, m_hi);
    }
}

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
template <typename T>
__global__ __launch_bounds__(256, 2) void k_stem_wgrad_sparse(const StemWgradArgs a) {
    __shared__ float wacc[49 * SW_MAXC][64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 49 * SW_MAXC * 64; i += 256) (&wacc[0][0])[i] = 0.f;
    __syncthreads();
    const T* img = reinterpret_cast<const T*>(a.img);
    const T* G = reinterpret_cast<const T*>(a.e.G);
    const T* X = reinterpret_cast<const T*>(a.e.X);
    const int n = lane;
    const bool nok = n < a.e.N;
    const float pn = nok ? a.e.P[n] : 0.f, qn = nok ? a.e.Q[n] : 0.f;
    float acc[49][3];
#pragma unroll
    for (int t = 0; t < 49; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; }
    const long gw = (long)blockIdx.x * 4 + (tid >> 6), nw = (long)gridDim.x * 4;
    // the coordinates of the NEXT hit are requested while this one is processed (a hit is a chain of dependent L2 round trips:
    // coordinates -> pixel value / gradient rows; with the first link prefetched a wave pays one per hit instead of two to three)
    int c_im = -1, c_y = 0, c_x = 0;
    if (gw < a.nnz) { c_im = a.coords[gw * 3]; c_y = a.coords[gw * 3 + 1]; c_x = a.coords[gw * 3 + 2]; }
    for (long hit = gw; hit < a.nnz; hit += nw) {
        const int im = __builtin_amdgcn_readfirstlane(c_im);
        const int y = __builtin_amdgcn_readfirstlane(c_y);
        const int x = __builtin_amdgcn_readfirstlane(c_x);
        if (hit + nw < a.nnz) { c_im = a.coords[(hit + nw) * 3]; c_y = a.coords[(hit + nw) * 3 + 1]; c_x = a.coords[(hit + nw) * 3 + 2]; }
        if (im < 0 || im >= a.n_img || y < 0 || y >= a.H || x < 0 || x >= a.W) continue;
        const T* px = img + (((long)im * a.H + y) * a.W + x) * a.Cpix;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = c < a.Cpix ? to_f<T>(px[c]) : 0.f;
        // The taps that see this hit have ky = (y+3) mod 2 (+2, +4, +6) and kx likewise: 9-16 of the 49.  One code copy per parity pair
        // keeps the accumulator indices static AND lets all <= 32 gradient-row loads of the hit be requested before the first use (the
        // tap-by-tap loop with its parity tests waited for every tap's two loads in turn: ~16 L2 round trips per hit).
        auto taps = [&](auto pyc, auto pxc) {
            constexpr int PY = decltype(pyc)::value, PX = decltype(pxc)::value;
            constexpr int NY = PY ? 3 : 4, NX = PX ? 3 : 4;
            float gq[NY][NX], xq[NY][NX];
            bool okq[NY][NX];
#pragma unroll
            for (int iy = 0; iy < NY; ++iy)
#pragma unroll
                for (int ix = 0; ix < NX; ++ix) {
                    const int oy = (y + 3 - (PY + 2 * iy)) >> 1, ox = (x + 3 - (PX + 2 * ix)) >> 1;       // (y+3-ky even by construction)
                    okq[iy][ix] = ((unsigned)oy < (unsigned)a.Hc) & ((unsigned)ox < (unsigned)a.Wc) & nok;
                    const long p = okq[iy][ix] ? ((long)im * a.Hc + oy) * a.Wc + ox : 0;                  // clamped: unconditional loads
                    gq[iy][ix] = to_f<T>(G[p * a.e.ldg + (nok ? n : 0)]);
                    xq[iy][ix] = to_f<T>(X[p * a.e.ldx + (nok ? n : 0)]);
                }
#pragma unroll
            for (int iy = 0; iy < NY; ++iy)
#pragma unroll
                for (int ix = 0; ix < NX; ++ix) {
                    const float eff = okq[iy][ix] ? gq[iy][ix] + pn * xq[iy][ix] + qn : 0.f;
                    constexpr int dummy = 0; (void)dummy;
                    const int t = (PY + 2 * iy) * 7 + (PX + 2 * ix);
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[t][c] = fmaf(v[c], eff, acc[t][c]);
                }
        };
        const int pyr = (y + 3) & 1, pxr = (x + 3) & 1;                       // wave-uniform
        if (pyr == 0) { if (pxr == 0) taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); else taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); }
        else { if (pxr == 0) taps(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); else taps(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); }
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



// Tiled variant (64 channels): a workgroup owns 8 x 32 conv0-output pixels of one map.  The effective gradient of the 5 x 17
// pooled pixels whose 3x3/2 windows touch the tile is computed ONCE into LDS (fp32); the per-pixel pass then sums its <= 4 windows
// from LDS instead of re-reading (G, x) of every window from L2 (4 x 256 B per pixel and chunk in k_pool0_bwd_vec).
constexpr int PB_TH = 8, PB_TW = 32, PB_PH = PB_TH / 2 + 1, PB_PW = PB_TW / 2 + 1;
template <typename T> __device__ __forceinline__ void store8_g(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8_g<float>(float* p, const float v[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8_g<bf16>(bf16* p, const float v[8]) {
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(v[j]);
    *reinterpret_cast<u16x8*>(p) = o;
}
template <typename T>
__global__ __launch_bounds__(256, 2) void k_pool0_bwd_tile(const Pool0BwdArgs a, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float effs[PB_PH * PB_PW * 64];
    __shared__ double red[4][8][8][3];
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* G = reinterpret_cast<const T*>(a.e.G);
    const T* D = reinterpret_cast<const T*>(a.e.X);
    T* DU = reinterpret_cast<T*>(a.DU);
    const int tid = threadIdx.x, c8 = tid & 7;
    // per-channel tables live in LDS and are re-read where they are used (a phase needs 16-24 of the 40 values: keeping all of them
    // and fp64 running sums in registers next to a 2x2 block's 64 operand registers would halve the occupancy)
    __shared__ __attribute__((aligned(16))) float tabs[5][64];
    __shared__ uint32_t actw[PB_TH];                                       // activity word of each tile row (PB_TW == 32: one word per row and tile)
    const int WW = (a.Win + 31) >> 5;
    if (tid < 64) { tabs[0][tid] = a.sc[tid]; tabs[1][tid] = a.sh[tid]; tabs[2][tid] = a.sl[tid]; tabs[3][tid] = a.e.P[tid]; tabs[4][tid] = a.e.Q[tid]; }
    auto tab8 = [&](int which, float (&v)[8]) {
        const float4 x0 = *reinterpret_cast<const float4*>(&tabs[which][c8 * 8]), x1 = *reinterpret_cast<const float4*>(&tabs[which][c8 * 8 + 4]);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    };
    // a thread's running sums cover a few hundred values: fp32 here, fp64 across threads and workgroups
    float s1[8], s2[8], s3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; s3[j] = 0; }
    const long ntiles = (long)a.n_img * tiles_x * tiles_y;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const long img = tile / ((long)tiles_x * tiles_y);
        const int h0 = ty * PB_TH, w0 = tx * PB_TW, ho0 = h0 / 2 - 1, wo0 = w0 / 2 - 1;
        __syncthreads();                                                   // previous tile's readers are done with effs
        if (tid < PB_TH) actw[tid] = (a.act != nullptr && h0 + tid < a.Hin) ? a.act[(img * a.Hin + h0 + tid) * WW + tx] : 0xffffffffu;
        {                                                                   // eff of the pooled pixels (zero outside the map)
            constexpr int NP1 = (PB_PH * PB_PW * 8 + 255) / 256;           // 3 trips: all their loads are requested before the first use
            float gv[NP1][8], dv[NP1][8];
            bool in[NP1];
#pragma unroll
            for (int k = 0; k < NP1; ++k) {
                const int i = tid + 256 * k, pp = i >> 3, py = pp / PB_PW, px = pp - py * PB_PW, ho = ho0 + py, wo = wo0 + px;
                in[k] = (i < PB_PH * PB_PW * 8) & (ho >= 0) & (ho < a.Ho) & (wo >= 0) & (wo < a.Wo);
                const long mo = in[k] ? (img * a.Ho + ho) * a.Wo + wo : img * a.Ho * a.Wo;      // clamped: unconditional loads
                load8<T>(G + mo * a.e.ldg + c8 * 8, gv[k]);
                load8<T>(D + mo * a.e.ldx + c8 * 8, dv[k]);
            }
            float cP[8], cQ[8];
            tab8(3, cP); tab8(4, cQ);
#pragma unroll
            for (int k = 0; k < NP1; ++k) {
                const int i = tid + 256 * k, pp = i >> 3;
                if (i < PB_PH * PB_PW * 8) {
                    float e8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e8[j] = in[k] ? gv[k][j] + cP[j] * dv[k][j] + cQ[j] : 0.f;
                    float4* o = reinterpret_cast<float4*>(effs + pp * 64 + c8 * 8);
                    o[0] = make_float4(e8[0], e8[1], e8[2], e8[3]); o[1] = make_float4(e8[4], e8[5], e8[6], e8[7]);
                }
            }
        }
        __syncthreads();
        // a thread takes 2x2 blocks of pixels (tile origins are even): the block's four pixels see the pooled windows
        //   (even,even) all four, (even,odd) the right two, (odd,even) the lower two, (odd,odd) the lower right one
        // so the four window sums are read once per block (4 LDS reads for 4 pixels instead of 9, no parity branches), and the
        // block's four x rows are requested together.  Out-of-map windows hold zeros in effs.
#pragma unroll 1
        for (int k = 0; k < PB_TH * PB_TW / 128; ++k) {
            const int bi = (tid >> 3) + 32 * k, by = bi / (PB_TW / 2), bx = bi - by * (PB_TW / 2);
            const int h = h0 + 2 * by, w = w0 + 2 * bx;
            float xv[4][8];
            bool ok[4], live[4];
            const uint32_t aw0 = actw[2 * by] >> (2 * bx), aw1 = actw[2 * by + 1] >> (2 * bx);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int hh = h + (t >> 1), ww = w + (t & 1);
                ok[t] = (hh < a.Hin) & (ww < a.Win);
                live[t] = ok[t] & ((((t >> 1) ? aw1 : aw0) >> (t & 1)) & 1u);                          // a hit reaches this position: its rows exist
                const long p = ok[t] ? (img * a.Hin + hh) * a.Win + ww : img * a.Hin * a.Win;       // clamped: the load is unconditional
                const T* src = (a.act == nullptr || live[t]) ? X + p * a.C + c8 * 8 : reinterpret_cast<const T*>(a.cline) + c8 * 8;
                load8<T>(src, xv[t]);
            }
            float e[4][8], sc[8], sh[8], sl[8];
            tab8(0, sc); tab8(1, sh); tab8(2, sl);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4* e4 = reinterpret_cast<const float4*>(effs + ((by + (t >> 1)) * PB_PW + bx + (t & 1)) * 64 + c8 * 8);
                const float4 x0 = e4[0], x1 = e4[1];
                e[t][0] = x0.x; e[t][1] = x0.y; e[t][2] = x0.z; e[t][3] = x0.w; e[t][4] = x1.x; e[t][5] = x1.y; e[t][6] = x1.z; e[t][7] = x1.w;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float lower = e[2][j] + e[3][j], right = e[1][j] + e[3][j];
                    const float dzs = t == 0 ? lower + e[0][j] + e[1][j] : t == 1 ? right : t == 2 ? lower : e[3][j];
                    const float z = ok[t] ? dzs * (1.0f / 9.0f) : 0.f;
                    const float x = xv[t][j];
                    const float u = fmaf(x, sc[j], sh[j]);
                    const float du = u > 0.f ? z : sl[j] * z;
                    s1[j] += du; s2[j] = fmaf(du, x, s2[j]); s3[j] += u > 0.f ? 0.f : z * u;
                    o[j] = sc[j] * du;
                }
                if (live[t]) store8_g<T>(DU + ((img * a.Hin + h + (t >> 1)) * a.Win + w + (t & 1)) * a.C + c8 * 8, o);
            }
        }
    }
    // threads with equal (tid & 7) hold the same channels: fold lanes 8, 16, 32 apart, then the four waves
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d1 = (double)s1[j], d2 = (double)s2[j], d3 = (double)s3[j];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o); }
        if (lane < 8) { red[wave][lane][j][0] = d1; red[wave][lane][j][1] = d2; red[wave][lane][j][2] = d3; }
    }
    __syncthreads();
    if (tid < 64) {
        const int ch = tid >> 3, j = tid & 7;
        double x = 0, y = 0, z = 0;
        for (int w = 0; w < 4; ++w) { x += red[w][ch][j][0]; y += red[w][ch][j][1]; z += red[w][ch][j][2]; }
        double* o = a.part + ((long)blockIdx.x * a.C + tid) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv0 forward (7x7, stride 2, pad 3, 3 -> 64 channels; reference layers/dense_net.py:112-116) as an implicit GEMM:
// one workgroup = 8 x 16 output pixels of one map; its 21 x 37 input patch is staged in LDS as NHWC4 (8 B per pixel, the
// 4th channel zero) so that the 8 contraction elements of an MFMA lane -- 2 neighbouring pixels of one kernel row -- are
// one aligned ds_read_b128.  Contraction order k' = ky*32 + kx*4 + c (kx = 7 and c = 3 carry zero weights): 14 k-steps
// of 16.  Each wave owns 2 output rows x 16 columns x 64 channels; the weights (28 fragments) stay in registers.
// The tile leaves through a bf16 C tile in LDS so that every lane stores 16 B of the NHWC output row.
// Algorithmic HBM bytes per map: 400*280*3*2 read + 200*140*64*2 written (the write dominates: 1.03 GB for 288 maps).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ST_TH = 8, ST_TW = 16;                        // output tile
constexpr int ST_PR = 2 * ST_TH + 5, ST_PW = 40;            // patch rows, patch pitch in pixels (37 used)
constexpr int ST_CP = 72;                                   // C tile pitch in bf16 (64 + 8: lane halves on different banks)

__global__ __launch_bounds__(256, 2) void k_stem_fwd_bf16(const ConvFwdArgs g, int n_img, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) unsigned short patch[2][ST_PR * ST_PW * 4];
    __shared__ __attribute__((aligned(16))) unsigned short ctile[ST_TH * ST_TW * ST_CP];
    __shared__ double red[4][64][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bf16* __restrict__ img = reinterpret_cast<const bf16*>(g.A);
    const bf16* __restrict__ Wk = reinterpret_cast<const bf16*>(g.Wk);          // [64][Kp], k = (ky*7 + kx)*3 + c
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(g.Out);
    const int Hin = g.Hin, Win = g.Win, Ho = g.H, Wo = g.W;
    const long ntiles = (long)n_img * tiles_x * tiles_y;

    // weight fragments: lane (r, h) of n-tile nt holds k' = 16 s + 8 h + i for output channel nt*32 + r
    bf16x8_t bw[2][14];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            u16x8 w;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kp = 16 * s + 8 * h + i, ky = kp >> 5, kx = (kp & 31) >> 2, c = kp & 3;
                w[i] = (kx < 7 && c < 3) ? Wk[(long)(nt * 32 + r) * g.Kp + (ky * 7 + kx) * 3 + c] : (bf16)0;
            }
            bw[nt][s] = __builtin_bit_cast(bf16x8_t, w);
        }
    const float bias0 = g.bias[r], bias1 = g.bias[32 + r];
    for (int i = tid; i < 2 * ST_PR * ST_PW * 4; i += 256) (&patch[0][0])[i] = 0;      // 4th channel + pitch padding stay zero
    double s1[2] = {0, 0}, s2[2] = {0, 0};

    // patch staging: element e of patch row pr is input (iy0 + pr, ix0 + e/3, channel e%3); 21 x 111 elements, <= 10 per thread.
    // The (row, pixel, channel) of a thread's k-th element does not depend on the tile: decode once, outside the tile loop.
    constexpr int NE = ST_PR * 37 * 3, PER = (NE + 255) / 256;
    int e_pr[PER], e_px[PER], e_goff[PER], e_loff[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + 256 * k;
        const int pr = i / 111, e = i - pr * 111, px = e / 3, c = e - px * 3;
        e_pr[k] = i < NE ? pr : -100000;                        // out-of-range elements fail the row test below
        e_px[k] = px;
        e_goff[k] = (pr * Win + px) * 3 + c;                     // element offset relative to the patch origin
        e_loff[k] = (pr * ST_PW + px) * 4 + c;
    }
    auto load_patch = [&](long tile, unsigned short (&v)[PER]) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const long n = tile / ((long)tiles_x * tiles_y);
        const int iy0 = 2 * ty * ST_TH - 3, ix0 = 2 * tx * ST_TW - 3;
        const bf16* base = img + ((n * Hin + iy0) * (long)Win + ix0) * 3;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int iy = iy0 + e_pr[k], ix = ix0 + e_px[k];
            v[k] = 0;
            if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) v[k] = base[e_goff[k]];
        }
    };
    auto store_patch = [&](int buf, const unsigned short (&v)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (e_pr[k] >= 0) patch[buf][e_loff[k]] = v[k];
    };

    unsigned short pv[PER];
    long tile = blockIdx.x;
    __syncthreads();
    if (tile < ntiles) { load_patch(tile, pv); store_patch(0, pv); }
    __syncthreads();
    int cur = 0;
    const int oy_l = 2 * wave + (r >> 4), ox_l = r & 15;          // this lane's pixel of the wave's 32-pixel M tile
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const bool more = tile + gridDim.x < ntiles;
        if (more) load_patch(tile + gridDim.x, pv);                // next patch travels under the MFMAs
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[nt][k] = 0.f;
        const unsigned short* pb = &patch[cur][0];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int gq = 2 * s + h, ky = gq >> 2, q4 = gq & 3;
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(pb + ((2 * oy_l + ky) * ST_PW + 2 * ox_l + 2 * q4) * 4);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[0][s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[1][s], acc[1], 0, 0, 0);
        }
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const long n = tile / ((long)tiles_x * tiles_y);
        const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
        // C layout: pixel row of the M tile = (k&3) + 8*(k>>2) + 4*h, channel = nt*32 + r
        float f1[2] = {0, 0}, f2[2] = {0, 0};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int pr = (k & 3) + 8 * (k >> 2) + 4 * h;         // 0..31 inside the wave's two rows
                const int py = 2 * wave + (pr >> 4), px = pr & 15;
                const bf16 o = f2bf(acc[nt][k] + (nt ? bias1 : bias0));
                ctile[(py * ST_TW + px) * ST_CP + nt * 32 + r] = o;
                if (oy0 + py < Ho && ox0 + px < Wo) { const float x = bf2f(o); f1[nt] += x; f2[nt] += x * x; }
            }
        s1[0] += (double)f1[0]; s2[0] += (double)f2[0]; s1[1] += (double)f1[1]; s2[1] += (double)f2[1];
        if (more) store_patch(cur ^ 1, pv);
        __syncthreads();
        // 128 pixels x 128 B: 4 chunks of 16 B per thread
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k, p = idx >> 3, c8 = idx & 7;
            const int py = p >> 4, px = p & 15;
            if (oy0 + py < Ho && ox0 + px < Wo)
                *reinterpret_cast<u16x8*>(Out + ((n * Ho + oy0 + py) * (long)Wo + ox0 + px) * g.ldo + c8 * 8) =
                    *reinterpret_cast<const u16x8*>(&ctile[p * ST_CP + c8 * 8]);
        }
        __syncthreads();
    }
    if (g.part != nullptr) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            double a = s1[nt], b = s2[nt];
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (lane < 32) { red[wave][nt * 32 + lane][0] = a; red[wave][nt * 32 + lane][1] = b; }
        }
        __syncthreads();
        if (tid < 64) {
            double a = 0, b = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[w][tid][0]; b += red[w][tid][1]; }
            g.part[((long)blockIdx.x * 64 + tid) * 2] = a;
            g.part[((long)blockIdx.x * 64 + tid) * 2 + 1] = b;
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// conv0 forward, second version.  Same tile, patch and contraction order as k_stem_fwd_bf16; what changed is everything around the
// 28 MFMAs, which was ~1 100 instructions per wave and tile (the kernel wrote its 1 GB at 1.36 TB/s, issue-bound):
//   * the MFMA operands are swapped -- weights are the A operand, the patch fragment the B operand -- so a lane holds, for ITS pixel,
//     4 consecutive channels per accumulator group: one v_cvt_pk pair and one 8-B LDS write per group (8 writes per lane and tile
//     instead of 32 two-byte writes), no per-element bounds branch (one validity flag per lane and tile);
//   * the 28 weight fragments live in LDS (28 KB), which frees 112 registers for the per-lane statistics (a lane accumulates the 64
//     sums of its 32 channels over all its tiles; the 32-lane fold happens once, at the end) and the batched patch loads;
//   * patch loads are unconditional from clamped addresses + select, 32-bit tile arithmetic.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_stem_fwd2_bf16(const ConvFwdArgs g, int n_img, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) unsigned short patch[2][ST_PR * ST_PW * 4];
    __shared__ __attribute__((aligned(16))) unsigned short ctile[ST_TH * ST_TW * ST_CP];
    __shared__ __attribute__((aligned(16))) unsigned short wl[2 * 14 * 64 * 8];          // [ct][s][lane] weight fragments
    __shared__ double red[4][64][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bf16* __restrict__ img = reinterpret_cast<const bf16*>(g.A);
    const bf16* __restrict__ Wk = reinterpret_cast<const bf16*>(g.Wk);          // [64][Kp], k = (ky*7 + kx)*3 + c
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(g.Out);
    const int Hin = g.Hin, Win = g.Win, Ho = g.H, Wo = g.W;
    const unsigned ntiles = (unsigned)n_img * tiles_x * tiles_y;

    // weight fragments: lane (r, h) of channel tile ct holds k' = 16 s + 8 h + i for output channel ct*32 + r
    for (int f = wave; f < 28; f += 4) {
        const int ct = f / 14, sstep = f - ct * 14;
        u16x8 w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kp = 16 * sstep + 8 * h + i, ky = kp >> 5, kx = (kp & 31) >> 2, c = kp & 3;
            w[i] = (kx < 7 && c < 3) ? Wk[(long)(ct * 32 + r) * g.Kp + (ky * 7 + kx) * 3 + c] : (bf16)0;
        }
        *reinterpret_cast<u16x8*>(&wl[(f * 64 + lane) * 8]) = w;
    }
    // this lane's 32 channels: ct*32 + (k&3) + 8*(k>>2) + 4*h
    float bias[2][16], f1[2][16], f2[2][16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int k = 0; k < 16; ++k) { bias[ct][k] = g.bias[ct * 32 + (k & 3) + 8 * (k >> 2) + 4 * h]; f1[ct][k] = 0.f; f2[ct][k] = 0.f; }
    for (int i = tid; i < 2 * ST_PR * ST_PW * 4; i += 256) (&patch[0][0])[i] = 0;      // 4th channel + pitch padding stay zero

    // patch staging: element e of patch row pr is input (iy0 + pr, ix0 + e/3, channel e%3); 21 x 111 elements, <= 10 per thread
    constexpr int NE = ST_PR * 37 * 3, PER = (NE + 255) / 256;
    int e_pr[PER], e_pc[PER];                                    // patch row; pixel * 4 + channel
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + 256 * k;
        const int pr = i / 111, e = i - pr * 111, px = e / 3, c = e - px * 3;
        e_pr[k] = i < NE ? pr : -100000;                        // out-of-range elements fail the row test below
        e_pc[k] = px * 4 + c;
    }
    auto load_patch = [&](unsigned tile, unsigned short (&v)[PER]) {
        const unsigned trow = tile / tiles_x;
        const int tx = (int)(tile - trow * tiles_x), ty = (int)(trow % tiles_y);
        const long n = trow / tiles_y;
        const int iy0 = 2 * ty * ST_TH - 3, ix0 = 2 * tx * ST_TW - 3;
        const bf16* base = img + ((n * Hin + iy0) * (long)Win + ix0) * 3;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int px = e_pc[k] >> 2, c = e_pc[k] & 3;
            const int iy = iy0 + e_pr[k], ix = ix0 + px;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const unsigned short x = *(ok ? base + (e_pr[k] * Win + px) * 3 + c : img);      // unconditional load, clamped address
            v[k] = ok ? x : (unsigned short)0;
        }
    };
    auto store_patch = [&](int buf, const unsigned short (&v)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (e_pr[k] >= 0) patch[buf][e_pr[k] * (ST_PW * 4) + e_pc[k]] = v[k];
    };

    unsigned short pv[PER];
    unsigned tile = blockIdx.x;
    __syncthreads();
    if (tile < ntiles) { load_patch(tile, pv); store_patch(0, pv); }
    __syncthreads();
    int cur = 0;
    const int oy_l = 2 * wave + (r >> 4), ox_l = r & 15;          // this lane's pixel of the tile
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const bool more = tile + gridDim.x < ntiles;
        if (more) load_patch(tile + gridDim.x, pv);                // next patch travels under the MFMAs
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[ct][k] = 0.f;
        const unsigned short* pb = &patch[cur][0];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int gq = 2 * s + h, ky = gq >> 2, q4 = gq & 3;
            const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(pb + ((2 * oy_l + ky) * ST_PW + 2 * ox_l + 2 * q4) * 4);
            const bf16x8_t w0 = *reinterpret_cast<const bf16x8_t*>(&wl[(s * 64 + lane) * 8]);
            const bf16x8_t w1 = *reinterpret_cast<const bf16x8_t*>(&wl[((14 + s) * 64 + lane) * 8]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, b, acc[1], 0, 0, 0);
        }
        const unsigned trow = tile / tiles_x;
        const int tx = (int)(tile - trow * tiles_x), ty = (int)(trow % tiles_y);
        const long n = trow / tiles_y;
        const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
        const float vm = ((oy0 + oy_l < Ho) & (ox0 + ox_l < Wo)) ? 1.f : 0.f;             // this lane's pixel is inside the map
        unsigned short* cp = &ctile[(oy_l * ST_TW + ox_l) * ST_CP + 4 * h];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                u16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = gq * 4 + j;
                    o[j] = f2bf(acc[ct][k] + bias[ct][k]);
                    const float x = bf2f(o[j]) * vm;
                    f1[ct][k] += x; f2[ct][k] = fmaf(x, x, f2[ct][k]);
                }
                *reinterpret_cast<u16x4*>(cp + ct * 32 + gq * 8) = o;
            }
        if (more) store_patch(cur ^ 1, pv);
        __syncthreads();
        // 128 pixels x 128 B: 4 chunks of 16 B per thread
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k, p = idx >> 3, c8 = idx & 7;
            const int py = p >> 4, px = p & 15;
            if ((oy0 + py < Ho) & (ox0 + px < Wo))
                *reinterpret_cast<u16x8*>(Out + ((n * Ho + oy0 + py) * (long)Wo + ox0 + px) * g.ldo + c8 * 8) =
                    *reinterpret_cast<const u16x8*>(&ctile[p * ST_CP + c8 * 8]);
        }
        __syncthreads();
    }
    if (g.part != nullptr) {
        // fold the 32 pixel lanes of each half wave; lanes 0 and 32 then hold the wave's sums of their 32 channels
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                double a = (double)f1[ct][k], b = (double)f2[ct][k];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (r == 0) { const int c = ct * 32 + (k & 3) + 8 * (k >> 2) + 4 * h; red[wave][c][0] = a; red[wave][c][1] = b; }
            }
        __syncthreads();
        if (tid < 64) {
            double a = 0, b = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[w][tid][0]; b += red[w][tid][1]; }
            g.part[((long)blockIdx.x * 64 + tid) * 2] = a;
            g.part[((long)blockIdx.x * 64 + tid) * 2 + 1] = b;
        }
    }
}

}  // namespace

int pool0_bwd_vec_grid(int n_img, int Hin, int Win) {
    const long g = ((long)n_img * Hin * Win + 31) / 32;
    return (int)(g < 2048 ? g : 2048);
}
bool pool0_bwd_vec_ok(const Pool0BwdArgs& a) {
    if (a.mode == MODE_F32) return conv3x3_tile_enabled() && a.C == 64 && (a.e.ldg & 7) == 0 && (a.e.ldx & 7) == 0;      // tile kernel only
    return a.mode == MODE_BF16 && (a.C == 8 || a.C == 16 || a.C == 32 || a.C == 64) && (a.e.ldg & 7) == 0 && (a.e.ldx & 7) == 0;
}
int pool0_bwd_vec(const Pool0BwdArgs& a, hipStream_t st) {
    if (!pool0_bwd_vec_ok(a)) return -2;
    if (a.nblk != pool0_bwd_vec_grid(a.n_img, a.Hin, a.Win)) return -3;
    static const bool old = TCVN_KNOB_SET("TCVN_POOL0_BWD_FLAT");        // A/B switch
    if (a.mode == MODE_F32) {
        hipLaunchKernelGGL(k_pool0_bwd_tile<float>, dim3(a.nblk), dim3(256), 0, st, a, cdiv(a.Win, PB_TW), cdiv(a.Hin, PB_TH));
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.C == 64 && !old) {
        hipLaunchKernelGGL(k_pool0_bwd_tile<bf16>, dim3(a.nblk), dim3(256), 0, st, a, cdiv(a.Win, PB_TW), cdiv(a.Hin, PB_TH));
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_pool0_bwd_vec, dim3(a.nblk), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int stem_mark(const int* coords, long nnz, int n_img, int H, int W, int Hc, int Wc, uint32_t* act, const float* bias, void* cline, hipStream_t st) {
    const int WW = stem_act_words(Wc);
    TCVN_CHECK(hipMemsetAsync(act, 0, (size_t)n_img * Hc * WW * 4, st));
    hipLaunchKernelGGL(k_stem_mark, dim3(cdiv(nnz > 0 ? nnz : 1, 256)), dim3(256), 0, st, coords, nnz, n_img, H, W, Hc, Wc, WW, act, bias, reinterpret_cast<bf16*>(cline));
    TCVN_LAUNCH_CHECK();
    return 0;
}

int stem_wgrad_sparse(const StemWgradArgs& a, float* dWk, hipStream_t st) {
    if (a.Cpix > 3 || a.e.N > 64 || a.Kp < 49 * a.Cpix) return -2;
    const int nb = 512;                    // latency-bound walk over the hit list: two workgroups per CU
    if ((long)nb * a.e.N * a.Kp * 4 > a.slab_bytes) return -3;
    {
        ProfScope ps("k_stem_wgrad_sparse", 2.0 * a.nnz * 12.25 * a.Cpix * a.e.N, 0.0, st);
        if (a.mode == MODE_F32) hipLaunchKernelGGL(k_stem_wgrad_sparse<float>, dim3(nb), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_stem_wgrad_sparse<bf16>, dim3(nb), dim3(256), 0, st, a);
        TCVN_LAUNCH_CHECK();
    }
    return slab_reduce(a.slab, nb, (long)a.e.N * a.Kp, dWk, st);
}


bool stem_fwd_ok(const ConvFwdArgs& a) {
    return conv3x3_tile_enabled() && a.mode == MODE_BF16 && a.amode == A_STEM && a.C == 3 && a.N == 64 && a.K == 147 && a.Kp >= 147 &&
           a.ldo == 64 && a.n_off == 0 && a.Hin == 2 * a.H && a.Win == 2 * a.W && a.M % (a.H * a.W) == 0 &&
           (reinterpret_cast<uintptr_t>(a.Out) & 15) == 0;
}
int stem_fwd_nblk(const ConvFwdArgs& a) {
    const long nt = (long)(a.M / (a.H * a.W)) * cdiv(a.W, ST_TW) * cdiv(a.H, ST_TH);
    return (int)(nt < 512 ? nt : 512);
}
int stem_fwd_bf16(const ConvFwdArgs& a, hipStream_t st) {
    const int n_img = a.M / (a.H * a.W);
    ProfScope ps("k_stem_fwd_bf16", 2.0 * a.M * (double)a.N * a.K, (double)a.M * 2.0 * (a.N + 4.0 * a.C), st);
    [[maybe_unused]] static const bool old_kernel = TCVN_KNOB_SET("TCVN_STEM_FWD_V1");         // validation build: the first version
    if (old_kernel) hipLaunchKernelGGL(k_stem_fwd_bf16, dim3(stem_fwd_nblk(a)), dim3(256), 0, st, a, n_img, cdiv(a.W, ST_TW), cdiv(a.H, ST_TH));
    else hipLaunchKernelGGL(k_stem_fwd2_bf16, dim3(stem_fwd_nblk(a)), dim3(256), 0, st, a, n_img, cdiv(a.W, ST_TW), cdiv(a.H, ST_TH));
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
