// Kernels of the SDXL-style embedder (reference: transformercvn/network/layers/sdxl_net.py:27-34 -> diffusers Encoder; block
// definitions restated in oracle/sdxl_oracle.py): NHWC convolutions of any kernel size / stride / top-left padding as implicit
// GEMMs on the matrix cores (fp32: v_mfma_f32_32x32x2_f32, bf16: v_mfma_f32_32x32x16_bf16; 128 pixels x {32,64,128} channels
// per 256-thread workgroup, K staged 32 deep through LDS with register prefetch), and GroupNorm(1 group) + SiLU as a
// two-phase reduction (per-image fp64 sums, then the normalising pass).  The convolution operands are plain tensors: the
// normalised + activated input of every convolution is materialised once by gn_act (it is read by the forward AND the
// weight-gradient GEMM), residual adds and biases are epilogues.
#include "conv_tile.h"
#include "prof.h"
#include "sdxl_ops.h"

namespace tcvn {

using namespace convk;

namespace {

struct RowPix { int img, h0, w0; bool valid; };

__device__ __forceinline__ RowPix out_pixel(const SConv& g, int m, int M) {          // forward / wgrad rows: output pixels
    RowPix r;
    r.valid = m < M;
    const int hw = g.Ho * g.Wo;
    r.img = m / hw;
    const int rem = m - r.img * hw;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    r.h0 = ho * g.stride - g.pad; r.w0 = wo * g.stride - g.pad;
    return r;
}

// 8 consecutive k (k = tap*Cin + c) of the im2col row of one output pixel; zero outside the map / beyond K
template <typename T>
__device__ __forceinline__ void load_im2col8(const SConv& g, const T* __restrict__ In, const RowPix& r, int k, int K, bool vec,
                                             float v[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (!r.valid || k >= K) return;
    if (vec) {
        const int tap = k / g.Cin, c = k - tap * g.Cin;
        const int ky = tap / g.ks, kx = tap - ky * g.ks;
        const int ih = r.h0 + ky, iw = r.w0 + kx;
        if (ih < 0 || ih >= g.Hin || iw < 0 || iw >= g.Win) return;
        load8<T>(In + (((long)r.img * g.Hin + ih) * g.Win + iw) * g.lda + c, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = k + j;
            if (kk < K) {
                const int tap = kk / g.Cin, c = kk - tap * g.Cin;
                const int ky = tap / g.ks, kx = tap - ky * g.ks;
                const int ih = r.h0 + ky, iw = r.w0 + kx;
                if (ih >= 0 && ih < g.Hin && iw >= 0 && iw < g.Win)
                    v[j] = to_f<T>(In[(((long)r.img * g.Hin + ih) * g.Win + iw) * g.lda + c]);
            }
        }
    }
}

template <typename T> __device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (sizeof(T) * 8 - 1)) == 0; }

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int BN_>
__global__ __launch_bounds__(NT) void k_sconv_fwd(const SConv g, const T* __restrict__ In, const T* __restrict__ Wk,
                                                  const float* __restrict__ bias, const T* __restrict__ Res, long ldres,
                                                  void* OutV, long ldo, int out_f32, int M, int K) {
    constexpr int WN = BN_ >= 64 ? 2 : 1, WM = 4 / WN, TM = BM / WM / 32, TN = BN_ / WN / 32;
    constexpr int A_OCT = BM * BK / 8 / NT;
    constexpr int B_OCT = (BN_ * BK / 8 + NT - 1) / NT;
    __shared__ Tile<T, BM> As;
    __shared__ Tile<T, BN_> Bs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN_;
    const int mtiles = (M + BM - 1) / BM, ktiles = g.Kp / BK;
    const bool vec = (g.Cin & 7) == 0 && (g.lda & 7) == 0 && aligned16<T>(In);
    const int oct = tid & 3, r0 = tid >> 2;
    for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
        const int m0 = mt * BM;
        RowPix rp[A_OCT];
#pragma unroll
        for (int i = 0; i < A_OCT; ++i) rp[i] = out_pixel(g, m0 + r0 + i * 64, M);
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
            for (int i = 0; i < A_OCT; ++i) load_im2col8<T>(g, In, rp[i], k, K, vec, ra[i]);
#pragma unroll
            for (int i = 0; i < B_OCT; ++i) {
                const int r = r0 + i * 64, n = n0 + r;
                if (r < BN_ && n < g.Cout) load8<T>(Wk + (long)n * g.Kp + k, rb[i]);
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
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN_ / WN) + j * 32 + (lane & 31);
            if (n >= g.Cout) continue;
            const float b = bias ? bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (m >= M) continue;
                    float v = acc[i][j][e] + b;
                    if (Res) v += to_f<T>(Res[(long)m * ldres + n]);
                    if (out_f32) reinterpret_cast<float*>(OutV)[(long)m * ldo + n] = v;
                    else reinterpret_cast<T*>(OutV)[(long)m * ldo + n] = from_f<T>(v);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// data gradient: rows = INPUT pixels, k = tap*Cout + n gathers the output gradient
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int BN_>
__global__ __launch_bounds__(NT) void k_sconv_dgrad(const SConv g, const T* __restrict__ dOut, long lddo, const T* __restrict__ Wt,
                                                    T* __restrict__ dIn, long lddi, int accumulate, int M, int K) {
    constexpr int WN = BN_ >= 64 ? 2 : 1, WM = 4 / WN, TM = BM / WM / 32, TN = BN_ / WN / 32;
    constexpr int A_OCT = BM * BK / 8 / NT;
    constexpr int B_OCT = (BN_ * BK / 8 + NT - 1) / NT;
    __shared__ Tile<T, BM> As;
    __shared__ Tile<T, BN_> Bs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN_;
    const int mtiles = (M + BM - 1) / BM, ktiles = g.Kpt / BK;
    const bool vec = (g.Cout & 7) == 0 && (lddo & 7) == 0 && aligned16<T>(dOut);
    const int oct = tid & 3, r0 = tid >> 2;
    for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
        const int m0 = mt * BM;
        int ri[A_OCT], ry[A_OCT], rx[A_OCT];
#pragma unroll
        for (int i = 0; i < A_OCT; ++i) {
            const int m = m0 + r0 + i * 64;
            const int hw = g.Hin * g.Win;
            ri[i] = m < M ? m / hw : -1;
            const int rem = m - (m / hw) * hw;
            ry[i] = rem / g.Win + g.pad; rx[i] = rem % g.Win + g.pad;
        }
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        float ra[A_OCT][8], rb[B_OCT][8];
        auto gather = [&](int i, int kk, int cnt, bool v8, float* v) {
            const int tap = kk / g.Cout, n = kk - tap * g.Cout;
            const int ky = tap / g.ks, kx = tap - ky * g.ks;
            const int ty = ry[i] - ky, tx = rx[i] - kx;
            if (ty < 0 || tx < 0) return;
            int oy = ty, ox = tx;
            if (g.stride == 2) { if ((ty | tx) & 1) return; oy = ty >> 1; ox = tx >> 1; }
            else if (g.stride != 1) { if (ty % g.stride || tx % g.stride) return; oy = ty / g.stride; ox = tx / g.stride; }
            if (oy >= g.Ho || ox >= g.Wo) return;
            const T* p = dOut + (((long)ri[i] * g.Ho + oy) * g.Wo + ox) * lddo + n;
            if (v8) load8<T>(p, v); else v[0] = to_f<T>(*p);
        };
        auto fetch = [&](int kt) {
            const int k = kt * BK + oct * 8;
#pragma unroll
            for (int i = 0; i < A_OCT; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ra[i][j] = 0.f;
                if (ri[i] < 0 || k >= K) continue;
                if (vec) gather(i, k, 8, true, ra[i]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k + j < K) gather(i, k + j, 1, false, &ra[i][j]);
                }
            }
#pragma unroll
            for (int i = 0; i < B_OCT; ++i) {
                const int r = r0 + i * 64, n = n0 + r;
                if (r < BN_ && n < g.Cin) load8<T>(Wt + (long)n * g.Kpt + k, rb[i]);
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
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN_ / WN) + j * 32 + (lane & 31);
            if (n >= g.Cin) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (m >= M) continue;
                    T* o = dIn + (long)m * lddi + n;
                    *o = from_f<T>(accumulate ? to_f<T>(*o) + acc[i][j][e] : acc[i][j][e]);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: C[i][j] = sum_m dOut[m][i] * a(m, j); pixels split over grid.z, fp32 atomics into dWk
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BJ = 128;

template <typename T, int BI>
__global__ __launch_bounds__(NT) void k_sconv_wgrad(const SConv g, const T* __restrict__ In, const T* __restrict__ dOut, long lddo,
                                                    float* __restrict__ dWk, float* __restrict__ dbias, int M, int K,
                                                    int rows_per_split) {
    constexpr int WI = BI >= 64 ? 2 : 1, WJ = 4 / WI, TM = BI / WI / 32, TN = BJ / WJ / 32;
    constexpr int L_OCT = (BK * BI / 8 + NT - 1) / NT;
    constexpr int R_OCT = BK * BJ / 8 / NT;
    constexpr int LPR = BI / 8, RPR = BJ / 8;
    __shared__ Tile<T, BI> As;      // rows = out channel i, k = pixel
    __shared__ Tile<T, BJ> Bs;      // rows = kernel index j, k = pixel
    __shared__ float bred[BK][BI + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WJ, wj = wave % WJ;
    const int j0 = blockIdx.x * BJ, i0 = blockIdx.y * BI;
    const bool avec = (g.Cin & 7) == 0 && (g.lda & 7) == 0 && aligned16<T>(In);
    const bool evec = (g.Cout & 7) == 0 && (lddo & 7) == 0 && aligned16<T>(dOut);
    const int m_begin = blockIdx.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const bool do_bias = dbias != nullptr && blockIdx.x == 0;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float bsum[L_OCT][8];
#pragma unroll
    for (int p = 0; p < L_OCT; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[p][j] = 0.f;
    float rl[L_OCT][8], rr[R_OCT][8];
    auto fetch = [&](int mc) {
#pragma unroll
        for (int p = 0; p < L_OCT; ++p) {
            const int idx = tid + p * NT;
            const int ml = idx / LPR, io = idx - ml * LPR;
            const int m = mc + ml, i = i0 + io * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) rl[p][j] = 0.f;
            if (ml < BK && m < m_end && i < g.Cout) {
                const T* q = dOut + (long)m * lddo + i;
                if (evec) load8<T>(q, rl[p]); else load8_guard<T>(q, min(8, g.Cout - i), rl[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < R_OCT; ++p) {
            const int idx = tid + p * NT;
            const int ml = idx / RPR, jo = idx - ml * RPR;
            const int m = mc + ml;
            const RowPix r = out_pixel(g, m, m_end);
            load_im2col8<T>(g, In, r, j0 + jo * 8, K, avec, rr[p]);
        }
    };
    if (m_begin < m_end) fetch(m_begin);
    for (int mc = m_begin; mc < m_end; mc += BK) {
#pragma unroll
        for (int p = 0; p < L_OCT; ++p) {
            const int idx = tid + p * NT;
            const int ml = idx / LPR, io = idx - ml * LPR;
            if (ml < BK) {
                store8_t(As, ml, io * 8, rl[p]);
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[p][j] += rl[p][j];
            }
        }
#pragma unroll
        for (int p = 0; p < R_OCT; ++p) {
            const int idx = tid + p * NT;
            const int ml = idx / RPR, jo = idx - ml * RPR;
            store8_t(Bs, ml, jo * 8, rr[p]);
        }
        __syncthreads();
        if (mc + BK < m_end) fetch(mc + BK);
        mma(As, Bs, wi * (BI / WI), wj * (BJ / WJ), lane, acc);
        __syncthreads();
    }
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) {
        const int j = j0 + wj * (BJ / WJ) + jt * 32 + (lane & 31);
        if (j >= K) continue;
#pragma unroll
        for (int it = 0; it < TM; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = i0 + wi * (BI / WI) + it * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (i < g.Cout) atomicAdd(dWk + (long)i * g.Kp + j, acc[it][jt][q]);
            }
    }
    if (do_bias) {
#pragma unroll
        for (int p = 0; p < L_OCT; ++p) {
            const int idx = tid + p * NT;
            const int ml = idx / LPR, io = idx - ml * LPR;
            if (ml < BK) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bred[ml][io * 8 + j] = bsum[p][j];
            }
        }
        __syncthreads();
        if (tid < BI && i0 + tid < g.Cout) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < BK; ++r) s += bred[r][tid];
            atomicAdd(dbias + i0 + tid, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm (one group) + SiLU
// ---------------------------------------------------------------------------------------------------------------------
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU instructions): the GroupNorm passes evaluate this once per element
__device__ __forceinline__ float silu(float u) { return u * __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
__device__ __forceinline__ float dsilu(float u) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-u)); return s * (1.f + u * (1.f - s)); }

constexpr int GN_CHUNK = 32768;       // elements of one image reduced by one workgroup

__device__ __forceinline__ void block_sum2(double& a, double& b, double (*red)[2]) {
    a = wave_sum(a); b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = a; red[wave][1] = b; }
    __syncthreads();
    a = 0; b = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[w][0]; b += red[w][1]; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_stats(const GnArgs a) {
    __shared__ double red[4][2];
    const T* X = reinterpret_cast<const T*>(a.X);
    const int img = blockIdx.y;
    const long per = (long)a.HW * a.C;
    const long e0 = (long)blockIdx.x * GN_CHUNK, e1 = min(per, e0 + GN_CHUNK);
    double s = 0, ss = 0;
    if ((a.C & 7) == 0 && (a.ldx & 7) == 0) {
        const int c8 = a.C >> 3;
        for (long o = e0 / 8 + threadIdx.x; o < e1 / 8; o += 256) {
            const long px = o / c8; const int c = (int)(o - px * c8) * 8;
            float v[8];
            load8<T>(X + ((long)img * a.HW + px) * a.ldx + c, v);
            float ls = 0.f, lss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { ls += v[j]; lss += v[j] * v[j]; }
            s += ls; ss += lss;
        }
    } else {
        for (long o = e0 + threadIdx.x; o < e1; o += 256) {
            const long px = o / a.C; const int c = (int)(o - px * a.C);
            const float v = to_f<T>(X[((long)img * a.HW + px) * a.ldx + c]);
            s += v; ss += (double)v * v;
        }
    }
    block_sum2(s, ss, red);
    if (threadIdx.x == 0) { atomicAdd(a.stats + img * 2, s); atomicAdd(a.stats + img * 2 + 1, ss); }
}

__device__ __forceinline__ void gn_mean_rstd(const double* stats, int img, double cnt, float eps, float& mean, float& rstd) {
    const double m = stats[img * 2] / cnt;
    double var = stats[img * 2 + 1] / cnt - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_act(const GnArgs a, T* __restrict__ Out, long ldo) {
    const T* X = reinterpret_cast<const T*>(a.X);
    const int img = blockIdx.y;
    float mean, rstd;
    gn_mean_rstd(a.stats, img, (double)a.HW * a.C, a.eps, mean, rstd);
    const long per = (long)a.HW * a.C;
    const bool vec = (a.C & 7) == 0 && (a.ldx & 7) == 0 && (ldo & 7) == 0;
    if (vec) {
        const int c8 = a.C >> 3;
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < per / 8; o += (long)gridDim.x * 256) {
            const long px = o / c8; const int c = (int)(o - px * c8) * 8;
            float v[8];
            load8<T>(X + ((long)img * a.HW + px) * a.ldx + c, v);
            T* dst = Out + ((long)img * a.HW + px) * ldo + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = (v[j] - mean) * rstd * a.gamma[c + j] + a.beta[c + j];
                dst[j] = from_f<T>(a.act ? silu(u) : u);
            }
        }
    } else {
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < per; o += (long)gridDim.x * 256) {
            const long px = o / a.C; const int c = (int)(o - px * a.C);
            const float u = (to_f<T>(X[((long)img * a.HW + px) * a.ldx + c]) - mean) * rstd * a.gamma[c] + a.beta[c];
            Out[((long)img * a.HW + px) * ldo + c] = from_f<T>(a.act ? silu(u) : u);
        }
    }
}

// per image: sum gamma*dU, sum gamma*dU*xhat (fp64 atomics); per channel: dgamma += sum dU*xhat, dbeta += sum dU.
// A workgroup owns `rows` consecutive pixels of one image and all channels: thread t handles channel group (t % cg), pixel
// lane (t / cg), so the per-channel sums reduce over the pixel lanes through LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_gn_bwd_reduce(const GnArgs a, const T* __restrict__ dA, long ldda, double* bsum,
                                                       float* dgamma, float* dbeta, int rows) {
    __shared__ double red[4][2];
    extern __shared__ float chs[];                 // [2][C]
    const T* X = reinterpret_cast<const T*>(a.X);
    const int img = blockIdx.y;
    float mean, rstd;
    gn_mean_rstd(a.stats, img, (double)a.HW * a.C, a.eps, mean, rstd);
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) chs[c] = 0.f;
    __syncthreads();
    const long p0 = (long)blockIdx.x * rows, p1 = min((long)a.HW, p0 + rows);
    double s1 = 0, s2 = 0;
    const int cols = min(a.C, 256), prl = 256 / cols;                        // pixel lanes per workgroup
    const int tc = threadIdx.x % cols, tp = threadIdx.x / cols;
    for (int c = tc; c < a.C; c += cols) {
        const float gmm = a.gamma[c], bt = a.beta[c];
        float dg = 0.f, db = 0.f;
        if (tp < prl) {
            for (long px = p0 + tp; px < p1; px += prl) {
                const long row = (long)img * a.HW + px;
                const float xh = (to_f<T>(X[row * a.ldx + c]) - mean) * rstd;
                const float u = xh * gmm + bt;
                const float du = to_f<T>(dA[row * ldda + c]) * (a.act ? dsilu(u) : 1.f);
                dg += du * xh; db += du;
                s1 += (double)(gmm * du); s2 += (double)(gmm * du * xh);
            }
        }
        atomicAdd(&chs[c], dg); atomicAdd(&chs[a.C + c], db);
    }
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) { atomicAdd(bsum + img * 2, s1); atomicAdd(bsum + img * 2 + 1, s2); }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += 256) { atomicAdd(dgamma + c, chs[c]); atomicAdd(dbeta + c, chs[a.C + c]); }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const GnArgs a, const T* __restrict__ dA, long ldda, const double* bsum,
                                                      T* __restrict__ dX, long lddx, int accumulate) {
    const T* X = reinterpret_cast<const T*>(a.X);
    const int img = blockIdx.y;
    float mean, rstd;
    const double cnt = (double)a.HW * a.C;
    gn_mean_rstd(a.stats, img, cnt, a.eps, mean, rstd);
    const float c1 = (float)(bsum[img * 2] / cnt), c2 = (float)(bsum[img * 2 + 1] / cnt);
    const long per = (long)a.HW * a.C;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < per; o += (long)gridDim.x * 256) {
        const long px = o / a.C; const int c = (int)(o - px * a.C);
        const long row = (long)img * a.HW + px;
        const float xh = (to_f<T>(X[row * a.ldx + c]) - mean) * rstd;
        const float gmm = a.gamma[c];
        const float u = xh * gmm + a.beta[c];
        const float du = to_f<T>(dA[row * ldda + c]) * (a.act ? dsilu(u) : 1.f);
        const float dx = rstd * (gmm * du - c1 - xh * c2);
        T* o2 = dX + row * lddx + c;
        *o2 = from_f<T>(accumulate ? to_f<T>(*o2) + dx : dx);
    }
}

// ---- dense-row vector variants (ldx == ldo == C, C % 8 == 0, HW*C < 2^31): 16 B per lane, no divisions, gamma/beta from LDS ----
template <typename T> __device__ __forceinline__ void store8g(T* p, const float v[8]);
template <> __device__ __forceinline__ void store8g<float>(float* p, const float v[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8g<bf16>(bf16* p, const float v[8]) {
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(v[j]);
    *reinterpret_cast<u16x8*>(p) = o;
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_act_v(const GnArgs a, T* __restrict__ Out) {
    extern __shared__ __attribute__((aligned(16))) float gb[];          // [2][C]
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) gb[c] = c < a.C ? a.gamma[c] : a.beta[c - a.C];
    __syncthreads();
    const int img = blockIdx.y;
    float mean, rstd;
    gn_mean_rstd(a.stats, img, (double)a.HW * a.C, a.eps, mean, rstd);
    const int per8 = a.HW * a.C / 8, stride = gridDim.x * 256;
    const T* X = reinterpret_cast<const T*>(a.X) + (long)img * a.HW * a.C;
    T* O = Out + (long)img * a.HW * a.C;
    int o = blockIdx.x * 256 + threadIdx.x;
    int c = (int)(((long)o * 8) % a.C);
    const int cstep = (int)(((long)stride * 8) % a.C);
    auto one = [&](int oo, int cc, float (&v)[8]) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + cc), g1 = *reinterpret_cast<const f32x4*>(gb + cc + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb + a.C + cc), b1 = *reinterpret_cast<const f32x4*>(gb + a.C + cc + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float u = (v[j] - mean) * rstd * (j < 4 ? g0[j & 3] : g1[j & 3]) + (j < 4 ? b0[j & 3] : b1[j & 3]);
            v[j] = a.act ? silu(u) : u;
        }
        store8g<T>(O + (long)oo * 8, v);
    };
    for (; o + stride < per8; o += 2 * stride) {            // two chunks per trip, both loads requested first
        float v0[8], v1[8];
        load8<T>(X + (long)o * 8, v0);
        load8<T>(X + (long)(o + stride) * 8, v1);
        int c1 = c + cstep; if (c1 >= a.C) c1 -= a.C;
        one(o, c, v0);
        one(o + stride, c1, v1);
        c = c1 + cstep; if (c >= a.C) c -= a.C;
    }
    if (o < per8) {
        float v[8];
        load8<T>(X + (long)o * 8, v);
        one(o, c, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_bwd_apply_v(const GnArgs a, const T* __restrict__ dA, const double* bsum, T* __restrict__ dX,
                                                        int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float gb[];          // [2][C]
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) gb[c] = c < a.C ? a.gamma[c] : a.beta[c - a.C];
    __syncthreads();
    const int img = blockIdx.y;
    float mean, rstd;
    const double cnt = (double)a.HW * a.C;
    gn_mean_rstd(a.stats, img, cnt, a.eps, mean, rstd);
    const float c1 = (float)(bsum[img * 2] / cnt), c2 = (float)(bsum[img * 2 + 1] / cnt);
    const int per8 = a.HW * a.C / 8, stride = gridDim.x * 256;
    const long base = (long)img * a.HW * a.C;
    const T* X = reinterpret_cast<const T*>(a.X) + base;
    int o = blockIdx.x * 256 + threadIdx.x;
    int c = (int)(((long)o * 8) % a.C);
    const int cstep = (int)(((long)stride * 8) % a.C);
    for (; o < per8; o += stride) {
        float x[8], d[8], acc8[8];
        load8<T>(X + (long)o * 8, x);
        load8<T>(dA + base + (long)o * 8, d);
        if (accumulate) load8<T>(dX + base + (long)o * 8, acc8);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + c), g1 = *reinterpret_cast<const f32x4*>(gb + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb + a.C + c), b1 = *reinterpret_cast<const f32x4*>(gb + a.C + c + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gmm = j < 4 ? g0[j & 3] : g1[j & 3], bt = j < 4 ? b0[j & 3] : b1[j & 3];
            const float xh = (x[j] - mean) * rstd;
            const float u = xh * gmm + bt;
            const float du = d[j] * (a.act ? dsilu(u) : 1.f);
            const float dx = rstd * (gmm * du - c1 - xh * c2);
            x[j] = accumulate ? acc8[j] + dx : dx;
        }
        store8g<T>(dX + base + (long)o * 8, x);
        c += cstep; if (c >= a.C) c -= a.C;
    }
}

// reduce: C/8 channel groups x 256/(C/8) pixel lanes per workgroup (C in {64, 128, 256, 512}: C/8 divides 256); a thread keeps its
// 8 channels for all its pixels, so dgamma / dbeta partial sums live in registers
template <typename T>
__global__ __launch_bounds__(256) void k_gn_bwd_reduce_v(const GnArgs a, const T* __restrict__ dA, double* bsum, float* dgamma, float* dbeta,
                                                         int rows) {
    __shared__ double red[4][2];
    extern __shared__ __attribute__((aligned(16))) float chs[];         // [2][C]
    const int img = blockIdx.y;
    float mean, rstd;
    gn_mean_rstd(a.stats, img, (double)a.HW * a.C, a.eps, mean, rstd);
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) chs[c] = 0.f;
    __syncthreads();
    const int c8 = a.C >> 3, lanes = 256 / c8;
    const int c = (threadIdx.x % c8) * 8, pl = threadIdx.x / c8;
    float gmm[8], bt[8], dg[8], db[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gmm[j] = a.gamma[c + j]; bt[j] = a.beta[c + j]; dg[j] = 0.f; db[j] = 0.f; }
    const long base = (long)img * a.HW * a.C;
    const T* X = reinterpret_cast<const T*>(a.X) + base;
    float f1 = 0.f, f2 = 0.f;
    double s1 = 0, s2 = 0;
    int k = 0;
    (void)rows;                                                         // pixels are dealt round-robin over the (few) workgroups of an image
    // four pixels per trip, all eight 16-B loads requested before the first use (round 4: one load -> use per trip kept one row in flight per
    // thread and the kernel moved its two tensors at 2.2 TB/s: 10.2 ms per SDXL step in 75 launches)
    const int pstep = gridDim.x * lanes;
    int px = blockIdx.x * lanes + pl;
    for (; px + 3 * pstep < a.HW; px += 4 * pstep) {
        float x[4][8], d[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            load8<T>(X + (long)(px + q * pstep) * a.C + c, x[q]);
            load8<T>(dA + base + (long)(px + q * pstep) * a.C + c, d[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (x[q][j] - mean) * rstd;
                const float u = xh * gmm[j] + bt[j];
                const float du = d[q][j] * (a.act ? dsilu(u) : 1.f);
                dg[j] += du * xh; db[j] += du;
                f1 += gmm[j] * du; f2 += gmm[j] * du * xh;
            }
        k += 4;
        if (k >= 8) { s1 += f1; s2 += f2; f1 = 0.f; f2 = 0.f; k = 0; }         // short fp32 runs, fp64 across them
    }
    for (; px < a.HW; px += pstep) {
        float x[8], d[8];
        load8<T>(X + (long)px * a.C + c, x);
        load8<T>(dA + base + (long)px * a.C + c, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (x[j] - mean) * rstd;
            const float u = xh * gmm[j] + bt[j];
            const float du = d[j] * (a.act ? dsilu(u) : 1.f);
            dg[j] += du * xh; db[j] += du;
            f1 += gmm[j] * du; f2 += gmm[j] * du * xh;
        }
        if (++k == 8) { s1 += f1; s2 += f2; f1 = 0.f; f2 = 0.f; k = 0; }
    }
    s1 += f1; s2 += f2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { atomicAdd(&chs[c + j], dg[j]); atomicAdd(&chs[a.C + c + j], db[j]); }
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) { atomicAdd(bsum + img * 2, s1); atomicAdd(bsum + img * 2 + 1, s2); }
    __syncthreads();
    for (int cc = threadIdx.x; cc < a.C; cc += 256) { atomicAdd(dgamma + cc, chs[cc]); atomicAdd(dbeta + cc, chs[a.C + cc]); }
}

template <typename T>
__global__ void k_cast_f32_to(const float* src, long lds, T* dst, long ldd, long rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols; const int c = (int)(i - r * cols);
    dst[r * ldd + c] = from_f<T>(src[r * lds + c]);
}
template <typename T>
__global__ void k_add_into(T* dst, long ldd, const T* src, long lds, long rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols; const int c = (int)(i - r * cols);
    dst[r * ldd + c] = from_f<T>(to_f<T>(dst[r * ldd + c]) + to_f<T>(src[r * lds + c]));
}

int grid_for(int M) { const int t = cdiv(M, BM); return t < 1024 ? t : 1024; }

template <typename T>
int fwd_t(const SConv& g, const void* In, const void* Wk, const float* bias, const void* Res, long ldres, void* Out, long ldo,
          int out_f32, hipStream_t st) {
    const int M = g.n * g.Ho * g.Wo, K = g.ks * g.ks * g.Cin;
    const T* in = reinterpret_cast<const T*>(In); const T* wk = reinterpret_cast<const T*>(Wk); const T* res = reinterpret_cast<const T*>(Res);
    const int gx = grid_for(M);
    if (g.Cout <= 32) hipLaunchKernelGGL((k_sconv_fwd<T, 32>), dim3(gx, 1), dim3(NT), 0, st, g, in, wk, bias, res, ldres, Out, ldo, out_f32, M, K);
    else if (g.Cout <= 64) hipLaunchKernelGGL((k_sconv_fwd<T, 64>), dim3(gx, 1), dim3(NT), 0, st, g, in, wk, bias, res, ldres, Out, ldo, out_f32, M, K);
    else hipLaunchKernelGGL((k_sconv_fwd<T, 128>), dim3(gx, cdiv(g.Cout, 128)), dim3(NT), 0, st, g, in, wk, bias, res, ldres, Out, ldo, out_f32, M, K);
    TCVN_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int dgrad_t(const SConv& g, const void* dOut, long lddo, const void* Wt, void* dIn, long lddi, int accumulate, hipStream_t st) {
    const int M = g.n * g.Hin * g.Win, K = g.ks * g.ks * g.Cout;
    const T* d = reinterpret_cast<const T*>(dOut); const T* wt = reinterpret_cast<const T*>(Wt); T* o = reinterpret_cast<T*>(dIn);
    const int gx = grid_for(M);
    if (g.Cin <= 32) hipLaunchKernelGGL((k_sconv_dgrad<T, 32>), dim3(gx, 1), dim3(NT), 0, st, g, d, lddo, wt, o, lddi, accumulate, M, K);
    else if (g.Cin <= 64) hipLaunchKernelGGL((k_sconv_dgrad<T, 64>), dim3(gx, 1), dim3(NT), 0, st, g, d, lddo, wt, o, lddi, accumulate, M, K);
    else hipLaunchKernelGGL((k_sconv_dgrad<T, 128>), dim3(gx, cdiv(g.Cin, 128)), dim3(NT), 0, st, g, d, lddo, wt, o, lddi, accumulate, M, K);
    TCVN_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int wgrad_t(const SConv& g, const void* In, const void* dOut, long lddo, float* dWk, float* dbias, hipStream_t st) {
    const int M = g.n * g.Ho * g.Wo, K = g.ks * g.ks * g.Cin;
    const int jt = cdiv(g.Kp, BJ);
    const int BIv = g.Cout <= 32 ? 32 : g.Cout <= 64 ? 64 : 128;
    const int it = cdiv(g.Cout, BIv);
    int split = cdiv(1024, jt * it);
    const int max_split = cdiv(M, 512);
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    const int rows = (int)round_up(cdiv(M, split), BK);
    split = cdiv(M, rows);
    const dim3 grid(jt, it, split);
    const T* in = reinterpret_cast<const T*>(In); const T* d = reinterpret_cast<const T*>(dOut);
    if (BIv == 32) hipLaunchKernelGGL((k_sconv_wgrad<T, 32>), grid, dim3(NT), 0, st, g, in, d, lddo, dWk, dbias, M, K, rows);
    else if (BIv == 64) hipLaunchKernelGGL((k_sconv_wgrad<T, 64>), grid, dim3(NT), 0, st, g, in, d, lddo, dWk, dbias, M, K, rows);
    else hipLaunchKernelGGL((k_sconv_wgrad<T, 128>), grid, dim3(NT), 0, st, g, in, d, lddo, dWk, dbias, M, K, rows);
    TCVN_LAUNCH_CHECK();
    return 0;
}

const char* label(char* buf, size_t cap, const char* what, const SConv& g) {
    snprintf(buf, cap, "k_sconv_%s<%s,%dx%d/%d,%d->%d>", what, g.mode == MODE_F32 ? "float" : "bf16", g.ks, g.ks, g.stride, g.Cin, g.Cout);
    return buf;
}

}  // namespace

int sconv_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, const void* Res, long ldres, void* Out, long ldo,
              int out_f32, hipStream_t st) {
    if (g.n <= 0) return 0;
    if (g.Kp % BK != 0 || g.Kp < g.ks * g.ks * g.Cin) return -2;
    const double M = (double)g.n * g.Ho * g.Wo, es = g.mode == MODE_F32 ? 4.0 : 2.0;
    char nm[96];
    ProfScope ps(label(nm, sizeof(nm), "fwd", g), 2.0 * M * g.Cout * g.ks * g.ks * g.Cin,
                 es * ((double)g.n * g.Hin * g.Win * g.Cin + M * g.Cout * (Res ? 2.0 : 1.0)), st);
    // g.stats: the consumer GroupNorm's statistics.  The tile kernels accumulate them in their epilogue; any other path is followed by
    // the stand-alone statistics pass, so the caller never runs one for a convolution it handed `stats` to.
    int rc;
    bool fused = true;
    if (sconv3_c64_fwd_ok(g, In, Wk, Res, ldres, Out, ldo, out_f32)) rc = sconv3_c64_fwd(g, In, Wk, bias, Res, Out, g.stats, st);
    else if (sconv_in_fwd_ok(g, In, Wk, Res, Out, ldo, out_f32)) rc = sconv_in_fwd(g, In, Wk, bias, Out, g.stats, st);
    else if (sconv3_g_fwd_ok(g, In, Wk, Res, ldres, Out, ldo, out_f32)) {
        fused = sconv3_g_fuses_stats(g);
        rc = sconv3_g_fwd(g, In, Wk, bias, Res, Out, g.stats, st);
    } else if (sconv3_s2_fwd_ok(g, In, Wk, Res, Out, ldo, out_f32)) rc = sconv3_s2_fwd(g, In, Wk, bias, Out, g.stats, st);
    else {
        fused = false;
        rc = g.mode == MODE_F32 ? fwd_t<float>(g, In, Wk, bias, Res, ldres, Out, ldo, out_f32, st)
                                : fwd_t<bf16>(g, In, Wk, bias, Res, ldres, Out, ldo, out_f32, st);
    }
    if (rc == 0 && g.stats != nullptr && !fused) {
        GnArgs ga{g.mode, Out, ldo, g.n, g.Ho * g.Wo, g.Cout, nullptr, nullptr, 0.f, 0, g.stats};
        rc = gn_stats(ga, st);
    }
    return rc;
}
int sconv_dgrad(const SConv& g, const void* dOut, long lddo, const void* Wt, void* dIn, long lddi, int accumulate, hipStream_t st) {
    if (g.n <= 0) return 0;
    if (g.Kpt % BK != 0 || g.Kpt < g.ks * g.ks * g.Cout) return -2;
    const double Mi = (double)g.n * g.Hin * g.Win, es = g.mode == MODE_F32 ? 4.0 : 2.0;
    char nm[96];
    ProfScope ps(label(nm, sizeof(nm), "dgrad", g), 2.0 * (double)g.n * g.Ho * g.Wo * g.Cout * g.ks * g.ks * g.Cin,
                 es * ((double)g.n * g.Ho * g.Wo * g.Cout + Mi * g.Cin), st);
    if (sconv3_c64_dgrad_ok(g, dOut, lddo, Wt, dIn, lddi)) return sconv3_c64_dgrad(g, dOut, Wt, dIn, accumulate, st);
    if (sconv3_g_dgrad_ok(g, dOut, lddo, Wt, dIn, lddi)) return sconv3_g_dgrad(g, dOut, Wt, dIn, accumulate, st);
    if (sconv3_s2_dgrad_ok(g, dOut, lddo, Wt, dIn, lddi)) return sconv3_s2_dgrad(g, dOut, Wt, dIn, accumulate, st);
    return g.mode == MODE_F32 ? dgrad_t<float>(g, dOut, lddo, Wt, dIn, lddi, accumulate, st)
                              : dgrad_t<bf16>(g, dOut, lddo, Wt, dIn, lddi, accumulate, st);
}
int sconv_wgrad(const SConv& g, const void* In, const void* dOut, long lddo, float* dWk, float* dbias, hipStream_t st) {
    if (g.n <= 0) return 0;
    const double M = (double)g.n * g.Ho * g.Wo, es = g.mode == MODE_F32 ? 4.0 : 2.0;
    char nm[96];
    ProfScope ps(label(nm, sizeof(nm), "wgrad", g), 2.0 * M * g.Cout * g.ks * g.ks * g.Cin,
                 es * ((double)g.n * g.Hin * g.Win * g.Cin + M * g.Cout), st);
    if (sconv3_c64_wgrad_ok(g, In, dOut, lddo)) return sconv3_c64_wgrad(g, In, dOut, dWk, dbias, st);
    if (sconv_in_wgrad_ok(g, dOut, lddo)) return sconv_in_wgrad(g, In, dOut, dWk, dbias, st);
    if (sconv3_g_wgrad_ok(g, In, dOut, lddo)) return sconv3_g_wgrad(g, In, dOut, dWk, dbias, st);
    if (sconv3_s2_wgrad_ok(g, In, dOut, lddo)) return sconv3_s2_wgrad(g, In, dOut, dWk, dbias, st);
    return g.mode == MODE_F32 ? wgrad_t<float>(g, In, dOut, lddo, dWk, dbias, st) : wgrad_t<bf16>(g, In, dOut, lddo, dWk, dbias, st);
}

namespace {
// dense rows, whole 16-B groups, 32-bit element indices within an image
bool gn_dense(const GnArgs& a) {
    return (a.C & 7) == 0 && a.ldx == a.C && (long)a.HW * a.C < (1L << 31) && (reinterpret_cast<uintptr_t>(a.X) & 15) == 0;
}
}  // namespace

int gn_stats(const GnArgs& a, hipStream_t st) {
    if (a.n <= 0) return 0;
    const dim3 grid(cdiv((long)a.HW * a.C, GN_CHUNK), a.n);
    if (a.mode == MODE_F32) hipLaunchKernelGGL(k_gn_stats<float>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_gn_stats<bf16>, grid, dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int gn_act(const GnArgs& a, void* Out, long ldo, hipStream_t st) {
    if (a.n <= 0) return 0;
    const long per = (long)a.HW * a.C;
    int gx = cdiv(per / 8 + 1, 256);
    if (gx > 2048) gx = 2048;
    const dim3 grid(gx, a.n);
    if (gn_dense(a) && ldo == a.C) {
        const size_t smem = 2 * (size_t)a.C * sizeof(float);
        if (a.mode == MODE_F32) hipLaunchKernelGGL(k_gn_act_v<float>, grid, dim3(256), smem, st, a, reinterpret_cast<float*>(Out));
        else hipLaunchKernelGGL(k_gn_act_v<bf16>, grid, dim3(256), smem, st, a, reinterpret_cast<bf16*>(Out));
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.mode == MODE_F32) hipLaunchKernelGGL(k_gn_act<float>, grid, dim3(256), 0, st, a, reinterpret_cast<float*>(Out), ldo);
    else hipLaunchKernelGGL(k_gn_act<bf16>, grid, dim3(256), 0, st, a, reinterpret_cast<bf16*>(Out), ldo);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int gn_bwd_reduce(const GnArgs& a, const void* dA, long ldda, double* bsum, float* dgamma, float* dbeta, hipStream_t st) {
    if (a.n <= 0) return 0;
    int rows = GN_CHUNK / a.C;
    if (rows < 1) rows = 1;
    const dim3 grid(cdiv(a.HW, rows), a.n);
    const size_t smem = 2 * (size_t)a.C * sizeof(float);
    if (gn_dense(a) && ldda == a.C && 256 % (a.C / 8) == 0) {
        // few workgroups per image: every workgroup ends with 2C global atomics on the same dgamma / dbeta words
        int gxv = cdiv(a.HW, 4096);
        const int cap = a.n >= 64 ? 8 : a.n >= 8 ? 32 : 128;
        if (gxv > cap) gxv = cap;
        const dim3 grid(gxv, a.n);
        if (a.mode == MODE_F32)
            hipLaunchKernelGGL(k_gn_bwd_reduce_v<float>, grid, dim3(256), smem, st, a, reinterpret_cast<const float*>(dA), bsum, dgamma, dbeta, rows);
        else
            hipLaunchKernelGGL(k_gn_bwd_reduce_v<bf16>, grid, dim3(256), smem, st, a, reinterpret_cast<const bf16*>(dA), bsum, dgamma, dbeta, rows);
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.mode == MODE_F32)
        hipLaunchKernelGGL(k_gn_bwd_reduce<float>, grid, dim3(256), smem, st, a, reinterpret_cast<const float*>(dA), ldda, bsum, dgamma, dbeta, rows);
    else
        hipLaunchKernelGGL(k_gn_bwd_reduce<bf16>, grid, dim3(256), smem, st, a, reinterpret_cast<const bf16*>(dA), ldda, bsum, dgamma, dbeta, rows);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int gn_bwd_apply(const GnArgs& a, const void* dA, long ldda, const double* bsum, void* dX, long lddx, int accumulate, hipStream_t st) {
    if (a.n <= 0) return 0;
    const long per = (long)a.HW * a.C;
    int gx = cdiv(per, 256 * 4);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, a.n);
    if (gn_dense(a) && ldda == a.C && lddx == a.C) {
        int gv = cdiv(per / 8, 256 * 2);
        if (gv > 2048) gv = 2048;
        if (gv < 1) gv = 1;
        const size_t smem = 2 * (size_t)a.C * sizeof(float);
        if (a.mode == MODE_F32)
            hipLaunchKernelGGL(k_gn_bwd_apply_v<float>, dim3(gv, a.n), dim3(256), smem, st, a, reinterpret_cast<const float*>(dA), bsum, reinterpret_cast<float*>(dX), accumulate);
        else
            hipLaunchKernelGGL(k_gn_bwd_apply_v<bf16>, dim3(gv, a.n), dim3(256), smem, st, a, reinterpret_cast<const bf16*>(dA), bsum, reinterpret_cast<bf16*>(dX), accumulate);
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.mode == MODE_F32)
        hipLaunchKernelGGL(k_gn_bwd_apply<float>, grid, dim3(256), 0, st, a, reinterpret_cast<const float*>(dA), ldda, bsum, reinterpret_cast<float*>(dX), lddx, accumulate);
    else
        hipLaunchKernelGGL(k_gn_bwd_apply<bf16>, grid, dim3(256), 0, st, a, reinterpret_cast<const bf16*>(dA), ldda, bsum, reinterpret_cast<bf16*>(dX), lddx, accumulate);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int cast_f32_to(int mode, const float* src, long lds, void* dst, long ldd, long rows, int cols, hipStream_t st) {
    if (rows <= 0) return 0;
    const int gx = cdiv(rows * cols, 256);
    if (mode == MODE_F32) hipLaunchKernelGGL(k_cast_f32_to<float>, dim3(gx), dim3(256), 0, st, src, lds, reinterpret_cast<float*>(dst), ldd, rows, cols);
    else hipLaunchKernelGGL(k_cast_f32_to<bf16>, dim3(gx), dim3(256), 0, st, src, lds, reinterpret_cast<bf16*>(dst), ldd, rows, cols);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int add_into(int mode, void* dst, long ldd, const void* src, long lds, long rows, int cols, hipStream_t st) {
    if (rows <= 0) return 0;
    const long tot = rows * cols;
    const int gx = cdiv(tot, 256);
    if (mode == MODE_F32) hipLaunchKernelGGL(k_add_into<float>, dim3(gx), dim3(256), 0, st, reinterpret_cast<float*>(dst), ldd, reinterpret_cast<const float*>(src), lds, rows, cols);
    else hipLaunchKernelGGL(k_add_into<bf16>, dim3(gx), dim3(256), 0, st, reinterpret_cast<bf16*>(dst), ldd, reinterpret_cast<const bf16*>(src), lds, rows, cols);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
