// Backward convolutions of the DenseNet embedder (autograd of layers/dense_net.py:18-45,84-94,112-118 in the reference,
// SURVEY.md 2.3 K14): data gradients and weight gradients as implicit GEMMs on the matrix cores, with the PReLU and
// BatchNorm backward of the neighbouring layers fused into the operand loaders and the epilogue.
//
// Gradient bookkeeping (see tcvn_ops.h EffSrc): a produced tensor X keeps G = sum over consumers of (gamma*rstd)*dU and
// per-channel (P, Q); its true gradient is G + P*X + Q, formed on the fly by the kernels that consume it.  The
// per-channel sums every BatchNorm backward needs leave the dgrad epilogue as one partial row per workgroup.
#include "conv_tile.h"
#include "prof.h"

namespace tcvn {

using namespace convk;

namespace {

template <typename T>
__device__ __forceinline__ void load_eff8(const EffSrc& e, const T* __restrict__ G, const T* __restrict__ X, long m, int n,
                                          bool vec, float v[8]) {
    const int cnt = min(8, e.N - n);
    float g[8], x[8];
    const T* gp = G + m * e.ldg + e.c_off + n;
    const T* xp = X + m * e.ldx + e.c_off + n;
    if (vec) { load8<T>(gp, g); load8<T>(xp, x); }
    else { load8_guard<T>(gp, cnt, g); load8_guard<T>(xp, cnt, x); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = 0.f;
        if (j < cnt) {
            t = g[j] + e.P[n + j] * x[j] + e.Q[n + j];
            if (e.drop_p > 0.f) t *= drop_scale_mn(e.drop_p, e.seed, e.stream_id, m, n + j, e.N);
        }
        v[j] = t;
    }
}

template <typename T>
__device__ __forceinline__ bool eff_vec(const EffSrc& e) {
    constexpr unsigned ALIGN = sizeof(T) * 8 - 1;
    return ((e.ldg & 7) == 0) && ((e.ldx & 7) == 0) && ((e.c_off & 7) == 0) && ((e.N & 7) == 0) &&
           ((reinterpret_cast<uintptr_t>(e.G) & ALIGN) == 0) && ((reinterpret_cast<uintptr_t>(e.X) & ALIGN) == 0);
}

// four consecutive channels of a row (16 B of fp32, 8 B of bf16); the address is a multiple of that size
template <typename T> __device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <> __device__ __forceinline__ void load4<bf16>(const bf16* p, float v[4]) {
    const u16x4 a = *reinterpret_cast<const u16x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bf2f(a[j]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float v[4]) {
    u16x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = f2bf(v[j]);
    *reinterpret_cast<u16x4*>(p) = a;
}

// ---------------------------------------------------------------------------------------------------------------------
// dgrad
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DMODE, int BN_>
__global__ __launch_bounds__(NT) void k_conv_dgrad(const ConvDgradArgs g) {
    constexpr int WN = BN_ >= 64 ? 2 : 1, WM = 4 / WN, TM = BM / WM / 32, TN = BN_ / WN / 32;
    constexpr int A_OCT = BM * BK / 8 / NT;
    constexpr int B_OCT = (BN_ * BK / 8 + NT - 1) / NT;

    __shared__ Tile<T, BM> As;
    __shared__ Tile<T, BN_> Bs;
    __shared__ double red[WM][BN_][3];
    __shared__ float cx[DMODE == DG_1X1_POOL ? 4 : 1][DMODE == DG_1X1_POOL ? 32 * 33 : 1];      // pooled epilogue: wave-private 32 x 32 exchange patch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN_;
    const EffSrc& e = g.e;
    const T* __restrict__ G = reinterpret_cast<const T*>(e.G);
    const T* __restrict__ X = reinterpret_cast<const T*>(e.X);
    const T* __restrict__ Wt = reinterpret_cast<const T*>(g.Wt);
    const T* __restrict__ Xin = reinterpret_cast<const T*>(g.Xin);
    T* __restrict__ Gout = reinterpret_cast<T*>(g.Gout);
    const int K = (DMODE == DG_3X3 ? 9 : 1) * e.N;
    const int mtiles = (g.M + BM - 1) / BM, ktiles = g.Kp / BK;
    const bool vec = eff_vec<T>(e);
    const int oct = tid & 3, r0 = tid >> 2;

    double s1[TN], s2[TN], s3[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0; s2[j] = 0; s3[j] = 0; }

    for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
        const int m0 = mt * BM;
        int rh[A_OCT], rw[A_OCT];
#pragma unroll
        for (int i = 0; i < A_OCT; ++i) {
            const int m = m0 + r0 + i * 64;
            rw[i] = DMODE == DG_3X3 ? m % g.W : 0;
            rh[i] = DMODE == DG_3X3 ? (m / g.W) % g.H : 0;
        }
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

        float ra[A_OCT][8], rb[B_OCT][8];
        auto fetch = [&](int kt) {
            const int k = kt * BK + oct * 8;
#pragma unroll
            for (int i = 0; i < A_OCT; ++i) {
                const int m = m0 + r0 + i * 64;
#pragma unroll
                for (int j = 0; j < 8; ++j) ra[i][j] = 0.f;
                if (m < g.M && k < K) {
                    if (DMODE == DG_3X3) {
                        const int tap = k / e.N, n = k - tap * e.N;
                        const int ky = tap / 3, kx = tap - ky * 3;
                        const int sh_ = rh[i] - (ky - 1), sw = rw[i] - (kx - 1);
                        if (sh_ >= 0 && sh_ < g.H && sw >= 0 && sw < g.W)
                            load_eff8<T>(e, G, X, (long)m - ((ky - 1) * g.W + (kx - 1)), n, vec, ra[i]);
                    } else {
                        load_eff8<T>(e, G, X, (long)m, k, vec, ra[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < B_OCT; ++i) {
                const int r = r0 + i * 64, n = n0 + r;
                if (r < BN_ && n < g.N) load8<T>(Wt + (long)n * g.Kp + k, rb[i]);
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

        // epilogue: PReLU + BatchNorm backward of the consumer norm whose input is Xin
        if (DMODE == DG_1X1_POOL && (g.N & 3) == 0) {
            // Pooled rows (a transition's data gradient): every accumulator element feeds FOUR source pixels (x read, G read-add-write).  In MFMA
            // layout that is 4-byte accesses, 12 per element -- the fp32 parity mode spent 6.7 ms per step here at a fifth of the HBM rate.  Each
            // 32 x 32 accumulator tile crosses a wave-private LDS patch instead, so that a lane owns four consecutive channels of a row
            // (16-B accesses of fp32) and all of a row's twelve accesses are requested before the first use.
            float* cw = &cx[wave][0];
            const int er = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN_ / WN) + j * 32 + c4;
                float sc[4], sh[4], sl[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const bool ok = n + k < g.N; sc[k] = ok ? g.sc[n + k] : 0.f; sh[k] = ok ? g.sh[n + k] : 0.f; sl[k] = ok ? g.sl[n + k] : 0.f; }
                float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f}, t3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) cw[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[i][j][q];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the patch is this wave's own: LDS operations of a wave complete in order
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 8 + er;
                        const int m = m0 + wm * (BM / WM) + i * 32 + row;
                        float cv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) cv[k] = cw[row * 33 + c4 + k];
                        if (m < g.M && n < g.N) {
                            const int hw = g.H * g.W;
                            const int img = m / hw, rem = m - img * hw;
                            const int ho = rem / g.W, wo = rem - ho * g.W;
                            const long p00 = ((long)img * g.Hin + 2 * ho) * g.Win + 2 * wo;
                            float xv[4][4], gv[4][4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const long px = p00 + (t >> 1) * g.Win + (t & 1);
                                load4<T>(Xin + px * g.ldxin + n, xv[t]);
                                if (g.accumulate) load4<T>(Gout + px * g.ldgo + n, gv[t]);
                                else { gv[t][0] = 0.f; gv[t][1] = 0.f; gv[t][2] = 0.f; gv[t][3] = 0.f; }
                            }
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const long px = p00 + (t >> 1) * g.Win + (t & 1);
                                float o[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const float x = xv[t][k];
                                    const float u = fmaf(x, sc[k], sh[k]);
                                    const float dA = 0.25f * cv[k];
                                    const float du = u > 0.f ? dA : sl[k] * dA;
                                    t1[k] += du; t2[k] = fmaf(du, x, t2[k]); t3[k] += u > 0.f ? 0.f : dA * u;
                                    o[k] = gv[t][k] + sc[k] * du;
                                }
                                store4<T>(Gout + px * g.ldgo + n, o);
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // all reads of the patch are done before the next tile overwrites it
                }
                // this lane's four channels -> the sums of column (lane & 31) of sub-tile j, as the reduction below expects them:
                // fold the 8 row lanes (lane bits 3..5), then lane l takes channel l & 3 of the group l >> 2
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double a = (double)t1[k], b = (double)t2[k], c = (double)t3[k];
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
                    // lanes 0..7 hold the totals of channels 4*lane + k: route them to lane 4*(lane & 7) + k
                    const double ra = __shfl(a, (lane & 31) >> 2), rb = __shfl(b, (lane & 31) >> 2), rc = __shfl(c, (lane & 31) >> 2);
                    if (((lane & 31) & 3) == k && lane < 32) { s1[j] += ra; s2[j] += rb; s3[j] += rc; }
                }
            }
        } else
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN_ / WN) + j * 32 + (lane & 31);            if (n >= g.N) continue;
            const float sc = g.sc[n], sh = g.sh[n], sl = g.sl[n];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    if (m >= g.M) continue;
                    if (DMODE == DG_1X1_POOL) {
                        const int hw = g.H * g.W;
                        const int img = m / hw, rem = m - img * hw;
                        const int ho = rem / g.W, wo = rem - ho * g.W;
                        const long p00 = ((long)img * g.Hin + 2 * ho) * g.Win + 2 * wo;
                        const float dA = 0.25f * acc[i][j][q];
                        // the four pixels of the window: all loads in flight before the first use (a dependent load per pixel costs a
                        // memory round trip each when few waves share the SIMD)
                        float xv[4], gv[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const long px = p00 + (t >> 1) * g.Win + (t & 1);
                            xv[t] = to_f<T>(Xin[px * g.ldxin + n]);
                            gv[t] = g.accumulate ? to_f<T>(Gout[px * g.ldgo + n]) : 0.f;
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const long px = p00 + (t >> 1) * g.Win + (t & 1);
                            const float x = xv[t];
                            const float u = fmaf(x, sc, sh);
                            const float du = u > 0.f ? dA : sl * dA;
                            s1[j] += du; s2[j] += (double)du * x; s3[j] += u > 0.f ? 0.f : dA * u;
                            Gout[px * g.ldgo + n] = from_f<T>(gv[t] + sc * du);
                        }
                    } else {
                        const float x = to_f<T>(Xin[(long)m * g.ldxin + n]);
                        const float u = fmaf(x, sc, sh);
                        const float dA = acc[i][j][q];
                        const float du = u > 0.f ? dA : sl * dA;
                        s1[j] += du; s2[j] += (double)du * x; s3[j] += u > 0.f ? 0.f : dA * u;
                        T* o = Gout + (long)m * g.ldgo + n;
                        const float val = sc * du;
                        *o = from_f<T>(g.accumulate ? to_f<T>(*o) + val : val);
                    }
                }
            }
        }
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        double a = s1[j], b = s2[j], c = s3[j];
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32); c += __shfl_xor(c, 32);
        if (lane < 32) {
            const int col = wn * (BN_ / WN) + j * 32 + lane;
            red[wm][col][0] = a; red[wm][col][1] = b; red[wm][col][2] = c;
        }
    }
    __syncthreads();
    if (tid < BN_ && n0 + tid < g.N) {
        double a = 0, b = 0, c = 0;
#pragma unroll
        for (int w = 0; w < WM; ++w) { a += red[w][tid][0]; b += red[w][tid][1]; c += red[w][tid][2]; }
        double* p = g.part + ((long)blockIdx.x * g.N + n0 + tid) * 3;
        p[0] = a; p[1] = b; p[2] = c;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad: C[i][j] = sum_m eff(m, i) * a(m, j); reduction over pixels split across grid.z, fp32 atomics into dWk
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BJ = 128;

template <typename T, int AMODE, int BI>
__global__ __launch_bounds__(NT) void k_conv_wgrad(const ConvWgradArgs g, int rows_per_split) {
    constexpr int WI = BI >= 64 ? 2 : 1, WJ = 4 / WI, TM = BI / WI / 32, TN = BJ / WJ / 32;
    constexpr int L_OCT = (BK * BI / 8 + NT - 1) / NT;        // octets of eff per thread per chunk
    constexpr int R_OCT = BK * BJ / 8 / NT;                   // 2
    constexpr int LPR = BI / 8, RPR = BJ / 8;                 // octets per row

    __shared__ Tile<T, BI> As;      // rows = out channel i, k = pixel
    __shared__ Tile<T, BJ> Bs;      // rows = kernel index j, k = pixel
    __shared__ float bred[BK][BI + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WJ, wj = wave % WJ;
    const int j0 = blockIdx.x * BJ, i0 = blockIdx.y * BI;
    const ConvFwdArgs& fa = g.fa;
    const EffSrc& e = g.e;
    const T* __restrict__ A = reinterpret_cast<const T*>(fa.A);
    const T* __restrict__ G = reinterpret_cast<const T*>(e.G);
    const T* __restrict__ X = reinterpret_cast<const T*>(e.X);
    constexpr unsigned ALIGN = sizeof(T) * 8 - 1;
    const bool avec = ((fa.lda & 7) == 0) && ((reinterpret_cast<uintptr_t>(A) & ALIGN) == 0) &&
                      (AMODE == A_3X3 ? (fa.C & 7) == 0 : (fa.K & 7) == 0) && AMODE != A_STEM;
    const bool evec = eff_vec<T>(e);
    const int m_begin = blockIdx.z * rows_per_split;
    const int m_end = min(fa.M, m_begin + rows_per_split);
    const bool do_bias = g.dbias != nullptr && blockIdx.x == 0;

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
            if (ml < BK && m < m_end && i < e.N) load_eff8<T>(e, G, X, (long)m, i, evec, rl[p]);
        }
#pragma unroll
        for (int p = 0; p < R_OCT; ++p) {
            const int idx = tid + p * NT;
            const int ml = idx / RPR, jo = idx - ml * RPR;
            const int m = mc + ml;
            if (m < m_end) {
                const RowInfo ri = row_info<T, AMODE>(fa, m);
                load_a8<T, AMODE>(fa, A, ri, j0 + jo * 8, avec, rr[p]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[p][j] = 0.f;
            }
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

    // epilogue: atomics into the kernel-layout gradient
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) {
        const int j = j0 + wj * (BJ / WJ) + jt * 32 + (lane & 31);
        if (j >= fa.K) continue;
#pragma unroll
        for (int it = 0; it < TM; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = i0 + wi * (BI / WI) + it * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (i < e.N) atomicAdd(g.dWk + (long)i * fa.Kp + j, acc[it][jt][q]);
            }
        }
    }
    if (do_bias) {
        // threads sharing an i-octet differ in their pixel row ml: reduce across ml through LDS
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
        if (tid < BI && i0 + tid < e.N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < BK; ++r) s += bred[r][tid];
            atomicAdd(g.dbias + i0 + tid, s);
        }
    }
}

template <typename T, int DMODE>
int launch_dgrad(const ConvDgradArgs& a, hipStream_t st) {
    const int gx = a.nblk;
    if (a.N <= 32) hipLaunchKernelGGL((k_conv_dgrad<T, DMODE, 32>), dim3(gx, 1), dim3(NT), 0, st, a);
    else if (a.N <= 64) hipLaunchKernelGGL((k_conv_dgrad<T, DMODE, 64>), dim3(gx, 1), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((k_conv_dgrad<T, DMODE, 128>), dim3(gx, cdiv(a.N, 128)), dim3(NT), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int dgrad_mode(const ConvDgradArgs& a, hipStream_t st) {
    switch (a.dmode) {
        case DG_1X1: return launch_dgrad<T, DG_1X1>(a, st);
        case DG_1X1_POOL: return launch_dgrad<T, DG_1X1_POOL>(a, st);
        case DG_3X3: return launch_dgrad<T, DG_3X3>(a, st);
    }
    return -1;
}

template <typename T, int AMODE>
int launch_wgrad(const ConvWgradArgs& a, hipStream_t st) {
    const int N = a.e.N, M = a.fa.M;
    const int jt = cdiv(a.fa.Kp, BJ);
    const int BIv = N <= 32 ? 32 : N <= 64 ? 64 : 128;
    const int it = cdiv(N, BIv);
    // enough splits to fill the chip, at least 512 pixels each, chunks of 32 rows
    int split = cdiv(1024, jt * it);
    const int max_split = cdiv(M, 512);
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    const int rows = (int)round_up(cdiv(M, split), BK);
    split = cdiv(M, rows);
    dim3 grid(jt, it, split);
    if (BIv == 32) hipLaunchKernelGGL((k_conv_wgrad<T, AMODE, 32>), grid, dim3(NT), 0, st, a, rows);
    else if (BIv == 64) hipLaunchKernelGGL((k_conv_wgrad<T, AMODE, 64>), grid, dim3(NT), 0, st, a, rows);
    else hipLaunchKernelGGL((k_conv_wgrad<T, AMODE, 128>), grid, dim3(NT), 0, st, a, rows);
    TCVN_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int wgrad_mode(const ConvWgradArgs& a, hipStream_t st) {
    switch (a.fa.amode) {
        case A_1X1: return launch_wgrad<T, A_1X1>(a, st);
        case A_1X1_POOL: return launch_wgrad<T, A_1X1_POOL>(a, st);
        case A_3X3: return launch_wgrad<T, A_3X3>(a, st);
        case A_STEM: return launch_wgrad<T, A_STEM>(a, st);
    }
    return -1;
}

}  // namespace

int conv_dgrad_nblk(const ConvDgradArgs& a) {
    if (conv3x3_dgrad_tile_ok(a)) return conv3x3_dgrad_tile_nblk(a);
    if (conv3x3_dgrad_f32_ok(a)) return conv3x3_dgrad_f32_nblk(a);
    if (conv1x1_dgrad_f32_ok(a)) return conv1x1_dgrad_f32_nblk(a);
    return conv_fwd_grid(a.M);
}

int conv_dgrad(const ConvDgradArgs& a, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (a.Kp % BK != 0) return -2;
    if (a.nblk != conv_dgrad_nblk(a)) { fprintf(stderr, "tcvn: conv_dgrad nblk mismatch\n"); return -3; }
    if (conv3x3_dgrad_tile_ok(a)) return conv3x3_dgrad_tile(a, st);
    if (conv3x3_dgrad_f32_ok(a)) return conv3x3_dgrad_f32(a, st);
    if (conv1x1_dgrad_f32_ok(a)) return conv1x1_dgrad_f32(a, st);
    char nm[96];
    snprintf(nm, sizeof(nm), "k_conv_dgrad<%s,%d,%d>", a.mode == MODE_F32 ? "float" : "bf16", a.dmode, a.N <= 32 ? 32 : a.N <= 64 ? 64 : 128);
    ProfScope ps(nm, 2.0 * a.M * (double)a.N * ((a.dmode == DG_3X3 ? 9 : 1) * a.e.N), 0.0, st);
    return a.mode == MODE_F32 ? dgrad_mode<float>(a, st) : dgrad_mode<bf16>(a, st);
}

int conv_wgrad(const ConvWgradArgs& a, hipStream_t st) {
    if (a.fa.M <= 0) return 0;
    if (a.nfast) {
        if (!conv3x3_wgrad_tile_ok(a)) { fprintf(stderr, "tcvn: conv_wgrad: tile kernel requested but not applicable\n"); return -4; }
        return conv3x3_wgrad_tile(a, st);
    }
    if (conv3x3_wgrad_f32_ok(a)) return conv3x3_wgrad_f32(a, st);
    if (gemm_tn_f32_ok(a)) return gemm_tn_f32(a, st);               // every fp32 1x1 weight gradient since round 5 (68 launches 6.8 ms against 9.0 ms
    if (conv1x1_wgrad_f32_ok(a)) return conv1x1_wgrad_f32(a, st);   // with the 128-output tile kernel on the bottleneck layers); that kernel serves callers without a slab
    char nm[96];
    snprintf(nm, sizeof(nm), "k_conv_wgrad<%s,%d,%d>", a.mode == MODE_F32 ? "float" : "bf16", a.fa.amode, a.e.N <= 32 ? 32 : a.e.N <= 64 ? 64 : 128);
    ProfScope ps(nm, 2.0 * a.fa.M * (double)a.e.N * a.fa.K, 0.0, st);
    return a.mode == MODE_F32 ? wgrad_mode<float>(a, st) : wgrad_mode<bf16>(a, st);
}

}  // namespace tcvn
