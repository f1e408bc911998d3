// fp32 (parity mode) backward of the 1x1 bottleneck convolution conv1 (Cin -> 128 channels; reference:
// transformercvn/network/layers/dense_net.py:18-27 and its autograd): data gradient and weight gradient.
//
// Arithmetic is v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).  One operand float per lane and k-step, so an operand that a
// wave owns alone is simply kept in registers, read from HBM in the layout it already has, and only the operand all four
// waves share goes through LDS:
//   data gradient    dA[pos][c] = sum_n eff[pos][n] * Wt[c][n]      a wave owns 32 positions: its 32 x 128 eff block is 64
//                    registers per lane (lane = position, k = n, read along the row it sits in); the weights stream through
//                    LDS in 32-channel tiles, double buffered; PReLU + BatchNorm backward of norm1 in the epilogue, the three
//                    per-channel sums go through LDS double atomics into one partial row per workgroup;
//   weight gradient  dW[n][c] = sum_pos eff[pos][n] * xa[pos][c]     a wave owns 32 output channels n for up to 256 input
//                    channels (8 accumulator tiles); eff[pos][n] is one coalesced 128-B load per half wave and k-step,
//                    xa = prelu(bn1(x)) tiles of 64 positions are staged in LDS, double buffered, loads of tile t+1 in
//                    flight under the MFMAs of tile t.
// At 4 B per element these layers sit near the ridge of the fp32 roofline (32 flop/B at Cin = 256): the kernels are laid out
// so that every input element is read from HBM once per launch.
#include "prof.h"
#include "tcvn_ops.h"

namespace tcvn {

namespace {

constexpr int WS = 132;                 // weight tile row stride (floats): b128 fragment reads of 16 lanes hit 16 bank groups
constexpr int MAXC = 512;

// ---------------------------------------------------------------------------------------------------------------------
// data gradient
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_conv1x1_dgrad_f32(const ConvDgradArgs g, int ntiles, int nct) {
    __shared__ __attribute__((aligned(16))) float ws[2][32 * WS];
    __shared__ __attribute__((aligned(16))) float pq[256];
    __shared__ double stat[MAXC * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const EffSrc& e = g.e;
    const float* __restrict__ G = reinterpret_cast<const float*>(e.G);
    const float* __restrict__ X = reinterpret_cast<const float*>(e.X);
    const float* __restrict__ Wt = reinterpret_cast<const float*>(g.Wt);
    const float* __restrict__ Xin = reinterpret_cast<const float*>(g.Xin);
    float* __restrict__ Gout = reinterpret_cast<float*>(g.Gout);
    pq[tid] = tid < 128 ? e.P[tid] : e.Q[tid - 128];
    for (int i = tid; i < g.N * 3; i += 256) stat[i] = 0.0;
    // weight tile loader: 32 rows x 128 floats = 1024 float4, four per thread
    const int wr = tid >> 5, wc = (tid & 31) * 4;
    f32x4 wreg[4];
    auto w_issue = [&](int ct) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = ct * 32 + wr + 8 * i;
            wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < g.N) wreg[i] = *reinterpret_cast<const f32x4*>(Wt + (long)c * g.Kp + wc);
        }
    };
    auto w_commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&ws[buf][(wr + 8 * i) * WS + wc]) = wreg[i];
    };
    w_issue(0);
    w_commit(0);
    __syncthreads();
    int buf = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = t * 128 + wave * 32;
        const long pos = m0 + l31;
        const bool valid = pos < g.M;
        float a[64];                                  // eff[pos][lh*64 + kk]
        {
            const float* gp = G + pos * e.ldg + e.c_off + lh * 64;
            const float* xp = X + pos * e.ldx + e.c_off + lh * 64;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                f32x4 gq[4], xq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; xq[j] = gq[j];
                    if (valid) {
                        gq[j] = *reinterpret_cast<const f32x4*>(gp + h * 16 + 4 * j);
                        xq[j] = *reinterpret_cast<const f32x4*>(xp + h * 16 + 4 * j);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 P = *reinterpret_cast<const f32x4*>(&pq[lh * 64 + h * 16 + 4 * j]);
                    const f32x4 Q = *reinterpret_cast<const f32x4*>(&pq[128 + lh * 64 + h * 16 + 4 * j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = gq[j][q] + P[q] * xq[j][q] + Q[q];
                        if (e.drop_p > 0.f) v *= drop_scale_mn(e.drop_p, e.seed, e.stream_id, pos, lh * 64 + h * 16 + 4 * j + q, e.N);
                        a[h * 16 + 4 * j + q] = valid ? v : 0.f;
                    }
                }
            }
        }
        for (int ct = 0; ct < nct; ++ct) {
            // next weight tile (wraps to tile 0 of the next position tile) in flight under this tile's MFMAs
            w_issue(ct + 1 < nct ? ct + 1 : 0);
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            const float* wp = &ws[buf][l31 * WS + lh * 64];
            f32x4 bc[4], bn[4];                                           // B fragments, one group of 16 k-steps ahead
#pragma unroll
            for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const f32x4*>(wp + 4 * j);
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                if (gi < 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const f32x4*>(wp + 16 * (gi + 1) + 4 * j);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[16 * gi + 4 * j + q], bc[j][q], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) bc[j] = bn[j];
            }
            // epilogue: u = sc*x + sh ; dU = dA * prelu'(u) ; Gout (+)= sc*dU ; sums (dU, dU*x, dA*min(u,0))
            const int c = ct * 32 + l31;
            if (c < g.N) {
                const float sc = g.sc[c], sh = g.sh[c], sl = g.sl[c];
                double s1 = 0, s2 = 0, s3 = 0;
                const float* xb = Xin + (long)(m0 + 4 * lh) * g.ldxin + c;
                float* gb = Gout + (long)(m0 + 4 * lh) * g.ldgo + c;
                const int mrem = g.M - (m0 + 4 * lh);             // rows r of this lane are valid while r < mrem
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    float xv[8], gv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = (q & 3) + 8 * (q >> 2) + 16 * hq;
                        xv[q] = 0.f; gv[q] = 0.f;
                        if (r < mrem) {
                            xv[q] = xb[r * (int)g.ldxin];
                            if (g.accumulate) gv[q] = gb[r * (int)g.ldgo];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = (q & 3) + 8 * (q >> 2) + 16 * hq;
                        if (r < mrem) {
                            const float x = xv[q], u = fmaf(x, sc, sh), dA = acc[q + 8 * hq];
                            const float du = u > 0.f ? dA : sl * dA;
                            s1 += du; s2 += (double)du * x; s3 += u > 0.f ? 0.f : dA * u;
                            gb[r * (int)g.ldgo] = gv[q] + sc * du;
                        }
                    }
                }
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
                if (lh == 0) { atomicAdd(&stat[c * 3], s1); atomicAdd(&stat[c * 3 + 1], s2); atomicAdd(&stat[c * 3 + 2], s3); }
            }
            w_commit(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    __syncthreads();
    for (int i = tid; i < g.N * 3; i += 256) g.part[(long)blockIdx.x * g.N * 3 + i] = stat[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WP = 64;                  // positions per staged tile

template <int NCT>
__global__ __launch_bounds__(256, 1) void k_conv1x1_wgrad_f32(const ConvWgradArgs g, int ntiles) {
    constexpr int CW = NCT * 32;                                  // channels of this workgroup's chunk
    constexpr int CWS = CW + ((CW % 64 == 0) ? 32 : 0);           // row stride: the two half waves read rows 2ks, 2ks+1 -> disjoint banks
    constexpr int NV = WP * CW / 4 / 256;                         // float4 per thread and tile = 2 * NCT
    extern __shared__ __attribute__((aligned(16))) float smw[];
    float* xs = smw;                                              // [2][WP][CWS]
    float* tab = xs + 2 * WP * CWS;                               // [3][CW]: scale, shift, slope of this chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const ConvFwdArgs& fa = g.fa;
    const EffSrc& e = g.e;
    const float* __restrict__ A = reinterpret_cast<const float*>(fa.A);
    const float* __restrict__ G = reinterpret_cast<const float*>(e.G);
    const float* __restrict__ X = reinterpret_cast<const float*>(e.X);
    const int j0 = blockIdx.y * CW;                               // first input channel of the chunk
    const int n = wave * 32 + l31;                                // this lane's output channel
    const float P = e.P[n], Q = e.Q[n];
    for (int i = tid; i < CW; i += 256) {
        const bool ok = j0 + i < fa.K;
        tab[i] = ok ? fa.sc[j0 + i] : 0.f; tab[CW + i] = ok ? fa.sh[j0 + i] : 0.f; tab[2 * CW + i] = ok ? fa.sl[j0 + i] : 0.f;
    }
    f32x16 acc[NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    float bsum = 0.f;

    f32x4 xr[NV];
    float gr[32], xe[32], a[32];
    auto issue = [&](int t) {
        const long m0 = (long)t * WP;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + p * 256, r = idx / (CW / 4), c4 = idx - r * (CW / 4);
            const long m = m0 + r;
            xr[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < fa.M && j0 + c4 * 4 < fa.K) xr[p] = *reinterpret_cast<const f32x4*>(A + m * fa.lda + j0 + c4 * 4);
        }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const long m = m0 + 2 * ks + lh;
            gr[ks] = 0.f; xe[ks] = 0.f;
            if (m < fa.M) { gr[ks] = G[m * e.ldg + e.c_off + n]; xe[ks] = X[m * e.ldx + e.c_off + n]; }
        }
    };
    auto commit = [&](int t, int buf) {
        const long m0 = (long)t * WP;
        float* d = xs + buf * WP * CWS;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + p * 256, r = idx / (CW / 4), c4 = idx - r * (CW / 4);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + c4 * 4), sh = *reinterpret_cast<const f32x4*>(tab + CW + c4 * 4),
                        sl = *reinterpret_cast<const f32x4*>(tab + 2 * CW + c4 * 4);
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m0 + r < fa.M) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = prelu(fmaf(xr[p][q], sc[q], sh[q]), sl[q]);
            }
            *reinterpret_cast<f32x4*>(d + r * CWS + c4 * 4) = v;
        }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const long m = m0 + 2 * ks + lh;
            float v = 0.f;
            if (m < fa.M) {
                v = gr[ks] + P * xe[ks] + Q;
                if (e.drop_p > 0.f) v *= drop_scale_mn(e.drop_p, e.seed, e.stream_id, m, n, e.N);
            }
            a[ks] = v;
            bsum += v;
        }
    };
    __syncthreads();                                              // tab
    int t = blockIdx.x, buf = 0;
    if (t < ntiles) { issue(t); commit(t, 0); }
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < ntiles) issue(tn);
        const float* bp = xs + buf * WP * CWS + lh * CWS + l31;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks)
#pragma unroll
            for (int j = 0; j < NCT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], bp[2 * ks * CWS + j * 32], acc[j], 0, 0, 0);
        if (tn < ntiles) commit(tn, buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // dWk[n][j0 + c] += acc ; accumulator row = output channel within the wave's 32, column = input channel within the tile
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
        const int c = j0 + j * 32 + l31;
        if (c < fa.K) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int nn = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                atomicAdd(g.dWk + (long)nn * fa.Kp + c, acc[j][q]);
            }
        }
    }
    if (g.dbias != nullptr && blockIdx.y == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (lh == 0) atomicAdd(g.dbias + n, bsum);
    }
}

template <int NCT>
size_t wgrad1_smem() {
    constexpr int CW = NCT * 32, CWS = CW + ((CW % 64 == 0) ? 32 : 0);
    return (size_t)(2 * WP * CWS + 3 * CW) * 4;
}

template <int NCT>
int launch_wgrad1(const ConvWgradArgs& a, int ntiles, int nchunk, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv1x1_wgrad_f32<NCT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)wgrad1_smem<NCT>()));
        attr = true;
    }
    int gx = 256 / nchunk;
    if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(k_conv1x1_wgrad_f32<NCT>, dim3(gx, nchunk), dim3(256), wgrad1_smem<NCT>(), st, a, ntiles);
    TCVN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: Out[pos][n_off + n] = bias[n] + sum_c prelu(sc[c]*X[pos][c] + sh[c], sl[c]) * Wk[n][c],  n < 128, + BN statistics of Out
// A wave owns 32 positions: the activated input of one 64-channel chunk is 32 registers per lane (lane = position, read along the
// row it sits in, transformed in registers); the 128 x 64 weight chunk goes through LDS, double buffered, 68-float pitch (the b128
// fragment reads of 16 lanes cover all 64 banks once); chunk k+1's weights and input are in flight under chunk k's 128 MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WSF = 68;

__global__ __launch_bounds__(256, 2) void k_conv1x1_fwd_f32(const ConvFwdArgs g, int ntiles, int nch) {
    extern __shared__ __attribute__((aligned(16))) float ws_dyn[];             // [2][128 * WSF]
    float (*ws)[128 * WSF] = reinterpret_cast<float (*)[128 * WSF]>(ws_dyn);
    __shared__ __attribute__((aligned(16))) float tab[3 * MAXC];
    __shared__ double stat[128 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* __restrict__ X = reinterpret_cast<const float*>(g.A);
    const float* __restrict__ Wk = reinterpret_cast<const float*>(g.Wk);
    float* __restrict__ Out = reinterpret_cast<float*>(g.Out);
    for (int c = tid; c < MAXC; c += 256) {
        const bool ok = c < g.K;
        tab[c] = ok ? g.sc[c] : 0.f; tab[MAXC + c] = ok ? g.sh[c] : 0.f; tab[2 * MAXC + c] = ok ? g.sl[c] : 0.f;
    }
    stat[tid] = 0.0;
    // weight chunk loader: 128 rows x 64 floats = 2048 float4, eight per thread
    const int wr = tid >> 4, wc = (tid & 15) * 4;
    f32x4 wreg[8], xreg[8];
    auto issue = [&](int t, int kc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = wr + 16 * i, k = kc * 64 + wc;
            wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < g.K) wreg[i] = *reinterpret_cast<const f32x4*>(Wk + (long)n * g.Kp + k);
        }
        const long pos = (long)t * 128 + wave * 32 + l31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kc * 64 + lh * 32 + 4 * i;
            xreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pos < g.M && k < g.K) xreg[i] = *reinterpret_cast<const f32x4*>(X + pos * g.lda + k);
        }
    };
    float a[32];
    auto commit = [&](int t, int kc, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&ws[buf][(wr + 16 * i) * WSF + wc]) = wreg[i];
        const bool valid = (long)t * 128 + wave * 32 + l31 < g.M;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kc * 64 + lh * 32 + 4 * i;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + k), sh = *reinterpret_cast<const f32x4*>(tab + MAXC + k),
                        sl = *reinterpret_cast<const f32x4*>(tab + 2 * MAXC + k);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[4 * i + q] = (valid && k + q < g.K) ? prelu(fmaf(xreg[i][q], sc[q], sh[q]), sl[q]) : 0.f;
        }
    };
    float bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = g.bias ? g.bias[nt * 32 + l31] : 0.f;
    double s1[4], s2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { s1[nt] = 0; s2[nt] = 0; }
    __syncthreads();                                              // tab, stat
    int t = blockIdx.x, kc = 0, buf = 0;
    if (t < ntiles) {
        issue(t, 0);
        commit(t, 0, 0);
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    while (t < ntiles) {
        int nkc = kc + 1, nt_ = t;
        if (nkc == nch) { nkc = 0; nt_ = t + gridDim.x; }
        const bool more = nt_ < ntiles;
        if (more) issue(nt_, nkc);
        const float* wp = &ws[buf][l31 * WSF + lh * 32];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wp + nt * 32 * WSF + 4 * j);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * j + q], b[q], acc[nt], 0, 0, 0);
            }
        }
        if (kc == nch - 1) {                                       // epilogue: bias, store, statistics
            const long m0 = (long)t * 128 + wave * 32 + 4 * lh;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = nt * 32 + l31;
                float f1 = 0.f, f2 = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long m = m0 + (e & 3) + 8 * (e >> 2);
                    if (m < g.M) {
                        const float o = acc[nt][e] + bias[nt];
                        Out[m * g.ldo + g.n_off + n] = o;
                        f1 += o; f2 += o * o;
                    }
                    acc[nt][e] = 0.f;
                }
                s1[nt] += f1; s2[nt] += f2;
            }
        }
        if (more) commit(nt_, nkc, buf ^ 1);
        __syncthreads();
        if (!more) break;
        t = nt_; kc = nkc; buf ^= 1;
    }
    if (g.part != nullptr) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            double x = s1[nt], y = s2[nt];
            x += __shfl_xor(x, 32); y += __shfl_xor(y, 32);
            if (lh == 0) { atomicAdd(&stat[(nt * 32 + l31) * 2], x); atomicAdd(&stat[(nt * 32 + l31) * 2 + 1], y); }
        }
        __syncthreads();
        g.part[(long)blockIdx.x * 256 + tid] = stat[tid];
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool conv1x1_fwd_f32_ok(const ConvFwdArgs& a) {
    if (!conv3x3_tile_enabled() || a.mode != MODE_F32 || a.amode != A_1X1 || a.N != 128 || a.K > MAXC || a.K < 1 || (a.K & 3) || a.drop_p > 0.f || a.sc == nullptr) return false;
    return aligned16(a.A) && aligned16(a.Wk) && (a.lda & 3) == 0 && (a.Kp & 3) == 0;
}
int conv1x1_fwd_f32_nblk(const ConvFwdArgs& a) {
    const int ntiles = (a.M + 127) / 128;
    return ntiles < 512 ? ntiles : 512;
}
int conv1x1_fwd_f32(const ConvFwdArgs& a, hipStream_t st) {
    const int ntiles = (a.M + 127) / 128;
    ProfScope ps("k_conv1x1_fwd_f32", 2.0 * a.M * (double)a.N * a.K, (double)a.M * 4.0 * (a.K + a.N), st);
    constexpr size_t smem = 2 * 128 * WSF * 4;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv1x1_fwd_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = true;
    }
    hipLaunchKernelGGL(k_conv1x1_fwd_f32, dim3(conv1x1_fwd_f32_nblk(a)), dim3(256), smem, st, a, ntiles, (a.K + 63) / 64);
    TCVN_LAUNCH_CHECK();
    return 0;
}

bool conv1x1_dgrad_f32_ok(const ConvDgradArgs& a) {
    const EffSrc& e = a.e;
    if (!conv3x3_tile_enabled() || a.mode != MODE_F32 || a.dmode != DG_1X1 || e.N != 128 || a.Kp != 128 || a.N > MAXC || a.N < 1) return false;
    return aligned16(e.G) && aligned16(e.X) && aligned16(a.Wt) && (e.ldg & 3) == 0 && (e.ldx & 3) == 0 && (e.c_off & 3) == 0;
}
int conv1x1_dgrad_f32_nblk(const ConvDgradArgs& a) {
    const int ntiles = (a.M + 127) / 128;
    return ntiles < 512 ? ntiles : 512;
}
int conv1x1_dgrad_f32(const ConvDgradArgs& a, hipStream_t st) {
    const int ntiles = (a.M + 127) / 128;
    ProfScope ps("k_conv1x1_dgrad_f32", 2.0 * a.M * (double)a.N * a.e.N, (double)a.M * 4.0 * (2 * a.e.N + (2 + (a.accumulate ? 1 : 0)) * a.N), st);
    hipLaunchKernelGGL(k_conv1x1_dgrad_f32, dim3(conv1x1_dgrad_f32_nblk(a)), dim3(256), 0, st, a, ntiles, (a.N + 31) / 32);
    TCVN_LAUNCH_CHECK();
    return 0;
}

bool conv1x1_wgrad_f32_ok(const ConvWgradArgs& a) {
    const ConvFwdArgs& f = a.fa;
    if (!conv3x3_tile_enabled() || a.mode != MODE_F32 || f.amode != A_1X1 || a.e.N != 128 || a.nfast || f.K > MAXC || f.K < 1 || (f.K & 3) || f.sc == nullptr) return false;
    return aligned16(f.A) && (f.lda & 3) == 0;
}
int conv1x1_wgrad_f32(const ConvWgradArgs& a, hipStream_t st) {
    const ConvFwdArgs& f = a.fa;
    const int ntiles = (f.M + WP - 1) / WP;
    const int nct = (f.K + 31) / 32, nchunk = (nct + 7) / 8, per = (nct + nchunk - 1) / nchunk;
    ProfScope ps("k_conv1x1_wgrad_f32", 2.0 * f.M * (double)a.e.N * f.K, (double)f.M * 4.0 * (f.K + 2 * a.e.N), st);
    switch (per) {
        case 1: return launch_wgrad1<1>(a, ntiles, nchunk, st);
        case 2: return launch_wgrad1<2>(a, ntiles, nchunk, st);
        case 3: return launch_wgrad1<3>(a, ntiles, nchunk, st);
        case 4: return launch_wgrad1<4>(a, ntiles, nchunk, st);
        case 5: return launch_wgrad1<5>(a, ntiles, nchunk, st);
        case 6: return launch_wgrad1<6>(a, ntiles, nchunk, st);
        case 7: return launch_wgrad1<7>(a, ntiles, nchunk, st);
        default: return launch_wgrad1<8>(a, ntiles, nchunk, st);
    }
}

}  // namespace tcvn
