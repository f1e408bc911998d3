// bf16 "NT" GEMM for the 1x1 convolutions on materialised operands:
//     C[m][n] = sum_k A[m][k] * W[n][k]        A = activated input / output gradient [M][K] (bf16, row-major, LDS-DMA'd)
// with three fused epilogues (reference: Bottleneck.bottleneck_block.conv1 / Transition.conv, layers/dense_net.py:18-27,84-94):
//   EPI_FWD        + bias -> bf16 -> Out[m][n_off+n], per-channel (sum, sum^2) partials for the next BatchNorm
//   EPI_DGRAD      PReLU + BatchNorm backward against the norm's input x = Xin[m][n]: G[m][n] += sc*dU, 3 partial sums
//   EPI_DGRAD_POOL same, the row being a 2x2-pooled pixel: each of its four source pixels receives dA/4
// One persistent workgroup per CU walks 128-row tiles; each wave owns 32 of the tile's 128 output columns for all rows, so its
// weight fragments (<= 32 k-steps) live in registers for the whole launch.  The fp32 C tile is exchanged through LDS so that the
// epilogue touches HBM with 16 B per lane (x, G read / G write) instead of 2-byte accesses in MFMA layout.
#include <type_traits>
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int CLD = 132;                        // C tile leading dimension (floats), padded
// rows per tile: 128 with one workgroup per CU, or 64 with two (their epilogue arithmetic and memory phases then overlap)

constexpr int KS_FWD = 40, KS_FWD_SMALL = 16, KS_DGRAD = 16, KS_DGRAD_SMALL = 8, KS_POOL = 24, KS_POOL_SMALL = 8;   // k-steps of 16 held in registers per instance

template <int ROWS>
__device__ __forceinline__ void dma_a(char* smem_base, int buf_off, const bf16* __restrict__ A, long lda, int K, int k0, long m0,
                                      long M, const char* __restrict__ zeros, int wave, int lane) {
    const int rsub = lane >> 4, slot = lane & 15;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        const int rg = wave + 4 * i;
        const int r = rg * 4 + rsub;
        const int col = k0 + ((slot ^ (r & 15)) << 3);
        const long m = m0 + r;
        const char* src = (m < M && col < K) ? reinterpret_cast<const char*>(A + m * lda + col) : zeros + (slot << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem_base + buf_off + rg * 1024), 16, 0, 0);
    }
}

// XF = 1 (forward only): A is the RAW BatchNorm input; every landed A tile is transformed in LDS to prelu(sc*x + sh) before the
// MFMAs read it (tables in LDS, 4-8 16-B chunks per thread), so the activated copy of the concat buffer is never written to HBM
// XF = 2 (forward, eval mode): the epilogue applies the NEXT BatchNorm (running statistics: no batch reduction to wait for) and
// PReLU to the fp32 result before the one rounding to bf16 -- the raw 1x1 output and the pass that activated it disappear;
// no statistics in this instance
template <int EPI, int MAXKS, int ROWS, int XF = 0>
__global__ __launch_bounds__(256, ROWS == 64 ? 2 : 1) void k_gemm_nt_bf16(const GemmNtArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = ROWS * 256;
    float* Cs = reinterpret_cast<float*>(smem + 2 * TILE);                 // [ROWS][CLD]
    double* red = reinterpret_cast<double*>(smem + 2 * TILE);              // [4][128][3] aliases Cs after the last tile
    float* atab = reinterpret_cast<float*>(smem + 2 * TILE + ROWS * CLD * 4);   // XF: [3][Kt] scale, shift, slope of the A transform
    const int Kt = (g.K + 7) & ~7;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * 128;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(g.A);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    const int ksteps = g.Kp >> 4, nkc = (g.K + 127) >> 7;
    const long mtiles = (g.M + ROWS - 1) / ROWS;

    // this wave's weight fragments: row tile (n0/32 + wave), all k-steps
    const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + (((long)(blockIdx.y * 4 + wave) * ksteps) * 64 + lane) * 8;
    const bool wave_live = n0 + wave * 32 < g.N;
    bf16x8_t bw[MAXKS];
#pragma unroll
    for (int i = 0; i < MAXKS; ++i)
        if (i < ksteps && wave_live) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + (long)i * 512);
        else
#pragma unroll
            for (int j = 0; j < 8; ++j) bw[i][j] = (__bf16)0.f;

    // epilogue role: 16 threads per row (8 channels each), rows c_r0 + 16*i
    const int c8 = tid & 15, c_r0 = tid >> 4;
    const int ncol = n0 + c8 * 8;                                         // first output column of this thread's chunk
    const bool col_ok = ncol < g.N;                                        // the last chunk may be partial (N % 8 != 0)
    float cb[8], csc[8], csh[8], csl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool ok = ncol + j < g.N;
        cb[j] = (EPI == EPI_FWD && ok) ? g.bias[ncol + j] : 0.f;
        if (XF == 2) {
            csc[j] = ok ? g.osc[ncol + j] : 0.f; csh[j] = ok ? g.osh[ncol + j] : 0.f; csl[j] = ok ? g.osl[ncol + j] : 0.f;
        } else {
            csc[j] = (EPI != EPI_FWD && ok) ? g.sc[ncol + j] : 0.f;
            csh[j] = (EPI != EPI_FWD && ok) ? g.sh[ncol + j] : 0.f;
            csl[j] = (EPI != EPI_FWD && ok) ? g.sl[ncol + j] : 0.f;
        }
    }
    // per-thread running sums: a thread sees at most a few hundred rows per channel, so the two-blocks-per-CU variants keep
    // them in fp32 (registers); everything across threads and workgroups is reduced in fp64
    typedef typename std::conditional<ROWS == 64, float, double>::type stat_t;
    stat_t st1[8], st2[8], st3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[j] = 0; st2[j] = 0; st3[j] = 0; }

    if (XF == 1) {
        for (int i = threadIdx.x; i < Kt; i += 256) {
            const bool ok = i < g.Kreal;
            atab[i] = ok ? g.asc[i] : 0.f; atab[Kt + i] = ok ? g.ash[i] : 0.f; atab[2 * Kt + i] = ok ? g.asl[i] : 0.f;
        }
    }
    // in-LDS transform of the k-chunk kc held in buffer `buf`: slot s of row r holds source chunk s ^ (r & 15)
    auto xform = [&](int buf, int kc) {
#pragma unroll
        for (int it = 0; it < ROWS * 16 / 256; ++it) {
            const int idx = tid + 256 * it, row = idx >> 4, slot = idx & 15;
            const int col = kc * 128 + ((slot ^ (row & 15)) << 3);
            if (col < Kt) {
                u16x8* p = reinterpret_cast<u16x8*>(smem + buf + row * 256 + (slot << 4));
                const u16x8 v = *p;
                const float4 s0 = *reinterpret_cast<const float4*>(atab + col), s1 = *reinterpret_cast<const float4*>(atab + col + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(atab + Kt + col), h1 = *reinterpret_cast<const float4*>(atab + Kt + col + 4);
                const float4 l0 = *reinterpret_cast<const float4*>(atab + 2 * Kt + col), l1 = *reinterpret_cast<const float4*>(atab + 2 * Kt + col + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                const float sl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = col + j < g.Kreal ? f2bf(prelu(fmaf(bf2f(v[j]), sc[j], sh[j]), sl[j])) : (bf16)0;
                *p = o;
            }
        }
    };
    long mt = blockIdx.x;
    if (mt < mtiles) dma_a<ROWS>(smem, 0, A, g.lda, g.K, 0, mt * ROWS, g.M, zeros, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (; mt < mtiles; mt += gridDim.x) {
        f32x16 acc[ROWS / 32];
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        // EPI_DGRAD: the epilogue's x / G rows of this tile are requested now, all at once, so that they travel under the
        // MFMA phase and next to the A prefetch (a load issued inside the epilogue loop cannot pass the G store of the
        // previous row group -- same base pointer -- and every row group would pay a full HBM round trip)
        constexpr bool PREF = EPI == EPI_DGRAD && (ROWS == 128 || MAXKS <= 8);     // 32 registers at 64 rows: the 16-k-step variant has none to spare
        u16x8 pxv[PREF ? ROWS / 16 : 1], pgv[PREF ? ROWS / 16 : 1];
        if (PREF && col_ok) {
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) {
                const long m = mt * ROWS + c_r0 + 16 * i;
                if (m < g.M) {
                    pxv[i] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16*>(g.Xin) + m * g.ldxin + ncol);
                    pgv[i] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16*>(g.Gout) + m * g.ldgo + ncol);
                }
            }
        }
#pragma unroll
        for (int kc = 0; kc < MAXKS / 8; ++kc) {
            if (kc < nkc) {
                // prefetch the next A tile (next k-chunk of this row tile, or the first chunk of the next row tile)
                if (kc + 1 < nkc) dma_a<ROWS>(smem, (cur ^ 1) * TILE, A, g.lda, g.K, (kc + 1) * 128, mt * ROWS, g.M, zeros, wave, lane);
                else if (mt + gridDim.x < mtiles) dma_a<ROWS>(smem, (cur ^ 1) * TILE, A, g.lda, g.K, 0, (mt + gridDim.x) * ROWS, g.M, zeros, wave, lane);
                const int ab = cur * TILE;
                if (XF == 1) {                  // bare barrier: a __syncthreads() would drain the prefetch just issued
                    xform(ab, kc);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (kc * 8 + ks < ksteps) {
#pragma unroll
                        for (int i = 0; i < ROWS / 32; ++i) {
                            const int row = i * 32 + r;
                            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(smem + ab + row * 256 + (((2 * ks + h) ^ (row & 15)) << 4));
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[kc * 8 + ks], acc[i], 0, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur ^= 1;
            }
        }
        // C tile -> LDS (fp32): row = i*32 + (e&3) + 8*(e>>2) + 4*h, column = wave*32 + r
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) Cs[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wave * 32 + r] = acc[i][e];
        __syncthreads();
        if (col_ok) {
            // per-tile sums: the fp32 running sums themselves where those are fp32 (64-row variants: registers are scarce there)
            constexpr bool DIRECT = std::is_same<stat_t, float>::value;
            float t1[DIRECT ? 1 : 8], t2[DIRECT ? 1 : 8], t3[DIRECT ? 1 : 8];
            float* const f1 = DIRECT ? reinterpret_cast<float*>(st1) : t1;
            float* const f2 = DIRECT ? reinterpret_cast<float*>(st2) : t2;
            float* const f3 = DIRECT ? reinterpret_cast<float*>(st3) : t3;
            if (!DIRECT)
#pragma unroll
                for (int j = 0; j < 8; ++j) { f1[j] = 0.f; f2[j] = 0.f; f3[j] = 0.f; }
            if (EPI == EPI_DGRAD_POOL) {
                // a pooled row feeds four source pixels: x of all four, for POOL_CH row groups at a time, is requested before the
                // first use (16 loads in flight per thread).  One load -> use -> store per trip left 4 KB in flight per workgroup,
                // i.e. every trip paid a full HBM round trip: 0.5-0.8 TB/s on the wide transitions.  This launch is always the first
                // contribution to G (g_write, checked by the launcher): G is written, never read.
                constexpr int POOL_CH = ROWS == 64 ? 2 : 4;       // two workgroups per CU at 64 rows: same bytes in flight per CU
                const int hw = g.H * g.W;
#pragma unroll
                for (int i0 = 0; i0 < ROWS / 16; i0 += POOL_CH) {
                    u16x8 xv[POOL_CH][4];
                    long p00[POOL_CH];
#pragma unroll
                    for (int c = 0; c < POOL_CH; ++c) {
                        const long m = mt * ROWS + c_r0 + 16 * (i0 + c);
                        const long mm = m < g.M ? m : g.M - 1;                          // clamped: the loads stay unconditional
                        const long img = mm / hw;
                        const int rem = (int)(mm - img * hw);
                        const int ho = rem / g.W, wo = rem - ho * g.W;
                        p00[c] = (img * g.Hin + 2 * ho) * g.Win + 2 * wo;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            xv[c][t] = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16*>(g.Xin) + (p00[c] + (t >> 1) * g.Win + (t & 1)) * g.ldxin + ncol);
                    }
#pragma unroll
                    for (int c = 0; c < POOL_CH; ++c) {
                        const int rr = c_r0 + 16 * (i0 + c);
                        if (mt * ROWS + rr < g.M) {
                            const float4 ca = *reinterpret_cast<const float4*>(Cs + rr * CLD + c8 * 8);
                            const float4 cc = *reinterpret_cast<const float4*>(Cs + rr * CLD + c8 * 8 + 4);
                            const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cc.x, cc.y, cc.z, cc.w};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                bf16* gp = reinterpret_cast<bf16*>(g.Gout) + (p00[c] + (t >> 1) * g.Win + (t & 1)) * g.ldgo + ncol;
                                u16x8 o;
                                if (ncol + 8 > g.N) o = *reinterpret_cast<const u16x8*>(gp);    // partial last chunk: the rest is not ours
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const float x = bf2f(xv[c][t][j]);
                                    const float u = fmaf(x, csc[j], csh[j]);
                                    const float dA = 0.25f * cv[j];
                                    const float du = u > 0.f ? dA : csl[j] * dA;
                                    f1[j] += du; f2[j] += du * x; f3[j] += u > 0.f ? 0.f : dA * u;
                                    if (ncol + j < g.N) o[j] = f2bf(csc[j] * du);
                                }
                                *reinterpret_cast<u16x8*>(gp) = o;
                            }
                        }
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) {
                const int rr = c_r0 + 16 * i;
                const long m = mt * ROWS + rr;
                if (m < g.M) {
                    const float4 ca = *reinterpret_cast<const float4*>(Cs + rr * CLD + c8 * 8);
                    const float4 cc = *reinterpret_cast<const float4*>(Cs + rr * CLD + c8 * 8 + 4);
                    const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cc.x, cc.y, cc.z, cc.w};
                    if (EPI == EPI_FWD) {
                        u16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float v = cv[j] + cb[j];
                            if (XF == 2) v = prelu(fmaf(v, csc[j], csh[j]), csl[j]);
                            o[j] = ncol + j < g.N ? f2bf(v) : (bf16)0;                   // channels beyond N: zero (written later)
                            if (XF != 2) { const float x = bf2f(o[j]); f1[j] += x; f2[j] += x * x; }
                        }
                        *reinterpret_cast<u16x8*>(reinterpret_cast<bf16*>(g.Out) + m * g.ldo + g.n_off + ncol) = o;
                    } else {
                        bf16* gp = reinterpret_cast<bf16*>(g.Gout) + m * g.ldgo + ncol;
                        u16x8 xv, gv;
                        if (PREF) { xv = pxv[i]; gv = pgv[i]; }
                        else {
                            xv = *reinterpret_cast<const u16x8*>(reinterpret_cast<const bf16*>(g.Xin) + m * g.ldxin + ncol);
                            gv = *reinterpret_cast<const u16x8*>(gp);
                        }
                        u16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float x = bf2f(xv[j]);
                            const float u = fmaf(x, csc[j], csh[j]);
                            const float dA = cv[j];
                            const float du = u > 0.f ? dA : csl[j] * dA;
                            f1[j] += du; f2[j] = fmaf(du, x, f2[j]); f3[j] = fmaf(u > 0.f ? 0.f : dA, u, f3[j]);
                            o[j] = ncol + j < g.N ? f2bf(fmaf(csc[j], du, bf2f(gv[j]))) : gv[j];   // beyond N: not ours
                        }
                        *reinterpret_cast<u16x8*>(gp) = o;
                    }
                }
            }
            if (!DIRECT)
#pragma unroll
                for (int j = 0; j < 8; ++j) { st1[j] += (stat_t)f1[j]; st2[j] += (stat_t)f2[j]; st3[j] += (stat_t)f3[j]; }
        }
        __syncthreads();
    }
    if (XF == 2 || g.part == nullptr) return;
    // reduce over the 16 row groups: 4 per wave by shuffles (lanes differing in bits 4,5), then across waves through LDS
    constexpr int NS = EPI == EPI_FWD ? 2 : 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d1 = (double)st1[j], d2 = (double)st2[j], d3 = (double)st3[j];
        d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32);
        d2 += __shfl_xor(d2, 16); d2 += __shfl_xor(d2, 32);
        if (NS == 3) { d3 += __shfl_xor(d3, 16); d3 += __shfl_xor(d3, 32); }
        if (j == 0) __syncthreads();                       // red aliases the C tile
        if (lane < 16) {
            double* p = red + ((wave * 128) + c8 * 8 + j) * 3;
            p[0] = d1; p[1] = d2; p[2] = d3;
        }
    }
    __syncthreads();
    if (tid < 128 && n0 + tid < g.N) {
        double a = 0, b = 0, c = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 128 + tid) * 3]; b += red[(w * 128 + tid) * 3 + 1]; c += red[(w * 128 + tid) * 3 + 2]; }
        double* p = g.part + ((long)blockIdx.x * g.N + n0 + tid) * NS;
        p[0] = a; p[1] = b;
        if (NS == 3) p[2] = c;
    }
}

}  // namespace

static int nt_max_ksteps(int epi) { return epi == EPI_FWD ? KS_FWD : epi == EPI_DGRAD ? KS_DGRAD : KS_POOL; }
// rows per tile of the instance that serves `a`
static int nt_rows(const GemmNtArgs& a) {
    if (a.epi == EPI_DGRAD) return 64;
    if (a.epi == EPI_DGRAD_POOL) return a.Kp <= KS_POOL_SMALL * 16 ? 64 : 128;
    return a.Kp <= KS_FWD_SMALL * 16 ? 64 : 128;
}
bool gemm_nt_ok(const GemmNtArgs& a) {
    if (!a.A || !a.Wfrag || !a.zeros || (a.lda & 7) || (a.K & 7) || a.Kp > nt_max_ksteps(a.epi) * 16 || (a.Kp & 15)) return false;
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.Wfrag) & 15)) return false;
    if (a.epi == EPI_FWD) return (a.ldo & 7) == 0 && (a.n_off & 7) == 0 && (reinterpret_cast<uintptr_t>(a.Out) & 15) == 0;
    if (a.epi == EPI_DGRAD_POOL && !a.g_write) return false;              // the pooled epilogue writes G (first contribution), it never adds
    return (a.ldxin & 7) == 0 && (a.ldgo & 7) == 0 && (reinterpret_cast<uintptr_t>(a.Xin) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.Gout) & 15) == 0;
}
int gemm_nt_nblk(const GemmNtArgs& a) {
    const int rows = nt_rows(a), nn = cdiv(a.N, 128);
    // resident capacity: two workgroups per CU for the 64-row tiles, shared by the nn column tiles of the grid -- more
    // workgroups than that only queue up and pay their prologue (weight fragments, tables) again
    int cap = (rows == 64 ? 512 : 256) / nn;
    if (cap < 64) cap = 64;
    const long mt = (a.M + rows - 1) / rows;
    return (int)(mt < cap ? mt : cap);
}
namespace {
__global__ void k_zero_pool_remainder(bf16* G, long ld, int n_img, int Hin, int Win, int Hc, int Wc) {
    // one thread per (remainder pixel, 8-channel chunk)
    const int rows_extra = Hin - Hc, cols_extra = Win - Wc;
    const long per_img = (long)rows_extra * Win + (long)Hc * cols_extra;
    const long chunks = ld / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img * per_img * chunks) return;
    const long px = i / chunks; const int ch = (int)(i - px * chunks);
    const int img = (int)(px / per_img); long r = px - (long)img * per_img;
    int y, x;
    if (r < (long)rows_extra * Win) { y = Hc + (int)(r / Win); x = (int)(r % Win); }
    else { r -= (long)rows_extra * Win; y = (int)(r / cols_extra); x = Wc + (int)(r % cols_extra); }
    u16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = 0;
    *reinterpret_cast<u16x8*>(G + (((long)img * Hin + y) * Win + x) * ld + ch * 8) = z;
}
}  // namespace

int zero_pool_remainder(void* G, long ld, int n_img, int Hin, int Win, int Ho, int Wo, hipStream_t st) {
    const int Hc = 2 * Ho, Wc = 2 * Wo;
    const long per_img = (long)(Hin - Hc) * Win + (long)Hc * (Win - Wc);
    if (per_img <= 0 || n_img <= 0) return 0;
    if (ld % 8) return -2;
    const long total = n_img * per_img * (ld / 8);
    hipLaunchKernelGGL(k_zero_pool_remainder, dim3(cdiv(total, 256)), dim3(256), 0, st, reinterpret_cast<bf16*>(G), ld, n_img, Hin, Win, Hc, Wc);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int gemm_nt_bf16(const GemmNtArgs& a, const char* label, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (!gemm_nt_ok(a)) return -2;
    if (a.part != nullptr && a.nblk != gemm_nt_nblk(a)) { fprintf(stderr, "tcvn: gemm_nt nblk mismatch\n"); return -3; }
    const int rows = nt_rows(a);
    const size_t smem = 2 * rows * 256 + (size_t)rows * CLD * 4 + (a.epi == EPI_FWD && a.asc != nullptr ? 3 * ((a.K + 7) & ~7) * 4 : 0);
    static bool attr = false;
    if (!attr) {
        const void* fns[10] = {reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_FWD, KS_FWD, 128, 2>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_FWD, KS_FWD_SMALL, 64, 2>),reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_FWD, KS_FWD, 128>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_FWD, KS_FWD_SMALL, 64>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_DGRAD, KS_DGRAD, 64>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_DGRAD, KS_DGRAD_SMALL, 64>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_DGRAD_POOL, KS_POOL, 128>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_DGRAD_POOL, KS_POOL_SMALL, 64>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_FWD, KS_FWD, 128, 1>),
                              reinterpret_cast<const void*>(k_gemm_nt_bf16<EPI_FWD, KS_FWD_SMALL, 64, 1>)};
        for (const void* f : fns) TCVN_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // algorithmic HBM bytes, SURVEY 8(d) strict (operands once, results once): A once; FWD writes N channels; the data-gradient
    // epilogues read x and write G (4 pixels per row when pooled).  The kernel also READS G (it accumulates into it): that
    // read is traffic, not algorithm -- with it the count would be K + 3 px N.
    const double px = a.epi == EPI_DGRAD_POOL ? 4.0 : 1.0;
    const double bytes = (double)a.M * 2.0 * (a.K + (a.epi == EPI_FWD ? (double)a.N : 2.0 * px * a.N));
    ProfScope ps(label, 2.0 * a.M * (double)a.N * a.K, bytes, st);
    const dim3 grid(gemm_nt_nblk(a), cdiv(a.N, 128));
    const bool xf = a.epi == EPI_FWD && a.asc != nullptr;
    const bool oact = a.epi == EPI_FWD && a.osc != nullptr;
    if (oact && (xf || a.part != nullptr || !a.osh || !a.osl)) return -2;      // eval-mode epilogue: no statistics, materialised A
    if (oact && a.Kp <= KS_FWD_SMALL * 16) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_FWD, KS_FWD_SMALL, 64, 2>), grid, dim3(256), smem, st, a);
    else if (oact) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_FWD, KS_FWD, 128, 2>), grid, dim3(256), smem, st, a);
    else if (xf && a.Kp <= KS_FWD_SMALL * 16) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_FWD, KS_FWD_SMALL, 64, 1>), grid, dim3(256), smem, st, a);
    else if (xf) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_FWD, KS_FWD, 128, 1>), grid, dim3(256), smem, st, a);
    else if (a.epi == EPI_FWD && a.Kp <= KS_FWD_SMALL * 16) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_FWD, KS_FWD_SMALL, 64>), grid, dim3(256), smem, st, a);
    else if (a.epi == EPI_FWD) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_FWD, KS_FWD, 128>), grid, dim3(256), smem, st, a);
    else if (a.epi == EPI_DGRAD && a.Kp <= KS_DGRAD_SMALL * 16) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_DGRAD, KS_DGRAD_SMALL, 64>), grid, dim3(256), smem, st, a);
    else if (a.epi == EPI_DGRAD) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_DGRAD, KS_DGRAD, 64>), grid, dim3(256), smem, st, a);
    else if (a.Kp <= KS_POOL_SMALL * 16) hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_DGRAD_POOL, KS_POOL_SMALL, 64>), grid, dim3(256), smem, st, a);
    else hipLaunchKernelGGL((k_gemm_nt_bf16<EPI_DGRAD_POOL, KS_POOL, 128>), grid, dim3(256), smem, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
