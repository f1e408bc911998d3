// HBM-bound helper kernels of the DenseNet forward path: pixel scatter, BatchNorm statistics plumbing,
// the fused BN+PReLU+AvgPool stem tail, the global-average head and the weight re-layout.
#include "tcvn_ops.h"

namespace tcvn {

namespace {

// ---- BatchNorm link: finalize producer statistics, build the consumer table, update running stats -------------
// Reference semantics: torch.nn.BatchNorm2d in training mode (biased batch variance for normalisation, unbiased for
// running_var, momentum 0.1), as instantiated at layers/dense_net.py:19,30,85,119,147.
__global__ __launch_bounds__(256) void k_bn_link(const BnLinkArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wave;
    if (c >= a.C) return;
    // the channel's parameters are requested FIRST, with the partial rows: read after the reduction they were a second, dependent round trip
    // in a kernel that is nothing but latency (132 launches per step on each embedder's chain)
    const float gamma = a.gamma[c], beta = a.beta[c];
    const bool upd = a.train && a.running_mean != nullptr;
    const float rmean = upd || !a.train ? a.running_mean[c] : 0.f, rvar = upd || !a.train ? a.running_var[c] : 0.f;
    double mean, var;
    if (a.train) {
        if (c >= a.c_new0 && c < a.c_new0 + a.n_new && a.isum != nullptr) {
            // the producer added its sums to fixed-point accumulators (bn_lf.h): nothing to reduce
            long long sa, sb;
            lf_sums(a.isum, a.isum_stride, c - a.c_new0, sa, sb);
            mean = (double)sa * ((1.0 / (double)a.count) / LF_S1);
            var = (double)sb * ((1.0 / (double)a.count) / LF_S2) - mean * mean;
            if (var < 0) var = 0;
            if (lane == 0) { a.bstat[c * 2] = mean; a.bstat[c * 2 + 1] = var; }
        } else if (c >= a.c_new0 && c < a.c_new0 + a.n_new) {
            double s1 = 0, s2 = 0;
            // eight partial rows per trip: every row of a launch with <= 512 workgroups in one round trip (see k_bn_bwd_link)
            for (int b = lane; b < a.nblk; b += 512) {
                double v[8][2];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int bb = b + 64 * i;
                    const double* p = a.part + ((long)(bb < a.nblk ? bb : b) * a.part_ld + (c - a.c_new0)) * 2;
                    v[i][0] = p[0]; v[i][1] = p[1];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (b + 64 * i < a.nblk) { s1 += v[i][0]; s2 += v[i][1]; }
            }
            s1 = wave_sum(s1); s2 = wave_sum(s2);
            mean = s1 / (double)a.count;
            var = s2 / (double)a.count - mean * mean;
            if (var < 0) var = 0;
            if (lane == 0) { a.bstat[c * 2] = mean; a.bstat[c * 2 + 1] = var; }
        } else {
            mean = a.bstat[c * 2]; var = a.bstat[c * 2 + 1];
        }
    } else {
        mean = rmean; var = rvar;
    }
    if (lane == 0) {
        const float r = (float)(1.0 / sqrt(var + (double)a.eps));
        const float sc = gamma * r;
        a.sc[c] = sc;
        a.sh[c] = beta - (float)mean * sc;
        if (upd) {
            const double unb = a.count > 1 ? var * (double)a.count / (double)(a.count - 1) : var;
            a.running_mean[c] = (1.f - a.momentum) * rmean + a.momentum * (float)mean;
            a.running_var[c] = (1.f - a.momentum) * rvar + a.momentum * (float)unb;
        }
    }
}

__global__ void k_bn_eval_tables(const BnEvalDesc* d, float eps) {
    const BnEvalDesc e = d[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
        const float r = 1.0f / sqrtf(e.rv[c] + eps);
        const float sc = e.gamma[c] * r;
        e.sc[c] = sc;
        e.sh[c] = e.beta[c] - e.rm[c] * sc;
    }
}

// ---- COO pixel list -> dense NHWC map (reference: trainers/neutrino_full_dense_trainer.py:15-24, :46-67) ---------
// The map must be zero-filled beforehand.  Duplicate coordinates are last-writer-wins like the reference's
// non-accumulating indexed `+=`.
template <typename T>
__global__ void k_scatter(const ScatterArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nnz) return;
    const int img = a.coords[i * 3], y = a.coords[i * 3 + 1], x = a.coords[i * 3 + 2];
    if (img < 0 || img >= a.n_img || y < 0 || y >= a.H || x < 0 || x >= a.W) return;
    T* o = reinterpret_cast<T*>(a.img) + (((long)img * a.H + y) * a.W + x) * a.Cpix;
    if (a.log_pixels == 3) {       // one_hot_pixels (reference :47-52): F.one_hot(values.long(), 256) per value channel, no scaling, no noise
        const int F = a.Cpix / 256;
        for (int f = 0; f < F; ++f) {
            int v = (int)a.values[i * F + f];
            v = v < 0 ? 0 : v > 255 ? 255 : v;
            o[f * 256 + v] = from_f<T>(1.f);
        }
        return;
    }
    for (int c = 0; c < a.Cpix; ++c) {
        float v = a.values[i * a.Cpix + c];
        v = a.log_pixels == 1 ? logf(v + 1.f) : a.log_pixels == 0 ? v / 255.0f : v;   // 2: already preprocessed
        if (a.noise_std != 0.f) {             // v * (1 + N(0,1) * std): Box-Muller on two counter-based uniforms
            const float u1 = fmaxf(rng_uniform(a.seed, 0x6e6f6973u, (uint64_t)(i * a.Cpix + c) * 2), 1e-7f);
            const float u2 = rng_uniform(a.seed, 0x6e6f6973u, (uint64_t)(i * a.Cpix + c) * 2 + 1);
            v *= 1.f + a.noise_std * sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
        }
        o[c] = from_f<T>(v);
    }
}

// ---- stem tail: BN + PReLU + AvgPool(3, stride 2) (layers/dense_net.py:119-121) ----------------------------------
constexpr int POOL_CJ = 4;   // up to 256 channels
template <typename T>
__global__ __launch_bounds__(256) void k_pool0(const Pool0Args a) {
    __shared__ double red[4][64 * POOL_CJ][2];
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const T* X = reinterpret_cast<const T*>(a.X);
    T* O = reinterpret_cast<T*>(a.Out);
    double s1[POOL_CJ], s2[POOL_CJ];
    float sc[POOL_CJ], sh[POOL_CJ], sl[POOL_CJ];
#pragma unroll
    for (int j = 0; j < POOL_CJ; ++j) {
        s1[j] = 0; s2[j] = 0;
        const int c = cl + 64 * j;
        sc[j] = c < a.C ? a.sc[c] : 0.f; sh[j] = c < a.C ? a.sh[c] : 0.f; sl[j] = c < a.C ? a.sl[c] : 0.f;
    }
    const long npix = (long)a.n_img * a.Ho * a.Wo;
    for (long p = (long)blockIdx.x * 4 + pg; p < npix; p += (long)gridDim.x * 4) {
        const int wo = (int)(p % a.Wo);
        const int ho = (int)((p / a.Wo) % a.Ho);
        const long img = p / ((long)a.Wo * a.Ho);
        const T* base = X + ((img * a.Hin + 2 * ho) * a.Win + 2 * wo) * a.C;
#pragma unroll
        for (int j = 0; j < POOL_CJ; ++j) {
            const int c = cl + 64 * j;
            if (c < a.C) {
                float acc = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc += prelu(fmaf(to_f<T>(base[((long)dy * a.Win + dx) * a.C + c]), sc[j], sh[j]), sl[j]);
                const T o = from_f<T>(acc * (1.0f / 9.0f));
                O[p * a.ldo + c] = o;
                const double v = (double)to_f<T>(o);
                s1[j] += v; s2[j] += v * v;
            }
        }
    }
    if (a.part == nullptr) return;
#pragma unroll
    for (int j = 0; j < POOL_CJ; ++j) { red[pg][cl + 64 * j][0] = s1[j]; red[pg][cl + 64 * j][1] = s2[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        double x = 0, y = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) { x += red[g][c][0]; y += red[g][c][1]; }
        a.part[((long)blockIdx.x * a.C + c) * 2] = x;
        a.part[((long)blockIdx.x * a.C + c) * 2 + 1] = y;
    }
}

// ---- head: final BN + PReLU + global average (layers/dense_net.py:147-154) ----------------------------------------
template <typename T>
__global__ void k_head_pool(const HeadPoolArgs a) {
    const int img = blockIdx.x;
    const T* X = reinterpret_cast<const T*>(a.X) + (long)img * a.HW * a.ldx;
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        const float sc = a.sc[c], sh = a.sh[c], sl = a.sl[c];
        float acc = 0.f;
        for (int p = 0; p < a.HW; ++p) acc += prelu(fmaf(to_f<T>(X[(long)p * a.ldx + c]), sc, sh), sl);
        a.F[(long)img * a.C + c] = acc / (float)a.HW;
    }
}

// ---- materialised activation: Out = bf16(prelu(X*sc + sh, sl)), 8 channels (16 B) per thread ------------------------
// Feeds the bf16 3x3 tile kernels, whose LDS images are then filled by LDS-DMA without touching the VALU.
// A thread keeps ONE 8-channel chunk for the whole launch (its 24 table values stay in registers) and walks pixels: a block pass
// covers 256/cpr pixels x cpr chunks, consecutive lanes = consecutive 16-B chunks of a row.  (Before: a flat index over (pixel, chunk)
// with a 64-bit division and 24 table loads per 16 B moved.)
struct ChanTab { float sc[8], sh[8], sl[8]; };
// The workgroup stages the three tables in LDS once (coalesced, 3*C loads per workgroup) and every thread takes its 24 values from
// there: per-thread table loads from global memory -- 24 per thread, the same few lines for every wave of every CU -- queue up on the
// two or three L2 channels that hold those lines (measured: +15-30 us on a 10 us launch).  Call with all threads of the block.
__device__ __forceinline__ ChanTab chan_tab(float* __restrict__ lds, const float* __restrict__ sc, const float* __restrict__ sh,
                                            const float* __restrict__ sl, int c, int C) {
    const int C8 = (C + 7) & ~7;
    for (int i = threadIdx.x; i < C8; i += blockDim.x) {
        const bool ok = i < C;
        const int k = ok ? i : C - 1;
        const float x = sc[k], y = sh[k], z = sl[k];
        lds[i] = ok ? x : 0.f; lds[C8 + i] = ok ? y : 0.f; lds[2 * C8 + i] = ok ? z : 0.f;
    }
    __syncthreads();
    ChanTab t;
    const int cc = c < C8 ? c : 0;                               // idle threads (beyond the last pixel of a block pass) read chunk 0
    const float4 a0 = *reinterpret_cast<const float4*>(lds + cc), a1 = *reinterpret_cast<const float4*>(lds + cc + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(lds + C8 + cc), b1 = *reinterpret_cast<const float4*>(lds + C8 + cc + 4);
    const float4 d0 = *reinterpret_cast<const float4*>(lds + 2 * C8 + cc), d1 = *reinterpret_cast<const float4*>(lds + 2 * C8 + cc + 4);
    t.sc[0] = a0.x; t.sc[1] = a0.y; t.sc[2] = a0.z; t.sc[3] = a0.w; t.sc[4] = a1.x; t.sc[5] = a1.y; t.sc[6] = a1.z; t.sc[7] = a1.w;
    t.sh[0] = b0.x; t.sh[1] = b0.y; t.sh[2] = b0.z; t.sh[3] = b0.w; t.sh[4] = b1.x; t.sh[5] = b1.y; t.sh[6] = b1.z; t.sh[7] = b1.w;
    t.sl[0] = d0.x; t.sl[1] = d0.y; t.sl[2] = d0.z; t.sl[3] = d0.w; t.sl[4] = d1.x; t.sl[5] = d1.y; t.sl[6] = d1.z; t.sl[7] = d1.w;
    return t;
}
__device__ __forceinline__ u16x8 act8(const u16x8 v, const ChanTab& t, int c, int C, bool tail) {
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(prelu(fmaf(bf2f(v[j]), t.sc[j], t.sh[j]), t.sl[j]));
    if (tail) {                                                  // channels >= C are written as zeros (v may hold anything there)
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c + j >= C) o[j] = 0;
    }
    return o;
}
__global__ __launch_bounds__(256) void k_act_bf16(const ActArgs a) {
    extern __shared__ __attribute__((aligned(16))) float act_lds[];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.X);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.Out);
    const int cpr = (a.C + 7) >> 3;                            // chunks per row; the tail chunk is zero padded in Out
    const int ppb = 256 / cpr;                                 // pixels per block pass (launcher: cpr <= 256)
    const int q = threadIdx.x / cpr, c = (threadIdx.x - q * cpr) * 8;
    const bool tail = c + 8 > a.C;
    const long stride = (long)gridDim.x * ppb;
    long m = (long)blockIdx.x * ppb + q;
    const bool live = q < ppb && m < a.M;
    // two rows per trip; the first trip's rows are requested BEFORE the tables are staged, so that a thread with a single trip (the
    // small maps of blocks 3-5) pays one memory round trip, not two.  Row stride ldx >= round_up(C, 8).
    const u16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    u16x8 v0 = z8, v1 = z8;                                    // (never a copy of a register still in flight)
    if (live) {
        v0 = *reinterpret_cast<const u16x8*>(X + m * a.ldx + c);
        if (m + stride < a.M) v1 = *reinterpret_cast<const u16x8*>(X + (m + stride) * a.ldx + c);
    }
    const ChanTab t = chan_tab(act_lds, a.sc, a.sh, a.sl, c, a.C);
    if (!live) return;
    for (;;) {
        const bool two = m + stride < a.M;
        const long mn = m + 2 * stride;
        const bool more = mn < a.M;
        u16x8 n0 = z8, n1 = z8;
        if (more) {
            n0 = *reinterpret_cast<const u16x8*>(X + mn * a.ldx + c);
            if (mn + stride < a.M) n1 = *reinterpret_cast<const u16x8*>(X + (mn + stride) * a.ldx + c);
        }
        *reinterpret_cast<u16x8*>(O + m * a.ldo + c) = act8(v0, t, c, a.C, tail);
        if (two) *reinterpret_cast<u16x8*>(O + (m + stride) * a.ldo + c) = act8(v1, t, c, a.C, tail);
        if (!more) break;
        v0 = n0; v1 = n1; m = mn;
    }
}

// ---- pooled activation in front of a transition conv: XP[img,ho,wo,c] = 1/4 sum_{2x2} prelu(bn(D)) ------------------------
// same thread mapping; 32-bit pixel arithmetic (launcher: n_img*Ho*Wo < 2^31)
__global__ __launch_bounds__(256) void k_act_pool_bf16(const ActPoolArgs a) {
    extern __shared__ __attribute__((aligned(16))) float act_lds[];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.X);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.Out);
    const int Ho = a.Hin / 2, Wo = a.Win / 2, cpr = (a.C + 7) >> 3;
    const int ppb = 256 / cpr;
    const int q = threadIdx.x / cpr, c = (threadIdx.x - q * cpr) * 8;
    const unsigned total = (unsigned)a.n_img * Ho * Wo, stride = gridDim.x * ppb;
    unsigned mo = blockIdx.x * ppb + q;
    const bool live = q < ppb && mo < total;
    auto window = [&](unsigned mo_, u16x8 (&v)[4]) {
        const unsigned row = mo_ / Wo, wo = mo_ - row * Wo;       // row = img*Ho + ho
        const unsigned img = row / Ho, ho = row - img * Ho;
        const long p00 = ((long)img * a.Hin + 2 * ho) * a.Win + 2 * wo;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const u16x8*>(X + (p00 + (k >> 1) * a.Win + (k & 1)) * a.ldx + c);
    };
    u16x8 v[4], vn[4];
    if (live) window(mo, v);                                     // requested before the tables: one round trip for a single-trip thread
    const ChanTab t = chan_tab(act_lds, a.sc, a.sh, a.sl, c, a.C);
    if (!live) return;
    for (;;) {
        const bool more = mo + stride < total;
        if (more) window(mo + stride, vn);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += prelu(fmaf(bf2f(v[k][j]), t.sc[j], t.sh[j]), t.sl[j]);
            o[j] = c + j < a.C ? f2bf(acc * 0.25f) : (bf16)0;    // channels >= C may hold anything (never written): select, not multiply
        }
        *reinterpret_cast<u16x8*>(O + (long)mo * a.ldo + c) = o;
        if (!more) break;
        v[0] = vn[0]; v[1] = vn[1]; v[2] = vn[2]; v[3] = vn[3];
        mo += stride;
    }
}

// fp32 parity mode (round 4): the same materialised operand for the transitions.  The generic implicit-GEMM kernels regenerated the
// pooled activation per operand element (four loads + four BatchNorm/PReLU evaluations, tables from global memory) in BOTH the forward
// and the weight-gradient loader -- k_conv_wgrad<float, A_1X1_POOL> was bound by that loader (9.6 ms per step).  One thread = one pooled
// pixel x 4 channels; columns [C, ldo) are written as zeros (they are GEMM K-padding).
__global__ __launch_bounds__(256) void k_act_pool_f32(const ActPoolArgs a) {
    const float* __restrict__ X = reinterpret_cast<const float*>(a.X);
    float* __restrict__ O = reinterpret_cast<float*>(a.Out);
    const int Ho = a.Hin / 2, Wo = a.Win / 2, cpr = (int)(a.ldo >> 2);
    const long total = (long)a.n_img * Ho * Wo * cpr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long mo = idx / cpr;
        const int c = (int)(idx - mo * cpr) * 4;
        const long row = mo / Wo;
        const int wo = (int)(mo - row * Wo);
        const long img = row / Ho;
        const int ho = (int)(row - img * Ho);
        const long p00 = ((long)img * a.Hin + 2 * ho) * a.Win + 2 * wo;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 4 <= a.C) {
            const float4 sc = make_float4(a.sc[c], a.sc[c + 1], a.sc[c + 2], a.sc[c + 3]), sh = make_float4(a.sh[c], a.sh[c + 1], a.sh[c + 2], a.sh[c + 3]);
            const float4 sl = make_float4(a.sl[c], a.sl[c + 1], a.sl[c + 2], a.sl[c + 3]);      // (parameter views: no alignment promise)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(X + (p00 + (k >> 1) * a.Win + (k & 1)) * a.ldx + c);
                acc[0] += prelu(fmaf(v.x, sc.x, sh.x), sl.x); acc[1] += prelu(fmaf(v.y, sc.y, sh.y), sl.y);
                acc[2] += prelu(fmaf(v.z, sc.z, sh.z), sl.z); acc[3] += prelu(fmaf(v.w, sc.w, sh.w), sl.w);
            }
        } else {
            for (int j = 0; j < 4 && c + j < a.C; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[j] += prelu(fmaf(X[(p00 + (k >> 1) * a.Win + (k & 1)) * a.ldx + c + j], a.sc[c + j], a.sh[c + j]), a.sl[c + j]);
        }
        *reinterpret_cast<float4*>(O + mo * a.ldo + c) = make_float4(acc[0] * 0.25f, acc[1] * 0.25f, acc[2] * 0.25f, acc[3] * 0.25f);
    }
}

// ---- weight re-layout -----------------------------------------------------------------------------------------------
// transpose == 0 : logical B[n][tap*Cin + c] = src[n][c][tap]                   (forward / wgrad operand layout)
// transpose == 1 : logical B[c][tap*N + n]   = src[n][c][tap]   (dgrad operand: rows = input channel, k = (tap, out ch))
// frag == 0 : dst = B row-major [rows][Kp]
// frag == 1 : dst in MFMA 32x32x16 B-fragment order: ((row/32 * Kp/16 + k/16) * 64 + ((k>>3)&1)*32 + row%32) * 8 + k%8, rows
//             zero padded to a multiple of 32 -- one wave-instruction then reads 1 KiB contiguous
// A thread produces EIGHT consecutive k of one row (round 5: one element per trip with two 64-bit divisions each moved the SDXL embedder's
// 20 M weights at 0.2 TB/s, 550 us per launch at the head of every forward pass): its eight source loads are requested together, the
// (tap, channel) walk is incremental, the result leaves as one 16-B (bf16) / 32-B (fp32) store -- contiguous in both layouts (Kp % 8 == 0).
template <typename T>
__global__ __launch_bounds__(256) void k_pack(const PackDesc* descs) {
    const PackDesc d = descs[blockIdx.y];
    T* dst = reinterpret_cast<T*>(d.dst);
    const int rows = d.transpose ? d.Cin : d.N;
    const int prow = d.frag ? (rows + 31) / 32 * 32 : rows;
    const int kc8 = d.Kp >> 3;
    const long total = (long)prow * kc8;
    const int inner = d.transpose ? d.N : d.Cin;                  // extent of the fast k index within a tap
    const int klim = d.taps * inner;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int r = (int)(q / kc8), k0 = (int)(q - (long)r * kc8) * 8;
        float v[8];
        int tap = k0 / inner, c = k0 - tap * inner;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = 0.f;
            if (r < rows && k0 + j < klim)
                v[j] = d.transpose ? d.src[((long)c * d.Cin + r) * d.taps + tap] : d.src[((long)r * d.Cin + c) * d.taps + tap];
            if (++c == inner) { c = 0; ++tap; }
        }
        const long o = d.frag ? ((((long)(r >> 5) * (d.Kp >> 4) + (k0 >> 4)) * 64 + ((k0 >> 3) & 1) * 32 + (r & 31)) * 8) : (long)r * d.Kp + k0;
        if constexpr (sizeof(T) == 2) {
            u16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = f2bf(v[j]);
            *reinterpret_cast<u16x8*>(dst + o) = w;
        } else {
            *reinterpret_cast<f32x4*>(dst + o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    }
}


// bf16, 64 channels: one thread per (pooled pixel, 8-channel chunk), nine 16-B loads of the conv0 output, one 16-B store
__global__ __launch_bounds__(256) void k_pool0_vec64(const Pool0Args a) {
    __shared__ double red[4][8][8][2];
    const bf16* X = reinterpret_cast<const bf16*>(a.X);
    bf16* O = reinterpret_cast<bf16*>(a.Out);
    const int tid = threadIdx.x, c8 = tid & 7;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = a.sc[c8 * 8 + j]; sh[j] = a.sh[c8 * 8 + j]; sl[j] = a.sl[c8 * 8 + j]; }
    double s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; }
    const long npix = (long)a.n_img * a.Ho * a.Wo;
    for (long p = (long)blockIdx.x * 32 + (tid >> 3); p < npix; p += (long)gridDim.x * 32) {
        const int wo = (int)(p % a.Wo);
        const int ho = (int)((p / a.Wo) % a.Ho);
        const long img = p / ((long)a.Wo * a.Ho);
        const bf16* base = X + ((img * a.Hin + 2 * ho) * a.Win + 2 * wo) * 64 + c8 * 8;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u16x8 v = *reinterpret_cast<const u16x8*>(base + ((long)dy * a.Win + dx) * 64);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += prelu(fmaf(bf2f(v[j]), sc[j], sh[j]), sl[j]);
            }
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = f2bf(acc[j] * (1.0f / 9.0f));
            const double x = (double)bf2f(o[j]);
            s1[j] += x; s2[j] += x * x;
        }
        *reinterpret_cast<u16x8*>(O + p * a.ldo + c8 * 8) = o;
    }
    if (a.part == nullptr) return;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
        if (lane < 8) { red[wave][lane][j][0] = s1[j]; red[wave][lane][j][1] = s2[j]; }
    }
    __syncthreads();
    if (tid < 64) {
        const int ch = tid >> 3, j = tid & 7;
        double x = 0, y = 0;
        for (int w = 0; w < 4; ++w) { x += red[w][ch][j][0]; y += red[w][ch][j][1]; }
        a.part[((long)blockIdx.x * 64 + tid) * 2] = x;
        a.part[((long)blockIdx.x * 64 + tid) * 2 + 1] = y;
    }
}
}  // namespace

int bn_link(const BnLinkArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_link, dim3(cdiv(a.C, 4)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int bn_eval_tables(const BnEvalDesc* d_descs, int n, float eps, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_bn_eval_tables, dim3(n), dim3(256), 0, st, d_descs, eps);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int scatter_pixels(const ScatterArgs& a, hipStream_t st) {
    if (a.nnz <= 0) return 0;
    if (a.mode == MODE_F32) hipLaunchKernelGGL(k_scatter<float>, dim3(cdiv(a.nnz, 256)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_scatter<bf16>, dim3(cdiv(a.nnz, 256)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int pool0_grid(int n_img, int Ho, int Wo) {
    const long npix = (long)n_img * Ho * Wo;
    const long g = (npix + 3) / 4;
    return (int)(g < 1024 ? g : 1024);
}

int pool0_fwd(const Pool0Args& a, hipStream_t st) {
    if (a.C > 64 * POOL_CJ) { fprintf(stderr, "tcvn: pool0 supports up to %d channels\n", 64 * POOL_CJ); return -2; }
    const int gx = pool0_grid(a.n_img, a.Ho, a.Wo);
    if (a.part != nullptr && a.nblk != gx) { fprintf(stderr, "tcvn: pool0 nblk mismatch\n"); return -3; }
    if (a.mode == MODE_F32) hipLaunchKernelGGL(k_pool0<float>, dim3(gx), dim3(256), 0, st, a);
    else if (a.C == 64 && (a.ldo & 7) == 0) hipLaunchKernelGGL(k_pool0_vec64, dim3(gx), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_pool0<bf16>, dim3(gx), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int head_pool_fwd(const HeadPoolArgs& a, hipStream_t st) {
    if (a.mode == MODE_F32) hipLaunchKernelGGL(k_head_pool<float>, dim3(a.n_img), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_head_pool<bf16>, dim3(a.n_img), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int act_bf16(const ActArgs& a, hipStream_t st) {
    if ((a.ldx & 7) || (a.ldo & 7) || a.ldx < ((a.C + 7) & ~7) || a.ldo < ((a.C + 7) & ~7)) return -2;
    const int cpr = (a.C + 7) >> 3;
    if (cpr > 256) return -2;
    const long g = (a.M + 2 * (256 / cpr) - 1) / (2 * (256 / cpr));                 // a thread handles >= 2 rows where there are that many
    hipLaunchKernelGGL(k_act_bf16, dim3((unsigned)(g < 4096 ? (g < 1 ? 1 : g) : 4096)), dim3(256), 3 * cpr * 8 * sizeof(float), st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int act_pool_bf16(const ActPoolArgs& a, hipStream_t st) {
    if ((a.ldx & 7) || (a.ldo & 7) || a.ldx < ((a.C + 7) & ~7) || a.ldo < ((a.C + 7) & ~7)) return -2;
    const int cpr = (a.C + 7) >> 3;
    const long px = (long)a.n_img * (a.Hin / 2) * (a.Win / 2);
    if (cpr > 256 || px >= (1L << 31) - 4096L * 256) return -2;
    if (px <= 0) return 0;
    const long g = (px + 256 / cpr - 1) / (256 / cpr);
    hipLaunchKernelGGL(k_act_pool_bf16, dim3((unsigned)(g < 4096 ? g : 4096)), dim3(256), 3 * cpr * 8 * sizeof(float), st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int act_pool_f32(const ActPoolArgs& a, hipStream_t st) {
    if ((a.ldx & 3) || (a.ldo & 3) || a.ldo < a.C || (reinterpret_cast<uintptr_t>(a.X) & 15) || (reinterpret_cast<uintptr_t>(a.Out) & 15)) return -2;
    const long total = (long)a.n_img * (a.Hin / 2) * (a.Win / 2) * (a.ldo >> 2);
    if (total <= 0) return 0;
    const long g = (total + 255) / 256;
    hipLaunchKernelGGL(k_act_pool_f32, dim3((unsigned)(g < 8192 ? g : 8192)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int pack_weights(const PackDesc* d_descs, int n, int mode, hipStream_t st) {
    if (n <= 0) return 0;
    // 256 workgroups per descriptor (grid-stride loop: small descriptors leave most of them idle): with 32, the 2.4 M-element layouts of the
    // SDXL embedder's 512 -> 512 3x3 layers took 860 us per launch, twice per step
    if (mode == MODE_F32) hipLaunchKernelGGL(k_pack<float>, dim3(256, n), dim3(256), 0, st, d_descs);
    else hipLaunchKernelGGL(k_pack<bf16>, dim3(256, n), dim3(256), 0, st, d_descs);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
