// Fused backward of a bottleneck 1x1 convolution (bf16): ONE pass over the pixels does what k_eff_mat + k_gemm_tn_bf16<conv1> +
// k_gemm_nt_bf16<dgrad1x1> did in three (reference: autograd of Bottleneck.bottleneck_block = BN - PReLU - conv1,
// layers/dense_net.py:18-27):
//     EY[m][k]  = bf16( DU[m][k] + PY[k]*Y[m][k] + QY[k] )                  gradient of the conv1 output (norm2's batch-mean terms folded in)
//     dbias[k] += sum_m EY[m][k]
//     dX[m][c]  = sum_k EY[m][k] * W1[k][c]                                  data gradient (K = 128)
//     u = sc*x + sh ; dU = dX * prelu'(u) ; G[m][c] += sc * dU ; partial sums (sum dU, sum dU*x, sum dX*min(u,0))   norm1 / PReLU1 backward
//     dW1[k][c] += sum_m EY[m][k] * prelu(u)[m][c]                           weight gradient (contraction over the pixels)
// A workgroup owns a 64-pixel tile and a 128-column slice of the cin input channels.  DU, Y and the raw BatchNorm input x arrive by
// LDS-DMA; EY is formed in LDS in place of DU; the data-gradient MFMAs read it row-wise, the weight-gradient MFMAs read it (and the
// activated input, which the epilogue leaves in place of x -- it computes u anyway) column-wise with ds_read_b64_tr_b16.  Neither EY nor
// the activated input exists in HBM for these kernels: per pixel and layer the backward of the 1x1 moves 512 + 6*cin bytes instead of
// 1792 + 8*cin (EY written once and read twice, the activated copy read once, x and G as here).
// The weight-gradient tile (128 x 128 fp32 = 64 accumulator registers per lane) stays in registers for the whole launch and leaves as one
// slab per workgroup (k_slab_reduce: the slabs of one y-slice are summed in a fixed order; launches with more than 32 slabs split them over
// up to 16 y-slices whose partial sums meet in fp32 atomics, i.e. the last bits of a weight gradient can differ from run to run).
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int ROWS = 64;
constexpr int TILE = ROWS * 256;                 // one [64][128] bf16 operand tile
constexpr int CLD = 132;                         // C tile leading dimension (floats), padded
constexpr int OFF_E = 0;                         // DU -> EY
constexpr int OFF_X = TILE;                      // x (this workgroup's 128-column slice) -> prelu(bn1(x))
constexpr int OFF_Y = 2 * TILE;                  // Y; free after EY is formed:
constexpr int OFF_C = 2 * TILE;                  //   the fp32 C tile [64][CLD] of the data gradient aliases it
constexpr int OFF_TAB = OFF_C + ROWS * CLD * 4;          // [5][128] floats: PY, QY (EY channels), sc, sh, sl (this workgroup's column slice)
constexpr int SMEM_BYTES = OFF_TAB + 5 * 128 * 4;        // 69 120 B: two workgroups per CU

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8_t tr_frag(const char* smem_base, int off_lo, int off_hi) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_hi));
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8_t, pr);
}

// [64 rows][128 columns] of a row-major bf16 matrix -> LDS, 16-B chunks XOR-swizzled by the row (slot s of row r holds source chunk s ^ swz16(r):
// conflict-free for the row-wise fragment reads of the data gradient AND for the transposed reads of the weight gradient, tcvn_common.h)
__device__ __forceinline__ void dma64(char* smem_base, int buf_off, const bf16* __restrict__ A, long lda, int K, int k0, long m0, long M,
                                      const char* __restrict__ zeros, int wave, int lane) {
    const int rsub = lane >> 4, slot = lane & 15;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        const int rg = wave + 4 * i;
        const int r = rg * 4 + rsub;
        const int col = k0 + ((slot ^ swz16(r)) << 3);
        const long m = m0 + r;
        const char* src = (m < M && col < K) ? reinterpret_cast<const char*>(A + m * lda + col) : zeros + (slot << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem_base + buf_off + rg * 1024), 16, 0, 0);
    }
}

__global__ __launch_bounds__(256, 2) void k_bwd1x1_fused_bf16(const Bwd1x1Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Cs = reinterpret_cast<float*>(smem + OFF_C);
    double* red = reinterpret_cast<double*>(smem + OFF_C);                 // [4][128][3], after the last tile
    float* csum = reinterpret_cast<float*>(smem + OFF_C + 4 * 128 * 3 * 8);   // [4][128], after the last tile
    float* tab = reinterpret_cast<float*>(smem + OFF_TAB);

    const int tid = threadIdx.x;
    const int N = g.cin, n0 = blockIdx.y * 128;
    const bf16* __restrict__ DU = reinterpret_cast<const bf16*>(g.DU);
    const bf16* __restrict__ Yp = reinterpret_cast<const bf16*>(g.Y);
    const bf16* __restrict__ Xp = reinterpret_cast<const bf16*>(g.Xin);
    bf16* __restrict__ Gp = reinterpret_cast<bf16*>(g.Gout);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    const long mtiles = (g.M + ROWS - 1) / ROWS;
    const u16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // data gradient: wave w owns output columns n0 + 32w .. +32 for all 64 rows; its 8 weight fragments (K = 128) stay in registers
    const bool wave_live = n0 + (tid >> 6) * 32 < N;
    bf16x8_t bw[8];
    {
        const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + (((long)(blockIdx.y * 4 + (tid >> 6)) * 8) * 64 + (tid & 63)) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (wave_live) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + (long)i * 512);
            else
#pragma unroll
                for (int j = 0; j < 8; ++j) bw[i][j] = (__bf16)0.f;
        }
    }
    // per-channel tables live in LDS (24 + 16 values per thread would otherwise sit in registers next to the 64 accumulator registers of
    // the weight-gradient tile): a phase reads the eight values of its chunk as two 16-B loads per table
    if (tid < 128) {
        const bool ok = n0 + tid < N;
        tab[tid] = g.PY[tid]; tab[128 + tid] = g.QY[tid];
        tab[256 + tid] = ok ? g.sc[n0 + tid] : 0.f; tab[384 + tid] = ok ? g.sh[n0 + tid] : 0.f; tab[512 + tid] = ok ? g.sl[n0 + tid] : 0.f;
    }
    float st1[8], st2[8], st3[8], cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[j] = 0.f; st2[j] = 0.f; st3[j] = 0.f; cs[j] = 0.f; }
    // weight gradient: 2 x 2 waves, 64 (EY channels) x 64 (input channels of this slice) each
    const bool wj_live = n0 + ((tid >> 6) & 1) * 64 < N;
    f32x16 accW[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) accW[a][b][e] = 0.f;

    // Operand requests.  Full tiles (all but at most one per launch): a uniform base per matrix and 16-row group plus ONE 32-bit lane
    // offset (the swizzled chunk of a lane does not depend on the row group: (4*wave + rsub + 16*i) & 15), i.e. three address registers for
    // the twelve DMAs and the four G loads -- per-lane 64-bit source pointers with their range selects are ~50 registers of loop
    // invariants next to the accumulators.  x chunks beyond cin are never requested: their LDS slots are zeroed once, nothing writes them.
    const int col_chunk = tid & 15;                                        // element-wise roles: 16 threads per row, 8 channels each
    const bool col_ok = n0 + col_chunk * 8 < N;
    const int col_rem = N - (n0 + col_chunk * 8);                          // < 8 only in a last, partial chunk (cin % 8 != 0): the foreign channels beside it
                                                                           // (the layer's own 3x3 output slice) are neither used as x nor rewritten in G
    const int d_r0 = (tid >> 6) * 4 + ((tid & 63) >> 4);                   // DMA: this lane's row in row group i is d_r0 + 16*i
    const int d_chunk = (tid & 15) ^ swz16(d_r0);
    const bool xchunk_ok = n0 + (d_chunk << 3) < N;
    const unsigned voffA = (unsigned)(d_r0 * 256 + (d_chunk << 4));
    const unsigned voffX = (unsigned)(d_r0 * (int)g.ldx * 2 + (d_chunk << 4));
    const unsigned voffG = (unsigned)(((tid >> 4) * (int)g.ldg + n0 + col_chunk * 8) * 2);
    if (!xchunk_ok) {
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) *reinterpret_cast<u16x8*>(smem + OFF_X + ((tid >> 6) + 4 * i) * 1024 + (tid & 63) * 16) = z8;
    }
    u16x8 pgv[ROWS / 16];
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) pgv[i] = z8;
    auto request = [&](long t) {          // the three operand tiles of row tile t, and this thread's G rows of it (for the epilogue)
        const long m0 = t * ROWS;
        const int wave = tid >> 6, lane = tid & 63;
        if (m0 + ROWS <= g.M) {
            const char* bE = reinterpret_cast<const char*>(DU) + m0 * 256;
            const char* bY = reinterpret_cast<const char*>(Yp) + m0 * 256;
            const char* bX = reinterpret_cast<const char*>(Xp) + (m0 * g.ldx + n0) * 2;
            const char* bG = reinterpret_cast<const char*>(Gp) + m0 * g.ldg * 2;
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bE + i * 4096 + voffA),
                                                 (__attribute__((address_space(3))) void*)(smem + OFF_E + (wave + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bY + i * 4096 + voffA),
                                                 (__attribute__((address_space(3))) void*)(smem + OFF_Y + (wave + 4 * i) * 1024), 16, 0, 0);
            if (xchunk_ok) {
#pragma unroll
                for (int i = 0; i < ROWS / 16; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bX + (long)i * 32 * g.ldx + voffX),
                                                     (__attribute__((address_space(3))) void*)(smem + OFF_X + (wave + 4 * i) * 1024), 16, 0, 0);
            }
            if (col_ok) {
#pragma unroll
                for (int i = 0; i < ROWS / 16; ++i) pgv[i] = *reinterpret_cast<const u16x8*>(bG + (long)i * 32 * g.ldg + voffG);
            }
        } else {                           // the launch's last, partial tile: rows beyond M come from the zero line
            dma64(smem, OFF_E, DU, 128, 128, 0, m0, g.M, zeros, wave, lane);
            dma64(smem, OFF_Y, Yp, 128, 128, 0, m0, g.M, zeros, wave, lane);
            if (xchunk_ok) dma64(smem, OFF_X, Xp, g.ldx, N, n0, m0, g.M, zeros, wave, lane);
            if (col_ok) {
#pragma unroll
                for (int i = 0; i < ROWS / 16; ++i) {
                    const long m = m0 + (tid >> 4) + 16 * i;
                    if (m < g.M) pgv[i] = *reinterpret_cast<const u16x8*>(Gp + m * g.ldg + n0 + col_chunk * 8);
                }
            }
        }
    };

    long mt = blockIdx.x;
    if (mt < mtiles) request(mt);
    for (; mt < mtiles; mt += gridDim.x) {
        const long m0 = mt * ROWS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // Index arithmetic of the phases below starts from an opaque copy of the thread index: left to loop-invariant code motion the
        // compiler keeps ~60 LDS addresses of all four phases alive across the whole loop, next to the accumulators, and spills.
        int t_o = tid;
        asm volatile("" : "+v"(t_o));
        const int c8 = t_o & 15, c_r0 = t_o >> 4;                          // element-wise roles: rows c_r0 + 16*i, channel chunk c8
        const int e_off = c_r0 * 256 + ((c8 ^ swz16(c_r0)) << 4);          // (c_r0 + 16*i) & 15 == c_r0 & 15: the row group adds i*4096
        auto tab8 = [&](int which, float (&v)[8]) {
            const float4 a = *reinterpret_cast<const float4*>(tab + which * 128 + c8 * 8), b = *reinterpret_cast<const float4*>(tab + which * 128 + c8 * 8 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        };
        // ---- EY = bf16(DU + PY*Y + QY) in place of DU (rows beyond M: zero), column sums for the bias gradient
        {
            float pP[8], pQ[8];
            tab8(0, pP); tab8(1, pQ);
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) {
                const int off = e_off + i * 4096;
                const u16x8 dv = *reinterpret_cast<const u16x8*>(smem + OFF_E + off);
                const u16x8 yv = *reinterpret_cast<const u16x8*>(smem + OFF_Y + off);
                const bool live = m0 + c_r0 + 16 * i < g.M;
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = eff3(bf2f(dv[j]), pP[j], bf2f(yv[j]), pQ[j]);
                    o[j] = live ? f2bf(t) : (bf16)0;
                    cs[j] += bf2f(o[j]);
                }
                *reinterpret_cast<u16x8*>(smem + OFF_E + off) = o;
            }
        }
        __syncthreads();
        // ---- data gradient: C[64][128-column slice] = EY x W1 (K = 128)
        {
            const int lane = t_o & 63, wave = t_o >> 6;
            const int r = lane & 31, h = lane >> 5;
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            if (wave_live) {
                // fragments one k-step ahead of the MFMAs that use them (left to the compiler all sixteen reads are hoisted: 64 registers);
                // chunk (2*ks + h) ^ swz16(r) == (2*ks) ^ w with w = h ^ swz16(r)
                const int a_base = OFF_E + r * 256, w4 = (h ^ swz16(r)) << 4;
                auto afrag = [&](int ks, int i) {
                    return *reinterpret_cast<const bf16x8_t*>(smem + a_base + i * 8192 + (w4 ^ (ks << 5)));
                };
                bf16x8_t a0 = afrag(0, 0), a1 = afrag(0, 1);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    bf16x8_t b0 = a0, b1 = a1;
                    if (ks + 1 < 8) { b0 = afrag(ks + 1, 0); b1 = afrag(ks + 1, 1); }
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bw[ks], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[ks], acc[1], 0, 0, 0);
                    a0 = b0; a1 = b1;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // C tile -> LDS (fp32; aliases the Y tile, last read before the barrier above): row = i*32 + (e&3) + 8*(e>>2) + 4*h, column = wave*32 + r
            float* cw = Cs + 4 * h * CLD + wave * 32 + r;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) cw[(i * 32 + (e & 3) + 8 * (e >> 2)) * CLD] = acc[i][e];
        }
        __syncthreads();
        // ---- epilogue: PReLU1 / norm1 backward against x (LDS), G += sc*dU (16-B lanes), activated input left in place of x
        if (col_ok) {
            float csc[8], csh[8], csl[8];
            tab8(2, csc); tab8(3, csh); tab8(4, csl);
            const float* crow = Cs + c_r0 * CLD + c8 * 8;
            char* gbase = reinterpret_cast<char*>(Gp) + m0 * g.ldg * 2;
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) {
                if (m0 + c_r0 + 16 * i < g.M) {
                    const int xoff = OFF_X + e_off + i * 4096;
                    u16x8 xv = *reinterpret_cast<const u16x8*>(smem + xoff);
                    if (col_rem < 8) {                                      // partial chunk (never in the tutorial widths): foreign x -> 0, so no
#pragma unroll                                                              // non-finite foreign value can reach the sums or the activated operand
                        for (int j = 0; j < 8; ++j) xv[j] = j < col_rem ? xv[j] : (bf16)0;
                    }
                    const float4 ca = *reinterpret_cast<const float4*>(crow + i * 16 * CLD);
                    const float4 cc = *reinterpret_cast<const float4*>(crow + i * 16 * CLD + 4);
                    const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cc.x, cc.y, cc.z, cc.w};
                    const u16x8 gv = pgv[i];
                    u16x8 o, xa;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = bf2f(xv[j]);
                        const float u = fmaf(x, csc[j], csh[j]);
                        const float dA = cv[j];
                        const float du = u > 0.f ? dA : csl[j] * dA;
                        st1[j] += du; st2[j] = fmaf(du, x, st2[j]); st3[j] = fmaf(u > 0.f ? 0.f : dA, u, st3[j]);
                        o[j] = f2bf(fmaf(csc[j], du, bf2f(gv[j])));
                        xa[j] = f2bf(prelu(u, csl[j]));
                    }
                    if (col_rem >= 8) *reinterpret_cast<u16x8*>(gbase + (long)i * 32 * g.ldg + voffG) = o;
                    else {                                                  // partial chunk: the layer's own channels only (the next ones belong to its 3x3
                        bf16* gp = reinterpret_cast<bf16*>(gbase + (long)i * 32 * g.ldg + voffG);     // output gradient, which other kernels may be reading)
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j < col_rem) gp[j] = o[j];
                    }
                    *reinterpret_cast<u16x8*>(smem + xoff) = xa;
                }
            }
        }
        __syncthreads();
        // ---- weight gradient: accW += EY^T x prelu(bn1(x)) over the tile's 64 pixels
        if (wj_live) {
            const int lane = t_o & 63, wave = t_o >> 6;
            const int wi = wave >> 1, wj = wave & 1;
            const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
            const int khalf = gq >> 1, chalf = gq & 1;
            const int sub = (tp & 1) * 8;
            const int rl = 8 * khalf + tq;                                  // row inside a 16-row k-step (rows rl and rl + 4); k-step ks adds ks*4096
            int a_lo[2], a_hi[2], b_lo[2], b_hi[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ac = wi * 8 + t * 4 + 2 * chalf + (tp >> 1), bc = wj * 8 + t * 4 + 2 * chalf + (tp >> 1);
                a_lo[t] = OFF_E + rl * 256 + ((ac ^ swz16(rl)) << 4) + sub; a_hi[t] = OFF_E + (rl + 4) * 256 + ((ac ^ swz16(rl + 4)) << 4) + sub;
                b_lo[t] = OFF_X + rl * 256 + ((bc ^ swz16(rl)) << 4) + sub; b_hi[t] = OFF_X + (rl + 4) * 256 + ((bc ^ swz16(rl + 4)) << 4) + sub;
            }
#pragma unroll
            for (int ks = 0; ks < ROWS / 16; ++ks) {
                bf16x8_t a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = tr_frag(smem, a_lo[t] + ks * 4096, a_hi[t] + ks * 4096);
                    b[t] = tr_frag(smem, b_lo[t] + ks * 4096, b_hi[t] + ks * 4096);
                }
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) accW[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], accW[ta][tb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                                   // every LDS tile is free again
        if (mt + gridDim.x < mtiles) request(mt + gridDim.x);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- per-workgroup results: statistics partials, bias column sums, the weight-gradient tile
    const int lane = tid & 63, wave = tid >> 6, c8 = tid & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d1 = (double)st1[j], d2 = (double)st2[j], d3 = (double)st3[j];
        float c1 = cs[j];
        d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32);
        d2 += __shfl_xor(d2, 16); d2 += __shfl_xor(d2, 32);
        d3 += __shfl_xor(d3, 16); d3 += __shfl_xor(d3, 32);
        c1 += __shfl_xor(c1, 16); c1 += __shfl_xor(c1, 32);
        if (lane < 16) {
            double* p = red + ((wave * 128) + c8 * 8 + j) * 3;
            p[0] = d1; p[1] = d2; p[2] = d3;
            csum[wave * 128 + c8 * 8 + j] = c1;
        }
    }
    __syncthreads();
    if (tid < 128) {
        if (n0 + tid < N) {
            double a = 0, b = 0, c = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[(w * 128 + tid) * 3]; b += red[(w * 128 + tid) * 3 + 1]; c += red[(w * 128 + tid) * 3 + 2]; }
            double* p = g.part + ((long)blockIdx.x * N + n0 + tid) * 3;
            p[0] = a; p[1] = b; p[2] = c;
        }
        if (blockIdx.y == 0) g.tail[(long)blockIdx.x * 128 + tid] = (csum[tid] + csum[128 + tid]) + (csum[256 + tid] + csum[384 + tid]);
    }
    const int wi = wave >> 1, wj = wave & 1;
    const int cj = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int j = n0 + wj * 64 + tb * 32 + cj;
            if (j >= g.ldc) continue;                                      // columns in [cin, ldc) are written as zeros (padding of the kernel layout)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = wi * 64 + ta * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                g.slab[((long)blockIdx.x * 128 + i) * g.ldc + j] = accW[ta][tb][e];
            }
        }
}

}  // namespace

bool bwd1x1_fused_ok(const Bwd1x1Args& a) {
    if (!a.DU || !a.Y || !a.Xin || !a.Gout || !a.Wfrag || !a.zeros || !a.part || !a.slab || !a.tail) return false;
    if (a.cin <= 0 || a.Kp != 128 || (a.ldx & 7) || (a.ldg & 7) || a.ldc < a.cin || a.ldc > cdiv(a.cin, 128) * 128) return false;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.DU) | reinterpret_cast<uintptr_t>(a.Y) | reinterpret_cast<uintptr_t>(a.Xin) |
                         reinterpret_cast<uintptr_t>(a.Gout) | reinterpret_cast<uintptr_t>(a.Wfrag) | reinterpret_cast<uintptr_t>(a.slab);
    if (al & 15) return false;
    return (long)bwd1x1_fused_nblk(a) * 128 * a.ldc * 4 <= a.slab_bytes;
}

int bwd1x1_fused_nblk(const Bwd1x1Args& a) {
    if (bwd1x1_wide_ok(a)) return bwd1x1_wide_nblk(a);       // 128 < cin <= 512: one workgroup per tile walks the column slices (bwd1x1_wide.hip)
    const int nn = cdiv(a.cin, 128);
    int cap = 512 / nn;                    // two resident workgroups per CU, shared by the nn column slices
    if (cap < 64) cap = 64;
    const long mt = (a.M + ROWS - 1) / ROWS;
    return (int)(mt < cap ? mt : cap);
}

int bwd1x1_fused_launch(const Bwd1x1Args& a, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (!bwd1x1_fused_ok(a)) return -2;
    if (a.nblk != bwd1x1_fused_nblk(a)) { fprintf(stderr, "tcvn: bwd1x1_fused nblk mismatch\n"); return -3; }
    if (bwd1x1_wide_ok(a)) return bwd1x1_wide_launch(a, st);
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd1x1_fused_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // SURVEY 8(d) strict bytes: operands DU, Y, x read once, the G contribution written once (its read is traffic, not algorithm)
    ProfScope ps("k_bwd1x1_fused_bf16", 4.0 * a.M * 128.0 * a.cin, (double)a.M * (512.0 + 4.0 * a.cin), st);
    hipLaunchKernelGGL(k_bwd1x1_fused_bf16, dim3(a.nblk, cdiv(a.cin, 128)), dim3(256), SMEM_BYTES, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

// the launch's per-workgroup weight-gradient tiles and bias column sums -> dWk [128][ldc], dbias [128] (accumulating); any stream ordered
// behind the launch -- nothing on the data-gradient chain waits for it
int bwd1x1_fused_reduce(const Bwd1x1Args& a, float* dWk, float* dbias, const SlabJob* extra, hipStream_t st, const BnBwdLinkArgs* link) {
    SlabJob jobs[4] = {};
    if (a.M > 0) { jobs[0] = slab_job(a.slab, a.nblk, 128L * a.ldc, dWk, 0); jobs[1] = slab_job(a.tail, a.nblk, 128, dbias, 0); }
    if (extra != nullptr) { jobs[2] = extra[0]; jobs[3] = extra[1]; }          // e.g. the same layer's 3x3 weight-gradient slabs: one launch for both
    if (link != nullptr) return slab_reduce4_link(jobs, 4, *link, st);         // ... and the norm1 backward link of the layer (round 5)
    return slab_reduce4(jobs, 4, st);
}

}  // namespace tcvn
