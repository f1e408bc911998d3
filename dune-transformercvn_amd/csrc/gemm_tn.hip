// bf16 "TN" GEMM over the pixel dimension for the 1x1-convolution weight gradients:
//     C[i][j] += sum_m L[m][i] * R[m][j]        L = materialised output gradient [M][Li], R = materialised activation [M][Rj]
// Reference call site: autograd of Bottleneck.bottleneck_block.conv1 and Transition.conv (layers/dense_net.py:21-26,87-92).
// Both operands are row-major in HBM with the contraction index as the ROW, so they are LDS-DMA'd untouched (128 rows x 256 B,
// XOR-swizzled 16-B chunks, double buffered) and read as MFMA fragments with ds_read_b64_tr_b16.  One workgroup owns a
// 128 x 128 tile of C for a slice of the pixels; slices are combined with fp32 atomics on contiguous 128-B rows.
// Algorithmic work: 2*M*Li*Rj FLOP; HBM: M*(Li+Rj)*2 B (L re-read once per 128-column tile of R).
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int ROWS = 64;                       // pixels per stage; 64 KB of LDS per workgroup -> two workgroups per CU
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8_t tr_frag(const char* smem_base, int off_lo, int off_hi) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_hi));
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8_t, pr);
}

// DMA one [ROWS][128-column] operand tile: source rows m0.., columns col0.. (zeros beyond `ncols` / `m_end`)
__device__ __forceinline__ void dma_tile(char* smem_base, int buf_off, const bf16* __restrict__ X, long ld, int ncols, int col0,
                                         long m0, long m_end, const char* __restrict__ zeros, int wave, int lane) {
    const int rsub = lane >> 4, slot = lane & 15;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        const int rg = wave + 4 * i;
        const int r = rg * 4 + rsub;
        const int col = col0 + ((slot ^ swz16(r)) << 3);
        const long m = m0 + r;
        const char* src = (m < m_end && col < ncols) ? reinterpret_cast<const char*>(X + m * ld + col) : zeros + (slot << 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem_base + buf_off + rg * 1024), 16, 0, 0);
    }
}

// XF = 1: R is the RAW BatchNorm input; each landed R tile is transformed in LDS to prelu(sc*x + sh) (rows beyond the slice and
// columns beyond Rreal stay zero), so the weight gradient needs no activated copy of the concat buffer in HBM
template <int XF>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_bf16(const GemmTnArgs g, long rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = ROWS * 256;                               // bytes of one operand tile
    // layout: L[0] L[1] R[0] R[1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;                       // 2 x 2 waves, 64 x 64 of C each
    const int i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
    const long m_begin = (long)blockIdx.z * rows_per_split;
    const long m_end = m_begin + rows_per_split < g.M ? m_begin + rows_per_split : g.M;
    const bf16* __restrict__ Lp = reinterpret_cast<const bf16*>(g.L);
    const bf16* __restrict__ Rp = reinterpret_cast<const bf16*>(g.R);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int khalf = gq >> 1, chalf = gq & 1;
    const int sub = (tp & 1) * 8;
    int achunk[2], bchunk[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        achunk[t] = wi * 8 + t * 4 + 2 * chalf + (tp >> 1);
        bchunk[t] = wj * 8 + t * 4 + 2 * chalf + (tp >> 1);
    }
    float* rtab = reinterpret_cast<float*>(smem + 4 * TILE);               // XF: [3][128] scale, shift, slope of columns j0..j0+127
    if (XF) {
        for (int i = tid; i < 128; i += 256) {
            const bool ok = j0 + i < g.Rreal;
            rtab[i] = ok ? g.rsc[j0 + i] : 0.f; rtab[128 + i] = ok ? g.rsh[j0 + i] : 0.f; rtab[256 + i] = ok ? g.rsl[j0 + i] : 0.f;
        }
    }
    auto xform = [&](int buf, long m0) {
#pragma unroll
        for (int it = 0; it < ROWS * 16 / 256; ++it) {
            const int idx = tid + 256 * it, row = idx >> 4, slot = idx & 15;
            const int cl = (slot ^ swz16(row)) << 3;                       // column inside the 128-column tile
            if (m0 + row < m_end && j0 + cl < g.Rreal) {
                u16x8* p = reinterpret_cast<u16x8*>(smem + buf + row * 256 + (slot << 4));
                const u16x8 v = *p;
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = j0 + cl + j < g.Rreal ? f2bf(prelu(fmaf(bf2f(v[j]), rtab[cl + j], rtab[128 + cl + j]), rtab[256 + cl + j])) : (bf16)0;
                *p = o;
            }
        }
    };
    if (m_begin < m_end) {
        dma_tile(smem, 0, Lp, g.ldl, g.Li, i0, m_begin, m_end, zeros, wave, lane);
        dma_tile(smem, 2 * TILE, Rp, g.ldr, g.Rj, j0, m_begin, m_end, zeros, wave, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (long m0 = m_begin; m0 < m_end; m0 += ROWS, cur ^= 1) {
        if (m0 + ROWS < m_end) {
            dma_tile(smem, (cur ^ 1) * TILE, Lp, g.ldl, g.Li, i0, m0 + ROWS, m_end, zeros, wave, lane);
            dma_tile(smem, 2 * TILE + (cur ^ 1) * TILE, Rp, g.ldr, g.Rj, j0, m0 + ROWS, m_end, zeros, wave, lane);
        }
        const int lb = cur * TILE, rb = 2 * TILE + cur * TILE;
        if (XF) {                               // bare barrier: a __syncthreads() would drain the prefetch just issued
            xform(rb, m0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll 2
        for (int ks = 0; ks < ROWS / 16; ++ks) {
            const int row = ks * 16 + 8 * khalf + tq, row2 = row + 4;
            bf16x8_t a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = tr_frag(smem, lb + row * 256 + ((achunk[t] ^ swz16(row)) << 4) + sub,
                               lb + row2 * 256 + ((achunk[t] ^ swz16(row2)) << 4) + sub);
                b[t] = tr_frag(smem, rb + row * 256 + ((bchunk[t] ^ swz16(row)) << 4) + sub,
                               rb + row2 * 256 + ((bchunk[t] ^ swz16(row2)) << 4) + sub);
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int cj = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int j = j0 + wj * 64 + tb * 32 + cj;
            if (j >= g.ldc) continue;                               // columns in [Rj, ldc) are written as zeros (padding)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wi * 64 + ta * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (i < g.Ci) g.slab[((long)blockIdx.z * g.Ci + i) * g.ldc + j] = acc[ta][tb][e];
            }
        }
}

}  // namespace

bool gemm_tn_ok(const GemmTnArgs& a) {
    return a.L && a.R && a.zeros && (a.ldl & 7) == 0 && (a.ldr & 7) == 0 && (a.Li & 7) == 0 && (a.Rj & 7) == 0 &&
           (reinterpret_cast<uintptr_t>(a.L) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.R) & 15) == 0;
}

int gemm_tn_bf16(const GemmTnArgs& a, const char* label, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (!gemm_tn_ok(a)) return -2;
    const int jt = cdiv(a.Rj, 128), it = cdiv(a.Li, 128);
    int split = cdiv(512, jt * it);
    if (a.slab != nullptr && a.Ci > 0) {                     // every slice needs its own Ci x ldc tile of the slab
        const long cap = a.slab_bytes / ((long)a.Ci * a.ldc * 4);
        if (split > cap) split = (int)cap;
    }
    const long stages = (a.M + ROWS - 1) / ROWS;
    if (split > stages) split = (int)stages;
    if (split < 1) split = 1;
    const long rows = ((stages + split - 1) / split) * ROWS;
    split = (int)((a.M + rows - 1) / rows);
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_bf16<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_bf16<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    if (a.slab == nullptr || a.Ci <= 0 || a.Ci > a.Li || (long)split * a.Ci * a.ldc * 4 > a.slab_bytes || cdiv(a.ldc, 128) > jt) return -3;
    {
        ProfScope ps(label, 2.0 * a.M * (double)a.Li * a.Rj, (double)a.M * 2.0 * (a.Li + a.Rj), st);
        if (a.rsc != nullptr) hipLaunchKernelGGL(k_gemm_tn_bf16<1>, dim3(jt, it, split), dim3(256), 4 * ROWS * 256 + 3 * 128 * 4, st, a, rows);
        else hipLaunchKernelGGL(k_gemm_tn_bf16<0>, dim3(jt, it, split), dim3(256), 4 * ROWS * 256, st, a, rows);
        TCVN_LAUNCH_CHECK();
    }
    return slab_reduce2(slab_job(a.slab, split, (long)a.Ci * a.ldc, a.C, 0), a.extra, st);
}

}  // namespace tcvn
