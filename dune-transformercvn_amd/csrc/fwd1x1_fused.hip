// Forward of a bottleneck 1x1 convolution with its input transform fused (bf16): Y = prelu(bn1(x)) x W1^T + bias, statistics partials of Y
// for norm2 (reference: Bottleneck.bottleneck_block = BN - PReLU - conv1, layers/dense_net.py:18-27).
// The RAW concat buffer is what arrives by LDS-DMA; the landed [64][128-channel] chunks are activated in LDS in place (one thread = one
// 8-channel chunk of four rows, tables from LDS once per chunk column), then read as MFMA fragments.  The activated copy XA of the 1x1
// input (k_act_bf16: 2*cin bytes written and read again per pixel and layer) does not exist for the layers this kernel serves; the fused
// 1x1 backward kernel (bwd1x1_fused.hip) rebuilds the same activation from x on its side.
// One workgroup = one 64-pixel tile, all 128 output channels (a wave owns 32 of them: its weight fragments for the whole K extent stay in
// registers).  K <= 256 (dense blocks 1-2): one or two resident chunks, three to four workgroups per CU; 256 < K <= 512 (k_fwd1x1_wide_bf16):
// the chunks stream through a two-slot ring, two workgroups per CU.  Arithmetic and summation orders are those of k_gemm_nt_bf16<EPI_FWD>
// on the materialised operand: identical Y.  OACT (eval mode): norm2 + PReLU2 of the consumer in the epilogue.
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int ROWS = 64;
constexpr int TILE = ROWS * 256;                 // one [64][128] bf16 chunk
constexpr int CLD = 136;                         // C tile leading dimension (bf16 elements), padded: 272-B rows keep the 16-B epilogue reads conflict-free

template <int NKC> struct FwdCfg {
    static constexpr int OFF_C = NKC * TILE;                       // bf16 C tile [64][CLD]: + bias and the one rounding happen in MFMA layout
    static constexpr int OFF_TAB = OFF_C + ROWS * CLD * 2;         // [3][NKC*128] floats: scale, shift, slope of the input channels
    static constexpr int SMEM = OFF_TAB + 3 * NKC * 128 * 4;       // NKC = 1: 35 328 B, NKC = 2: 53 248 B (three workgroups per CU)
};

// OACT (eval mode: running statistics, so no batch reduction separates the 1x1 output from its BatchNorm): the epilogue applies norm2 + PReLU2
// to the fp32 result before the one rounding -- Out is the activated map the 3x3 kernel stages, the raw bottleneck map does not exist.
template <int NKC, bool OACT>
__global__ __launch_bounds__(256, 3) void k_fwd1x1_fused_bf16(const Fwd1x1Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef FwdCfg<NKC> C;
    constexpr int KS = NKC * 8;
    bf16* Cs = reinterpret_cast<bf16*>(smem + C::OFF_C);
    double* red = reinterpret_cast<double*>(smem);                         // [4][128][2], after the last tile (over the x chunks)
    float* tab = reinterpret_cast<float*>(smem + C::OFF_TAB);

    const int tid = threadIdx.x;
    const int K = g.cin;
    const bf16* __restrict__ Xp = reinterpret_cast<const bf16*>(g.Xin);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    const long mtiles = (g.M + ROWS - 1) / ROWS;
    const int ksteps = g.Kp >> 4;
    const u16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8_t bw[KS];
    {
        const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + (((long)(tid >> 6) * ksteps) * 64 + (tid & 63)) * 8;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            if (i < ksteps) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + (long)i * 512);
            else
#pragma unroll
                for (int j = 0; j < 8; ++j) bw[i][j] = (__bf16)0.f;
        }
    }
    for (int i = tid; i < NKC * 128; i += 256) {
        const bool ok = i < K;
        float tsc = 0.f, tsh = 0.f;
        if (ok) {
            if (g.lf.isum != nullptr) lf_table(g.lf, i, blockIdx.x == 0, tsc, tsh);      // link-free: norm1's table from the producers' sums (bn_lf.h)
            else { tsc = g.sc[i]; tsh = g.sh[i]; }
        }
        tab[i] = tsc; tab[NKC * 128 + i] = tsh; tab[2 * NKC * 128 + i] = ok ? g.sl[i] : 0.f;
    }
    const int ocol = (tid >> 6) * 32 + (tid & 31);                         // this lane's output column in MFMA layout: wave*32 + (lane & 31)
    const float cbias = g.bias[ocol];
    const float o_sc = OACT ? g.osc[ocol] : 1.f, o_sh = OACT ? g.osh[ocol] : 0.f, o_sl = OACT ? g.osl[ocol] : 1.f;
    float st1[8], st2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[j] = 0.f; st2[j] = 0.f; }

    // requests: see bwd1x1_fused.hip -- a uniform base per 16-row group and chunk column plus one 32-bit lane offset for full tiles;
    // chunks beyond cin are never requested (their LDS slots are zeroed once and stay zero)
    const int d_r0 = (tid >> 6) * 4 + ((tid & 63) >> 4);
    const int d_chunk = (tid & 15) ^ (d_r0 & 15);
    const unsigned voffX = (unsigned)(d_r0 * (int)g.ldx * 2 + (d_chunk << 4));
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
        if (kc * 128 + (d_chunk << 3) >= K) {
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) *reinterpret_cast<u16x8*>(smem + kc * TILE + ((tid >> 6) + 4 * i) * 1024 + (tid & 63) * 16) = z8;
        }
    auto request = [&](long t) {
        const long m0 = t * ROWS;
        const int wave = tid >> 6, lane = tid & 63;
        if (m0 + ROWS <= g.M) {
            const char* bX = reinterpret_cast<const char*>(Xp) + m0 * g.ldx * 2;
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
                if (kc * 128 + (d_chunk << 3) < K) {
#pragma unroll
                    for (int i = 0; i < ROWS / 16; ++i)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bX + kc * 256 + (long)i * 32 * g.ldx + voffX),
                                                         (__attribute__((address_space(3))) void*)(smem + kc * TILE + (wave + 4 * i) * 1024), 16, 0, 0);
                }
        } else {                           // the launch's last, partial tile: rows beyond M come from the zero line
            const int rsub = lane >> 4, slot = lane & 15;
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
                if (kc * 128 + (d_chunk << 3) < K) {
#pragma unroll
                    for (int i = 0; i < ROWS / 16; ++i) {
                        const int rg = wave + 4 * i;
                        const long m = m0 + rg * 4 + rsub;
                        const char* src = m < g.M ? reinterpret_cast<const char*>(Xp + m * g.ldx + kc * 128 + (d_chunk << 3)) : zeros + (slot << 4);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(smem + kc * TILE + rg * 1024), 16, 0, 0);
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
        int t_o = tid;                                                     // opaque copy: keeps the phases' LDS addresses out of loop-invariant registers
        asm volatile("" : "+v"(t_o));
        const int c8 = t_o & 15, c_r0 = t_o >> 4;
        const int e_off = c_r0 * 256 + ((c8 ^ (c_r0 & 15)) << 4);
        // ---- x -> prelu(sc*x + sh, sl) in place (rows beyond M hold zeros and stay what they become: they are never stored)
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int col = kc * 128 + c8 * 8;
            if (col < K) {
                const float* tp = tab + col;
                const float4 s0 = *reinterpret_cast<const float4*>(tp), s1 = *reinterpret_cast<const float4*>(tp + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(tp + NKC * 128), h1 = *reinterpret_cast<const float4*>(tp + NKC * 128 + 4);
                const float4 l0 = *reinterpret_cast<const float4*>(tp + 2 * NKC * 128), l1 = *reinterpret_cast<const float4*>(tp + 2 * NKC * 128 + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                const float sl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                const int nrem = K - col;                                  // < 8 only in a last, partial chunk: foreign channels -> 0
#pragma unroll
                for (int i = 0; i < ROWS / 16; ++i) {
                    u16x8* p = reinterpret_cast<u16x8*>(smem + kc * TILE + e_off + i * 4096);
                    const u16x8 v = *p;
                    u16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = j < nrem ? f2bf(prelu(fmaf(bf2f(v[j]), sc[j], sh[j]), sl[j])) : (bf16)0;
                    *p = o;
                }
            }
        }
        __syncthreads();
        // ---- C[64][128] = A x W1^T over the K extent
        const int lane = t_o & 63, wave = t_o >> 6;
        const int r = lane & 31, h = lane >> 5;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        {
            const int a_base = r * 256, w4 = (h ^ (r & 15)) << 4;          // chunk (2*ks + h) ^ (r & 15) == (2*ks) ^ (h ^ (r & 15))
            auto afrag = [&](int s, int i) {                               // s = global k-step: chunk column s / 8, k-step s % 8 inside it
                return *reinterpret_cast<const bf16x8_t*>(smem + (s >> 3) * TILE + a_base + i * 8192 + (w4 ^ ((s & 7) << 5)));
            };
            bf16x8_t a0 = afrag(0, 0), a1 = afrag(0, 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s < ksteps) {
                    bf16x8_t b0 = a0, b1 = a1;
                    if (s + 1 < KS && s + 1 < ksteps) { b0 = afrag(s + 1, 0); b1 = afrag(s + 1, 1); }
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bw[s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[s], acc[1], 0, 0, 0);
                    a0 = b0; a1 = b1;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                                   // the x chunks are free: the next tile travels under the epilogue
        if (mt + gridDim.x < mtiles) request(mt + gridDim.x);
        {
            bf16* cw = Cs + 4 * h * CLD + wave * 32 + r;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][e] + cbias;
                    if (OACT) v = prelu(fmaf(v, o_sc, o_sh), o_sl);
                    cw[(i * 32 + (e & 3) + 8 * (e >> 2)) * CLD] = f2bf(v);
                }
        }
        __syncthreads();
        // ---- epilogue: statistics of the rounded values, 16-B stores (a tile of Y is 16 KB contiguous)
        {
            const bf16* crow = Cs + c_r0 * CLD + c8 * 8;
            bf16* yb = reinterpret_cast<bf16*>(g.Out) + (m0 + c_r0) * 128 + c8 * 8;
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) {
                if (m0 + c_r0 + 16 * i < g.M) {
                    const u16x8 o = *reinterpret_cast<const u16x8*>(crow + i * 16 * CLD);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = bf2f(o[j]);
                        st1[j] += x; st2[j] += x * x;
                    }
                    *reinterpret_cast<u16x8*>(yb + (long)i * 16 * 128) = o;
                }
            }
        }
        // (the next iteration's first barrier separates this read of Cs from the next tile's write)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g.part == nullptr && g.isum_out == nullptr) return;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, c8 = tid & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d1 = (double)st1[j], d2 = (double)st2[j];
        d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32);
        d2 += __shfl_xor(d2, 16); d2 += __shfl_xor(d2, 32);
        if (lane < 16) {
            double* p = red + ((wave * 128) + c8 * 8 + j) * 2;
            p[0] = d1; p[1] = d2;
        }
    }
    __syncthreads();
    if (tid < 128) {
        double a = 0, b = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 128 + tid) * 2]; b += red[(w * 128 + tid) * 2 + 1]; }
        if (g.isum_out != nullptr) lf_add(g.isum_out, g.isum_stride, tid, a, b);          // link-free: the consumer derives norm2's table itself (bn_lf.h)
        else { double* p = g.part + ((long)blockIdx.x * 128 + tid) * 2; p[0] = a; p[1] = b; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Wide layers (256 < K <= 512: the deep half of dense block 3, blocks 4-5): the K extent does not fit LDS next to a second workgroup, so its
// 128-channel chunks stream through a two-slot ring -- chunk kc + 1 (or the next tile's first chunk) is requested before chunk kc is
// activated and multiplied.  Same arithmetic and summation order as above (k ascending); all 32 weight fragments of a wave in registers.
// ---------------------------------------------------------------------------------------------------------------------------------
struct WideCfg {
    static constexpr int OFF_C = 2 * TILE;
    static constexpr int OFF_TAB = OFF_C + ROWS * CLD * 2;
    static constexpr int SMEM = OFF_TAB + 3 * 512 * 4;             // 56 320 B: two workgroups per CU
};

template <bool OACT>
__global__ __launch_bounds__(256, 2) void k_fwd1x1_wide_bf16(const Fwd1x1Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 32;
    bf16* Cs = reinterpret_cast<bf16*>(smem + WideCfg::OFF_C);
    double* red = reinterpret_cast<double*>(smem);                         // [4][128][2], after the last tile (over the chunk ring)
    float* tab = reinterpret_cast<float*>(smem + WideCfg::OFF_TAB);

    const int tid = threadIdx.x;
    const int K = g.cin, nkc = (K + 127) >> 7;
    const bf16* __restrict__ Xp = reinterpret_cast<const bf16*>(g.Xin);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    const long mtiles = (g.M + ROWS - 1) / ROWS;
    const int ksteps = g.Kp >> 4;

    // the table first (link-free: fp64 arithmetic on the producers' sums), THEN the 32 weight fragments: with the fragments' 128 destination
    // registers already live the table code spilled 21 more registers into the tile loop
    for (int i = tid; i < 512; i += 256) {
        const bool ok = i < K;
        float tsc = 0.f, tsh = 0.f;
        if (ok) {
            if (g.lf.isum != nullptr) lf_table(g.lf, i, blockIdx.x == 0, tsc, tsh);      // link-free: norm1's table from the producers' sums (bn_lf.h)
            else { tsc = g.sc[i]; tsh = g.sh[i]; }
        }
        tab[i] = tsc; tab[512 + i] = tsh; tab[1024 + i] = ok ? g.sl[i] : 0.f;
    }
    bf16x8_t bw[KS];
    {
        const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + (((long)(tid >> 6) * ksteps) * 64 + (tid & 63)) * 8;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            if (i < ksteps) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + (long)i * 512);
            else
#pragma unroll
                for (int j = 0; j < 8; ++j) bw[i][j] = (__bf16)0.f;
        }
    }
    const int ocol = (tid >> 6) * 32 + (tid & 31);
    const float cbias = g.bias[ocol];
    const float o_sc = OACT ? g.osc[ocol] : 1.f, o_sh = OACT ? g.osh[ocol] : 0.f, o_sl = OACT ? g.osl[ocol] : 1.f;
    float st1[8], st2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[j] = 0.f; st2[j] = 0.f; }

    const int d_r0 = (tid >> 6) * 4 + ((tid & 63) >> 4);
    const int d_chunk = (tid & 15) ^ (d_r0 & 15);
    const unsigned voffX = (unsigned)(d_r0 * (int)g.ldx * 2 + (d_chunk << 4));
    // chunk kc of row tile t -> ring slot; lanes whose 16-B piece lies beyond the K extent or beyond the last row fetch the zero line
    auto request = [&](long t, int kc, int slot) {
        const long m0 = t * ROWS;
        const int wave = tid >> 6;
        const char* bX = reinterpret_cast<const char*>(Xp) + m0 * g.ldx * 2 + kc * 256;
        const bool colok = kc * 128 + (d_chunk << 3) < K;
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) {
            const bool ok = colok && m0 + d_r0 + 16 * i < g.M;
            const char* src = ok ? bX + (long)i * 32 * g.ldx + voffX : zeros + ((tid & 15) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + slot * TILE + (wave + 4 * i) * 1024), 16, 0, 0);
        }
    };

    long mt = blockIdx.x;
    int slot = 0;
    if (mt < mtiles) request(mt, 0, 0);
    for (; mt < mtiles; mt += gridDim.x) {
        const long m0 = mt * ROWS;
        int t_o = tid;
        asm volatile("" : "+v"(t_o));
        const int c8 = t_o & 15, c_r0 = t_o >> 4;
        const int e_off = c_r0 * 256 + ((c8 ^ (c_r0 & 15)) << 4);
        const int lane = t_o & 63, wave = t_o >> 6;
        const int r = lane & 31, h = lane >> 5;
        const int a_base = r * 256, w4 = (h ^ (r & 15)) << 4;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            if (kc < nkc) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                           // chunk kc has landed; the other slot's readers (previous chunk's MFMAs) are done
                if (kc + 1 < nkc) request(mt, kc + 1, slot ^ 1);
                else if (mt + gridDim.x < mtiles) request(mt + gridDim.x, 0, slot ^ 1);
                // ---- activate chunk kc in place
                const int col = kc * 128 + c8 * 8;
                if (col < K) {
                    const float* tp = tab + col;
                    const float4 s0 = *reinterpret_cast<const float4*>(tp), s1 = *reinterpret_cast<const float4*>(tp + 4);
                    const float4 h0 = *reinterpret_cast<const float4*>(tp + 512), h1 = *reinterpret_cast<const float4*>(tp + 516);
                    const float4 l0 = *reinterpret_cast<const float4*>(tp + 1024), l1 = *reinterpret_cast<const float4*>(tp + 1028);
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    const float sl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                    const int nrem = K - col;
#pragma unroll
                    for (int i = 0; i < ROWS / 16; ++i) {
                        u16x8* p = reinterpret_cast<u16x8*>(smem + slot * TILE + e_off + i * 4096);
                        const u16x8 v = *p;
                        u16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = j < nrem ? f2bf(prelu(fmaf(bf2f(v[j]), sc[j], sh[j]), sl[j])) : (bf16)0;
                        *p = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // bare barrier: a __syncthreads() would drain the request just issued
                // ---- acc += A(chunk kc) x W1^T(k-steps 8*kc ..)
                auto afrag = [&](int ks, int i) {
                    return *reinterpret_cast<const bf16x8_t*>(smem + slot * TILE + a_base + i * 8192 + (w4 ^ (ks << 5)));
                };
                bf16x8_t a0 = afrag(0, 0), a1 = afrag(0, 1);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (kc * 8 + ks < ksteps) {
                        bf16x8_t b0 = a0, b1 = a1;
                        if (ks + 1 < 8 && kc * 8 + ks + 1 < ksteps) { b0 = afrag(ks + 1, 0); b1 = afrag(ks + 1, 1); }
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bw[kc * 8 + ks], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[kc * 8 + ks], acc[1], 0, 0, 0);
                        a0 = b0; a1 = b1;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                slot ^= 1;
            }
        }
        {
            bf16* cw = Cs + 4 * h * CLD + wave * 32 + r;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][e] + cbias;
                    if (OACT) v = prelu(fmaf(v, o_sc, o_sh), o_sl);
                    cw[(i * 32 + (e & 3) + 8 * (e >> 2)) * CLD] = f2bf(v);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const bf16* crow = Cs + c_r0 * CLD + c8 * 8;
            bf16* yb = reinterpret_cast<bf16*>(g.Out) + (m0 + c_r0) * 128 + c8 * 8;
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) {
                if (m0 + c_r0 + 16 * i < g.M) {
                    const u16x8 o = *reinterpret_cast<const u16x8*>(crow + i * 16 * CLD);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = bf2f(o[j]);
                        st1[j] += x; st2[j] += x * x;
                    }
                    *reinterpret_cast<u16x8*>(yb + (long)i * 16 * 128) = o;
                }
            }
        }
        // (the next tile's C tile is written three barriers from here)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (g.part == nullptr && g.isum_out == nullptr) return;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, c8 = tid & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d1 = (double)st1[j], d2 = (double)st2[j];
        d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32);
        d2 += __shfl_xor(d2, 16); d2 += __shfl_xor(d2, 32);
        if (lane < 16) {
            double* p = red + ((wave * 128) + c8 * 8 + j) * 2;
            p[0] = d1; p[1] = d2;
        }
    }
    __syncthreads();
    if (tid < 128) {
        double a = 0, b = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 128 + tid) * 2]; b += red[(w * 128 + tid) * 2 + 1]; }
        if (g.isum_out != nullptr) lf_add(g.isum_out, g.isum_stride, tid, a, b);          // link-free: the consumer derives norm2's table itself (bn_lf.h)
        else { double* p = g.part + ((long)blockIdx.x * 128 + tid) * 2; p[0] = a; p[1] = b; }
    }
}

}  // namespace

bool fwd1x1_fused_ok(const Fwd1x1Args& a) {
    if (!a.Xin || !a.Out || !a.Wfrag || !a.zeros || !a.bias || !a.sl) return false;
    if (a.lf.isum == nullptr && (!a.sc || !a.sh)) return false;
    if (a.lf.isum != nullptr && (!a.lf.bstat || !a.lf.gamma || !a.lf.beta || !a.lf.sc_out || !a.lf.sh_out || a.lf.count <= 0)) return false;
    if (a.isum_out != nullptr && a.osc != nullptr) return false;
    if (a.osc != nullptr && (a.part != nullptr || !a.osh || !a.osl)) return false;      // the output activation is the eval-mode epilogue: no statistics
    if (a.cin <= 0 || a.cin > 512 || a.Kp < a.cin || a.Kp > 512 || (a.Kp & 15) || (a.ldx & 7) || a.ldx < a.cin) return false;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.Xin) | reinterpret_cast<uintptr_t>(a.Out) | reinterpret_cast<uintptr_t>(a.Wfrag);
    return (al & 15) == 0;
}

int fwd1x1_fused_nblk(const Fwd1x1Args& a) {
    const int cap = a.Kp <= 256 ? 768 : 512;         // resident workgroups: three per CU (two for the wide-layer kernel)
    const long mt = (a.M + ROWS - 1) / ROWS;
    return (int)(mt < cap ? mt : cap);
}

int fwd1x1_fused(const Fwd1x1Args& a, hipStream_t st) {
    if (a.M <= 0) return 0;
    if (!fwd1x1_fused_ok(a)) return -2;
    if (a.part != nullptr && a.isum_out == nullptr && a.nblk != fwd1x1_fused_nblk(a)) { fprintf(stderr, "tcvn: fwd1x1_fused nblk mismatch\n"); return -3; }
    static bool attr = false;
    if (!attr) {
        const void* fns[4] = {reinterpret_cast<const void*>(k_fwd1x1_fused_bf16<1, false>), reinterpret_cast<const void*>(k_fwd1x1_fused_bf16<2, false>),
                              reinterpret_cast<const void*>(k_fwd1x1_fused_bf16<1, true>), reinterpret_cast<const void*>(k_fwd1x1_fused_bf16<2, true>)};
        for (const void* f : fns) TCVN_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd1x1_wide_bf16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd1x1_wide_bf16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // SURVEY 8(d) strict bytes: x read once, Y written once
    ProfScope ps("k_fwd1x1_fused_bf16", 2.0 * a.M * 128.0 * a.cin, (double)a.M * 2.0 * (a.cin + 128.0), st);
    const int nblk = fwd1x1_fused_nblk(a);
    const bool oact = a.osc != nullptr;
    if (a.Kp > 256 && oact) hipLaunchKernelGGL(k_fwd1x1_wide_bf16<true>, dim3(nblk), dim3(256), WideCfg::SMEM, st, a);
    else if (a.Kp > 256) hipLaunchKernelGGL(k_fwd1x1_wide_bf16<false>, dim3(nblk), dim3(256), WideCfg::SMEM, st, a);
    else if (a.Kp <= 128 && oact) hipLaunchKernelGGL((k_fwd1x1_fused_bf16<1, true>), dim3(nblk), dim3(256), FwdCfg<1>::SMEM, st, a);
    else if (a.Kp <= 128) hipLaunchKernelGGL((k_fwd1x1_fused_bf16<1, false>), dim3(nblk), dim3(256), FwdCfg<1>::SMEM, st, a);
    else if (oact) hipLaunchKernelGGL((k_fwd1x1_fused_bf16<2, true>), dim3(nblk), dim3(256), FwdCfg<2>::SMEM, st, a);
    else hipLaunchKernelGGL((k_fwd1x1_fused_bf16<2, false>), dim3(nblk), dim3(256), FwdCfg<2>::SMEM, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
