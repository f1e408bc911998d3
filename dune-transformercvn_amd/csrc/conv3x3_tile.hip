// bf16 3x3 convolutions on padded LDS tiles (see tile3x3.h): forward, data gradient, weight gradient.
// Reference call site: Bottleneck.output_block (transformercvn/network/layers/dense_net.py:29-40) and its autograd.
//
// Forward: one workgroup = 128 padded output positions x 32 output channels; the BatchNorm+PReLU-transformed bf16 input
// image (128 + 2*(W+3) rows x 128 channels) is staged ONCE in LDS, then 9 taps x 8 k-steps of v_mfma_f32_32x32x16_bf16
// read it with row offsets (ds_read_b128, XOR-swizzled, conflict free); weights stream from L2 in fragment order.
// Algorithmic work per launch: 2 * pixels * 32 * 1152 FLOP; HBM: read 128 ch + write 32 ch per pixel.
#include <cstdlib>
#include <type_traits>
#include "tile3x3.h"
#include "prof.h"

namespace tcvn {

using namespace t3;

namespace {


__device__ __forceinline__ int fdiv(int a, int d, float inv, int& rem) {     // a in [0, 2^24)
    int q = (int)((float)a * inv);
    rem = a - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
}
// pixel index of padded position g (or -1)
__device__ __forceinline__ int pix_of(const PadGeom& q, int g, float invWp, float invHp) {
    if (g < 0 || g >= (int)q.gtot) return -1;
    int wp, hp;
    const int row = fdiv(g, q.Wp, invWp, wp);
    const int img = fdiv(row, q.Hp, invHp, hp);
    if (hp < 1 || hp > q.H || wp < 1 || wp > q.W) return -1;
    return (img * q.H + (hp - 1)) * q.W + (wp - 1);
}

// In-LDS activation of one landed 16-B chunk (ConvFwdArgs::act_fused): the same arithmetic as k_act_bf16 (fp32 fma, PReLU, one
// rounding to bf16), so a fused launch and a materialised one stage bit-identical images.
struct Act8 { float sc[8], sh[8], sl[8]; };
__device__ __forceinline__ Act8 act8_load(const float* __restrict__ xtab, int cc) {      // xtab: [3][128] floats in LDS
    Act8 t;
    const float4 a0 = *reinterpret_cast<const float4*>(xtab + cc * 8), a1 = *reinterpret_cast<const float4*>(xtab + cc * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(xtab + 128 + cc * 8), b1 = *reinterpret_cast<const float4*>(xtab + 128 + cc * 8 + 4);
    const float4 d0 = *reinterpret_cast<const float4*>(xtab + 256 + cc * 8), d1 = *reinterpret_cast<const float4*>(xtab + 256 + cc * 8 + 4);
    t.sc[0] = a0.x; t.sc[1] = a0.y; t.sc[2] = a0.z; t.sc[3] = a0.w; t.sc[4] = a1.x; t.sc[5] = a1.y; t.sc[6] = a1.z; t.sc[7] = a1.w;
    t.sh[0] = b0.x; t.sh[1] = b0.y; t.sh[2] = b0.z; t.sh[3] = b0.w; t.sh[4] = b1.x; t.sh[5] = b1.y; t.sh[6] = b1.z; t.sh[7] = b1.w;
    t.sl[0] = d0.x; t.sl[1] = d0.y; t.sl[2] = d0.z; t.sl[3] = d0.w; t.sl[4] = d1.x; t.sl[5] = d1.y; t.sl[6] = d1.z; t.sl[7] = d1.w;
    return t;
}
__device__ __forceinline__ u16x8 act8_apply(const u16x8 v, const Act8& t) {
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(prelu(fmaf(bf2f(v[j]), t.sc[j], t.sh[j]), t.sl[j]));
    return o;
}
__device__ __forceinline__ void act_tab_fill(float* __restrict__ xtab, const float* __restrict__ sc, const float* __restrict__ sh,
                                             const float* __restrict__ sl, int tid, int nthreads) {
    for (int i = tid; i < 128; i += nthreads) { xtab[i] = sc[i]; xtab[128 + i] = sh[i]; xtab[256 + i] = sl[i]; }
}

// LDS-DMA one padded image (rows [g_first, g_first + nrows4)) of a pre-activated [pixels,128] bf16 tensor into `buf`:
// every wave-instruction writes 1 KiB = 4 image rows, lane -> (row = lane>>4, slot = lane&15); the XOR swizzle is applied
// on the SOURCE chunk (slot s of row r holds channel chunk s ^ (r & 15)), padding rows come from a page of zeros.
__device__ __forceinline__ void dma_image(char* smem_base, int buf_off, const bf16* __restrict__ XA, const char* __restrict__ zeros,
                                          const int* rowpix, int nrows4, int wave, int lane) {
    const int rsub = lane >> 4, slot = lane & 15;
    // all table entries first, then all DMA instructions: one table read per DMA was an exposed LDS round trip each (4 500 of the weight
    // gradient's 17 400 cycles per wave and tile in the phase counters).  Up to DMA_RG row groups per wave = 384 image rows.
    constexpr int DMA_RG = 24;
    int mrow[DMA_RG];
#pragma unroll
    for (int i = 0; i < DMA_RG; ++i) {
        const int rg = wave + 4 * i;
        mrow[i] = rg * 4 < nrows4 ? rowpix[rg * 4 + rsub] : -1;
    }
#pragma unroll
    for (int i = 0; i < DMA_RG; ++i) {
        const int rg = wave + 4 * i, r = rg * 4 + rsub;
        if (rg * 4 < nrows4) {
            const int m = mrow[i];                     // pixel index of this image row or -1 (table filled a tile ahead)
            const char* src = m >= 0 ? reinterpret_cast<const char*>(XA + (long)m * 128) + ((slot ^ (r & 15)) << 4)
                                     : zeros + (slot << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem_base + buf_off + rg * 1024), 16, 0, 0);
        }
    }
}

// One workgroup per CU (persistent): weights live in registers for the whole launch (72 fragments = 288 VGPRs), two LDS
// images double-buffer the LDS-DMA of tile t+1 under the 72 MFMAs + epilogue of tile t.
__global__ __launch_bounds__(256, 1) void k_conv3x3_fwd_bf16(const ConvFwdArgs g, int n_img, int ntiles, int swz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PadGeom q(n_img, g.H, g.W);
    const int nrows4 = (q.rows() + 3) & ~3;
    const int img_bytes = nrows4 * 256;      // images at byte offsets 0 and img_bytes (kept as offsets: LDS address space)
    int* tbl = reinterpret_cast<int*>(smem + 2 * nrows4 * 256);               // [3][nrows4] pixel index per image row
    double* red = reinterpret_cast<double*>(smem + 2 * nrows4 * 256 + 3 * nrows4 * 4);   // [4][32][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const bf16* __restrict__ YA = reinterpret_cast<const bf16*>(g.Aact);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + lane * 8;   // fragment order: 1 KiB per wave load
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(g.Out);
    const int nb = gridDim.x;
    const int lb = swz ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;   // neighbours share an XCD's L2
    const bool nok = r < g.N;
    const float bias = nok ? g.bias[r] : 0.f;
    const bool drop = g.drop_p > 0.f;
    const uint32_t dkey = drop_key(g.seed, g.stream_id);

    bf16x8_t bw[72];
#pragma unroll
    for (int i = 0; i < 72; ++i) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + i * 512);

    double s1 = 0, s2 = 0;
    auto fill_tbl = [&](int slot, int tile) {
        for (int rr = tid; rr < nrows4; rr += 256) tbl[slot * nrows4 + rr] = pix_of(q, tile * TP - q.halo + rr, invWp, invHp);
    };
    if (lb < ntiles) fill_tbl(0, lb);
    if (lb + nb < ntiles) fill_tbl(1, lb + nb);
    __syncthreads();
    if (lb < ntiles) dma_image(smem, 0, YA, zeros, tbl, nrows4, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0, ts = 0;                                         // image buffer / table slot of the current tile
    for (int t = lb; t < ntiles; t += nb, cur ^= 1, ts = ts == 2 ? 0 : ts + 1) {
        const int tn = ts == 2 ? 0 : ts + 1, tnn = tn == 2 ? 0 : tn + 1;
        if (t + nb < ntiles && !TCVN_DBG_BIT(g.dbg, 1))                     // prefetch the next tile's image under this tile's MFMAs
            dma_image(smem, (cur ^ 1) * img_bytes, YA, zeros, tbl + tn * nrows4, nrows4, wave, lane);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        asm volatile("s_nop 4" : "+a"(acc));          // accvgpr writes -> first MFMA (inside asm) needs its wait states
        const int lrow0 = wave * 32 + r + q.halo;
        const int image = cur * img_bytes;
        // A fragments of tap+1 are read from LDS while the 8 MFMAs of tap run; weights are consumed straight from AGPRs
        bf16x8_t af[2][8];
        {
            const int lr = lrow0 - q.Wp - 1;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                af[0][ks] = *reinterpret_cast<const bf16x8_t*>(smem + image + lr * 256 + (((2 * ks + h) ^ (lr & 15)) << 4));
        }
        if (!TCVN_DBG_BIT(g.dbg, 2))
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) {
                const int lr = lrow0 + ((tap + 1) / 3 - 1) * q.Wp + ((tap + 1) % 3 - 1);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    af[(tap + 1) & 1][ks] =
                        *reinterpret_cast<const bf16x8_t*>(smem + image + lr * 256 + (((2 * ks + h) ^ (lr & 15)) << 4));
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                // taps 0-6 keep their weights in AGPRs (16 acc + 224), taps 7-8 in arch VGPRs.  hipcc's hazard recognizer does
                // not see inside asm: the leading s_nop 1 covers (a) the two wait states gfx950 needs between a VALU write of an
                // operand register (the allocator's v_accvgpr_read/v_mov copies land right in front of a statement) and the MFMA
                // reading it, and (b) the wait state between back-to-back MFMAs chained through the accumulator.  Without it a
                // wave occasionally (~1e-4 of tiles) computed a whole tile with one stale operand dword.  The nops are free:
                // they sit in the shadow of the previous MFMA's 8 passes.
                if (tap < 7) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(af[tap & 1][ks]), "a"(bw[tap * 8 + ks]));
                else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(af[tap & 1][ks]), "v"(bw[tap * 8 + ks]));
            }
        }
        // the MFMAs sit inside asm statements: hipcc pads no hazard for them -- wait out the last MFMA's result latency
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc));
        // epilogue: bias, dropout (one Philox call per 4 consecutive pixels of a channel), store, statistics
        long cur_grp = -1;
        uint32_t bits = 0;
        const int* px = tbl + ts * nrows4 + q.halo;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int lp = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int m = px[lp];
            if (m >= 0 && nok && !TCVN_DBG_BIT(g.dbg, 4)) {
                float v = acc[e] + bias;
                if (drop) {
                    if ((m >> 1) != cur_grp) { cur_grp = m >> 1; bits = drop_bits(dkey, m, r, g.N); }
                    v *= drop_pick(bits, m, g.drop_p);
                }
                const bf16 o = f2bf(v);
                Out[(long)m * g.ldo + g.n_off + r] = o;
                const double x = (double)bf2f(o);
                s1 += x; s2 += x * x;
            }
        }
        if (t + 2 * nb < ntiles) fill_tbl(tnn, t + 2 * nb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (g.part != nullptr) {
        double a = s1, b = s2;
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
        if (lane < 32) { red[(wave * 32 + lane) * 2] = a; red[(wave * 32 + lane) * 2 + 1] = b; }
        __syncthreads();
        if (tid < g.N) {
            double x = 0, y = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { x += red[(w * 32 + tid) * 2]; y += red[(w * 32 + tid) * 2 + 1]; }
            g.part[((long)blockIdx.x * g.N + tid) * 2] = x;
            g.part[((long)blockIdx.x * g.N + tid) * 2 + 1] = y;
        }
    }
}

// Ring variant: a workgroup walks CONSECUTIVE tiles, so tile t+1's image shares its first rows() - 128 rows with tile t's.
// The LDS image is a ring of 512 rows (padded position g -> row (g - g_org) & 511, g_org = first row of the workgroup's first
// tile): per tile only the 128 new rows are fetched (32 KB instead of 128 + 2*(W+3) rows = 70 KB at W = 69).  Measured on the
// strip kernel above: the LDS-DMA fill alone cost 173 of the 335 us of a block-1 launch.  Needs rows() + 128 <= 512.
constexpr int RING = 512;
__global__ __launch_bounds__(256, 1) void k_conv3x3_fwd_ring_bf16(const ConvFwdArgs g, int n_img, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PadGeom q(n_img, g.H, g.W);
    const int nrows4 = (q.rows() + 3) & ~3;
    int* tbl = reinterpret_cast<int*>(smem + RING * 256);                   // [RING] pixel index per ring row
    double* red = reinterpret_cast<double*>(smem + RING * 256 + RING * 4);  // [4][32][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const bf16* __restrict__ YA = reinterpret_cast<const bf16*>(g.Aact);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + lane * 8;
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(g.Out);
    const int nb = gridDim.x, base = ntiles / nb, rem = ntiles % nb;
    const int t0 = blockIdx.x * base + min((int)blockIdx.x, rem), t1 = t0 + base + ((int)blockIdx.x < rem ? 1 : 0);
    const int g_org = t0 * TP - q.halo;
    const bool nok = r < g.N;
    const float bias = nok ? g.bias[r] : 0.f;
    const bool drop = g.drop_p > 0.f;
    const uint32_t dkey = drop_key(g.seed, g.stream_id);

    bf16x8_t bw[72];
#pragma unroll
    for (int i = 0; i < 72; ++i) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + i * 512);

    double s1 = 0, s2 = 0;
    // rows [row0, row0 + n) of this workgroup's row space (row 0 = g_org); ring slot = row & 511
    auto fill_rows = [&](int row0, int n) {
        for (int i = tid; i < n; i += 256) tbl[(row0 + i) & (RING - 1)] = pix_of(q, g_org + row0 + i, invWp, invHp);
    };
    auto dma_rows = [&](int row0, int n) {                       // n multiple of 4; 4 rows (1 KiB) per wave instruction
        const int rsub = lane >> 4, slot = lane & 15;
        for (int rg = wave; rg * 4 < n; rg += 4) {
            const int ring_row = (row0 + rg * 4) & (RING - 1);   // row0 and RING are multiples of 4: a group never wraps
            const int rr = ring_row + rsub;
            const int m = tbl[rr];
            const char* src = m >= 0 ? reinterpret_cast<const char*>(YA + (long)m * 128) + ((slot ^ (rr & 15)) << 4) : zeros + (slot << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + ring_row * 256), 16, 0, 0);
        }
    };
    if (t0 < t1) fill_rows(0, nrows4);
    if (t0 + 1 < t1) fill_rows(nrows4, TP);
    __syncthreads();
    if (t0 < t1) dma_rows(0, nrows4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int k128 = (t - t0) * TP;                          // row of this tile's first image row
        if (t + 1 < t1 && !TCVN_DBG_BIT(g.dbg, 1)) dma_rows(nrows4 + k128, TP);     // the next tile's 128 new rows, under this tile's MFMAs
        // two accumulator chains (even / odd k-steps): a dependent MFMA chain issues one MFMA per ~54 cycles, two independent
        // chains keep the matrix pipe at its 32-cycle cadence
        f32x16 acc, acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
        asm volatile("s_nop 4" : "+a"(acc), "+a"(acc2));   // accvgpr writes -> first MFMA (inside asm) needs its wait states
        const int lrow0 = k128 + wave * 32 + r + q.halo;
        bf16x8_t af[2][8];
        {
            const int lr = (lrow0 - q.Wp - 1) & (RING - 1);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                af[0][ks] = *reinterpret_cast<const bf16x8_t*>(smem + lr * 256 + (((2 * ks + h) ^ (lr & 15)) << 4));
        }
        if (!TCVN_DBG_BIT(g.dbg, 2))
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) {
                const int lr = (lrow0 + ((tap + 1) / 3 - 1) * q.Wp + ((tap + 1) % 3 - 1)) & (RING - 1);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    af[(tap + 1) & 1][ks] = *reinterpret_cast<const bf16x8_t*>(smem + lr * 256 + (((2 * ks + h) ^ (lr & 15)) << 4));
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {      // hazard padding: see k_conv3x3_fwd_bf16
                if (ks & 1) {
                    if (tap < 7) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc2) : "v"(af[tap & 1][ks]), "a"(bw[tap * 8 + ks]));
                    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc2) : "v"(af[tap & 1][ks]), "v"(bw[tap * 8 + ks]));
                } else {
                    if (tap < 7) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(af[tap & 1][ks]), "a"(bw[tap * 8 + ks]));
                    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(af[tap & 1][ks]), "v"(bw[tap * 8 + ks]));
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc), "+a"(acc2));
        long cur_grp = -1;
        uint32_t bits = 0;
        float f1 = 0.f, f2 = 0.f;                       // this tile's 16 values per lane in fp32, folded into fp64 once per tile
        int mrow[16];                                    // all 16 table reads in one batch (one LDS wait, not one per element)
#pragma unroll
        for (int e = 0; e < 16; ++e) mrow[e] = tbl[(k128 + q.halo + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) & (RING - 1)];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = mrow[e];
            if (m >= 0 && nok && !TCVN_DBG_BIT(g.dbg, 4)) {
                float v = acc[e] + acc2[e] + bias;
                if (drop) {
                    if ((m >> 1) != cur_grp) { cur_grp = m >> 1; bits = drop_bits(dkey, m, r, g.N); }
                    v *= drop_pick(bits, m, g.drop_p);
                }
                const bf16 o = f2bf(v);
                Out[(long)m * g.ldo + g.n_off + r] = o;
                const float x = bf2f(o);
                f1 += x; f2 = fmaf(x, x, f2);
            }
        }
        s1 += (double)f1; s2 += (double)f2;
        if (t + 2 < t1) fill_rows(nrows4 + k128 + TP, TP);        // table of the rows the next iteration will fetch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (g.part != nullptr) {
        double a = s1, b = s2;
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
        if (lane < 32) { red[(wave * 32 + lane) * 2] = a; red[(wave * 32 + lane) * 2 + 1] = b; }
        __syncthreads();
        if (tid < g.N) {
            double x = 0, y = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { x += red[(w * 32 + tid) * 2]; y += red[(w * 32 + tid) * 2 + 1]; }
            g.part[((long)blockIdx.x * g.N + tid) * 2] = x;
            g.part[((long)blockIdx.x * g.N + tid) * 2 + 1] = y;
        }
    }
}
size_t fwd_ring_smem() { return RING * 256 + RING * 4 + 4 * 32 * 16; }

// Pair variant of the ring kernel: 512 threads = two waves per SIMD.  The ablation of the ring kernel (tools/ablate_conv3x3.py: block 1,
// 277 us = 76 fixed + 47 DMA issue + 86 MFMA + 68 epilogue, nothing overlapping) says a single wave per SIMD serialises its phases; here the
// two waves of a SIMD split the NINE TAPS of the same 32 positions -- wave w (role A) owns taps 0-4 (40 weight fragments = 160 registers),
// wave w+4 (role B) taps 5-8 (32 fragments) -- which is what fits the 256 registers a wave has at two waves per SIMD, and they run
// skewed by half a tile: after the tile barrier, B first finishes tile t-1 (adds A's partial sums, exchanged through LDS, applies bias /
// dropout, stores, statistics) while A already multiplies tile t; then B multiplies its taps of tile t.  A also fills the pixel table.
// The matrix pipe of the SIMD sees the same 72 MFMAs per tile, but the epilogue, the table arithmetic, the LDS waits and the DMA issue of
// one wave now sit under the other wave's MFMAs.  Ring rows are sized from the map width (rows() + 128, multiple of 16) so that the LDS
// also holds the two exchange buffers: 100 KB + 32 KB at W = 69.
// phase counters of the pair kernel (validation build only): wave-cycles per phase, reported by every 16th workgroup
#ifdef TCVN_DEBUG_KNOBS
__device__ unsigned long long g_pair_ph[16];
__device__ unsigned long long g_wg_ph[16];                   // the same for k_conv3x3_wgrad_bf16
#define PAIR_T0() unsigned long long ph_t = clock64()
#define PAIR_PH(i) do { const unsigned long long n_ = clock64(); ph[i] += n_ - ph_t; ph_t = n_; } while (0)
#else
#define PAIR_T0() do {} while (0)
#define PAIR_PH(i) do {} while (0)
#endif
constexpr int PAIR_TBL = 1024;       // table entries (power of two > 512 + halo: an entry lives two tiles longer than its image row)
__global__ __launch_bounds__(512, 2) void k_conv3x3_fwd_pair_bf16(const ConvFwdArgs g, int n_img, int ntiles, int ring) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PadGeom q(n_img, g.H, g.W);
    const int nrows4 = (q.rows() + 3) & ~3;
    int* tbl = reinterpret_cast<int*>(smem + ring * 256);                             // [1024] pixel index of row (row-space index & 1023): an entry
                                                                                      // outlives its image row (the deferred epilogue reads it a tile later)
    float* xchg = reinterpret_cast<float*>(smem + ring * 256 + PAIR_TBL * 4);         // [2][4 pairs][16][64] role A's partial sums
    bf16* ctile = reinterpret_cast<bf16*>(xchg + 2 * 4 * 16 * 64);                    // [4 pairs][32][32] bf16 output tiles of the epilogue
    double* red = reinterpret_cast<double*>(ctile + 4 * 32 * 32);                     // [4][32][2]
    float* xtab = reinterpret_cast<float*>(red + 4 * 32 * 2);                         // act_fused: [3][128] scale, shift, slope of the input BatchNorm + PReLU

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool roleB = wave >= 4;
    const bool xf = g.act_fused != 0;
    const int pw = wave & 3;                                                          // pair index = 32-position block of the tile
    const int r = lane & 31, h = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const bf16* __restrict__ YA = reinterpret_cast<const bf16*>(g.Aact);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(g.Out);
    const int nb = gridDim.x, base = ntiles / nb, rem = ntiles % nb;
    const int t0 = blockIdx.x * base + min((int)blockIdx.x, rem), t1 = t0 + base + ((int)blockIdx.x < rem ? 1 : 0);
    const int g_org = t0 * TP - q.halo;
    const bool nok = r < g.N;
    const float bias = nok ? g.bias[r] : 0.f;
    const bool drop = g.drop_p > 0.f;
    const uint32_t dkey = drop_key(g.seed, g.stream_id);
    // all 32 channels present and the output slice 16-B aligned: the tile leaves in 16-B stores through LDS
    const bool vec_store = g.N == 32 && (g.n_off & 7) == 0 && (g.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(g.Out) & 15) == 0;
    auto wrap = [&](int x) { return x >= ring ? x - ring : x; };                      // x in [0, 2 * ring)

    // weight fragment (tap, ks) sits at ((tap*8 + ks)*64 + lane)*8; role A keeps taps 0..4 in registers, role B taps 5..8
    const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + lane * 8;

    // rows [row0, row0 + n) of this workgroup's row space (row 0 = g_org); image row -> ring slot (row mod ring), table entry row & 1023
    auto fill_rows = [&](int row0, int n, int t, int nt) {                            // by threads [t, t + nt)
        for (int i = t; i < n; i += nt) tbl[(row0 + i) & (PAIR_TBL - 1)] = pix_of(q, g_org + row0 + i, invWp, invHp);
    };
    auto dma_rows = [&](int row0, int slot0, int n, int w0, int nw) {                 // n multiple of 4; 4 rows (1 KiB) per wave instruction, issued by
        const int rsub = lane >> 4, slot = lane & 15;                                 // wave w0 of nw
        // four instructions per trip with their table reads batched in front (one dependent LDS read per instruction: 1 200 cycles per
        // wave and tile in the phase counters, 750-900 batched)
        for (int rg0 = w0; rg0 * 4 < n; rg0 += 4 * nw) {
            int m4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m4[j] = tbl[(row0 + (rg0 + nw * j) * 4 + rsub) & (PAIR_TBL - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rg = rg0 + nw * j;
                if (rg * 4 < n) {
                    const int ring_row = wrap(slot0 + rg * 4);                        // slot0, ring multiples of 4: a group never wraps
                    const int rr = ring_row + rsub;
                    const char* src = m4[j] >= 0 ? reinterpret_cast<const char*>(YA + (long)m4[j] * 128) + ((slot ^ (rr & 15)) << 4) : zeros + (slot << 4);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(smem + ring_row * 256), 16, 0, 0);
                }
            }
        }
    };
    // act_fused: the rows arrive RAW; the wave that requested a row group activates it in place once its own DMAs have landed (vmcnt(0) at the
    // top of the tile loop) and before the tile barrier that publishes the rows -- same (w0, nw) assignment as dma_rows.  The rows of the NEXT
    // tile are disjoint from every row the current tile's taps read, so no other wave touches them meanwhile.  A lane keeps one logical
    // 8-channel chunk (cc) for all rows: its 24 table values are read from LDS once per call.
    auto xform_rows = [&](int row0, int slot0, int n, int w0, int nw) {
        const int rsub = lane >> 4, cc = lane & 15;
        for (int rg0 = w0; rg0 * 4 < n; rg0 += 4 * nw) {
            int m4[4];
            u16x8 v[4];
            char* p[4];
            // every LDS read of the trip -- table entries, image chunks, the 24 table values -- is requested before the first use
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rg = rg0 + nw * j;
                const bool in = rg * 4 < n;
                m4[j] = tbl[(row0 + (in ? rg * 4 : 0) + rsub) & (PAIR_TBL - 1)];
                if (!in) m4[j] = -1;
                const int rr = wrap(slot0 + (in ? rg * 4 : 0)) + rsub;
                p[j] = smem + rr * 256 + ((cc ^ (rr & 15)) << 4);
                v[j] = *reinterpret_cast<const u16x8*>(p[j]);
            }
            const Act8 tb = act8_load(xtab, cc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (m4[j] >= 0) *reinterpret_cast<u16x8*>(p[j]) = act8_apply(v[j], tb);      // padding rows stay the zeros the DMA wrote
        }
    };
    if (t0 < t1) fill_rows(0, nrows4, tid, 512);
    if (t0 + 1 < t1) fill_rows(nrows4, TP, tid, 512);
    if (xf && g.lf.isum != nullptr) {                                                  // link-free (round 5): norm2's table from the 1x1 kernel's sums (bn_lf.h)
        if (tid < 128) {
            float tsc, tsh;
            lf_table(g.lf, tid, blockIdx.x == 0, tsc, tsh);
            xtab[tid] = tsc; xtab[128 + tid] = tsh; xtab[256 + tid] = g.sl[tid];
        }
    } else if (xf) act_tab_fill(xtab, g.sc, g.sh, g.sl, tid, 512);
    __syncthreads();
    if (t0 < t1) dma_rows(0, 0, nrows4, wave, 8);

#ifdef TCVN_DEBUG_KNOBS
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    double s1 = 0, s2 = 0;
    f32x16 accp;                                                                      // role B: its partial sums of the previous tile
#pragma unroll
    for (int e = 0; e < 16; ++e) accp[e] = 0.f;
    int slot_tile = 0;                                                                // ring slot of the current tile's first image row
    int slot_new = wrap(nrows4);                                                      // ring slot of the NEXT tile's first new row
    // role B's deferred epilogue of tile `te`: partial sums of both waves, bias, dropout, store, statistics.  Branch-free: the first
    // version tested `m >= 0`, the dropout group and the store path per element -- ~100 taken branches per tile, 5 460 cycles per wave and
    // tile in the phase counters (4 000 now) -- so everything is computed for all 16 elements and selected.  With all 32 channels present the
    // tile leaves through a bf16 tile in LDS as 16-B stores (two per lane instead of sixteen 2-byte scattered stores).
    auto epilogue_impl = [&](auto dropc, auto vecc, int te, const f32x16& mine) {
        constexpr bool DROP = decltype(dropc)::value, VEC = decltype(vecc)::value;
        const float* xc = xchg + (((te - t0) & 1) * 4 + pw) * 16 * 64 + lane;
        float part[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) part[e] = TCVN_DBG_BIT(g.dbg, 512) ? 0.f : xc[e * 64];
        int mrow[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) mrow[e] = tbl[((te - t0) * TP + q.halo + pw * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) & (PAIR_TBL - 1)];
        float f1 = 0.f, f2 = 0.f;
        bf16* ct = ctile + pw * 32 * 32;                                              // this pair's [32 positions][32 channels] bf16 tile (wave private)
        uint32_t kword = 0;                                                           // lane L < 32: keep flags of position pos(e = L >> 1, h = L & 1)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = mrow[e];
            const bool ok = m >= 0 && nok;
            float v = mine[e] + part[e] + bias;
            if (DROP) {
                const int mm = m < 0 ? 0 : m;
                const float dsc = drop_pick(drop_bits32(dkey, mm, r, g.N), mm, g.drop_p);   // (the launcher checks pixels * N < 2^32)
                v *= dsc;
                // the 32 channels of a position sit in the 32 lanes of a wave half: one ballot = the keep words of two positions
                const unsigned long long bal = __ballot(dsc != 0.f);
                kword = lane == 2 * e ? (uint32_t)bal : lane == 2 * e + 1 ? (uint32_t)(bal >> 32) : kword;
            }
            const bf16 o = ok ? f2bf(v) : (bf16)0;
            const float x = bf2f(o);                                                  // 0 for padding positions / absent channels
            f1 += x; f2 = fmaf(x, x, f2);
            if (VEC) { if (!TCVN_DBG_BIT(g.dbg, 256)) ct[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = o; }
            else if (ok) Out[(long)m * g.ldo + g.n_off + r] = o;
        }
        s1 += (double)f1; s2 += (double)f2;
        if (DROP && g.keep_out != nullptr && lane < 32) {                                 // the backward kernels test these bits instead of hashing
            const int el = lane >> 1, pos = (el & 3) + 8 * (el >> 2) + 4 * (lane & 1);
            const int m = tbl[((te - t0) * TP + q.halo + pw * 32 + pos) & (PAIR_TBL - 1)];
            if (m >= 0) g.keep_out[m] = kword;
        }
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {                                             // 32 positions x 64 B = 128 chunks of 16 B, two per lane
                const int c = lane + 64 * i, pos = c >> 2, chunk = c & 3;
                const int m = tbl[((te - t0) * TP + q.halo + pw * 32 + pos) & (PAIR_TBL - 1)];
                const u16x8 v8 = *reinterpret_cast<const u16x8*>(ct + pos * 32 + chunk * 8);
                if (m >= 0 && !TCVN_DBG_BIT(g.dbg, 128)) *reinterpret_cast<u16x8*>(Out + (long)m * g.ldo + g.n_off + chunk * 8) = v8;
            }
        }
    };
    auto epilogue = [&](int te, const f32x16& mine) {                                  // uniform dispatch, once per tile
        if (drop) {
            if (vec_store) epilogue_impl(std::true_type{}, std::true_type{}, te, mine);
            else epilogue_impl(std::true_type{}, std::false_type{}, te, mine);
        } else {
            if (vec_store) epilogue_impl(std::false_type{}, std::true_type{}, te, mine);
            else epilogue_impl(std::false_type{}, std::false_type{}, te, mine);
        }
    };

    // one multiply pass: NT taps starting at tap `tap_first` over the 32 positions of this pair, image of the tile in ring slot `slot_t`.
    // The A fragments travel LDS -> registers four k-steps (half a tap) ahead of the MFMAs that consume them (two register groups of four
    // fragments); the lgkmcnt wait is placed by hand BEFORE the next group's reads are issued -- hipcc would sink the reads next to their
    // uses or put the wait behind the new reads.  (Two accumulator chains with two-fragment groups measured slower: 212 vs 197 us.)
    auto multiply = [&](auto& bwr, auto ntc, int tap_first, int slot_t, f32x16& acc) {
        constexpr int NT = decltype(ntc)::value, NG = NT * 2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int lrow0 = slot_t + pw * 32 + r + q.halo;                               // < 2 * ring; tap shifts add at most Wp + 1 < ring more
        auto row_of = [&](int tp) {
            const int tap = tap_first + tp;
            int lr = lrow0 + (tap / 3 - 1) * q.Wp + (tap % 3 - 1);
            lr = lr >= ring ? lr - ring : lr;
            return lr >= ring ? lr - ring : lr;
        };
        bf16x8_t af[2][4];
        auto load_group = [&](int gI, bf16x8_t (&dst)[4]) {
            const int lr = row_of(gI >> 1), ks0 = (gI & 1) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dst[i] = *reinterpret_cast<const bf16x8_t*>(smem + lr * 256 + (((2 * (ks0 + i) + h) ^ (lr & 15)) << 4));
        };
        load_group(0, af[0]);
#pragma unroll
        for (int gI = 0; gI < NG; ++gI) {
            __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0): this group's fragments (read four MFMAs ago) are in
            __builtin_amdgcn_sched_barrier(0);
            if (gI + 1 < NG) load_group(gI + 1, af[(gI + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[gI & 1][i], bwr[gI * 4 + i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // The two roles run the same barrier sequence (one __syncthreads per tile + one after the loop) on their own register sets.
    if (!roleB) {
        bf16x8_t bw[40];
#pragma unroll
        for (int i = 0; i < 40; ++i) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + i * 512);
        PAIR_T0();
        int slot_dma = 0;                                                             // ring slot of the rows requested during the previous tile
        for (int t = t0; t < t1; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's share of tile t's rows has landed
            if (xf) {
                if (t == t0) xform_rows(0, 0, nrows4, wave, 8);
                else xform_rows(nrows4 + (t - 1 - t0) * TP, slot_dma, TP, wave, 8);
            }
            PAIR_PH(0);
            __syncthreads();                                                          // ... everybody's; tile t-1's MFMAs are done; xchg / tbl of the last phase visible
            PAIR_PH(1);
            slot_dma = slot_new;
            // the next tile's 128 new rows travel under this tile's work.  Materialised input: issued by role A alone, so that role B's stores do
            // not queue behind loads.  act_fused: every wave requests a sixteenth and activates exactly the rows it requested (both roles share
            // the in-LDS activation: role A alone carried +28 % on the launch, 2.17 -> 2.77 ms per step)
            if (t + 1 < t1) { if (xf) dma_rows(nrows4 + (t - t0) * TP, slot_new, TP, wave, 8); else dma_rows(nrows4 + (t - t0) * TP, slot_new, TP, pw, 4); }
            PAIR_PH(2);
            f32x16 acc;
            multiply(bw, std::integral_constant<int, 5>{}, 0, slot_tile, acc);
            float* xc = xchg + (((t - t0) & 1) * 4 + pw) * 16 * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) xc[e * 64] = acc[e];
            PAIR_PH(3);
            // table of the rows the NEXT iteration will fetch (tile t + 2's new rows), by the 256 role-A threads
            if (t + 2 < t1) fill_rows(nrows4 + (t - t0 + 1) * TP, TP, tid, 256);
            PAIR_PH(4);
            slot_tile = wrap(slot_tile + TP);
            slot_new = wrap(slot_new + TP);
        }
    } else {
        bf16x8_t bw[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + (40 + i) * 512);
        if (!TCVN_DBG_BIT(g.dbg, 1024)) __builtin_amdgcn_s_setprio(1);     // the second-dispatched half loses every issue arbitration otherwise
        PAIR_T0();
        int slot_dma = 0;
        for (int t = t0; t < t1; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // DMA share landed, stores of the last epilogue left
            if (xf) {                                                                  // this wave's share of the tile's new rows (see role A)
                if (t == t0) xform_rows(0, 0, nrows4, wave, 8);
                else xform_rows(nrows4 + (t - 1 - t0) * TP, slot_dma, TP, wave, 8);
            }
            PAIR_PH(8);
            __syncthreads();
            PAIR_PH(9);
            slot_dma = slot_new;
            if (xf && t + 1 < t1) dma_rows(nrows4 + (t - t0) * TP, slot_new, TP, wave, 8);
            PAIR_PH(10);
            if (t > t0) epilogue(t - 1, accp);                                         // finish tile t-1 while role A multiplies tile t
            PAIR_PH(11);
            multiply(bw, std::integral_constant<int, 4>{}, 5, slot_tile, accp);
            PAIR_PH(12);
            slot_tile = wrap(slot_tile + TP);
            slot_new = wrap(slot_new + TP);
        }
    }
#ifdef TCVN_DEBUG_KNOBS
    if (lane == 0 && (blockIdx.x & 15) == 0)                  // every 16th workgroup reports (the atomics of all of them cost ~70 us per launch)
        for (int i = 0; i < 16; ++i)
            if (ph[i]) atomicAdd(&g_pair_ph[i], ph[i]);
#endif
    __syncthreads();                                                                  // the last tile's exchange buffer is complete
    if (roleB && t1 > t0) epilogue(t1 - 1, accp);
    if (g.part != nullptr || g.isum_out != nullptr) {
        double a = s1, b = s2;
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
        if (roleB && lane < 32) { red[(pw * 32 + lane) * 2] = a; red[(pw * 32 + lane) * 2 + 1] = b; }
        __syncthreads();
        if (tid < g.N) {
            double x = 0, y = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { x += red[(w * 32 + tid) * 2]; y += red[(w * 32 + tid) * 2 + 1]; }
            if (g.isum_out != nullptr) lf_add(g.isum_out, g.isum_stride, tid, x, y);                   // link-free: the next consumer derives its table itself
            else {
                g.part[((long)blockIdx.x * g.N + tid) * 2] = x;
                g.part[((long)blockIdx.x * g.N + tid) * 2 + 1] = y;
            }
        }
    }
}
int fwd_pair_ring(const PadGeom& q) { return (int)((((q.rows() + 3) & ~3) + TP + 15) & ~15); }
size_t fwd_pair_smem(const PadGeom& q) { const size_t ring = fwd_pair_ring(q); return ring * 256 + PAIR_TBL * 4 + 2 * 4 * 16 * 64 * 4 + 4 * 32 * 32 * 2 + 4 * 32 * 16 + 3 * 128 * 4; }


size_t fwd_smem(const PadGeom& q) { const size_t r4 = (q.rows() + 3) & ~3; return 2 * r4 * 256 + 3 * r4 * 4 + 4 * 32 * 16; }

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient: dW[tap][c][n] = sum over padded positions p of Yact[p + shift(tap)][c] * eff[p][n].
// The contraction runs over pixels, i.e. over the ROW index of both row-major LDS images, so both MFMA operands are
// read transposed with ds_read_b64_tr_b16 (semantics checked by tools/micro/tr_read_test.hip): per 16-position k-step a
// wave reads the eff fragment once and nine shifted Yact fragments, and owns the 9 x (32 c x 32 n) accumulators of its
// 32-channel slice for the whole launch (144 accumulator registers); one atomic pass at the end.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8_t tr_frag(const char* smem_base, int off_lo, int off_hi) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_hi));
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8_t, pr);
}

constexpr int WG_RING = 512, WG_TBL = 1024;
// Round 5: the ring carries WG_MIRROR extra rows behind its end that MIRROR its first rows (whoever writes ring row r < WG_MIRROR also writes
// row WG_RING + r).  A transposed read takes the 16 consecutive rows of a k-step starting anywhere in the ring: with the mirror it never
// wraps inside an instruction, so its address is (per-lane part, fixed per tap) + (a SCALAR start, one s_and per tap and k-step) -- one VALU
// add per fragment instead of an add and a mask per read (the multiplying waves issued ~560 instructions per tile around their 72 MFMAs
// and shared the SIMD's issue slots with the helper wave: 4 800 cycles per tile for 2 300 of matrix work, tools/wgrad_phases.py).  The
// ring's swizzle moves only the 64-B quad (wswz: row & 3), so rows r and r + 4 -- the two halves of a fragment -- differ by exactly 1 KiB:
// the second read of a fragment is the first one's address with an immediate offset.
constexpr int WG_MIRROR = 16, WG_RING_BYTES = (WG_RING + WG_MIRROR) * 256;
__device__ __forceinline__ int wswz(int r) { return (r & 3) << 2; }
__device__ __forceinline__ void wg_kloop(f32x16 (&acc)[9], const char* smem, int base_row, const int (&arow0)[9], const int (&lp)[9], const char* ebase) {
    bf16x8_t fr[3][3], fb[2];
    auto issue = [&](int j) {                                       // j = 3 * ks + third
        const int ks = j / 3, third = j - 3 * ks;
        if (third == 0) fb[ks & 1] = tr_frag(ebase, ks * 1024, ks * 1024 + 256);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int tap = third * 3 + i;
            const int s0 = ((base_row + arow0[tap] + 16 * ks) & (WG_RING - 1)) << 8;      // uniform: scalar ALU
            fr[j % 3][i] = tr_frag(smem + s0, lp[tap], lp[tap] + 1024);
        }
    };
    constexpr int NG = 3 * (TP / 16);
    issue(0);
    issue(1);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        // group j complete <=> at most the reads of group j+1 outstanding (6, or 8 when it opens a k-step)
        if (j + 1 < NG) { if ((j + 1) % 3 == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (8 << 8)); else __builtin_amdgcn_s_waitcnt(0xC07F | (6 << 8)); }
        else __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 2 < NG) issue(j + 2);
        __builtin_amdgcn_sched_barrier(0);
        const int ks = j / 3, third = j - 3 * ks;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            acc[third * 3 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % 3][i], fb[ks & 1], acc[third * 3 + i], 0, 0, 0);
    }
}

// Two waves per SIMD with different roles (as in the forward pair kernel): 512 threads.  Waves 0-3 only multiply -- the pipelined
// fragment reads + 72 MFMAs of tile t.  Waves 4-7 prepare tile t+1 meanwhile: image rows by LDS-DMA, the eff tile (slice loads,
// BatchNorm mean terms, dropout keep bits), the pixel table of tile t+2.  One barrier per tile.
// A workgroup walks CONSECUTIVE tiles and keeps the image in a ring of WG_RING = 512 rows (128 KB): tile t+1 shares all but its last
// 128 rows with tile t, so only those are fetched -- 32 KB per tile instead of the whole 70 KB image (272 rows at W = 69).  The phase
// counters of the one-image-per-tile version showed the helper waves, not the MFMAs, bounding the tile: 6 700 cycles stalled in the DMA
// issue and 5 100 more until the slice loads queued behind it returned, i.e. the kernel moved its 2.1x redundant image traffic at
// the HBM rate (3.2 TB/s) while the multiplying waves waited 9 600 of 13 700 cycles at the barrier.
// Row space of a workgroup: row 0 = padded position t0 * TP - halo; ring slot = row & 511; table entry = row & 1023; the XOR
// swizzle of a row's 16-B chunks is wswz(row) = (row & 3) << 2 (TP and the ring are multiples of 16, so it equals the tile-relative
// value).  Every LDS read of the multiplying waves is a transposed read of four consecutive rows x 64 B: the forward kernels' `row & 15`
// swizzle put those four rows on the same 16 banks (two- to four-way conflicts on all 160 reads per wave and tile, rocprofv3 round 4:
// LDS_BANK_CONFLICT = 2.6 x the LDS-active cycles); with row & 3 selecting the 64-B quad they cover all 64 banks.
__global__ __launch_bounds__(512, 1) void k_conv3x3_wgrad_bf16(const ConvWgradArgs g, int n_img, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvFwdArgs& fa = g.fa;
    const EffSrc& e = g.e;
    const PadGeom q(n_img, fa.H, fa.W);
    const int nrows4 = (q.rows() + 3) & ~3;
    constexpr int eff_off = WG_RING_BYTES;                                // two [TP][32] bf16 tiles behind the ring (+ mirror rows), 64-B rows, unswizzled
    int* tbl = reinterpret_cast<int*>(smem + eff_off + 2 * TP * 64);      // [1024] pixel index of row (row & 1023)
    float* bred = reinterpret_cast<float*>(smem + eff_off);               // [64][32], aliases the eff tiles after the last barrier
    float* xtab = reinterpret_cast<float*>(smem + eff_off + 2 * TP * 64 + WG_TBL * 4);      // act_fused: [3][128] tables of the image's BatchNorm + PReLU
    const bool xf = fa.act_fused != 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w4 = wave & 3, htid = tid & 255;                             // wave / thread index inside the role
    const int nb = gridDim.x, per = ntiles / nb, rem = ntiles % nb;
    const int t0 = blockIdx.x * per + min((int)blockIdx.x, rem), t1 = t0 + per + ((int)blockIdx.x < rem ? 1 : 0);
    const int g_org = t0 * TP - q.halo;
#ifdef TCVN_DEBUG_KNOBS
    unsigned long long ph[16] = {0};
#endif

    if (wave >= 4) {
        // ---------------- helper role: tables, image DMA, eff tiles ----------------
        const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
        const bf16* __restrict__ YA = reinterpret_cast<const bf16*>(fa.Aact);
        const char* __restrict__ zeros = reinterpret_cast<const char*>(fa.zeros);
        const bf16* __restrict__ G = reinterpret_cast<const bf16*>(e.G);
        const bf16* __restrict__ D = reinterpret_cast<const bf16*>(e.X);
        const bool drop = e.drop_p > 0.f;
        const uint32_t dkey = drop_key(e.seed, e.stream_id);
        const int ec = htid & 3, ra = htid >> 2;                           // eff staging: rows ra and ra + 64, channel chunk ec
        float cP[8], cQ[8], bsum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = ec * 8 + j;
            cP[j] = n < e.N ? e.P[n] : 0.f; cQ[j] = n < e.N ? e.Q[n] : 0.f; bsum[j] = 0.f;
        }
        const uint32_t* __restrict__ KM = e.keep;                          // keep words written by the forward kernel (or nullptr: hash)
        const float dinv = 1.f / (1.f - e.drop_p);
        auto eff_fetch = [&](int m, u16x8& gv, u16x8& xv, uint32_t& kw) {                     // slice rows + keep word of pixel m: three loads
            const long o = (long)(m >= 0 ? m : 0);
            gv = *reinterpret_cast<const u16x8*>(G + o * e.ldg + e.c_off + ec * 8);
            xv = *reinterpret_cast<const u16x8*>(D + o * e.ldx + e.c_off + ec * 8);
            kw = *(KM != nullptr ? KM + o : reinterpret_cast<const uint32_t*>(zeros));      // always three loads per row (the barrier counts them)
        };
        auto eff_load = [&](int row0, int i, u16x8& gv, u16x8& xv, uint32_t& kw) -> int {      // row0: first body row of the tile
            const int m = tbl[(row0 + ra + 64 * i) & (WG_TBL - 1)];
            eff_fetch(m, gv, xv, kw);
            return m;
        };
        auto eff_store = [&](int buf, int i, int m, const u16x8& gv, const u16x8& xv, uint32_t kw) {
            u16x8 o;
            const uint32_t kb = kw >> (ec * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = 0.f;
                const int n = ec * 8 + j;
                if (m >= 0 && n < e.N) {
                    t = eff3(bf2f(gv[j]), cP[j], bf2f(xv[j]), cQ[j]);
                    if (drop) t *= KM != nullptr ? (((kb >> j) & 1u) ? dinv : 0.f)
                                                 : drop_pick(drop_bits32(dkey, m, n, e.N), m, e.drop_p);      // (pixels * N < 2^32: conv3x3_wgrad_tile_ok)
                }
                o[j] = f2bf(t);
                bsum[j] += bf2f(o[j]);
            }
            *reinterpret_cast<u16x8*>(smem + eff_off + buf * TP * 64 + (ra + 64 * i) * 64 + ec * 16) = o;
        };
        auto fill_rows = [&](int row0, int n) {                            // table entries of rows [row0, row0 + n)
            for (int i = htid; i < n; i += 256) tbl[(row0 + i) & (WG_TBL - 1)] = pix_of(q, g_org + row0 + i, invWp, invHp);
        };
        // rows [row0, row0 + n) -> ring (n multiple of 4): 4 rows = 1 KiB per wave instruction; all table reads first, then the DMAs
        auto dma_rows = [&](int row0, int n) {
            const int rsub = lane >> 4, slot = lane & 15;
            constexpr int DMA_RG = 24;                                     // row groups per wave: up to 384 rows per call
            int mrow[DMA_RG];
#pragma unroll
            for (int i = 0; i < DMA_RG; ++i) {
                const int rg = w4 + 4 * i;
                mrow[i] = rg * 4 < n ? tbl[(row0 + rg * 4 + rsub) & (WG_TBL - 1)] : -1;
            }
#pragma unroll
            for (int i = 0; i < DMA_RG; ++i) {
                const int rg = w4 + 4 * i;
                if (rg * 4 < n) {
                    const int row = row0 + rg * 4, rr = row + rsub;       // row0 multiple of 4: the group stays inside the ring
                    const char* src = mrow[i] >= 0 ? reinterpret_cast<const char*>(YA + (long)mrow[i] * 128) + ((slot ^ wswz(rr)) << 4)
                                                   : zeros + (slot << 4);
                    // the DMA as inline assembly: behind the builtin the compiler orders every later LDS access of this wave (eff tile,
                    // table) behind vmcnt(0) -- it cannot know they touch other rows -- which would put the DMAs last in the iteration with
                    // their latency exposed at the barrier.  The waits that matter are placed by hand (vmcnt(6) in front of the barrier).
                    const unsigned lds = __builtin_amdgcn_readfirstlane(
                        (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)(smem + (row & (WG_RING - 1)) * 256)));
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(src) : "memory");      // (m0 is not allocatable: nothing else in this kernel uses it)
                    if ((row & (WG_RING - 1)) < WG_MIRROR) {               // the ring's first rows also live behind its end (uniform: row is a multiple of 4)
                        const unsigned lds2 = lds + WG_RING * 256;
                        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds2), "v"(src) : "memory");
                    }
                }
            }
        };
        // act_fused: the image rows arrive RAW (the bottleneck map itself); the helper wave that requested a row group applies the
        // layer's BatchNorm + PReLU to it in LDS once its DMAs have landed, in front of the tile barrier (same row groups as dma_rows)
        // (phase counters, first version: 3 630 cycles per wave and tile for 8 chunks -- three dependent LDS round trips per group of four
        // chunks and the 24 table values re-read per call; now every LDS read of a call is requested before the first use and the
        // thread's table values stay in registers for the whole launch)
        Act8 xtb;
        auto xform_rows = [&](int row0, int n) {
            const int rsub = lane >> 4, cc = lane & 15;
            for (int i0 = 0; (w4 + 4 * i0) * 4 < n; i0 += 8) {
                int mr[8];
                u16x8 v[8];
                char* p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int rg = w4 + 4 * (i0 + j);
                    const bool in = rg * 4 < n;
                    const int row = row0 + (in ? rg * 4 : 0) + rsub;
                    mr[j] = tbl[row & (WG_TBL - 1)];
                    if (!in) mr[j] = -1;
                    p[j] = smem + (row & (WG_RING - 1)) * 256 + ((cc ^ wswz(row)) << 4);
                    v[j] = *reinterpret_cast<const u16x8*>(p[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (mr[j] >= 0) {
                        const u16x8 o = act8_apply(v[j], xtb);
                        *reinterpret_cast<u16x8*>(p[j]) = o;
                        if (p[j] < smem + WG_MIRROR * 256) *reinterpret_cast<u16x8*>(p[j] + WG_RING * 256) = o;      // mirror of the ring's first rows
                    }
            }
        };
        // With the eff rows materialised by the data-gradient kernel (e.ey): a tile's 128 rows x 64 B arrive by DMA like the image rows,
        // 16 rows per wave instruction (lane = 4 * row + chunk); the helper then only adds up the bias gradient from the landed tile.
        const bf16* __restrict__ EYs = reinterpret_cast<const bf16*>(e.ey);
        auto dma_eff = [&](int row0, int buf) {                            // row0: first body row of the tile
            const int rsub = lane >> 2, chunk = lane & 3;
            int mr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) mr[i] = tbl[(row0 + (w4 + 4 * i) * 16 + rsub) & (WG_TBL - 1)];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* src = mr[i] >= 0 ? reinterpret_cast<const char*>(EYs + (long)mr[i] * 32 + chunk * 8) : zeros + chunk * 16;
                const unsigned lds = __builtin_amdgcn_readfirstlane(
                    (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)(smem + eff_off + buf * TP * 64 + (w4 + 4 * i) * 1024)));
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(src) : "memory");
            }
        };
        auto bias_from_tile = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u16x8 v = *reinterpret_cast<const u16x8*>(smem + eff_off + buf * TP * 64 + (ra + 64 * i) * 64 + ec * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[j] += bf2f(v[j]);
            }
        };
        // Pipeline of the helper role, iteration i (tile t0 + i is being multiplied): eff tile of tile i+1 from the slice rows requested
        // one iteration earlier (a tile time ago: they have arrived), DMA of tile i+1's new image rows, slice loads of tile i+2,
        // table entries of tile i+3's new rows.  The barrier waits for the DMAs only -- vmcnt(6): the six slice loads behind them stay in
        // flight (waiting for them there cost 5 300 cycles per tile, the whole memory latency under load).
        const int ntl = t1 - t0;
        if (ntl > 0) fill_rows(0, nrows4 + min(ntl - 1, 2) * TP);          // tile 0's image rows, the new rows of tiles 1 and 2
        if (xf) act_tab_fill(xtab, fa.sc, fa.sh, fa.sl, htid, 256);
        __syncthreads();                                                    // (1)
        if (xf) xtb = act8_load(xtab, lane & 15);
        u16x8 gv[2], xv[2];
        uint32_t kw[2];
        int mm[2] = {-1, -1};
        if (ntl > 0) {
            dma_rows(0, nrows4);
            if (EYs != nullptr) dma_eff(q.halo, 0);
            else {
#pragma unroll
                for (int i = 0; i < 2; ++i) { u16x8 g0, x0; uint32_t k0; const int m = eff_load(q.halo, i, g0, x0, k0); eff_store(0, i, m, g0, x0, k0); }
            }
        }
        if (ntl > 1 && EYs == nullptr) {
#pragma unroll
            for (int i = 0; i < 2; ++i) mm[i] = eff_load(TP + q.halo, i, gv[i], xv[i], kw[i]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (xf && ntl > 0) xform_rows(0, nrows4);
        __syncthreads();                                                    // (2)
        int cur = 0;
        PAIR_T0();
        if (EYs != nullptr && xf) {
            // Round 5 (the path the dense blocks 1-2 run).  Per tile the helper moves 128 image rows x 256 B (activated on the way) and 128 eff rows
            // x 64 B from HBM to LDS.  Requested one tile ahead and waited for in the same iteration (round 4), a helper wave spent ~1 500 of its
            // 6 100 cycles per tile waiting for those loads, and the multiplying waves waited for the helpers (tools/wgrad_phases.py: k loop 3 800
            // cycles, barrier 2 700).  Now BOTH travel HBM -> registers TWO tiles ahead (two register sets, the loop unrolled by two): a request has
            // a whole tile time to arrive, nothing in the loop waits on vmcnt for the current iteration's requests (the barrier needs lgkmcnt
            // only), the eff tile needs no DMA and its bias sums come from the registers.  32-bit byte offsets against a uniform base (the
            // launchers bound pixels * 256 B below 4 GB).
            const int rsub_x = lane >> 4, cc_x = lane & 15;
            const char* __restrict__ yab = reinterpret_cast<const char*>(YA);
            const char* __restrict__ eyb = reinterpret_cast<const char*>(EYs);
            u16x8 rva[10], rvb[10];                                           // [0..7] image chunks, [8..9] eff chunks (rows ra, ra + 64; chunk ec)
            int ma[10], mb[10];
            auto fetch = [&](int tl, u16x8 (&rv)[10], int (&rm)[10]) {      // tile tl's 128 new image rows and its 128 eff rows -> registers
                const int row0 = (tl - 1) * TP + nrows4, body = tl * TP + q.halo;
#pragma unroll
                for (int j = 0; j < 8; ++j) rm[j] = tbl[(row0 + (w4 + 4 * j) * 4 + rsub_x) & (WG_TBL - 1)];
#pragma unroll
                for (int i = 0; i < 2; ++i) rm[8 + i] = tbl[(body + ra + 64 * i) & (WG_TBL - 1)];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    rv[j] = *reinterpret_cast<const u16x8*>(yab + (size_t)((unsigned)(rm[j] >= 0 ? rm[j] : 0) * 256u + (unsigned)(cc_x * 16)));
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    rv[8 + i] = *reinterpret_cast<const u16x8*>(eyb + (size_t)((unsigned)(rm[8 + i] >= 0 ? rm[8 + i] : 0) * 64u + (unsigned)(ec * 16)));
            };
            auto commit = [&](int tl, const u16x8 (&rv)[10], const int (&rm)[10]) {      // ... -> the ring (activated) and eff buffer tl & 1
                const int row0 = (tl - 1) * TP + nrows4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = row0 + (w4 + 4 * j) * 4 + rsub_x;
                    u16x8 o = act8_apply(rv[j], xtb);
                    if (rm[j] < 0) o = u16x8{0, 0, 0, 0, 0, 0, 0, 0};                      // padding position: a zero row
                    char* wp = smem + (row & (WG_RING - 1)) * 256 + ((cc_x ^ wswz(row)) << 4);
                    *reinterpret_cast<u16x8*>(wp) = o;
                    if ((row & (WG_RING - 1)) < WG_MIRROR) *reinterpret_cast<u16x8*>(wp + WG_RING * 256) = o;       // mirror of the ring's first rows
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    u16x8 o = rv[8 + i];
                    if (rm[8 + i] < 0) o = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<u16x8*>(smem + eff_off + (tl & 1) * TP * 64 + (ra + 64 * i) * 64 + ec * 16) = o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum[j] += bf2f(o[j]);
                }
            };
            if (ntl > 0) bias_from_tile(0);                                 // tile 0's eff rows came by DMA above
            if (ntl > 1) fetch(1, rva, ma);
            for (int il = 0; il < ntl; il += 2) {
                // ---- even half: the multiplying waves work on tile il; set A holds tile il+1 (requested a tile ago) ----
                if (il + 2 < ntl) fetch(il + 2, rvb, mb);
                PAIR_PH(8);
                if (il + 3 < ntl) fill_rows((il + 2) * TP + nrows4, TP);                     // new rows of tile il+3
                PAIR_PH(10);
                if (il + 1 < ntl) commit(il + 1, rva, ma);
                PAIR_PH(13);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // (tile)
                PAIR_PH(12);
                if (il + 1 >= ntl) break;
                // ---- odd half: tile il+1 is multiplied; set B holds tile il+2 ----
                if (il + 3 < ntl) fetch(il + 3, rva, ma);
                PAIR_PH(8);
                if (il + 4 < ntl) fill_rows((il + 3) * TP + nrows4, TP);                     // new rows of tile il+4
                PAIR_PH(10);
                if (il + 2 < ntl) commit(il + 2, rvb, mb);
                PAIR_PH(13);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                PAIR_PH(12);
            }
        } else if (EYs != nullptr) {
            // act_fused: the 128 new image rows of tile il+1 travel HBM -> registers (8 x 16 B per lane) instead of HBM -> LDS, are activated in
            // registers and written to the ring once.  The DMA + in-place variant needed an LDS read and a second LDS write per chunk, queued
            // behind the multiplying waves' 160 transposed reads per tile on the one LDS pipe of the CU: 3 200-3 600 cycles per wave and tile
            // in the phase counters (the multiplying waves then waited 2 500 of 7 400 cycles at the barrier).
            u16x8 rv[8];
            int rm[8];
            const int rsub_x = lane >> 4, cc_x = lane & 15;
            for (int il = 0; il < ntl; ++il, cur ^= 1) {
                if (il + 1 < ntl) {
                    if (xf) {
                        const int row0 = il * TP + nrows4;
#pragma unroll
                        for (int j = 0; j < 8; ++j) rm[j] = tbl[(row0 + (w4 + 4 * j) * 4 + rsub_x) & (WG_TBL - 1)];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            rv[j] = *reinterpret_cast<const u16x8*>(YA + (long)(rm[j] >= 0 ? rm[j] : 0) * 128 + cc_x * 8);
                    } else dma_rows(il * TP + nrows4, TP);                 // image: the 128 rows tile il+1 does not share with tile il
                    dma_eff((il + 1) * TP + q.halo, cur ^ 1);              // its eff rows
                }
                PAIR_PH(8);
                bias_from_tile(cur);
                PAIR_PH(9);
                if (il + 3 < ntl) fill_rows((il + 2) * TP + nrows4, TP);   // new rows of tile il+3
                PAIR_PH(10);
                if (xf && il + 1 < ntl) {
                    const int row0 = il * TP + nrows4;
                    PAIR_PH(11);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = row0 + (w4 + 4 * j) * 4 + rsub_x;
                        u16x8 o = act8_apply(rv[j], xtb);
                        if (rm[j] < 0) o = u16x8{0, 0, 0, 0, 0, 0, 0, 0};                  // padding position: a zero row
                        char* wp = smem + (row & (WG_RING - 1)) * 256 + ((cc_x ^ wswz(row)) << 4);
                        *reinterpret_cast<u16x8*>(wp) = o;
                        if ((row & (WG_RING - 1)) < WG_MIRROR) *reinterpret_cast<u16x8*>(wp + WG_RING * 256) = o;       // mirror of the ring's first rows
                    }
                    PAIR_PH(13);
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (tile)
                PAIR_PH(12);
            }
        } else
        for (int il = 0; il < ntl; ++il, cur ^= 1) {
            // the six slice loads of the previous iteration landed a tile ago: this wait is free, and it tells the compiler's scoreboard
            // that their registers are ready, so that nothing below waits on the vmcnt counter behind the DMAs
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (il + 1 < ntl) dma_rows(il * TP + nrows4, TP);              // the 128 rows tile il+1 does not share with tile il: requested first,
            PAIR_PH(8);                                                    // they travel under the eff tile and the table work
            if (il + 1 < ntl) {
#pragma unroll
                for (int i = 0; i < 2; ++i) eff_store(cur ^ 1, i, mm[i], gv[i], xv[i], kw[i]);      // tile il+1 (loads of the previous iteration)
            }
            PAIR_PH(9);
            const int body2 = il + 2 < ntl ? (il + 2) * TP + q.halo : q.halo;                     // (past the end: any rows -- keeps six loads behind the DMAs)
#pragma unroll
            for (int i = 0; i < 2; ++i) mm[i] = tbl[(body2 + ra + 64 * i) & (WG_TBL - 1)];
            if (il + 3 < ntl) fill_rows((il + 2) * TP + nrows4, TP);       // new rows of tile il+3
            PAIR_PH(10);
            __builtin_amdgcn_sched_barrier(0);                             // program order = issue order: the six loads below stay BEHIND the DMAs
#pragma unroll
            for (int i = 0; i < 2; ++i) eff_fetch(mm[i], gv[i], xv[i], kw[i]);
            __builtin_amdgcn_sched_barrier(0);
            if (xf) {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // the DMAs (in front of the six slice loads) have landed
                if (il + 1 < ntl) xform_rows(il * TP + nrows4, TP);
            }
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (tile) DMAs landed, eff tile and table written
            PAIR_PH(12);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (g.dbias != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bred[ra * 32 + ec * 8 + j] = bsum[j];
        }
    } else {
        // ---------------- multiplying role ----------------
        f32x16 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        // tr-read lane roles: group gq = lane>>4 -> (k half = gq>>1, column half = gq&1); lane 4q+p supplies row q, cols 4p..4p+3
        const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
        const int khalf = gq >> 1, chalf = gq & 1;
        const int a_chunk = w4 * 4 + 2 * chalf + (tp >> 1), a_sub = (tp & 1) * 8;         // Yact: this wave's 32 channels
        const int b_colbyte = (16 * chalf + 4 * tp) * 2;                                  // eff: 32 channels
        // a fragment's address = scalar start of its 16-row k-step window (wg_kloop) + this lane's part: row 8*khalf + tq of the window, the
        // swizzled 16-B chunk (the swizzle takes row & 3: window starts differ from the tile's first row by multiples of 16 plus the tap's
        // constant shift, so it is fixed per tap) and the 8-B half; the second read of the fragment is 4 rows = 1 KiB further
        int arow0[9], lp[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            arow0[tap] = q.halo + (tap / 3 - 1) * q.Wp + (tap % 3 - 1);
            const int arow = arow0[tap] + 8 * khalf + tq;
            lp[tap] = (8 * khalf + tq) * 256 + ((a_chunk ^ wswz(arow)) << 4) + a_sub;
        }
        const int b_off0 = (8 * khalf + tq) * 64 + b_colbyte;
        __syncthreads();                                                    // (1)
        __syncthreads();                                                    // (2)
        int cur = 0;
        PAIR_T0();
        // 8 k-steps x (1 eff + 9 image fragments, two transposed LDS reads each) as a software pipeline over 24 groups of three image
        // fragments (the first group of a k-step also carries the eff fragment): group j+2 is requested while group j is multiplied, so
        // 12-14 LDS reads stay in flight (lgkmcnt holds 15) instead of every k-step waiting for its own 20 reads (7 200 cycles per wave and
        // tile in the phase counters against 2 300 of MFMA issue).  The waits are placed by hand in front of the new requests -- left to
        // itself the compiler sinks the requests behind the MFMAs or waits for all of them.
        for (int t = t0; t < t1; ++t, cur ^= 1) {
            const int base_row = ((t - t0) * TP) & (WG_RING - 1);
            const char* ebase = smem + eff_off + cur * TP * 64 + b_off0;
            wg_kloop(acc, smem, base_row, arow0, lp, ebase);
            PAIR_PH(0);
            __syncthreads();                                                // (tile)
            PAIR_PH(1);
        }
        // dW[tap*128 + c][n] ; rows of the C tile are this wave's channels, columns the 32 output channels
        const int n = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = w4 * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                g.slab[(long)blockIdx.x * (9 * 128 * 32) + ((long)tap * 128 + c) * 32 + n] = acc[tap][i];
            }
    }
#ifdef TCVN_DEBUG_KNOBS
    if (lane == 0 && (blockIdx.x & 15) == 0)
        for (int i = 0; i < 16; ++i)
            if (ph[i]) atomicAdd(&g_wg_ph[i], ph[i]);
#endif
    if (g.dbias != nullptr) {
        __syncthreads();
        if (tid < 32) {
            float sum = 0.f;
            for (int rr = 0; rr < 64; ++rr) sum += bred[rr * 32 + tid];
            g.slab[(long)gridDim.x * (9 * 128 * 32) + blockIdx.x * 32 + tid] = tid < e.N ? sum : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Data gradient: dA[p][c] = sum_tap sum_n eff[p - shift(tap)][n] * W2[n][c][tap]  (128 channels out, K = 9 x 32), followed
// by the PReLU + BatchNorm backward of norm2 on the bottleneck tensor Y.  The 32-channel eff image (gradient of the layer's
// concat slice, dropout mask and BN mean-terms applied) is built once per tile in LDS; each wave owns 32 of the 128 output
// channels for all 128 positions, so its 18 weight fragments stay in registers.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_conv3x3_dgrad_bf16(const ConvDgradArgs g, int n_img, int ntiles, int swz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const EffSrc& e = g.e;
    const PadGeom q(n_img, g.H, g.W);
    const int nrows4 = (q.rows() + 3) & ~3;
    int* tbl = reinterpret_cast<int*>(smem + nrows4 * 64);                 // [nrows4] pixel index per image row
    constexpr int CLD3 = 132;
    double* sred = reinterpret_cast<double*>(smem + nrows4 * 68);           // [128][3] per-channel sums of this workgroup
    float* Cs = reinterpret_cast<float*>(smem + nrows4 * 68 + 128 * 24);    // [64][CLD3] fp32 dA tile (one pass)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const bf16* __restrict__ G = reinterpret_cast<const bf16*>(e.G);
    const bf16* __restrict__ D = reinterpret_cast<const bf16*>(e.X);
    const bf16* __restrict__ Y = reinterpret_cast<const bf16*>(g.Xin);
    bf16* __restrict__ DU = reinterpret_cast<bf16*>(g.Gout);
    const int nb = gridDim.x;
    const int lb = swz ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const bool drop = e.drop_p > 0.f;
    const uint32_t dkey = drop_key(e.seed, e.stream_id);

    // weights of this wave's 32 output channels: fragment (row tile = wave, k-step) at ((wave*18 + ks)*64 + lane)*8
    const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + ((long)wave * 18 * 64 + lane) * 8;
    bf16x8_t bw[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + i * 512);
    const int e_c8 = tid & 15, e_r0 = tid >> 4;                            // epilogue role: 8-channel chunk, rows e_r0 + 16*i
    float esc[8], esh[8], esl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { esc[j] = g.sc[e_c8 * 8 + j]; esh[j] = g.sh[e_c8 * 8 + j]; esl[j] = g.sl[e_c8 * 8 + j]; }
    for (int i = tid; i < 128 * 3; i += 256) sred[i] = 0.0;

    const int ec = tid & 3, er0 = tid >> 2;                                // eff staging: chunk ec of rows er0, er0+64, ...
    float cP[8], cQ[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = ec * 8 + j;
        cP[j] = n < e.N ? e.P[n] : 0.f; cQ[j] = n < e.N ? e.Q[n] : 0.f;
    }
    for (int t = lb; t < ntiles; t += nb) {
        const int g0 = t * TP;
        __syncthreads();
        for (int rr = tid; rr < nrows4; rr += 256) tbl[rr] = pix_of(q, g0 - q.halo + rr, invWp, invHp);
        __syncthreads();
        // eff image: 16-B chunk ec of row rr at rr*64 + ((ec ^ ((rr>>2)&3)) << 4)
        for (int rr = er0; rr < nrows4; rr += 64) {
            const int m = tbl[rr];
            u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m >= 0) {
                const u16x8 gv = *reinterpret_cast<const u16x8*>(G + (long)m * e.ldg + e.c_off + ec * 8);
                const u16x8 xv = *reinterpret_cast<const u16x8*>(D + (long)m * e.ldx + e.c_off + ec * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = ec * 8 + j;
                    float v = 0.f;
                    if (n < e.N) {
                        v = eff3(bf2f(gv[j]), cP[j], bf2f(xv[j]), cQ[j]);
                        if (drop) v *= drop_pick(drop_bits(dkey, m, n, e.N), m, e.drop_p);
                    }
                    o[j] = f2bf(v);
                }
            }
            *reinterpret_cast<u16x8*>(smem + off64(rr, ec)) = o;
        }
        __syncthreads();

        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int base = q.halo - ((tap / 3 - 1) * q.Wp + (tap % 3 - 1)) + r;      // source position = p - shift(tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int lr = base + mt * 32;
                    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(smem + off64(lr, 2 * ks + h));
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[tap * 2 + ks], acc[mt], 0, 0, 0);
                }
            }
        }
        // epilogue through LDS (two passes of 64 rows): the fp32 dA tile is re-read as 8-channel chunks so that Y is loaded and
        // DU stored 16 B per lane; u = sc*y + sh ; dU = dA * prelu'(u) ; DU = sc*dU ; sums (dU, dU*y, dA*min(u,0)) per channel
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int mt = pass * 2 + half;
#pragma unroll
                for (int k = 0; k < 16; ++k) Cs[(half * 32 + (k & 3) + 8 * (k >> 2) + 4 * h) * CLD3 + wave * 32 + r] = acc[mt][k];
            }
            __syncthreads();
            float f1[8], f2[8], f3[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { f1[j] = 0.f; f2[j] = 0.f; f3[j] = 0.f; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = e_r0 + 16 * i;
                const int m = tbl[q.halo + pass * 64 + rr];
                if (m >= 0) {
                    const float4 ca = *reinterpret_cast<const float4*>(Cs + rr * CLD3 + e_c8 * 8);
                    const float4 cc = *reinterpret_cast<const float4*>(Cs + rr * CLD3 + e_c8 * 8 + 4);
                    const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cc.x, cc.y, cc.z, cc.w};
                    const u16x8 yv = *reinterpret_cast<const u16x8*>(Y + (long)m * g.ldxin + e_c8 * 8);
                    u16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = bf2f(yv[j]);
                        const float u = fmaf(y, esc[j], esh[j]);
                        const float du = u > 0.f ? cv[j] : esl[j] * cv[j];
                        f1[j] += du; f2[j] += du * y; f3[j] += u > 0.f ? 0.f : cv[j] * u;
                        o[j] = f2bf(esc[j] * du);
                    }
                    *reinterpret_cast<u16x8*>(DU + (long)m * g.ldgo + e_c8 * 8) = o;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                atomicAdd(&sred[(e_c8 * 8 + j) * 3], (double)f1[j]);
                atomicAdd(&sred[(e_c8 * 8 + j) * 3 + 1], (double)f2[j]);
                atomicAdd(&sred[(e_c8 * 8 + j) * 3 + 2], (double)f3[j]);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid < 128) {
        double* p = g.part + ((long)blockIdx.x * g.N + tid) * 3;
        p[0] = sred[tid * 3]; p[1] = sred[tid * 3 + 1]; p[2] = sred[tid * 3 + 2];
    }
}

// Pipelined variant (concat slice 16-B aligned, 32 channels): the raw (G, x) slice rows of tile t+1 travel by LDS-DMA while
// tile t is multiplied and its epilogue runs; the epilogue's Y rows are requested one 32-row pass ahead; the per-channel sums
// stay in registers until the end of the launch.  Barriers inside the pipeline are bare s_barrier + lgkmcnt(0): a
// __syncthreads() would drain the DMA and the prefetched loads (its fence waits for vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256, 2) void k_conv3x3_dgrad2_bf16(const ConvDgradArgs g, int n_img, int ntiles, int swz, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const EffSrc& e = g.e;
    const PadGeom q(n_img, g.H, g.W);
    const int nr = (q.rows() + 15) & ~15;                       // image rows, whole DMA row groups (16 rows x 64 B = 1 KiB)
    constexpr int CLD3 = 132;
    const int o_rg = nr * 64, o_rd = 2 * nr * 64, o_tbl = 3 * nr * 64, o_cs = o_tbl + 2 * nr * 4;     // eff image at 0
    const int o_km = o_cs + 32 * 132 * 4 + 448 * 4;             // [nr rounded up to 64] keep words of the tile's rows (when the forward stored them)
    int* tbl = reinterpret_cast<int*>(smem + o_tbl);            // [2][nr] pixel index per image row (this tile / next tile)
    float* Cs = reinterpret_cast<float*>(smem + o_cs);          // [32][CLD3] fp32 dA rows of one pass
    double* red = reinterpret_cast<double*>(smem + o_cs);       // [4][128][3] after the last tile
    float* tab = reinterpret_cast<float*>(smem + o_cs + 32 * CLD3 * 4);      // sc, sh, sl of norm2 [3][128]; P, Q of the slice [2][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const bf16* __restrict__ G = reinterpret_cast<const bf16*>(e.G);
    const bf16* __restrict__ D = reinterpret_cast<const bf16*>(e.X);
    const bf16* __restrict__ Y = reinterpret_cast<const bf16*>(g.Xin);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    bf16* __restrict__ DU = reinterpret_cast<bf16*>(g.Gout);
    const int nb = gridDim.x;
    const int lb = swz ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const bool drop = e.drop_p > 0.f;
    const uint32_t dkey = drop_key(e.seed, e.stream_id);

    const bf16* __restrict__ Wf = reinterpret_cast<const bf16*>(g.Wfrag) + ((long)wave * 18 * 64 + lane) * 8;
    bf16x8_t bw[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) bw[i] = *reinterpret_cast<const bf16x8_t*>(Wf + i * 512);
    const int e_c8 = tid & 15, e_r0 = tid >> 4;                 // epilogue role: 8-channel chunk, rows e_r0 + 16*i of a pass
    if (tid < 128) { tab[tid] = g.sc[tid]; tab[128 + tid] = g.sh[tid]; tab[256 + tid] = g.sl[tid]; }
    if (tid < 32) { tab[384 + tid] = e.P[tid]; tab[416 + tid] = e.Q[tid]; }
    const int ec = tid & 3, er0 = tid >> 2;                     // eff role: chunk ec of rows er0 + 64*k
    float st1[8], st2[8], st3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[j] = 0.f; st2[j] = 0.f; st3[j] = 0.f; }

    auto fill_tbl = [&](int slot, int tile) {
        for (int rr = tid; rr < nr; rr += 256) tbl[slot * nr + rr] = pix_of(q, tile * TP - q.halo + rr, invWp, invHp);
    };
    const uint32_t* __restrict__ KM = e.keep;                    // keep words of the forward kernel (or nullptr: hash)
    const float dinv = 1.f / (1.f - e.drop_p);
    auto dma_raw = [&](int slot) {                              // both slices, this wave's row groups; padding rows <- zeros
        const int rsub = lane >> 2, chunk = lane & 3;
        if (KM != nullptr) {                                    // one word per row: 64 rows per instruction (lane = row), 4 B per lane
            for (int r64 = wave; r64 * 64 < nr; r64 += 4) {
                const int rr = r64 * 64 + lane;
                const int m = rr < nr ? tbl[slot * nr + rr] : -1;
                const char* sk = m >= 0 ? reinterpret_cast<const char*>(KM + m) : zeros;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sk,
                                                 (__attribute__((address_space(3))) void*)(smem + o_km + r64 * 256), 4, 0, 0);
            }
        }
        for (int rg = wave; rg * 16 < nr; rg += 4) {
            const int m = tbl[slot * nr + rg * 16 + rsub];
            const char* sg = m >= 0 ? reinterpret_cast<const char*>(G + (long)m * e.ldg + e.c_off) + chunk * 16 : zeros + chunk * 16;
            const char* sd = m >= 0 ? reinterpret_cast<const char*>(D + (long)m * e.ldx + e.c_off) + chunk * 16 : zeros + chunk * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sg,
                                             (__attribute__((address_space(3))) void*)(smem + o_rg + rg * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sd,
                                             (__attribute__((address_space(3))) void*)(smem + o_rd + rg * 1024), 16, 0, 0);
        }
    };
    auto load_y = [&](int slot, int pass, u16x8 (&yv)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tbl[slot * nr + q.halo + pass * 32 + e_r0 + 16 * i];
            yv[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (m >= 0) yv[i] = *reinterpret_cast<const u16x8*>(Y + (long)m * g.ldxin + e_c8 * 8);
        }
    };

    int cur = 0;
    if (lb < ntiles) fill_tbl(0, lb);
    __syncthreads();
    if (lb < ntiles) dma_raw(0);
    // the epilogue's Y rows travel TWO passes ahead of their use, across tile boundaries: pass p of a tile lives in yq[p]; passes
    // 0/1 of the next tile are requested during passes 2/3 of this one (one pass ahead left a pass shorter than an HBM round trip)
    u16x8 yq[4][2];
    if (lb < ntiles) { load_y(0, 0, yq[0]); load_y(0, 1, yq[1]); }
    for (int t = lb; t < ntiles; t += nb, cur ^= 1) {
        if (t + nb < ntiles) fill_tbl(cur ^ 1, t + nb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's share of tile t's raw rows has landed
        __syncthreads();                                         // ... and everybody else's; next table visible
        // eff image: 16-B chunk ec of row rr at off64(rr, ec);  eff = (G + P*x + Q) * dropout, exactly 0 on padding rows
        float cP[8], cQ[8];
        {
            const float4 p0 = *reinterpret_cast<const float4*>(tab + 384 + ec * 8), p1 = *reinterpret_cast<const float4*>(tab + 388 + ec * 8);
            const float4 q0 = *reinterpret_cast<const float4*>(tab + 416 + ec * 8), q1 = *reinterpret_cast<const float4*>(tab + 420 + ec * 8);
            cP[0] = p0.x; cP[1] = p0.y; cP[2] = p0.z; cP[3] = p0.w; cP[4] = p1.x; cP[5] = p1.y; cP[6] = p1.z; cP[7] = p1.w;
            cQ[0] = q0.x; cQ[1] = q0.y; cQ[2] = q0.z; cQ[3] = q0.w; cQ[4] = q1.x; cQ[5] = q1.y; cQ[6] = q1.z; cQ[7] = q1.w;
        }
        for (int rr = er0; rr < nr; rr += 64) {
            const int m = tbl[cur * nr + rr];
            u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m >= 0 && !TCVN_DBG_BIT(dbg, 8)) {
                const u16x8 gv = *reinterpret_cast<const u16x8*>(smem + o_rg + rr * 64 + ec * 16);
                const u16x8 xv = *reinterpret_cast<const u16x8*>(smem + o_rd + rr * 64 + ec * 16);
                const uint32_t kb = KM != nullptr ? *reinterpret_cast<const uint32_t*>(smem + o_km + rr * 4) >> (ec * 8) : 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = eff3(bf2f(gv[j]), cP[j], bf2f(xv[j]), cQ[j]);
                    if (drop) v *= KM != nullptr ? (((kb >> j) & 1u) ? dinv : 0.f)
                                                 : drop_pick(drop_bits32(dkey, m, ec * 8 + j, e.N), m, e.drop_p);      // (pixels * N < 2^32: conv3x3_dgrad_tile_ok)
                    o[j] = f2bf(v);
                }
            }
            *reinterpret_cast<u16x8*>(smem + off64(rr, ec)) = o;
        }
        __syncthreads();                                         // image complete, raw buffers free again
        if (t + nb < ntiles) dma_raw(cur ^ 1);                   // next tile's slices travel under the MFMAs + epilogue

        // the 128 positions are multiplied in two halves of 64 (two accumulator tiles live instead of four: the kernel sits at
        // the 256-register limit of two workgroups per CU); each half is followed by its two 32-row epilogue passes
#pragma unroll
        for (int half = 0; half < 2; ++half) {
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        if (!TCVN_DBG_BIT(dbg, 2))
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int base = q.halo - ((tap / 3 - 1) * q.Wp + (tap % 3 - 1)) + r + half * 64;      // source position = p - shift(tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int lr = base + mt * 32;
                    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(smem + off64(lr, 2 * ks + h));
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[tap * 2 + ks], acc[mt], 0, 0, 0);
                }
            }
        }
        // epilogue, four passes of 32 rows through the fp32 C tile: u = sc*y + sh ; dU = dA * prelu'(u) ; DU = sc*dU ;
        // sums (dU, dU*y, dA*min(u,0)) per channel
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pass = half * 2 + pp;
            u16x8 (&yc)[2] = yq[pass];
            if (pass < 2) load_y(cur, pass + 2, yq[pass + 2]);
            else if (t + nb < ntiles) load_y(cur ^ 1, pass - 2, yq[pass - 2]);      // next tile's table: filled at the top of this iteration
#pragma unroll
            for (int k = 0; k < 16; ++k) Cs[((k & 3) + 8 * (k >> 2) + 4 * h) * CLD3 + wave * 32 + r] = acc[pp][k];
            lds_barrier();
            float esc[8], esh[8], esl[8];
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const float4 a4 = *reinterpret_cast<const float4*>(tab + e_c8 * 8 + j4 * 4);
                const float4 b4 = *reinterpret_cast<const float4*>(tab + 128 + e_c8 * 8 + j4 * 4);
                const float4 c4 = *reinterpret_cast<const float4*>(tab + 256 + e_c8 * 8 + j4 * 4);
                esc[j4 * 4] = a4.x; esc[j4 * 4 + 1] = a4.y; esc[j4 * 4 + 2] = a4.z; esc[j4 * 4 + 3] = a4.w;
                esh[j4 * 4] = b4.x; esh[j4 * 4 + 1] = b4.y; esh[j4 * 4 + 2] = b4.z; esh[j4 * 4 + 3] = b4.w;
                esl[j4 * 4] = c4.x; esl[j4 * 4 + 1] = c4.y; esl[j4 * 4 + 2] = c4.z; esl[j4 * 4 + 3] = c4.w;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rr = e_r0 + 16 * i;
                const int m = tbl[cur * nr + q.halo + pass * 32 + rr];
                if (m >= 0 && !TCVN_DBG_BIT(dbg, 4)) {
                    const float4 ca = *reinterpret_cast<const float4*>(Cs + rr * CLD3 + e_c8 * 8);
                    const float4 cc = *reinterpret_cast<const float4*>(Cs + rr * CLD3 + e_c8 * 8 + 4);
                    const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cc.x, cc.y, cc.z, cc.w};
                    u16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = bf2f(yc[i][j]);
                        const float u = fmaf(y, esc[j], esh[j]);
                        const float du = u > 0.f ? cv[j] : esl[j] * cv[j];
                        st1[j] += du; st2[j] += du * y; st3[j] += u > 0.f ? 0.f : cv[j] * u;
                        o[j] = f2bf(esc[j] * du);
                    }
                    *reinterpret_cast<u16x8*>(DU + (long)m * g.ldgo + e_c8 * 8) = o;
                }
            }
            lds_barrier();
        }
        }   // half
    }
    // reduce the 16 row groups: 4 per wave by shuffles (lane bits 4, 5), then across waves through LDS
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d1 = (double)st1[j], d2 = (double)st2[j], d3 = (double)st3[j];
        d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32);
        d2 += __shfl_xor(d2, 16); d2 += __shfl_xor(d2, 32);
        d3 += __shfl_xor(d3, 16); d3 += __shfl_xor(d3, 32);
        if (lane < 16) {
            double* p = red + ((wave * 128) + e_c8 * 8 + j) * 3;
            p[0] = d1; p[1] = d2; p[2] = d3;
        }
    }
    __syncthreads();
    if (tid < 128) {
        double a = 0, b = 0, c = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 128 + tid) * 3]; b += red[(w * 128 + tid) * 3 + 1]; c += red[(w * 128 + tid) * 3 + 2]; }
        double* p = g.part + ((long)blockIdx.x * g.N + tid) * 3;
        p[0] = a; p[1] = b; p[2] = c;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Third data-gradient kernel (the one the dense layers of blocks 1-3 run): consecutive tiles per workgroup, the eff image in a ring
// of 512 rows, one barrier per tile.  k_conv3x3_dgrad2_bf16 rebuilds the whole (128 + 2 halo)-row eff image per tile (2.1x the rows
// at W = 69, each with its dropout flags and BatchNorm mean terms) and walks ten barriers per tile -- fp32 C tile exchange between the
// four waves, pass by pass -- which its ablations put at half of its time ("other": 177 of 358 us).  Here:
//   * 512 threads = 8 waves: wave (cs, ph) owns output channels [32 cs, +32) for positions [64 ph, +64) of the tile; its 18 weight
//     fragments stay in registers; two waves per SIMD, so one wave's epilogue runs under the other's MFMAs;
//   * every eff row is built ONCE, as one of the 128 new rows of the next tile: thread -> (row, 16-B chunk), slice loads requested
//     two tiles ahead, keep bits or hash;
//   * the epilogue is wave-private: the wave's 32 x 32 fp32 tile goes through its own LDS patch (no workgroup barrier), a lane then
//     owns (row, 8 channels): Y load (requested before the MFMAs), PReLU / BatchNorm backward, 16-B store, fp32 running sums.
// Row space as in the weight-gradient kernel: row 0 = padded position t0 * TP - halo, ring slot = row & 511, table entry = row & 1023.
constexpr int DG_RING = 512, DG_TBL = 1024, DG_CP = 36;
__global__ __launch_bounds__(512, 1) void k_conv3x3_dgrad3_bf16(const ConvDgradArgs g, int n_img, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const EffSrc& e = g.e;
    const PadGeom q(n_img, g.H, g.W);
    const int nrows = q.rows();
    constexpr int o_tbl = DG_RING * 64, o_tab = o_tbl + DG_TBL * 4, o_cw = o_tab + 448 * 4, o_w = o_cw + 8 * 32 * DG_CP * 4;
    int* tbl = reinterpret_cast<int*>(smem + o_tbl);
    float* tab = reinterpret_cast<float*>(smem + o_tab);        // sc, sh, sl of norm2 [3][128]; P, Q of the slice [2][32]
    double* red = reinterpret_cast<double*>(smem + o_cw);       // [8][32][3] after the last tile (aliases the C patches)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cs = wave & 3, ph = wave >> 2;
    float* Cw = reinterpret_cast<float*>(smem + o_cw) + wave * 32 * DG_CP;      // this wave's [32][DG_CP] fp32 patch
    const int r = lane & 31, h = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const bf16* __restrict__ G = reinterpret_cast<const bf16*>(e.G);
    const bf16* __restrict__ D = reinterpret_cast<const bf16*>(e.X);
    const bf16* __restrict__ Y = reinterpret_cast<const bf16*>(g.Xin);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    bf16* __restrict__ DU = reinterpret_cast<bf16*>(g.Gout);
    const uint32_t* __restrict__ KM = e.keep;
    bf16* __restrict__ EY = reinterpret_cast<bf16*>(g.ey_out);
    const int nb = gridDim.x, per = ntiles / nb, rem = ntiles % nb;
    const int t0 = blockIdx.x * per + min((int)blockIdx.x, rem), ntl = per + ((int)blockIdx.x < rem ? 1 : 0);
    const int g_org = t0 * TP - q.halo;
    const bool drop = e.drop_p > 0.f;
    const uint32_t dkey = drop_key(e.seed, e.stream_id);
    const float dinv = 1.f / (1.f - e.drop_p);

    // the 4 x 18 weight fragments (72 KB) live in LDS: in registers (72 per lane) they push the kernel past the 256 registers two waves
    // per SIMD leave each, next to the accumulators, the Y rows in flight and the running sums
    {
        const u16x8* __restrict__ wsrc = reinterpret_cast<const u16x8*>(g.Wfrag);
        u16x8* wdst = reinterpret_cast<u16x8*>(smem + o_w);
        for (int i = tid; i < 4 * 18 * 64; i += 512) wdst[i] = wsrc[i];
    }
    const char* wl = smem + o_w + (cs * 18 * 64 + lane) * 16;
    if (tid < 128) { tab[tid] = g.sc[tid]; tab[128 + tid] = g.sh[tid]; tab[256 + tid] = g.sl[tid]; }
    if (tid < 32) { tab[384 + tid] = e.P[tid]; tab[416 + tid] = e.Q[tid]; }

    // eff role of a thread: chunk ec of one row per 128-row batch
    const int ec = tid & 3, er = tid >> 2;
    auto fill_rows = [&](int row0, int n) {
        for (int i = tid; i < n; i += 512) tbl[(row0 + i) & (DG_TBL - 1)] = pix_of(q, g_org + row0 + i, invWp, invHp);
    };
    auto eff_fetch = [&](int m, u16x8& gv, u16x8& xv, uint32_t& kw) {
        const long o = (long)(m >= 0 ? m : 0);
        gv = *reinterpret_cast<const u16x8*>(G + o * e.ldg + e.c_off + ec * 8);
        xv = *reinterpret_cast<const u16x8*>(D + o * e.ldx + e.c_off + ec * 8);
        kw = *(KM != nullptr ? KM + o : reinterpret_cast<const uint32_t*>(zeros));
    };
    auto eff_store = [&](int row, int m, const u16x8& gv, const u16x8& xv, uint32_t kw) {      // eff = (G + P*x + Q) * dropout; 0 on padding rows
        u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m >= 0) {
            const float4 p0 = *reinterpret_cast<const float4*>(tab + 384 + ec * 8), p1 = *reinterpret_cast<const float4*>(tab + 388 + ec * 8);
            const float4 q0 = *reinterpret_cast<const float4*>(tab + 416 + ec * 8), q1 = *reinterpret_cast<const float4*>(tab + 420 + ec * 8);
            const float cP[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w}, cQ[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const uint32_t kb = kw >> (ec * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = eff3(bf2f(gv[j]), cP[j], bf2f(xv[j]), cQ[j]);
                if (drop) v *= KM != nullptr ? (((kb >> j) & 1u) ? dinv : 0.f)
                                             : drop_pick(drop_bits32(dkey, m, ec * 8 + j, e.N), m, e.drop_p);      // (pixels * N < 2^32: launcher)
                o[j] = f2bf(v);
            }
        }
        *reinterpret_cast<u16x8*>(smem + off64(row & (DG_RING - 1), ec)) = o;
        if (EY != nullptr && m >= 0) *reinterpret_cast<u16x8*>(EY + (long)m * 32 + ec * 8) = o;      // the weight gradient's operand, built once
    };

    if (ntl > 0) fill_rows(0, nrows + min(ntl - 1, 2) * TP);                // tile 0's rows, the new rows of tiles 1 and 2
    __syncthreads();
    if (ntl > 0) {
        for (int row = er; row < nrows; row += 128) {                       // tile 0: all its rows
            u16x8 g0, x0; uint32_t k0;
            const int m = tbl[row & (DG_TBL - 1)];
            eff_fetch(m, g0, x0, k0);
            eff_store(row, m, g0, x0, k0);
        }
    }
    u16x8 gv, xv;
    uint32_t kw = 0;
    int mm = -1;
    if (ntl > 1) { mm = tbl[(nrows + er) & (DG_TBL - 1)]; eff_fetch(mm, gv, xv, kw); }       // tile 1's new rows
    else { gv = u16x8{0, 0, 0, 0, 0, 0, 0, 0}; xv = gv; }
    __syncthreads();

    // epilogue role of a lane in its wave's 32 x 32 patch: rows el and el + 16, channel chunk e4
    const int e4 = lane & 3, el = lane >> 2;
    float st1[8], st2[8], st3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st1[j] = 0.f; st2[j] = 0.f; st3[j] = 0.f; }
    // Y rows of the wave's two 32-position passes: slot mt is reloaded with the NEXT tile's rows as soon as this tile's pass mt has
    // consumed it, so a request has a whole tile to arrive (requested at the top of its own tile it had the MFMA phase only: ~1 us
    // against 2-4 us of HBM latency under load, and the first version of this kernel ran 45 % slower than the kernel it replaces)
    u16x8 yq[2][2];
    auto y_fetch = [&](int tile_row, int mt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tbl[(tile_row + q.halo + ph * 64 + mt * 32 + el + 16 * i) & (DG_TBL - 1)];
            const long o = (long)(m >= 0 ? m : 0);
            yq[mt][i] = *reinterpret_cast<const u16x8*>(Y + o * g.ldxin + cs * 32 + e4 * 8);
        }
    };
    if (ntl > 0) { y_fetch(0, 0); y_fetch(0, 1); }

    for (int il = 0; il < ntl; ++il) {
        const int trow = il * TP;                                            // first image row of this tile
        // ---- next tiles: eff rows of tile il+1 into the ring, slice loads of tile il+2, table of tile il+3 ----
        if (il + 1 < ntl) eff_store(trow + nrows + er, mm, gv, xv, kw);
        if (il + 2 < ntl) { mm = tbl[(trow + TP + nrows + er) & (DG_TBL - 1)]; eff_fetch(mm, gv, xv, kw); }
        if (il + 3 < ntl) fill_rows(trow + 2 * TP + nrows, TP);
        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[mt][k] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int base = trow + q.halo - ((tap / 3 - 1) * q.Wp + (tap % 3 - 1)) + ph * 64 + r;      // source position = p - shift(tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(wl + (tap * 2 + ks) * 1024);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(smem + off64((base + mt * 32) & (DG_RING - 1), 2 * ks + h));
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mt], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: u = sc*y + sh ; dU = dA * prelu'(u) ; DU = sc*dU ; sums (dU, dU*y, dA*min(u,0)) per channel ----
        float esc[8], esh[8], esl[8];                                        // norm2's table of this lane's 8 channels (re-read per tile: 24 registers
#pragma unroll                                                               // the MFMA phase needs more)
        for (int j4 = 0; j4 < 2; ++j4) {
            const float4 a4 = *reinterpret_cast<const float4*>(tab + cs * 32 + e4 * 8 + j4 * 4);
            const float4 b4 = *reinterpret_cast<const float4*>(tab + 128 + cs * 32 + e4 * 8 + j4 * 4);
            const float4 c4 = *reinterpret_cast<const float4*>(tab + 256 + cs * 32 + e4 * 8 + j4 * 4);
            esc[j4 * 4] = a4.x; esc[j4 * 4 + 1] = a4.y; esc[j4 * 4 + 2] = a4.z; esc[j4 * 4 + 3] = a4.w;
            esh[j4 * 4] = b4.x; esh[j4 * 4 + 1] = b4.y; esh[j4 * 4 + 2] = b4.z; esh[j4 * 4 + 3] = b4.w;
            esl[j4 * 4] = c4.x; esl[j4 * 4 + 1] = c4.y; esl[j4 * 4 + 2] = c4.z; esl[j4 * 4 + 3] = c4.w;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int k = 0; k < 16; ++k) Cw[((k & 3) + 8 * (k >> 2) + 4 * h) * DG_CP + r] = acc[mt][k];
            // (the same wave reads what it wrote: LDS operations of a wave execute in order, no barrier)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = tbl[(trow + q.halo + ph * 64 + mt * 32 + el + 16 * i) & (DG_TBL - 1)];
                const float4 ca = *reinterpret_cast<const float4*>(Cw + (el + 16 * i) * DG_CP + e4 * 8);
                const float4 cc = *reinterpret_cast<const float4*>(Cw + (el + 16 * i) * DG_CP + e4 * 8 + 4);
                const float cv[8] = {ca.x, ca.y, ca.z, ca.w, cc.x, cc.y, cc.z, cc.w};
                u16x8 o;
                const bool ok = m >= 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = bf2f(yq[mt][i][j]);
                    const float u = fmaf(y, esc[j], esh[j]);
                    const float c = ok ? cv[j] : 0.f;
                    const float du = u > 0.f ? c : esl[j] * c;
                    st1[j] += du; st2[j] = fmaf(du, y, st2[j]); st3[j] += u > 0.f ? 0.f : c * u;
                    o[j] = f2bf(esc[j] * du);
                }
                if (ok) *reinterpret_cast<u16x8*>(DU + (long)m * g.ldgo + cs * 32 + e4 * 8) = o;
            }
            if (il + 1 < ntl) y_fetch(trow + TP, mt);                        // this slot's rows of the next tile
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // tile il+1's rows and the table entries are in place
    }
    // per-channel sums: lanes with equal e4 hold the same 8 channels (fold lane bits 2..5), the two position halves through LDS
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double d1 = (double)st1[j], d2 = (double)st2[j], d3 = (double)st3[j];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o); }
        if (lane < 4) {
            double* p = red + ((wave * 32) + e4 * 8 + j) * 3;
            p[0] = d1; p[1] = d2; p[2] = d3;
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int c_cs = tid >> 5, c_in = tid & 31;
        double a = 0, b = 0, c = 0;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const double* p = red + (((hh * 4 + c_cs) * 32) + c_in) * 3;
            a += p[0]; b += p[1]; c += p[2];
        }
        double* p = g.part + ((long)blockIdx.x * g.N + tid) * 3;
        p[0] = a; p[1] = b; p[2] = c;
    }
}
static long cu_tiles() { return 256; }      // workgroups of a full persistent grid (one per CU)
size_t dgrad3_smem() { return size_t(DG_RING) * 64 + DG_TBL * 4 + 448 * 4 + 8 * 32 * DG_CP * 4 + 4 * 18 * 1024; }
bool dgrad3_ok(const ConvDgradArgs& a, const PadGeom& q) {
    [[maybe_unused]] static const int dbg = TCVN_KNOB_INT("TCVN_DBG");
    // from 8 tiles per workgroup on: below that its prologue (72 KB of weights into LDS, the whole first eff image) costs more than the
    // ring saves (block 3, 816 tiles: 37 us against 32 us for the two-workgroup kernel; block 2, 3 600 tiles: 94 against 102)
    return a.zeros != nullptr && a.e.N == 32 && (a.e.c_off & 7) == 0 && q.rows() + TP + 8 <= DG_RING && !TCVN_DBG_BIT(dbg, 4096) &&
           (q.tiles() >= 8 * cu_tiles() || TCVN_DBG_BIT(dbg, 8192)) &&                 // TCVN_DBG=8192 (validation build): at any size
           (reinterpret_cast<uintptr_t>(a.e.G) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.e.X) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(a.Xin) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.Gout) & 15) == 0;      // TCVN_DBG=4096: the two-workgroup kernel
}

size_t dgrad2_smem(const PadGeom& q) { const size_t nr = (q.rows() + 15) & ~15; return 3 * nr * 64 + 2 * nr * 4 + 32 * 132 * 4 + 448 * 4 + ((nr + 63) & ~size_t(63)) * 4; }
size_t dgrad_smem(const PadGeom& q) { const size_t r4 = (q.rows() + 3) & ~3; return r4 * 68 + 128 * 24 + 64 * 132 * 4; }
int tile_grid2(long ntiles) {           // two workgroups per CU
    if (ntiles >= 512) return 512;
    if (ntiles >= 8) return (int)(ntiles / 8 * 8);
    return (int)ntiles;
}

// ring + two eff tiles + table; the ring must hold a tile's rows and the 128 rows being fetched for the next one
size_t wgrad_smem(const PadGeom& q) { const size_t r4 = (q.rows() + 3) & ~3; return r4 + TP + 8 <= 512 ? size_t(WG_RING_BYTES) + 2 * TP * 64 + 1024 * 4 + 3 * 128 * 4 : size_t(1) << 30; }

int tile_grid(long ntiles) {            // one persistent workgroup per CU
    [[maybe_unused]] static const int dbg = TCVN_KNOB_INT("TCVN_DBG");
    if (TCVN_DBG_BIT(dbg, 16)) return (int)ntiles;                 // debug: one tile per workgroup
    if (ntiles >= 256) return 256;
    if (ntiles >= 8) return (int)(ntiles / 8 * 8);
    return (int)ntiles;
}

}  // namespace

// n_img is recovered from M = n*H*W
static bool tile_disabled() {
    static const bool off = TCVN_KNOB_SET("TCVN_DISABLE_TILE");      // validation switch: force the generic kernels
    return off;
}

bool conv3x3_tile_enabled() { return !tile_disabled(); }

bool conv3x3_tile_ok(const ConvFwdArgs& a) {
    if (tile_disabled()) return false;
    if (a.Wfrag == nullptr || (reinterpret_cast<uintptr_t>(a.Wfrag) & 15) || a.Aact == nullptr || a.zeros == nullptr) return false;
    if (a.mode != MODE_BF16 || a.amode != A_3X3 || a.C != 128 || a.lda != 128 || a.N > 32 || a.Kp != 1152) return false;
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.Wk) & 15)) return false;
    if (a.M % (a.H * a.W) != 0) return false;
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return q.gtot < (1L << 24) && fwd_smem(q) <= 160 * 1024;
}
int conv3x3_tile_nblk(const ConvFwdArgs& a) {
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return tile_grid(q.tiles());
}
static bool fwd_pair_ok(const ConvFwdArgs& a, const PadGeom& q, int dbg) {
    return fwd_pair_smem(q) <= 160 * 1024 && 4 * TP + q.halo + q.Wp + 1 < PAIR_TBL && (long)a.M * a.N < (1L << 32) && !TCVN_DBG_BIT(dbg, 32) &&
           !TCVN_DBG_BIT(dbg, 64);
}
bool conv3x3_act_fusable(const ConvFwdArgs& a) {
    if (!conv3x3_tile_ok(a) || a.sc == nullptr || a.sh == nullptr || a.sl == nullptr) return false;
    [[maybe_unused]] static const int dbg = TCVN_KNOB_INT("TCVN_DBG");
    static const bool off = TCVN_KNOB_SET("TCVN_NO_ACT_FUSE");      // validation build: keep the materialised activation (A/B and variant tests)
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    // forward: only the pair kernel activates in LDS; backward: the weight-gradient tile kernel (same geometry conditions as conv3x3_wgrad_tile_ok)
    return !off && fwd_pair_ok(a, q, dbg) && wgrad_smem(q) <= 160 * 1024;
}
bool conv3x3_fwd_pair(const ConvFwdArgs& a) {          // conv_fwd(a) runs k_conv3x3_fwd_pair_bf16 (the kernel that honours lf / isum_out)
    if (!conv3x3_tile_ok(a)) return false;
    [[maybe_unused]] static const int dbg = TCVN_KNOB_INT("TCVN_DBG");
    return fwd_pair_ok(a, PadGeom(a.M / (a.H * a.W), a.H, a.W), dbg);
}
bool conv3x3_fwd_writes_keep(const ConvFwdArgs& a) {
    if (a.keep_out == nullptr || a.drop_p <= 0.f || !conv3x3_tile_ok(a)) return false;
    [[maybe_unused]] static const int dbg = TCVN_KNOB_INT("TCVN_DBG");
    return fwd_pair_ok(a, PadGeom(a.M / (a.H * a.W), a.H, a.W), dbg);
}
int conv3x3_fwd_tile(const ConvFwdArgs& a, hipStream_t st) {
    const int n_img = a.M / (a.H * a.W);
    const PadGeom q(n_img, a.H, a.W);
    const int ntiles = (int)q.tiles();
    const int nb = tile_grid(ntiles);
    const size_t smem = fwd_smem(q);
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_fwd_bf16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024));
        attr = true;
    }
    ProfScope ps("k_conv3x3_fwd_bf16", 2.0 * a.M * (double)a.N * a.K, (double)a.M * 2.0 * (a.C + a.N), st);   // read 128 ch, write N ch
    ConvFwdArgs b = a;
    static const int dbg = TCVN_KNOB_INT("TCVN_DBG");
    b.dbg = dbg;
    if (fwd_pair_ok(a, q, dbg)) {   // two waves per SIMD, taps split (TCVN_DBG=64: one-wave ring kernel)
        static bool attr3 = false;
        if (!attr3) {
            TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_fwd_pair_bf16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024));
            attr3 = true;
        }
        hipLaunchKernelGGL(k_conv3x3_fwd_pair_bf16, dim3(nb), dim3(512), fwd_pair_smem(q), st, b, n_img, ntiles, fwd_pair_ring(q));
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.lf.isum != nullptr || a.isum_out != nullptr) return -2;            // only the pair kernel derives / adds link-free statistics (conv3x3_fwd_pair)
    if (!a.act_fused && ((q.rows() + 3) & ~3) + TP <= RING && !TCVN_DBG_BIT(dbg, 32)) {       // consecutive tiles per workgroup, ring image (TCVN_DBG=32: strips)
        static bool attr2 = false;
        if (!attr2) {
            TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_fwd_ring_bf16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024));
            attr2 = true;
        }
        hipLaunchKernelGGL(k_conv3x3_fwd_ring_bf16, dim3(nb), dim3(256), fwd_ring_smem(), st, b, n_img, ntiles);
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.act_fused) return -2;                                               // only the pair kernel activates in LDS (conv3x3_act_fusable)
    hipLaunchKernelGGL(k_conv3x3_fwd_bf16, dim3(nb), dim3(256), smem, st, b, n_img, ntiles, (nb >= 8 && nb % 8 == 0 && !TCVN_DBG_BIT(dbg, 8)) ? 1 : 0);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn

namespace tcvn {
using namespace t3;

bool conv3x3_wgrad_tile_ok(const ConvWgradArgs& a) {
    const ConvFwdArgs& fa = a.fa;
    if (!conv3x3_tile_enabled() || a.mode != MODE_BF16 || fa.amode != A_3X3 || fa.C != 128 || a.e.N > 32) return false;
    if (fa.Aact == nullptr || fa.zeros == nullptr || (a.e.ldg & 7) || (a.e.ldx & 7) || (a.e.c_off & 1)) return false;
    if (fa.M % (fa.H * fa.W) != 0) return false;
    const PadGeom q(fa.M / (fa.H * fa.W), fa.H, fa.W);
    return q.gtot < (1L << 24) && wgrad_smem(q) <= 160 * 1024 && (long)fa.M * a.e.N < (1L << 32);
}

int conv3x3_wgrad_tile(const ConvWgradArgs& a, hipStream_t st) {
    const int n_img = a.fa.M / (a.fa.H * a.fa.W);
    const PadGeom q(n_img, a.fa.H, a.fa.W);
    const int ntiles = (int)q.tiles();
    const int nb = tile_grid(ntiles);
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_wgrad_bf16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024));
        attr = true;
    }
    if (a.slab == nullptr || (long)nb * (9 * 128 * 32 + 32) * 4 > a.slab_bytes || a.dbias == nullptr) return -3;
    {
        ProfScope ps("k_conv3x3_wgrad_bf16", 2.0 * a.fa.M * (double)a.e.N * a.fa.K, (double)a.fa.M * 2.0 * (a.fa.C + 2 * a.e.N), st);   // YA + (G, x) slices
        hipLaunchKernelGGL(k_conv3x3_wgrad_bf16, dim3(nb), dim3(512), wgrad_smem(q), st, a, n_img, ntiles);
        TCVN_LAUNCH_CHECK();
    }
    // weight partials [nb][9*128*32] -> dWk and bias partials [nb][32] -> dbias[0:N) (32-wide rows, zero beyond N): one launch
    SlabJob jb{};
    if (a.dbias != nullptr) jb = slab_job(a.slab + (long)nb * (9 * 128 * 32), nb, a.e.N, a.dbias, 32);
    if (a.deferred != nullptr) { a.deferred[0] = slab_job(a.slab, nb, 9 * 128 * 32, a.dWk, 0); a.deferred[1] = jb; return 0; }
    return slab_reduce2(slab_job(a.slab, nb, 9 * 128 * 32, a.dWk, 0), jb, st);
}

}  // namespace tcvn

namespace tcvn {
using namespace t3;

bool conv3x3_dgrad_tile_ok(const ConvDgradArgs& a) {
    if (!conv3x3_tile_enabled() || a.mode != MODE_BF16 || a.dmode != DG_3X3 || a.N != 128 || a.e.N > 32 || a.Kp != 288) return false;
    if (a.Wfrag == nullptr || a.accumulate || a.ldxin != 128 || a.ldgo != 128) return false;
    if ((a.e.ldg & 7) || (a.e.ldx & 7) || (a.e.c_off & 1) || a.M % (a.H * a.W) != 0) return false;
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return q.gtot < (1L << 24) && (long)a.M * a.e.N < (1L << 32);
}
bool conv3x3_dgrad_writes_ey(const ConvDgradArgs& a) {
    if (a.ey_out == nullptr || !conv3x3_dgrad_tile_ok(a)) return false;
    return dgrad3_ok(a, PadGeom(a.M / (a.H * a.W), a.H, a.W));
}
int conv3x3_dgrad_tile_nblk(const ConvDgradArgs& a) {
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return dgrad3_ok(a, q) ? tile_grid(q.tiles()) : tile_grid2(q.tiles());      // one 512-thread workgroup per CU, or two of 256
}
int conv3x3_dgrad_tile(const ConvDgradArgs& a, hipStream_t st) {
    const int n_img = a.M / (a.H * a.W);
    const PadGeom q(n_img, a.H, a.W);
    const int ntiles = (int)q.tiles();
    ProfScope ps("k_conv3x3_dgrad_bf16", 2.0 * a.M * (double)a.N * 9 * a.e.N, (double)a.M * 2.0 * (2 * a.e.N + 2 * a.N), st);   // (G, x) slices in; Y in, DU out
    if (dgrad3_ok(a, q)) {                 // consecutive tiles, eff ring, wave-private epilogue
        static bool attr3 = false;
        if (!attr3) {
            TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_dgrad3_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr3 = true;
        }
        hipLaunchKernelGGL(k_conv3x3_dgrad3_bf16, dim3(tile_grid(ntiles)), dim3(512), dgrad3_smem(), st, a, n_img, ntiles);
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    const int nb = tile_grid2(ntiles);
    // pipelined variant: needs the concat slice 16-B aligned for the LDS-DMA and all 32 channels present
    if (a.zeros != nullptr && a.e.N == 32 && (a.e.c_off & 7) == 0 && dgrad2_smem(q) <= 80 * 1024 &&
        (reinterpret_cast<uintptr_t>(a.e.G) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.e.X) & 15) == 0) {
        static bool attr = false;
        if (!attr) {
            TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_dgrad2_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            attr = true;
        }
        static const int dbgd = TCVN_KNOB_INT("TCVN_DBG");
        hipLaunchKernelGGL(k_conv3x3_dgrad2_bf16, dim3(nb), dim3(256), dgrad2_smem(q), st, a, n_img, ntiles, (nb >= 8 && nb % 8 == 0) ? 1 : 0, dbgd);
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_conv3x3_dgrad_bf16, dim3(nb), dim3(256), dgrad_smem(q), st, a, n_img, ntiles,
                       (nb >= 8 && nb % 8 == 0) ? 1 : 0);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn

#ifdef TCVN_DEBUG_KNOBS
extern "C" void tcvn_debug_pair_phases(unsigned long long* out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(tcvn::g_pair_ph), 16 * 8);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(tcvn::g_pair_ph), z, 16 * 8); }
}
extern "C" void tcvn_debug_wgrad_phases(unsigned long long* out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(tcvn::g_wg_ph), 16 * 8);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(tcvn::g_wg_ph), z, 16 * 8); }
}
#endif
