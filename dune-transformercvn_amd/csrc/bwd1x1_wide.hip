// Fused backward of a bottleneck 1x1 convolution for WIDE layers (128 < cin <= 512, cin % 4 == 0: dense blocks 2-4), round 5.
// A VALIDATED VARIANT, NOT THE PRODUCT PATH: it moves less data than k_bwd1x1_fused_bf16 and takes 20-30 % longer (numbers below).
// Reference: autograd of Bottleneck.bottleneck_block = BN - PReLU - conv1 (layers/dense_net.py:18-27).  Same arithmetic, expression
// by expression, as k_bwd1x1_fused_bf16 (bwd1x1_fused.hip); what differs is who walks the 128-column slices of the cin input channels:
//
//   k_bwd1x1_fused_bf16   one workgroup per (64-pixel tile, slice): gridDim.y slices, each workgroup re-reads DU / Y and rebuilds
//                         EY = bf16(DU + PY*Y + QY) for its own slice -- two to four times per pixel in blocks 3-5 (306 MB of HBM / L2
//                         traffic per average launch against 190 MB of strict bytes).  Two workgroups per CU, 2 waves per SIMD.
//   this kernel           one workgroup per TILE walks all NS slices: DU / Y are fetched and EY is formed ONCE per pixel; the NS
//                         weight-gradient tiles (NS x 128 x 128 fp32 = NS x 64 registers per lane) stay in the accumulation registers
//                         for the whole launch -- which is why a workgroup has a CU to itself (one wave per SIMD: 512 registers per lane).
//
// With one wave per SIMD nothing hides a memory round trip, so every operand travels ahead of its use on a fixed schedule:
// DU / Y one TILE ahead (E double buffered), the x slices three SLICES ahead (ring of three LDS slots), the slice's weight fragments
// one slice ahead and its G rows two slices ahead (registers).  All of these are issued by hand (inline assembly: LDS-DMA, register
// loads, the G stores) in an order that is the SAME for every wave and every iteration -- lanes / waves without real work fetch the zero
// page or store to a dump row -- so that the `s_waitcnt vmcnt(N)` in front of each consumer can carry a compile-time N (vmcnt retires in
// issue order).  Barriers inside the pipeline are bare `s_barrier`s behind an LDS-only wait.
//
// Measured (MI355X, config-2 step, same box, tools/r05_ab2.sh + tools/per_block_trace.sh; profiles/r05_wide_variant_per_block.txt):
// every launch it takes over is slower -- prong block 2 (two slices, 6 664 tiles) 235 us, block 3 two / three / four slices 70 / 101 / 130 us,
// against 97 / 49 / 59 us AVERAGES of the per-slice kernel over the same launch groups (which include the event embedder's small launches):
// 5.21 vs 4.46 ms per step in total, step 19.58 vs 19.18 ms.  4.6 us per (tile, slice) is ~10 000 cycles for ~1 400 instructions of one wave:
// the memory schedule works (no vmcnt(0) drain, no scratch: tools/check_asm_loads.sh), but with ONE wave per SIMD every LDS round trip, every
// dependent MFMA chain and every VALU dependency of the element-wise phases is exposed, and the phases of the four lock-stepped waves cannot
// overlap MFMA with VALU work the way two independent workgroups per CU do.  The per-slice kernel's redundant DU / Y reads are L2 / MALL hits
// that its second resident workgroup hides; the traffic this variant saves was not what bound the launch.
//
// Hand-issued loads and the compiler: a register the compiler believes loaded may be copied or spilled by it BEFORE the wait (seen twice
// while this file was written: spills of the G-row registers into AGPRs right behind the load).  tools/check_asm_loads.sh replays the
// generated ISA's main loop against the vmcnt queue and fails the build (Makefile: build/bwd1x1_wide.chk) on any such access.
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int ROWS = 64;
constexpr int TILE = ROWS * 256;                 // one [64][128] bf16 operand tile
constexpr int CLD = 132;                         // C tile leading dimension (floats), padded
constexpr int OFF_E = 0;                         // E[2]: DU -> EY, double buffered over tiles
constexpr int OFF_Y = 2 * TILE;                  // Y of the current tile (dead once EY is formed)
constexpr int OFF_X = 3 * TILE;                  // X[3]: ring of x slices (raw -> activated in place)
constexpr int OFF_C = 6 * TILE;                  // fp32 C tile [64][CLD] of the data gradient
constexpr int OFF_TAB = OFF_C + ROWS * CLD * 4;  // floats: PY[128], QY[128], sc[512], sh[512], sl[512]
constexpr int SMEM_BYTES = OFF_TAB + (256 + 3 * 512) * 4;      // 139 264 B

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8_t tr_frag(const char* smem_base, int off_lo, int off_hi) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_hi));
    struct { s16x4 a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8_t, pr);
}

// ---- hand-issued vector memory operations (every one counts once in vmcnt, in this order) ----
__device__ __forceinline__ void dma16(const char* src, unsigned lds_addr) {      // 16 B per lane -> LDS rows at lds_addr (wave-uniform) + lane * 16
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory");      // (m0: nothing else here uses it)
}
__device__ __forceinline__ void gload16(u16x8& v, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload8(u16x4& v, const char* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void gstore8(char* p, const u16x4& v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// after a vmwait: ties the registers an asm load wrote to the program order (the compiler must not move their first use above the wait)
__device__ __forceinline__ void pin(u16x8& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(u16x4& a) { asm volatile("" : "+v"(a)); }
// ... and an accumulator to the phase that updates it: with several basic blocks in the loop body the compiler otherwise SINKS the whole
// accumulation chain (and every per-row operand it needs: ~100 registers per slice) to the loop latch, where the next use is
__device__ __forceinline__ void pin(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pinp(const char*& a) { asm volatile("" : "+v"(a)); }

template <int NS>
__global__ __launch_bounds__(256, 1) void k_bwd1x1_wide_bf16(const Bwd1x1Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Cs = reinterpret_cast<float*>(smem + OFF_C);
    float* tab = reinterpret_cast<float*>(smem + OFF_TAB);
    double* red = reinterpret_cast<double*>(smem);                          // [4][128][3] (after the last tile, over E)
    float* csum = reinterpret_cast<float*>(smem + 4 * 128 * 3 * 8);          // [4][128]
    // register sets of the G rows in flight (two slices ahead): slice s of every tile uses set s % NSETS, a compile-time index.  With an even
    // NS two sets alternate (the set just consumed is refilled); NS = 3 takes three (slice s refills set (s + 2) % 3, consumed a slice ago)
    constexpr int NSETS = (NS & 1) ? 3 : 2;
    static_assert(NS == 2 || NS == 3 || NS == 4, "slice count");

    const int tid = threadIdx.x;
    const int N = g.cin;
    const char* __restrict__ DU = reinterpret_cast<const char*>(g.DU);
    const char* __restrict__ Yp = reinterpret_cast<const char*>(g.Y);
    const char* __restrict__ Xp = reinterpret_cast<const char*>(g.Xin);
    char* __restrict__ Gp = reinterpret_cast<char*>(g.Gout);
    const char* __restrict__ zeros = reinterpret_cast<const char*>(g.zeros);
    const char* __restrict__ Wfr = reinterpret_cast<const char*>(g.Wfrag);
    char* dump0 = reinterpret_cast<char*>(g.slab + (long)blockIdx.x * 128 * g.ldc);      // scratch row of masked G stores (the slab is written last)
    const long mtiles = (g.M + ROWS - 1) / ROWS;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    if (tid < 128) { tab[tid] = g.PY[tid]; tab[128 + tid] = g.QY[tid]; }
    for (int i = tid; i < 512; i += 256) {
        const bool ok = i < N;
        tab[256 + i] = ok ? g.sc[i] : 0.f; tab[256 + 512 + i] = ok ? g.sh[i] : 0.f; tab[256 + 1024 + i] = ok ? g.sl[i] : 0.f;
    }
    float st1[NS][4], st2[NS][4], st3[NS][4], cs[8];
    f32x16 accW[NS][2][2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[s][j] = 0.f; st2[s][j] = 0.f; st3[s][j] = 0.f; }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) accW[s][a][b][e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;

    // ---- request helpers (t = the caller's opaque copy of the thread index: addresses are rebuilt per use, not kept in registers across the loop).
    // A tile is named by (m0, rows) with rows = its live pixel rows (0 for the requests past the workgroup's last tile).  Both candidate
    // addresses of a masked request are formed and SELECTED (pinned: the compiler otherwise branches around the address arithmetic, and the
    // dozens of small basic blocks that makes defeat its scheduling limits).
    // wave w issues row groups w, w + 4, w + 8, w + 12 (4 rows = 1 KiB each); lane -> (row = lane >> 4, slot = lane & 15)
    auto dma_rows128 = [&](int t, const char* base128, long m0, int rows, int buf) {   // a [64][128] tile of a 256-B-row matrix (DU or Y)
        const int d_rsub = (t >> 4) & 3, d_slot = t & 15, wv = t >> 6;
        const char* z = zeros + (d_slot << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rg = wv + 4 * i, r = rg * 4 + d_rsub;
            const char* real = base128 + (m0 + r) * 256 + ((d_slot ^ swz16(r)) << 4);
            pinp(real);
            dma16(r < rows ? real : z, __builtin_amdgcn_readfirstlane(lds0 + buf + rg * 1024));
        }
    };
    auto dma_x = [&](int t, long m0, int rows, int n0, int buf) {            // x slice [64][128] of the concat buffer (row pitch ldx)
        const int d_rsub = (t >> 4) & 3, d_slot = t & 15, wv = t >> 6;
        const char* z = zeros + (d_slot << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rg = wv + 4 * i, r = rg * 4 + d_rsub;
            const int col = n0 + ((d_slot ^ swz16(r)) << 3);
            const char* real = Xp + ((m0 + r) * g.ldx + col) * 2;
            pinp(real);
            dma16((r < rows && col < N) ? real : z, __builtin_amdgcn_readfirstlane(lds0 + buf + rg * 1024));
        }
    };
    // element-wise role in the epilogue: columns 4 c4 .. 4 c4 + 3 of the slice, rows r0 + 8 i (a half wave covers one 256-B row)
    auto load_g = [&](int t, long m0, int rows, int n0, u16x4 (&pv)[8]) {    // masked lanes read the zero page
        const int c4 = t & 31, r0 = t >> 5;
        const char* z = zeros + (c4 << 3);
        const char* grow = Gp + ((m0 + r0) * g.ldg + n0 + c4 * 4) * 2;
        const long gstride = g.ldg * 16;
        const bool col_ok = n0 + c4 * 4 < N;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char* real = grow + i * gstride;
            pinp(real);
            gload8(pv[i], (col_ok && r0 + 8 * i < rows) ? real : z);
        }
    };
    auto load_w = [&](int t, int s, u16x8 (&w)[8]) {                         // the wave's eight weight fragments of slice s (K = 128)
        const int wv = t >> 6;
        const int sw = s * 128 + wv * 32 < N ? s * 4 + wv : wv;              // a wave whose columns lie beyond cin (no MFMA) re-reads slice 0: the
        const char* p = Wfr + (((long)sw * 8 * 64 + (t & 63)) << 4);         // fragment matrix ends with the layer's last 32-column group
#pragma unroll
        for (int i = 0; i < 8; ++i) gload16(w[i], p + i * 1024);
    };
    auto rows_of = [&](long m0) { const long r = g.M - m0; return (int)(r < ROWS ? r : ROWS); };

    // ---- the sequence of (tile, slice) pairs of this workgroup; requests past the end fetch the zero page / slice 0 (uniform counts) ----
    const long t_first = blockIdx.x, t_step = gridDim.x;
    auto tile_of = [&](long lt) { return t_first + lt * t_step; };           // global tile of local tile lt (may be >= mtiles: then m0 >= M)
    const long ntl = t_first < mtiles ? (mtiles - t_first + t_step - 1) / t_step : 0;
    u16x8 bw[8];
    u16x4 pgv[NSETS][8];

    if (ntl > 0) {
        // prologue: everything tile 0 / slices 0..2 need, then one full drain
        const long mA = tile_of(0) * ROWS;
        const int rA = rows_of(mA);
        dma_rows128(tid, DU, mA, rA, OFF_E);
        dma_rows128(tid, Yp, mA, rA, OFF_Y);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int lt = k / NS, s = k % NS;
            const long mk = lt < ntl ? tile_of(lt) * ROWS : g.M;
            dma_x(tid, mk, rows_of(mk), s * 128, OFF_X + k * TILE);
        }
        load_w(tid, 0, bw);
        load_g(tid, mA, rA, 0, pgv[0]);
        load_g(tid, mA, rA, 128, pgv[1]);                                    // NS >= 2
        vmwait<0>();
#pragma unroll
        for (int i = 0; i < 8; ++i) { pin(bw[i]); pin(pgv[0][i]); pin(pgv[1][i]); }
    }
    __syncthreads();

    int xslot = 0;                                                           // ring slot of the current slice
    for (long lt = 0; lt < ntl; ++lt) {
        const long m0 = tile_of(lt) * ROWS;
        const int rows = rows_of(m0);
        const int ebuf = OFF_E + (int)(lt & 1) * TILE;
        int t_o = tid;                                                       // opaque copy: keeps the phases' addresses out of loop-invariant registers
        asm volatile("" : "+v"(t_o));
        // ---- (A) DU / Y of this tile have landed (requested a tile ago: behind them the 28 NS operations of the last tile's slices)
        if (lt > 0) vmwait<56>();
        lds_barrier();
        // ---- EY = bf16(DU + PY*Y + QY) in place of DU (rows beyond M: zero), column sums for the bias gradient
        {
            const int c8 = t_o & 15, c_r0 = t_o >> 4;
            const int e_off = c_r0 * 256 + ((c8 ^ swz16(c_r0)) << 4);
            float pP[8], pQ[8];
            {
                const float4 a = *reinterpret_cast<const float4*>(tab + c8 * 8), b = *reinterpret_cast<const float4*>(tab + c8 * 8 + 4);
                pP[0] = a.x; pP[1] = a.y; pP[2] = a.z; pP[3] = a.w; pP[4] = b.x; pP[5] = b.y; pP[6] = b.z; pP[7] = b.w;
                const float4 c = *reinterpret_cast<const float4*>(tab + 128 + c8 * 8), d = *reinterpret_cast<const float4*>(tab + 128 + c8 * 8 + 4);
                pQ[0] = c.x; pQ[1] = c.y; pQ[2] = c.z; pQ[3] = c.w; pQ[4] = d.x; pQ[5] = d.y; pQ[6] = d.z; pQ[7] = d.w;
            }
#pragma unroll
            for (int i = 0; i < ROWS / 16; ++i) {
                const int off = e_off + i * 4096;
                const u16x8 dv = *reinterpret_cast<const u16x8*>(smem + ebuf + off);
                const u16x8 yv = *reinterpret_cast<const u16x8*>(smem + OFF_Y + off);
                const bool live = c_r0 + 16 * i < rows;
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = eff3(bf2f(dv[j]), pP[j], bf2f(yv[j]), pQ[j]);
                    o[j] = live ? f2bf(t) : (bf16)0;
                    cs[j] += bf2f(o[j]);
                }
                *reinterpret_cast<u16x8*>(smem + ebuf + off) = o;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) pin(cs[j]);
        }
        lds_barrier();                                                       // (B) EY complete; the Y buffer and the other E buffer are free
        // T1, T2: the next tile's DU and Y (past the end: rows >= M -> the zero page)
        {
            const long mn = lt + 1 < ntl ? tile_of(lt + 1) * ROWS : g.M;
            const int rn = rows_of(mn);
            dma_rows128(t_o, DU, mn, rn, OFF_E + (TILE - (ebuf - OFF_E)));
            dma_rows128(t_o, Yp, mn, rn, OFF_Y);
        }

#pragma unroll
        for (int s = 0; s < NS; ++s) {
            asm volatile("" : "+v"(t_o));
            const int n0 = s * 128;
            const int xbuf = OFF_X + xslot * TILE;
            u16x4 (&pg)[8] = pgv[s % NSETS];
            // ---- data gradient: C[64][slice] = EY x W1 (K = 128); this slice's fragments were requested a slice ago
            if (s == 0) vmwait<28>(); else vmwait<20>();                     // behind them: S2 S3 S4 of the last slice (+ T1 T2 at a tile's first slice)
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(bw[i]);
            {
                const int ln = t_o & 63, wv = t_o >> 6;
                const int r = ln & 31, h = ln >> 5;
                const bool wave_live = n0 + wv * 32 < N;
                f32x16 acc[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
                if (wave_live) {
                    const int a_base = ebuf + r * 256, w4 = (h ^ swz16(r)) << 4;
                    auto afrag = [&](int ks, int i) { return *reinterpret_cast<const bf16x8_t*>(smem + a_base + i * 8192 + (w4 ^ (ks << 5))); };
                    bf16x8_t a0 = afrag(0, 0), a1 = afrag(0, 1);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        bf16x8_t b0 = a0, b1 = a1;
                        if (ks + 1 < 8) { b0 = afrag(ks + 1, 0); b1 = afrag(ks + 1, 1); }
                        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, bw[ks]);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf, acc[1], 0, 0, 0);
                        a0 = b0; a1 = b1;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // S1: the next slice's weight fragments (past the end: slice 0 again)
                load_w(t_o, s + 1 < NS ? s + 1 : 0, bw);
                float* cw = Cs + 4 * h * CLD + wv * 32 + r;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) cw[(i * 32 + (e & 3) + 8 * (e >> 2)) * CLD] = acc[i][e];
            }
            // ---- (C1) this wave's x rows and its G rows of the slice have landed
            if (s <= 1) vmwait<48>(); else vmwait<40>();                     // behind them: S4, a whole slice, S1 (+ T1 T2 when a tile began in between)
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(pg[i]);
            lds_barrier();
            // ---- epilogue: PReLU1 / norm1 backward against x (LDS), G += sc*dU, activated input left in place of x
            {
                const int c4 = t_o & 31, r0 = t_o >> 5;
                const bool col_ok = n0 + c4 * 4 < N;
                const float4 csc = *reinterpret_cast<const float4*>(tab + 256 + n0 + c4 * 4);
                const float4 csh = *reinterpret_cast<const float4*>(tab + 256 + 512 + n0 + c4 * 4);
                const float4 csl = *reinterpret_cast<const float4*>(tab + 256 + 1024 + n0 + c4 * 4);
                const float kc[4] = {csc.x, csc.y, csc.z, csc.w}, kh[4] = {csh.x, csh.y, csh.z, csh.w}, kl[4] = {csl.x, csl.y, csl.z, csl.w};
                const float* crow = Cs + r0 * CLD + c4 * 4;
                const char* dump = dump0 + (t_o << 3);
                const char* grow = Gp + ((m0 + r0) * g.ldg + n0 + c4 * 4) * 2;
                const long gstride = g.ldg * 16;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = r0 + 8 * i;
                    const bool live = col_ok && r < rows;
                    const int xoff = xbuf + r * 256 + (((c4 >> 1) ^ swz16(r)) << 4) + ((c4 & 1) << 3);
                    u16x4 xv = *reinterpret_cast<const u16x4*>(smem + xoff);
                    if (!col_ok) xv = u16x4{0, 0, 0, 0};                      // a neighbour layer's channels in the slice's last 16-B chunk
                    const float4 ca = *reinterpret_cast<const float4*>(crow + i * 8 * CLD);
                    const float cv[4] = {ca.x, ca.y, ca.z, ca.w};
                    const u16x4 gv = pg[i];
                    u16x4 o, xa;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = bf2f(xv[j]);
                        const float uu = fmaf(x, kc[j], kh[j]);
                        const float dA = live ? cv[j] : 0.f;
                        const float du = uu > 0.f ? dA : kl[j] * dA;
                        st1[s][j] += du; st2[s][j] = fmaf(du, x, st2[s][j]); st3[s][j] = fmaf(uu > 0.f ? 0.f : dA, uu, st3[s][j]);
                        o[j] = f2bf(fmaf(kc[j], du, bf2f(gv[j])));
                        xa[j] = f2bf(prelu(uu, kl[j]));
                    }
                    // S2: one store per row, ALWAYS issued (masked lanes store to the dump row); cin % 4 == 0: no partial chunks here
                    const char* real = grow + i * gstride;
                    pinp(real);
                    gstore8(const_cast<char*>(live ? real : dump), o);
                    *reinterpret_cast<u16x4*>(smem + xoff) = xa;
                    if (i == 3) __builtin_amdgcn_sched_barrier(0);          // two groups of four rows in flight (register pressure)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { pin(st1[s][j]); pin(st2[s][j]); pin(st3[s][j]); }
                // S3: the G rows of the slice two ahead
                {
                    const int s2 = (s + 2) % NS, dl = (s + 2) / NS;
                    const long m2 = lt + dl < ntl ? tile_of(lt + dl) * ROWS : g.M;
                    load_g(t_o, m2, rows_of(m2), s2 * 128, pgv[(NS & 1) ? s2 : s % NSETS]);
                }
            }
            lds_barrier();                                                   // (C2)
            // ---- weight gradient: accW[s] += EY^T x prelu(bn1(x)) over the tile's 64 pixels
            if (n0 + ((t_o >> 6) & 1) * 64 < N) {
                const int ln = t_o & 63, wv = t_o >> 6;
                const int wi = wv >> 1, wj = wv & 1;
                const int gq = ln >> 4, tq = (ln >> 2) & 3, tp = ln & 3;
                const int khalf = gq >> 1, chalf = gq & 1;
                const int sub = (tp & 1) * 8;
                const int rl = 8 * khalf + tq;
                int a_lo[2], a_hi[2], b_lo[2], b_hi[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ac = wi * 8 + t * 4 + 2 * chalf + (tp >> 1), bc = wj * 8 + t * 4 + 2 * chalf + (tp >> 1);
                    a_lo[t] = ebuf + rl * 256 + ((ac ^ swz16(rl)) << 4) + sub; a_hi[t] = ebuf + (rl + 4) * 256 + ((ac ^ swz16(rl + 4)) << 4) + sub;
                    b_lo[t] = xbuf + rl * 256 + ((bc ^ swz16(rl)) << 4) + sub; b_hi[t] = xbuf + (rl + 4) * 256 + ((bc ^ swz16(rl + 4)) << 4) + sub;
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
                        for (int tb = 0; tb < 2; ++tb) accW[s][ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], accW[s][ta][tb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            lds_barrier();                                                   // (C3) this slice's ring slot is free
            // S4: the x slice three ahead goes into it
            {
                const int s3 = (s + 3) % NS, dl = (s + 3) / NS;
                const long m3 = lt + dl < ntl ? tile_of(lt + dl) * ROWS : g.M;
                dma_x(t_o, m3, rows_of(m3), s3 * 128, xbuf);
            }
            xslot = xslot == 2 ? 0 : xslot + 1;
        }
    }
    vmwait<0>();
    __syncthreads();

    // ---- per-workgroup results: statistics partials, bias column sums, the weight-gradient tiles
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int n0 = s * 128, c4 = lane & 31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double d1 = (double)st1[s][j], d2 = (double)st2[s][j], d3 = (double)st3[s][j];
            d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32); d3 += __shfl_xor(d3, 32);
            if (lane < 32) {
                double* p = red + ((wave * 128) + c4 * 4 + j) * 3;
                p[0] = d1; p[1] = d2; p[2] = d3;
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            double a = 0, b = 0, c = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[(w * 128 + tid) * 3]; b += red[(w * 128 + tid) * 3 + 1]; c += red[(w * 128 + tid) * 3 + 2]; }
            double* p = g.part + ((long)blockIdx.x * N + n0 + tid) * 3;
            p[0] = a; p[1] = b; p[2] = c;
        }
        __syncthreads();
    }
    {
        const int c8 = tid & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float c1 = cs[j];
            c1 += __shfl_xor(c1, 16); c1 += __shfl_xor(c1, 32);
            if (lane < 16) csum[wave * 128 + c8 * 8 + j] = c1;
        }
    }
    __syncthreads();
    if (tid < 128) g.tail[(long)blockIdx.x * 128 + tid] = (csum[tid] + csum[128 + tid]) + (csum[256 + tid] + csum[384 + tid]);
    const int wi = wave >> 1, wj = wave & 1;
    const int cj = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int j = s * 128 + wj * 64 + tb * 32 + cj;
                if (j >= g.ldc) continue;                                    // columns in [cin, ldc) are written as zeros (padding of the kernel layout)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = wi * 64 + ta * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    g.slab[((long)blockIdx.x * 128 + i) * g.ldc + j] = accW[s][ta][tb][e];
                }
            }
}

}  // namespace

bool bwd1x1_wide_ok(const Bwd1x1Args& a) {
    // OPT-IN (validation build, TCVN_BWD1_WIDE=1): measured 20-30 % SLOWER than the per-slice kernel on every wide launch of the config-2
    // step (header) -- the product library never selects it; tests/test_densenet_gpu.py keeps it correct
    static const bool on = TCVN_KNOB_SET("TCVN_BWD1_WIDE");
    return on && a.cin > 128 && a.cin <= 512 && (a.cin & 3) == 0 && a.Kp == 128 && a.ldc >= a.cin && a.M * 256L < (1L << 40);
}
int bwd1x1_wide_nblk(const Bwd1x1Args& a) {
    const long mt = (a.M + ROWS - 1) / ROWS;
    if (mt <= 256) return (int)(mt < 1 ? 1 : mt);                      // one workgroup per CU ...
    const long per = (mt + 255) / 256;                                 // ... each with the same number of tiles (+- 1): the fewest workgroups
    return (int)((mt + per - 1) / per);                                // that reach the makespan of 256 (every slab costs NS x 64 KB of HBM twice)
}
int bwd1x1_wide_launch(const Bwd1x1Args& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd1x1_wide_bf16<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd1x1_wide_bf16<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd1x1_wide_bf16<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const int ns = cdiv(a.cin, 128);
    // SURVEY 8(d) strict bytes: operands DU, Y, x read once, the G contribution written once (its read is traffic, not algorithm)
    ProfScope ps("k_bwd1x1_fused_bf16", 4.0 * a.M * 128.0 * a.cin, (double)a.M * (512.0 + 4.0 * a.cin), st);
    if (ns == 2) hipLaunchKernelGGL(k_bwd1x1_wide_bf16<2>, dim3(a.nblk), dim3(256), SMEM_BYTES, st, a);
    else if (ns == 3) hipLaunchKernelGGL(k_bwd1x1_wide_bf16<3>, dim3(a.nblk), dim3(256), SMEM_BYTES, st, a);
    else hipLaunchKernelGGL(k_bwd1x1_wide_bf16<4>, dim3(a.nblk), dim3(256), SMEM_BYTES, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
