// bf16 3x3 / stride 1 / pad 1 convolutions with 64 input and 64 output channels on full-resolution NHWC maps: the layers of the
// first two stages of the SDXL-style embedder (diffusers ResnetBlock2D conv1 / conv2 at 400x280 and 200x140; reference call
// site transformercvn/network/layers/sdxl_net.py:27-34, block definitions restated in oracle/sdxl_oracle.py), forward and data
// gradient.  These maps are 40 % of the embedder's FLOPs and, in the generic implicit-GEMM kernels, re-read every input pixel
// nine times from L2 with one dependent load per 32-wide K chunk.
//
// Here a 512-thread workgroup owns an 8 x 32 pixel tile of one map: its 10 x 34 pixel halo patch (64 channels, 128 B per pixel,
// 144-B pitch so that the 16-B fragment reads of 16 lanes fall on 16 different bank groups) is staged ONCE in LDS, double
// buffered -- the loads of the next tile are in flight under the MFMAs of this one.  A wave owns two tile rows (2 x 32
// positions) and 32 of the 64 output channels; its 36 weight fragments (9 taps x 4 chunks of 16 input channels,
// v_mfma_f32_32x32x16_bf16) stay in registers for the whole launch, the nine taps are plain offsets into the patch.  The tile
// leaves through an fp32 C tile in LDS (two passes of 128 positions), so bias / residual are added in fp32 before the one rounding
// to bf16 -- the same arithmetic as the generic kernel -- and every lane stores 16 contiguous bytes of an NHWC row.
//
// The data gradient of such a layer is the same convolution over the output gradient with the taps flipped and the channel
// roles exchanged: the kernel reads the transposed weight pack ([Cin][tap*Cout + n]) at tap 8 - t.
// Algorithmic HBM bytes per map pixel: 128 B read + 128 B written (+ 128 B residual); MFMA work 2 * 576 * 64 flop.
#include "prof.h"
#include "sdxl_ops.h"
#include "tcvn_ops.h"

namespace tcvn {

namespace {

constexpr int TH = 8, TW = 32;                        // output tile
constexpr int PW = TW + 2, PH = TH + 2;               // halo patch
constexpr int PS = 144;                               // bytes per patch pixel (128 + 16)
constexpr int PATCH_BYTES = ((PH * PW + 6) / 7) * 7 * PS;    // 49 392: whole LDS-DMA groups of 7 pixels
constexpr int CP = 68;                                // fp32 C tile pitch (floats)
constexpr int CT_BYTES = 128 * CP * 4;                // 34 816

// Tile -> pixel mapping of the 8 x 32 stride-1 tile.  Large maps: tiles_x x tiles_y tiles per image.  Small maps would waste most
// of a tile (a 7 x 5 map fills 14 % of it) and re-read the weights once per image, so maps of width <= 15 are PACKED side by side
// (px images per tile row, one zero column between neighbours -- the padding both of them need) and maps of height <= 3 are stacked
// (py per tile).  A tile then carries px * py images.
struct TileMap {
    int n, H, W, tiles_x, tiles_y, px, py;
    __host__ __device__ int ntiles() const { return ((n + px * py - 1) / (px * py)) * tiles_x * tiles_y; }
    // pixel of tile t at tile-relative (ry, rx), ry in [-1, 8], rx in [-1, 32]; false = zero padding / outside / no such image
    __device__ __forceinline__ bool pixel(int t, int ry, int rx, int& img, int& y, int& x) const {
        const int tx = t % tiles_x, r = t / tiles_x, ty = r % tiles_y, grp = r / tiles_y;
        int sx = 0, sy = 0;
        if (px > 1) { if (rx < 0) return false; sx = rx / (W + 1); x = rx - sx * (W + 1); if (sx >= px) return false; }
        else x = tx * TW + rx;
        if (py > 1) { if (ry < 0) return false; sy = ry / (H + 1); y = ry - sy * (H + 1); if (sy >= py) return false; }
        else y = ty * TH + ry;
        img = grp * (px * py) + sy * px + sx;
        return x >= 0 && x < W && y >= 0 && y < H && img < n;
    }
};

// barrier that orders LDS traffic only: __syncthreads() would also wait for every outstanding global store (vmcnt(0))
__device__ __forceinline__ void lds_barrier_() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// GroupNorm statistics of a tile's STORED values (sum, sum of squares over pixels x channels of one image), fused into the producing
// convolution's epilogue: wave partials through two LDS float atomics, one pair of fp64 atomics per tile by thread 0 afterwards.
__device__ __forceinline__ void stats_wave_add(float s, float ss, float* lacc) {
    s = wave_sum(s); ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&lacc[0], s); atomicAdd(&lacc[1], ss); }
}
__device__ __forceinline__ void stats_flush(float* lacc, double* stats, int img) {      // thread 0, after a barrier behind the adds
    atomicAdd(stats + 2 * img, (double)lacc[0]); atomicAdd(stats + 2 * img + 1, (double)lacc[1]);
    lacc[0] = 0.f; lacc[1] = 0.f;
}

__device__ __attribute__((aligned(16))) unsigned int g_zero_line[4];       // 16 B of zeros: LDS-DMA source of padding

#ifdef TCVN_PHASE_PROF
__device__ unsigned long long g_ph2[16];
#define PH2_INIT long long ph_last = clock64();
#define PH2(i) if (threadIdx.x == 0) { const long long ph_now = clock64(); atomicAdd(&g_ph2[i], (unsigned long long)(ph_now - ph_last)); ph_last = ph_now; }
#else
#define PH2_INIT
#define PH2(i)
#endif

struct C64Args {
    const bf16* In; const bf16* W; const float* bias; const bf16* Res; bf16* Out; double* stats;
    int n, H, W_, flip;
    int tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(512, 1) void k_sconv3_c64(const C64Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                                            // [2][PATCH_BYTES]
    float* Cs = reinterpret_cast<float*>(smem + 2 * PATCH_BYTES);  // [128][CP]
    float* lacc = reinterpret_cast<float*>(smem + 2 * PATCH_BYTES + CT_BYTES);     // [2] statistics of the tile
    if (threadIdx.x == 0) { lacc[0] = 0.f; lacc[1] = 0.f; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wp = wave >> 1, wn = wave & 1;                       // tile rows 2wp, 2wp+1; output channels [32wn, +32)

    // weight fragments: B[k][j = n]: lane (n = l31, k half = lh) holds W[32wn + n][tap'*64 + kc*16 + lh*8 .. +8]
    bf16x8_t bw[36];
    {
        const bf16* wrow = g.W + (long)(32 * wn + l31) * 576 + lh * 8;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                bw[tap * 4 + kc] = *reinterpret_cast<const bf16x8_t*>(wrow + (g.flip ? 8 - tap : tap) * 64 + kc * 16);
    }
    const float bias = g.bias ? g.bias[32 * wn + l31] : 0.f;

    // patch staging by LDS-DMA (global_load_lds, 16 B per lane, no registers): one wave instruction writes 1 KiB of consecutive LDS,
    // so it carries 7 pixels of 144 B -- lanes 9q .. 9q+7 the eight chunks of pixel q, lane 9q+8 the pad slot (fed from zeros),
    // lane 63 switched off.  Pixels outside the map read the zero line as well.
    const int dq = lane / 9, dslot = lane - dq * 9;
    auto issue = [&](int t, int buf) {
        const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        for (int j = wave; j * 7 < PH * PW; j += 8) {
            const int pix = j * 7 + dq;
            const int py = pix / PW, px = pix - py * PW;
            const int y = y0 + py, x = x0 + px;
            const bool ok = dslot < 8 && pix < PH * PW && y >= 0 && y < g.H && x >= 0 && x < g.W_;
            const bf16* src = ok ? g.In + (((long)img * g.H + y) * g.W_ + x) * 64 + dslot * 8 : reinterpret_cast<const bf16*>(g_zero_line);
            if (lane < 63)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(patch + buf * PATCH_BYTES + j * 7 * PS), 16, 0, 0);
        }
    };

    // workgroups of one XCD (blockIdx % 8) walk neighbouring tiles, so the halo pixels two tiles share are served by one L2
    const int nb = gridDim.x;
    const int lb = (nb % 8 == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    int t = lb, buf = 0;
    if (t < g.ntiles) issue(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PH2_INIT
    for (; t < g.ntiles; t += nb, buf ^= 1) {
        const int tn = t + nb;
        PH2(4)
        if (tn < g.ntiles) issue(tn, buf ^ 1);
        PH2(0)
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const char* pb = patch + buf * PATCH_BYTES + ((2 * wp) * PW + l31) * PS + lh * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char* pt = pb + ((tap / 3) * PW + (tap % 3)) * PS;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(pt + kc * 32);
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(pt + PW * PS + kc * 32);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bw[tap * 4 + kc], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[tap * 4 + kc], acc[1], 0, 0, 0);
            }
        }
        PH2(1)
        // the next patch was requested before the MFMAs: collect it here, in front of the epilogue, so that the epilogue's global
        // stores (which retire through the same in-order vmcnt) drain under the next tile instead of being waited for
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PH2(2)
        // epilogue: two passes of 128 positions (tile rows 0-3, 4-7) through the fp32 C tile
        const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
        float ts = 0.f, tss = 0.f;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if ((wp >> 1) == pass) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Cs[(((wp & 1) * 2 + rt) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CP + 32 * wn + l31] = acc[rt][e] + bias;
            }
            lds_barrier_();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 512, pos = idx >> 3, ch = idx & 7;     // 128 positions x 8 chunks
                const int y = ty * TH + pass * 4 + (pos >> 5), x = tx * TW + (pos & 31);
                if (y < g.H && x < g.W_) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8 + 4);
                    float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                    const long o = (((long)img * g.H + y) * g.W_ + x) * 64 + ch * 8;
                    if (g.Res) {
                        const u16x8 rv = *reinterpret_cast<const u16x8*>(g.Res + o);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += bf2f(rv[j]);
                    }
                    u16x8 ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = f2bf(v[j]);
                    if (g.stats) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float f = bf2f(ov[j]); ts += f; tss += f * f; }
                    }
                    *reinterpret_cast<u16x8*>(g.Out + o) = ov;
                }
            }
            if (pass == 0) lds_barrier_();
        }
        if (g.stats) stats_wave_add(ts, tss, lacc);
        PH2(3)
        lds_barrier_();                                          // C tile free again, next patch complete
        if (g.stats && threadIdx.x == 0) stats_flush(lacc, g.stats, img);
    }
}

TileMap make_map(int n, int H, int W) {
    TileMap m{};
    m.n = n; m.H = H; m.W = W;
    m.px = W <= 15 ? TW / (W + 1) : 1;
    m.py = (H <= 3 && m.px > 1) ? TH / (H + 1) : 1;                // stacking only together with side-by-side packing
    m.tiles_x = m.px > 1 ? 1 : (W + TW - 1) / TW;
    m.tiles_y = m.py > 1 ? 1 : (H + TH - 1) / TH;
    return m;
}

constexpr size_t C64_SMEM = 2 * PATCH_BYTES + CT_BYTES + 16;       // 132 736

// ---------------------------------------------------------------------------------------------------------------------
// General width (input and output channels multiples of 64; the 128 / 256 / 512-channel stages): same tile, same patch, but the
// input channels pass through LDS in chunks of 64 and a workgroup produces one group of 64 output channels (blockIdx.y).
// The weight fragments no longer fit in registers, so they stream from L2 one tap ahead of the MFMAs in a ring of five slots
// (tap t lives in slot t % 5; the loads of tap t+2 are issued under tap t).  vmcnt retires in order, so the LDS-DMA of the next
// patch is issued only after the last fragment loads of the chunk (and the first two taps of the next chunk) are in flight:
// nothing the MFMAs wait for is queued behind an HBM access.
// ---------------------------------------------------------------------------------------------------------------------
struct CGArgs {
    const bf16* In; const bf16* W; const float* bias; const bf16* Res; bf16* Out; double* stats;   // stats: unpacked maps only
    int n, H, W_, flip;
    int cin, cout, nc;                                  // channels of this pass' input / output, cin / 64
    TileMap map; int ntiles;
};

__global__ __launch_bounds__(512, 1) void k_sconv3_g(const CGArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    float* Cs = reinterpret_cast<float*>(smem + 2 * PATCH_BYTES);
    float* lacc = reinterpret_cast<float*>(smem + 2 * PATCH_BYTES + CT_BYTES);
    if (threadIdx.x == 0) { lacc[0] = 0.f; lacc[1] = 0.f; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wp = wave >> 1, wn = wave & 1;
    const int n0 = 64 * blockIdx.y;                                // this workgroup's output channels [n0, n0 + 64)
    const bf16* wrow = g.W + (long)(n0 + 32 * wn + l31) * (9 * g.cin) + lh * 8;
    const float bias = g.bias ? g.bias[n0 + 32 * wn + l31] : 0.f;

    bf16x8_t bq[5][4];
#define TCVN_BLOAD(slot, chunk, tap)                                                                              \
    {                                                                                                             \
        const bf16* p_ = wrow + (g.flip ? 8 - (tap) : (tap)) * g.cin + (chunk) * 64;                              \
        _Pragma("unroll") for (int kc = 0; kc < 4; ++kc) bq[slot][kc] = *reinterpret_cast<const bf16x8_t*>(p_ + kc * 16); \
    }
    const int dq = lane / 9, dslot = lane - dq * 9;
    auto issue = [&](int t, int chunk, int buf) {
        for (int j = wave; j * 7 < PH * PW; j += 8) {
            const int pix = j * 7 + dq;
            const int py = pix / PW, px = pix - py * PW;
            int img = 0, y = 0, x = 0;
            const bool ok = g.map.pixel(t, py - 1, px - 1, img, y, x) && dslot < 8 && pix < PH * PW;
            const bf16* src = ok ? g.In + (((long)img * g.H + y) * g.W_ + x) * g.cin + chunk * 64 + dslot * 8
                                 : reinterpret_cast<const bf16*>(g_zero_line);
            if (lane < 63)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(patch + buf * PATCH_BYTES + j * 7 * PS), 16, 0, 0);
        }
    };

    const int nb = gridDim.x;
    const int lb = (nb % 8 == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    int t = lb, chunk = 0, buf = 0;
    if (t >= g.ntiles) return;
    issue(t, 0, 0);
    TCVN_BLOAD(0, 0, 0)
    TCVN_BLOAD(1, 0, 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    while (true) {
        int nchunk = chunk + 1, nt = t;
        if (nchunk == g.nc) { nchunk = 0; nt = t + nb; }
        const bool more = nt < g.ntiles;
        const char* pb = patch + buf * PATCH_BYTES + ((2 * wp) * PW + l31) * PS + lh * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap <= 6) TCVN_BLOAD((tap + 2) % 5, chunk, tap + 2)
            if (tap == 6 && more) TCVN_BLOAD(0, nchunk, 0)
            if (tap == 7 && more) {
                TCVN_BLOAD(1, nchunk, 1)
                issue(nt, nchunk, buf ^ 1);
            }
            const char* pt = pb + ((tap / 3) * PW + (tap % 3)) * PS;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(pt + kc * 32);
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(pt + PW * PS + kc * 32);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[tap % 5][kc], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[tap % 5][kc], acc[1], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // next patch + next fragments; in front of the epilogue's stores
        if (chunk == g.nc - 1) {
            float ts = 0.f, tss = 0.f;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                if ((wp >> 1) == pass) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            Cs[(((wp & 1) * 2 + rt) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CP + 32 * wn + l31] = acc[rt][e] + bias;
                }
                lds_barrier_();
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = tid + i * 512, pos = idx >> 3, ch = idx & 7;
                    int img = 0, y = 0, x = 0;
                    if (g.map.pixel(t, pass * 4 + (pos >> 5), pos & 31, img, y, x)) {
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8);
                        const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8 + 4);
                        float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                        const long o = (((long)img * g.H + y) * g.W_ + x) * g.cout + n0 + ch * 8;
                        if (g.Res) {
                            const u16x8 rv = *reinterpret_cast<const u16x8*>(g.Res + o);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += bf2f(rv[j]);
                        }
                        u16x8 ov;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ov[j] = f2bf(v[j]);
                    if (g.stats) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float f = bf2f(ov[j]); ts += f; tss += f * f; }
                    }
                        *reinterpret_cast<u16x8*>(g.Out + o) = ov;
                    }
                }
                if (pass == 0) lds_barrier_();
            }
            if (g.stats) stats_wave_add(ts, tss, lacc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        }
        lds_barrier_();
        if (g.stats && chunk == g.nc - 1 && threadIdx.x == 0) stats_flush(lacc, g.stats, t / (g.map.tiles_x * g.map.tiles_y));
        if (!more) break;
        t = nt; chunk = nchunk; buf ^= 1;
    }
#undef TCVN_BLOAD
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 down-sampler forward (3x3, pad 0, zeros beyond the map; channels multiples of 64): the general-width scheme on a
// 2 x 32 output tile -- its 5 x 65 input patch has the size of the stride-1 patch -- with four waves (tile row x channel half),
// input pixel = 2 * output pixel + tap.  Input-bound: four input pixels are read per output pixel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int S2_PH = 5, S2_PW = 65;                  // patch of a 2 x 32 output tile
constexpr size_t S2_SMEM = 2 * PATCH_BYTES + 64 * CP * 4 + 16;

struct S2Args {
    const bf16* In; const bf16* W; const float* bias; bf16* Out; double* stats;
    int n, Hi, Wi, Ho, Wo, cin, cout, nc;
    int tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(256, 1) void k_sconv3_s2_fwd(const S2Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    float* Cs = reinterpret_cast<float*>(smem + 2 * PATCH_BYTES);
    float* lacc = reinterpret_cast<float*>(smem + 2 * PATCH_BYTES + 64 * CP * 4);
    if (threadIdx.x == 0) { lacc[0] = 0.f; lacc[1] = 0.f; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wn = wave & 1;                       // output tile row, channel half
    const int n0 = 64 * blockIdx.y;
    const bf16* wrow = g.W + (long)(n0 + 32 * wn + l31) * (9 * g.cin) + lh * 8;
    const float bias = g.bias ? g.bias[n0 + 32 * wn + l31] : 0.f;
    bf16x8_t bq[5][4];
#define TCVN_BLOAD(slot, chunk, tap)                                                                              \
    {                                                                                                             \
        const bf16* p_ = wrow + (tap) * g.cin + (chunk) * 64;                                                     \
        _Pragma("unroll") for (int kc = 0; kc < 4; ++kc) bq[slot][kc] = *reinterpret_cast<const bf16x8_t*>(p_ + kc * 16); \
    }
    const int dq = lane / 9, dslot = lane - dq * 9;
    auto issue = [&](int t, int chunk, int buf) {
        const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
        const int y0 = 2 * (ty * 2), x0 = 2 * (tx * TW);
        for (int j = wave; j * 7 < S2_PH * S2_PW; j += 4) {
            const int pix = j * 7 + dq;
            const int py = pix / S2_PW, px = pix - py * S2_PW;
            const int y = y0 + py, x = x0 + px;
            const bool ok = dslot < 8 && pix < S2_PH * S2_PW && y < g.Hi && x < g.Wi;
            const bf16* src = ok ? g.In + (((long)img * g.Hi + y) * g.Wi + x) * g.cin + chunk * 64 + dslot * 8
                                 : reinterpret_cast<const bf16*>(g_zero_line);
            if (lane < 63)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(patch + buf * PATCH_BYTES + j * 7 * PS), 16, 0, 0);
        }
    };
    const int nb = gridDim.x;
    const int lb = (nb % 8 == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    int t = lb, chunk = 0, buf = 0;
    if (t >= g.ntiles) return;
    issue(t, 0, 0);
    TCVN_BLOAD(0, 0, 0)
    TCVN_BLOAD(1, 0, 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    while (true) {
        int nchunk = chunk + 1, nt = t;
        if (nchunk == g.nc) { nchunk = 0; nt = t + nb; }
        const bool more = nt < g.ntiles;
        const char* pb = patch + buf * PATCH_BYTES + ((2 * wr) * S2_PW + 2 * l31) * PS + lh * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap <= 6) TCVN_BLOAD((tap + 2) % 5, chunk, tap + 2)
            if (tap == 6 && more) TCVN_BLOAD(0, nchunk, 0)
            if (tap == 7 && more) {
                TCVN_BLOAD(1, nchunk, 1)
                issue(nt, nchunk, buf ^ 1);
            }
            const char* pt = pb + ((tap / 3) * S2_PW + (tap % 3)) * PS;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(pt + kc * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[tap % 5][kc], acc, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (chunk == g.nc - 1) {
            const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
#pragma unroll
            for (int e = 0; e < 16; ++e) Cs[(wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CP + 32 * wn + l31] = acc[e] + bias;
            lds_barrier_();
            float ts = 0.f, tss = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256, pos = idx >> 3, ch = idx & 7;     // 64 positions x 8 chunks
                const int y = ty * 2 + (pos >> 5), x = tx * TW + (pos & 31);
                if (y < g.Ho && x < g.Wo) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8 + 4);
                    u16x8 ov;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ov[j] = f2bf(c0[j]); ov[4 + j] = f2bf(c1[j]); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = bf2f(ov[j]); ts += f; tss += f * f; }
                    *reinterpret_cast<u16x8*>(g.Out + (((long)img * g.Ho + y) * g.Wo + x) * g.cout + n0 + ch * 8) = ov;
                }
            }
            if (g.stats) stats_wave_add(ts, tss, lacc);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        }
        lds_barrier_();
        if (g.stats && chunk == g.nc - 1 && threadIdx.x == 0) stats_flush(lacc, g.stats, t / (g.tiles_x * g.tiles_y));
        if (!more) break;
        t = nt; chunk = nchunk; buf ^= 1;
    }
#undef TCVN_BLOAD
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 down-sampler data gradient, 64 output-gradient channels (the two full-resolution down-samplers):
//   dIn[y][x][c] = sum over taps with (y - ky), (x - kx) even of dOut[(y - ky)/2][(x - kx)/2][n] * W[n][c][ky][kx].
// The input pixels fall into four parity classes (y & 1, x & 1) with 4 / 2 / 2 / 1 contributing taps; each class is a small
// stride-1 convolution over the half-resolution output gradient.  A workgroup owns 4 x 32 half-resolution positions (8 x 64 input
// pixels): the 5 x 33 dOut patch arrives by LDS-DMA (double buffered), the 36 weight fragments of the transposed pack stay in
// registers, and the four classes are multiplied one after the other through the fp32 C tile (16-B stores of every other pixel).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int D2_PH = 5, D2_PW = 33;
constexpr int D2_PATCH = ((D2_PH * D2_PW + 6) / 7) * 7 * PS;          // 24 192
constexpr size_t D2_SMEM = 2 * D2_PATCH + CT_BYTES;

struct D2Args {
    const bf16* dOut; const bf16* Wt; bf16* dIn;
    int n, Hi, Wi, Ho, Wo, cin, accumulate;
    int tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(512, 1) void k_sconv3_s2_dgrad(const D2Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    float* Cs = reinterpret_cast<float*>(smem + 2 * D2_PATCH);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wy = wave >> 1, wn = wave & 1;
    const int n0 = 64 * blockIdx.y;                                // this workgroup's input channels [n0, n0 + 64)
    bf16x8_t bw[36];
    {
        const bf16* wrow = g.Wt + (long)(n0 + 32 * wn + l31) * 576 + lh * 8;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) bw[tap * 4 + kc] = *reinterpret_cast<const bf16x8_t*>(wrow + tap * 64 + kc * 16);
    }
    const int dq = lane / 9, dslot = lane - dq * 9;
    auto issue = [&](int t, int buf) {
        const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
        const int y0 = ty * 4 - 1, x0 = tx * TW - 1;
        for (int j = wave; j * 7 < D2_PH * D2_PW; j += 8) {
            const int pix = j * 7 + dq;
            const int py = pix / D2_PW, px = pix - py * D2_PW;
            const int y = y0 + py, x = x0 + px;
            const bool ok = dslot < 8 && pix < D2_PH * D2_PW && y >= 0 && y < g.Ho && x >= 0 && x < g.Wo;
            const bf16* src = ok ? g.dOut + (((long)img * g.Ho + y) * g.Wo + x) * 64 + dslot * 8 : reinterpret_cast<const bf16*>(g_zero_line);
            if (lane < 63)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(patch + buf * D2_PATCH + j * 7 * PS), 16, 0, 0);
        }
    };
    const int nb = gridDim.x;
    const int lb = (nb % 8 == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    int t = lb, buf = 0;
    if (t < g.ntiles) issue(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; t < g.ntiles; t += nb, buf ^= 1) {
        const int tn = t + nb;
        if (tn < g.ntiles) issue(tn, buf ^ 1);
        const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
        const char* pb = patch + buf * D2_PATCH + lh * 16;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if ((ky & 1) != py) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if ((kx & 1) != px) continue;
                    const char* pt = pb + ((wy + (ky == 2 ? 0 : 1)) * D2_PW + l31 + (kx == 2 ? 0 : 1)) * PS;
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) {
                        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(pt + kc * 32);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[(ky * 3 + kx) * 4 + kc], acc, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) Cs[(wy * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * CP + 32 * wn + l31] = acc[e];
            lds_barrier_();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 512, pos = idx >> 3, ch = idx & 7;     // 128 half-resolution positions x 8 chunks
                const int y = 2 * (ty * 4 + (pos >> 5)) + py, x = 2 * (tx * TW + (pos & 31)) + px;
                if (y < g.Hi && x < g.Wi) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(Cs + pos * CP + ch * 8 + 4);
                    float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                    bf16* o = g.dIn + (((long)img * g.Hi + y) * g.Wi + x) * g.cin + n0 + ch * 8;
                    if (g.accumulate) {
                        const u16x8 rv = *reinterpret_cast<const u16x8*>(o);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += bf2f(rv[j]);
                    }
                    u16x8 ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = f2bf(v[j]);
                    *reinterpret_cast<u16x8*>(o) = ov;
                }
            }
            lds_barrier_();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // next patch (the stores of this tile with it: once per tile)
        lds_barrier_();
    }
}

int launch_g(const SConv& g, const void* In, int cin, const void* W, int cout, const float* bias, const void* Res, void* Out, int flip,
             double* stats, hipStream_t st) {
    CGArgs a{};
    a.In = reinterpret_cast<const bf16*>(In); a.W = reinterpret_cast<const bf16*>(W); a.bias = bias;
    a.Res = reinterpret_cast<const bf16*>(Res); a.Out = reinterpret_cast<bf16*>(Out);
    a.n = g.n; a.H = g.Hin; a.W_ = g.Win; a.flip = flip; a.cin = cin; a.cout = cout; a.nc = cin / 64;
    a.map = make_map(g.n, g.Hin, g.Win); a.ntiles = a.map.ntiles();
    a.stats = (a.map.px == 1 && a.map.py == 1) ? stats : nullptr;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv3_g), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64_SMEM));
        attr = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL(k_sconv3_g, dim3(grid, cout / 64), dim3(512), C64_SMEM, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int launch_c64(const SConv& g, const void* In, const void* W, const float* bias, const void* Res, void* Out, int flip, double* stats,
               hipStream_t st) {
    C64Args a{};
    a.stats = stats;
    a.In = reinterpret_cast<const bf16*>(In); a.W = reinterpret_cast<const bf16*>(W); a.bias = bias;
    a.Res = reinterpret_cast<const bf16*>(Res); a.Out = reinterpret_cast<bf16*>(Out);
    a.n = g.n; a.H = g.Hin; a.W_ = g.Win; a.flip = flip;
    a.tiles_x = (g.Win + TW - 1) / TW; a.tiles_y = (g.Hin + TH - 1) / TH; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv3_c64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C64_SMEM));
        attr = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL(k_sconv3_c64, dim3(grid), dim3(512), C64_SMEM, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient: dW[n][tap*64 + c] = sum over positions p of In[p + shift(tap)][c] * dOut[p][n].
// The contraction runs over pixels, i.e. over the ROW index of both pixel-major LDS images, so both MFMA operands are read
// transposed with ds_read_b64_tr_b16 (lane roles as in conv3x3_tile.hip, checked by tools/micro/tr_read_test.hip): per
// 16-position k-step (half a tile row) a wave reads one dOut fragment and nine shifted In fragments.  A wave owns one
// (32 c x 32 n) quarter of all nine taps for the whole launch (144 accumulator registers); the workgroup's partial
// gradient leaves as one slab in the kernel layout [n][576], summed by k_slab_reduce.  Pixel rows are 128 B, 16-B chunks
// XOR-swizzled with the pixel index so that the four rows of a transpose group fall on different banks.
// Two workgroups per CU (76 KB of LDS each): one stages while the other multiplies.
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_;
__device__ __forceinline__ bf16x8_t tr_frag_(const char* smem_base, int off_lo, int off_hi) {
    typedef __attribute__((address_space(3))) s16x4_* lds_p;
    const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_lo));
    const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem_base + off_hi));
    struct { s16x4_ a, b; } pr = {lo, hi};
    return __builtin_bit_cast(bf16x8_t, pr);
}
// 128-B pixel rows: two rows per 256 B of banks.  A transposed read (half a wave) takes four consecutive pixels x one 64-B half row; with
// `chunk ^ (pix & 7)` pixels p and p + 2 of such a group landed on the same 16 banks (same row parity, same half) -- bit 1 of the pixel index
// now selects the half, so the four pixels of a group cover four different 64-B bank ranges whatever their alignment.
__device__ __forceinline__ int swz(int pix, int chunk, int sub) { return pix * 128 + ((chunk ^ ((((pix >> 1) & 1) << 2) | (pix & 3))) << 4) + sub; }

constexpr int WG_PATCH = PH * PW * 128;               // 43 520
constexpr int WG_DOUT = TH * TW * 128;                // 32 768
constexpr size_t WG_SMEM = WG_PATCH + WG_DOUT + PH * PW * 4;        // stride 1 (+ the packed-map table); the stride-2 instance needs less

struct W64Args {
    const bf16* In; const bf16* dOut; float* slab; float* bslab;
    int n, H, W_;                         // output map (= the dOut tensor)
    int tiles_x, tiles_y, ntiles;
    int cin, cout;                        // row strides; blockIdx.y selects the (64 c) x (64 n) sub-block: c chunk fastest
    int Hi, Wi;                           // input map
    TileMap map;                          // stride 1: tile -> pixel mapping (small maps packed)
};

// S = 1: 3x3 / stride 1 / pad 1 (8 x 32 output tile, 10 x 34 patch); S = 2: 3x3 / stride 2 / pad 0 with zeros beyond the map
// (diffusers' F.pad(0,1,0,1) down-sampler: 2 x 32 output tile, 5 x 65 patch, input pixel = 2 * output pixel + tap)
template <int S, bool PACK>
__global__ __launch_bounds__(256, 2) void k_sconv3_c64_wgrad(const W64Args g) {
    constexpr int OTH = S == 1 ? 8 : 2;                            // output tile rows
    constexpr int PHs = S * OTH + (S == 1 ? 2 : 1), PWs = S * TW + (S == 1 ? 2 : 1), ORG = S == 1 ? 1 : 0;
    constexpr int NPC = PHs * PWs * 8, NPL = (NPC + 255) / 256;    // patch chunks, per thread
    constexpr int NDL = OTH * TW * 8 / 256;                        // dOut chunks per thread
    constexpr int PATCH_B = PHs * PWs * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                                            // [PH*PW][128 B] In halo patch
    char* dout = smem + PATCH_B;
    // packed small maps: tile-independent part of the tile -> pixel mapping, one entry per patch pixel:
    // (image slot within the group) << 24 | pixel offset within the group's images, -1 = padding
    int* ptab = reinterpret_cast<int*>(smem + PATCH_B + OTH * TW * 128);
    if (PACK) {
        for (int p = threadIdx.x; p < PHs * PWs; p += 256) {
            const int ry = p / PWs - 1, rx = p % PWs - 1;
            int e = -1;                                            // packing in y implies packing in x (make_map); x-only packing
            if (rx >= 0 && (g.map.py == 1 || ry >= 0)) {           // keeps y for the use site (rows follow the y tile)
                const int sx = rx / (g.map.W + 1), x = rx - sx * (g.map.W + 1);
                int sy = 0, y = 0;
                if (g.map.py > 1) { sy = ry / (g.map.H + 1); y = ry - sy * (g.map.H + 1); }
                if (sx < g.map.px && x < g.map.W && (g.map.py == 1 || (sy < g.map.py && y < g.map.H))) {
                    const int slot = sy * g.map.px + sx;
                    e = (slot << 24) | (slot * g.map.H * g.map.W + y * g.map.W + x);
                }
            }
            ptab[p] = e;
        }
        __syncthreads();
    }                                  // [TH*TW][128 B] output-gradient tile
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wn = wave & 1;                       // input-channel half, output-channel half
    const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int khalf = gq >> 1, chalf = gq & 1;
    const int a_chunk = 4 * wc + 2 * chalf + (tp >> 1), b_chunk = 4 * wn + 2 * chalf + (tp >> 1), sub = (tp & 1) * 8;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

    const int ncc = g.cin >> 6;
    const int c0 = 64 * (blockIdx.y % ncc), n0 = 64 * (blockIdx.y / ncc);
    const int nb = gridDim.x;
    const int lb = (nb % 8 == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    for (int t = lb; t < g.ntiles; t += nb) {
        const int tx = t % g.tiles_x, r = t / g.tiles_x, ty = r % g.tiles_y, img = r / g.tiles_y;
        const int y0 = ty * OTH, x0 = tx * TW;
        const int gbase = img * (g.map.px * g.map.py);             // first image of this tile's group (packed maps)
        __syncthreads();                                           // previous tile's readers are done
        {
            // two batches: the tile mapping of the packed small maps needs registers too, and this kernel sits at the 256-register
            // limit of two workgroups per CU
            constexpr int NBAT = PACK ? 2 : 1, NPH = (NPL + NBAT - 1) / NBAT;
#pragma unroll
            for (int hb = 0; hb < NBAT; ++hb) {
                u16x8 v[NPH];
#pragma unroll
                for (int i = 0; i < NPH; ++i) {
                    const int idx = tid + (hb * NPH + i) * 256, pix = idx >> 3, ch = idx & 7;
                    const int py = pix / PWs, px = pix - py * PWs;
                    int im = img, y = S * y0 - ORG + py, x = S * x0 - ORG + px;
                    bool ok = idx < NPC;
                    v[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (PACK) {                                    // tiles of packed maps: t = image group (unless the map is tiled in y)
                        const int e = ok ? ptab[pix] : -1;
                        if (e >= 0 && gbase + (e >> 24) < g.n && g.map.py > 1)
                            v[i] = *reinterpret_cast<const u16x8*>(g.In + ((long)gbase * g.Hi * g.Wi + (e & 0xffffff)) * g.cin + c0 + ch * 8);
                        else if (e >= 0 && gbase + (e >> 24) < g.n) {           // packed in x only: rows follow the y tile
                            const int yy = ty * OTH + py - 1, slot = e >> 24, xx = (e & 0xffffff) - slot * g.Hi * g.Wi;      // y == ry here
                            if (yy >= 0 && yy < g.Hi)
                                v[i] = *reinterpret_cast<const u16x8*>(g.In + (((long)(gbase + slot) * g.Hi + yy) * g.Wi + xx) * g.cin + c0 + ch * 8);
                        }
                    } else {
                        ok = ok && y >= 0 && y < g.Hi && x >= 0 && x < g.Wi;
                        if (ok) v[i] = *reinterpret_cast<const u16x8*>(g.In + (((long)im * g.Hi + y) * g.Wi + x) * g.cin + c0 + ch * 8);
                    }
                }
#pragma unroll
                for (int i = 0; i < NPH; ++i) {
                    const int idx = tid + (hb * NPH + i) * 256, pix = idx >> 3, ch = idx & 7;
                    if (idx < NPC) *reinterpret_cast<u16x8*>(patch + swz(pix, ch, 0)) = v[i];
                }
            }
        }
        {
            u16x8 d[NDL];
#pragma unroll
            for (int i = 0; i < NDL; ++i) {
                const int idx = tid + i * 256, pos = idx >> 3, ch = idx & 7;
                int im = img, y = y0 + (pos >> 5), x = x0 + (pos & 31);
                bool ok = true;
                d[i] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (PACK) {
                    const int e = ptab[((pos >> 5) + 1) * PWs + (pos & 31) + 1];
                    if (e >= 0 && gbase + (e >> 24) < g.n && g.map.py > 1)
                        d[i] = *reinterpret_cast<const u16x8*>(g.dOut + ((long)gbase * g.H * g.W_ + (e & 0xffffff)) * g.cout + n0 + ch * 8);
                    else if (e >= 0 && gbase + (e >> 24) < g.n) {
                        const int yy = ty * OTH + (pos >> 5), slot = e >> 24, xx = (e & 0xffffff) - slot * g.H * g.W_;
                        if (yy < g.H) d[i] = *reinterpret_cast<const u16x8*>(g.dOut + (((long)(gbase + slot) * g.H + yy) * g.W_ + xx) * g.cout + n0 + ch * 8);
                    }
                } else {
                    ok = y < g.H && x < g.W_;
                    if (ok) d[i] = *reinterpret_cast<const u16x8*>(g.dOut + (((long)im * g.H + y) * g.W_ + x) * g.cout + n0 + ch * 8);
                }
            }
#pragma unroll
            for (int i = 0; i < NDL; ++i) {
                const int idx = tid + i * 256, pos = idx >> 3, ch = idx & 7;
                *reinterpret_cast<u16x8*>(dout + swz(pos, ch, 0)) = d[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[j] += bf2f(d[i][j]);
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < OTH * TW / 16; ++ks) {
            const int ry = ks >> 1, xb = (ks & 1) * 16 + 8 * khalf + tq;           // this lane's row of the transpose group
            const int bp = ry * TW + xb;
            const bf16x8_t b = tr_frag_(dout, swz(bp, b_chunk, sub), swz(bp + 4, b_chunk, sub));
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ap = (S * ry + tap / 3) * PWs + S * xb + tap % 3;
                const bf16x8_t a = tr_frag_(patch, swz(ap, a_chunk, sub), swz(ap + 4 * S, a_chunk, sub));
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tap], 0, 0, 0);
            }
        }
    }
    // slab[blockIdx][n][576]: accumulator row = input channel within the wave's half, column = output channel
    {
        const int l31 = lane & 31, lh = lane >> 5;
        float* out = g.slab + ((long)blockIdx.x * gridDim.y + blockIdx.y) * (64 * 576) + (long)(32 * wn + l31) * 576 + 32 * wc;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int e = 0; e < 16; ++e) out[tap * 64 + (e & 3) + 8 * (e >> 2) + 4 * lh] = acc[tap][e];
    }
    if (g.bslab == nullptr) return;                                // uniform
    const int ncc_ = g.cin / 64;                                   // general width (2-D grid): the sub-blocks of one output-channel group saw the same
    if ((int)blockIdx.y % ncc_ != 0) return;                       // dOut tiles; the first of them leaves the group's column sums (uniform per workgroup)
    // bias gradient: threads with equal tid & 7 hold the same 8 channels
    __syncthreads();
    float* br = reinterpret_cast<float*>(smem);                    // [32][64]
#pragma unroll
    for (int j = 0; j < 8; ++j) br[(tid >> 3) * 64 + (tid & 7) * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int q = 0; q < 32; ++q) s += br[q * 64 + tid];
        g.bslab[((long)blockIdx.x * (g.cout / 64) + blockIdx.y / ncc_) * 64 + tid] = s;
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool geom_ok(const SConv& g) {
    return conv3x3_tile_enabled() && g.mode == MODE_BF16 && g.ks == 3 && g.stride == 1 && g.pad == 1 && g.Cin == 64 && g.Cout == 64 && g.lda == 64 &&
           g.Ho == g.Hin && g.Wo == g.Win && g.Kp == 576 && g.Kpt == 576 && (long)g.n * g.Hin * g.Win < (1L << 31) / 64;
}

}  // namespace

namespace {
bool geom_g_ok(const SConv& g) {
    return conv3x3_tile_enabled() && g.mode == MODE_BF16 && g.ks == 3 && g.stride == 1 && g.pad == 1 && g.Cin % 64 == 0 && g.Cout % 64 == 0 &&
           g.Cin <= 512 && g.Cout <= 512 && g.lda == g.Cin && g.Ho == g.Hin && g.Wo == g.Win && g.Kp == 9 * g.Cin && g.Kpt == 9 * g.Cout &&
           !(g.Cin == 64 && g.Cout == 64);
}
}  // namespace
bool sconv3_g_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, long ldres, const void* Out, long ldo, int out_f32) {
    return geom_g_ok(g) && !out_f32 && ldo == g.Cout && (Res == nullptr || ldres == g.Cout) && al16(In) && al16(Wk) && al16(Res) && al16(Out);
}
bool sconv3_g_fuses_stats(const SConv& g) { const TileMap m = make_map(g.n, g.Hin, g.Win); return m.px == 1 && m.py == 1; }
int sconv3_g_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, const void* Res, void* Out, double* stats, hipStream_t st) {
    return launch_g(g, In, g.Cin, Wk, g.Cout, bias, Res, Out, 0, stats, st);
}
bool sconv3_g_dgrad_ok(const SConv& g, const void* dOut, long lddo, const void* Wt, const void* dIn, long lddi) {
    return geom_g_ok(g) && lddo == g.Cout && lddi == g.Cin && al16(dOut) && al16(Wt) && al16(dIn);
}
int sconv3_g_dgrad(const SConv& g, const void* dOut, const void* Wt, void* dIn, int accumulate, hipStream_t st) {
    return launch_g(g, dOut, g.Cout, Wt, g.Cin, nullptr, accumulate ? dIn : nullptr, dIn, 1, nullptr, st);
}

bool sconv3_c64_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, long ldres, const void* Out, long ldo, int out_f32) {
    return geom_ok(g) && !out_f32 && ldo == 64 && (Res == nullptr || ldres == 64) && al16(In) && al16(Wk) && al16(Res) && al16(Out);
}
int sconv3_c64_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, const void* Res, void* Out, double* stats, hipStream_t st) {
    return launch_c64(g, In, Wk, bias, Res, Out, 0, stats, st);
}
bool sconv3_c64_dgrad_ok(const SConv& g, const void* dOut, long lddo, const void* Wt, const void* dIn, long lddi) {
    return geom_ok(g) && lddo == 64 && lddi == 64 && al16(dOut) && al16(Wt) && al16(dIn);
}
int sconv3_c64_dgrad(const SConv& g, const void* dOut, const void* Wt, void* dIn, int accumulate, hipStream_t st) {
    return launch_c64(g, dOut, Wt, nullptr, accumulate ? dIn : nullptr, dIn, 1, nullptr, st);
}

namespace {
// dWk[(n0 + n)*Kp + tap*Cin + c0 + c] += sum_x slab[x][sub][n][tap*64 + c]
__global__ __launch_bounds__(256) void k_sub_reduce(const float* __restrict__ slab, int nx, int nsub, int ncc, int cin, float* __restrict__ dWk) {
    const int sub = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;                  // element of the 64 x 576 sub-block
    if (i >= 64 * 576) return;
    float s0 = 0.f, s1 = 0.f;
    int x = 0;
    for (; x + 7 < nx; x += 8) {                                   // eight slabs requested together (two per trip was a memory round trip per pair:
        float v[8];                                                // up to 128 dependent trips per launch, 73 launches per step)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = slab[((long)(x + q) * nsub + sub) * (64 * 576) + i];
        s0 += (v[0] + v[2]) + (v[4] + v[6]);
        s1 += (v[1] + v[3]) + (v[5] + v[7]);
    }
    for (; x + 1 < nx; x += 2) {
        s0 += slab[((long)x * nsub + sub) * (64 * 576) + i];
        s1 += slab[((long)(x + 1) * nsub + sub) * (64 * 576) + i];
    }
    if (x < nx) s0 += slab[((long)x * nsub + sub) * (64 * 576) + i];
    const int n = i / 576, k = i - n * 576, tap = k >> 6, c = k & 63;
    const int c0 = 64 * (sub % ncc), n0 = 64 * (sub / ncc);
    dWk[(long)(n0 + n) * (9 * cin) + tap * cin + c0 + c] += s0 + s1;
}
}  // namespace

bool sconv3_g_wgrad_ok(const SConv& g, const void* In, const void* dOut, long lddo) {
    return geom_g_ok(g) && lddo == g.Cout && al16(In) && al16(dOut) && g.slab != nullptr && g.slab_bytes >= kSconvSlabBytes;
}
int sconv3_g_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st) {
    W64Args a{};
    a.In = reinterpret_cast<const bf16*>(In); a.dOut = reinterpret_cast<const bf16*>(dOut);
    a.n = g.n; a.H = g.Hin; a.W_ = g.Win; a.cin = g.Cin; a.cout = g.Cout; a.Hi = g.Hin; a.Wi = g.Win;
    a.map = make_map(g.n, g.Hin, g.Win); a.tiles_x = a.map.tiles_x; a.tiles_y = a.map.tiles_y; a.ntiles = a.map.ntiles();
    const int ncc = g.Cin / 64, nsub = ncc * (g.Cout / 64);
    int gx = 512 / nsub;                                           // slab holds 512 sub-block partials; two workgroups per CU
    if (gx > a.ntiles) gx = a.ntiles;
    if (gx < 1) gx = 1;
    // bias gradient from the kernel's own dOut tiles (round 5; before: a separate column-sum pass over dOut, 73 launches and ~8 GB of reads per step):
    // one row of cout sums per x-workgroup, behind the 512 weight sub-block slabs (gx * cout <= 512 * 64 floats)
    a.slab = g.slab; a.bslab = dbias != nullptr ? g.slab + 512L * 64 * 576 : nullptr;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv3_c64_wgrad<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_SMEM));
        attr = true;
    }
    if (a.map.px > 1 || a.map.py > 1) {
        static bool attr2 = false;
        if (!attr2) {
            TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv3_c64_wgrad<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_SMEM));
            attr2 = true;
        }
        hipLaunchKernelGGL((k_sconv3_c64_wgrad<1, true>), dim3(gx, nsub), dim3(256), WG_SMEM, st, a);
    } else {
        hipLaunchKernelGGL((k_sconv3_c64_wgrad<1, false>), dim3(gx, nsub), dim3(256), WG_SMEM, st, a);
    }
    TCVN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sub_reduce, dim3((64 * 576 + 255) / 256, nsub), dim3(256), 0, st, a.slab, gx, nsub, ncc, g.Cin, dWk);
    TCVN_LAUNCH_CHECK();
    if (dbias != nullptr) return slab_reduce(a.bslab, gx, g.Cout, dbias, st);
    return 0;
}

namespace {
bool geom_s2_ok(const SConv& g) {
    return conv3x3_tile_enabled() && g.mode == MODE_BF16 && g.ks == 3 && g.stride == 2 && g.pad == 0 && g.Cin % 64 == 0 && g.Cout % 64 == 0 &&
           g.Cin <= 512 && g.Cout <= 512 && g.lda == g.Cin && g.Ho == (g.Hin - 2) / 2 + 1 && g.Wo == (g.Win - 2) / 2 + 1 &&
           g.Kp == 9 * g.Cin && g.Kpt == 9 * g.Cout;
}
}  // namespace
bool sconv3_s2_fwd_ok(const SConv& g, const void* In, const void* Wk, const void* Res, const void* Out, long ldo, int out_f32) {
    return geom_s2_ok(g) && Res == nullptr && !out_f32 && ldo == g.Cout && al16(In) && al16(Wk) && al16(Out);
}
int sconv3_s2_fwd(const SConv& g, const void* In, const void* Wk, const float* bias, void* Out, double* stats, hipStream_t st) {
    S2Args a{};
    a.stats = stats;
    a.In = reinterpret_cast<const bf16*>(In); a.W = reinterpret_cast<const bf16*>(Wk); a.bias = bias; a.Out = reinterpret_cast<bf16*>(Out);
    a.n = g.n; a.Hi = g.Hin; a.Wi = g.Win; a.Ho = g.Ho; a.Wo = g.Wo; a.cin = g.Cin; a.cout = g.Cout; a.nc = g.Cin / 64;
    a.tiles_x = (g.Wo + TW - 1) / TW; a.tiles_y = (g.Ho + 1) / 2; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv3_s2_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2_SMEM));
        attr = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL(k_sconv3_s2_fwd, dim3(grid, g.Cout / 64), dim3(256), S2_SMEM, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}
bool sconv3_s2_dgrad_ok(const SConv& g, const void* dOut, long lddo, const void* Wt, const void* dIn, long lddi) {
    return geom_s2_ok(g) && g.Cout == 64 && lddo == 64 && lddi == g.Cin && al16(dOut) && al16(Wt) && al16(dIn);
}
int sconv3_s2_dgrad(const SConv& g, const void* dOut, const void* Wt, void* dIn, int accumulate, hipStream_t st) {
    D2Args a{};
    a.dOut = reinterpret_cast<const bf16*>(dOut); a.Wt = reinterpret_cast<const bf16*>(Wt); a.dIn = reinterpret_cast<bf16*>(dIn);
    a.n = g.n; a.Hi = g.Hin; a.Wi = g.Win; a.Ho = g.Ho; a.Wo = g.Wo; a.cin = g.Cin; a.accumulate = accumulate;
    // half-resolution positions (Y, X) cover input pixels (2Y + py, 2X + px): Y up to ceil(Hi / 2) - 1
    const int hy = (g.Hin + 1) / 2, hx = (g.Win + 1) / 2;
    a.tiles_x = (hx + TW - 1) / TW; a.tiles_y = (hy + 3) / 4; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv3_s2_dgrad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)D2_SMEM));
        attr = true;
    }
    const int grid = a.ntiles < 256 ? a.ntiles : 256;
    hipLaunchKernelGGL(k_sconv3_s2_dgrad, dim3(grid, g.Cin / 64), dim3(512), D2_SMEM, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}
bool sconv3_s2_wgrad_ok(const SConv& g, const void* In, const void* dOut, long lddo) {
    return geom_s2_ok(g) && lddo == g.Cout && al16(In) && al16(dOut) && g.slab != nullptr && g.slab_bytes >= kSconvSlabBytes;
}
int sconv3_s2_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st) {
    W64Args a{};
    a.In = reinterpret_cast<const bf16*>(In); a.dOut = reinterpret_cast<const bf16*>(dOut);
    a.n = g.n; a.H = g.Ho; a.W_ = g.Wo; a.Hi = g.Hin; a.Wi = g.Win; a.cin = g.Cin; a.cout = g.Cout;
    a.tiles_x = (g.Wo + TW - 1) / TW; a.tiles_y = (g.Ho + 1) / 2; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    const int ncc = g.Cin / 64, nsub = ncc * (g.Cout / 64);
    int gx = 512 / nsub;
    if (gx > a.ntiles) gx = a.ntiles;
    if (gx < 1) gx = 1;
    a.slab = g.slab; a.bslab = dbias != nullptr ? g.slab + 512L * 64 * 576 : nullptr;
    hipLaunchKernelGGL((k_sconv3_c64_wgrad<2, false>), dim3(gx, nsub), dim3(256), (size_t)(5 * 65 * 128 + 2 * TW * 128), st, a);
    TCVN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sub_reduce, dim3((64 * 576 + 255) / 256, nsub), dim3(256), 0, st, a.slab, gx, nsub, ncc, g.Cin, dWk);
    TCVN_LAUNCH_CHECK();
    if (dbias != nullptr) return slab_reduce(a.bslab, gx, g.Cout, dbias, st);
    return 0;
}

bool sconv3_c64_wgrad_ok(const SConv& g, const void* In, const void* dOut, long lddo) {
    return geom_ok(g) && lddo == 64 && al16(In) && al16(dOut) && g.slab != nullptr &&
           g.slab_bytes >= kSconvSlabBytes;
}
int sconv3_c64_wgrad(const SConv& g, const void* In, const void* dOut, float* dWk, float* dbias, hipStream_t st) {
    W64Args a{};
    a.In = reinterpret_cast<const bf16*>(In); a.dOut = reinterpret_cast<const bf16*>(dOut);
    a.n = g.n; a.H = g.Hin; a.W_ = g.Win; a.cin = 64; a.cout = 64; a.Hi = g.Hin; a.Wi = g.Win;
    a.tiles_x = (g.Win + TW - 1) / TW; a.tiles_y = (g.Hin + TH - 1) / TH; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    const int grid = a.ntiles < 512 ? (a.ntiles < 256 ? a.ntiles : 256) : 512;
    a.slab = g.slab; a.bslab = g.slab + (long)grid * 64 * 576;
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sconv3_c64_wgrad<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_SMEM));
        attr = true;
    }
    hipLaunchKernelGGL((k_sconv3_c64_wgrad<1, false>), dim3(grid), dim3(256), WG_SMEM, st, a);
    TCVN_LAUNCH_CHECK();
    SlabJob none{};
    SlabJob jw = slab_job(a.slab, grid, 64L * 576, dWk, 0);
    SlabJob jb = dbias ? slab_job(a.bslab, grid, 64, dbias, 0) : none;
    return slab_reduce2(jw, jb, st);
}

}  // namespace tcvn

#ifdef TCVN_PHASE_PROF
extern "C" int tcvn_debug_phases2(unsigned long long* out16) {
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(tcvn::g_ph2), 16 * 8);
    return 0;
}
#endif
