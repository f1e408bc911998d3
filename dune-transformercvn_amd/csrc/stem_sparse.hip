// Sparse-aware stem (bf16 throughput mode): conv0 7x7/2 + BatchNorm0 + PReLU0 + AvgPool 3/2 and their backward straight from
// the COO hit list (reference: trainers/neutrino_full_dense_trainer.py:15-24 `sparse_to_dense` + layers/dense_net.py:112-121).
//
// The pixel maps are mostly empty: a 400 x 280 prong map holds 20-800 hits, an event map 500-4000.  The dense path scatters them into
// a [n,400,280,3] map, runs a dense MFMA convolution over all 28 000 output positions of every map, writes the [n,200,140,64] conv0
// output (1 GB for 288 maps), reads it again for the pooling and twice more in backward.  Here neither the dense map nor the conv0
// output exists in HBM: a workgroup owns a small region of conv0-output positions, finds the hits inside the region's input window
// through a per-step bucket index (32 x 32-pixel cells), lays them out in an LDS window map (duplicate coordinates: the hit with the
// highest list index wins -- a valid outcome of the reference's non-accumulating indexed write) and evaluates conv0 for the region by
// a gather over the 49 taps of every position: positions whose window holds no hit cost 49 LDS reads and equal the bias exactly.
// Because the region is recomputed wherever it is needed (cost proportional to the hits), four passes replace the dense kernels:
//   k_stem_sparse_stats   sum / sum of squares of the conv0 output per channel (BatchNorm0's batch statistics)        [train only]
//   k_stem_sparse_pool    conv0 -> BN0 -> PReLU0 -> AvgPool(3, stride 2) -> first 64 channels of dense block 1 (+ their statistics)
//   k_stem_sparse_bwd<0>  pooling / PReLU0 / BN0 backward sums (sum dU, sum dU*x, sum dz*min(u,0)) per channel
//   k_stem_sparse_bwd<1>  conv0 weight gradient: eff0 = sc*dU + P0*x + Q0 rebuilt per region in LDS, contracted with the hits
// conv0's output stays fp32 on chip (the dense bf16 path rounds it to bf16 when it stores it).  Deterministic: no floating-point
// atomics, fixed summation orders.  HBM traffic per step: the hit list, the pooled map (written once, read twice in backward).
#include <cstdlib>
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int SS_CELL = 5;                      // 32 x 32-pixel cells
constexpr int SS_N = 64;                        // conv0 output channels
constexpr int SS_K = 147;                       // 7*7*3
constexpr int SS_WLD = 33;                      // LDS weight row pitch in 32-bit words (64 bf16 + 1 pad word)
constexpr int SS_XLD = 68;                      // LDS region row pitch in floats (64 + 4: 16-B aligned rows, shifted banks)

// ---------------------------------------------------------------------------------------------------------------------
// bucket index: cell_start[cell] .. cell_start[cell+1] lists the hits whose pixel lies in 32x32-pixel cell `cell`
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float preprocess_value(float v, int mode, float noise_std, uint64_t seed, long flat_index) {
    v = mode == 1 ? logf(v + 1.f) : mode == 0 ? v / 255.0f : v;        // same arithmetic as k_scatter (elementwise.hip)
    if (noise_std != 0.f) {
        const float u1 = fmaxf(rng_uniform(seed, 0x6e6f6973u, (uint64_t)flat_index * 2), 1e-7f);
        const float u2 = rng_uniform(seed, 0x6e6f6973u, (uint64_t)flat_index * 2 + 1);
        v *= 1.f + noise_std * sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
    }
    return bf2f(f2bf(v));                                               // what the dense bf16 map would hold
}

__global__ void k_stem_index_count(const StemSparseArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nnz) return;
    const int img = a.coords[i * 3], y = a.coords[i * 3 + 1], x = a.coords[i * 3 + 2];
    if (img < 0 || img >= a.n_img || y < 0 || y >= a.H || x < 0 || x >= a.W) return;
    atomicAdd(&a.cell_fill[((long)img * a.cells_y + (y >> SS_CELL)) * a.cells_x + (x >> SS_CELL)], 1);
    float v[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < a.Cpix && c < 3; ++c) v[c] = preprocess_value(a.values[i * a.Cpix + c], a.value_mode, a.noise_std, a.seed, i * a.Cpix + c);
    a.pv[i] = make_float4(v[0], v[1], v[2], 0.f);
}

// exclusive prefix sum of the per-cell counts (one workgroup); resets the counters to zero for the fill pass
__global__ __launch_bounds__(1024) void k_stem_index_scan(const StemSparseArgs a, int ncells) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (ncells + 1023) / 1024;
    const int lo = t * per, hi = min(ncells, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a.cell_fill[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                      // Hillis-Steele inclusive scan
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                                    // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int c = a.cell_fill[i];
        a.cell_start[i] = run;
        a.cell_fill[i] = 0;
        run += c;
    }
    if (t == 1023) a.cell_start[ncells] = part[1023];
}

__global__ void k_stem_index_fill(const StemSparseArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nnz) return;
    const int img = a.coords[i * 3], y = a.coords[i * 3 + 1], x = a.coords[i * 3 + 2];
    if (img < 0 || img >= a.n_img || y < 0 || y >= a.H || x < 0 || x >= a.W) return;
    const long cell = ((long)img * a.cells_y + (y >> SS_CELL)) * a.cells_x + (x >> SS_CELL);
    a.cell_hits[a.cell_start[cell] + atomicAdd(&a.cell_fill[cell], 1)] = (int)i;      // order inside a cell is irrelevant (see window_map)
}

// ---------------------------------------------------------------------------------------------------------------------
// region machinery shared by the four passes
// ---------------------------------------------------------------------------------------------------------------------
// LDS window map of one region: input pixels [iy0, iy0 + WH) x [ix0, ix0 + WW) of image `img`.
//   idx[pixel] = highest hit index at that pixel or -1        (atomicMax: independent of the order the hits are visited in)
//   vm[pixel]  = (v0 | v1 << 16, v2 | flag << 16) bf16 values of that hit, flag = 1 when the pixel holds a hit
template <int WH, int WW>
__device__ __forceinline__ void window_map(const StemSparseArgs& a, int img, int iy0, int ix0, int* idx, uint2* vm, int tid) {
    for (int i = tid; i < WH * WW; i += 256) idx[i] = -1;
    __syncthreads();
    const int y_lo = max(iy0, 0), y_hi = min(iy0 + WH - 1, a.H - 1), x_lo = max(ix0, 0), x_hi = min(ix0 + WW - 1, a.W - 1);
    if (y_lo <= y_hi && x_lo <= x_hi) {
        const int cx_lo = x_lo >> SS_CELL, cx_hi = x_hi >> SS_CELL;
        for (int cy = y_lo >> SS_CELL; cy <= (y_hi >> SS_CELL); ++cy) {        // the cells of one cell row are contiguous in the index
            const long c0 = ((long)img * a.cells_y + cy) * a.cells_x;
            const int k0 = a.cell_start[c0 + cx_lo], k1 = a.cell_start[c0 + cx_hi + 1];
            for (int k = k0 + tid; k < k1; k += 256) {
                const int h = a.cell_hits[k];
                const int y = a.coords[(long)h * 3 + 1] - iy0, x = a.coords[(long)h * 3 + 2] - ix0;
                if (y >= 0 && y < WH && x >= 0 && x < WW) atomicMax(&idx[y * WW + x], h);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < WH * WW; i += 256) {
        const int h = idx[i];
        uint2 e = make_uint2(0u, 0u);
        if (h >= 0) {
            const float4 v = a.pv[h];
            e.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
            e.y = (unsigned)f2bf(v.z) | (1u << 16);
        }
        vm[i] = e;
    }
    __syncthreads();
}

// conv0 weights -> LDS as bf16 pairs: word (k, j2) = channels (2 j2, 2 j2 + 1) of contraction index k = (ky*7 + kx)*3 + c
__device__ __forceinline__ void load_weights(const StemSparseArgs& a, unsigned* wl, float* bias_l, int tid) {
    const bf16* Wk = reinterpret_cast<const bf16*>(a.Wk);
    for (int i = tid; i < SS_K * 32; i += 256) {
        const int k = i >> 5, j2 = i & 31;
        wl[k * SS_WLD + j2] = (unsigned)Wk[(long)(2 * j2) * a.Kp + k] | ((unsigned)Wk[(long)(2 * j2 + 1) * a.Kp + k] << 16);
    }
    if (tid < SS_N) bias_l[tid] = a.bias[tid];
}

// conv0 output of region position (ply, plx) (window origin = 2 * region origin - 3), channels [half*32, half*32 + 32)
template <int WW>
__device__ __forceinline__ void gather_c0(const uint2* vm, const unsigned* wl, const float* bias_l, int ply, int plx, int half, float (&acc)[32]) {
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = bias_l[half * 32 + j];
    const uint2* row = vm + (2 * ply) * WW + 2 * plx;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll 1
        for (int kx = 0; kx < 7; ++kx) {
            const uint2 e = row[ky * WW + kx];
            if (e.y >> 16) {
                const float v[3] = {bf2f((bf16)(e.x & 0xffffu)), bf2f((bf16)(e.x >> 16)), bf2f((bf16)(e.y & 0xffffu))};
                const unsigned* w = wl + ((ky * 7 + kx) * 3) * SS_WLD + half * 16;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int j2 = 0; j2 < 16; ++j2) {
                        const unsigned ww = w[c * SS_WLD + j2];
                        acc[2 * j2] = fmaf(v[c], __uint_as_float(ww << 16), acc[2 * j2]);
                        acc[2 * j2 + 1] = fmaf(v[c], __uint_as_float(ww & 0xffff0000u), acc[2 * j2 + 1]);
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// pass 1 (train): per-channel sum and sum of squares of the conv0 output over all Hc x Wc positions of every map
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ST_RH = 8, ST_RW = 16, ST_WH = 2 * ST_RH + 5, ST_WW = 2 * ST_RW + 5;       // region 8 x 16, window 21 x 37

__global__ __launch_bounds__(256, 2) void k_stem_sparse_stats(const StemSparseArgs a, int tiles_y, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float xt[ST_RH * ST_RW * SS_XLD];
    __shared__ unsigned wl[SS_K * SS_WLD];
    __shared__ float bias_l[SS_N];
    __shared__ int idx[ST_WH * ST_WW];
    __shared__ uint2 vm[ST_WH * ST_WW];
    __shared__ double red[4][SS_N][2];
    const int tid = threadIdx.x;
    load_weights(a, wl, bias_l, tid);
    const long ntiles = (long)a.n_img * tiles_y * tiles_x;
    const int ch = tid & 63, grp = tid >> 6;                    // reduction role: channel, quarter of the region's positions
    double s1 = 0, s2 = 0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y), img = (int)(tile / ((long)tiles_x * tiles_y));
        const int cy0 = ty * ST_RH, cx0 = tx * ST_RW;
        window_map<ST_WH, ST_WW>(a, img, 2 * cy0 - 3, 2 * cx0 - 3, idx, vm, tid);         // ends with a barrier: xt readers of the last tile are done
        {
            const int p = tid >> 1, half = tid & 1, ply = p / ST_RW, plx = p - ply * ST_RW;
            float acc[32];
            gather_c0<ST_WW>(vm, wl, bias_l, ply, plx, half, acc);
            float4* o = reinterpret_cast<float4*>(xt + p * SS_XLD + half * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        }
        __syncthreads();
        float f1 = 0.f, f2 = 0.f;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int p = grp * 32 + i, ply = p / ST_RW, plx = p - ply * ST_RW;
            if (cy0 + ply < a.Hc && cx0 + plx < a.Wc) { const float x = xt[p * SS_XLD + ch]; f1 += x; f2 = fmaf(x, x, f2); }
        }
        s1 += (double)f1; s2 += (double)f2;
    }
    red[grp][ch][0] = s1; red[grp][ch][1] = s2;
    __syncthreads();
    if (tid < SS_N) {
        double x = 0, y = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) { x += red[g][tid][0]; y += red[g][tid][1]; }
        a.part[((long)blockIdx.x * SS_N + tid) * 2] = x;
        a.part[((long)blockIdx.x * SS_N + tid) * 2 + 1] = y;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// pass 2: pooled, activated map.  A workgroup owns 4 x 8 pooled pixels = the 9 x 17 conv0 positions their 3x3/2 windows cover.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PL_PH = 4, PL_PW = 8, PL_RH = 2 * PL_PH + 1, PL_RW = 2 * PL_PW + 1, PL_WH = 2 * PL_RH + 5, PL_WW = 2 * PL_RW + 5;

__global__ __launch_bounds__(256, 2) void k_stem_sparse_pool(const StemSparseArgs a, int tiles_y, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float xt[PL_RH * PL_RW * SS_XLD];
    __shared__ unsigned wl[SS_K * SS_WLD];
    __shared__ float bias_l[SS_N];
    __shared__ __attribute__((aligned(16))) float tab[3][SS_N];  // BatchNorm0 scale, shift; PReLU0 slope
    __shared__ int idx[PL_WH * PL_WW];
    __shared__ uint2 vm[PL_WH * PL_WW];
    __shared__ double red[4][8][8][2];
    const int tid = threadIdx.x;
    load_weights(a, wl, bias_l, tid);
    if (tid < SS_N) { tab[0][tid] = a.sc[tid]; tab[1][tid] = a.sh[tid]; tab[2][tid] = a.sl[tid]; }
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(a.Out);
    const long ntiles = (long)a.n_img * tiles_y * tiles_x;
    const int q = tid >> 3, c8 = tid & 7, qy = q / PL_PW, qx = q - qy * PL_PW;       // output role: pooled pixel, 8-channel chunk
    double s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; }
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y), img = (int)(tile / ((long)tiles_x * tiles_y));
        const int cy0 = 2 * ty * PL_PH, cx0 = 2 * tx * PL_PW;                       // first conv0 position of the region
        window_map<PL_WH, PL_WW>(a, img, 2 * cy0 - 3, 2 * cx0 - 3, idx, vm, tid);
#pragma unroll 1
        for (int task = tid; task < PL_RH * PL_RW * 2; task += 256) {
            const int p = task >> 1, half = task & 1, ply = p / PL_RW, plx = p - ply * PL_RW;
            float acc[32];
            gather_c0<PL_WW>(vm, wl, bias_l, ply, plx, half, acc);
            float4* o = reinterpret_cast<float4*>(xt + p * SS_XLD + half * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = half * 32 + 4 * j;
                const float4 c = *reinterpret_cast<const float4*>(&tab[0][n]), h = *reinterpret_cast<const float4*>(&tab[1][n]);
                const float4 l = *reinterpret_cast<const float4*>(&tab[2][n]);
                o[j] = make_float4(prelu(fmaf(acc[4 * j], c.x, h.x), l.x), prelu(fmaf(acc[4 * j + 1], c.y, h.y), l.y),
                                   prelu(fmaf(acc[4 * j + 2], c.z, h.z), l.z), prelu(fmaf(acc[4 * j + 3], c.w, h.w), l.w));
                __builtin_amdgcn_sched_barrier(0);           // keep the table loads of the eight groups from piling up in registers
            }
        }
        __syncthreads();
        const int ho = ty * PL_PH + qy, wo = tx * PL_PW + qx;
        if (ho < a.Ho && wo < a.Wo) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float4* e = reinterpret_cast<const float4*>(xt + ((2 * qy + dy) * PL_RW + 2 * qx + dx) * SS_XLD + c8 * 8);
                    const float4 x0 = e[0], x1 = e[1];
                    s[0] += x0.x; s[1] += x0.y; s[2] += x0.z; s[3] += x0.w; s[4] += x1.x; s[5] += x1.y; s[6] += x1.z; s[7] += x1.w;
                }
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = f2bf(s[j] * (1.0f / 9.0f));
                const double x = (double)bf2f(o[j]);
                s1[j] += x; s2[j] += x * x;
            }
            *reinterpret_cast<u16x8*>(Out + (((long)img * a.Ho + ho) * a.Wo + wo) * a.ldo + c8 * 8) = o;
        }
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
    if (tid < SS_N) {
        const int ch = tid >> 3, j = tid & 7;
        double x = 0, y = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { x += red[w][ch][j][0]; y += red[w][ch][j][1]; }
        a.part[((long)blockIdx.x * SS_N + tid) * 2] = x;
        a.part[((long)blockIdx.x * SS_N + tid) * 2 + 1] = y;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  PASS 0: sums of the pooling + PReLU0 + BatchNorm0 backward over all conv0 positions.  PASS 1: with (P0, Q0) from
// those sums, the effective conv0 output gradient eff0 = sc*dU + P0*x + Q0 of the region in LDS, contracted with the region's hits:
//   dW0[n][ky][kx][c] += v[hit][c] * eff0[(y + 3 - ky)/2, (x + 3 - kx)/2][n]      for the taps that land on a position of the region
// (each (hit, tap) pair belongs to exactly one region).  One wave per hit, lane = output channel, 147 accumulators per lane.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BW_PH = ST_RH / 2 + 1, BW_PW = ST_RW / 2 + 1;

template <int PASS>
__global__ __launch_bounds__(256, 2) void k_stem_sparse_bwd(const StemSparseArgs a, int tiles_y, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xt = reinterpret_cast<float*>(smem);                                         // [128][SS_XLD]   x, then eff0 (PASS 1)
    float* effs = xt + ST_RH * ST_RW * SS_XLD;                                           // [45][64]        gradient of the pooled pixels
    unsigned* wl = reinterpret_cast<unsigned*>(effs + BW_PH * BW_PW * SS_N);             // [147][33]
    float* bias_l = reinterpret_cast<float*>(wl + SS_K * SS_WLD);                        // [64]
    int* idx = reinterpret_cast<int*>(bias_l + SS_N);                                    // [777]
    uint2* vm = reinterpret_cast<uint2*>(idx + ((ST_WH * ST_WW + 1) & ~1));              // [777]
    unsigned short* hits = reinterpret_cast<unsigned short*>(vm + ST_WH * ST_WW);        // [777 + pad] window pixels that hold a hit, ascending
    int* cnt = reinterpret_cast<int*>(hits + ((ST_WH * ST_WW + 3) & ~3));                // [4] per-wave counts of the compaction
    float* ctab = reinterpret_cast<float*>(cnt + 8);                                     // [7][64]: sc, sh, sl of BN0/PReLU0; P, Q of the pooled map; P0, Q0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    load_weights(a, wl, bias_l, tid);
    if (tid < SS_N) {
        ctab[tid] = a.sc[tid]; ctab[64 + tid] = a.sh[tid]; ctab[128 + tid] = a.sl[tid];
        ctab[192 + tid] = a.e.P[tid]; ctab[256 + tid] = a.e.Q[tid];
        ctab[320 + tid] = PASS ? a.P0[tid] : 0.f; ctab[384 + tid] = PASS ? a.Q0[tid] : 0.f;
    }
    const bf16* __restrict__ G = reinterpret_cast<const bf16*>(a.e.G);
    const bf16* __restrict__ D = reinterpret_cast<const bf16*>(a.e.X);
    const int c8 = tid & 7;
    auto tab8 = [&](int which, float (&v)[8]) {
        const float4 x0 = *reinterpret_cast<const float4*>(ctab + which * 64 + c8 * 8), x1 = *reinterpret_cast<const float4*>(ctab + which * 64 + c8 * 8 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    };
    double s1[8], s2[8], s3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; s3[j] = 0; }
    // weight-gradient accumulators: wave w owns kernel rows ky = 2w and 2w + 1 (every wave walks all hits of a region in the same
    // ascending order, so no cross-wave reduction exists and the sums are deterministic)
    float wacc[PASS ? 14 : 1][3];
#pragma unroll
    for (int t = 0; t < (PASS ? 14 : 1); ++t) { wacc[t][0] = 0.f; wacc[t][1] = 0.f; wacc[t][2] = 0.f; }

    const long ntiles = (long)a.n_img * tiles_y * tiles_x;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y), img = (int)(tile / ((long)tiles_x * tiles_y));
        const int cy0 = ty * ST_RH, cx0 = tx * ST_RW, ho0 = cy0 / 2 - 1, wo0 = cx0 / 2 - 1;
        window_map<ST_WH, ST_WW>(a, img, 2 * cy0 - 3, 2 * cx0 - 3, idx, vm, tid);         // (its first barrier also fences the last tile's readers)
        float cP[8], cQ[8];
        tab8(3, cP); tab8(4, cQ);
        for (int i = tid; i < BW_PH * BW_PW * 8; i += 256) {                              // gradient of the pooled pixels whose windows touch the region
            const int pp = i >> 3, py = pp / BW_PW, px = pp - py * BW_PW, ho = ho0 + py, wo = wo0 + px;
            float e8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ho >= 0 && ho < a.Ho && wo >= 0 && wo < a.Wo) {
                const long mo = ((long)img * a.Ho + ho) * a.Wo + wo;
                float gv[8], dv[8];
                load8<bf16>(G + mo * a.e.ldg + c8 * 8, gv);
                load8<bf16>(D + mo * a.e.ldx + c8 * 8, dv);
#pragma unroll
                for (int j = 0; j < 8; ++j) e8[j] = gv[j] + cP[j] * dv[j] + cQ[j];
            }
            float4* o = reinterpret_cast<float4*>(effs + pp * SS_N + c8 * 8);
            o[0] = make_float4(e8[0], e8[1], e8[2], e8[3]); o[1] = make_float4(e8[4], e8[5], e8[6], e8[7]);
        }
        {
            const int p = tid >> 1, half = tid & 1, ply = p / ST_RW, plx = p - ply * ST_RW;
            float acc[32];
            gather_c0<ST_WW>(vm, wl, bias_l, ply, plx, half, acc);
            float4* o = reinterpret_cast<float4*>(xt + p * SS_XLD + half * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
        }
        __syncthreads();
        float sc[8], sh[8], sl[8], p0[8], q0[8];
        tab8(0, sc); tab8(1, sh); tab8(2, sl); tab8(5, p0); tab8(6, q0);
#pragma unroll 1
        for (int k = 0; k < ST_RH * ST_RW / 32; ++k) {
            const int p = (tid >> 3) + 32 * k, ply = p / ST_RW, plx = p - ply * ST_RW, h = cy0 + ply, w = cx0 + plx;
            float4* xp = reinterpret_cast<float4*>(xt + p * SS_XLD + c8 * 8);
            if (h >= a.Hc || w >= a.Wc) {
                if (PASS) { xp[0] = make_float4(0.f, 0.f, 0.f, 0.f); xp[1] = make_float4(0.f, 0.f, 0.f, 0.f); }
                continue;
            }
            const int py1 = h / 2 - ho0, px1 = w / 2 - wo0;
            float dz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                if (dy == 1 && (h & 1)) continue;                 // windows containing row h: ho = h/2, and h/2 - 1 when h is even
                if (h / 2 - dy >= a.Ho) continue;                 // (out-of-map windows with ho < 0 hold zeros in effs)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    if (dx == 1 && (w & 1)) continue;
                    if (w / 2 - dx >= a.Wo) continue;
                    const float4* e4 = reinterpret_cast<const float4*>(effs + ((py1 - dy) * BW_PW + px1 - dx) * SS_N + c8 * 8);
                    const float4 x0 = e4[0], x1 = e4[1];
                    dz[0] += x0.x; dz[1] += x0.y; dz[2] += x0.z; dz[3] += x0.w; dz[4] += x1.x; dz[5] += x1.y; dz[6] += x1.z; dz[7] += x1.w;
                }
            }
            const float4 xa = xp[0], xb = xp[1];
            const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = dz[j] * (1.0f / 9.0f);
                const float x = xv[j];
                const float u = fmaf(x, sc[j], sh[j]);
                const float du = u > 0.f ? z : sl[j] * z;
                if (!PASS) { s1[j] += du; s2[j] += (double)du * x; s3[j] += u > 0.f ? 0.f : z * u; }
                o[j] = fmaf(sc[j], du, fmaf(p0[j], x, q0[j]));
            }
            if (PASS) { xp[0] = make_float4(o[0], o[1], o[2], o[3]); xp[1] = make_float4(o[4], o[5], o[6], o[7]); }
        }
        if (PASS) {
            // ordered compaction of the window pixels that hold a hit (ascending pixel index -> fixed summation order)
            constexpr int NPX = ST_WH * ST_WW, ROUNDS = (NPX + 255) / 256;
            int base = 0;
#pragma unroll 1
            for (int r = 0; r < ROUNDS; ++r) {
                const int i = r * 256 + tid;
                const bool has = i < NPX && (vm[i].y >> 16);
                const unsigned long long m = __ballot(has);
                if (lane == 0) cnt[wave] = __popcll(m);
                __syncthreads();
                int off = base;
                for (int w = 0; w < wave; ++w) off += cnt[w];
                if (has) hits[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
                base += cnt[0] + cnt[1] + cnt[2] + cnt[3];
                __syncthreads();
            }
            // (the second barrier of the last round also publishes eff0 in xt)
            const int n = lane;
            for (int hI = 0; hI < base; ++hI) {
                const int px = __builtin_amdgcn_readfirstlane((int)hits[hI]);
                const int wy = px / ST_WW, wx = px - wy * ST_WW;
                const uint2 e = vm[px];
                const float v[3] = {bf2f((bf16)(e.x & 0xffffu)), bf2f((bf16)(e.x >> 16)), bf2f((bf16)(e.y & 0xffffu))};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int ky = 2 * wave + kk;
                    const int ry = wy - ky;                                    // = 2 * ply
                    if (ky > 6 || (ry & 1) || ry < 0 || (ry >> 1) >= ST_RH) continue;
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        const int rx = wx - kx;
                        if ((rx & 1) || rx < 0 || (rx >> 1) >= ST_RW) continue;
                        const float ef = xt[((ry >> 1) * ST_RW + (rx >> 1)) * SS_XLD + n];          // 0 for positions outside the map
#pragma unroll
                        for (int c = 0; c < 3; ++c) wacc[kk * 7 + kx][c] = fmaf(v[c], ef, wacc[kk * 7 + kx][c]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!PASS) {
        double* red = reinterpret_cast<double*>(xt);                                     // [4][8][8][3]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); s3[j] += __shfl_xor(s3[j], o); }
            if (lane < 8) {
                double* r = red + (((wave * 8 + lane) * 8) + j) * 3;
                r[0] = s1[j]; r[1] = s2[j]; r[2] = s3[j];
            }
        }
        __syncthreads();
        if (tid < SS_N) {
            const int ch = tid >> 3, j = tid & 7;
            double x = 0, y = 0, z = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const double* r = red + (((w * 8 + ch) * 8) + j) * 3; x += r[0]; y += r[1]; z += r[2]; }
            double* o = a.part + ((long)blockIdx.x * SS_N + tid) * 3;
            o[0] = x; o[1] = y; o[2] = z;
        }
    } else {
        // every wave stores its own kernel rows: slab[block][n][Kp], k = (ky*7 + kx)*3 + c, through an LDS image [147][64]
        float* wsum = xt;                                                                // runs over xt into effs (both free now)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ky = 2 * wave + kk;
                    if (ky <= 6) wsum[((ky * 7 + kx) * 3 + c) * SS_N + lane] = wacc[PASS ? kk * 7 + kx : 0][c];
                }
        __syncthreads();
        float* out = a.slab + (long)blockIdx.x * SS_N * a.Kp;
        for (int i = tid; i < SS_N * a.Kp; i += 256) {
            const int nn = i / a.Kp, k = i - nn * a.Kp;
            out[i] = k < SS_K ? wsum[k * SS_N + nn] : 0.f;
        }
    }
}

size_t bwd_smem() {
    size_t b = (size_t)ST_RH * ST_RW * SS_XLD * 4 + (size_t)BW_PH * BW_PW * SS_N * 4 + (size_t)SS_K * SS_WLD * 4 + SS_N * 4;
    b += (size_t)((ST_WH * ST_WW + 1) & ~1) * 4 + (size_t)ST_WH * ST_WW * 8 + (size_t)((ST_WH * ST_WW + 3) & ~3) * 2 + 32 + 7 * 64 * 4;
    return b;
}

int region_grid(long ntiles) { return (int)(ntiles < 1024 ? ntiles : 1024); }          // persistent: <= 4 workgroups per CU worth of regions

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
long stem_sparse_hit_capacity(int n_img) { return (long)n_img * 8192; }                 // average occupancy up to 7 % of a 400 x 280 map
long stem_sparse_index_bytes(int n_img, int H, int W) {
    const long ncells = (long)n_img * cdiv(H, 32) * cdiv(W, 32);
    return round_up((ncells + 1) * 4, 256) + round_up(ncells * 4, 256) + round_up(stem_sparse_hit_capacity(n_img) * 4, 256) +
           round_up(stem_sparse_hit_capacity(n_img) * 16, 256);
}
void stem_sparse_carve(StemSparseArgs& a, char* base) {
    const long ncells = (long)a.n_img * a.cells_y * a.cells_x;
    a.cell_start = reinterpret_cast<int*>(base); base += round_up((ncells + 1) * 4, 256);
    a.cell_fill = reinterpret_cast<int*>(base); base += round_up(ncells * 4, 256);
    a.cell_hits = reinterpret_cast<int*>(base); base += round_up(stem_sparse_hit_capacity(a.n_img) * 4, 256);
    a.pv = reinterpret_cast<float4*>(base);
}
bool stem_sparse_ok(int mode, int in_ch, int init_ch, int H, int W, int value_mode, long nnz, int n_img, long ldo) {
    return conv3x3_tile_enabled() && mode == MODE_BF16 && in_ch >= 1 && in_ch <= 3 && init_ch == SS_N && value_mode >= 0 && value_mode <= 2 &&
           nnz <= stem_sparse_hit_capacity(n_img) && H >= 7 && W >= 7 && H <= 8192 && W <= 8192 && (ldo & 7) == 0 &&
           (long)n_img * cdiv(H, 32) * cdiv(W, 32) < (1L << 30);
}
int stem_sparse_stats_grid(const StemSparseArgs& a) { return region_grid((long)a.n_img * cdiv(a.Hc, ST_RH) * cdiv(a.Wc, ST_RW)); }
int stem_sparse_pool_grid(const StemSparseArgs& a) { return region_grid((long)a.n_img * cdiv(a.Ho, PL_PH) * cdiv(a.Wo, PL_PW)); }
int stem_sparse_bwd_grid(const StemSparseArgs& a) { return region_grid((long)a.n_img * cdiv(a.Hc, ST_RH) * cdiv(a.Wc, ST_RW)); }

int stem_sparse_index(const StemSparseArgs& a, hipStream_t st) {
    const long ncells = (long)a.n_img * a.cells_y * a.cells_x;
    TCVN_CHECK(hipMemsetAsync(a.cell_fill, 0, (size_t)ncells * 4, st));
    if (a.nnz > 0) {
        hipLaunchKernelGGL(k_stem_index_count, dim3(cdiv(a.nnz, 256)), dim3(256), 0, st, a);
        TCVN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_stem_index_scan, dim3(1), dim3(1024), 0, st, a, (int)ncells);
    TCVN_LAUNCH_CHECK();
    if (a.nnz > 0) {
        hipLaunchKernelGGL(k_stem_index_fill, dim3(cdiv(a.nnz, 256)), dim3(256), 0, st, a);
        TCVN_LAUNCH_CHECK();
    }
    return 0;
}
int stem_sparse_stats(const StemSparseArgs& a, hipStream_t st) {
    const int ty = cdiv(a.Hc, ST_RH), tx = cdiv(a.Wc, ST_RW);
    ProfScope ps("k_stem_sparse_stats", 2.0 * a.nnz * 12.25 * a.Cpix * SS_N, (double)a.nnz * (12.0 + 4.0 * a.Cpix), st);
    hipLaunchKernelGGL(k_stem_sparse_stats, dim3(stem_sparse_stats_grid(a)), dim3(256), 0, st, a, ty, tx);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int stem_sparse_pool(const StemSparseArgs& a, hipStream_t st) {
    const int ty = cdiv(a.Ho, PL_PH), tx = cdiv(a.Wo, PL_PW);
    ProfScope ps("k_stem_sparse_pool", 2.0 * a.nnz * 12.25 * a.Cpix * SS_N, (double)a.n_img * a.Ho * a.Wo * SS_N * 2.0 + (double)a.nnz * (12.0 + 4.0 * a.Cpix), st);
    hipLaunchKernelGGL(k_stem_sparse_pool, dim3(stem_sparse_pool_grid(a)), dim3(256), 0, st, a, ty, tx);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int stem_sparse_bwd(const StemSparseArgs& a, int pass, hipStream_t st) {
    const int ty = cdiv(a.Hc, ST_RH), tx = cdiv(a.Wc, ST_RW);
    const int nb = stem_sparse_bwd_grid(a);
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_sparse_bwd<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_sparse_bwd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr = true;
    }
    if (bwd_smem() > 80 * 1024) return -2;
    const double bytes = (double)a.n_img * a.Ho * a.Wo * SS_N * 2.0 * 2.0 + (double)a.nnz * (12.0 + 4.0 * a.Cpix);      // (G, x) of the pooled map + the hit list
    if (pass == 0) {
        ProfScope ps("k_stem_sparse_bwd<sums>", 2.0 * a.nnz * 12.25 * a.Cpix * SS_N, bytes, st);
        hipLaunchKernelGGL(k_stem_sparse_bwd<0>, dim3(nb), dim3(256), bwd_smem(), st, a, ty, tx);
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.slab == nullptr || (long)nb * SS_N * a.Kp * 4 > a.slab_bytes || a.Kp < SS_K) return -3;
    {
        ProfScope ps("k_stem_sparse_bwd<wgrad>", 4.0 * a.nnz * 12.25 * a.Cpix * SS_N, bytes, st);
        hipLaunchKernelGGL(k_stem_sparse_bwd<1>, dim3(nb), dim3(256), bwd_smem(), st, a, ty, tx);
        TCVN_LAUNCH_CHECK();
    }
    return slab_reduce(a.slab, nb, (long)SS_N * a.Kp, a.dWk, st);
}

}  // namespace tcvn
