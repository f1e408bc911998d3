// Sparse-aware stem (bf16 throughput mode): conv0 7x7/2 + BatchNorm0 + PReLU0 + AvgPool 3/2 and their backward straight from
// the COO hit list (reference: trainers/neutrino_full_dense_trainer.py:15-24 `sparse_to_dense` + layers/dense_net.py:112-121).
//
// The pixel maps are mostly empty: a 400 x 280 prong map holds 20-800 hits, an event map 500-4000.  The dense path scatters them into
// a [n,400,280,3] map, runs a dense MFMA convolution over all 28 000 output positions of every map, writes the [n,200,140,64] conv0
// output (1 GB for 288 maps), reads it again for the pooling and twice more in backward.  Here neither the dense map nor the conv0
// output exists in HBM, and the work is proportional to the hits:
//   * index (once per forward): the hits are bucketed by (map, pixel row) -- count, prefix sum, fill -- as 16-byte records
//     (y, x, three bf16 values, list index); duplicate coordinates keep the record with the highest list index (a valid outcome of the
//     reference's non-accumulating indexed write, and a deterministic one).
//   * every pass gives a workgroup one map (or a range of row bands of one map).  Per band it copies the band's records -- one
//     contiguous range of the index -- into LDS, lays out an occupancy bitmap and a pixel -> record map, and from then on works on chip:
//     a conv0 position whose 7x7 window holds no hit equals the bias exactly, so it is never evaluated; the non-empty positions are
//     collected in an ordered worklist and evaluated by eight threads each (8 channels per thread), one FMA chain per hit in window order.
//   k_stem_sparse_stats   sum / sum of squares of the conv0 output per channel (BatchNorm0's batch statistics)        [train only]
//   k_stem_sparse_pool    conv0 -> BN0 -> PReLU0 -> AvgPool(3, stride 2) -> first 64 channels of dense block 1 (+ their statistics);
//                         pooled pixels whose 11x11 input window is empty are one constant vector
//   k_stem_sparse_bwd<0>  pooling / PReLU0 / BN0 backward sums (sum dU, sum dU*x, sum dz*min(u,0)): the empty positions enter through
//                         the identity sum_p dz[p] = sum_q eff[q] (every pooled pixel spreads eff/9 over nine existing positions)
//   k_stem_sparse_bwd<1>  conv0 weight gradient: eff0 = sc*dU + P0*x + Q0 of the non-empty positions, contracted with the hits
// conv0's output stays fp32 on chip (the dense bf16 path rounds it to bf16 when it stores it).  Deterministic: integer atomics only,
// fixed summation orders.  HBM traffic per step: the index, the pooled map (written once, read in backward).
#include <cstdlib>
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int SS_N = 64;                        // conv0 output channels
constexpr int SS_K = 147;                       // 7*7*3
constexpr int SS_HCAP = 2048;                   // records of one band kept in LDS (longer bands read their records from the index)
constexpr int SS_BROWS = 23;                    // input rows of the tallest band (4 pooled rows: 4*4 + 7)
constexpr unsigned SS_DEAD = 0x80000000u;       // record flag (word z): a later list entry has the same pixel

// ---------------------------------------------------------------------------------------------------------------------
// index: row_start[map*H + y] .. row_start[map*H + y + 1] lists the hits of pixel row y of a map
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float preprocess_value(float v, int mode, float noise_std, uint64_t seed, long flat_index) {
    v = mode == 1 ? logf(v + 1.f) : mode == 0 ? v / 255.0f : v;        // same arithmetic as k_scatter (elementwise.hip)
    if (noise_std != 0.f) {
        const float u1 = fmaxf(rng_uniform(seed, 0x6e6f6973u, (uint64_t)flat_index * 2), 1e-7f);
        const float u2 = rng_uniform(seed, 0x6e6f6973u, (uint64_t)flat_index * 2 + 1);
        v *= 1.f + noise_std * sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
    }
    return v;                                                           // rounded to bf16 when the record is packed (= the dense bf16 map)
}

__global__ void k_stem_index_count(const StemSparseArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nnz) return;
    const int img = a.coords[i * 3], y = a.coords[i * 3 + 1], x = a.coords[i * 3 + 2];
    if (img < 0 || img >= a.n_img || y < 0 || y >= a.H || x < 0 || x >= a.W) return;
    atomicAdd(&a.row_fill[(long)img * a.H + y], 1);
}

// exclusive prefix sum of the per-row counts (one workgroup); resets the counters to zero for the fill pass.  Each of the 16 waves owns a
// contiguous region and walks it in 64-element strips (lane = consecutive element: coalesced) with a shuffle scan per strip and a running
// carry -- two passes over the counts.  (Before: one contiguous chunk of ~110 counts per THREAD, i.e. 64 different lines per wave load:
// 110-210 us per launch at 288 maps, a quarter of the eval-mode stem.)
__global__ __launch_bounds__(1024) void k_stem_index_scan(const StemSparseArgs a, int nbins) {
    __shared__ int wsum[16];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int R = ((nbins + 15) / 16 + 63) & ~63;              // region per wave, whole strips
    const int lo = wave * R, hi = min(nbins, lo + R);
    int s = 0;
    for (int i = lo + lane; i < hi; i += 64) s += a.row_fill[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    int carry = 0;
    for (int w = 0; w < wave; ++w) carry += wsum[w];
    for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        const int v = i < hi ? a.row_fill[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o);
            if (lane >= o) incl += n;
        }
        if (i < hi) { a.row_start[i] = carry + incl - v; a.row_fill[i] = 0; }
        carry += __shfl(incl, 63);
    }
    if (t == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += wsum[w]; a.row_start[nbins] = tot; }
}

__global__ void k_stem_index_fill(const StemSparseArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nnz) return;
    const int img = a.coords[i * 3], y = a.coords[i * 3 + 1], x = a.coords[i * 3 + 2];
    if (img < 0 || img >= a.n_img || y < 0 || y >= a.H || x < 0 || x >= a.W) return;
    const long bin = (long)img * a.H + y;
    const int pos = atomicAdd(&a.row_fill[bin], 1);
    float v[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < a.Cpix && c < 3; ++c) v[c] = preprocess_value(a.values[i * a.Cpix + c], a.value_mode, a.noise_std, a.seed, i * a.Cpix + c);
    // a row listed more than 65535 times (possible only with > 200 duplicates per pixel) keeps its first 65535 entries in arrival order:
    // still one of the duplicates per pixel, like the reference's write
    uint4 r;
    r.x = ((unsigned)y << 16) | (unsigned)x;
    r.y = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    r.z = (unsigned)f2bf(v[2]) | (pos >= 65535 ? SS_DEAD : 0u);
    r.w = (unsigned)i;
    a.rec[a.row_start[bin] + pos] = r;                        // order inside a row is arbitrary: lookups go through the pixel map
}

// duplicates: of the records of one pixel only the one with the highest list index stays alive
__global__ void k_stem_index_dedup(const StemSparseArgs a, const int* total) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *total) return;
    const uint4 r = a.rec[k];
    const long bin = (long)a.coords[(long)r.w * 3] * a.H + (r.x >> 16);
    const int k0 = a.row_start[bin], k1 = a.row_start[bin + 1];
    bool dead = false;
    for (int j = k0; j < k1 && !dead; ++j) {
        const uint4 o = a.rec[j];
        dead = o.x == r.x && o.w > r.w && j - k0 < 65535;
    }
    if (dead) a.rec[k].z = r.z | SS_DEAD;
}

// ---------------------------------------------------------------------------------------------------------------------
// band machinery shared by the passes (all LDS pointers are carved from one dynamic array by the kernels)
// ---------------------------------------------------------------------------------------------------------------------
struct BandLds {
    uint4* recs;            // [SS_HCAP] records of the band (when they fit)
    int* rowbase;           // [SS_BROWS + 1] first record of each band row, relative to the band's first record
    unsigned* bits;         // [SS_BROWS][bw] occupancy bitmap, one zero word in front of and two behind every row
    unsigned short* map;    // [SS_BROWS][W] position of the pixel's record inside its row (valid where the bit is set)
    int bw;                 // words per bitmap row incl. padding
};
struct Band { int in_lo, in_hi, k0, nrec; bool in_lds; };

// input rows [in_lo, in_hi] of map img (clipped to the map by the caller; in_hi - in_lo < SS_BROWS)
__device__ __forceinline__ Band load_band(const StemSparseArgs& a, const BandLds& L, int img, int in_lo, int in_hi, int tid) {
    Band b;
    b.in_lo = in_lo; b.in_hi = in_hi;
    const int* rs = a.row_start + (long)img * a.H;
    b.k0 = rs[in_lo];
    b.nrec = rs[in_hi + 1] - b.k0;
    b.in_lds = b.nrec <= SS_HCAP;
    const int nrows = in_hi - in_lo + 1;
    __syncthreads();                                          // the previous band's readers are done
    for (int i = tid; i < nrows * L.bw; i += 256) L.bits[i] = 0u;
    if (tid <= nrows) L.rowbase[tid] = rs[in_lo + tid] - b.k0;
    __syncthreads();
    for (int k = tid; k < b.nrec; k += 256) {
        const uint4 r = a.rec[b.k0 + k];
        if (b.in_lds) L.recs[k] = r;
        if (!(r.z & SS_DEAD)) {
            const int yr = (int)(r.x >> 16) - in_lo, x = (int)(r.x & 0xffffu);
            atomicOr(&L.bits[yr * L.bw + 1 + (x >> 5)], 1u << (x & 31));
            L.map[yr * a.W + x] = (unsigned short)(k - L.rowbase[yr]);
        }
    }
    __syncthreads();
    return b;
}

// bits [x0, x0 + n) (n <= 16) of pixel row y; rows outside the band and columns outside the map read as zero
__device__ __forceinline__ unsigned rowbits(const BandLds& L, const Band& b, int y, int x0, int n) {
    if (y < b.in_lo || y > b.in_hi) return 0u;
    const unsigned* row = L.bits + (y - b.in_lo) * L.bw + 1;
    const int w = x0 >> 5, sh = x0 & 31;                      // x0 >= -3: w >= -1 (the zero word in front)
    const unsigned long long v = (unsigned long long)row[w] | ((unsigned long long)row[w + 1] << 32);
    return (unsigned)(v >> sh) & ((1u << n) - 1u);
}

__device__ __forceinline__ uint4 band_record(const StemSparseArgs& a, const BandLds& L, const Band& b, int y, int x) {
    const int yr = y - b.in_lo;
    const int k = L.rowbase[yr] + (int)L.map[yr * a.W + x];
    return b.in_lds ? L.recs[k] : a.rec[b.k0 + k];
}

// conv0 weights -> LDS [147][64] bf16 (k = (ky*7 + kx)*3 + c), bias
__device__ __forceinline__ void load_weights(const StemSparseArgs& a, bf16* wl, float* bias_l, int tid) {
    const bf16* Wk = reinterpret_cast<const bf16*>(a.Wk);
    for (int i = tid; i < SS_K * SS_N; i += 256) {
        const int k = i >> 6, n = i & 63;
        wl[i] = Wk[(long)n * a.Kp + k];
    }
    if (tid < SS_N) bias_l[tid] = a.bias[tid];
}

// acc[j] += v[c] * w[tap*3 + c][c8*8 + j]
__device__ __forceinline__ void fma_tap(const bf16* wl, int tap, int c8, const float (&v)[3], float (&acc)[8]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const u16x8 w = *reinterpret_cast<const u16x8*>(wl + (tap * 3 + c) * SS_N + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[c], bf2f(w[j]), acc[j]);
    }
}

// conv0 output of position (h, w) of the map, channels [c8*8, c8*8 + 8): bias + the hits of its 7x7 window in window order
__device__ __forceinline__ void conv0_at(const StemSparseArgs& a, const BandLds& L, const Band& b, const bf16* wl, const float* bias_l,
                                         int h, int w, int c8, float (&acc)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias_l[c8 * 8 + j];
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        const int y = 2 * h - 3 + ky;
        unsigned m = rowbits(L, b, y, 2 * w - 3, 7);
        while (m) {
            const int kx = __ffs(m) - 1;
            m &= m - 1;
            const uint4 r = band_record(a, L, b, y, 2 * w - 3 + kx);
            const float v[3] = {bf2f((bf16)(r.y & 0xffffu)), bf2f((bf16)(r.y >> 16)), bf2f((bf16)(r.z & 0xffffu))};
            fma_tap(wl, ky * 7 + kx, c8, v, acc);
        }
    }
}

// ordered worklist of the candidates [0, n) for which `test(i)` holds: wlist[.] = i ascending; returns the count (uniform)
template <typename F>
__device__ __forceinline__ int build_worklist(unsigned short* wlist, int* cnt, int n, int tid, F test) {
    const int lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int r0 = 0; r0 < n; r0 += 256) {
        const int i = r0 + tid;
        const bool has = i < n && test(i);
        const unsigned long long m = __ballot(has);
        if (lane == 0) cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += cnt[w];
        if (has) wlist[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
        base += cnt[0] + cnt[1] + cnt[2] + cnt[3];
        __syncthreads();
    }
    return base;
}

// carve the band structures out of dynamic LDS; returns the first free byte
__device__ __forceinline__ char* carve_band(char* p, BandLds& L, int W) {
    L.recs = reinterpret_cast<uint4*>(p); p += SS_HCAP * 16;
    L.bw = (W + 31) / 32 + 3;
    L.bits = reinterpret_cast<unsigned*>(p); p += ((SS_BROWS * L.bw * 4 + 15) & ~15);
    L.map = reinterpret_cast<unsigned short*>(p); p += ((SS_BROWS * W * 2 + 15) & ~15);
    L.rowbase = reinterpret_cast<int*>(p); p += 128;
    return p;
}
size_t band_bytes(int W) { return (size_t)SS_HCAP * 16 + (((size_t)SS_BROWS * ((W + 31) / 32 + 3) * 4 + 15) & ~15) + (((size_t)SS_BROWS * W * 2 + 15) & ~15) + 128; }

// work units: (map, split) -> bands [b0, b1) of nb bands
__device__ __forceinline__ void unit_bands(int unit, int nsplit, int nb, int& img, int& b0, int& b1) {
    img = unit / nsplit;
    const int s = unit - img * nsplit;
    b0 = (int)((long)s * nb / nsplit); b1 = (int)((long)(s + 1) * nb / nsplit);
}

// ---------------------------------------------------------------------------------------------------------------------
// pass 1 (train): per-channel sum and sum of squares of the conv0 output over all Hc x Wc positions of every map
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ST_CR = 8;                                     // conv0 rows per band (21 input rows)

__global__ __launch_bounds__(256, 2) void k_stem_sparse_stats(const StemSparseArgs a, int nsplit, int nunits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BandLds L;
    char* p = carve_band(smem, L, a.W);
    bf16* wl = reinterpret_cast<bf16*>(p); p += SS_K * SS_N * 2;
    float* bias_l = reinterpret_cast<float*>(p); p += SS_N * 4;
    unsigned short* wlist = reinterpret_cast<unsigned short*>(p); p += ((ST_CR * a.Wc * 2 + 15) & ~15);
    int* cnt = reinterpret_cast<int*>(p); p += 32;
    double* red = reinterpret_cast<double*>(p);              // [4 waves][8][8 x 2]
    const int tid = threadIdx.x, c8 = tid & 7;
    load_weights(a, wl, bias_l, tid);
    const int nb = (a.Hc + ST_CR - 1) / ST_CR;
    double s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; }
    long n_empty = 0;                                        // counted by thread 0 only
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        int img, b0, b1;
        unit_bands(unit, nsplit, nb, img, b0, b1);
        for (int bi = b0; bi < b1; ++bi) {
            const int h0 = bi * ST_CR, nh = min(ST_CR, a.Hc - h0), npos = nh * a.Wc;
            const Band b = load_band(a, L, img, max(0, 2 * h0 - 3), min(a.H - 1, 2 * (h0 + nh - 1) + 3), tid);
            const int nw = build_worklist(wlist, cnt, npos, tid, [&](int i) {
                const int h = h0 + i / a.Wc, w = i % a.Wc;
                unsigned any = 0;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) any |= rowbits(L, b, 2 * h - 3 + ky, 2 * w - 3, 7);
                return any != 0u;
            });
            if (tid == 0) n_empty += npos - nw;
            for (int t = tid; t < nw * 8; t += 256) {
                const int i = wlist[t >> 3];
                float acc[8];
                conv0_at(a, L, b, wl, bias_l, h0 + i / a.Wc, i % a.Wc, c8, acc);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] += (double)acc[j] * acc[j]; }
            }
        }
    }
    __syncthreads();
    // lanes with equal (tid & 7) hold the same channels: fold lane bits 3..5, then the four waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
        if ((tid & 63) < 8) { red[(((tid >> 6) * 8 + c8) * 8 + j) * 2] = s1[j]; red[(((tid >> 6) * 8 + c8) * 8 + j) * 2 + 1] = s2[j]; }
    }
    if (tid == 0) reinterpret_cast<long*>(cnt)[1] = n_empty;
    __syncthreads();
    if (tid < SS_N) {
        const int cc = tid >> 3, j = tid & 7;
        double x = 0, y = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) { x += red[((g * 8 + cc) * 8 + j) * 2]; y += red[((g * 8 + cc) * 8 + j) * 2 + 1]; }
        const double ne = (double)reinterpret_cast<long*>(cnt)[1], bb = (double)bias_l[tid];     // empty positions hold the bias exactly
        a.part[((long)blockIdx.x * SS_N + tid) * 2] = x + ne * bb;
        a.part[((long)blockIdx.x * SS_N + tid) * 2 + 1] = y + ne * bb * bb;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// pass 2: pooled, activated map.  Band = 4 pooled rows (23 input rows).  A pooled pixel covers conv0 rows 2q..2q+2 = input rows
// 4q-3 .. 4q+7; if that 11 x 11 window holds no hit its value is avgpool of nine copies of prelu(bn(bias)).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PL_PR = 4;

__global__ __launch_bounds__(256, 2) void k_stem_sparse_pool(const StemSparseArgs a, int nsplit, int nunits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BandLds L;
    char* p = carve_band(smem, L, a.W);
    bf16* wl = reinterpret_cast<bf16*>(p); p += SS_K * SS_N * 2;
    float* bias_l = reinterpret_cast<float*>(p); p += SS_N * 4;
    float* tab = reinterpret_cast<float*>(p); p += 3 * SS_N * 4;           // BatchNorm0 scale, shift; PReLU0 slope
    bf16* cvec = reinterpret_cast<bf16*>(p); p += SS_N * 2;                // pooled value of an all-empty window
    unsigned short* wlist = reinterpret_cast<unsigned short*>(p); p += ((PL_PR * a.Wo * 2 + 15) & ~15);
    unsigned char* flag = reinterpret_cast<unsigned char*>(p); p += ((PL_PR * a.Wo + 15) & ~15);
    int* cnt = reinterpret_cast<int*>(p); p += 32;
    double* red = reinterpret_cast<double*>(p);                           // [4 waves][8][8 x 2]
    const int tid = threadIdx.x, c8 = tid & 7;
    load_weights(a, wl, bias_l, tid);
    if (tid < SS_N) { tab[tid] = a.sc[tid]; tab[64 + tid] = a.sh[tid]; tab[128 + tid] = a.sl[tid]; }
    __syncthreads();
    if (tid < SS_N) {
        const float ae = prelu(fmaf(bias_l[tid], tab[tid], tab[64 + tid]), tab[128 + tid]);
        float s = 0.f;
        for (int i = 0; i < 9; ++i) s += ae;                              // the same nine additions the general path makes
        cvec[tid] = f2bf(s * (1.0f / 9.0f));
    }
    float sc[8], sh[8], sl[8];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = tab[c8 * 8 + j]; sh[j] = tab[64 + c8 * 8 + j]; sl[j] = tab[128 + c8 * 8 + j]; }
    bf16* __restrict__ Out = reinterpret_cast<bf16*>(a.Out);
    const int nb = (a.Ho + PL_PR - 1) / PL_PR;
    double s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0; s2[j] = 0; }
    long n_empty = 0;
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        int img, b0, b1;
        unit_bands(unit, nsplit, nb, img, b0, b1);
        for (int bi = b0; bi < b1; ++bi) {
            const int q0 = bi * PL_PR, nq = min(PL_PR, a.Ho - q0), npix = nq * a.Wo;
            const Band b = load_band(a, L, img, max(0, 4 * q0 - 3), min(a.H - 1, 4 * (q0 + nq - 1) + 7), tid);
            const int nw = build_worklist(wlist, cnt, npix, tid, [&](int i) {
                const int qy = q0 + i / a.Wo, qx = i % a.Wo;
                unsigned any = 0;
#pragma unroll
                for (int wy = 0; wy < 11; ++wy) any |= rowbits(L, b, 4 * qy - 3 + wy, 4 * qx - 3, 11);
                flag[i] = any != 0u;
                return any != 0u;
            });
            if (tid == 0) n_empty += npix - nw;
            // non-empty pooled pixels: the nine conv0 positions from the hits of the 11 x 11 window, window order
            for (int t = tid; t < nw * 8; t += 256) {
                const int i = wlist[t >> 3];
                const int qy = q0 + i / a.Wo, qx = i % a.Wo;
                float acc[9][8];
#pragma unroll
                for (int ps = 0; ps < 9; ++ps)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[ps][j] = bias_l[c8 * 8 + j];
#pragma unroll 1
                for (int wy = 0; wy < 11; ++wy) {
                    const int y = 4 * qy - 3 + wy;
                    unsigned m = rowbits(L, b, y, 4 * qx - 3, 11);
                    while (m) {
                        const int wx = __ffs(m) - 1;
                        m &= m - 1;
                        const uint4 r = band_record(a, L, b, y, 4 * qx - 3 + wx);
                        const float v[3] = {bf2f((bf16)(r.y & 0xffffu)), bf2f((bf16)(r.y >> 16)), bf2f((bf16)(r.z & 0xffffu))};
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int ky = wy - 2 * dy;
                            if (ky < 0 || ky > 6) continue;
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) {
                                const int kx = wx - 2 * dx;
                                if (kx < 0 || kx > 6) continue;
                                fma_tap(wl, ky * 7 + kx, c8, v, acc[dy * 3 + dx]);
                            }
                        }
                    }
                }
                float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ps = 0; ps < 9; ++ps)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += prelu(fmaf(acc[ps][j], sc[j], sh[j]), sl[j]);
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o[j] = f2bf(s[j] * (1.0f / 9.0f));
                    const double x = (double)bf2f(o[j]);
                    s1[j] += x; s2[j] += x * x;
                }
                *reinterpret_cast<u16x8*>(Out + (((long)img * a.Ho + qy) * a.Wo + qx) * a.ldo + c8 * 8) = o;
            }
            // empty pooled pixels: the constant vector
            const u16x8 cv = *reinterpret_cast<const u16x8*>(cvec + c8 * 8);
            for (int t = tid; t < npix * 8; t += 256) {
                const int i = t >> 3;
                if (!flag[i]) *reinterpret_cast<u16x8*>(Out + (((long)img * a.Ho + q0 + i / a.Wo) * a.Wo + i % a.Wo) * a.ldo + c8 * 8) = cv;
            }
        }
    }
    if (a.part == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
        if ((tid & 63) < 8) { red[(((tid >> 6) * 8 + c8) * 8 + j) * 2] = s1[j]; red[(((tid >> 6) * 8 + c8) * 8 + j) * 2 + 1] = s2[j]; }
    }
    if (tid == 0) reinterpret_cast<long*>(cnt)[1] = n_empty;
    __syncthreads();
    if (tid < SS_N) {
        const int cc = tid >> 3, j = tid & 7;
        double x = 0, y = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) { x += red[((g * 8 + cc) * 8 + j) * 2]; y += red[((g * 8 + cc) * 8 + j) * 2 + 1]; }
        const double ne = (double)reinterpret_cast<long*>(cnt)[1], cvv = (double)bf2f(cvec[tid]);
        a.part[((long)blockIdx.x * SS_N + tid) * 2] = x + ne * cvv;
        a.part[((long)blockIdx.x * SS_N + tid) * 2 + 1] = y + ne * cvv * cvv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  dz[p] = 1/9 sum over the <= 4 pooled pixels q whose 3x3/2 window holds position p of eff[q] = G[q] + P*D[q] + Q.
// PASS 0: sums (dU, dU*x, dz*min(u,0)) per channel: explicit over the non-empty positions; over the empty ones (x = bias, u = u_e)
//         they are slope_e * S, bias * slope_e * S and min(u_e, 0) * S with S = sum over empty positions of dz = sum_q eff[q] - sum over
//         non-empty positions of dz (every pooled pixel spreads eff/9 over nine existing positions).
// PASS 1: eff0 = sc*dU + P0*x + Q0 of the non-empty positions (chunks of 64 in LDS), contracted with the hits of their windows:
//         dW0[n][ky][kx][c] += v[hit][c] * eff0[p][n]; wave w owns the kernel rows ky = 2w, 2w + 1 of an LDS accumulator [147][64].
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BW_CHUNK = 64;

template <int PASS>
__global__ __launch_bounds__(256, 2) void k_stem_sparse_bwd(const StemSparseArgs a, int nsplit, int nunits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BandLds L;
    char* p = carve_band(smem, L, a.W);
    bf16* wl = reinterpret_cast<bf16*>(p); p += SS_K * SS_N * 2;
    float* bias_l = reinterpret_cast<float*>(p); p += SS_N * 4;
    float* ctab = reinterpret_cast<float*>(p); p += 7 * SS_N * 4;          // sc, sh, sl of BN0/PReLU0; P, Q of the pooled map; P0, Q0
    unsigned short* wlist = reinterpret_cast<unsigned short*>(p); p += ((ST_CR * a.Wc * 2 + 15) & ~15);
    int* cnt = reinterpret_cast<int*>(p); p += 32;
    float* eff0 = reinterpret_cast<float*>(p); p += PASS ? BW_CHUNK * SS_N * 4 : 0;       // [64][64]
    float* wsum = reinterpret_cast<float*>(p); p += PASS ? SS_K * SS_N * 4 : 0;            // [147][64]
    double* red = reinterpret_cast<double*>(smem);                                        // PASS 0, after the last band: [4 waves][8][8 x 5] over the records
    const int tid = threadIdx.x, c8 = tid & 7, lane = tid & 63, wave = tid >> 6;
    load_weights(a, wl, bias_l, tid);
    if (tid < SS_N) {
        ctab[tid] = a.sc[tid]; ctab[64 + tid] = a.sh[tid]; ctab[128 + tid] = a.sl[tid];
        ctab[192 + tid] = a.e.P[tid]; ctab[256 + tid] = a.e.Q[tid];
        ctab[320 + tid] = PASS ? a.P0[tid] : 0.f; ctab[384 + tid] = PASS ? a.Q0[tid] : 0.f;
    }
    if (PASS) for (int i = tid; i < SS_K * SS_N; i += 256) wsum[i] = 0.f;
    __syncthreads();
    float sc[8], sh[8], sl[8], cP[8], cQ[8], p0[PASS ? 8 : 1], q0[PASS ? 8 : 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = ctab[c8 * 8 + j]; sh[j] = ctab[64 + c8 * 8 + j]; sl[j] = ctab[128 + c8 * 8 + j];
        cP[j] = ctab[192 + c8 * 8 + j]; cQ[j] = ctab[256 + c8 * 8 + j];
        if (PASS) { p0[PASS ? j : 0] = ctab[320 + c8 * 8 + j]; q0[PASS ? j : 0] = ctab[384 + c8 * 8 + j]; }
    }
    const bf16* __restrict__ G = reinterpret_cast<const bf16*>(a.e.G);
    const bf16* __restrict__ D = reinterpret_cast<const bf16*>(a.e.X);
    auto eff_at = [&](int img, int qy, int qx, float (&e8)[8]) {           // gradient of pooled pixel (qy, qx), this thread's 8 channels
        const long mo = ((long)img * a.Ho + qy) * a.Wo + qx;
        float gv[8], dv[8];
        load8<bf16>(G + mo * a.e.ldg + c8 * 8, gv);
        load8<bf16>(D + mo * a.e.ldx + c8 * 8, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) e8[j] = gv[j] + cP[j] * dv[j] + cQ[j];
    };
    // dz of conv0 position (h, w): windows containing row h are ho = h/2 and, when h is even, h/2 - 1 (rows 2ho .. 2ho + 2)
    auto dz_at = [&](int img, int h, int w, float (&dz)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] = 0.f;
#pragma unroll 1
        for (int d = 0; d < 4; ++d) {
            const int dy = d >> 1, dx = d & 1;
            const int qy = h / 2 - dy, qx = w / 2 - dx;
            if ((dy == 1 && (h & 1)) || qy < 0 || qy >= a.Ho || (dx == 1 && (w & 1)) || qx < 0 || qx >= a.Wo) continue;
            float e8[8];
            eff_at(img, qy, qx, e8);
#pragma unroll
            for (int j = 0; j < 8; ++j) dz[j] += e8[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) dz[j] *= (1.0f / 9.0f);
    };
    const int nb = (a.Hc + ST_CR - 1) / ST_CR;
    double t1[8], t2[8], t3[8], tz[8], tq[8];          // non-empty positions: sums dU, dU*x, dz*min(u,0), dz;  pooled pixels: eff
#pragma unroll
    for (int j = 0; j < 8; ++j) { t1[j] = 0; t2[j] = 0; t3[j] = 0; tz[j] = 0; tq[j] = 0; }
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        int img, b0, b1;
        unit_bands(unit, nsplit, nb, img, b0, b1);
        if (!PASS) {       // sum_q eff[q] over this unit's share of the pooled rows (same split fractions as the bands)
            const int qa = (int)((long)(unit - img * nsplit) * a.Ho / nsplit), qb = (int)((long)(unit - img * nsplit + 1) * a.Ho / nsplit);
#pragma unroll 1
            for (int t = tid; t < (qb - qa) * a.Wo * 8; t += 256) {
                const int i = t >> 3;
                float e8[8];
                eff_at(img, qa + i / a.Wo, i % a.Wo, e8);
#pragma unroll
                for (int j = 0; j < 8; ++j) tq[j] += e8[j];
            }
        }
        for (int bi = b0; bi < b1; ++bi) {
            const int h0 = bi * ST_CR, nh = min(ST_CR, a.Hc - h0), npos = nh * a.Wc;
            const Band b = load_band(a, L, img, max(0, 2 * h0 - 3), min(a.H - 1, 2 * (h0 + nh - 1) + 3), tid);
            const int nw = build_worklist(wlist, cnt, npos, tid, [&](int i) {
                const int h = h0 + i / a.Wc, w = i % a.Wc;
                unsigned any = 0;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) any |= rowbits(L, b, 2 * h - 3 + ky, 2 * w - 3, 7);
                return any != 0u;
            });
            for (int c0 = 0; c0 < nw; c0 += (PASS ? BW_CHUNK : nw)) {
                const int cn = PASS ? min(BW_CHUNK, nw - c0) : nw;
#pragma unroll 1
                for (int t = tid; t < cn * 8; t += 256) {
                    const int i = wlist[c0 + (t >> 3)];
                    const int h = h0 + i / a.Wc, w = i % a.Wc;
                    float x[8], dz[8];
                    conv0_at(a, L, b, wl, bias_l, h, w, c8, x);
                    dz_at(img, h, w, dz);
                    float ef[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float u = fmaf(x[j], sc[j], sh[j]);
                        const float du = u > 0.f ? dz[j] : sl[j] * dz[j];
                        if (!PASS) { t1[j] += du; t2[j] += (double)du * x[j]; t3[j] += u > 0.f ? 0.f : dz[j] * u; tz[j] += dz[j]; }
                        ef[j] = PASS ? fmaf(sc[j], du, fmaf(p0[PASS ? j : 0], x[j], q0[PASS ? j : 0])) : 0.f;
                    }
                    if (PASS) {
                        float4* o = reinterpret_cast<float4*>(eff0 + (t >> 3) * SS_N + c8 * 8);
                        o[0] = make_float4(ef[0], ef[1], ef[2], ef[3]); o[1] = make_float4(ef[4], ef[5], ef[6], ef[7]);
                    }
                }
                if (PASS) {
                    __syncthreads();
                    // every wave walks the chunk's (position, hit) pairs in the same order and adds the pairs of its own kernel rows
                    for (int s = 0; s < cn; ++s) {
                        const int i = wlist[c0 + s];
                        const int h = h0 + i / a.Wc, w = i % a.Wc;
                        const float ef = eff0[s * SS_N + lane];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const int ky = 2 * wave + kk;
                            if (ky > 6) continue;
                            const int y = 2 * h - 3 + ky;
                            unsigned m = rowbits(L, b, y, 2 * w - 3, 7);
                            while (m) {
                                const int kx = __ffs(m) - 1;
                                m &= m - 1;
                                const uint4 r = band_record(a, L, b, y, 2 * w - 3 + kx);
                                const float v[3] = {bf2f((bf16)(r.y & 0xffffu)), bf2f((bf16)(r.y >> 16)), bf2f((bf16)(r.z & 0xffffu))};
                                float* d = wsum + ((ky * 7 + kx) * 3) * SS_N + lane;
                                d[0] = fmaf(v[0], ef, d[0]); d[SS_N] = fmaf(v[1], ef, d[SS_N]); d[2 * SS_N] = fmaf(v[2], ef, d[2 * SS_N]);
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }
    __syncthreads();
    if (!PASS) {
        double* r = red + (wave * 8 + c8) * 40;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                t1[j] += __shfl_xor(t1[j], o); t2[j] += __shfl_xor(t2[j], o); t3[j] += __shfl_xor(t3[j], o);
                tz[j] += __shfl_xor(tz[j], o); tq[j] += __shfl_xor(tq[j], o);
            }
            if (lane < 8) { r[j * 5] = t1[j]; r[j * 5 + 1] = t2[j]; r[j * 5 + 2] = t3[j]; r[j * 5 + 3] = tz[j]; r[j * 5 + 4] = tq[j]; }
        }
        __syncthreads();
        if (tid < SS_N) {
            const int cc = tid >> 3, j = tid & 7;
            double x1 = 0, x2 = 0, x3 = 0, xz = 0, xq = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double* q = red + (g * 8 + cc) * 40 + j * 5;
                x1 += q[0]; x2 += q[1]; x3 += q[2]; xz += q[3]; xq += q[4];
            }
            const double bb = (double)bias_l[tid], ue = (double)fmaf(bias_l[tid], ctab[tid], ctab[64 + tid]);
            const double slope_e = ue > 0.0 ? 1.0 : (double)ctab[128 + tid];
            const double S = xq - xz;                                      // sum of dz over the empty positions
            double* o = a.part + ((long)blockIdx.x * SS_N + tid) * 3;
            o[0] = x1 + slope_e * S;
            o[1] = x2 + bb * slope_e * S;
            o[2] = x3 + (ue > 0.0 ? 0.0 : ue * S);
        }
    } else {
        float* out = a.slab + (long)blockIdx.x * SS_N * a.Kp;              // slab[block][n][Kp], k = (ky*7 + kx)*3 + c
        for (int i = tid; i < SS_N * a.Kp; i += 256) {
            const int nn = i / a.Kp, k = i - nn * a.Kp;
            out[i] = k < SS_K ? wsum[k * SS_N + nn] : 0.f;
        }
    }
}

size_t common_smem(int W) { return band_bytes(W) + (size_t)SS_K * SS_N * 2 + SS_N * 4; }
size_t stats_smem(const StemSparseArgs& a) { return common_smem(a.W) + ((ST_CR * a.Wc * 2 + 15) & ~15) + 32 + 4 * 8 * 16 * 8; }
size_t pool_smem(const StemSparseArgs& a) {
    return common_smem(a.W) + 3 * SS_N * 4 + SS_N * 2 + ((PL_PR * a.Wo * 2 + 15) & ~15) + ((PL_PR * a.Wo + 15) & ~15) + 32 + 4 * 8 * 16 * 8;
}
size_t bwd_smem(const StemSparseArgs& a, int pass) {
    return common_smem(a.W) + 7 * SS_N * 4 + ((ST_CR * a.Wc * 2 + 15) & ~15) + 32 +
           (pass ? (size_t)BW_CHUNK * SS_N * 4 + (size_t)SS_K * SS_N * 4 : (size_t)0);
}

// (map, split) work units: enough of them to fill the chip twice, never more splits than bands
int split_count(int n_img, int nbands) {
    int ns = (512 + n_img - 1) / n_img;
    return ns < 1 ? 1 : ns > nbands ? nbands : ns;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
long stem_sparse_hit_capacity(int n_img) { return (long)n_img * 8192; }                 // average occupancy up to 7 % of a 400 x 280 map
long stem_sparse_index_bytes(int n_img, int H, int W) {
    const long nbins = (long)n_img * H;
    return round_up((nbins + 1) * 4, 256) + round_up(nbins * 4, 256) + round_up(stem_sparse_hit_capacity(n_img) * 16, 256);
}
void stem_sparse_carve(StemSparseArgs& a, char* base) {
    const long nbins = (long)a.n_img * a.H;
    a.row_start = reinterpret_cast<int*>(base); base += round_up((nbins + 1) * 4, 256);
    a.row_fill = reinterpret_cast<int*>(base); base += round_up(nbins * 4, 256);
    a.rec = reinterpret_cast<uint4*>(base);
}
bool stem_sparse_ok(int mode, int in_ch, int init_ch, int H, int W, int value_mode, long nnz, int n_img, long ldo) {
    return conv3x3_tile_enabled() && mode == MODE_BF16 && in_ch >= 1 && in_ch <= 3 && init_ch == SS_N && value_mode >= 0 && value_mode <= 2 &&
           nnz <= stem_sparse_hit_capacity(n_img) && H >= 11 && W >= 11 && H <= 32767 && W <= 384 && (ldo & 7) == 0 &&
           (long)n_img * H < (1L << 30);
}
static int unit_grid(int nunits) { return nunits < 1024 ? nunits : 1024; }
int stem_sparse_stats_grid(const StemSparseArgs& a) { return unit_grid(a.n_img * split_count(a.n_img, cdiv(a.Hc, ST_CR))); }
int stem_sparse_pool_grid(const StemSparseArgs& a) { return unit_grid(a.n_img * split_count(a.n_img, cdiv(a.Ho, PL_PR))); }
int stem_sparse_bwd_grid(const StemSparseArgs& a) { return unit_grid(a.n_img * split_count(a.n_img, cdiv(a.Hc, ST_CR))); }

int stem_sparse_index(const StemSparseArgs& a, hipStream_t st) {
    const long nbins = (long)a.n_img * a.H;
    TCVN_CHECK(hipMemsetAsync(a.row_fill, 0, (size_t)nbins * 4, st));
    if (a.nnz > 0) {
        hipLaunchKernelGGL(k_stem_index_count, dim3(cdiv(a.nnz, 256)), dim3(256), 0, st, a);
        TCVN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_stem_index_scan, dim3(1), dim3(1024), 0, st, a, (int)nbins);
    TCVN_LAUNCH_CHECK();
    if (a.nnz > 0) {
        hipLaunchKernelGGL(k_stem_index_fill, dim3(cdiv(a.nnz, 256)), dim3(256), 0, st, a);
        TCVN_LAUNCH_CHECK();
        // the number of records (hits inside the maps) is row_start[nbins]: the kernel reads it on the device
        hipLaunchKernelGGL(k_stem_index_dedup, dim3(cdiv(a.nnz, 256)), dim3(256), 0, st, a, a.row_start + nbins);
        TCVN_LAUNCH_CHECK();
    }
    return 0;
}
template <typename K>
static int set_smem(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return -2;
    TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}
int stem_sparse_stats(const StemSparseArgs& a, hipStream_t st) {
    const int ns = split_count(a.n_img, cdiv(a.Hc, ST_CR)), nunits = a.n_img * ns;
    int rc;
    if ((rc = set_smem(k_stem_sparse_stats, stats_smem(a)))) return rc;
    ProfScope ps("k_stem_sparse_stats", 2.0 * a.nnz * 12.25 * a.Cpix * SS_N, (double)a.nnz * 16.0, st);
    hipLaunchKernelGGL(k_stem_sparse_stats, dim3(unit_grid(nunits)), dim3(256), stats_smem(a), st, a, ns, nunits);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int stem_sparse_pool(const StemSparseArgs& a, hipStream_t st) {
    const int ns = split_count(a.n_img, cdiv(a.Ho, PL_PR)), nunits = a.n_img * ns;
    int rc;
    if ((rc = set_smem(k_stem_sparse_pool, pool_smem(a)))) return rc;
    ProfScope ps("k_stem_sparse_pool", 2.0 * a.nnz * 12.25 * a.Cpix * SS_N * 2.25, (double)a.n_img * a.Ho * a.Wo * SS_N * 2.0 + (double)a.nnz * 16.0, st);
    hipLaunchKernelGGL(k_stem_sparse_pool, dim3(unit_grid(nunits)), dim3(256), pool_smem(a), st, a, ns, nunits);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int stem_sparse_bwd(const StemSparseArgs& a, int pass, hipStream_t st) {
    const int ns = split_count(a.n_img, cdiv(a.Hc, ST_CR)), nunits = a.n_img * ns;
    const int nb = unit_grid(nunits);
    int rc;
    const double bytes = (double)a.n_img * a.Ho * a.Wo * SS_N * 2.0 * 2.0 + (double)a.nnz * 16.0;      // (G, x) of the pooled map + the index
    if (pass == 0) {
        if ((rc = set_smem(k_stem_sparse_bwd<0>, bwd_smem(a, 0)))) return rc;
        ProfScope ps("k_stem_sparse_bwd<sums>", 2.0 * a.nnz * 12.25 * a.Cpix * SS_N, bytes, st);
        hipLaunchKernelGGL(k_stem_sparse_bwd<0>, dim3(nb), dim3(256), bwd_smem(a, 0), st, a, ns, nunits);
        TCVN_LAUNCH_CHECK();
        return 0;
    }
    if (a.slab == nullptr || (long)nb * SS_N * a.Kp * 4 > a.slab_bytes || a.Kp < SS_K) return -3;
    if ((rc = set_smem(k_stem_sparse_bwd<1>, bwd_smem(a, 1)))) return rc;
    {
        ProfScope ps("k_stem_sparse_bwd<wgrad>", 4.0 * a.nnz * 12.25 * a.Cpix * SS_N, bytes, st);
        hipLaunchKernelGGL(k_stem_sparse_bwd<1>, dim3(nb), dim3(256), bwd_smem(a, 1), st, a, ns, nunits);
        TCVN_LAUNCH_CHECK();
    }
    return slab_reduce(a.slab, nb, (long)SS_N * a.Kp, a.dWk, st);
}

}  // namespace tcvn
