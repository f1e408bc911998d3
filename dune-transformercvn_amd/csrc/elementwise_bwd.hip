// HBM-bound helper kernels of the DenseNet backward path.
#include "tcvn_ops.h"

namespace tcvn {

namespace {

// BatchNorm backward bookkeeping for one norm layer (train mode):
//   s1 = sum dU, t2 = sum dU*x, s3 = sum dA*min(u,0)   (partials from the dgrad epilogue / pooling backward kernels)
//   dbeta = s1 ; dgamma = sum dU*xhat = r*(t2 - mu*s1) ; dslope = s3
//   dx = sc*dU + Px*x + Qx  with  Px = -sc*dgamma*r/M ,  Qx = -sc*s1/M + sc*dgamma*r*mu/M      (sc = gamma*r)
__device__ __forceinline__ void bn_bwd_link_body(const BnBwdLinkArgs& a, int blk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blk * 4 + wave;
    if (c >= a.C) return;
    // everything the closing arithmetic needs is requested up front, with the partial rows (one round trip instead of two: see k_bn_link)
    const double mu = a.bstat[c * 2], var = a.bstat[c * 2 + 1];
    const float gam = a.gamma[c];
    const float g_dg = a.dgamma[c], g_db = a.dbeta[c], g_ds = a.dslope[c];
    const float p_old = a.accumulate_pq ? a.P[c] : 0.f, q_old = a.accumulate_pq ? a.Q[c] : 0.f;
    double s1 = 0, t2 = 0, s3 = 0;
    int b = lane;
    // Eight partial rows per trip = every row of a launch with <= 512 workgroups (all of them) in ONE round trip: 24 loads in flight
    // per lane.  This kernel sits between two convolutions of the critical chain while the other embedder's kernels keep the memory system
    // busy: each dependent trip cost a full loaded-latency round trip (fp32 mode: 25 us per launch with two trips, 132 launches per step).
    for (; b < a.nblk; b += 512) {
        double v[8][3];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bb = b + 64 * i;
            const double* p = a.part + ((long)(bb < a.nblk ? bb : b) * a.C + c) * 3;
            v[i][0] = p[0]; v[i][1] = p[1]; v[i][2] = p[2];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (b + 64 * i < a.nblk) { s1 += v[i][0]; t2 += v[i][1]; s3 += v[i][2]; }
    }
    s1 = wave_sum(s1); t2 = wave_sum(t2); s3 = wave_sum(s3);
    if (lane != 0) return;
    const double r = 1.0 / sqrt(var + (double)a.eps);
    const double dgamma = r * (t2 - mu * s1);
    const double sc = (double)gam * r;
    const double M = (double)a.count;
    a.dgamma[c] = g_dg + (float)dgamma;
    a.dbeta[c] = g_db + (float)s1;
    a.dslope[c] = g_ds + (float)s3;
    const float Px = (float)(-sc * dgamma * r / M);
    const float Qx = (float)(-sc * s1 / M + sc * dgamma * r * mu / M);
    a.P[c] = p_old + Px; a.Q[c] = q_old + Qx;
}
__global__ __launch_bounds__(256) void k_bn_bwd_link(const BnBwdLinkArgs a) { bn_bwd_link_body(a, blockIdx.x); }

// global-average head backward: dz = dF/HW on every pixel, then PReLU+BN backward bookkeeping of final_norm
constexpr int HP_CJ = 4;     // up to 1024 channels with 256 threads
template <typename T>
__global__ __launch_bounds__(256) void k_head_pool_bwd(const HeadPoolBwdArgs a) {
    const T* X = reinterpret_cast<const T*>(a.X);
    T* Gout = reinterpret_cast<T*>(a.Gout);
    double s1[HP_CJ], s2[HP_CJ], s3[HP_CJ];
#pragma unroll
    for (int j = 0; j < HP_CJ; ++j) { s1[j] = 0; s2[j] = 0; s3[j] = 0; }
    const float inv = 1.0f / (float)a.HW;
    for (int img = blockIdx.x; img < a.n_img; img += gridDim.x) {
#pragma unroll
        for (int j = 0; j < HP_CJ; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c >= a.C) continue;
            const float sc = a.sc[c], sh = a.sh[c], sl = a.sl[c];
            const float dz = a.dF[(long)img * a.C + c] * inv;
            for (int p = 0; p < a.HW; ++p) {
                const long m = (long)img * a.HW + p;
                const float x = to_f<T>(X[m * a.ldx + c]);
                const float u = fmaf(x, sc, sh);
                const float du = u > 0.f ? dz : sl * dz;
                s1[j] += du; s2[j] += (double)du * x; s3[j] += u > 0.f ? 0.f : dz * u;
                Gout[m * a.ldgo + c] = from_f<T>(sc * du);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < HP_CJ; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < a.C) {
            double* p = a.part + ((long)blockIdx.x * a.C + c) * 3;
            p[0] = s1[j]; p[1] = s2[j]; p[2] = s3[j];
        }
    }
}

// stem tail backward: AvgPool(3, s2) -> PReLU -> BN0, over the conv0 output pixels
constexpr int PB_CJ = 4;
template <typename T>
__global__ __launch_bounds__(256) void k_pool0_bwd(const Pool0BwdArgs a) {
    __shared__ double red[4][64 * PB_CJ][3];
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* G = reinterpret_cast<const T*>(a.e.G);
    const T* D = reinterpret_cast<const T*>(a.e.X);
    T* DU = reinterpret_cast<T*>(a.DU);
    double s1[PB_CJ], s2[PB_CJ], s3[PB_CJ];
#pragma unroll
    for (int j = 0; j < PB_CJ; ++j) { s1[j] = 0; s2[j] = 0; s3[j] = 0; }
    const long npix = (long)a.n_img * a.Hin * a.Win;
    for (long p = (long)blockIdx.x * 4 + pg; p < npix; p += (long)gridDim.x * 4) {
        const int w = (int)(p % a.Win);
        const int h = (int)((p / a.Win) % a.Hin);
        const long img = p / ((long)a.Win * a.Hin);
        const int ho_lo = max(0, (h - 1) / 2), ho_hi = min(a.Ho - 1, h / 2);     // windows 2*ho <= h <= 2*ho+2
        const int wo_lo = max(0, (w - 1) / 2), wo_hi = min(a.Wo - 1, w / 2);
#pragma unroll
        for (int j = 0; j < PB_CJ; ++j) {
            const int c = cl + 64 * j;
            if (c >= a.C) continue;
            float dz = 0.f;
            for (int ho = ho_lo; ho <= ho_hi; ++ho)
                for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                    if (2 * ho > h || 2 * ho + 2 < h || 2 * wo > w || 2 * wo + 2 < w) continue;
                    const long mo = (img * a.Ho + ho) * a.Wo + wo;
                    dz += to_f<T>(G[mo * a.e.ldg + c]) + a.e.P[c] * to_f<T>(D[mo * a.e.ldx + c]) + a.e.Q[c];
                }
            dz *= (1.0f / 9.0f);
            const float x = to_f<T>(X[p * a.C + c]);
            const float sc = a.sc[c];
            const float u = fmaf(x, sc, a.sh[c]);
            const float du = u > 0.f ? dz : a.sl[c] * dz;
            s1[j] += du; s2[j] += (double)du * x; s3[j] += u > 0.f ? 0.f : dz * u;
            DU[p * a.C + c] = from_f<T>(sc * du);
        }
    }
#pragma unroll
    for (int j = 0; j < PB_CJ; ++j) {
        red[pg][cl + 64 * j][0] = s1[j]; red[pg][cl + 64 * j][1] = s2[j]; red[pg][cl + 64 * j][2] = s3[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        double x = 0, y = 0, z = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { x += red[q][c][0]; y += red[q][c][1]; z += red[q][c][2]; }
        double* o = a.part + ((long)blockIdx.x * a.C + c) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

// materialised effective gradient (see EffMatArgs); block = W chunk-lanes (W = pow2 >= N/8) x 256/W rows, 16 B per thread
__global__ __launch_bounds__(256) void k_eff_mat(const EffMatArgs a, int wlog) {
    __shared__ float red[256][8];
    const EffSrc& e = a.e;
    const int W = 1 << wlog, rpb = 256 >> wlog;
    const int tx = threadIdx.x & (W - 1), ty = threadIdx.x >> wlog;
    const int cpr = (e.N + 7) >> 3;                        // the tail chunk is zero padded in Out
    const bf16* G = reinterpret_cast<const bf16*>(e.G);
    const bf16* X = reinterpret_cast<const bf16*>(e.X);
    bf16* O = reinterpret_cast<bf16*>(a.Out);
    float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tx < cpr) {
        float cP[8], cQ[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const bool ok = tx * 8 + j < e.N; cP[j] = ok ? e.P[tx * 8 + j] : 0.f; cQ[j] = ok ? e.Q[tx * 8 + j] : 0.f; }
        const uint32_t dkey = drop_key(e.seed, e.stream_id);
        for (long m = (long)blockIdx.x * rpb + ty; m < a.M; m += (long)gridDim.x * rpb) {
            const u16x8 gv = *reinterpret_cast<const u16x8*>(G + m * e.ldg + e.c_off + tx * 8);
            const u16x8 xv = *reinterpret_cast<const u16x8*>(X + m * e.ldx + e.c_off + tx * 8);
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = tx * 8 + j < e.N ? eff3(bf2f(gv[j]), cP[j], bf2f(xv[j]), cQ[j]) : 0.f;
                if (e.drop_p > 0.f) t *= drop_pick(drop_bits(dkey, m, tx * 8 + j, e.N), m, e.drop_p);
                o[j] = f2bf(t);
                cs[j] += bf2f(o[j]);
            }
            *reinterpret_cast<u16x8*>(O + m * a.ldo + tx * 8) = o;
        }
    }
    if (a.colsum == nullptr) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = cs[j];
    __syncthreads();
    if (ty == 0 && tx < cpr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float sum = 0.f;
            for (int q = 0; q < rpb; ++q) sum += red[q * W + tx][j];
            if (tx * 8 + j < e.N) a.slab[(long)blockIdx.x * e.N + tx * 8 + j] = sum;       // reduced by k_slab_reduce (same-line atomics serialise)
        }
    }
}

// dst[i] += sum_s slab[s*stride + i]: block = 64 columns x 4 slab lanes; grid.y splits the slabs (<= 16 atomics per element);
// grid.z selects one of two independent jobs (e.g. a weight-gradient slab and the bias column sums of the same layer: one
// launch instead of two -- these launches sit at the ~5 us dispatch floor)
__device__ __forceinline__ void slab_reduce_body(const SlabJob& j, float (*red)[256]) {
    if (j.count <= 0) return;
    const int cx = threadIdx.x & 63, sg = threadIdx.x >> 6;
    if ((int)blockIdx.y >= j.ny) return;                                               // uniform per workgroup
    const int s0 = blockIdx.y * j.per_y, s1 = min(j.nslab, s0 + j.per_y);
    const float* __restrict__ slab = j.slab;
    if (j.v4) {
        // 16 B per lane: a block covers 128 columns with eight slab lanes, eight loads in flight per thread.  (Round 4: the scalar version
        // moved the 37.7 MB slabs of a 3x3 weight gradient at 0.7 TB/s.  Round 5: with the slabs split over up to 16 y-slices the launch
        // spent its time in the fp32 atomics that merge the slices -- 557 K of them per dense layer; one y-slice per job (every launch of
        // the DenseNet plan: <= 512 slabs) adds into dst directly, in a fixed order: 19.40 -> 19.17 ms per step, tools/r05_ab2.sh.)
        const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
        const long i = ((long)blockIdx.x * 32 + cl) * 4;
        if ((long)blockIdx.x * 128 >= j.count) return;
        typedef __attribute__((ext_vector_type(4))) float f4;
        f4 acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
        if (i < j.count) {
            const float* base = slab + i;
            int k = s0 + sl;
            for (; k + 56 < s1; k += 64) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += *reinterpret_cast<const f4*>(base + (long)(k + 8 * q) * j.stride);
            }
            for (; k < s1; k += 8) acc[0] += *reinterpret_cast<const f4*>(base + (long)k * j.stride);
        }
        f4 (*red4)[32] = reinterpret_cast<f4 (*)[32]>(&red[0][0]);         // [8][32] f4 = the 4 x 256 floats of `red`
        red4[sl][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (sl == 0 && i < j.count) {
            const f4 v = ((red4[0][cl] + red4[1][cl]) + (red4[2][cl] + red4[3][cl])) + ((red4[4][cl] + red4[5][cl]) + (red4[6][cl] + red4[7][cl]));
            if (j.ny == 1) { f4* d = reinterpret_cast<f4*>(j.dst + i); *d = *d + v; }
            else { atomicAdd(j.dst + i, v.x); atomicAdd(j.dst + i + 1, v.y); atomicAdd(j.dst + i + 2, v.z); atomicAdd(j.dst + i + 3, v.w); }
        }
        return;
    }
    const long i = (long)blockIdx.x * 64 + cx;
    if ((long)blockIdx.x * 64 >= j.count) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < j.count) {
        int k = s0 + sg;
        for (; k + 12 < s1; k += 16) {
            a0 += slab[(long)k * j.stride + i]; a1 += slab[(long)(k + 4) * j.stride + i];
            a2 += slab[(long)(k + 8) * j.stride + i]; a3 += slab[(long)(k + 12) * j.stride + i];
        }
        for (; k < s1; k += 4) a0 += slab[(long)k * j.stride + i];
    }
    red[sg][cx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sg == 0 && i < j.count) {
        const float v = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
        if (j.ny == 1) j.dst[i] += v; else atomicAdd(j.dst + i, v);
    }
}
__global__ __launch_bounds__(256) void k_slab_reduce(const SlabJob j0, const SlabJob j1, const SlabJob j2, const SlabJob j3) {
    __shared__ __attribute__((aligned(16))) float red[4][256];
    slab_reduce_body(blockIdx.z == 0 ? j0 : blockIdx.z == 1 ? j1 : blockIdx.z == 2 ? j2 : j3, red);
}
// The same launch with one more z-plane that runs a BatchNorm backward link (round 5): after the fused 1x1 backward kernel both its slab
// reduction and the norm1 link are ~5 us latency-floor launches on the critical chain, independent of each other (different inputs, different
// outputs) -- one launch instead of two per dense layer.  Plane `nz` = the link: its first cdiv(C, 4) workgroups of y-slice 0.
__global__ __launch_bounds__(256) void k_slab_reduce_link(const SlabJob j0, const SlabJob j1, const SlabJob j2, const SlabJob j3, int nz,
                                                          const BnBwdLinkArgs link) {
    __shared__ __attribute__((aligned(16))) float red[4][256];
    if ((int)blockIdx.z == nz) {
        if (blockIdx.y == 0) bn_bwd_link_body(link, blockIdx.x);
        return;
    }
    slab_reduce_body(blockIdx.z == 0 ? j0 : blockIdx.z == 1 ? j1 : blockIdx.z == 2 ? j2 : j3, red);
}

__global__ void k_unpack(const UnpackDesc* descs) {
    const UnpackDesc d = descs[blockIdx.y];
    const long total = (long)d.N * d.Cin * d.taps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % d.taps);
        const int c = (int)((i / d.taps) % d.Cin);
        const int n = (int)(i / ((long)d.taps * d.Cin));
        d.dst[i] += d.nfast ? d.src[((long)tap * d.Cin + c) * 32 + n] : d.src[(long)n * d.Kp + tap * d.Cin + c];
    }
}

}  // namespace

int bn_bwd_link(const BnBwdLinkArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_bwd_link, dim3(cdiv(a.C, 4)), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int head_pool_bwd_grid(int n_img) { return n_img < 256 ? n_img : 256; }
int head_pool_bwd(const HeadPoolBwdArgs& a, hipStream_t st) {
    if (a.C > 256 * HP_CJ) return -2;
    if (a.nblk != head_pool_bwd_grid(a.n_img)) return -3;
    if (a.mode == MODE_F32) hipLaunchKernelGGL(k_head_pool_bwd<float>, dim3(a.nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_head_pool_bwd<bf16>, dim3(a.nblk), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int pool0_bwd_grid(int n_img, int Hin, int Win) {
    const long g = ((long)n_img * Hin * Win + 3) / 4;
    return (int)(g < 2048 ? g : 2048);
}
int pool0_bwd(const Pool0BwdArgs& a, hipStream_t st) {
    if (a.C > 64 * PB_CJ) return -2;
    if (a.nblk != pool0_bwd_grid(a.n_img, a.Hin, a.Win)) return -3;
    if (a.mode == MODE_F32) hipLaunchKernelGGL(k_pool0_bwd<float>, dim3(a.nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_pool0_bwd<bf16>, dim3(a.nblk), dim3(256), 0, st, a);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int eff_materialize_bf16(const EffMatArgs& a, hipStream_t st) {
    if (a.M <= 0) return 0;
    const EffSrc& e = a.e;
    if (e.N > 512 || (e.ldg & 7) || (e.ldx & 7) || (e.c_off & 7) || (a.ldo & 7) || a.ldo < ((e.N + 7) & ~7)) return -2;
    int wlog = 0;
    while ((1 << wlog) < ((e.N + 7) >> 3)) ++wlog;
    const int rpb = 256 >> wlog;
    const long g = (a.M + rpb - 1) / rpb;
    const int nb = (int)(g < 1024 ? g : 1024);
    if (a.colsum != nullptr && a.slab == nullptr) return -3;
    hipLaunchKernelGGL(k_eff_mat, dim3(nb), dim3(256), 0, st, a, wlog);
    TCVN_LAUNCH_CHECK();
    if (a.colsum != nullptr) {
        if (a.deferred != nullptr) *a.deferred = slab_job(a.slab, nb, e.N, a.colsum, 0);      // folded into the caller's next reduction
        else return slab_reduce(a.slab, nb, e.N, a.colsum, st);
    }
    return 0;
}

SlabJob slab_job(const float* slab, int nslab, long count, float* dst, long stride) {
    SlabJob j{slab, dst, nslab, count, stride > 0 ? stride : count, 1, 1, 0};
    if (count <= 0 || nslab <= 0) { j.count = 0; return j; }
    j.v4 = (count % 4 == 0 && j.stride % 4 == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && count >= 1024) ? 1 : 0;
    static const int per_knob = TCVN_KNOB_INT("TCVN_SLAB_PER");     // validation build: slabs per y-slice of the 16-B path (A/B)
    int ny = cdiv(nslab, j.v4 ? (per_knob > 0 ? per_knob : 512) : 64);           // 16-B path: one y-slice up to 512 slabs (no atomics; see slab_reduce_body)
    if (ny > 16) ny = 16;
    j.ny = ny; j.per_y = cdiv(nslab, ny);
    return j;
}
// up to four independent jobs in one launch (grid.z); empty jobs (count == 0) are skipped
int slab_reduce4(const SlabJob* jobs, int n, hipStream_t st) {
    SlabJob j[4] = {};
    long gmax = 0; int nymax = 1, nz = 0;
    for (int i = 0; i < n && i < 4; ++i) {
        if (jobs[i].count <= 0) continue;
        j[nz++] = jobs[i];
        const long g = cdiv(jobs[i].count, jobs[i].v4 ? 128 : 64);
        gmax = g > gmax ? g : gmax; nymax = jobs[i].ny > nymax ? jobs[i].ny : nymax;
    }
    if (nz == 0) return 0;
    hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)gmax, nymax, nz), dim3(256), 0, st, j[0], j[1], j[2], j[3]);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int slab_reduce4_link(const SlabJob* jobs, int n, const BnBwdLinkArgs& link, hipStream_t st) {
    SlabJob j[4] = {};
    long gmax = cdiv(link.C, 4); int nymax = 1, nz = 0;
    for (int i = 0; i < n && i < 4; ++i) {
        if (jobs[i].count <= 0) continue;
        j[nz++] = jobs[i];
        const long g = cdiv(jobs[i].count, jobs[i].v4 ? 128 : 64);
        gmax = g > gmax ? g : gmax; nymax = jobs[i].ny > nymax ? jobs[i].ny : nymax;
    }
    hipLaunchKernelGGL(k_slab_reduce_link, dim3((unsigned)gmax, nymax, nz + 1), dim3(256), 0, st, j[0], j[1], j[2], j[3], nz, link);
    TCVN_LAUNCH_CHECK();
    return 0;
}
int slab_reduce2(const SlabJob& a, const SlabJob& b, hipStream_t st) {
    const SlabJob jobs[2] = {a, b};
    return slab_reduce4(jobs, 2, st);
}
int slab_reduce(const float* slab, int nslab, long count, float* dst, hipStream_t st, long stride) {
    SlabJob none{};
    return slab_reduce2(slab_job(slab, nslab, count, dst, stride), none, st);
}

int unpack_wgrads(const UnpackDesc* d_descs, int n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_unpack, dim3(32, n), dim3(256), 0, st, d_descs);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn
