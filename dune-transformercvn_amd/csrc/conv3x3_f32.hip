// fp32 (parity mode) 3x3 convolutions of the DenseNet bottleneck layers on padded LDS tiles: forward, data gradient and weight
// gradient of conv2 (128 -> growth <= 32 channels; reference: transformercvn/network/layers/dense_net.py:29-40 and its autograd).
//
// Same padded index space as the bf16 tile kernels (tile3x3.h): a workgroup stages ONE transformed image of 128 + 2*(W+3)
// consecutive padded positions in LDS, so BatchNorm + PReLU run once per element instead of once per tap and the nine taps are
// plain row offsets.  Arithmetic is v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 64 cycles per instruction): one operand float
// per lane, so fragments are read from LDS as 4-byte words along the contiguous channel axis -- no transposes, no swizzles:
//   weight gradient  C[n][c] (per tap) = sum_pos eff[pos][n] * img[pos + shift(tap)][c]:  36 (tap, 32-channel) tiles, nine per wave
//                    (144 accumulator registers), A = eff^T and B = img both read along their channel axis;
//   forward          each wave owns a quarter of K (32 input channels x 9 taps, its 144 weight floats register resident) for all
//                    128 positions, the four partial tiles are summed through LDS in the epilogue;
//   data gradient    each wave owns 32 of the 128 output channels (144 weight floats register resident) for all 128 positions.
// 576 MFMAs per wave and tile (36.9 k cycles, 22 us at the 1.64 GHz the chip holds under this load); the measured MFMA phases run at
// that rate, staging and epilogue add 40-60 % (one workgroup per CU: nothing overlaps them), see DESIGN.md section 4.
#include "prof.h"
#include "tcvn_ops.h"
#include "tile3x3.h"

namespace tcvn {

using namespace t3;

namespace {

constexpr int C128 = 128;

#ifdef TCVN_PHASE_PROF
__device__ unsigned long long g_ph[32];
#define PH_INIT long long ph_last = clock64();
#define PH(i) if (threadIdx.x == 0) { const long long ph_now = clock64(); atomicAdd(&g_ph[i], (unsigned long long)(ph_now - ph_last)); ph_last = ph_now; }
#else
#define PH_INIT
#define PH(i)
#endif

__device__ __forceinline__ int fdiv_(int a, int d, float inv, int& rem) {
    int q = (int)((float)a * inv);
    rem = a - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
}
__device__ __forceinline__ int pix_of_(const PadGeom& q, int g, float invWp, float invHp) {
    if (g < 0 || g >= (int)q.gtot) return -1;
    int wp, hp;
    const int row = fdiv_(g, q.Wp, invWp, wp);
    const int img = fdiv_(row, q.Hp, invHp, hp);
    if (hp < 1 || hp > q.H || wp < 1 || wp > q.W) return -1;
    return (img * q.H + (hp - 1)) * q.W + (wp - 1);
}

// image rows [0, nrows): prelu(sc*y + sh, sl) of Y[tbl[row]] (zeros for padding), row stride IS floats.  Loads are issued in
// batches of SB rows per thread before any of them is consumed: one workgroup per CU cannot hide a dependent load per row.
template <int IS, int SB>
__device__ __forceinline__ void stage_image(float* img, const int* tbl, int nrows, const float* __restrict__ Y, long lda,
                                            const float* sc, const float* sh, const float* sl) {
    const int ch = threadIdx.x & 31, r0 = threadIdx.x >> 5;                 // 16-B chunk, row lane
    const f32x4 vsc = *reinterpret_cast<const f32x4*>(sc + ch * 4), vsh = *reinterpret_cast<const f32x4*>(sh + ch * 4),
                vsl = *reinterpret_cast<const f32x4*>(sl + ch * 4);
    for (int rb = r0; rb < nrows; rb += 8 * SB) {
        f32x4 y[SB];
        int mm[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) mm[u] = rb + 8 * u < nrows ? tbl[rb + 8 * u] : -1;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            y[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (mm[u] >= 0) y[u] = *reinterpret_cast<const f32x4*>(Y + (long)mm[u] * lda + ch * 4);
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int rr = rb + 8 * u;
            if (rr < nrows) {
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (mm[u] >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = prelu(fmaf(y[u][j], vsc[j], vsh[j]), vsl[j]);
                }
                float* d = img + rr * IS + ch * 4;
                if (IS % 4 == 0) *reinterpret_cast<f32x4*>(d) = v;
                else { d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
            }
        }
    }
}

// eff[pos][n] = drop * (G + P*x + Q) of the layer's output slice for nrows rows starting at row `first` of tbl; row stride ES
template <int ES, int SB>
__device__ __forceinline__ void stage_eff(float* eff, const int* tbl, int first, int nrows, const EffSrc& e, float* bsum) {
    const float* __restrict__ G = reinterpret_cast<const float*>(e.G);
    const float* __restrict__ X = reinterpret_cast<const float*>(e.X);
    const int n = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const bool nok = n < e.N;
    const float P = nok ? e.P[n] : 0.f, Q = nok ? e.Q[n] : 0.f;
    float s = 0.f;
    for (int rb = r0; rb < nrows; rb += 8 * SB) {
        float gv[SB], xv[SB];
        int mm[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) mm[u] = (rb + 8 * u < nrows && nok) ? tbl[first + rb + 8 * u] : -1;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            gv[u] = 0.f; xv[u] = 0.f;
            if (mm[u] >= 0) { gv[u] = G[(long)mm[u] * e.ldg + e.c_off + n]; xv[u] = X[(long)mm[u] * e.ldx + e.c_off + n]; }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int rr = rb + 8 * u;
            if (rr < nrows) {
                float v = 0.f;
                if (mm[u] >= 0) {
                    v = gv[u] + P * xv[u] + Q;
                    if (e.drop_p > 0.f) v *= drop_scale_mn(e.drop_p, e.seed, e.stream_id, mm[u], n, e.N);
                }
                eff[rr * ES + n] = v;
                s += v;
            }
        }
    }
    if (bsum) *bsum += s;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void k_conv3x3_wgrad_f32(const ConvWgradArgs g, int n_img, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const ConvFwdArgs& fa = g.fa;
    const PadGeom q(n_img, fa.H, fa.W);
    const int nrows = q.rows();
    float* img = smf;                                     // [nrows][128]
    float* eff = img + nrows * C128;                      // [TP][32]
    int* tbl = reinterpret_cast<int*>(eff + TP * 32);     // [nrows]
    float* bred = reinterpret_cast<float*>(tbl + ((nrows + 3) & ~3));      // [8][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const float* __restrict__ Y = reinterpret_cast<const float*>(fa.A);
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float bsum = 0.f;
    // this wave's nine (tap, 32-channel tile) pairs: pair = wave*9 + i -> tap = pair >> 2, ct = pair & 3
    int boff[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int pair = wave * 9 + i, tap = pair >> 2, ct = pair & 3;
        boff[i] = (q.halo + (tap / 3 - 1) * q.Wp + (tap % 3 - 1)) * C128 + ct * 32 + l31;
    }
    PH_INIT
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();
        PH(0)
        for (int rr = tid; rr < nrows; rr += 256) tbl[rr] = pix_of_(q, t * TP - q.halo + rr, invWp, invHp);
        __syncthreads();
        PH(1)
        stage_image<C128, 17>(img, tbl, nrows, Y, fa.lda, fa.sc, fa.sh, fa.sl);
        stage_eff<32, 16>(eff, tbl, q.halo, TP, g.e, &bsum);
        __syncthreads();
        PH(2)
#pragma unroll 2
        for (int ks = 0; ks < TP / 2; ++ks) {
            const int pos = 2 * ks + lh;
            const float a = eff[pos * 32 + l31];
            const float* bp = img + pos * C128;
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[boff[i]], acc[i], 0, 0, 0);
        }
    }
    // dWk[n][tap*128 + c] += acc ; accumulator row = output channel n, column = input channel within the tile
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int pair = wave * 9 + i, tap = pair >> 2, ct = pair & 3;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (n < g.e.N) atomicAdd(g.dWk + (long)n * fa.Kp + tap * C128 + ct * 32 + l31, acc[i][e]);
        }
    }
    if (g.dbias != nullptr) {
        __syncthreads();
        bred[(tid >> 5) * 32 + (tid & 31)] = bsum;
        __syncthreads();
        if (tid < 32 && tid < g.e.N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += bred[r * 32 + tid];
            atomicAdd(g.dbias + tid, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: Out[pos][n_off + n] = drop * (bias[n] + sum_tap sum_c img[pos + shift(tap)][c] * Wk[n][tap*128 + c]), statistics
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ISF = 129;          // image row stride in floats: A fragments run along positions -> odd stride, conflict free

__global__ __launch_bounds__(256, 1) void k_conv3x3_fwd_f32(const ConvFwdArgs g, int n_img, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const PadGeom q(n_img, g.H, g.W);
    const int nrows = q.rows();
    float* img = smf;                                             // [nrows][ISF]; later the 4 partial tiles [4][TP][32]
    int* tbl = reinterpret_cast<int*>(img + nrows * ISF);         // [nrows]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const float* __restrict__ Y = reinterpret_cast<const float*>(g.A);
    const float* __restrict__ Wk = reinterpret_cast<const float*>(g.Wk);
    float* __restrict__ Out = reinterpret_cast<float*>(g.Out);
    // this wave's quarter of K: input channels [32*wave, +32) of every tap; B[k][j = n]: lane holds Wk[n = l31][tap*128 + 32*wave + 2*kk + lh]
    float bw[144];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            bw[tap * 16 + kk] = l31 < g.N ? Wk[(long)l31 * g.Kp + tap * C128 + wave * 32 + 2 * kk + lh] : 0.f;
    // epilogue role: position pos = tid >> 1, channels [16*(tid&1), +16)
    const int epos = tid >> 1, en0 = (tid & 1) * 16;
    float ebias[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) ebias[j] = en0 + j < g.N ? g.bias[en0 + j] : 0.f;
    double s1[16], s2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { s1[j] = 0; s2[j] = 0; }
    PH_INIT
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();
        PH(8)
        for (int rr = tid; rr < nrows; rr += 256) tbl[rr] = pix_of_(q, t * TP - q.halo + rr, invWp, invHp);
        __syncthreads();
        PH(9)
        stage_image<ISF, 17>(img, tbl, nrows, Y, g.lda, g.sc, g.sh, g.sl);
        __syncthreads();
        PH(10)
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* ap = img + (q.halo + (tap / 3 - 1) * q.Wp + (tap % 3 - 1) + l31) * ISF + wave * 32 + lh;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[pt * 32 * ISF + 2 * kk], bw[tap * 16 + kk], acc[pt], 0, 0, 0);
        }
        __syncthreads();                                          // every wave is done reading the image
        PH(11)
        float* part = img;                                        // [4][TP][32]
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                part[((wave * TP) + pt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = acc[pt][e];
        __syncthreads();
        const int m = tbl[q.halo + epos];
        if (m >= 0) {
            float v[16];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                float4 sum = *reinterpret_cast<const float4*>(part + epos * 32 + en0 + j4 * 4);
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(part + (w * TP + epos) * 32 + en0 + j4 * 4);
                    sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
                }
                v[j4 * 4] = sum.x; v[j4 * 4 + 1] = sum.y; v[j4 * 4 + 2] = sum.z; v[j4 * 4 + 3] = sum.w;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = en0 + j;
                float o = v[j] + ebias[j];
                if (g.drop_p > 0.f && n < g.N) o *= drop_scale_mn(g.drop_p, g.seed, g.stream_id, m, n, g.N);
                v[j] = o;
                if (n < g.N) { s1[j] += (double)o; s2[j] += (double)o * o; }
            }
            float* op = Out + (long)m * g.ldo + g.n_off + en0;
            if (g.N == 32) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) *reinterpret_cast<f32x4*>(op + j4 * 4) = f32x4{v[j4 * 4], v[j4 * 4 + 1], v[j4 * 4 + 2], v[j4 * 4 + 3]};
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (en0 + j < g.N) op[j] = v[j];
            }
        }
    }
    if (g.part != nullptr) {                                      // per-channel sums of this workgroup: reduce the 128 threads of each channel half
        __syncthreads();
        double* red = reinterpret_cast<double*>(smf);             // [256][2] per channel slot j, processed 4 channels at a time
        for (int j0 = 0; j0 < 16; j0 += 1) {
            red[tid * 2] = s1[j0]; red[tid * 2 + 1] = s2[j0];
            __syncthreads();
            if (tid < 2) {                                        // tid = channel half
                double a = 0, b = 0;
                for (int r = tid; r < 256; r += 2) { a += red[r * 2]; b += red[r * 2 + 1]; }
                const int n = tid * 16 + j0;
                if (n < g.N) { double* p = g.part + ((long)blockIdx.x * g.N + n) * 2; p[0] = a; p[1] = b; }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// data gradient (N = 32 output-gradient channels): dA[pos][c] = sum_tap sum_n eff[pos - shift(tap)][n] * Wt[c][tap*32 + n], then the
// PReLU + BatchNorm backward of norm2 on Y: u = sc*y + sh ; dU = dA * prelu'(u) ; Gout = sc*dU ; sums (dU, dU*y, dA*min(u,0))
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ESD = 33;

__global__ __launch_bounds__(256, 1) void k_conv3x3_dgrad_f32(const ConvDgradArgs g, int n_img, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const PadGeom q(n_img, g.H, g.W);
    const int nrows = q.rows();
    float* eff = smf;                                             // [nrows][ESD]
    int* tbl = reinterpret_cast<int*>(eff + nrows * ESD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float invWp = 1.0f / q.Wp, invHp = 1.0f / q.Hp;
    const float* __restrict__ Wt = reinterpret_cast<const float*>(g.Wt);
    const float* __restrict__ Yin = reinterpret_cast<const float*>(g.Xin);
    float* __restrict__ Gout = reinterpret_cast<float*>(g.Gout);
    const int c = wave * 32 + l31;                                // this lane's output channel
    float bw[144];                                                // B[k = tap*32 + n][j = c]: Wt[c][2*kk + lh]
#pragma unroll
    for (int kk = 0; kk < 144; ++kk) bw[kk] = Wt[(long)c * g.Kp + 2 * kk + lh];
    const float sc = g.sc[c], sh = g.sh[c], sl = g.sl[c];
    double s1 = 0, s2 = 0, s3 = 0;
    PH_INIT
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();
        PH(16)
        for (int rr = tid; rr < nrows; rr += 256) tbl[rr] = pix_of_(q, t * TP - q.halo + rr, invWp, invHp);
        __syncthreads();
        PH(17)
        stage_eff<ESD, 17>(eff, tbl, 0, nrows, g.e, nullptr);
        __syncthreads();
        PH(18)
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* ap = eff + (q.halo - ((tap / 3 - 1) * q.Wp + (tap % 3 - 1)) + l31) * ESD + lh;     // source = pos - shift(tap)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[pt * 32 * ESD + 2 * kk], bw[tap * 16 + kk], acc[pt], 0, 0, 0);
        }
        PH(19)
        // epilogue in batches of 16 rows: all loads of a batch are in flight before the first is used
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            int mm[16];
            float yv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) mm[e] = tbl[q.halo + pt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh];
#pragma unroll
            for (int e = 0; e < 16; ++e) yv[e] = Yin[(long)(mm[e] < 0 ? 0 : mm[e]) * g.ldxin + c];
            float f1 = 0.f, f3 = 0.f;
            double d2 = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool ok = mm[e] >= 0;
                const float y = yv[e], u = fmaf(y, sc, sh);
                const float dA = ok ? acc[pt][e] : 0.f;
                const float du = u > 0.f ? dA : sl * dA;
                f1 += du; d2 += (double)du * y; f3 += u > 0.f ? 0.f : dA * u;
                if (ok) Gout[(long)mm[e] * g.ldgo + c] = sc * du;
            }
            s1 += f1; s2 += d2; s3 += f3;
        }
    }
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
    if (lh == 0) {
        double* p = g.part + ((long)blockIdx.x * g.N + c) * 3;
        p[0] = s1; p[1] = s2; p[2] = s3;
    }
}

size_t fwd_f32_smem(const PadGeom& q) {
    const size_t img = (size_t)q.rows() * ISF, part = 4 * TP * 32;
    return ((img > part ? img : part) + ((q.rows() + 3) & ~3)) * 4 + 64;
}
size_t dgrad_f32_smem(const PadGeom& q) { return ((size_t)q.rows() * ESD + ((q.rows() + 3) & ~3)) * 4 + 64; }

size_t wgrad_f32_smem(const PadGeom& q) { return ((size_t)q.rows() * C128 + TP * 32 + ((q.rows() + 3) & ~3) + 8 * 32) * 4; }

int grid_f32(long ntiles) { return (int)(ntiles < 256 ? ntiles : 256); }

}  // namespace

bool conv3x3_wgrad_f32_ok(const ConvWgradArgs& a) {
    const ConvFwdArgs& f = a.fa;
    if (!conv3x3_tile_enabled() || a.mode != MODE_F32 || f.amode != A_3X3 || f.C != C128 || f.lda != C128 || a.e.N > 32 || f.Kp != 9 * C128 || a.nfast) return false;
    if (f.M % (f.H * f.W) != 0 || (reinterpret_cast<uintptr_t>(f.A) & 15)) return false;
    const PadGeom q(f.M / (f.H * f.W), f.H, f.W);
    return q.gtot < (1L << 30) && wgrad_f32_smem(q) <= 160 * 1024;
}

int conv3x3_wgrad_f32(const ConvWgradArgs& a, hipStream_t st) {
    const ConvFwdArgs& f = a.fa;
    const int n_img = f.M / (f.H * f.W);
    const PadGeom q(n_img, f.H, f.W);
    const int ntiles = (int)q.tiles();
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_wgrad_f32), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    ProfScope ps("k_conv3x3_wgrad_f32", 2.0 * f.M * (double)a.e.N * f.K, (double)f.M * 4.0 * (f.C + 2 * a.e.N), st);
    hipLaunchKernelGGL(k_conv3x3_wgrad_f32, dim3(grid_f32(ntiles)), dim3(256), wgrad_f32_smem(q), st, a, n_img, ntiles);
    TCVN_LAUNCH_CHECK();
    return 0;
}

}  // namespace tcvn

namespace tcvn {
bool conv3x3_fwd_f32_ok(const ConvFwdArgs& a) {
    if (!conv3x3_tile_enabled() || a.mode != MODE_F32 || a.amode != A_3X3 || a.C != C128 || a.lda != C128 || a.N > 32 || a.Kp != 9 * C128) return false;
    if (a.M % (a.H * a.W) != 0 || (reinterpret_cast<uintptr_t>(a.A) & 15) || (a.ldo & 3) || (a.n_off & 3) ||
        (reinterpret_cast<uintptr_t>(a.Out) & 15)) return false;
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return q.gtot < (1L << 30) && fwd_f32_smem(q) <= 160 * 1024 && 256 * 2 * 8 <= (long)fwd_f32_smem(q);
}
int conv3x3_fwd_f32_nblk(const ConvFwdArgs& a) { return grid_f32(PadGeom(a.M / (a.H * a.W), a.H, a.W).tiles()); }
int conv3x3_fwd_f32(const ConvFwdArgs& a, hipStream_t st) {
    const int n_img = a.M / (a.H * a.W);
    const PadGeom q(n_img, a.H, a.W);
    const int ntiles = (int)q.tiles();
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_fwd_f32), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    ProfScope ps("k_conv3x3_fwd_f32", 2.0 * a.M * (double)a.N * a.K, (double)a.M * 4.0 * (a.C + a.N), st);
    hipLaunchKernelGGL(k_conv3x3_fwd_f32, dim3(grid_f32(ntiles)), dim3(256), fwd_f32_smem(q), st, a, n_img, ntiles);
    TCVN_LAUNCH_CHECK();
    return 0;
}
bool conv3x3_dgrad_f32_ok(const ConvDgradArgs& a) {
    if (!conv3x3_tile_enabled() || a.mode != MODE_F32 || a.dmode != DG_3X3 || a.N != C128 || a.e.N != 32 || a.Kp != 288 || a.accumulate || a.ldxin != C128) return false;
    if (a.M % (a.H * a.W) != 0) return false;
    const PadGeom q(a.M / (a.H * a.W), a.H, a.W);
    return q.gtot < (1L << 30) && dgrad_f32_smem(q) <= 160 * 1024;
}
int conv3x3_dgrad_f32_nblk(const ConvDgradArgs& a) { return grid_f32(PadGeom(a.M / (a.H * a.W), a.H, a.W).tiles()); }
int conv3x3_dgrad_f32(const ConvDgradArgs& a, hipStream_t st) {
    const int n_img = a.M / (a.H * a.W);
    const PadGeom q(n_img, a.H, a.W);
    const int ntiles = (int)q.tiles();
    static bool attr = false;
    if (!attr) {
        TCVN_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_dgrad_f32), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    ProfScope ps("k_conv3x3_dgrad_f32", 2.0 * a.M * (double)a.N * 9 * a.e.N, (double)a.M * 4.0 * (2 * a.e.N + 2 * a.N), st);
    hipLaunchKernelGGL(k_conv3x3_dgrad_f32, dim3(grid_f32(ntiles)), dim3(256), dgrad_f32_smem(q), st, a, n_img, ntiles);
    TCVN_LAUNCH_CHECK();
    return 0;
}
}  // namespace tcvn

#ifdef TCVN_PHASE_PROF
extern "C" int tcvn_debug_phases(unsigned long long* out32, int reset) {
    hipMemcpyFromSymbol(out32, HIP_SYMBOL(tcvn::g_ph), 32 * 8);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(tcvn::g_ph), z, 32 * 8); }
    return 0;
}
#endif
