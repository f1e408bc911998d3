// fp32 (parity mode) weight gradient of a 1x1 convolution, any channel counts -- the transitions (operand already activated) and the
// bottleneck layers (raw operand, activated on the way into LDS)
// (reference: Transition = BN - PReLU - conv1x1 - AvgPool2, layers/dense_net.py:78-94, and its autograd; the pooled + activated input XP is
// materialised by the forward pass, densenet.hip):
//     dW[n][c] += sum_pos eff[pos][n] * XP[pos][c] ,   db[n] += sum_pos eff[pos][n] ,   eff = G + P * x + Q  (EffSrc)
// Round 5.  The bottleneck tile kernel (k_conv1x1_wgrad_f32: a wave owns 32 output channels over ALL positions, closing fp32 atomics) does not
// cover the transitions' shapes and measured slower than this kernel on its own layers too (9.0 -> 6.8 ms per step over the 68 launches;
// a 256-input-channel tile variant that halves the re-reads of eff spilled and lost: 9.3 ms).  A plain split-K TN GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation): a workgroup owns a 128 x 128 tile of dW and a slice of the positions,
// 2 x 2 waves of 64 x 64 (four accumulator tiles each); both operands of a 32-position chunk are staged in LDS as [position][channel]
// (one float per lane and k-step: lane = channel, the two positions of a k-step on the lane halves), the next chunk travels in registers
// while the current one multiplies; the per-slice tiles leave as slabs laid out like dW and are summed in a fixed order (k_slab_reduce).
#include "tcvn_ops.h"
#include "prof.h"

namespace tcvn {

namespace {

constexpr int PT = 32;                 // positions per staged chunk
constexpr int TLD = 160;               // LDS pitch of a chunk row (floats): the two positions of a k-step fall on disjoint bank halves

__global__ __launch_bounds__(256, 2) void k_gemm_tn_f32(const ConvWgradArgs g, int nsplit, int per_split, float* __restrict__ slab, float* __restrict__ tail) {
    __shared__ __attribute__((aligned(16))) float Es[PT][TLD];      // eff[pos][n0 + 0..127]
    __shared__ __attribute__((aligned(16))) float Bs[PT][TLD];      // XP[pos][c0 + 0..127]
    __shared__ float bred[8][128];
    const ConvFwdArgs& fa = g.fa;
    const EffSrc& e = g.e;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int n0 = blockIdx.x * 128, c0 = blockIdx.y * 128, sp = blockIdx.z;
    const float* __restrict__ G = reinterpret_cast<const float*>(e.G);
    const float* __restrict__ X = reinterpret_cast<const float*>(e.X);
    const float* __restrict__ A = reinterpret_cast<const float*>(fa.A);
    const long m_begin = (long)sp * per_split, m_end = m_begin + per_split < fa.M ? m_begin + per_split : fa.M;

    // staging role: 4 consecutive channels (c4) of positions pr, pr + 8, pr + 16, pr + 24 of a chunk
    const int c4 = (tid & 31) * 4, pr = tid >> 5;
    // A 16-B chunk that STARTS inside the channel range is loaded whole (the row pitches are multiples of 4 floats >= the range: in
    // bounds) and masked per element: channel counts need not be multiples of 4 (transition 4: 226 outputs; dense block 5: cin % 4 == 2)
    const bool n_ok = n0 + c4 < e.N, c_ok = c0 + c4 < fa.K;
    const bool act = fa.sc != nullptr;                              // raw operand: prelu(sc * x + sh, sl) on the way into LDS (bottleneck 1x1)
    f32x4 Pv = {0.f, 0.f, 0.f, 0.f}, Qv = Pv, nm = Pv, scv = Pv, shv = Pv, slv = Pv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (n0 + c4 + q < e.N) { Pv[q] = e.P[n0 + c4 + q]; Qv[q] = e.Q[n0 + c4 + q]; nm[q] = 1.f; }
        if (c0 + c4 + q < fa.K) { scv[q] = act ? fa.sc[c0 + c4 + q] : 1.f; shv[q] = act ? fa.sh[c0 + c4 + q] : 0.f; slv[q] = act ? fa.sl[c0 + c4 + q] : 1.f; }
    }
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 gr[4], xr[4], br[4];
    auto issue = [&](long m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + pr + 8 * i;
            gr[i] = f32x4{0.f, 0.f, 0.f, 0.f}; xr[i] = gr[i]; br[i] = gr[i];
            if (m < m_end) {
                if (n_ok) {
                    gr[i] = *reinterpret_cast<const f32x4*>(G + m * e.ldg + e.c_off + n0 + c4);
                    xr[i] = *reinterpret_cast<const f32x4*>(X + m * e.ldx + e.c_off + n0 + c4);
                }
                if (c_ok) br[i] = *reinterpret_cast<const f32x4*>(A + m * fa.lda + c0 + c4);
            }
        }
    };
    auto commit = [&](long m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + pr + 8 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = nm[q] * (gr[i][q] + Pv[q] * xr[i][q] + Qv[q]);                  // channels beyond N: 0 * finite neighbour data
                    w[q] = prelu(fmaf(br[i][q], scv[q], shv[q]), slv[q]);                  // beyond K: tables are zero; plain operand: identity
                }
            }
            bsum += v;
            *reinterpret_cast<f32x4*>(&Es[pr + 8 * i][c4]) = v;
            *reinterpret_cast<f32x4*>(&Bs[pr + 8 * i][c4]) = w;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    if (m_begin < m_end) issue(m_begin);
    for (long m0 = m_begin; m0 < m_end; m0 += PT) {
        __syncthreads();                                           // the previous chunk has been multiplied
        commit(m0);
        __syncthreads();
        if (m0 + PT < m_end) issue(m0 + PT);
        const float* ep = &Es[lh][wi * 64 + l31];
        const float* bp = &Bs[lh][wj * 64 + l31];
#pragma unroll
        for (int ks = 0; ks < PT / 2; ++ks) {
            const float a0 = ep[2 * ks * TLD], a1 = ep[2 * ks * TLD + 32];
            const float b0 = bp[2 * ks * TLD], b1 = bp[2 * ks * TLD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // the slice's tile -> its slab (laid out like dW: [N][Kp]); accumulator row = output channel, column (lane) = input channel
    float* sl = slab + (long)sp * e.N * fa.Kp;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = c0 + wj * 64 + b * 32 + l31;
            if (c >= fa.Kp) continue;                               // columns [K, Kp) are zeros (padding of the kernel layout; tk * 128 >= Kp)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int n = n0 + wi * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                if (n < e.N) sl[(long)n * fa.Kp + c] = acc[a][b][q];
            }
        }
    if (blockIdx.y == 0 && tail != nullptr) {                       // bias gradient: column sums of eff over the slice's positions
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) bred[pr][c4 + q] = bsum[q];
        __syncthreads();
        if (tid < 128 && n0 + tid < e.N) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += bred[r][tid];
            tail[(long)sp * e.N + n0 + tid] = s;
        }
    }
}

}  // namespace

bool gemm_tn_f32_ok(const ConvWgradArgs& a) {
    const ConvFwdArgs& f = a.fa;
    if (!conv3x3_tile_enabled() || a.mode != MODE_F32 || f.amode != A_1X1 || a.nfast) return false;
    if ((f.sc == nullptr) != (f.sh == nullptr) || (f.sc == nullptr) != (f.sl == nullptr)) return false;      // all three tables (raw operand) or none
    if (a.e.N < 1 || f.K < 1 || (f.lda & 3) || (a.e.ldg & 3) || (a.e.ldx & 3) || (a.e.c_off & 3) || f.Kp < f.K) return false;
    if (f.lda < ((f.K + 3) & ~3) || a.e.ldg < a.e.c_off + ((a.e.N + 3) & ~3) || a.e.ldx < a.e.c_off + ((a.e.N + 3) & ~3)) return false;      // whole chunks in bounds
    if (a.e.drop_p > 0.f || a.slab == nullptr || cdiv(f.K, 128) * 128 < f.Kp) return false;
    const uintptr_t al = reinterpret_cast<uintptr_t>(f.A) | reinterpret_cast<uintptr_t>(a.e.G) | reinterpret_cast<uintptr_t>(a.e.X) | reinterpret_cast<uintptr_t>(a.slab);
    if (al & 15) return false;
    return (long)(a.e.N * (long)f.Kp + a.e.N) * 4 * 8 <= a.slab_bytes;              // at least eight position slices must fit
}

int gemm_tn_f32(const ConvWgradArgs& a, hipStream_t st) {
    const ConvFwdArgs& f = a.fa;
    const int tn = cdiv(a.e.N, 128), tk = cdiv(f.K, 128);
    const long per_slab = (long)a.e.N * f.Kp + a.e.N;                               // weights + bias column sums, floats
    int nsplit = 512 / (tn * tk);                                                   // two workgroups per CU
    const long fit = a.slab_bytes / (per_slab * 4);
    if (nsplit > fit) nsplit = (int)fit;
    const long chunks = cdiv(f.M, PT);
    if (nsplit > chunks) nsplit = (int)chunks;
    if (nsplit < 1) nsplit = 1;
    const int per_split = (int)(cdiv(chunks, nsplit) * PT);
    nsplit = cdiv(f.M, per_split);
    float* tail = a.dbias ? a.slab + (long)nsplit * a.e.N * f.Kp : nullptr;
    {
        ProfScope ps("k_gemm_tn_f32<transition>", 2.0 * f.M * (double)a.e.N * f.K, (double)f.M * 4.0 * (f.K + 2 * a.e.N), st);
        hipLaunchKernelGGL(k_gemm_tn_f32, dim3(tn, tk, nsplit), dim3(256), 0, st, a, nsplit, per_split, a.slab, tail);
        TCVN_LAUNCH_CHECK();
    }
    const SlabJob jw = slab_job(a.slab, nsplit, (long)a.e.N * f.Kp, a.dWk, 0);
    SlabJob jb{};
    if (tail != nullptr) jb = slab_job(tail, nsplit, a.e.N, a.dbias, 0);
    return slab_reduce2(jw, jb, st);
}

}  // namespace tcvn
