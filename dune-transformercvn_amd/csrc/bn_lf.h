// Link-free BatchNorm statistics (round 5; reference: nn.BatchNorm2d in train mode, layers/dense_net.py:18-40).
//
// Train-mode BatchNorm needs the batch statistics of a tensor before its consumer may normalise it.  Rounds 1-4 bridged that with a
// ~5.5 us launch per BatchNorm (k_bn_link: reduce the producer's per-workgroup partial rows, build the consumer's (scale, shift) table):
// 132 launches per step on each embedder's critical chain.  Here the producer adds its per-workgroup sums to ONE row of 64-bit
// FIXED-POINT accumulators (no-return agent-scope atomics: they execute at the memory side, integer addition is associative, so the
// result does not depend on arrival order -- deterministic, no fence, no last-workgroup tail), and every workgroup of the CONSUMER
// derives the table in its prologue from those 2 x 8 bytes per channel, under the weight-fragment loads it waits for anyway.  Workgroup
// 0 of the consumer also publishes what the backward pass and the module state need: the (scale, shift) table, (mean, biased
// variance), the running statistics.
//
// Fixed point: sum x scaled by 2^24, sum x^2 by 2^16.  Range: |sum x| < 2^39 and sum x^2 < 2^47, i.e. an r.m.s. activation below 8 000
// over 2 M positions (BatchNorm-ed DenseNet maps are O(1)); resolution per added partial 6e-8 / 1.5e-5 absolute against sums of
// 10^3..10^7: the mean / variance carry ~1e-10 relative error, far below the fp32 table they feed.
#pragma once
#include "tcvn_common.h"

namespace tcvn {

constexpr double LF_S1 = 16777216.0, LF_S2 = 65536.0;       // 2^24, 2^16

// what a consumer kernel needs to derive the (scale, shift) table of its input BatchNorm by itself
// Replicas: the adds of one address are applied one after the other at the memory side (measured, round 5: 500-768 workgroups adding to
// the same 256 addresses cost the producing kernel +13 us, more than the link launch they were to replace), so a workgroup adds to
// replica (blockIdx.x % LF_REP) of the row and the consumer sums the LF_REP replicas -- integer sums: still order-independent.
constexpr int LF_REP = 16;

struct LfLink {
    const long long* isum;             // [LF_REP][rep_stride]: replica r holds [n_new][2] fixed-point (sum, sum of squares) of channels
                                       // [c_new0, c_new0 + n_new) at isum + r * rep_stride; nullptr = not link-free
    long rep_stride;                   // in long longs
    int c_new0, n_new;                 // channels outside the window take (mean, var) from bstat (published by an earlier consumer / link)
    double* bstat;                     // [C][2] (mean, biased variance): read outside the window, written inside it by workgroup 0
    double inv_count; long count;      // positions per channel
    const float *gamma, *beta;
    float *running_mean, *running_var; // updated by workgroup 0 (may be null)
    float *sc_out, *sh_out;            // the table in HBM, for the backward kernels (workgroup 0)
    float eps, momentum;
};

#ifdef __HIPCC__
// producer side: this workgroup's per-channel sums -> the accumulators (one call per channel and workgroup)
__device__ __forceinline__ void lf_add(long long* isum, long rep_stride, int c, double s1, double s2) {
    const long long a = __double2ll_rn(s1 * LF_S1), b = __double2ll_rn(s2 * LF_S2);
    long long* p = isum + (long)(blockIdx.x % LF_REP) * rep_stride + 2 * c;
    __hip_atomic_fetch_add(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(p + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the window entry of channel index i (relative to the window): sum over the replicas (16-B loads, all in flight together)
__device__ __forceinline__ void lf_sums(const long long* isum, long rep_stride, int i, long long& a, long long& b) {
    typedef __attribute__((ext_vector_type(2))) long long ll2;
    ll2 v[LF_REP];
#pragma unroll
    for (int r = 0; r < LF_REP; ++r) v[r] = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(isum + (long)r * rep_stride + 2 * i));
    a = 0; b = 0;
#pragma unroll
    for (int r = 0; r < LF_REP; ++r) { a += v[r].x; b += v[r].y; }
}

// consumer side: (scale, shift) of channel c < C; `publish` = this is workgroup 0 (one thread per channel calls this)
__device__ __forceinline__ void lf_table(const LfLink& k, int c, bool publish, float& sc, float& sh) {
    double mean, var;
    const bool fresh = c >= k.c_new0 && c < k.c_new0 + k.n_new;
    if (fresh) {
        long long a, b;
        lf_sums(k.isum, k.rep_stride, c - k.c_new0, a, b);
        mean = (double)a * (k.inv_count / LF_S1);
        var = (double)b * (k.inv_count / LF_S2) - mean * mean;
        if (var < 0) var = 0;
    } else {
        mean = k.bstat[2 * c]; var = k.bstat[2 * c + 1];
    }
    const float r = (float)(1.0 / sqrt(var + (double)k.eps));        // the link kernel's arithmetic, expression by expression (k_bn_link): a layer
                                                                      // gets the same table whichever of the two builds it
    sc = k.gamma[c] * r;
    sh = k.beta[c] - (float)mean * sc;
    if (publish) {
        k.sc_out[c] = sc; k.sh_out[c] = sh;
        if (fresh) { k.bstat[2 * c] = mean; k.bstat[2 * c + 1] = var; }
        if (k.running_mean != nullptr) {
            const double unb = k.count > 1 ? var * (double)k.count / (double)(k.count - 1) : var;
            k.running_mean[c] = (1.f - k.momentum) * k.running_mean[c] + k.momentum * (float)mean;
            k.running_var[c] = (1.f - k.momentum) * k.running_var[c] + k.momentum * (float)unb;
        }
    }
}
#endif

}  // namespace tcvn
