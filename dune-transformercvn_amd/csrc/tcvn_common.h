// Common device/host helpers for the TransformerCVN gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

// Validation / ablation switches exist only in builds made with -DTCVN_DEBUG_KNOBS (`make debug` -> libtcvn_hip_dbg.so, used by
// the tests that compare kernel variants).  The default library reads no environment variable and contains none of the
// work-dropping branches: TCVN_KNOB_* fold to constants and the compiler removes the code behind them.
#ifdef TCVN_DEBUG_KNOBS
#define TCVN_KNOB_INT(name) (getenv(name) ? atoi(getenv(name)) : 0)
#define TCVN_KNOB_SET(name) (getenv(name) != nullptr)
#define TCVN_DBG_BIT(v, bit) (((v) & (bit)) != 0)
#else
#define TCVN_KNOB_INT(name) 0
#define TCVN_KNOB_SET(name) false
#define TCVN_DBG_BIT(v, bit) false
#endif

namespace tcvn {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

typedef unsigned short bf16;   // storage type for bf16 activations / weights

enum { MODE_F32 = 0, MODE_BF16 = 1 };

__device__ __forceinline__ float bf2f(bf16 v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16 f2bf(float f) {          // round-to-nearest-even (v_cvt_pk_bf16_f32), NaN stays NaN
    const __bf16 b = static_cast<__bf16>(f);
    return __builtin_bit_cast(unsigned short, b);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return f2bf(v); }
// value as the consumer will see it after storage rounding
template <typename T> __device__ __forceinline__ float round_to(float v) { return to_f<T>(from_f<T>(v)); }

__device__ __forceinline__ float prelu(float u, float a) { return u > 0.f ? u : a * u; }
// The effective gradient g + P*x + Q and the gradient accumulation g + s*d with the FMA spelled out: left as `a + b*c` the compiler
// contracts or not depending on the surrounding code (an unrelated edit of the fused 1x1 backward epilogue un-fused four of its eight
// G updates: kernels that must agree bit for bit -- the fused backward and the three kernels it replaces -- then differed by an fp32 ulp
// in front of a bf16 rounding, 4e-3 in the stem's gradients).
__device__ __forceinline__ float eff3(float g, float P, float x, float Q) { return fmaf(P, x, g) + Q; }

// XOR swizzle of the sixteen 16-B chunks of a 256-B LDS row that is read BOTH ways: slot = chunk ^ swz16(row).  Row-wise b128 fragment reads
// (16 consecutive rows, one chunk) need a bijection of row & 15 onto the slots; the transposed reads (ds_read_b64_tr_b16: half a wave = four
// consecutive rows x one aligned 64-B quad of chunks) need the QUAD index (slot >> 2) to differ between four consecutive rows, whatever
// their alignment -- with the plain `row & 15` the quad moves with (row >> 2), so the four rows of a transpose group fall on the same 16 banks
// (rocprofv3 round 4: LDS_BANK_CONFLICT = 2.6 x the LDS-active cycles of the 3x3 weight-gradient kernel).  Here row & 3 picks the quad.
__device__ __forceinline__ int swz16(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// load 8 consecutive elements as float (vector path requires 16B/32B alignment)
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float v[8]) {
    const u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf2f(a[j]);
}
template <typename T> __device__ __forceinline__ void load8_guard(const T* p, int n, float v[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = j < n ? to_f<T>(p[j]) : 0.f;
}

// counter-based RNG (Philox-4x32-10) for dropout masks and pixel noise: reproducible in backward
__device__ __forceinline__ uint4 philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// one uniform in [0,1) per (stream id, element index)
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t stream, uint64_t idx) {
    const uint4 r = philox4((uint32_t)(idx >> 2), (uint32_t)(idx >> 34), stream, 0x5443564eu, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
    return (w >> 8) * (1.0f / 16777216.0f);
}
// dropout keep-scale for element idx: 0 or 1/(1-p)
__device__ __forceinline__ float drop_scale(float p, uint64_t seed, uint32_t stream, uint64_t idx) {
    if (p <= 0.f) return 1.f;
    return rng_uniform(seed, stream, idx) >= p ? 1.f / (1.f - p) : 0.f;
}
// The same draws for callers that walk consecutive element indices: elements 4q .. 4q+3 are the four words of ONE Philox call (counter q), so
// the call is repeated only when idx >> 2 changes (a Philox-4x32-10 is 40 quarter-rate integer multiplies; the fused encoder spent ~20 % of
// its time recomputing the same block four times).  `cur` / `r` are the caller's cache (cur = ~0 initially).
struct DropCache { uint64_t cur; uint4 r; };
__device__ __forceinline__ float drop_scale_cached(DropCache& dc, float p, uint64_t seed, uint32_t stream, uint64_t idx) {
    if (p <= 0.f) return 1.f;
    const uint64_t q = idx >> 2;
    if (q != dc.cur) {
        dc.r = philox4((uint32_t)q, (uint32_t)(q >> 32), stream, 0x5443564eu, (uint32_t)seed, (uint32_t)(seed >> 32));
        dc.cur = q;
    }
    const uint32_t w = (idx & 3) == 0 ? dc.r.x : (idx & 3) == 1 ? dc.r.y : (idx & 3) == 2 ? dc.r.z : dc.r.w;
    return (w >> 8) * (1.0f / 16777216.0f) >= p ? 1.f / (1.f - p) : 0.f;
}

// Dropout mask of a [pixels, N] activation slice.  Counter-based and stateless (recomputed in backward): one 32-bit
// avalanche hash (lowbias32) per (pixel pair, channel) yields two 16-bit uniform draws; pixel & 1 selects the draw, so
// kernels that own consecutive pixels of one channel share a hash.  keep <=> draw >= round(p * 65536).
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_key(uint64_t seed, uint32_t stream) {
    return lowbias32((uint32_t)seed ^ lowbias32((uint32_t)(seed >> 32) + stream * 0x9E3779B9u));
}
__device__ __forceinline__ uint32_t drop_bits(uint32_t key, long m, int n, int N) {
    const uint64_t grp = (uint64_t)(m >> 1) * (uint64_t)N + (uint64_t)n;
    return lowbias32(((uint32_t)grp * 0x9E3779B1u) ^ key ^ ((uint32_t)(grp >> 32) * 0x85ebca77u));
}
// the same hash for slices whose group index (pixels/2 * N + n) fits 32 bits (every realistic launch: the launchers check pixels * N < 2^32):
// the high word of the group index is zero, so its term drops out and the 64-bit multiply-add becomes a 32-bit one
__device__ __forceinline__ uint32_t drop_bits32(uint32_t key, int m, int n, int N) {
    const uint32_t grp = (uint32_t)(m >> 1) * (uint32_t)N + (uint32_t)n;
    return lowbias32((grp * 0x9E3779B1u) ^ key);
}
__device__ __forceinline__ float drop_pick(uint32_t bits, long m, float p) {
    const uint32_t draw = (m & 1) ? (bits >> 16) : (bits & 0xffffu);
    return draw >= (uint32_t)(p * 65536.0f + 0.5f) ? 1.f / (1.f - p) : 0.f;
}
__device__ __forceinline__ float drop_scale_mn(float p, uint64_t seed, uint32_t stream, long m, int n, int N) {
    if (p <= 0.f) return 1.f;
    return drop_pick(drop_bits(drop_key(seed, stream), m, n, N), m, p);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

#define TCVN_CHECK(expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "tcvn: %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, hipGetErrorString(e_)); \
            return (int)e_;                                                                \
        }                                                                                  \
    } while (0)

#define TCVN_LAUNCH_CHECK()                                                                \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "tcvn: launch failed at %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return (int)e_;                                                                \
        }                                                                                  \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

}  // namespace tcvn
