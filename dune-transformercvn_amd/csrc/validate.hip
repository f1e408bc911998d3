// Validation aid of the C ABI: evaluates the stateless dropout mask the kernels apply, as a tensor.
// The forward and backward kernels never store a mask: every site recomputes keep(seed, stream id, element) from the
// counter-based generators in tcvn_common.h.  tcvn_dropout_keep() exposes those generators so that a test can (a) compare
// them with the zeros of real kernel outputs and (b) hand the masks of a GPU step to the CPU oracle.
#include "../../include/tcvn_hip.h"
#include "tcvn_common.h"

using namespace tcvn;

namespace {
__global__ void k_dropout_keep(int kind, float p, uint64_t seed, uint32_t sid, long rows, int cols, float* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    out[i] = kind == 0 ? drop_scale(p, seed, sid, (uint64_t)i) : drop_scale_mn(p, seed, sid, r, c, cols);
}
}  // namespace

extern "C" int tcvn_dropout_keep(int kind, float p, uint64_t seed, uint32_t stream_id, int64_t rows, int cols, float* out,
                                 void* stream) {
    if (kind < 0 || kind > 1 || rows < 0 || cols <= 0 || !out || p < 0.f || p >= 1.f) return -1;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(k_dropout_keep, dim3(cdiv(rows * cols, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), kind, p,
                       seed, stream_id, (long)rows, cols, out);
    TCVN_LAUNCH_CHECK();
    return 0;
}
