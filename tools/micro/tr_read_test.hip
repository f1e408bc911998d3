// Semantics check of ds_read_b64_tr_b16 (cdna_hip_programming.md T10) with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const short* in, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    const int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
    __attribute__((address_space(3))) s16x4* a = (__attribute__((address_space(3))) s16x4*)(lds + (4 * g + q) * 64 + 4 * p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    std::vector<short> h(64 * 64), o(256);
    for (int i = 0; i < 64 * 64; ++i) h[i] = (short)i;
    short *d, *e;
    hipMalloc(&d, h.size() * 2); hipMalloc(&e, 512);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o.data(), e, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int expect = (4 * (l >> 4) + j) * 64 + (l & 15);
            if (o[l * 4 + j] != expect) { if (bad < 8) printf("lane %d elem %d got %d expect %d\n", l, j, o[l * 4 + j], expect); ++bad; }
        }
    printf("tr_read semantics %s (%d mismatches)\n", bad ? "DIFFER" : "as documented", bad);
    return bad != 0;
}
