// Padded-tile machinery of the bf16 3x3 kernels (forward, dgrad, wgrad).
//
// Pixels are addressed in a zero-padded index space: image n, padded row hp in [0,H+2), padded column wp in [0,W+2);
//   g = (n*(H+2) + hp)*(W+2) + wp ;  real pixel <=> 1<=hp<=H && 1<=wp<=W ;  m = (n*H + hp-1)*W + wp-1.
// In that space a 3x3 tap is the constant shift (ky-1)*(W+2) + (kx-1) and every out-of-image neighbour is an explicit
// zero row, so a workgroup stages ONE transformed (BatchNorm+PReLU applied, bf16) image of 128 + 2*(W+3) consecutive
// padded positions in LDS and all nine taps read it with plain row offsets: the transform runs once per element instead
// of once per tap, and no per-tap validity masks exist.  Useful fraction of the padded space: H*W/((H+2)*(W+2)) (95 % at
// 99x69).  LDS rows are 128 channels = 256 B (or 32 channels = 64 B for gradient images), 16-B chunks XOR-swizzled so
// that the 16 lanes of a ds_read_b128 group (consecutive rows, same chunk) hit 16 different bank slots.
#pragma once
#include "tcvn_ops.h"

namespace tcvn {
namespace t3 {

constexpr int TP = 128;                       // padded positions per tile (4 waves x 32 rows)

struct PadGeom {
    int n, H, W, Hp, Wp, halo;                // halo = Wp + 1 rows on each side
    long gtot;                                // n*Hp*Wp
    __host__ __device__ PadGeom(int n_, int H_, int W_) : n(n_), H(H_), W(W_), Hp(H_ + 2), Wp(W_ + 2), halo(W_ + 3),
                                                           gtot((long)n_ * (H_ + 2) * (W_ + 2)) {}
    __host__ __device__ int rows() const { return TP + 2 * halo; }
    __host__ __device__ long tiles() const { return (gtot + TP - 1) / TP; }
};

struct Pos { int img, hp, wp; };

__device__ __forceinline__ Pos decode(const PadGeom& q, long g) {
    Pos p;
    const long row = g / q.Wp;
    p.wp = (int)(g - row * q.Wp);
    p.img = (int)(row / q.Hp);
    p.hp = (int)(row - (long)p.img * q.Hp);
    return p;
}
__device__ __forceinline__ void advance(const PadGeom& q, Pos& p, int d) {    // d >= 0, any size
    p.wp += d;
    while (p.wp >= q.Wp) { p.wp -= q.Wp; ++p.hp; }
    while (p.hp >= q.Hp) { p.hp -= q.Hp; ++p.img; }
}
// pixel index of a padded position or -1
__device__ __forceinline__ long pixel(const PadGeom& q, const Pos& p, long g) {
    if (g < 0 || g >= q.gtot || p.hp < 1 || p.hp > q.H || p.wp < 1 || p.wp > q.W) return -1;
    return ((long)p.img * q.H + (p.hp - 1)) * q.W + (p.wp - 1);
}

// byte offset of 16-B chunk `c` of LDS row `r`: 256-B rows (128 ch) and 64-B rows (32 ch)
__device__ __forceinline__ int off256(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }
__device__ __forceinline__ int off64(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ u16x8 pack8(const float v[8]) {
    u16x8 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = f2bf(v[j]);
    return p;
}

}  // namespace t3
}  // namespace tcvn
