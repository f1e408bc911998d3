// DenseNet plan: slot table, geometry, workspace layout, forward/backward drivers.
#pragma once
#include <string>
#include <vector>

#include "../../include/tcvn_hip.h"
#include "tcvn_common.h"

namespace tcvn {

struct Bwd1x1Args;
struct Slot { std::string name; long numel; int kind; };
struct BnSlots { int w = -1, b = -1, rm = -1, rv = -1, nbt = -1, C = 0, id = -1; };
struct LayerSlots { BnSlots n1, n2; int a1, w1, b1, a2, w2, b2, cin; };
struct BlockGeom {
    int H, W, C0, L, Ctot, ld;        // ld: channel pitch of the concat / gradient buffers (bf16: a multiple of 64 = whole 128-B lines per row)
    int ldp;                          // channel pitch of the pooled transition input XP (Ctot rounded up to 8: also its GEMM's K extent)
    std::vector<LayerSlots> layers;
    bool has_trans = false;
    BnSlots tn; int ta = -1, tw = -1, tb = -1;
};
struct WkEntry { int slot, N, Cin, taps, transpose, rows, Kp, frag; long off; };

struct Layout {
    long img, c0, bstat0, part, tabs, F, Z, head_stat, wk, fwd_end, total;
    std::vector<long> D, bstatD;
    std::vector<std::vector<long>> Y, bstatY, YA;     // YA: activated bf16 copies of Y (bf16 mode)
    std::vector<std::vector<long>> EY3;               // bf16 mode: [pixels][32] eff rows of a layer's 3x3 output gradient (data gradient -> weight gradient)
    std::vector<std::vector<long>> KM;                // bf16 mode with dropout: one keep word per pixel and dense layer (3x3 output dropout)
    long zeros, ey, ey2, slab;
    long isum_bytes = 0;                 // link-free BatchNorm statistics (bn_lf.h): fixed-point accumulators, directly behind the zero page
    std::vector<long> isumD;             //   per dense block: [ld][2] int64 (sum, sum of squares) of the concat buffer's channels
    std::vector<std::vector<long>> isumY;    //   per dense layer: [128][2] of the bottleneck map
    long slab1;                          // bf16 mode: slabs of the fused 1x1 backward kernel (main stream; `slab` may be in use by the 3x3 weight gradient
                                         // on the side stream when tcvn_backward_overlap is on)
    long sact;                           // bf16 dense stem: activity bitmap of the conv0 output (stem_mark), -1 when unused
    long sidx;                           // sparse-stem bucket index (stem_sparse.hip), -1 when the plan cannot use it
    std::vector<std::vector<long>> XA;   // activated bf16 copies of the 1x1-conv inputs (per layer, -1 if absent)
    std::vector<long> XP;                // pooled activated transition inputs (-1 if absent)
    // backward
    long du, du0, pq0, pqY, gwk, bpart, dF, dZ, zero_end;
    std::vector<long> G, pqD;
};

bool backward_overlap_enabled();
void set_backward_overlap(int on);
bool xa_materialize();      // true unless TCVN_XA_ONTHEFLY is set: keep the activated copy of every 1x1 input in HBM (A/B switch)

struct DenseNetPlan {
    tcvn_densenet_cfg cfg;
    int esz, Hc, Wc, Cf, n_bn = 0;
    std::vector<Slot> slots;
    std::vector<float*> data, grad;
    std::vector<BlockGeom> blocks;
    int s_w0, s_b0, s_a0, s_af, s_wl, s_al;
    BnSlots n0, nf, nl;
    std::vector<WkEntry> wk_cache;
    bool bound = false;
    bool fast1_ok(int cin) const;    // 1x1 bottleneck conv on the bf16 NT/TN GEMMs
    bool fastt_ok(int Ctot) const;   // transition conv on the bf16 NT/TN GEMMs
    bool fast3x3 = false;            // bf16 padded-tile 3x3 kernels in use (decided at bind time)
    // device descriptor table (pack + eval BN descriptors)
    char* d_desc = nullptr; size_t desc_cap = 0; std::vector<char> h_desc;
    char* desc_ws = nullptr; long desc_total = 0; int n_pack = 0, n_bneval = 0;
    uint64_t last_seed = 0; int last_n = 0;
    const int32_t* last_coords = nullptr; long last_nnz = 0;     // COO list of the last forward (sparse stem weight gradient)
    const float* last_values = nullptr; int last_value_mode = 0; float last_noise = 0.f;
    std::vector<std::vector<char>> keep_valid;   // [block][layer]: the last train-mode forward stored the layer's dropout keep words
    std::vector<std::vector<char>> act_fused;    // [block][layer]: the last forward ran the 3x3 pair kernel on the RAW bottleneck map (activation applied
                                                 // in LDS, ConvFwdArgs::act_fused): no activated copy YA of that layer exists, the weight gradient does the same
    std::vector<std::vector<char>> xa_skipped;   // [block][layer]: the last forward ran the fused 1x1 kernel on the raw concat buffer: no activated copy XA of that
                                                 // layer's input exists, backward must take the fused 1x1 kernel (which rebuilds it from x)
    bool last_fused_ya = false;          // the last forward was an eval pass whose 1x1 GEMMs wrote the activated bottleneck maps only (no raw Y)
    bool last_stem_act = false;          // the last forward's dense stem skipped the conv0-output rows no hit reaches (backward must use the same bitmap)
    bool last_sparse_stem = false;       // the last forward ran the sparse-aware stem: no dense map / conv0 output exists (backward must match)
    bool sparse_stem_possible() const;   // plan-level condition (bf16, 3 -> 64 channels); the hit count decides per call
    // weight-gradient side stream of backward (3x3 and 1x1 weight gradients run beside the data-gradient chain)
    hipStream_t side_st = nullptr; hipEvent_t ev_fork_a = nullptr, ev_fork_b = nullptr, ev_done[2] = {nullptr, nullptr}, ev_drain = nullptr;
    int ensure_side();
    char* d_undesc = nullptr; std::vector<char> h_undesc; char* undesc_ws = nullptr; long undesc_total = 0; int n_unpack = 0;

    explicit DenseNetPlan(const tcvn_densenet_cfg& c);
    ~DenseNetPlan();
    int add_slot(const std::string& name, long numel, int kind);
    BnSlots add_bn(const std::string& p, int c);
    void layout(int n, bool bwd, Layout& L) const;
    void layout_bwd(int n, long start, long maxY, Layout& L) const;
    long tab_floats() const;
    long tab_off(const BnSlots& s) const;
    long wk_bytes() const;
    std::vector<WkEntry> wk_list() const;
    const WkEntry& wk_find(int slot, int transpose, int frag = 0) const;
    const void* wk_frag(const char* ws, const Layout& L, int slot, int transpose) const;   // null when absent
    int bind(void* const* d, void* const* g);
    int upload_descs(char* ws, const Layout& L, hipStream_t st);
    int forward(int n, const int32_t* coords, const float* values, long nnz, int log_pixels, float noise_std, float* out,
                long out_ld, char* ws, long ws_bytes, int train, uint64_t seed, hipStream_t st);
    int backward(int n, const float* d_out, long d_out_ld, char* ws, long ws_bytes, hipStream_t st, int bi_hi = -1, int bi_lo = 0);
    // Arguments of the fused 1x1 backward kernel for block bi, layer l (PY / QY / part left to the caller) -- the ONE place that decides whether a
    // layer's backward can take it: the forward asks before it drops the activated copy XA, the backward fills its launch from the same function
    bool bwd1x1_fill(int bi, int l, long M, char* ws, const Layout& L, Bwd1x1Args& fa) const;
    bool bwd1x1_fusable(int bi, int l, long M, char* ws, const Layout& L) const;
    std::vector<int> unpack_first;   // first unpack descriptor of every block (+ total): partial backward calls unpack their own blocks
    int tap(int n, const char* name, long* off, int* tn, int* th, int* tw, int* tc, int* tld, int* tes) const;
};

}  // namespace tcvn
