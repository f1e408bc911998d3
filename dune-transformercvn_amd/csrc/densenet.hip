// DenseNet embedder engine: owns the layer schedule of layers/dense_net.py:97-167 (reference) and drives the gfx950
// kernels on one stream.  Data layout in HBM (per call, inside the caller's workspace):
//   img    [n,H,W,in_ch]                dense pixel map, NHWC
//   c0     [n,H/2,W/2,init]             conv0 output (pre-BN)
//   D[b]   [n,Hb,Wb,ld_b]               concat buffer of dense block b; layer l writes channels [C0_b+l*g, +g) in place
//   Y[b,l] [n,Hb,Wb,bn_size*g]          bottleneck (1x1 conv) output, kept for backward
//   bstat  (mean, biased var) per produced channel, fp64; BN (scale, shift) tables per BatchNorm layer, fp32
// Activations are fp32 (TCVN_MODE_F32) or bf16 (TCVN_MODE_BF16); statistics and tables are always fp64/fp32.
#include <string>
#include <vector>
#include <cstring>

#include "../../include/tcvn_hip.h"
#include "tcvn_ops.h"
#include "tcvn_rows.h"
#include <cstdlib>
#include "densenet_plan.h"

using namespace tcvn;

namespace {
constexpr float kEps = 1e-5f, kMom = 0.1f;
}

// ---------------------------------------------------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------------------------------------------------
int DenseNetPlan::add_slot(const std::string& name, long numel, int kind) {
    slots.push_back({name, numel, kind});
    return (int)slots.size() - 1;
}
BnSlots DenseNetPlan::add_bn(const std::string& p, int c) {
    BnSlots s;
    s.w = add_slot(p + ".weight", c, TCVN_SLOT_PARAM);
    s.b = add_slot(p + ".bias", c, TCVN_SLOT_PARAM);
    s.rm = add_slot(p + ".running_mean", c, TCVN_SLOT_BUFFER);
    s.rv = add_slot(p + ".running_var", c, TCVN_SLOT_BUFFER);
    s.nbt = add_slot(p + ".num_batches_tracked", 1, TCVN_SLOT_COUNTER);
    s.C = c;
    s.id = n_bn++;
    return s;
}

DenseNetPlan::DenseNetPlan(const tcvn_densenet_cfg& c) : cfg(c) {
    esz = cfg.mode == MODE_F32 ? 4 : 2;
    const int g = cfg.growth, mid = cfg.bn_size * cfg.growth;
    Hc = (cfg.H + 6 - 7) / 2 + 1; Wc = (cfg.W + 6 - 7) / 2 + 1;
    int h = (Hc - 3) / 2 + 1, w = (Wc - 3) / 2 + 1;
    int ch = cfg.init_ch;
    const std::string f = "features";
    s_w0 = add_slot(f + ".conv0.weight", (long)ch * cfg.in_ch * 49, TCVN_SLOT_PARAM);
    s_b0 = add_slot(f + ".conv0.bias", ch, TCVN_SLOT_PARAM);
    n0 = add_bn(f + ".norm0", ch);
    s_a0 = add_slot(f + ".relu0.weight", ch, TCVN_SLOT_PARAM);
    for (int b = 0; b < cfg.n_blocks; ++b) {
        BlockGeom bg;
        bg.H = h; bg.W = w; bg.C0 = ch; bg.L = cfg.layers[b]; bg.Ctot = ch + bg.L * g; bg.ldp = (int)round_up(bg.Ctot, 8);
        // bf16 rows start on 128-B lines: the kernels read and write channel PREFIXES of these rows (1x1 input, its gradient's
        // read-modify-write); with a 320-B pitch (160 channels) every second prefix straddles one line more than it has to
        bg.ld = cfg.mode == MODE_BF16 ? (int)round_up(bg.Ctot, 64) : bg.ldp;
        for (int l = 0; l < bg.L; ++l) {
            LayerSlots ls;
            const int cin = ch + l * g;
            const std::string p = f + ".dense" + std::to_string(b + 1) + ".layers." + std::to_string(l);
            ls.n1 = add_bn(p + ".bottleneck_block.norm1", cin);
            ls.a1 = add_slot(p + ".bottleneck_block.relu1.weight", cin, TCVN_SLOT_PARAM);
            ls.w1 = add_slot(p + ".bottleneck_block.conv1.weight", (long)mid * cin, TCVN_SLOT_PARAM);
            ls.b1 = add_slot(p + ".bottleneck_block.conv1.bias", mid, TCVN_SLOT_PARAM);
            ls.n2 = add_bn(p + ".output_block.norm2", mid);
            ls.a2 = add_slot(p + ".output_block.relu2.weight", mid, TCVN_SLOT_PARAM);
            ls.w2 = add_slot(p + ".output_block.conv2.weight", (long)g * mid * 9, TCVN_SLOT_PARAM);
            ls.b2 = add_slot(p + ".output_block.conv2.bias", g, TCVN_SLOT_PARAM);
            ls.cin = cin;
            bg.layers.push_back(ls);
        }
        ch = bg.Ctot;
        if (b != cfg.n_blocks - 1) {
            const std::string p = f + ".transition" + std::to_string(b + 1);
            bg.has_trans = true;
            bg.tn = add_bn(p + ".norm", ch);
            bg.ta = add_slot(p + ".relu.weight", ch, TCVN_SLOT_PARAM);
            bg.tw = add_slot(p + ".conv.weight", (long)(ch / 2) * ch, TCVN_SLOT_PARAM);
            bg.tb = add_slot(p + ".conv.bias", ch / 2, TCVN_SLOT_PARAM);
            ch = ch / 2; h = h / 2; w = w / 2;
        }
        blocks.push_back(bg);
    }
    Cf = ch;
    nf = add_bn(f + ".final_norm", ch);
    s_af = add_slot(f + ".final_relu.weight", ch, TCVN_SLOT_PARAM);
    s_wl = add_slot("output_block.linear.weight", (long)cfg.out_dim * ch, TCVN_SLOT_PARAM);
    nl = add_bn("output_block.norm", cfg.out_dim);
    s_al = add_slot("output_block.relu.weight", cfg.out_dim, TCVN_SLOT_PARAM);
    data.assign(slots.size(), nullptr);
    grad.assign(slots.size(), nullptr);
}

DenseNetPlan::~DenseNetPlan() {
    if (d_desc) (void)hipFree(d_desc);
    if (d_undesc) (void)hipFree(d_undesc);
    if (side_st) {
        (void)hipStreamSynchronize(side_st);
        (void)hipStreamDestroy(side_st);
        (void)hipEventDestroy(ev_fork_a); (void)hipEventDestroy(ev_fork_b); (void)hipEventDestroy(ev_done[0]);
        (void)hipEventDestroy(ev_done[1]); (void)hipEventDestroy(ev_drain);
    }
}
int DenseNetPlan::ensure_side() {
    if (side_st) return 0;
    TCVN_CHECK(hipStreamCreateWithFlags(&side_st, hipStreamNonBlocking));
    hipEvent_t* evs[5] = {&ev_fork_a, &ev_fork_b, &ev_done[0], &ev_done[1], &ev_drain};
    for (hipEvent_t* e : evs) TCVN_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// workspace layout
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Bump {
    long off = 0;
    long take(long bytes) { long o = off; off += round_up(bytes, 256); return o; }
};
}  // namespace

void DenseNetPlan::layout(int n, bool bwd, Layout& L) const {
    Bump b;
    const int mid = cfg.bn_size * cfg.growth;
    L.img = b.take((long)n * cfg.H * cfg.W * cfg.in_ch * esz);
    L.c0 = b.take((long)n * Hc * Wc * cfg.init_ch * esz);
    L.D.clear(); L.Y.clear(); L.bstatD.clear(); L.bstatY.clear(); L.YA.clear(); L.XA.clear(); L.XP.clear(); L.KM.clear();
    L.zeros = b.take(1024);
    // link-free statistics accumulators: contiguous with the zero page, so that ONE memset per forward clears both
    L.isumD.clear(); L.isumY.clear(); L.isum_bytes = 0;
    if (cfg.mode == MODE_BF16) {
        const long i0 = b.off;
        for (const auto& bg : blocks) {
            L.isumD.push_back(b.take((long)LF_REP * bg.ld * 16));             // LF_REP replicas of [ld][2]
            std::vector<long> ys;
            for (int l = 0; l < bg.L; ++l) ys.push_back(b.take((long)LF_REP * mid * 16));
            L.isumY.push_back(ys);
        }
        L.isum_bytes = b.off - i0;
    }
    L.sact = (cfg.mode == MODE_BF16 && cfg.init_ch == 64) ? b.take((long)n * Hc * stem_act_words(Wc) * 4) : -1;
    L.sidx = sparse_stem_possible() ? b.take(stem_sparse_index_bytes(n, cfg.H, cfg.W)) : -1;
    long max_part = (long)pool0_grid(n, blocks[0].H, blocks[0].W) * cfg.init_ch * 16;
    max_part = std::max(max_part, 1024L * cfg.init_ch * 16);      // sparse stem passes: <= 1024 workgroups
    max_part = std::max(max_part, 512L * cfg.init_ch * 16);
    long maxY = 0;
    for (const auto& bg : blocks) {
        const long M = (long)n * bg.H * bg.W;
        L.D.push_back(b.take(M * bg.ld * esz));
        std::vector<long> ys, bs;
        for (int l = 0; l < bg.L; ++l) { ys.push_back(b.take(M * mid * esz)); bs.push_back(b.take(mid * 16)); }
        L.Y.push_back(ys); L.bstatY.push_back(bs);
        std::vector<long> yas;
        if (cfg.mode == MODE_BF16)
            for (int l = 0; l < bg.L; ++l) yas.push_back(b.take(M * mid * esz));
        L.YA.push_back(yas);
        std::vector<long> kms;
        if (cfg.mode == MODE_BF16 && cfg.dropout > 0.f)
            for (int l = 0; l < bg.L; ++l) kms.push_back(b.take(M * 4));
        L.KM.push_back(kms);
        std::vector<long> xas;
        for (int l = 0; l < bg.L; ++l) {
            const int cin = bg.C0 + l * cfg.growth;
            // >= 0 marks the bf16 GEMM path; the activated copy itself is skipped with TCVN_XA_ONTHEFLY (then the GEMMs
            // transform the raw concat buffer in LDS)
            xas.push_back(fast1_ok(cin) ? b.take(xa_materialize() ? M * round_up(cin, 8) * esz : 0) : -1);
        }
        L.XA.push_back(xas);
        const bool tfast = bg.has_trans && fastt_ok(bg.Ctot);
        // fp32 mode (round 4): the same materialised operand feeds the generic forward / weight-gradient kernels of the transition
        const bool tmat32 = bg.has_trans && cfg.mode == MODE_F32 && conv3x3_tile_enabled() && (bg.ld & 3) == 0;
        L.XP.push_back((tfast || tmat32) ? b.take((long)n * (bg.H / 2) * (bg.W / 2) * bg.ldp * esz) : -1);   // row stride bg.ldp, zero padded
        L.bstatD.push_back(b.take((long)bg.ld * 16));
        max_part = std::max(max_part, 512L * std::max(mid, bg.Ctot) * 16);   // the conv launchers use <= 512 workgroups ...
        max_part = std::max(max_part, 768L * mid * 16);                      // ... but the fused 1x1 forward up to 768 (fwd1x1_fused_nblk: three per CU)
        maxY = std::max(maxY, M * mid);
    }
    L.bstat0 = b.take((long)cfg.init_ch * 16);
    L.part = b.take(max_part * 2);                     // x2: backward partials carry 3 doubles per channel
    L.tabs = b.take((long)tab_floats() * 4);
    L.F = b.take((long)n * Cf * 4);
    L.Z = b.take((long)n * cfg.out_dim * 4);
    L.head_stat = b.take((long)cfg.out_dim * 8);
    L.wk = b.take(wk_bytes());
    L.fwd_end = b.off;
    if (bwd) layout_bwd(n, b.off, maxY, L);
    else L.total = b.off;
}

long DenseNetPlan::tab_floats() const {
    // (scale, shift) per BN layer, channel count rounded to 8
    long t = 0;
    auto add = [&](const BnSlots& s) { t += 2 * round_up(s.C, 8); };
    add(n0);
    for (const auto& bg : blocks) {
        for (const auto& ls : bg.layers) { add(ls.n1); add(ls.n2); }
        if (bg.has_trans) add(bg.tn);
    }
    add(nf);
    return t;
}

// offsets (in floats) of the table of BN layer `s` inside the tabs region: sc at off, sh at off + round_up(C, 8)
long DenseNetPlan::tab_off(const BnSlots& s) const {
    long t = 0;
    bool found = false;
    auto add = [&](const BnSlots& q) {
        if (found) return;
        if (q.id == s.id) { found = true; return; }
        t += 2 * round_up(q.C, 8);
    };
    add(n0);
    for (const auto& bg : blocks) {
        for (const auto& ls : bg.layers) { add(ls.n1); add(ls.n2); }
        if (bg.has_trans) add(bg.tn);
    }
    add(nf);
    return t;
}

long DenseNetPlan::wk_bytes() const {
    long t = 0;
    for (const auto& w : wk_list()) t += round_up(round_up(w.rows, 32) * w.Kp * esz, 256);
    return t;
}

// bf16 GEMM paths for the 1x1 convolutions: operands are materialised with row strides rounded up to 8 channels (zero
// padded), so any channel count works as long as the K extent fits the NT kernel's register-resident weights (<= 640).
bool DenseNetPlan::fast1_ok(int cin) const {
    const int mid = cfg.bn_size * cfg.growth;
    return cfg.mode == MODE_BF16 && conv3x3_tile_enabled() && mid % 8 == 0 && mid <= 256 && round_up(cin, 32) <= 640;
}
bool DenseNetPlan::sparse_stem_possible() const {
    return cfg.mode == MODE_BF16 && cfg.in_ch >= 1 && cfg.in_ch <= 3 && cfg.init_ch == 64 && !blocks.empty() && (blocks[0].ld & 7) == 0;
}
bool tcvn::xa_materialize() {
    static const bool on = !TCVN_KNOB_SET("TCVN_XA_ONTHEFLY");   // default: write prelu(bn1(x)) to HBM once per layer.  A/B on
    // MI355X (B=32 x 8 prongs): transforming the raw tile in LDS inside the two GEMMs instead saves the copy (2.4 GB, 1.5 ms of
    // k_act_bf16) but costs more than it saves at one or two waves per SIMD: fwd1x1 2.2 -> 4.8 ms, dW1 2.0 -> 3.0 ms, step 29.3 -> 30.7 ms
    return on;
}
bool DenseNetPlan::fastt_ok(int Ctot) const {
    return cfg.mode == MODE_BF16 && conv3x3_tile_enabled() && round_up(Ctot, 32) <= 640 && Ctot / 2 <= 512;
}

// every conv weight in kernel layout; the order defines the offsets in the wk region
std::vector<WkEntry> DenseNetPlan::wk_list() const {
    std::vector<WkEntry> v;
    const int mid = cfg.bn_size * cfg.growth, g = cfg.growth;
    long off = 0;
    auto add = [&](int slot, int N, int Cin, int taps, int transpose, int frag = 0) {
        WkEntry e;
        e.slot = slot; e.N = N; e.Cin = Cin; e.taps = taps; e.transpose = transpose; e.frag = frag;
        e.rows = transpose ? Cin : N;
        e.Kp = (int)round_up((long)taps * (transpose ? N : Cin), 32);
        e.off = off;
        off += round_up(round_up(e.rows, 32) * e.Kp * esz, 256);
        v.push_back(e);
    };
    const bool frag = cfg.mode == MODE_BF16;      // bf16 fast paths read weights in MFMA fragment order
    add(s_w0, cfg.init_ch, cfg.in_ch, 49, 0);
    for (const auto& bg : blocks) {
        for (const auto& ls : bg.layers) {
            add(ls.w1, mid, ls.cin, 1, 0);
            add(ls.w2, g, mid, 9, 0);
            add(ls.w1, mid, ls.cin, 1, 1);     // dgrad layouts
            add(ls.w2, g, mid, 9, 1);
            if (frag) {
                add(ls.w2, g, mid, 9, 0, 1); add(ls.w2, g, mid, 9, 1, 1);
                add(ls.w1, mid, ls.cin, 1, 0, 1); add(ls.w1, mid, ls.cin, 1, 1, 1);
            }
        }
        if (bg.has_trans) {
            add(bg.tw, bg.Ctot / 2, bg.Ctot, 1, 0); add(bg.tw, bg.Ctot / 2, bg.Ctot, 1, 1);
            if (frag) { add(bg.tw, bg.Ctot / 2, bg.Ctot, 1, 0, 1); add(bg.tw, bg.Ctot / 2, bg.Ctot, 1, 1, 1); }
        }
    }
    return v;
}

const void* DenseNetPlan::wk_frag(const char* ws, const Layout& L, int slot, int transpose) const {
    for (const auto& e : wk_cache)
        if (e.slot == slot && e.transpose == transpose && e.frag == 1) return ws + L.wk + e.off;
    return nullptr;
}

const WkEntry& DenseNetPlan::wk_find(int slot, int transpose, int frag) const {
    for (const auto& e : wk_cache)
        if (e.slot == slot && e.transpose == transpose && e.frag == frag) return e;
    fprintf(stderr, "tcvn: wk_find miss\n");
    abort();
}

int DenseNetPlan::bind(void* const* d, void* const* g) {
    for (size_t i = 0; i < slots.size(); ++i) {
        data[i] = reinterpret_cast<float*>(d[i]);
        grad[i] = g ? reinterpret_cast<float*>(g[i]) : nullptr;
        if (slots[i].kind != TCVN_SLOT_COUNTER && data[i] == nullptr) {
            fprintf(stderr, "tcvn: slot %s unbound\n", slots[i].name.c_str());
            return -10;
        }
    }
    wk_cache = wk_list();
    fast3x3 = cfg.mode == MODE_BF16 && cfg.bn_size * cfg.growth == 128 && cfg.growth <= 32 && conv3x3_tile_enabled();
    bound = true;
    desc_ws = nullptr;   // device descriptor tables are rebuilt on the next forward
    undesc_ws = nullptr;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Tab { float* sc; float* sh; };
}

int DenseNetPlan::upload_descs(char* ws, const Layout& L, hipStream_t st) {
    // Pack descriptors (weights -> kernel layout) and eval-mode BN descriptors live in a small device table that
    // depends on the workspace address; rebuilt only when the workspace base or the bindings change.
    if (desc_ws == ws && desc_total == L.total) return 0;
    std::vector<PackDesc> pd;
    for (const auto& e : wk_cache) {
        PackDesc d;
        d.src = data[e.slot]; d.dst = ws + L.wk + e.off; d.N = e.N; d.Cin = e.Cin; d.taps = e.taps; d.Kp = e.Kp;
        d.transpose = e.transpose; d.frag = e.frag;
        pd.push_back(d);
    }
    std::vector<BnEvalDesc> bd;
    float* tabs = reinterpret_cast<float*>(ws + L.tabs);
    auto addbn = [&](const BnSlots& s) {
        BnEvalDesc d;
        d.gamma = data[s.w]; d.beta = data[s.b]; d.rm = data[s.rm]; d.rv = data[s.rv];
        d.sc = tabs + tab_off(s); d.sh = d.sc + round_up(s.C, 8); d.C = s.C;
        bd.push_back(d);
    };
    addbn(n0);
    for (const auto& bg : blocks) {
        for (const auto& ls : bg.layers) { addbn(ls.n1); addbn(ls.n2); }
        if (bg.has_trans) addbn(bg.tn);
    }
    addbn(nf);
    n_pack = (int)pd.size(); n_bneval = (int)bd.size();
    const size_t bytes = pd.size() * sizeof(PackDesc) + bd.size() * sizeof(BnEvalDesc);
    if (bytes > desc_cap) {
        if (d_desc) TCVN_CHECK(hipFree(d_desc));
        TCVN_CHECK(hipMalloc(&d_desc, bytes));
        desc_cap = bytes;
    }
    h_desc.resize(bytes);
    memcpy(h_desc.data(), pd.data(), pd.size() * sizeof(PackDesc));
    memcpy(h_desc.data() + pd.size() * sizeof(PackDesc), bd.data(), bd.size() * sizeof(BnEvalDesc));
    TCVN_CHECK(hipMemcpyAsync(d_desc, h_desc.data(), bytes, hipMemcpyHostToDevice, st));
    desc_ws = ws; desc_total = L.total;
    return 0;
}

int DenseNetPlan::forward(int n, const int32_t* coords, const float* values, long nnz, int log_pixels, float noise_std,
                          float* out, long out_ld, char* ws, long ws_bytes, int train, uint64_t seed, hipStream_t st) {
    if (!bound) return -11;
    if (n <= 0) return 0;
    Layout L;
    layout(n, train != 0, L);
    if (ws_bytes < L.total) { fprintf(stderr, "tcvn: densenet workspace too small (%ld < %ld)\n", ws_bytes, L.total); return -12; }
    int rc;
    if ((rc = upload_descs(ws, L, st))) return rc;
    const int mode = cfg.mode, g = cfg.growth, mid = cfg.bn_size * cfg.growth;
    float* tabs = reinterpret_cast<float*>(ws + L.tabs);
    auto tab = [&](const BnSlots& s) { Tab t; t.sc = tabs + tab_off(s); t.sh = t.sc + round_up(s.C, 8); return t; };
    double* part = reinterpret_cast<double*>(ws + L.part);
    const PackDesc* d_pack = reinterpret_cast<const PackDesc*>(d_desc);
    const BnEvalDesc* d_bn = reinterpret_cast<const BnEvalDesc*>(d_desc + n_pack * sizeof(PackDesc));

    // weights -> kernel layout (fp32 -> T); eval: all BN tables from the running statistics in one launch
    if ((rc = pack_weights(d_pack, n_pack, mode, st))) return rc;
    if (!train && (rc = bn_eval_tables(d_bn, n_bneval, kEps, st))) return rc;

    auto link = [&](const BnSlots& s, const double* prt, int nblk, int part_ld, int c_new0, int n_new, double* bstat,
                    long count, const long long* isum = nullptr, long isum_stride = 0) -> int {
        if (!train) return 0;
        BnLinkArgs a{};
        a.isum = isum; a.isum_stride = isum_stride;                  // window sums in fixed-point accumulators (a link-free producer) instead of partial rows
        a.part = prt; a.nblk = nblk; a.part_ld = part_ld; a.c_new0 = c_new0; a.n_new = n_new; a.bstat = bstat;
        a.count = count; a.C = s.C; a.gamma = data[s.w]; a.beta = data[s.b];
        a.running_mean = data[s.rm]; a.running_var = data[s.rv];
        Tab t = tab(s); a.sc = t.sc; a.sh = t.sh; a.train = 1; a.eps = kEps; a.momentum = kMom;
        return bn_link(a, st);
    };

    // ---- stem ----
    // Round 5, link-free BatchNorm statistics (bn_lf.h): the fused 1x1 kernels and the 3x3 pair kernel ADD their per-workgroup sums to
    // fixed-point accumulators and derive their input BatchNorm's table in their own prologue -- no k_bn_link launch between them (120 of
    // the 132 per step and embedder).  TCVN_NO_LF (validation build): the link kernels of rounds 1-4.
    static const bool no_lf = TCVN_KNOB_SET("TCVN_NO_LF");
    const bool lf_on = train && !no_lf && mode == MODE_BF16 && L.isum_bytes > 0;
    TCVN_CHECK(hipMemsetAsync(ws + L.zeros, 0, 1024 + (lf_on ? L.isum_bytes : 0), st));
    auto lf_of = [&](const BnSlots& s, const long long* isum, long rep_stride, int c_new0, int n_new, double* bstat, long count) {
        LfLink k{};
        k.isum = isum; k.rep_stride = rep_stride; k.c_new0 = c_new0; k.n_new = n_new; k.bstat = bstat; k.inv_count = 1.0 / (double)count; k.count = count;
        k.gamma = data[s.w]; k.beta = data[s.b]; k.running_mean = data[s.rm]; k.running_var = data[s.rv];
        Tab t = tab(s); k.sc_out = t.sc; k.sh_out = t.sh; k.eps = kEps; k.momentum = kMom;
        return k;
    };
    // Sparse-aware stem (bf16, 3 -> 64 channels, the hit list fits the index): conv0 + BN0 + PReLU0 + AvgPool straight from the COO
    // list, neither the dense map nor the conv0 output is materialised (stem_sparse.hip).  Otherwise: scatter + dense kernels.
    // Measured on MI355X (256 prong maps / 32 event maps, round 3): inference -- index + pooled pass 0.60 / 0.27 ms against 0.97 / 0.12 ms
    // for the dense conv0 + pooling kernels, so eval-mode forwards take the sparse stem.  Training -- the statistics pass adds 0.34 / 0.16 ms
    // (forward at parity with the dense kernels) and the sparse backward passes (0.87 + 4.3 ms) lose clearly to the dense backward
    // (0.40 + 0.50 ms: its weight gradient already walks the hit list), so train-mode steps keep the dense stem.  The validation build
    // can force either (TCVN_DENSE_STEM / TCVN_SPARSE_STEM_TRAIN) for the parity tests of all four sparse passes.
    static const bool dense_stem_knob = TCVN_KNOB_SET("TCVN_DENSE_STEM");
    static const bool sparse_train_knob = TCVN_KNOB_SET("TCVN_SPARSE_STEM_TRAIN");
    const bool sparse_stem = !dense_stem_knob && (!train || sparse_train_knob) && L.sidx >= 0 &&
                             stem_sparse_ok(mode, cfg.in_ch, cfg.init_ch, cfg.H, cfg.W, log_pixels, nnz, n, blocks[0].ld);
    last_sparse_stem = sparse_stem;
    last_stem_act = false;
    last_fused_ya = false;
    last_values = values; last_value_mode = log_pixels; last_noise = train ? noise_std : 0.f;
    int init_nblk = 0, init_ld = cfg.init_ch;
    if (sparse_stem) {
        const BlockGeom& b0 = blocks[0];
        const WkEntry& e = wk_find(s_w0, 0);
        StemSparseArgs sa{};
        sa.coords = coords; sa.values = values; sa.nnz = nnz; sa.n_img = n; sa.H = cfg.H; sa.W = cfg.W; sa.Cpix = cfg.in_ch;
        sa.value_mode = log_pixels; sa.noise_std = train ? noise_std : 0.f; sa.seed = seed;
        stem_sparse_carve(sa, ws + L.sidx);
        sa.Wk = ws + L.wk + e.off; sa.Kp = e.Kp; sa.bias = data[s_b0];
        sa.Hc = Hc; sa.Wc = Wc; sa.Ho = b0.H; sa.Wo = b0.W;
        if ((rc = stem_sparse_index(sa, st))) return rc;
        if (train) {
            sa.part = part;
            if ((rc = stem_sparse_stats(sa, st))) return rc;
            if ((rc = link(n0, part, stem_sparse_stats_grid(sa), cfg.init_ch, 0, cfg.init_ch, reinterpret_cast<double*>(ws + L.bstat0), (long)n * Hc * Wc))) return rc;
        }
        Tab t = tab(n0);
        sa.sc = t.sc; sa.sh = t.sh; sa.sl = data[s_a0]; sa.Out = ws + L.D[0]; sa.ldo = b0.ld; sa.part = train ? part : nullptr;
        if ((rc = stem_sparse_pool(sa, st))) return rc;
        init_nblk = stem_sparse_pool_grid(sa);
    } else {
    TCVN_CHECK(hipMemsetAsync(ws + L.img, 0, (size_t)n * cfg.H * cfg.W * cfg.in_ch * esz, st));
    {
        ScatterArgs a{mode, coords, values, nnz, n, ws + L.img, cfg.H, cfg.W, cfg.in_ch, log_pixels, train ? noise_std : 0.f, seed};
        if ((rc = scatter_pixels(a, st))) return rc;
    }
    const long M0 = (long)n * Hc * Wc;
    // Round 4: the maps are mostly empty, so most conv0 outputs are exactly bf16(bias) (a sum of zeros plus the bias).  stem_mark builds a bitmap
    // of the output positions some hit reaches (~16 % of a prong map's 28 000 at 20-800 hits); the pooling BACKWARD kernel reads one shared row for
    // every other position and does not store their gradient rows (only the hit-list weight gradient reads that tensor).  Bit-identical results.
    // Measured and NOT kept: the same bitmap in conv0 (stores skipped) and in the forward pooling kernel -- the forward pooling got slower
    // (prong embedder 271 -> 345 us: its nine loads per pixel turn into a mix of L2 hits and isolated 128-B HBM lines, which DRAM serves far
    // below its streaming rate), the backward gained 10 %, the step did not move (19.65-19.76 against 19.74-19.86 ms).
    static const bool no_stem_skip = TCVN_KNOB_SET("TCVN_NO_STEM_SKIP") || TCVN_KNOB_SET("TCVN_POOL0_BWD_FLAT");   // (the flat backward kernel reads every row)
    uint32_t* sact = nullptr;
    const void* cline = ws + L.zeros + 512;                 // the zero page is 1 KB; DMA sources use its first 256 B
    {
        const WkEntry& e0 = wk_find(s_w0, 0);
        ConvFwdArgs probe{};
        probe.mode = mode; probe.amode = A_STEM; probe.M = (int)M0; probe.N = cfg.init_ch; probe.K = 49 * cfg.in_ch; probe.Kp = e0.Kp; probe.C = cfg.in_ch;
        probe.H = Hc; probe.W = Wc; probe.Hin = cfg.H; probe.Win = cfg.W; probe.ldo = cfg.init_ch; probe.n_off = 0; probe.Out = ws + L.c0;
        if (!no_stem_skip && train && L.sact >= 0 && stem_fwd_ok(probe) && (blocks[0].ld & 7) == 0 && coords != nullptr) {
            sact = reinterpret_cast<uint32_t*>(ws + L.sact);
            if ((rc = stem_mark(coords, nnz, n, cfg.H, cfg.W, Hc, Wc, sact, data[s_b0], const_cast<void*>(cline), st))) return rc;
        }
    }
    last_stem_act = sact != nullptr;
    {
        const WkEntry& e = wk_find(s_w0, 0);
        ConvFwdArgs a{};
        a.mode = mode; a.amode = A_STEM; a.A = ws + L.img; a.lda = cfg.in_ch; a.M = (int)M0; a.N = cfg.init_ch;
        a.K = 49 * cfg.in_ch; a.Kp = e.Kp; a.C = cfg.in_ch; a.H = Hc; a.W = Wc; a.Hin = cfg.H; a.Win = cfg.W;
        a.Wk = ws + L.wk + e.off; a.bias = data[s_b0]; a.Out = ws + L.c0; a.ldo = cfg.init_ch; a.n_off = 0;
        a.part = train ? part : nullptr; a.nblk = conv_fwd_nblk(a);
        if ((rc = conv_fwd(a, st))) return rc;
        if ((rc = link(n0, part, a.nblk, cfg.init_ch, 0, cfg.init_ch, reinterpret_cast<double*>(ws + L.bstat0), M0))) return rc;
    }
    {
        const BlockGeom& b0 = blocks[0];
        Tab t = tab(n0);
        Pool0Args a{mode, ws + L.c0, n, Hc, Wc, cfg.init_ch, t.sc, t.sh, data[s_a0], ws + L.D[0], b0.ld, b0.H, b0.W,
                    train ? part : nullptr, pool0_grid(n, b0.H, b0.W)};
        if ((rc = pool0_fwd(a, st))) return rc;
    }
    init_nblk = pool0_grid(n, blocks[0].H, blocks[0].W);
    }
    // statistics of the block's initial channels are described by (init_nblk, init_ld)

    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        const BlockGeom& bg = blocks[bi];
        const long M = (long)n * bg.H * bg.W;
        char* D = ws + L.D[bi];
        double* bstatD = reinterpret_cast<double*>(ws + L.bstatD[bi]);
        int new_c0 = 0, new_n = bg.C0, new_nblk = init_nblk, new_ld = init_ld;   // channels whose stats are fresh in `part`
        long long* isumD = lf_on ? reinterpret_cast<long long*>(ws + L.isumD[bi]) : nullptr;      // [ld][2], indexed by channel
        bool new_isum = false;                                                    // ... or in isumD (added there by a link-free producer)
        for (int l = 0; l < bg.L; ++l) {
            const LayerSlots& ls = bg.layers[l];
            const bool fast1 = L.XA[bi][l] >= 0;
            const int cin8 = (int)round_up(ls.cin, 8);
            // Eval mode (running statistics: no batch reduction between the 1x1 output and its BatchNorm): the 1x1 GEMM's epilogue
            // applies norm2 + PReLU and writes the activated map the 3x3 tile kernel stages -- the raw bottleneck output Y and the
            // k_act_bf16 pass over it (512 B per pixel and layer, one launch) do not exist.  Train mode needs Y for the statistics.
            bool fuse_ya = false;
            if (mode == MODE_BF16 && !train && fast1 && xa_materialize()) {
                ConvFwdArgs c3{};
                c3.mode = mode; c3.amode = A_3X3; c3.A = ws + L.Y[bi][l]; c3.lda = mid; c3.M = (int)M; c3.N = g; c3.K = 9 * mid;
                c3.Kp = wk_find(ls.w2, 0).Kp; c3.C = mid; c3.H = bg.H; c3.W = bg.W; c3.Wk = ws + L.wk + wk_find(ls.w2, 0).off;
                c3.Wfrag = wk_frag(ws, L, ls.w2, 0); c3.Aact = ws + L.YA[bi][l]; c3.zeros = ws + L.zeros;
                fuse_ya = conv3x3_tile_ok(c3);
            }
            if (fuse_ya) last_fused_ya = true;
            if (xa_skipped.size() != blocks.size()) xa_skipped.assign(blocks.size(), std::vector<char>());
            if ((int)xa_skipped[bi].size() != bg.L) xa_skipped[bi].assign(bg.L, 0);
            xa_skipped[bi][l] = 0;
            // Train mode (round 4): the 1x1 runs on the RAW concat buffer, norm1 + PReLU1 applied to the landed LDS tiles (fwd1x1_fused.hip);
            // the fused 1x1 backward kernel rebuilds that activation from x, so the activated copy XA is neither written nor read.
            static const bool no_fuse1 = TCVN_KNOB_SET("TCVN_NO_FWD1_FUSE") || TCVN_KNOB_SET("TCVN_NO_BWD1_FUSE");
            static const bool no_wide1 = TCVN_KNOB_SET("TCVN_NO_FWD1_WIDE");         // validation build: wide layers on k_act_bf16 + the 128-row GEMM
            bool lf2 = false;                  // the 3x3 pair kernel derives norm2's table itself (no link launch in front of it)
            bool n1_linked = false;
            auto link_n1 = [&]() -> int {      // norm1's table by the link kernel (window sums from the partial rows or from isumD)
                n1_linked = true;
                return link(ls.n1, part, new_nblk, new_ld, new_c0, new_n, bstatD, M, new_isum ? isumD + 2 * new_c0 : nullptr, 2L * bg.ld);
            };
            if ((train || fuse_ya) && fast1 && !no_fuse1 && mid == 128 && mode == MODE_BF16 && !(no_wide1 && wk_find(ls.w1, 0, 1).Kp > 256)) {
                const WkEntry& e = wk_find(ls.w1, 0, 1);
                Tab t1 = tab(ls.n1);
                Fwd1x1Args fa{};
                fa.Xin = D; fa.ldx = bg.ld; fa.cin = ls.cin; fa.sc = t1.sc; fa.sh = t1.sh; fa.sl = data[ls.a1]; fa.M = M;
                fa.Wfrag = ws + L.wk + e.off; fa.Kp = e.Kp; fa.bias = data[ls.b1]; fa.Out = ws + L.Y[bi][l]; fa.zeros = ws + L.zeros;
                fa.part = train ? part : nullptr; fa.nblk = fwd1x1_fused_nblk(fa);
                if (!train) {                  // eval: norm2 + PReLU2 in the epilogue, the activated map is the only output (as k_gemm_nt_bf16<.., XF = 2>)
                    Tab t2 = tab(ls.n2);
                    fa.osc = t2.sc; fa.osh = t2.sh; fa.osl = data[ls.a2]; fa.Out = ws + L.YA[bi][l];
                }
                // train mode: the activated copy XA is only dropped when the backward's fused 1x1 kernel will accept this layer (it rebuilds
                // the activation from x); otherwise the step would die in backward after the forward has already run
                if (fwd1x1_fused_ok(fa) && (!train || bwd1x1_fusable((int)bi, l, M, ws, L))) {
                    long long* isumY = lf_on ? reinterpret_cast<long long*>(ws + L.isumY[bi][l]) : nullptr;
                    // link-free consumer of norm1: the fresh window was added to isumD by the previous layer's 3x3 kernel (the block's first
                    // layer follows a transition / the stem, whose statistics still leave as partial rows: link kernel)
                    static const bool lf_no1 = TCVN_KNOB_SET("TCVN_LF_NO1"), lf_no2 = TCVN_KNOB_SET("TCVN_LF_NO2");      // validation build: the link kernel
                    if (lf_on && new_isum && !lf_no1) fa.lf = lf_of(ls.n1, isumD + 2 * new_c0, 2L * bg.ld, new_c0, new_n, bstatD, M);
                    else if ((rc = link_n1())) return rc;
                    fa.isum_out = isumY; fa.isum_stride = 2L * mid;       // link-free producer of norm2's statistics
                    if ((rc = fwd1x1_fused(fa, st))) return rc;
                    // norm2's consumer: the 3x3 pair kernel with the activation in LDS derives the table itself; anything else takes the link kernel
                    if (isumY != nullptr) {
                        ConvFwdArgs c3{};
                        c3.mode = mode; c3.amode = A_3X3; c3.A = ws + L.Y[bi][l]; c3.lda = mid; c3.M = (int)M; c3.N = g; c3.K = 9 * mid;
                        c3.Kp = wk_find(ls.w2, 0).Kp; c3.C = mid; c3.H = bg.H; c3.W = bg.W; c3.Wk = ws + L.wk + wk_find(ls.w2, 0).off;
                        c3.Wfrag = wk_frag(ws, L, ls.w2, 0); c3.Aact = ws + L.Y[bi][l]; c3.zeros = ws + L.zeros;
                        Tab t2 = tab(ls.n2);
                        c3.sc = t2.sc; c3.sh = t2.sh; c3.sl = data[ls.a2];
                        lf2 = conv3x3_act_fusable(c3) && conv3x3_fwd_pair(c3) && !lf_no2;
                    }
                    if (!lf2 && (rc = link(ls.n2, part, fa.nblk, mid, 0, mid, reinterpret_cast<double*>(ws + L.bstatY[bi][l]), M, isumY, 2L * mid))) return rc;
                    xa_skipped[bi][l] = 1;
                    goto conv3;
                }
            }
            if (!n1_linked && (rc = link_n1())) return rc;
            if (fast1 && xa_materialize()) {     // activated copy of the 1x1 input: operand of the bf16 GEMMs (forward, weight gradient)
                Tab t1 = tab(ls.n1);
                ActArgs act{D, bg.ld, M, ls.cin, t1.sc, t1.sh, data[ls.a1], ws + L.XA[bi][l], cin8};
                if ((rc = act_bf16(act, st))) return rc;
            }
            if (fast1) {   // bottleneck 1x1 on the NT GEMM: XA x W1^T -> Y
                const WkEntry& e = wk_find(ls.w1, 0, 1);
                GemmNtArgs a{};
                a.epi = EPI_FWD; a.A = ws + L.XA[bi][l]; a.lda = cin8; a.K = cin8; a.M = M; a.N = mid;
                if (!xa_materialize()) {      // raw concat buffer in, BatchNorm + PReLU applied to every landed LDS tile
                    Tab t1 = tab(ls.n1);
                    a.A = D; a.lda = bg.ld; a.asc = t1.sc; a.ash = t1.sh; a.asl = data[ls.a1]; a.Kreal = ls.cin;
                }
                a.Wfrag = ws + L.wk + e.off; a.Kp = e.Kp; a.zeros = ws + L.zeros; a.bias = data[ls.b1];
                a.Out = ws + L.Y[bi][l]; a.ldo = mid; a.n_off = 0; a.part = train ? part : nullptr; a.nblk = gemm_nt_nblk(a);
                if (fuse_ya) {                 // eval: norm2 + PReLU in the GEMM epilogue, the activated map is the only output
                    Tab t2 = tab(ls.n2);
                    a.osc = t2.sc; a.osh = t2.sh; a.osl = data[ls.a2]; a.Out = ws + L.YA[bi][l];
                }
                if ((rc = gemm_nt_bf16(a, "k_gemm_nt_bf16<fwd1x1>", st))) return rc;
                if ((rc = link(ls.n2, part, a.nblk, mid, 0, mid, reinterpret_cast<double*>(ws + L.bstatY[bi][l]), M))) return rc;
            } else {   // bottleneck 1x1: D[:, 0:cin] -> Y
                const WkEntry& e = wk_find(ls.w1, 0);
                Tab t = tab(ls.n1);
                ConvFwdArgs a{};
                a.mode = mode; a.amode = A_1X1; a.A = D; a.lda = bg.ld; a.M = (int)M; a.N = mid; a.K = ls.cin; a.Kp = e.Kp;
                a.C = ls.cin; a.H = bg.H; a.W = bg.W; a.sc = t.sc; a.sh = t.sh; a.sl = data[ls.a1];
                a.Wk = ws + L.wk + e.off; a.bias = data[ls.b1]; a.Out = ws + L.Y[bi][l]; a.ldo = mid; a.n_off = 0;
                a.part = train ? part : nullptr; a.nblk = conv_fwd_nblk(a);
                if ((rc = conv_fwd(a, st))) return rc;
                if ((rc = link(ls.n2, part, a.nblk, mid, 0, mid, reinterpret_cast<double*>(ws + L.bstatY[bi][l]), M))) return rc;
            }
        conv3:
            {   // 3x3: Y -> D[:, cin:cin+g]
                const WkEntry& e = wk_find(ls.w2, 0);
                Tab t = tab(ls.n2);
                ConvFwdArgs a{};
                a.mode = mode; a.amode = A_3X3; a.A = ws + L.Y[bi][l]; a.lda = mid; a.M = (int)M; a.N = g; a.K = 9 * mid;
                a.Kp = e.Kp; a.C = mid; a.H = bg.H; a.W = bg.W; a.sc = t.sc; a.sh = t.sh; a.sl = data[ls.a2];
                a.Wk = ws + L.wk + e.off; a.bias = data[ls.b2]; a.Out = D; a.ldo = bg.ld; a.n_off = ls.cin;
                a.Wfrag = wk_frag(ws, L, ls.w2, 0);
                if (act_fused.size() != blocks.size()) act_fused.assign(blocks.size(), std::vector<char>());
                if ((int)act_fused[bi].size() != bg.L) act_fused[bi].assign(bg.L, 0);
                act_fused[bi][l] = 0;
                if (mode == MODE_BF16) {
                    a.zeros = ws + L.zeros;
                    // Round 4: norm2 + PReLU are applied INSIDE the 3x3 kernel (and inside the layer's weight-gradient kernel): the raw
                    // bottleneck map is staged by LDS-DMA and the wave that fetched a row activates it in LDS once -- the activated copy YA
                    // (256 B written + 256 B read per pixel and layer) and the k_act_bf16 launch over Y do not exist.  Bit-identical images.
                    a.Aact = ws + L.Y[bi][l];
                    if (!fuse_ya && conv3x3_act_fusable(a)) { a.act_fused = 1; act_fused[bi][l] = 1; }
                    else {                     // materialise prelu(bn(Y)) once; the tile kernel stages it by LDS-DMA
                        if (!fuse_ya) {
                            ActArgs act{ws + L.Y[bi][l], mid, M, mid, t.sc, t.sh, data[ls.a2], ws + L.YA[bi][l], mid};
                            if ((rc = act_bf16(act, st))) return rc;
                        }
                        a.Aact = ws + L.YA[bi][l];
                    }
                }
                a.part = train ? part : nullptr;
                a.drop_p = train ? cfg.dropout : 0.f; a.seed = seed; a.stream_id = (uint32_t)(bi * 64 + l + 1);
                a.nblk = conv_fwd_nblk(a);
                if (lf2) {                     // (decided above on the same arguments: pair kernel + in-LDS activation)
                    if (!a.act_fused) { fprintf(stderr, "tcvn: link-free norm2 without the in-LDS activation\n"); return -17; }
                    a.lf = lf_of(ls.n2, reinterpret_cast<long long*>(ws + L.isumY[bi][l]), 2L * mid, 0, mid, reinterpret_cast<double*>(ws + L.bstatY[bi][l]), M);
                }
                const bool out_isum = lf_on && mode == MODE_BF16 && conv3x3_fwd_pair(a) && g <= 32;      // link-free producer of the new channels' statistics
                if (out_isum) { a.isum_out = isumD + 2 * ls.cin; a.isum_stride = 2L * bg.ld; }
                if (train && !L.KM[bi].empty()) {           // the pair kernel leaves the keep flags it drew for the backward kernels
                    a.keep_out = reinterpret_cast<uint32_t*>(ws + L.KM[bi][l]);
                    if (keep_valid.size() != blocks.size()) keep_valid.assign(blocks.size(), std::vector<char>());
                    if ((int)keep_valid[bi].size() != bg.L) keep_valid[bi].assign(bg.L, 0);
                    keep_valid[bi][l] = conv3x3_fwd_writes_keep(a) ? 1 : 0;
                }
                if ((rc = conv_fwd(a, st))) return rc;
                new_c0 = ls.cin; new_n = g; new_nblk = a.nblk; new_ld = g; new_isum = out_isum;
            }
        }
        if (bg.has_trans) {
            const BlockGeom& nb = blocks[bi + 1];
            if ((rc = link(bg.tn, part, new_nblk, new_ld, new_c0, new_n, bstatD, M, new_isum ? isumD + 2 * new_c0 : nullptr, 2L * bg.ld))) return rc;
            const WkEntry& e = wk_find(bg.tw, 0);
            Tab t = tab(bg.tn);
            const long Mn = (long)n * nb.H * nb.W;
            const bool fastt = L.XP[bi] >= 0 && mode == MODE_BF16;
            const bool mat32 = L.XP[bi] >= 0 && mode == MODE_F32;
            if (fastt || mat32) {
                ActPoolArgs ap{D, bg.ld, n, bg.H, bg.W, bg.Ctot, t.sc, t.sh, data[bg.ta], ws + L.XP[bi], bg.ldp};
                if ((rc = fastt ? act_pool_bf16(ap, st) : act_pool_f32(ap, st))) return rc;
            }
            if (fastt) {
                const WkEntry& ef = wk_find(bg.tw, 0, 1);
                GemmNtArgs ga{};
                ga.epi = EPI_FWD; ga.A = ws + L.XP[bi]; ga.lda = bg.ldp; ga.K = bg.ldp; ga.M = Mn; ga.N = bg.Ctot / 2;
                ga.Wfrag = ws + L.wk + ef.off; ga.Kp = ef.Kp; ga.zeros = ws + L.zeros; ga.bias = data[bg.tb];
                ga.Out = ws + L.D[bi + 1]; ga.ldo = nb.ld; ga.n_off = 0; ga.part = train ? part : nullptr; ga.nblk = gemm_nt_nblk(ga);
                if ((rc = gemm_nt_bf16(ga, "k_gemm_nt_bf16<fwdtrans>", st))) return rc;
                init_nblk = ga.nblk; init_ld = bg.Ctot / 2;
                continue;
            }
            ConvFwdArgs a{};
            a.mode = mode; a.amode = A_1X1_POOL; a.A = D; a.lda = bg.ld; a.M = (int)Mn; a.N = bg.Ctot / 2; a.K = bg.Ctot;
            a.Kp = e.Kp; a.C = bg.Ctot; a.H = nb.H; a.W = nb.W; a.Hin = bg.H; a.Win = bg.W;
            a.sc = t.sc; a.sh = t.sh; a.sl = data[bg.ta];
            if (mat32) {            // pooled + activated operand materialised above: a plain 1x1 convolution over it (K padded to ldp: zero columns x zero weights)
                a.amode = A_1X1; a.A = ws + L.XP[bi]; a.lda = bg.ldp; a.K = bg.ldp; a.C = bg.ldp; a.sc = nullptr; a.sh = nullptr; a.sl = nullptr;
            }
            a.Wk = ws + L.wk + e.off; a.bias = data[bg.tb]; a.Out = ws + L.D[bi + 1]; a.ldo = nb.ld; a.n_off = 0;
            a.part = train ? part : nullptr; a.nblk = conv_fwd_nblk(a);
            if ((rc = conv_fwd(a, st))) return rc;
            init_nblk = a.nblk; init_ld = bg.Ctot / 2;
        } else {
            if ((rc = link(nf, part, new_nblk, new_ld, new_c0, new_n, bstatD, M, new_isum ? isumD + 2 * new_c0 : nullptr, 2L * bg.ld))) return rc;
            Tab t = tab(nf);
            HeadPoolArgs a{mode, D, bg.ld, n, bg.H * bg.W, Cf, t.sc, t.sh, data[s_af], reinterpret_cast<float*>(ws + L.F)};
            if ((rc = head_pool_fwd(a, st))) return rc;
        }
    }
    // ---- output block: Linear(no bias) - BatchNorm1d - PReLU - Dropout (layers/dense_net.py:157-162) ----
    float* F = reinterpret_cast<float*>(ws + L.F);
    float* Z = reinterpret_cast<float*>(ws + L.Z);
    if ((rc = linear_fwd(F, Cf, data[s_wl], nullptr, Z, cfg.out_dim, n, cfg.out_dim, Cf, st))) return rc;
    float* hs = reinterpret_cast<float*>(ws + L.head_stat);
    RowsBnArgs r{};
    r.X = Z; r.ldx = cfg.out_dim; r.R = n; r.C = cfg.out_dim; r.gamma = data[nl.w]; r.beta = data[nl.b]; r.slope = data[s_al];
    r.running_mean = data[nl.rm]; r.running_var = data[nl.rv]; r.Y = out; r.ldy = out_ld;
    r.save_mean = hs; r.save_rstd = hs + cfg.out_dim; r.train = train; r.eps = kEps; r.momentum = kMom;
    r.drop_p = train ? cfg.dropout : 0.f; r.seed = seed; r.stream_id = 0x4000u;
    if ((rc = rows_bn_fwd(r, st))) return rc;
    last_seed = seed; last_n = n; last_coords = coords; last_nnz = nnz;
    return 0;
}

int DenseNetPlan::tap(int n, const char* name, long* off, int* tn, int* th, int* tw, int* tc, int* tld, int* tes) const {
    Layout L;
    layout(n, false, L);
    std::string s(name);
    *tn = n; *tes = esz;
    if ((s == "img" || s == "conv0") && last_sparse_stem && n == last_n) return -1;      // the sparse stem materialises neither
    if (s == "img") { *off = L.img; *th = cfg.H; *tw = cfg.W; *tc = cfg.in_ch; *tld = cfg.in_ch; return 0; }
    if (s == "conv0") { *off = L.c0; *th = Hc; *tw = Wc; *tc = cfg.init_ch; *tld = cfg.init_ch; return 0; }
    if (s == "condense") { *off = L.F; *th = 1; *tw = 1; *tc = Cf; *tld = Cf; *tes = 4; return 0; }
    if (s == "raw:wk") { *tn = *th = *tw = 1; *off = L.wk; *tc = *tld = (int)(wk_bytes() / esz); return 0; }
    if (s == "raw:tabs") { *tn = *th = *tw = 1; *off = L.tabs; *tc = *tld = (int)tab_floats(); *tes = 4; return 0; }
    if (s.rfind("raw:ystat", 0) == 0) {                          // (mean, biased var) of a bottleneck map: "raw:ystat<block>.<layer>"
        int b = 0, l = 0;
        if (sscanf(s.c_str(), "raw:ystat%d.%d", &b, &l) != 2) return -1;
        b -= 1;
        if (b < 0 || b >= (int)blocks.size() || l < 0 || l >= blocks[b].L) return -1;
        *tn = *th = *tw = 1; *off = L.bstatY[b][l]; *tc = *tld = 2 * cfg.bn_size * cfg.growth; *tes = 8; return 0;
    }
    if (s.rfind("raw:bstat", 0) == 0) {
        const int b = atoi(s.c_str() + 9) - 1;
        if (b < 0 || b >= (int)blocks.size()) return -1;
        *tn = *th = *tw = 1; *off = L.bstatD[b]; *tc = *tld = blocks[b].ld * 2; *tes = 8; return 0;
    }
    if (s.rfind("dense", 0) == 0) {
        const int b = atoi(s.c_str() + 5) - 1;
        if (b < 0 || b >= (int)blocks.size()) return -1;
        *off = L.D[b]; *th = blocks[b].H; *tw = blocks[b].W; *tc = blocks[b].Ctot; *tld = blocks[b].ld;
        return 0;
    }
    if (s.rfind("bottleneck", 0) == 0) {
        int b = 0, l = 0;
        if (sscanf(s.c_str(), "bottleneck%d.%d", &b, &l) != 2) return -1;
        b -= 1;
        if (b < 0 || b >= (int)blocks.size() || l < 0 || l >= blocks[b].L) return -1;
        if (last_fused_ya && n == last_n) return -1;             // eval pass with the activation in the GEMM epilogue: no raw Y exists
        *off = L.Y[b][l]; *th = blocks[b].H; *tw = blocks[b].W; *tc = cfg.bn_size * cfg.growth; *tld = *tc;
        return 0;
    }
    if (s.rfind("xa", 0) == 0 || s.rfind("ya", 0) == 0) {      // bf16 mode: materialised activations (1x1 / 3x3 operands)
        int b = 0, l = 0;
        if (sscanf(s.c_str() + 2, "%d.%d", &b, &l) != 2) return -1;
        b -= 1;
        if (b < 0 || b >= (int)blocks.size() || l < 0 || l >= blocks[b].L) return -1;
        const bool xa = s[0] == 'x';
        if ((xa && (!xa_materialize() || L.XA[b].empty() || L.XA[b][l] < 0)) || (!xa && L.YA[b].empty())) return -1;
        if (xa && n == last_n && b < (int)xa_skipped.size() && l < (int)xa_skipped[b].size() && xa_skipped[b][l]) return -1;                 // 1x1 ran on the raw buffer
        if (!xa && n == last_n && b < (int)act_fused.size() && l < (int)act_fused[b].size() && act_fused[b][l]) return -1;   // activated in LDS only
        *off = xa ? L.XA[b][l] : L.YA[b][l]; *th = blocks[b].H; *tw = blocks[b].W;
        *tc = xa ? blocks[b].layers[l].cin : cfg.bn_size * cfg.growth; *tld = xa ? (int)round_up(*tc, 8) : *tc;
        return 0;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
struct tcvn_densenet { DenseNetPlan plan; explicit tcvn_densenet(const tcvn_densenet_cfg& c) : plan(c) {} };

extern "C" {

int tcvn_version(void) { return 1; }
void tcvn_backward_overlap(int on) { tcvn::set_backward_overlap(on); }

int tcvn_densenet_create(const tcvn_densenet_cfg* cfg, tcvn_densenet** out) {
    if (!cfg || !out || cfg->n_blocks < 1 || cfg->n_blocks > 8) return -1;
    if (cfg->mode != TCVN_MODE_F32 && cfg->mode != TCVN_MODE_BF16) return -1;
    *out = new tcvn_densenet(*cfg);
    return 0;
}
void tcvn_densenet_destroy(tcvn_densenet* p) { delete p; }
int tcvn_densenet_num_slots(const tcvn_densenet* p) { return (int)p->plan.slots.size(); }
int tcvn_densenet_slot(const tcvn_densenet* p, int i, char* name, int cap, int64_t* numel, int* kind) {
    if (i < 0 || i >= (int)p->plan.slots.size()) return -1;
    const auto& s = p->plan.slots[i];
    if (name && cap > 0) { strncpy(name, s.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (numel) *numel = s.numel;
    if (kind) *kind = s.kind;
    return 0;
}
int tcvn_densenet_bind(tcvn_densenet* p, void* const* data, void* const* grad) { return p->plan.bind(data, grad); }
int64_t tcvn_densenet_workspace_bytes(const tcvn_densenet* p, int n_img, int with_backward) {
    Layout L;
    p->plan.layout(n_img, with_backward != 0, L);
    return L.total;
}
int tcvn_densenet_forward(tcvn_densenet* p, int n_img, const int32_t* coords, const float* values, int64_t nnz, int log_pixels,
                          float noise_std, float* out, int64_t out_ld, void* ws, int64_t ws_bytes, int train, uint64_t seed,
                          void* stream) {
    return p->plan.forward(n_img, coords, values, nnz, log_pixels, noise_std, out, out_ld, reinterpret_cast<char*>(ws), ws_bytes,
                           train, seed, reinterpret_cast<hipStream_t>(stream));
}
int tcvn_densenet_backward(tcvn_densenet* p, int n_img, const float* d_out, int64_t d_out_ld, void* ws, int64_t ws_bytes,
                           void* stream) {
    return p->plan.backward(n_img, d_out, d_out_ld, reinterpret_cast<char*>(ws), ws_bytes, reinterpret_cast<hipStream_t>(stream));
}
int tcvn_densenet_num_blocks(const tcvn_densenet* p) { return (int)p->plan.blocks.size(); }
int tcvn_densenet_backward_blocks(tcvn_densenet* p, int n_img, const float* d_out, int64_t d_out_ld, void* ws, int64_t ws_bytes,
                                  int block_hi, int block_lo, void* stream) {
    return p->plan.backward(n_img, d_out, d_out_ld, reinterpret_cast<char*>(ws), ws_bytes, reinterpret_cast<hipStream_t>(stream), block_hi,
                            block_lo);
}
int tcvn_densenet_tap(const tcvn_densenet* p, int n_img, const char* name, int64_t* byte_off, int* n, int* h, int* w, int* c,
                      int* ld, int* elem_bytes) {
    long off = 0;
    int rc = p->plan.tap(n_img, name, &off, n, h, w, c, ld, elem_bytes);
    *byte_off = off;
    return rc;
}

}  // extern "C"
