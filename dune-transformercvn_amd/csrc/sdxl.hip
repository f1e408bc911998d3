// SDXL-style embedder engine (reference: transformercvn/network/layers/sdxl_net.py:7-42 = diffusers' VAE Encoder with
// block_out_channels [d,d,2d,2d,4d,4d,8d,8d,out], norm_num_groups 1, then Flatten + Linear(out,out); selected by
// networks/neutrino_full_sdxl_network.py:6-20 / train.py --sdxl).  PARITY UNPINNED: diffusers is not vendored, pinned or
// installed; the block definitions this schedule follows are restated and cited in oracle/sdxl_oracle.py.
//
// The network is held as a TAPE of two operator kinds over NHWC activation buffers in the caller's workspace:
//   CONV  out = conv(in; w, b, kernel, stride, top-left pad) [+ residual buffer]        (sdxl_kernels.hip, implicit GEMM)
//   GN    out = silu?(groupnorm_1group(in; gamma, beta, eps 1e-6))                        (two-phase reduction)
// forward walks the tape, backward walks it in reverse: every buffer has a gradient buffer, the first contribution
// writes it and later ones accumulate.  The mid-block attention runs over the H*W tokens of the final map; the plan requires
// that map to be 1x1 (as the reference's Flatten + Linear(out,out) does), where softmax == 1 and the block reduces to
// x + to_out(to_v(groupnorm(x))); to_q / to_k keep their slots and receive zero gradients.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tcvn_hip.h"
#include "densenet_plan.h"      // Slot
#include "sdxl_ops.h"
#include "tcvn_ops.h"           // scatter_pixels, pack_weights, unpack_wgrads

using namespace tcvn;

namespace {
constexpr float kGnEps = 1e-6f;
enum { OP_CONV = 0, OP_GN = 1 };
struct SBuf { int H, W, C; };
struct SOp { int kind, in, out, res, w, b, ks, stride, pad, act, conv_id, gn_id, final_f32; };
struct Bump {
    long off = 0;
    long take(long bytes) { long o = off; off += round_up(bytes, 256); return o; }
};
struct SLayout {
    std::vector<long> act, grad;            // per buffer (act[0] = image); grad[0] unused
    std::vector<long> wk, wkt, gwk;         // per conv
    long stats, bstats, zero_begin, zero_end, slab, total;
};
}  // namespace

struct tcvn_sdxl {
    tcvn_sdxl_cfg cfg;
    int esz;
    std::vector<Slot> slots;
    std::vector<float*> data, grad;
    std::vector<SBuf> bufs;
    std::vector<SOp> ops;
    int n_conv = 0, n_gn = 0;
    bool bound = false;
    int last_n = 0;
    const int32_t* last_coords = nullptr; long last_nnz = 0;       // COO list of the last forward (conv_in weight gradient)
    char* d_desc = nullptr; size_t desc_cap = 0; char* desc_ws = nullptr; long desc_total = 0; int n_pack = 0;
    char* d_undesc = nullptr; char* undesc_ws = nullptr; long undesc_total = 0; int n_unpack = 0;
    std::vector<char> h_desc, h_undesc;

    explicit tcvn_sdxl(const tcvn_sdxl_cfg& c);
    ~tcvn_sdxl() { if (d_desc) (void)hipFree(d_desc); if (d_undesc) (void)hipFree(d_undesc); }
    int slot(const std::string& name, long numel) { slots.push_back({name, numel, TCVN_SLOT_PARAM}); return (int)slots.size() - 1; }
    int new_buf(int H, int W, int C) { bufs.push_back({H, W, C}); return (int)bufs.size() - 1; }
    int conv(const std::string& p, int in, int cout, int ks, int stride, int pad, int res, int Ho, int Wo, bool linear_keys = false);
    int gn(const std::string& p, int in, int act);
    int resnet(const std::string& p, int in, int cout);
    void layout(int n, bool bwd, SLayout& L) const;
    int Kp(const SOp& o) const { return (int)round_up((long)o.ks * o.ks * bufs[o.in].C, 32); }
    int Kpt(const SOp& o) const { return (int)round_up((long)o.ks * o.ks * bufs[o.out].C, 32); }
    SConv geom(const SOp& o, int n) const;
    int forward(int n, const int32_t* coords, const float* values, long nnz, int log_pixels, float noise_std, float* out, long out_ld,
                char* ws, long ws_bytes, int train, uint64_t seed, hipStream_t st);
    int backward(int n, const float* d_out, long d_out_ld, char* ws, long ws_bytes, hipStream_t st);
};

int tcvn_sdxl::conv(const std::string& p, int in, int cout, int ks, int stride, int pad, int res, int Ho, int Wo, bool linear_keys) {
    SOp o{};
    o.kind = OP_CONV; o.in = in; o.res = res; o.ks = ks; o.stride = stride; o.pad = pad; o.conv_id = n_conv++;
    const int cin = bufs[in].C;
    o.w = slot(p + ".weight", (long)cout * cin * ks * ks);
    o.b = slot(p + ".bias", cout);
    (void)linear_keys;
    o.out = new_buf(Ho, Wo, cout);
    ops.push_back(o);
    return o.out;
}
int tcvn_sdxl::gn(const std::string& p, int in, int act) {
    SOp o{};
    o.kind = OP_GN; o.in = in; o.res = -1; o.act = act; o.gn_id = n_gn++;
    o.w = slot(p + ".weight", bufs[in].C);
    o.b = slot(p + ".bias", bufs[in].C);
    o.out = new_buf(bufs[in].H, bufs[in].W, bufs[in].C);
    ops.push_back(o);
    return o.out;
}
// ResnetBlock2D: x + conv2(silu(gn2(conv1(silu(gn1(x)))))), x through a 1x1 conv_shortcut when the width changes.
// Slot order = module registration order: norm1, conv1, norm2, conv2, conv_shortcut.
int tcvn_sdxl::resnet(const std::string& p, int in, int cout) {
    const int H = bufs[in].H, W = bufs[in].W, cin = bufs[in].C;
    const int a1 = gn(p + ".norm1", in, 1);
    const int h1 = conv(p + ".conv1", a1, cout, 3, 1, 1, -1, H, W);
    const int a2 = gn(p + ".norm2", h1, 1);
    // conv2 is registered before conv_shortcut, but the shortcut's output is conv2's residual: emit the op first, fix the slots after
    if (cin == cout) return conv(p + ".conv2", a2, cout, 3, 1, 1, in, H, W);
    const size_t s0 = slots.size();
    const int w2 = slot(p + ".conv2.weight", (long)cout * cout * 9), b2 = slot(p + ".conv2.bias", cout);
    const int sc = conv(p + ".conv_shortcut", in, cout, 1, 1, 0, -1, H, W);
    (void)s0;
    SOp o{};
    o.kind = OP_CONV; o.in = a2; o.res = sc; o.ks = 3; o.stride = 1; o.pad = 1; o.conv_id = n_conv++; o.w = w2; o.b = b2;
    o.out = new_buf(H, W, cout);
    ops.push_back(o);
    return o.out;
}

tcvn_sdxl::tcvn_sdxl(const tcvn_sdxl_cfg& c) : cfg(c) {
    esz = cfg.mode == MODE_F32 ? 4 : 2;
    std::vector<int> chans;
    int d = cfg.init_ch;
    for (int b = 0; b < cfg.num_blocks; ++b) { for (int r = 0; r < cfg.repeat; ++r) chans.push_back(d); d *= 2; }
    chans.push_back(cfg.out_dim);
    const std::string e = "encoder";
    int h = new_buf(cfg.H, cfg.W, cfg.in_ch);                                     // buffer 0: the scattered pixel map
    h = conv(e + ".conv_in", h, chans[0], 3, 1, 1, -1, cfg.H, cfg.W);
    for (size_t i = 0; i < chans.size(); ++i) {
        const std::string p = e + ".down_blocks." + std::to_string(i);
        for (int j = 0; j < 2; ++j) h = resnet(p + ".resnets." + std::to_string(j), h, chans[i]);
        if (i + 1 != chans.size()) {                                              // F.pad(0,1,0,1) + 3x3 stride 2, padding 0
            const int Ho = (bufs[h].H + 1 - 3) / 2 + 1, Wo = (bufs[h].W + 1 - 3) / 2 + 1;
            h = conv(p + ".downsamplers.0.conv", h, chans[i], 3, 2, 0, -1, Ho, Wo);
        }
    }
    const int C = chans.back();
    // mid block.  Registration order in diffusers: attentions, then resnets; execution: resnet0, attention, resnet1.
    const std::string m = e + ".mid_block";
    const int s_gn_w = slot(m + ".attentions.0.group_norm.weight", C), s_gn_b = slot(m + ".attentions.0.group_norm.bias", C);
    slot(m + ".attentions.0.to_q.weight", (long)C * C); slot(m + ".attentions.0.to_q.bias", C);
    slot(m + ".attentions.0.to_k.weight", (long)C * C); slot(m + ".attentions.0.to_k.bias", C);
    const int s_v_w = slot(m + ".attentions.0.to_v.weight", (long)C * C), s_v_b = slot(m + ".attentions.0.to_v.bias", C);
    const int s_o_w = slot(m + ".attentions.0.to_out.0.weight", (long)C * C), s_o_b = slot(m + ".attentions.0.to_out.0.bias", C);
    h = resnet(m + ".resnets.0", h, C);
    {
        const int H = bufs[h].H, W = bufs[h].W;
        SOp g{}; g.kind = OP_GN; g.in = h; g.res = -1; g.act = 0; g.gn_id = n_gn++; g.w = s_gn_w; g.b = s_gn_b; g.out = new_buf(H, W, C);
        ops.push_back(g);
        SOp v{}; v.kind = OP_CONV; v.in = g.out; v.res = -1; v.ks = 1; v.stride = 1; v.pad = 0; v.conv_id = n_conv++; v.w = s_v_w; v.b = s_v_b;
        v.out = new_buf(H, W, C); ops.push_back(v);
        SOp o{}; o.kind = OP_CONV; o.in = v.out; o.res = h; o.ks = 1; o.stride = 1; o.pad = 0; o.conv_id = n_conv++; o.w = s_o_w; o.b = s_o_b;
        o.out = new_buf(H, W, C); ops.push_back(o);
        h = o.out;
    }
    h = resnet(m + ".resnets.1", h, C);
    const int a = gn(e + ".conv_norm_out", h, 1);
    h = conv(e + ".conv_out", a, cfg.out_dim, 3, 1, 1, -1, bufs[a].H, bufs[a].W);
    h = conv("output_layer.1", h, cfg.out_dim, 1, 1, 0, -1, bufs[h].H, bufs[h].W);
    ops.back().final_f32 = 1;
    data.assign(slots.size(), nullptr);
    grad.assign(slots.size(), nullptr);
}

void tcvn_sdxl::layout(int n, bool bwd, SLayout& L) const {
    Bump b;
    L.act.assign(bufs.size(), -1); L.grad.assign(bufs.size(), -1);
    for (size_t i = 0; i + 1 < bufs.size(); ++i) L.act[i] = b.take((long)n * bufs[i].H * bufs[i].W * bufs[i].C * esz);
    L.wk.assign(n_conv, -1); L.wkt.assign(n_conv, -1); L.gwk.assign(n_conv, -1);
    for (const auto& o : ops)
        if (o.kind == OP_CONV) {
            L.wk[o.conv_id] = b.take((long)bufs[o.out].C * Kp(o) * esz);
            if (o.in != 0) L.wkt[o.conv_id] = b.take((long)bufs[o.in].C * Kpt(o) * esz);
        }
    L.zero_begin = b.off;
    L.stats = b.take((long)n_gn * n * 16);
    if (bwd) {
        L.bstats = b.take((long)n_gn * n * 16);
        for (const auto& o : ops)
            if (o.kind == OP_CONV) L.gwk[o.conv_id] = b.take((long)bufs[o.out].C * Kp(o) * 4);
    } else L.bstats = -1;
    L.zero_end = b.off;
    L.slab = bwd ? b.take(kSconvSlabBytes) : -1;
    if (bwd)
        for (size_t i = 1; i < bufs.size(); ++i) L.grad[i] = b.take((long)n * bufs[i].H * bufs[i].W * bufs[i].C * esz);
    L.total = b.off;
}

SConv tcvn_sdxl::geom(const SOp& o, int n) const {
    SConv g{};
    g.mode = cfg.mode; g.n = n; g.Hin = bufs[o.in].H; g.Win = bufs[o.in].W; g.Cin = bufs[o.in].C; g.lda = g.Cin;
    g.Ho = bufs[o.out].H; g.Wo = bufs[o.out].W; g.Cout = bufs[o.out].C; g.ks = o.ks; g.stride = o.stride; g.pad = o.pad;
    g.Kp = Kp(o); g.Kpt = Kpt(o);
    return g;
}

int tcvn_sdxl::forward(int n, const int32_t* coords, const float* values, long nnz, int log_pixels, float noise_std, float* out,
                       long out_ld, char* ws, long ws_bytes, int train, uint64_t seed, hipStream_t st) {
    if (!bound) return -11;
    if (n <= 0) return 0;
    const SBuf& last = bufs.back();
    if (last.H != 1 || last.W != 1) {
        fprintf(stderr, "tcvn: SDXL embedder needs a 1x1 final map (Flatten + Linear(out,out)); %dx%d maps end at %dx%d\n", cfg.H, cfg.W, last.H, last.W);
        return -22;
    }
    SLayout L;
    layout(n, train != 0, L);
    if (ws_bytes < L.total) { fprintf(stderr, "tcvn: sdxl workspace too small (%ld < %ld)\n", ws_bytes, L.total); return -12; }
    int rc;
    if (desc_ws != ws || desc_total != L.total) {                       // weight packing descriptors depend on the workspace address
        std::vector<PackDesc> pd;
        for (const auto& o : ops)
            if (o.kind == OP_CONV) {
                const int cin = bufs[o.in].C, cout = bufs[o.out].C;
                pd.push_back(PackDesc{data[o.w], ws + L.wk[o.conv_id], cout, cin, o.ks * o.ks, Kp(o), 0, 0});
                if (o.in != 0) pd.push_back(PackDesc{data[o.w], ws + L.wkt[o.conv_id], cout, cin, o.ks * o.ks, Kpt(o), 1, 0});
            }
        n_pack = (int)pd.size();
        const size_t bytes = pd.size() * sizeof(PackDesc);
        if (bytes > desc_cap) { if (d_desc) TCVN_CHECK(hipFree(d_desc)); TCVN_CHECK(hipMalloc(&d_desc, bytes)); desc_cap = bytes; }
        h_desc.assign(reinterpret_cast<char*>(pd.data()), reinterpret_cast<char*>(pd.data()) + bytes);
        TCVN_CHECK(hipMemcpyAsync(d_desc, h_desc.data(), bytes, hipMemcpyHostToDevice, st));
        desc_ws = ws; desc_total = L.total;
    }
    if ((rc = pack_weights(reinterpret_cast<const PackDesc*>(d_desc), n_pack, cfg.mode, st))) return rc;
    TCVN_CHECK(hipMemsetAsync(ws + L.act[0], 0, (size_t)n * cfg.H * cfg.W * cfg.in_ch * esz, st));
    TCVN_CHECK(hipMemsetAsync(ws + L.stats, 0, (size_t)n_gn * n * 16, st));
    {
        ScatterArgs a{cfg.mode, coords, values, nnz, n, ws + L.act[0], cfg.H, cfg.W, cfg.in_ch, log_pixels, train ? noise_std : 0.f, seed};
        if ((rc = scatter_pixels(a, st))) return rc;
    }
    // a buffer consumed by exactly one GroupNorm and produced by a convolution gets its statistics from that convolution's epilogue
    std::vector<int> gn_of(bufs.size(), -1), conv_made(bufs.size(), 0);
    for (const auto& o : ops) {
        if (o.kind == OP_GN) gn_of[o.in] = gn_of[o.in] == -1 ? o.gn_id : -2;
        else if (!o.final_f32) conv_made[o.out] = 1;
    }
    for (const auto& o : ops) {
        if (o.kind == OP_GN) {
            GnArgs a{cfg.mode, ws + L.act[o.in], bufs[o.in].C, n, bufs[o.in].H * bufs[o.in].W, bufs[o.in].C, data[o.w], data[o.b], kGnEps, o.act,
                     reinterpret_cast<double*>(ws + L.stats) + (long)o.gn_id * n * 2};
            if (!(gn_of[o.in] >= 0 && conv_made[o.in]) && (rc = gn_stats(a, st))) return rc;
            if ((rc = gn_act(a, ws + L.act[o.out], bufs[o.out].C, st))) return rc;
        } else {
            SConv g = geom(o, n);
            if (!o.final_f32 && gn_of[o.out] >= 0) g.stats = reinterpret_cast<double*>(ws + L.stats) + (long)gn_of[o.out] * n * 2;
            void* dst = o.final_f32 ? reinterpret_cast<void*>(out) : reinterpret_cast<void*>(ws + L.act[o.out]);
            const long ldo = o.final_f32 ? out_ld : g.Cout;
            if ((rc = sconv_fwd(g, ws + L.act[o.in], ws + L.wk[o.conv_id], data[o.b], o.res >= 0 ? ws + L.act[o.res] : nullptr,
                                o.res >= 0 ? bufs[o.res].C : 0, dst, ldo, o.final_f32, st))) return rc;
        }
    }
    last_n = n; last_coords = coords; last_nnz = nnz;
    return 0;
}

int tcvn_sdxl::backward(int n, const float* d_out, long d_out_ld, char* ws, long ws_bytes, hipStream_t st) {
    if (!bound) return -11;
    if (n <= 0) return 0;
    if (n != last_n) { fprintf(stderr, "tcvn: sdxl backward without matching forward\n"); return -13; }
    for (size_t i = 0; i < slots.size(); ++i)
        if (grad[i] == nullptr) { fprintf(stderr, "tcvn: grad of %s unbound\n", slots[i].name.c_str()); return -14; }
    SLayout L;
    layout(n, true, L);
    if (ws_bytes < L.total) return -12;
    int rc;
    if (undesc_ws != ws || undesc_total != L.total) {
        std::vector<UnpackDesc> ud;
        for (const auto& o : ops)
            if (o.kind == OP_CONV)
                ud.push_back(UnpackDesc{reinterpret_cast<const float*>(ws + L.gwk[o.conv_id]), grad[o.w], bufs[o.out].C, bufs[o.in].C,
                                        o.ks * o.ks, Kp(o), 0});
        n_unpack = (int)ud.size();
        if (!d_undesc) TCVN_CHECK(hipMalloc(&d_undesc, ud.size() * sizeof(UnpackDesc)));
        h_undesc.assign(reinterpret_cast<char*>(ud.data()), reinterpret_cast<char*>(ud.data()) + ud.size() * sizeof(UnpackDesc));
        TCVN_CHECK(hipMemcpyAsync(d_undesc, h_undesc.data(), h_undesc.size(), hipMemcpyHostToDevice, st));
        undesc_ws = ws; undesc_total = L.total;
    }
    // zero: backward GroupNorm sums and the kernel-layout weight gradients (forward statistics sit in front of them and stay)
    TCVN_CHECK(hipMemsetAsync(ws + L.bstats, 0, (size_t)(L.zero_end - L.bstats), st));
    std::vector<char> written(bufs.size(), 0);
    // gradient region of every buffer for THIS call: a residual input that has no contribution yet simply TAKES OVER the region of the output
    // gradient (same shape; dead once its producer has been processed) instead of receiving a copy of it (round 5: ten device-to-device copies of
    // up to 2 GB, 3.1 ms per step)
    std::vector<long> goff(L.grad.begin(), L.grad.end());
    const int last = (int)bufs.size() - 1;
    // the last buffer is the caller's fp32 output; its gradient arrives as fp32 too
    if ((rc = cast_f32_to(cfg.mode, d_out, d_out_ld, ws + goff[last], bufs[last].C, n, bufs[last].C, st))) return rc;
    written[last] = 1;
    for (int oi = (int)ops.size() - 1; oi >= 0; --oi) {
        const SOp& o = ops[oi];
        if (!written[o.out]) { fprintf(stderr, "tcvn: sdxl backward: gradient of buffer %d never produced\n", o.out); return -15; }
        const char* dO = ws + goff[o.out];
        const long rows_in = (long)n * bufs[o.in].H * bufs[o.in].W;
        if (o.kind == OP_GN) {
            GnArgs a{cfg.mode, ws + L.act[o.in], bufs[o.in].C, n, bufs[o.in].H * bufs[o.in].W, bufs[o.in].C, data[o.w], data[o.b], kGnEps, o.act,
                     reinterpret_cast<double*>(ws + L.stats) + (long)o.gn_id * n * 2};
            double* bs = reinterpret_cast<double*>(ws + L.bstats) + (long)o.gn_id * n * 2;
            if ((rc = gn_bwd_reduce(a, dO, bufs[o.out].C, bs, grad[o.w], grad[o.b], st))) return rc;
            if ((rc = gn_bwd_apply(a, dO, bufs[o.out].C, bs, ws + goff[o.in], bufs[o.in].C, written[o.in], st))) return rc;
            written[o.in] = 1;
        } else {
            SConv g = geom(o, n);
            g.slab = reinterpret_cast<float*>(ws + L.slab); g.slab_bytes = kSconvSlabBytes;
            if (o.in == 0) { g.hits = last_coords; g.nnz = last_nnz; }
            if ((rc = sconv_wgrad(g, ws + L.act[o.in], dO, g.Cout, reinterpret_cast<float*>(ws + L.gwk[o.conv_id]), grad[o.b], st))) return rc;
            if (o.in != 0) {
                if ((rc = sconv_dgrad(g, dO, g.Cout, ws + L.wkt[o.conv_id], ws + goff[o.in], g.Cin, written[o.in], st))) return rc;
                written[o.in] = 1;
            }
            if (o.res >= 0) {                                            // out = conv(...) + res: the residual receives dOut as is
                const long rows = (long)n * bufs[o.res].H * bufs[o.res].W;
                if (written[o.res]) { if ((rc = add_into(cfg.mode, ws + goff[o.res], bufs[o.res].C, dO, g.Cout, rows, g.Cout, st))) return rc; }
                else if (bufs[o.res].C == g.Cout && rows == (long)n * bufs[o.out].H * bufs[o.out].W) std::swap(goff[o.res], goff[o.out]);      // stream order: this op's kernels read dO first
                else TCVN_CHECK(hipMemcpyAsync(ws + goff[o.res], dO, (size_t)rows * g.Cout * esz, hipMemcpyDeviceToDevice, st));
                written[o.res] = 1;
            }
        }
        (void)rows_in;
    }
    return unpack_wgrads(reinterpret_cast<const UnpackDesc*>(d_undesc), n_unpack, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" {

int tcvn_sdxl_create(const tcvn_sdxl_cfg* cfg, tcvn_sdxl** out) {
    if (!cfg || !out || cfg->num_blocks < 1 || cfg->num_blocks > 6 || cfg->repeat < 1 || cfg->repeat > 4 || cfg->init_ch < 1) return -1;
    if (cfg->mode != TCVN_MODE_F32 && cfg->mode != TCVN_MODE_BF16) return -1;
    *out = new tcvn_sdxl(*cfg);
    return 0;
}
void tcvn_sdxl_destroy(tcvn_sdxl* p) { delete p; }
int tcvn_sdxl_num_slots(const tcvn_sdxl* p) { return (int)p->slots.size(); }
int tcvn_sdxl_slot(const tcvn_sdxl* p, int i, char* name, int cap, int64_t* numel, int* kind) {
    if (i < 0 || i >= (int)p->slots.size()) return -1;
    const auto& s = p->slots[i];
    if (name && cap > 0) { strncpy(name, s.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (numel) *numel = s.numel;
    if (kind) *kind = s.kind;
    return 0;
}
int tcvn_sdxl_bind(tcvn_sdxl* p, void* const* d, void* const* g) {
    for (size_t i = 0; i < p->slots.size(); ++i) {
        p->data[i] = reinterpret_cast<float*>(d[i]);
        p->grad[i] = g ? reinterpret_cast<float*>(g[i]) : nullptr;
        if (p->data[i] == nullptr) { fprintf(stderr, "tcvn: slot %s unbound\n", p->slots[i].name.c_str()); return -10; }
    }
    p->bound = true; p->desc_ws = nullptr; p->undesc_ws = nullptr;
    return 0;
}
int64_t tcvn_sdxl_workspace_bytes(const tcvn_sdxl* p, int n_img, int with_backward) {
    SLayout L;
    p->layout(n_img, with_backward != 0, L);
    return L.total;
}
int tcvn_sdxl_forward(tcvn_sdxl* p, int n_img, const int32_t* coords, const float* values, int64_t nnz, int log_pixels, float noise_std,
                      float* out, int64_t out_ld, void* ws, int64_t ws_bytes, int train, uint64_t seed, void* stream) {
    return p->forward(n_img, coords, values, nnz, log_pixels, noise_std, out, out_ld, reinterpret_cast<char*>(ws), ws_bytes, train, seed,
                      reinterpret_cast<hipStream_t>(stream));
}
int tcvn_sdxl_backward(tcvn_sdxl* p, int n_img, const float* d_out, int64_t d_out_ld, void* ws, int64_t ws_bytes, void* stream) {
    return p->backward(n_img, d_out, d_out_ld, reinterpret_cast<char*>(ws), ws_bytes, reinterpret_cast<hipStream_t>(stream));
}
/* tap: "conv_in", "block<i>" (output of down block i before its downsampler), "mid" -> byte offset + NHWC shape */
int tcvn_sdxl_tap(const tcvn_sdxl* p, int n_img, const char* name, int64_t* byte_off, int* n, int* h, int* w, int* c, int* ld,
                  int* elem_bytes) {
    SLayout L;
    p->layout(n_img, false, L);
    const std::string s(name);
    int buf = -1;
    int convs = 0, target = -1;
    if (s == "conv_in") buf = p->ops[0].out;
    else if (s == "img") buf = 0;
    else {
        // walk the tape: down block i ends at the input of its stride-2 conv (or, for the last block, at the first mid-block GN)
        std::vector<int> block_end;
        for (const auto& o : p->ops)
            if (o.kind == OP_CONV && o.stride == 2) block_end.push_back(o.in);
        (void)convs; (void)target;
        if (s.rfind("block", 0) == 0) {
            const int i = atoi(s.c_str() + 5);
            const int nb = p->cfg.num_blocks * p->cfg.repeat + 1;
            if (i < 0 || i >= nb) return -1;
            if (i < (int)block_end.size()) buf = block_end[i];
            else {                                   // last block: two resnets after the last downsampler
                int seen = 0;
                for (size_t k = 0; k < p->ops.size(); ++k)
                    if (p->ops[k].kind == OP_CONV && p->ops[k].stride == 2 && ++seen == (int)block_end.size()) {
                        int resn = 0;
                        for (size_t q = k + 1; q < p->ops.size(); ++q)
                            if (p->ops[q].kind == OP_CONV && p->ops[q].ks == 3 && p->ops[q].res >= 0 && ++resn == 2) { buf = p->ops[q].out; break; }
                        break;
                    }
            }
        } else if (s == "mid") {
            for (const auto& o : p->ops)
                if (o.kind == OP_GN && o.act == 1) buf = o.in;      // the last SiLU GroupNorm is conv_norm_out: its input is the mid block's output
        }
    }
    if (buf < 0 || L.act[buf] < 0) return -1;
    *byte_off = L.act[buf]; *n = n_img; *h = p->bufs[buf].H; *w = p->bufs[buf].W; *c = p->bufs[buf].C; *ld = p->bufs[buf].C;
    *elem_bytes = p->esz;
    return 0;
}

}  // extern "C"
