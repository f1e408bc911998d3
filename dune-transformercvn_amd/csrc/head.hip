// Token path engine: combined embedding (LinearBlock), transformer encoder, event / prong decoders and the softmax
// focal loss; forward, loss and backward on one stream.  Reference call chain:
//   networks/neutrino_full_base_network.py:113-125 (combined embedding, pad), :184-188 (encoder, decoders)
//   layers/prong_custom_bert_encoder.py:57-75, layers/prong_decoder.py:15-16, layers/prong_target_decoder.py:34-41
//   trainers/neutrino_full_base_trainer.py:148-177 (loss)
// All arithmetic fp32 (this path is < 0.03 % of the FLOPs).
#include <string>
#include <vector>
#include <cstring>

#include "../../include/tcvn_hip.h"
#include "head_plan.h"
#include "tcvn_rows.h"
#include "tcvn_encoder.h"

using namespace tcvn;

namespace {
constexpr float kEps = 1e-5f, kMom = 0.1f;

__global__ void k_combine_loss(const float* oe, const float* op, float ew, float* losses, float* accs) {
    losses[0] = ew * oe[0] + (1.f - ew) * op[0];
    losses[1] = oe[0]; losses[2] = op[0];
    accs[0] = oe[1]; accs[1] = op[1];
}
__global__ void k_scale(float* x, long n, float s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s;
}
struct Bump {
    long off = 0;
    long take(long bytes) { long o = off; off += round_up(bytes, 256); return o; }
};
}  // namespace

int HeadPlan::add_slot(const std::string& name, long numel, int kind) {
    slots.push_back({name, numel, kind});
    return (int)slots.size() - 1;
}
HBn HeadPlan::add_bn(const std::string& p, int c) {
    HBn s;
    s.w = add_slot(p + ".weight", c, TCVN_SLOT_PARAM);
    s.b = add_slot(p + ".bias", c, TCVN_SLOT_PARAM);
    s.rm = add_slot(p + ".running_mean", c, TCVN_SLOT_BUFFER);
    s.rv = add_slot(p + ".running_var", c, TCVN_SLOT_BUFFER);
    add_slot(p + ".num_batches_tracked", 1, TCVN_SLOT_COUNTER);
    return s;
}

HeadPlan::HeadPlan(const tcvn_head_cfg& c) : cfg(c) {
    const int D = cfg.hidden_dim;
    const std::string ce = "prong_embedding.combined_embedding";
    // LinearBlock (layers/prong_feature_embedding.py:7-33): Linear(bias = not linear_batch_norm) - BatchNorm1d | Identity - PReLU | ReLU - Dropout
    const bool bn = !cfg.no_linear_bn, prelu_act = !cfg.linear_relu;
    cw = add_slot(ce + ".linear.weight", (long)D * cfg.in_dim, TCVN_SLOT_PARAM);
    if (!bn) cb = add_slot(ce + ".linear.bias", D, TCVN_SLOT_PARAM);
    if (bn) cn = add_bn(ce + ".norm", D);
    if (prelu_act) ca = add_slot(ce + ".activation.weight", D, TCVN_SLOT_PARAM);
    for (int l = 0; l < cfg.n_layers; ++l) {
        const std::string p = "encoder.encoder.layers." + std::to_string(l);
        HLayer L;
        L.win = add_slot(p + ".self_attn.in_proj_weight", 3L * D * D, TCVN_SLOT_PARAM);
        L.bin = add_slot(p + ".self_attn.in_proj_bias", 3 * D, TCVN_SLOT_PARAM);
        L.wo = add_slot(p + ".self_attn.out_proj.weight", (long)D * D, TCVN_SLOT_PARAM);
        L.bo = add_slot(p + ".self_attn.out_proj.bias", D, TCVN_SLOT_PARAM);
        L.w1 = add_slot(p + ".linear1.weight", (long)D * D, TCVN_SLOT_PARAM);
        L.b1 = add_slot(p + ".linear1.bias", D, TCVN_SLOT_PARAM);
        L.w2 = add_slot(p + ".linear2.weight", (long)D * D, TCVN_SLOT_PARAM);
        L.b2 = add_slot(p + ".linear2.bias", D, TCVN_SLOT_PARAM);
        L.g1 = add_slot(p + ".norm1.weight", D, TCVN_SLOT_PARAM);
        L.be1 = add_slot(p + ".norm1.bias", D, TCVN_SLOT_PARAM);
        L.g2 = add_slot(p + ".norm2.weight", D, TCVN_SLOT_PARAM);
        L.be2 = add_slot(p + ".norm2.bias", D, TCVN_SLOT_PARAM);
        layers.push_back(L);
    }
    ew = add_slot("event_decoder.hidden_layer.weight", (long)cfg.event_classes * D, TCVN_SLOT_PARAM);
    eb = add_slot("event_decoder.hidden_layer.bias", cfg.event_classes, TCVN_SLOT_PARAM);
    int idx = 0, in = D;
    for (int i = 0; i < cfg.n_dec; ++i) {
        const std::string p = "prong_decoder.hidden_layers.";
        HDec d;
        d.in = in; d.out = cfg.dec_dims[i];
        // create_linear_block (layers/encoder.py:10-24): [Linear, BatchNorm1d?, PReLU | ReLU, Dropout?] -- the Sequential indices follow
        d.w = add_slot(p + std::to_string(idx) + ".weight", (long)d.out * d.in, TCVN_SLOT_PARAM);
        d.b = add_slot(p + std::to_string(idx) + ".bias", d.out, TCVN_SLOT_PARAM);
        if (bn) d.n = add_bn(p + std::to_string(idx + 1), d.out);
        if (prelu_act) d.a = add_slot(p + std::to_string(idx + 1 + (bn ? 1 : 0)) + ".weight", d.out, TCVN_SLOT_PARAM);
        idx += 2 + (bn ? 1 : 0) + (cfg.dropout_modules ? 1 : 0);
        in = d.out;
        dec.push_back(d);
    }
    dec_width = in;
    ow = add_slot("prong_decoder.output_layer.weight", (long)cfg.prong_classes * cfg.dec_out_in, TCVN_SLOT_PARAM);
    ob = add_slot("prong_decoder.output_layer.bias", cfg.prong_classes, TCVN_SLOT_PARAM);
    data.assign(slots.size(), nullptr);
    grad.assign(slots.size(), nullptr);
}

int HeadPlan::bind(void* const* d, void* const* g) {
    for (size_t i = 0; i < slots.size(); ++i) {
        data[i] = reinterpret_cast<float*>(d[i]);
        grad[i] = g ? reinterpret_cast<float*>(g[i]) : nullptr;
        if (slots[i].kind != TCVN_SLOT_COUNTER && data[i] == nullptr) return -10;
    }
    bound = true;
    return 0;
}

void HeadPlan::layout(int B, int P, int nP, HLayout& L) const {
    Bump b;
    const int D = cfg.hidden_dim, S = 1 + P, T = S * B, R = B + nP, TP = P * B, H = cfg.heads;
    const long td = (long)T * D * 4;
    L.Zc = b.take((long)R * D * 4); L.C = b.take((long)R * D * 4); L.cstat = b.take(2L * D * 4);
    L.X.clear(); L.lay.clear();
    for (int l = 0; l <= cfg.n_layers; ++l) L.X.push_back(b.take(td));
    for (int l = 0; l < cfg.n_layers; ++l) {
        HLayBuf q;
        q.qkv = b.take(3 * td); q.probs = b.take((long)B * H * S * S * 4); q.ctx = b.take(td); q.ao = b.take(td);
        q.xh1 = b.take(td); q.rstd1 = b.take((long)T * 4); q.x1 = b.take(td); q.hpre = b.take(td); q.hact = b.take(td);
        q.f = b.take(td); q.xh2 = b.take(td); q.rstd2 = b.take((long)T * 4);
        q.h1 = cfg.norm_first ? b.take(td) : -1; q.h2 = cfg.norm_first ? b.take(td) : -1;
        L.lay.push_back(q);
    }
    L.HID = b.take(td);
    L.Zd.clear(); L.Ad.clear(); L.dstat.clear();
    for (const auto& d : dec) {
        L.Zd.push_back(b.take((long)TP * d.out * 4)); L.Ad.push_back(b.take((long)TP * d.out * 4));
        L.dstat.push_back(b.take(2L * d.out * 4));
    }
    L.LG = b.take((long)TP * cfg.prong_classes * 4);
    L.dEv = b.take((long)B * cfg.event_classes * 4); L.dPr = b.take((long)B * P * cfg.prong_classes * 4);
    L.lossbuf = b.take(64);
    // backward scratch
    L.dLG = b.take((long)TP * cfg.prong_classes * 4);
    L.t0 = b.take(3 * td); L.t1 = b.take(3 * td); L.t2 = b.take(td); L.t3 = b.take(td); L.dHID = b.take(td);
    L.dC = b.take((long)R * D * 4); L.dZc = b.take((long)R * D * 4);
    // fused encoder backward: per-layer weight-gradient operands and per-event LayerNorm sums
    for (int l = 0; l < cfg.n_layers; ++l) {
        HLayBuf& q = L.lay[l];
        q.g_dqkv = b.take(3 * td); q.g_dao = b.take(td); q.g_dhp = b.take(td); q.g_df = b.take(td);
    }
    L.lnp = b.take((long)B * cfg.n_layers * 4 * D * 4);
    L.total = b.off;
}

// ---- stage 1: combined embedding (LinearBlock over event + packed prong rows) and the token gather -> L.X[0] ------------
int HeadPlan::embed(int B, int P, int nP, const float* rows, const int32_t* tok_row, char* ws, const HLayout& L, int train,
                    uint64_t seed, hipStream_t st) {
    const int D = cfg.hidden_dim, S = 1 + P, R = B + nP;
    const float dp = train ? cfg.dropout : 0.f;
    auto F = [&](long off) { return reinterpret_cast<float*>(ws + off); };
    int rc;
    if ((rc = linear_fwd(rows, cfg.in_dim, data[cw], cb >= 0 ? data[cb] : nullptr, F(L.Zc), D, R, D, cfg.in_dim, st))) return rc;
    RowsBnArgs r{};
    r.X = F(L.Zc); r.ldx = D; r.R = R; r.C = D; r.slope = ca >= 0 ? data[ca] : nullptr; r.no_norm = cfg.no_linear_bn;
    if (!r.no_norm) { r.gamma = data[cn.w]; r.beta = data[cn.b]; r.running_mean = data[cn.rm]; r.running_var = data[cn.rv]; }
    r.Y = F(L.C); r.ldy = D;
    r.save_mean = F(L.cstat); r.save_rstd = F(L.cstat) + D; r.train = train; r.eps = kEps; r.momentum = kMom;
    r.drop_p = dp; r.seed = seed; r.stream_id = 0x5000u;
    if ((rc = rows_bn_fwd(r, st))) return rc;
    return gather_tokens(F(L.C), tok_row, F(L.X[0]), B, S, D, st);
}

// ---- stage 2: transformer encoder, L.X[0] (sequence-major tokens, padding rows zero) -> L.HID (masked) ------------------
int HeadPlan::encode(int B, int P, const int32_t* tok_row, char* ws, const HLayout& L, int train, uint64_t seed, hipStream_t st) {
    const int D = cfg.hidden_dim, S = 1 + P, T = S * B, H = cfg.heads, hd = D / H;
    const float dp = train ? cfg.dropout : 0.f;
    auto F = [&](long off) { return reinterpret_cast<float*>(ws + off); };
    int rc;
    if (fused_encoder && encoder_fused_ok(S, D, H, cfg.n_layers, cfg.norm_first)) {      // one launch for the whole stack (+ mask)
        EncFusedArgs a{};
        a.X0 = F(L.X[0]); a.tok_row = tok_row; a.HID = F(L.HID); a.B = B; a.S = S; a.H = H; a.L = cfg.n_layers; a.gelu = cfg.gelu;
        a.save = 1; a.eps = kEps; a.drop_p = dp; a.seed = seed;
        for (int l = 0; l < cfg.n_layers; ++l) {
            const HLayer& W = layers[l];
            const HLayBuf& q = L.lay[l];
            a.w[l] = EncLayerW{data[W.win], data[W.bin], data[W.wo], data[W.bo], data[W.w1], data[W.b1], data[W.w2], data[W.b2],
                               data[W.g1], data[W.be1], data[W.g2], data[W.be2]};
            a.buf[l] = EncLayerBuf{F(q.qkv), F(q.probs), F(q.ctx), F(q.xh1), F(q.rstd1), F(q.x1), F(q.hpre), F(q.hact), F(q.xh2),
                                   F(q.rstd2), F(L.X[l + 1])};
        }
        return encoder_fused_fwd(a, st);
    }
    if (cfg.norm_first) {
        // pre-norm variant (prong_custom_bert_encoder.py:45-52 with transformer_norm_first): x += drop(sa(LN1(x))); x += drop(ff(LN2(x))).
        // LN(x) is add_ln_fwd with a zero residual branch (t0 is scratch of the backward pass, free here).
        float* zero = F(L.t0);
        TCVN_CHECK(hipMemsetAsync(zero, 0, (size_t)T * D * 4, st));
        for (int l = 0; l < cfg.n_layers; ++l) {
            const HLayer& W = layers[l];
            const HLayBuf& q = L.lay[l];
            const uint32_t sid = 0x6000u + l * 8;
            AddLnArgs n1{F(L.X[l]), zero, data[W.g1], data[W.be1], F(q.h1), F(q.xh1), F(q.rstd1), T, D, kEps, 0.f, seed, sid + 1};
            if ((rc = add_ln_fwd(n1, st))) return rc;
            if ((rc = linear_fwd(F(q.h1), D, data[W.win], data[W.bin], F(q.qkv), 3 * D, T, 3 * D, D, st))) return rc;
            AttnArgs a{F(q.qkv), tok_row, F(q.probs), F(q.ctx), B, S, H, hd, dp, seed, sid};
            if ((rc = attn_fwd(a, st))) return rc;
            if ((rc = linear_fwd(F(q.ctx), D, data[W.wo], data[W.bo], F(q.ao), D, T, D, D, st))) return rc;
            if ((rc = add_drop(F(L.X[l]), F(q.ao), F(q.x1), (long)T * D, dp, seed, sid + 1, st))) return rc;
            AddLnArgs n2{F(q.x1), zero, data[W.g2], data[W.be2], F(q.h2), F(q.xh2), F(q.rstd2), T, D, kEps, 0.f, seed, sid + 3};
            if ((rc = add_ln_fwd(n2, st))) return rc;
            if ((rc = linear_fwd(F(q.h2), D, data[W.w1], data[W.b1], F(q.hpre), D, T, D, D, st))) return rc;
            if ((rc = act_fwd(F(q.hpre), F(q.hact), (long)T * D, cfg.gelu, dp, seed, sid + 2, st))) return rc;
            if ((rc = linear_fwd(F(q.hact), D, data[W.w2], data[W.b2], F(q.f), D, T, D, D, st))) return rc;
            if ((rc = add_drop(F(q.x1), F(q.f), F(L.X[l + 1]), (long)T * D, dp, seed, sid + 3, st))) return rc;
        }
        return mask_rows(F(L.X[cfg.n_layers]), tok_row, F(L.HID), B, S, D, st);
    }
    for (int l = 0; l < cfg.n_layers; ++l) {
        const HLayer& W = layers[l];
        const HLayBuf& q = L.lay[l];
        const uint32_t sid = 0x6000u + l * 8;
        if ((rc = linear_fwd(F(L.X[l]), D, data[W.win], data[W.bin], F(q.qkv), 3 * D, T, 3 * D, D, st))) return rc;
        AttnArgs a{F(q.qkv), tok_row, F(q.probs), F(q.ctx), B, S, H, hd, dp, seed, sid};
        if ((rc = attn_fwd(a, st))) return rc;
        if ((rc = linear_fwd(F(q.ctx), D, data[W.wo], data[W.bo], F(q.ao), D, T, D, D, st))) return rc;
        AddLnArgs n1{F(L.X[l]), F(q.ao), data[W.g1], data[W.be1], F(q.x1), F(q.xh1), F(q.rstd1), T, D, kEps, dp, seed, sid + 1};
        if ((rc = add_ln_fwd(n1, st))) return rc;
        if ((rc = linear_fwd(F(q.x1), D, data[W.w1], data[W.b1], F(q.hpre), D, T, D, D, st))) return rc;
        if ((rc = act_fwd(F(q.hpre), F(q.hact), (long)T * D, cfg.gelu, dp, seed, sid + 2, st))) return rc;
        if ((rc = linear_fwd(F(q.hact), D, data[W.w2], data[W.b2], F(q.f), D, T, D, D, st))) return rc;
        AddLnArgs n2{F(q.x1), F(q.f), data[W.g2], data[W.be2], F(L.X[l + 1]), F(q.xh2), F(q.rstd2), T, D, kEps, dp, seed, sid + 3};
        if ((rc = add_ln_fwd(n2, st))) return rc;
    }
    return mask_rows(F(L.X[cfg.n_layers]), tok_row, F(L.HID), B, S, D, st);
}

// ---- stage 3: event decoder on token 0, prong decoder on tokens 1..P of L.HID -------------------------------------------
int HeadPlan::decode(int B, int P, float* ev_logits, float* pr_logits, char* ws, const HLayout& L, int train, uint64_t seed,
                     hipStream_t st) {
    const int D = cfg.hidden_dim, TP = P * B;
    const float dp = train ? cfg.dropout : 0.f;
    auto F = [&](long off) { return reinterpret_cast<float*>(ws + off); };
    int rc;
    if (ev_logits && (rc = linear_fwd(F(L.HID), D, data[ew], data[eb], ev_logits, cfg.event_classes, B, cfg.event_classes, D, st))) return rc;
    if (!pr_logits || TP == 0) return 0;
    const float* in = F(L.HID) + (long)B * D;
    int inw = D;
    for (size_t i = 0; i < dec.size(); ++i) {
        const HDec& d = dec[i];
        if ((rc = linear_fwd(in, inw, data[d.w], data[d.b], F(L.Zd[i]), d.out, TP, d.out, d.in, st))) return rc;
        RowsBnArgs r{};
        r.X = F(L.Zd[i]); r.ldx = d.out; r.R = TP; r.C = d.out; r.slope = d.a >= 0 ? data[d.a] : nullptr; r.no_norm = cfg.no_linear_bn;
        if (!r.no_norm) { r.gamma = data[d.n.w]; r.beta = data[d.n.b]; r.running_mean = data[d.n.rm]; r.running_var = data[d.n.rv]; }
        r.Y = F(L.Ad[i]); r.ldy = d.out;
        r.save_mean = F(L.dstat[i]); r.save_rstd = F(L.dstat[i]) + d.out; r.train = train; r.eps = kEps; r.momentum = kMom;
        r.drop_p = cfg.dropout_modules ? dp : 0.f; r.seed = seed; r.stream_id = 0x7000u + (uint32_t)i;
        if ((rc = rows_bn_fwd(r, st))) return rc;
        in = F(L.Ad[i]); inw = d.out;
    }
    if ((rc = linear_fwd(in, inw, data[ow], data[ob], F(L.LG), cfg.prong_classes, TP, cfg.prong_classes, cfg.dec_out_in, st))) return rc;
    return permute_rows(F(L.LG), pr_logits, B, P, cfg.prong_classes, 1, st);
}

int HeadPlan::check(int B, int P, int nP, long ws_bytes, HLayout& L) const {
    if (!bound) return -11;
    if (cfg.dec_out_in != dec_width) { fprintf(stderr, "tcvn: prong decoder width mismatch (reference would fail too)\n"); return -21; }
    if (B <= 0 || P < 0 || nP < 0 || 1 + P > 64) return -1;          // attention kernel: sequences up to 64 tokens
    layout(B, P, nP, L);
    return ws_bytes < L.total ? -12 : 0;
}

int HeadPlan::forward(int B, int P, int nP, const float* rows, const int32_t* tok_row, float* ev_logits, float* pr_logits,
                      char* ws, long ws_bytes, int train, uint64_t seed, hipStream_t st) {
    HLayout L;
    int rc;
    if ((rc = check(B, P, nP, ws_bytes, L))) return rc;
    if ((rc = embed(B, P, nP, rows, tok_row, ws, L, train, seed, st))) return rc;
    if ((rc = encode(B, P, tok_row, ws, L, train, seed, st))) return rc;
    if ((rc = decode(B, P, ev_logits, pr_logits, ws, L, train, seed, st))) return rc;
    last_seed = seed; last_train = train;
    return 0;
}

int HeadPlan::loss(int B, int P, const float* ev_logits, const float* pr_logits, const int64_t* et, const int8_t* pt, float* losses,
                   float* accs, float* dEv, float* dPr, hipStream_t st) {
    int rc;
    float* lb = accs + 2;                       // accs has room for 2 + 4 floats (scratch behind the two accuracies)
    if ((rc = focal_i64(ev_logits, et, B, cfg.event_classes, cfg.gamma, cfg.event_weight, dEv, lb, st))) return rc;
    if ((rc = focal_i8(pr_logits, pt, B * P, cfg.prong_classes, cfg.gamma, 1.f - cfg.event_weight, dPr, lb + 2, st))) return rc;
    hipLaunchKernelGGL(k_combine_loss, dim3(1), dim3(1), 0, st, lb, lb + 2, cfg.event_weight, losses, accs);
    TCVN_LAUNCH_CHECK();
    return 0;
}

int HeadPlan::backward(int B, int P, int nP, const float* rows, const int32_t* tok_row, const float* dEv, const float* dPr,
                       float* d_rows, char* ws, long ws_bytes, hipStream_t st) {
    if (!bound) return -11;
    for (size_t i = 0; i < slots.size(); ++i)
        if (slots[i].kind == TCVN_SLOT_PARAM && grad[i] == nullptr) return -14;
    HLayout L;
    layout(B, P, nP, L);
    if (ws_bytes < L.total) return -12;
    const int D = cfg.hidden_dim, S = 1 + P, T = S * B, R = B + nP, TP = P * B, H = cfg.heads, hd = D / H;
    const int Ce = cfg.event_classes, Cp = cfg.prong_classes;
    const float dp = cfg.dropout;
    const uint64_t seed = last_seed;
    auto F = [&](long off) { return reinterpret_cast<float*>(ws + off); };
    int rc;
    // ---- prong decoder ----
    if ((rc = permute_rows(dPr, F(L.dLG), B, P, Cp, 0, st))) return rc;
    float* dHID = F(L.dHID);
    const float* last_in = dec.empty() ? F(L.HID) + (long)B * D : F(L.Ad[dec.size() - 1]);
    const int last_w = dec.empty() ? D : dec.back().out;
    if ((rc = linear_bwd_dw(F(L.dLG), Cp, last_in, last_w, grad[ow], grad[ob], TP, Cp, last_w, st))) return rc;
    float* dA = F(L.t0);
    float* dZ = F(L.t1);
    float* dst0 = dec.empty() ? dHID + (long)B * D : dA;
    if ((rc = linear_bwd_dx(F(L.dLG), Cp, data[ow], dst0, dec.empty() ? D : last_w, TP, Cp, last_w, 0, st))) return rc;
    for (int i = (int)dec.size() - 1; i >= 0; --i) {
        const HDec& d = dec[i];
        RowsBnBwdArgs r{};
        r.X = F(L.Zd[i]); r.ldx = d.out; r.dY = dA; r.lddy = d.out; r.R = TP; r.C = d.out;
        r.no_norm = cfg.no_linear_bn; r.slope = d.a >= 0 ? data[d.a] : nullptr; r.dslope = d.a >= 0 ? grad[d.a] : nullptr;
        if (!r.no_norm) { r.gamma = data[d.n.w]; r.beta = data[d.n.b]; r.dgamma = grad[d.n.w]; r.dbeta = grad[d.n.b]; }
        r.save_mean = F(L.dstat[i]); r.save_rstd = F(L.dstat[i]) + d.out;
        r.dX = dZ; r.lddx = d.out;
        r.drop_p = cfg.dropout_modules ? dp : 0.f; r.seed = seed; r.stream_id = 0x7000u + (uint32_t)i;
        if ((rc = rows_bn_bwd(r, st))) return rc;
        const float* in = i == 0 ? F(L.HID) + (long)B * D : F(L.Ad[i - 1]);
        const int inw = i == 0 ? D : dec[i - 1].out;
        if ((rc = linear_bwd_dw(dZ, d.out, in, inw, grad[d.w], grad[d.b], TP, d.out, d.in, st))) return rc;
        float* dIn = i == 0 ? dHID + (long)B * D : dA;
        if ((rc = linear_bwd_dx(dZ, d.out, data[d.w], dIn, inw, TP, d.out, d.in, 0, st))) return rc;
    }
    // ---- event decoder ----
    if ((rc = linear_bwd_dw(dEv, Ce, F(L.HID), D, grad[ew], grad[eb], B, Ce, D, st))) return rc;
    if ((rc = linear_bwd_dx(dEv, Ce, data[ew], dHID, D, B, Ce, D, 0, st))) return rc;
    // ---- encoder ----
    float* dX = F(L.t2);
    if ((rc = mask_rows(dHID, tok_row, dX, B, S, D, st))) return rc;
    if (fused_encoder && encoder_fused_ok(S, D, H, cfg.n_layers, cfg.norm_first)) {
        // chain kernel (one workgroup per event) + grouped weight-gradient launch; dX is rewritten in place (t3 -> t2 not needed)
        EncFusedBwdArgs a{};
        a.dY = dX; a.dX = F(L.t3); a.lnp = F(L.lnp); a.B = B; a.S = S; a.H = H; a.L = cfg.n_layers; a.gelu = cfg.gelu;
        a.drop_p = dp; a.seed = seed;
        EncWgradArgs w{};
        w.T = T; w.B = B; w.L = cfg.n_layers; w.lnp = F(L.lnp);
        int nj = 0, nt = 0;
        for (int l = 0; l < cfg.n_layers; ++l) {
            const HLayer& W = layers[l];
            const HLayBuf& q = L.lay[l];
            a.w[l] = EncLayerW{data[W.win], data[W.bin], data[W.wo], data[W.bo], data[W.w1], data[W.b1], data[W.w2], data[W.b2],
                               data[W.g1], data[W.be1], data[W.g2], data[W.be2]};
            a.buf[l] = EncLayerBuf{F(q.qkv), F(q.probs), F(q.ctx), F(q.xh1), F(q.rstd1), F(q.x1), F(q.hpre), F(q.hact), F(q.xh2),
                                   F(q.rstd2), F(L.X[l + 1])};
            a.g[l] = EncLayerGrad{F(q.g_dqkv), F(q.g_dao), F(q.g_dhp), F(q.g_df)};
            w.job[nj++] = EncWgradJob{F(q.g_dqkv), 3 * D, F(L.X[l]), grad[W.win], grad[W.bin], 3 * D / 32};
            w.job[nj++] = EncWgradJob{F(q.g_dao), D, F(q.ctx), grad[W.wo], grad[W.bo], D / 32};
            w.job[nj++] = EncWgradJob{F(q.g_dhp), D, F(q.x1), grad[W.w1], grad[W.b1], D / 32};
            w.job[nj++] = EncWgradJob{F(q.g_df), D, F(q.hact), grad[W.w2], grad[W.b2], D / 32};
            nt += 3 * D / 32 + 3 * (D / 32);
            w.ln_dst[l][0] = grad[W.g1]; w.ln_dst[l][1] = grad[W.be1]; w.ln_dst[l][2] = grad[W.g2]; w.ln_dst[l][3] = grad[W.be2];
        }
        w.n_jobs = nj; w.n_tiles = nt;
        if ((rc = encoder_fused_bwd(a, w, st))) return rc;
        dX = F(L.t3);
        goto combined;
    }
    {
    float* d1 = F(L.t3);
    float* dR = F(L.t0);
    float* dT = F(L.t1);
    if (cfg.norm_first) {
        // dX = d x_{l+1}.  d x1 = dX + LN2'(d h2) ; d x_l = d x1 + LN1'(d h1); the dropped branches carry drop * gradient.
        float* dS = F(L.dHID);                                   // LayerNorm input gradients (dHID is consumed by now)
        for (int l = cfg.n_layers - 1; l >= 0; --l) {
            const HLayer& W = layers[l];
            const HLayBuf& q = L.lay[l];
            const uint32_t sid = 0x6000u + l * 8;
            const long n = (long)T * D;
            if ((rc = mul_drop(dX, dR, n, dp, seed, sid + 3, st))) return rc;                               // dR = d f
            if ((rc = linear_bwd_dw(dR, D, F(q.hact), D, grad[W.w2], grad[W.b2], T, D, D, st))) return rc;
            if ((rc = linear_bwd_dx(dR, D, data[W.w2], dT, D, T, D, D, 0, st))) return rc;                  // dT = d hact
            if ((rc = act_bwd(F(q.hpre), dT, dR, n, cfg.gelu, dp, seed, sid + 2, st))) return rc;           // dR = d hpre
            if ((rc = linear_bwd_dw(dR, D, F(q.h2), D, grad[W.w1], grad[W.b1], T, D, D, st))) return rc;
            if ((rc = linear_bwd_dx(dR, D, data[W.w1], dT, D, T, D, D, 0, st))) return rc;                  // dT = d h2
            AddLnBwdArgs n2{dT, F(q.xh2), F(q.rstd2), data[W.g2], dS, d1, grad[W.g2], grad[W.be2], T, D, 0.f, seed, sid + 3};
            if ((rc = add_ln_bwd(n2, st))) return rc;                                                       // dS = LN2 input gradient
            if ((rc = add_inplace(dX, dS, n, st))) return rc;                                               // dX = d x1
            if ((rc = mul_drop(dX, dR, n, dp, seed, sid + 1, st))) return rc;                               // dR = d ao
            if ((rc = linear_bwd_dw(dR, D, F(q.ctx), D, grad[W.wo], grad[W.bo], T, D, D, st))) return rc;
            if ((rc = linear_bwd_dx(dR, D, data[W.wo], dT, D, T, D, D, 0, st))) return rc;                  // dT = d ctx
            AttnBwdArgs ab{F(q.qkv), F(q.probs), dT, dR, B, S, H, hd, dp, seed, sid};                       // dR = d qkv [T, 3D]
            if ((rc = attn_bwd(ab, st))) return rc;
            if ((rc = linear_bwd_dw(dR, 3 * D, F(q.h1), D, grad[W.win], grad[W.bin], T, 3 * D, D, st))) return rc;
            if ((rc = linear_bwd_dx(dR, 3 * D, data[W.win], dT, D, T, 3 * D, D, 0, st))) return rc;         // dT = d h1
            AddLnBwdArgs n1{dT, F(q.xh1), F(q.rstd1), data[W.g1], dS, d1, grad[W.g1], grad[W.be1], T, D, 0.f, seed, sid + 1};
            if ((rc = add_ln_bwd(n1, st))) return rc;
            if ((rc = add_inplace(dX, dS, n, st))) return rc;                                               // dX = d x_l
        }
    } else
    for (int l = cfg.n_layers - 1; l >= 0; --l) {
        const HLayer& W = layers[l];
        const HLayBuf& q = L.lay[l];
        const uint32_t sid = 0x6000u + l * 8;
        AddLnBwdArgs n2{dX, F(q.xh2), F(q.rstd2), data[W.g2], d1, dR, grad[W.g2], grad[W.be2], T, D, dp, seed, sid + 3};
        if ((rc = add_ln_bwd(n2, st))) return rc;                                   // d1 = d(x1) residual, dR = d(f)
        if ((rc = linear_bwd_dw(dR, D, F(q.hact), D, grad[W.w2], grad[W.b2], T, D, D, st))) return rc;
        if ((rc = linear_bwd_dx(dR, D, data[W.w2], dT, D, T, D, D, 0, st))) return rc;      // dT = d(hact)
        if ((rc = act_bwd(F(q.hpre), dT, dR, (long)T * D, cfg.gelu, dp, seed, sid + 2, st))) return rc;   // dR = d(hpre)
        if ((rc = linear_bwd_dw(dR, D, F(q.x1), D, grad[W.w1], grad[W.b1], T, D, D, st))) return rc;
        if ((rc = linear_bwd_dx(dR, D, data[W.w1], d1, D, T, D, D, 1, st))) return rc;      // d1 += through FFN
        AddLnBwdArgs n1{d1, F(q.xh1), F(q.rstd1), data[W.g1], dX, dR, grad[W.g1], grad[W.be1], T, D, dp, seed, sid + 1};
        if ((rc = add_ln_bwd(n1, st))) return rc;                                   // dX = d(x_l) residual, dR = d(ao)
        if ((rc = linear_bwd_dw(dR, D, F(q.ctx), D, grad[W.wo], grad[W.bo], T, D, D, st))) return rc;
        if ((rc = linear_bwd_dx(dR, D, data[W.wo], dT, D, T, D, D, 0, st))) return rc;      // dT = d(ctx)
        AttnBwdArgs ab{F(q.qkv), F(q.probs), dT, dR, B, S, H, hd, dp, seed, sid};       // dR = d(qkv) [T, 3D]
        if ((rc = attn_bwd(ab, st))) return rc;
        if ((rc = linear_bwd_dw(dR, 3 * D, F(L.X[l]), D, grad[W.win], grad[W.bin], T, 3 * D, D, st))) return rc;
        if ((rc = linear_bwd_dx(dR, 3 * D, data[W.win], dX, D, T, 3 * D, D, 1, st))) return rc;
    }
    }
combined:
    // ---- combined embedding ----
    if ((rc = scatter_tokens_bwd(dX, tok_row, F(L.dC), B, S, D, st))) return rc;
    {
        RowsBnBwdArgs r{};
        r.X = F(L.Zc); r.ldx = D; r.dY = F(L.dC); r.lddy = D; r.R = R; r.C = D;
        r.no_norm = cfg.no_linear_bn; r.slope = ca >= 0 ? data[ca] : nullptr; r.dslope = ca >= 0 ? grad[ca] : nullptr;
        if (!r.no_norm) { r.gamma = data[cn.w]; r.beta = data[cn.b]; r.dgamma = grad[cn.w]; r.dbeta = grad[cn.b]; }
        r.save_mean = F(L.cstat); r.save_rstd = F(L.cstat) + D;
        r.dX = F(L.dZc); r.lddx = D;
        r.drop_p = dp; r.seed = seed; r.stream_id = 0x5000u;
        if ((rc = rows_bn_bwd(r, st))) return rc;
    }
    if ((rc = linear_bwd_dw(F(L.dZc), D, rows, cfg.in_dim, grad[cw], cb >= 0 ? grad[cb] : nullptr, R, D, cfg.in_dim, st))) return rc;
    return linear_bwd_dx(F(L.dZc), D, data[cw], d_rows, cfg.in_dim, R, D, cfg.in_dim, 0, st);
}

// ---------------------------------------------------------------------------------------------------------------------
struct tcvn_head { HeadPlan plan; int last_np = 0; explicit tcvn_head(const tcvn_head_cfg& c) : plan(c) {} };

extern "C" {
int tcvn_head_create(const tcvn_head_cfg* cfg, tcvn_head** out) {
    if (!cfg || !out || cfg->n_dec < 0 || cfg->n_dec > 8 || cfg->hidden_dim % cfg->heads != 0) return -1;
    *out = new tcvn_head(*cfg);
    return 0;
}
void tcvn_head_destroy(tcvn_head* p) { delete p; }
int tcvn_head_num_slots(const tcvn_head* p) { return (int)p->plan.slots.size(); }
int tcvn_head_slot(const tcvn_head* p, int i, char* name, int cap, int64_t* numel, int* kind) {
    if (i < 0 || i >= (int)p->plan.slots.size()) return -1;
    const auto& s = p->plan.slots[i];
    if (name && cap > 0) { strncpy(name, s.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (numel) *numel = s.numel;
    if (kind) *kind = s.kind;
    return 0;
}
int tcvn_head_bind(tcvn_head* p, void* const* data, void* const* grad) { return p->plan.bind(data, grad); }
void tcvn_head_set_fused_encoder(tcvn_head* p, int on) { p->plan.fused_encoder = on != 0; }
int64_t tcvn_head_workspace_bytes(const tcvn_head* p, int batch, int max_prongs, int n_prongs) {
    HLayout L;
    p->plan.layout(batch, max_prongs, n_prongs, L);
    return L.total;
}
int tcvn_head_forward(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row,
                      float* event_logits, float* prong_logits, void* ws, int64_t ws_bytes, int train, uint64_t seed, void* stream) {
    p->last_np = n_prongs;
    return p->plan.forward(batch, max_prongs, n_prongs, rows, tok_row, event_logits, prong_logits, reinterpret_cast<char*>(ws),
                           ws_bytes, train, seed, reinterpret_cast<hipStream_t>(stream));
}
/* stage entry points (forward only): see include/tcvn_hip.h */
int tcvn_head_embed(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row, float* tokens,
                    void* ws, int64_t ws_bytes, int train, uint64_t seed, void* stream) {
    HLayout L;
    int rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((rc = p->plan.check(batch, max_prongs, n_prongs, ws_bytes, L))) return rc;
    char* w = reinterpret_cast<char*>(ws);
    if ((rc = p->plan.embed(batch, max_prongs, n_prongs, rows, tok_row, w, L, train, seed, st))) return rc;
    return permute_rows(reinterpret_cast<float*>(w + L.X[0]), tokens, batch, 1 + max_prongs, p->plan.cfg.hidden_dim, 1, st);
}
int tcvn_head_encode(tcvn_head* p, int batch, int max_prongs, const float* tokens, const int32_t* tok_row, float* hidden, void* ws,
                     int64_t ws_bytes, int train, uint64_t seed, void* stream) {
    HLayout L;
    int rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((rc = p->plan.check(batch, max_prongs, 0, ws_bytes, L))) return rc;
    char* w = reinterpret_cast<char*>(ws);
    const int D = p->plan.cfg.hidden_dim, S = 1 + max_prongs;
    float* X0 = reinterpret_cast<float*>(w + L.X[0]);
    if ((rc = permute_rows(tokens, X0, batch, S, D, 0, st))) return rc;            // [B,S,D] -> sequence-major rows
    if ((rc = mask_rows(X0, tok_row, X0, batch, S, D, st))) return rc;             // embeddings * sequence_mask (:69)
    if ((rc = p->plan.encode(batch, max_prongs, tok_row, w, L, train, seed, st))) return rc;
    TCVN_CHECK(hipMemcpyAsync(hidden, w + L.HID, (size_t)S * batch * D * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}
int tcvn_head_decode(tcvn_head* p, int batch, int max_prongs, const float* hidden, float* event_logits, float* prong_logits,
                     void* ws, int64_t ws_bytes, int train, uint64_t seed, void* stream) {
    HLayout L;
    int rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((rc = p->plan.check(batch, max_prongs, 0, ws_bytes, L))) return rc;
    char* w = reinterpret_cast<char*>(ws);
    const int D = p->plan.cfg.hidden_dim, S = 1 + max_prongs;
    TCVN_CHECK(hipMemcpyAsync(w + L.HID, hidden, (size_t)S * batch * D * 4, hipMemcpyDeviceToDevice, st));
    return p->plan.decode(batch, max_prongs, event_logits, prong_logits, w, L, train, seed, st);
}

int tcvn_head_loss(tcvn_head* p, int batch, int max_prongs, const float* event_logits, const float* prong_logits,
                   const int64_t* event_targets, const int8_t* prong_targets, float* losses, float* accs, float* d_event_logits,
                   float* d_prong_logits, void* stream) {
    return p->plan.loss(batch, max_prongs, event_logits, prong_logits, event_targets, prong_targets, losses, accs, d_event_logits,
                        d_prong_logits, reinterpret_cast<hipStream_t>(stream));
}
int tcvn_head_backward(tcvn_head* p, int batch, int max_prongs, int n_prongs, const float* rows, const int32_t* tok_row,
                       const float* d_event_logits, const float* d_prong_logits, float* d_rows, void* ws, int64_t ws_bytes,
                       void* stream) {
    return p->plan.backward(batch, max_prongs, n_prongs, rows, tok_row, d_event_logits, d_prong_logits, d_rows,
                            reinterpret_cast<char*>(ws), ws_bytes, reinterpret_cast<hipStream_t>(stream));
}
}

// Stand-alone row operators behind the holder modules' own forward() (LinearBlock, ProngDecoder, ProngTargetDecoder,
// DenseNet.output_block): forward only, fp32, caller-owned tensors.
extern "C" int tcvn_linear_forward(const float* x, int64_t ldx, const float* weight, const float* bias, float* y, int64_t ldy,
                                   int rows, int n_out, int n_in, void* stream) {
    if (!x || !weight || !y || rows < 0 || n_out <= 0 || n_in <= 0) return -1;
    if (rows == 0) return 0;
    return linear_fwd(x, ldx, weight, bias, y, ldy, rows, n_out, n_in, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int tcvn_rows_bn_prelu_forward(const float* x, int64_t ldx, int rows, int channels, const float* gamma, const float* beta,
                                          const float* slope, float* running_mean, float* running_var, float* y, int64_t ldy,
                                          float* save_mean_rstd, int train, float drop_p, uint64_t seed, uint32_t stream_id,
                                          void* stream) {
    const bool no_norm = !gamma && !beta && !running_mean && !running_var;      // LinearBlock without BatchNorm1d (norm = Identity)
    if (!x || !y || rows <= 0 || channels <= 0) return -1;
    if (!no_norm && (!gamma || !beta || !running_mean || !running_var || !save_mean_rstd)) return -1;
    RowsBnArgs r{};
    r.no_norm = no_norm ? 1 : 0;
    r.X = x; r.ldx = ldx; r.R = rows; r.C = channels; r.gamma = gamma; r.beta = beta; r.slope = slope;      // slope NULL: ReLU
    r.running_mean = running_mean; r.running_var = running_var; r.Y = y; r.ldy = ldy;
    r.save_mean = save_mean_rstd; r.save_rstd = save_mean_rstd ? save_mean_rstd + channels : nullptr; r.train = train; r.eps = kEps; r.momentum = kMom;
    r.drop_p = train ? drop_p : 0.f; r.seed = seed; r.stream_id = stream_id;
    return rows_bn_fwd(r, reinterpret_cast<hipStream_t>(stream));
}

// Backward of the two row operators (smart-feature MLP, layers/prong_feature_embedding.py:36-78: the only LinearBlocks outside the head plan)
extern "C" int tcvn_linear_backward(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* weight, float* dx, int64_t lddx,
                                    float* dweight, float* dbias, int rows, int n_out, int n_in, void* stream) {
    if (!dy || !x || !weight || rows <= 0 || n_out <= 0 || n_in <= 0) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (dweight && (rc = linear_bwd_dw(dy, lddy, x, ldx, dweight, dbias, rows, n_out, n_in, st))) return rc;
    if (dx && (rc = linear_bwd_dx(dy, lddy, weight, dx, lddx, rows, n_out, n_in, 0, st))) return rc;
    return 0;
}
extern "C" int tcvn_rows_bn_prelu_backward(const float* x, int64_t ldx, const float* dy, int64_t lddy, int rows, int channels,
                                           const float* gamma, const float* beta, const float* slope, const float* save_mean_rstd,
                                           float* dx, int64_t lddx, float* dgamma, float* dbeta, float* dslope, float drop_p,
                                           uint64_t seed, uint32_t stream_id, void* stream) {
    const bool no_norm = !gamma && !beta && !dgamma && !dbeta;                  // LinearBlock without BatchNorm1d
    if (!x || !dy || !dx || rows <= 0 || channels <= 0 || (slope != nullptr) != (dslope != nullptr)) return -1;
    if (!no_norm && (!gamma || !beta || !save_mean_rstd || !dgamma || !dbeta)) return -1;
    RowsBnBwdArgs r{};
    r.no_norm = no_norm ? 1 : 0;
    r.X = x; r.ldx = ldx; r.dY = dy; r.lddy = lddy; r.R = rows; r.C = channels; r.gamma = gamma; r.beta = beta; r.slope = slope;
    r.save_mean = save_mean_rstd; r.save_rstd = save_mean_rstd ? save_mean_rstd + channels : nullptr; r.dX = dx; r.lddx = lddx;
    r.dgamma = dgamma; r.dbeta = dbeta; r.dslope = dslope; r.drop_p = drop_p; r.seed = seed; r.stream_id = stream_id;
    return rows_bn_bwd(r, reinterpret_cast<hipStream_t>(stream));
}

// Stand-alone softmax focal loss of one logit matrix (reference: NeutrinoFullBaseTrainer.loss, :148-160)
extern "C" int tcvn_focal_loss(const float* logits, const int64_t* targets, int rows, int classes, float gamma, float weight,
                               float* d_logits, float* out2, void* stream) {
    return focal_i64(logits, targets, rows, classes, gamma, weight, d_logits, out2, reinterpret_cast<hipStream_t>(stream));
}
