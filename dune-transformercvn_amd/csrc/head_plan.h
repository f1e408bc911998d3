// Token-path plan: slot table, workspace layout, forward / loss / backward drivers.
#pragma once
#include <string>
#include <vector>

#include "../../include/tcvn_hip.h"
#include "tcvn_common.h"
#include "densenet_plan.h"   // Slot

namespace tcvn {

struct HBn { int w = -1, b = -1, rm = -1, rv = -1; };
struct HLayer { int win, bin, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2; };
struct HDec { int w, b, a = -1, in, out; HBn n; };      // a = -1: ReLU; n.w = -1: no BatchNorm1d
struct HLayBuf { long qkv, probs, ctx, ao, xh1, rstd1, x1, hpre, hact, f, xh2, rstd2, g_dqkv, g_dao, g_dhp, g_df,
                 h1, h2; };     // h1 / h2: LayerNorm outputs of the pre-norm variant (transformer_norm_first)
struct HLayout {
    long Zc, C, cstat, HID, LG, dEv, dPr, lossbuf, dLG, t0, t1, t2, t3, dHID, dC, dZc, lnp, total;
    std::vector<long> X, Zd, Ad, dstat;
    std::vector<HLayBuf> lay;
};

struct HeadPlan {
    tcvn_head_cfg cfg;
    std::vector<Slot> slots;
    std::vector<float*> data, grad;
    int cw, cb = -1, ca = -1, ew, eb, ow, ob, dec_width;      // cb: combined linear bias (only without BatchNorm1d); ca: PReLU slope (-1: ReLU)
    HBn cn;
    std::vector<HLayer> layers;
    std::vector<HDec> dec;
    bool bound = false;
    bool fused_encoder = true;     // encoder_fused.hip when the shape allows (tcvn_head_set_fused_encoder(p, 0): unfused kernels, for A/B tests)
    uint64_t last_seed = 0; int last_train = 0;

    explicit HeadPlan(const tcvn_head_cfg& c);
    int add_slot(const std::string& name, long numel, int kind);
    HBn add_bn(const std::string& p, int c);
    int bind(void* const* d, void* const* g);
    void layout(int B, int P, int nP, HLayout& L) const;
    int check(int B, int P, int nP, long ws_bytes, HLayout& L) const;
    int embed(int B, int P, int nP, const float* rows, const int32_t* tok_row, char* ws, const HLayout& L, int train, uint64_t seed,
              hipStream_t st);
    int encode(int B, int P, const int32_t* tok_row, char* ws, const HLayout& L, int train, uint64_t seed, hipStream_t st);
    int decode(int B, int P, float* ev_logits, float* pr_logits, char* ws, const HLayout& L, int train, uint64_t seed, hipStream_t st);
    int forward(int B, int P, int nP, const float* rows, const int32_t* tok_row, float* ev_logits, float* pr_logits, char* ws,
                long ws_bytes, int train, uint64_t seed, hipStream_t st);
    int loss(int B, int P, const float* ev_logits, const float* pr_logits, const int64_t* et, const int8_t* pt, float* losses,
             float* accs, float* dEv, float* dPr, hipStream_t st);
    int backward(int B, int P, int nP, const float* rows, const int32_t* tok_row, const float* dEv, const float* dPr, float* d_rows,
                 char* ws, long ws_bytes, hipStream_t st);
};

}  // namespace tcvn
