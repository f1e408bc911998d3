// DenseNet backward driver: reverse schedule of densenet.hip::forward on the same workspace.
// Extra HBM state: G[b] (gradient accumulators shaped like the concat buffers), (P,Q) per channel, one bottleneck-sized
// scratch DU, a conv0-sized scratch DU0, fp32 kernel-layout weight gradients (converted to OIHW at the end).
#include <cstring>
#include <vector>

#include <cstdlib>
#include "densenet_plan.h"
#include "tcvn_ops.h"
#include "tcvn_rows.h"

using namespace tcvn;

// tcvn_backward_overlap(): the 3x3 weight gradients on a side stream beside the data-gradient chain.  ON by default in rounds 2-3 (26.0-26.1 ms
// against 26.4 ms when the side stream also carried the 1x1 TN GEMMs).  OFF by default since round 4: with the fused 1x1 backward kernel on the
// main stream only the 3x3 weight gradient is left to overlap, and the same-box A/B reads 19.55 ms/step serial against 19.55-19.60 with the side
// stream (19.8 when the fused kernel's slab reductions also went there) -- nothing to gain, and serial kernel timings are what profiles show.
static int g_backward_overlap = 0;
bool tcvn::backward_overlap_enabled() { return g_backward_overlap != 0; }
void tcvn::set_backward_overlap(int on) { g_backward_overlap = on; }

namespace {
constexpr float kEps = 1e-5f;
constexpr long kSlabBytes = 48L << 20;      // per-workgroup partial weight gradients (<= 256 x 147 KB) and column sums
constexpr long kSlab1GemmBytes = 64L << 20; // fused 1x1 backward: 256 workgroups x [128][512] fp32 (bwd1x1_wide.hip), bias column sums behind them
constexpr long kSlab1Bytes = 68L << 20;
constexpr long kSlabGemmBytes = 44L << 20;  // GEMM slabs; the tail [44 MB, 48 MB) holds the bias column-sum partials (<= 1024 x 512 floats)
struct Bump {
    long off;
    long take(long bytes) { long o = off; off += round_up(bytes, 256); return o; }
};
}  // namespace

void DenseNetPlan::layout_bwd(int n, long start, long maxY, Layout& L) const {
    Bump b{start};
    const int mid = cfg.bn_size * cfg.growth;
    L.G.clear(); L.pqD.clear();
    // --- zeroed at the start of every backward: G, pqD, gwk (contiguous) ---
    long bpart = (long)std::max(pool0_bwd_grid(n, Hc, Wc), pool0_bwd_vec_grid(n, Hc, Wc)) * cfg.init_ch * 24;
    bpart = std::max(bpart, 1024L * cfg.init_ch * 24);              // sparse stem backward: <= 1024 workgroups
    bpart = std::max(bpart, (long)head_pool_bwd_grid(n) * Cf * 24);
    for (const auto& bg : blocks) {
        const long M = (long)n * bg.H * bg.W;
        L.G.push_back(b.take(M * bg.ld * esz));
        bpart = std::max(bpart, 512L * std::max(mid, bg.Ctot) * 24);
    }
    for (const auto& bg : blocks) L.pqD.push_back(b.take((long)bg.ld * 8));
    long gw = 0;
    for (const auto& e : wk_list())
        if (!e.transpose && !e.frag) gw += round_up((long)e.N * e.Kp * 4, 256);
    L.gwk = b.take(gw);
    L.zero_end = b.off;
    L.du = b.take(maxY * esz);
    L.ey = b.take(maxY * esz);
    L.ey2 = b.take(maxY * esz);                    // second EY buffer: the weight-gradient stream may still read the previous one
    L.slab = b.take(kSlabBytes);
    L.slab1 = cfg.mode == MODE_BF16 ? b.take(kSlab1Bytes) : -1;
    L.pqY = b.take((long)mid * 8);
    L.du0 = b.take((long)n * Hc * Wc * cfg.init_ch * esz);
    L.pq0 = b.take((long)cfg.init_ch * 8);
    L.bpart = b.take(bpart);
    L.dF = b.take((long)n * Cf * 4);
    L.dZ = b.take((long)n * cfg.out_dim * 4);
    L.EY3.clear();                                 // per dense layer: the 3x3 output gradient's eff rows, data gradient -> weight gradient
    for (const auto& bg : blocks) {
        std::vector<long> eys;
        if (cfg.mode == MODE_BF16 && cfg.growth == 32)
            for (int l = 0; l < bg.L; ++l) eys.push_back(b.take((long)n * bg.H * bg.W * 32 * esz));
        L.EY3.push_back(eys);
    }
    L.total = b.off;
}

bool DenseNetPlan::bwd1x1_fill(int bi, int l, long M, char* ws, const Layout& L, Bwd1x1Args& fa) const {
    const BlockGeom& bg = blocks[bi];
    const LayerSlots& ls = bg.layers[l];
    const int mid = cfg.bn_size * cfg.growth;
    if (L.XA[bi][l] < 0 || mid != 128 || L.slab1 < 0 || cfg.mode != MODE_BF16) return false;
    float* tabs = reinterpret_cast<float*>(ws + L.tabs);
    const WkEntry& etf = wk_find(ls.w1, 1, 1);
    fa = Bwd1x1Args{};
    fa.DU = ws + L.du; fa.Y = ws + L.Y[bi][l]; fa.PY = reinterpret_cast<float*>(ws + L.pqY); fa.QY = fa.PY + mid; fa.M = M;
    fa.Xin = ws + L.D[bi]; fa.ldx = bg.ld; fa.cin = ls.cin;
    fa.sc = tabs + tab_off(ls.n1); fa.sh = fa.sc + round_up(ls.n1.C, 8); fa.sl = data[ls.a1]; fa.Gout = ws + L.G[bi]; fa.ldg = bg.ld;
    fa.Wfrag = ws + L.wk + etf.off; fa.Kp = etf.Kp; fa.zeros = ws + L.zeros; fa.part = reinterpret_cast<double*>(ws + L.bpart);
    fa.slab = reinterpret_cast<float*>(ws + L.slab1); fa.slab_bytes = kSlab1GemmBytes; fa.ldc = wk_find(ls.w1, 0).Kp;
    fa.tail = reinterpret_cast<float*>(ws + L.slab1 + kSlab1GemmBytes);
    fa.nblk = bwd1x1_fused_nblk(fa);
    return bwd1x1_fused_ok(fa);
}
bool DenseNetPlan::bwd1x1_fusable(int bi, int l, long M, char* ws, const Layout& L) const {
    static const bool no_fuse1 = TCVN_KNOB_SET("TCVN_NO_BWD1_FUSE");
    Bwd1x1Args fa;
    return !no_fuse1 && bwd1x1_fill(bi, l, M, ws, L, fa);
}

// Blocks [bi_lo, bi_hi] of the backward pass (bi_hi == last block: also the output block; bi_lo == 0: also the stem).  A caller that
// wants each block's parameter gradients as soon as they are final (data-parallel exchange overlapped with the rest of backward)
// walks the blocks from the last to the first; one call with (n_blocks - 1, 0) is the whole backward.
int DenseNetPlan::backward(int n, const float* d_out, long d_out_ld, char* ws, long ws_bytes, hipStream_t st, int bi_hi, int bi_lo) {
    if (!bound) return -11;
    if (bi_hi < 0) bi_hi = (int)blocks.size() - 1;
    if (bi_lo < 0 || bi_lo > bi_hi || bi_hi >= (int)blocks.size()) return -1;
    const bool first_part = bi_hi == (int)blocks.size() - 1, last_part = bi_lo == 0;
    if (n <= 0) return 0;
    if (n != last_n) { fprintf(stderr, "tcvn: densenet backward without matching forward\n"); return -13; }
    for (size_t i = 0; i < slots.size(); ++i)
        if (slots[i].kind == TCVN_SLOT_PARAM && grad[i] == nullptr) { fprintf(stderr, "tcvn: grad of %s unbound\n", slots[i].name.c_str()); return -14; }
    Layout L;
    layout(n, true, L);
    if (ws_bytes < L.total) return -12;
    int rc;
    const int mode = cfg.mode, g = cfg.growth, mid = cfg.bn_size * cfg.growth;
    const uint64_t seed = last_seed;
    // Weight gradients (3x3 and 1x1, with their slab reductions) do not feed the data-gradient chain: in bf16 mode they run on a
    // side stream beside it.  Shared state: the slab (side stream only between drains), EY (double buffered, released by
    // ev_done), the bias column-sum partials (two halves of the slab tail).  TCVN_BWD_SERIAL=1 keeps everything on `st`.
    static const bool serial_env = TCVN_KNOB_SET("TCVN_BWD_SERIAL");
    static const bool no_fuse1 = TCVN_KNOB_SET("TCVN_NO_BWD1_FUSE");      // validation build: the three-kernel 1x1 backward (eff copy, TN GEMM, NT GEMM)
    // Every width takes the fused kernel.  A/B on MI355X (B = 32 x 8 prongs, validation build): fused for cin <= 256 only 19.76 ms/step, <= 384
    // 19.39, all layers 19.16 -- although a launch with three or four 128-column slices (block 3: every slice re-reads DU / Y and rebuilds EY)
    // takes longer than k_eff_mat + k_gemm_nt did (105 against ~58 us at four slices), the step is shorter without the two extra launches
    // per layer on the main stream and the TN GEMM competing on the side stream.  TCVN_BWD1_MAXCIN (validation build) restores a limit.
    static const int fuse1_maxcin = TCVN_KNOB_INT("TCVN_BWD1_MAXCIN") > 0 ? TCVN_KNOB_INT("TCVN_BWD1_MAXCIN") : (1 << 30);
    const bool side_on = fast3x3 && !serial_env && backward_overlap_enabled();
    if (side_on && (rc = ensure_side())) return rc;
    int seq = 0;                                   // parity of the EY buffer / tail half; reset by drain()
    bool side_busy = false;
    auto drain = [&]() -> int {                    // `st` waits for everything enqueued on the side stream
        if (side_on && side_busy) {
            TCVN_CHECK(hipEventRecord(ev_drain, side_st));
            TCVN_CHECK(hipStreamWaitEvent(st, ev_drain, 0));
            side_busy = false;
        }
        seq = 0;
        return 0;
    };
    float* tabs = reinterpret_cast<float*>(ws + L.tabs);
    auto sc_of = [&](const BnSlots& s) { return tabs + tab_off(s); };
    auto sh_of = [&](const BnSlots& s) { return tabs + tab_off(s) + round_up(s.C, 8); };
    double* part = reinterpret_cast<double*>(ws + L.bpart);

    // kernel-layout gradient offsets follow wk_list order (non-transposed entries only)
    std::vector<long> gw_off(wk_cache.size(), -1);
    {
        long o = 0;
        for (size_t i = 0; i < wk_cache.size(); ++i)
            if (!wk_cache[i].transpose && !wk_cache[i].frag) { gw_off[i] = o; o += round_up((long)wk_cache[i].N * wk_cache[i].Kp * 4, 256); }
    }
    auto gw_of = [&](int slot) -> float* {
        for (size_t i = 0; i < wk_cache.size(); ++i)
            if (wk_cache[i].slot == slot && !wk_cache[i].transpose && !wk_cache[i].frag) return reinterpret_cast<float*>(ws + L.gwk + gw_off[i]);
        return nullptr;
    };
    // device table of unpack descriptors (depends on ws)
    if (undesc_ws != ws || undesc_total != L.total) {
        std::vector<UnpackDesc> ud;
        unpack_first.assign(blocks.size() + 1, 0);            // descriptors are in wk_list order: conv0, then block by block
        for (size_t i = 0; i < wk_cache.size(); ++i) {
            const WkEntry& e = wk_cache[i];
            if (e.transpose || e.frag) continue;
            for (size_t bi = 0; bi < blocks.size(); ++bi) {   // first descriptor of block bi = its first layer's conv1
                if (!blocks[bi].layers.empty() && e.slot == blocks[bi].layers[0].w1) unpack_first[bi] = (int)ud.size();
            }
            UnpackDesc d{reinterpret_cast<const float*>(ws + L.gwk + gw_off[i]), grad[e.slot], e.N, e.Cin, e.taps, e.Kp,
                         (fast3x3 && e.taps == 9) ? 1 : 0};
            ud.push_back(d);
        }
        n_unpack = (int)ud.size();
        unpack_first[blocks.size()] = n_unpack;
        if (!d_undesc) TCVN_CHECK(hipMalloc(&d_undesc, ud.size() * sizeof(UnpackDesc)));
        h_undesc.assign(reinterpret_cast<char*>(ud.data()), reinterpret_cast<char*>(ud.data()) + ud.size() * sizeof(UnpackDesc));
        TCVN_CHECK(hipMemcpyAsync(d_undesc, h_undesc.data(), h_undesc.size(), hipMemcpyHostToDevice, st));
        undesc_ws = ws; undesc_total = L.total;
    }

    if (first_part) {
    // Zeroed per backward: the (P, Q) tables and the kernel-layout weight gradients (one contiguous range behind the G buffers),
    // and those gradient buffers G[b] whose first contribution accumulates.  In the bf16 fast path the first writer of G[b] -- the
    // transition's pooled data gradient, or the head for the last block -- writes instead (g_write), which saves a 0.9 GB memset and
    // the read of it at B = 32 x 8 prongs; only the pixels no 2x2 window covers (odd map sizes) are zeroed explicitly.
    TCVN_CHECK(hipMemsetAsync(ws + L.pqD[0], 0, (size_t)(L.zero_end - L.pqD[0]), st));
    for (size_t bi = 0; bi + 1 < blocks.size(); ++bi) {
        const long bytes = (long)n * blocks[bi].H * blocks[bi].W * blocks[bi].ld * esz;
        if (L.XP[bi] >= 0 && mode == MODE_BF16) {
            if ((rc = zero_pool_remainder(ws + L.G[bi], blocks[bi].ld, n, blocks[bi].H, blocks[bi].W, blocks[bi + 1].H, blocks[bi + 1].W, st))) return rc;
        } else TCVN_CHECK(hipMemsetAsync(ws + L.G[bi], 0, (size_t)bytes, st));
    }

    }
    auto bwd_link = [&](const BnSlots& s, int nblk, const double* bstat, long count, float* P, float* Q, int acc, int a_slot) -> int {
        BnBwdLinkArgs a{part, nblk, s.C, bstat, count, kEps, data[s.w], grad[s.w], grad[s.b], grad[a_slot], P, Q, acc};
        return bn_bwd_link(a, st);
    };

    // ---- output block backward: Dropout - PReLU - BatchNorm1d - Linear ----
    float* F = reinterpret_cast<float*>(ws + L.F);
    float* Z = reinterpret_cast<float*>(ws + L.Z);
    float* dF = reinterpret_cast<float*>(ws + L.dF);
    float* dZ = reinterpret_cast<float*>(ws + L.dZ);
    float* hs = reinterpret_cast<float*>(ws + L.head_stat);
    if (first_part) {
        RowsBnBwdArgs r{};
        r.X = Z; r.ldx = cfg.out_dim; r.dY = d_out; r.lddy = d_out_ld; r.R = n; r.C = cfg.out_dim;
        r.gamma = data[nl.w]; r.beta = data[nl.b]; r.slope = data[s_al]; r.save_mean = hs; r.save_rstd = hs + cfg.out_dim;
        r.dX = dZ; r.lddx = cfg.out_dim; r.dgamma = grad[nl.w]; r.dbeta = grad[nl.b]; r.dslope = grad[s_al];
        r.drop_p = cfg.dropout; r.seed = seed; r.stream_id = 0x4000u;
        if ((rc = rows_bn_bwd(r, st))) return rc;
        if ((rc = linear_bwd_dw(dZ, cfg.out_dim, F, Cf, grad[s_wl], nullptr, n, cfg.out_dim, Cf, st))) return rc;
        if ((rc = linear_bwd_dx(dZ, cfg.out_dim, data[s_wl], dF, Cf, n, cfg.out_dim, Cf, 0, st))) return rc;
    }

    for (int bi = bi_hi; bi >= bi_lo; --bi) {
        const BlockGeom& bg = blocks[bi];
        const long M = (long)n * bg.H * bg.W;
        char* D = ws + L.D[bi];
        char* G = ws + L.G[bi];
        float* P = reinterpret_cast<float*>(ws + L.pqD[bi]);
        float* Q = P + bg.ld;
        const double* bstatD = reinterpret_cast<const double*>(ws + L.bstatD[bi]);

        if (!bg.has_trans) {
            // final_norm + global average
            HeadPoolBwdArgs a{mode, D, bg.ld, n, bg.H * bg.W, Cf, sc_of(nf), sh_of(nf), data[s_af], dF, G, bg.ld, part,
                              head_pool_bwd_grid(n)};
            if ((rc = head_pool_bwd(a, st))) return rc;
            if ((rc = bwd_link(nf, a.nblk, bstatD, M, P, Q, 1, s_af))) return rc;
        } else {
            // transition: BN - PReLU - (pool commuted) - 1x1 conv, output = first channels of block bi+1
            const BlockGeom& nb = blocks[bi + 1];
            const long Mn = (long)n * nb.H * nb.W;
            const int Nt = bg.Ctot / 2;
            float* Pn = reinterpret_cast<float*>(ws + L.pqD[bi + 1]);
            float* Qn = Pn + nb.ld;
            EffSrc e{ws + L.G[bi + 1], nb.ld, ws + L.D[bi + 1], nb.ld, 0, Nt, Pn, Qn, 0.f, 0, 0};
            const WkEntry& ef = wk_find(bg.tw, 0);
            const WkEntry& et = wk_find(bg.tw, 1);
            ConvWgradArgs w{};
            w.mode = mode; w.e = e; w.dWk = gw_of(bg.tw); w.dbias = grad[bg.tb];
            w.fa.mode = mode; w.fa.amode = A_1X1_POOL; w.fa.A = D; w.fa.lda = bg.ld; w.fa.M = (int)Mn; w.fa.N = Nt;
            w.fa.K = bg.Ctot; w.fa.Kp = ef.Kp; w.fa.C = bg.Ctot; w.fa.H = nb.H; w.fa.W = nb.W; w.fa.Hin = bg.H; w.fa.Win = bg.W;
            w.fa.sc = sc_of(bg.tn); w.fa.sh = sh_of(bg.tn); w.fa.sl = data[bg.ta];
            const int Nt8 = (int)round_up(Nt, 8);
            if ((rc = drain())) return rc;            // the transition uses EY and the slab on `st`
            const bool xp_bf16 = L.XP[bi] >= 0 && mode == MODE_BF16;
            if (L.XP[bi] >= 0 && mode == MODE_F32) {  // fp32: the materialised pooled activation of the forward is the weight gradient's operand
                w.fa.amode = A_1X1; w.fa.A = ws + L.XP[bi]; w.fa.lda = bg.ldp; w.fa.K = bg.ldp; w.fa.C = bg.ldp;
                w.fa.sc = nullptr; w.fa.sh = nullptr; w.fa.sl = nullptr;
                w.slab = reinterpret_cast<float*>(ws + L.slab); w.slab_bytes = kSlabBytes;      // split-K slabs of k_gemm_tn_f32 (the stream was drained above)
            }
            if (xp_bf16) {
                // materialise the output gradient once (+ bias gradient), then dW = ET^T x XP on the TN GEMM
                SlabJob bias_job{};
                EffMatArgs em{e, Mn, ws + L.ey, Nt8, grad[bg.tb], reinterpret_cast<float*>(ws + L.slab + kSlabGemmBytes), &bias_job};
                if ((rc = eff_materialize_bf16(em, st))) return rc;
                GemmTnArgs ga{ws + L.ey, Nt8, Nt8, ws + L.XP[bi], bg.ldp, bg.ldp, Mn, gw_of(bg.tw), ef.Kp, ws + L.zeros,
                              reinterpret_cast<float*>(ws + L.slab), kSlabGemmBytes, Nt, bias_job};
                if ((rc = gemm_tn_bf16(ga, "k_gemm_tn_bf16<transition>", st))) return rc;
            } else if ((rc = conv_wgrad(w, st))) return rc;
            if (xp_bf16) {
                const WkEntry& etf = wk_find(bg.tw, 1, 1);
                GemmNtArgs ga{};
                ga.epi = EPI_DGRAD_POOL; ga.A = ws + L.ey; ga.lda = Nt8; ga.K = Nt8; ga.M = Mn; ga.N = bg.Ctot;
                ga.Wfrag = ws + L.wk + etf.off; ga.Kp = etf.Kp; ga.zeros = ws + L.zeros;
                ga.Xin = D; ga.ldxin = bg.ld; ga.sc = sc_of(bg.tn); ga.sh = sh_of(bg.tn); ga.sl = data[bg.ta];
                ga.Gout = G; ga.ldgo = bg.ld; ga.g_write = 1; ga.H = nb.H; ga.W = nb.W; ga.Hin = bg.H; ga.Win = bg.W;
                ga.part = part; ga.nblk = gemm_nt_nblk(ga);
                if ((rc = gemm_nt_bf16(ga, "k_gemm_nt_bf16<dgradtrans>", st))) return rc;
                if ((rc = bwd_link(bg.tn, ga.nblk, bstatD, M, P, Q, 1, bg.ta))) return rc;
                goto layers;
            }
            ConvDgradArgs d{};
            d.mode = mode; d.dmode = DG_1X1_POOL; d.e = e; d.M = (int)Mn; d.N = bg.Ctot; d.Kp = et.Kp;
            d.H = nb.H; d.W = nb.W; d.Hin = bg.H; d.Win = bg.W; d.Wt = ws + L.wk + et.off;
            d.Xin = D; d.ldxin = bg.ld; d.sc = sc_of(bg.tn); d.sh = sh_of(bg.tn); d.sl = data[bg.ta];
            d.Gout = G; d.ldgo = bg.ld; d.accumulate = 1; d.part = part; d.nblk = conv_dgrad_nblk(d);
            if ((rc = conv_dgrad(d, st))) return rc;
            if ((rc = bwd_link(bg.tn, d.nblk, bstatD, M, P, Q, 1, bg.ta))) return rc;
        }

    layers:
        for (int l = bg.L - 1; l >= 0; --l) {
            const LayerSlots& ls = bg.layers[l];
            char* Y = ws + L.Y[bi][l];
            char* DU = ws + L.du;
            float* PY = reinterpret_cast<float*>(ws + L.pqY);
            float* QY = PY + mid;
            const uint32_t sid = (uint32_t)(bi * 64 + l + 1);
            // gradient of this layer's output slice D[:, cin:cin+g]
            EffSrc e2{G, bg.ld, D, bg.ld, ls.cin, g, P + ls.cin, Q + ls.cin, cfg.dropout, seed, sid};
            if (bi < (int)keep_valid.size() && l < (int)keep_valid[bi].size() && keep_valid[bi][l] && !L.KM[bi].empty())
                e2.keep = reinterpret_cast<const uint32_t*>(ws + L.KM[bi][l]);      // keep words of this layer's forward (else: the hash)
            bool ey_valid = false;
            {   // conv2 data gradient -> DU (= sc2 * dU2) + norm2 partials (+ the eff rows for the weight gradient below)
                const WkEntry& et = wk_find(ls.w2, 1);
                ConvDgradArgs d{};
                d.mode = mode; d.dmode = DG_3X3; d.e = e2; d.M = (int)M; d.N = mid; d.Kp = et.Kp; d.H = bg.H; d.W = bg.W;
                d.Wt = ws + L.wk + et.off; d.Xin = Y; d.ldxin = mid; d.sc = sc_of(ls.n2); d.sh = sh_of(ls.n2); d.sl = data[ls.a2];
                d.Gout = DU; d.ldgo = mid; d.accumulate = 0; d.part = part;
                d.Wfrag = wk_frag(ws, L, ls.w2, 1); d.zeros = ws + L.zeros;
                if (fast3x3 && !L.EY3[bi].empty()) { d.ey_out = ws + L.EY3[bi][l]; ey_valid = conv3x3_dgrad_writes_ey(d); if (!ey_valid) d.ey_out = nullptr; }
                d.nblk = conv_dgrad_nblk(d);
                if ((rc = conv_dgrad(d, st))) return rc;
                if ((rc = bwd_link(ls.n2, d.nblk, reinterpret_cast<const double*>(ws + L.bstatY[bi][l]), M, PY, QY, 0, ls.a2))) return rc;
            }
            // Fused 1x1 backward (round 4, bwd1x1_fused.hip): effective gradient formed in LDS, bias / data / weight gradient and the norm1
            // backward epilogue in one pass -- no EY in HBM, no read of the activated copy XA, one launch instead of three.  It writes G, so it
            // runs on `st`; its slabs are its own (L.slab belongs to the 3x3 weight gradient, which may be on the side stream).
            const bool xa_absent = bi < (int)xa_skipped.size() && l < (int)xa_skipped[bi].size() && xa_skipped[bi][l];
            Bwd1x1Args fa{};
            bool fuse1 = L.XA[bi][l] >= 0 && !no_fuse1 && mid == 128 && L.slab1 >= 0 && (ls.cin <= fuse1_maxcin || xa_absent);
            if (fuse1) fuse1 = bwd1x1_fill(bi, l, M, ws, L, fa);       // (the same function the forward asked before it dropped the activated copy)
            SlabJob w3jobs[2] = {};        // the 3x3 weight gradient's slab reductions, folded into the fused kernel's reduction launch (same stream only)
            bool w3_deferred = false;
            {   // conv2 (3x3) weight gradient: beside the rest of this layer's data-gradient chain
                const WkEntry& ef = wk_find(ls.w2, 0);
                ConvWgradArgs w{};
                w.mode = mode; w.e = e2; w.dWk = gw_of(ls.w2); w.dbias = grad[ls.b2];
                if (ey_valid) w.e.ey = ws + L.EY3[bi][l];
                w.fa.mode = mode; w.fa.amode = A_3X3; w.fa.A = Y; w.fa.lda = mid; w.fa.M = (int)M; w.fa.N = g; w.fa.K = 9 * mid;
                w.fa.Kp = ef.Kp; w.fa.C = mid; w.fa.H = bg.H; w.fa.W = bg.W;
                w.fa.sc = sc_of(ls.n2); w.fa.sh = sh_of(ls.n2); w.fa.sl = data[ls.a2];
                if (fast3x3) {
                    const bool fused = bi < (int)act_fused.size() && l < (int)act_fused[bi].size() && act_fused[bi][l];
                    w.nfast = 1; w.fa.Aact = fused ? Y : ws + L.YA[bi][l]; w.fa.act_fused = fused ? 1 : 0; w.fa.zeros = ws + L.zeros;
                    w.slab = reinterpret_cast<float*>(ws + L.slab); w.slab_bytes = kSlabBytes;
                }
                const bool par = side_on && L.XA[bi][l] >= 0;
                if (par) {                    // G slice, its (P, Q), the materialised YA and the eff rows are final: fork
                    TCVN_CHECK(hipEventRecord(ev_fork_a, st));
                    TCVN_CHECK(hipStreamWaitEvent(side_st, ev_fork_a, 0));
                    side_busy = true;
                }
                if (!par && fuse1 && fast3x3 && conv3x3_wgrad_tile_ok(w)) { w.deferred = w3jobs; w3_deferred = true; }
                if ((rc = conv_wgrad(w, par ? side_st : st))) return rc;
            }
            EffSrc e1{DU, mid, Y, mid, 0, mid, PY, QY, 0.f, 0, 0};
            if (fuse1) {
                // The slab reductions stay on `st` behind the launch (on the side stream, with double-buffered slabs, the step was 0.25 ms LONGER);
                // one launch reduces this kernel's slabs and the 3x3 weight gradient's.
                if ((rc = bwd1x1_fused_launch(fa, st))) return rc;
                // Round 5: the norm1 link rides in the reduction launch (an extra z-plane of k_slab_reduce_link): both are ~5 us latency-floor
                // launches on the critical chain and independent of each other -- 60 launches fewer per step (TCVN_SPLIT_LINK: two launches)
                static const bool split_link = TCVN_KNOB_SET("TCVN_SPLIT_LINK");
                if (split_link) {
                    if ((rc = bwd1x1_fused_reduce(fa, gw_of(ls.w1), grad[ls.b1], w3_deferred ? w3jobs : nullptr, st))) return rc;
                    if ((rc = bwd_link(ls.n1, fa.nblk, bstatD, M, P, Q, 1, ls.a1))) return rc;
                } else {
                    const BnSlots& s1 = ls.n1;
                    BnBwdLinkArgs la{part, fa.nblk, s1.C, bstatD, M, kEps, data[s1.w], grad[s1.w], grad[s1.b], grad[ls.a1], P, Q, 1};
                    if ((rc = bwd1x1_fused_reduce(fa, gw_of(ls.w1), grad[ls.b1], w3_deferred ? w3jobs : nullptr, st, &la))) return rc;
                }
                continue;
            }
            if (bi < (int)xa_skipped.size() && l < (int)xa_skipped[bi].size() && xa_skipped[bi][l]) {
                fprintf(stderr, "tcvn: the forward skipped the activated 1x1 input of block %d layer %d but the fused 1x1 backward cannot run\n", bi, l);
                return -15;
            }
            {   // conv1 (1x1) weight gradient
                const WkEntry& ef = wk_find(ls.w1, 0);
                ConvWgradArgs w{};
                w.mode = mode; w.e = e1; w.dWk = gw_of(ls.w1); w.dbias = grad[ls.b1];
                w.fa.mode = mode; w.fa.amode = A_1X1; w.fa.A = D; w.fa.lda = bg.ld; w.fa.M = (int)M; w.fa.N = mid; w.fa.K = ls.cin;
                w.fa.Kp = ef.Kp; w.fa.C = ls.cin; w.fa.H = bg.H; w.fa.W = bg.W;
                w.fa.sc = sc_of(ls.n1); w.fa.sh = sh_of(ls.n1); w.fa.sl = data[ls.a1];
                if (mode == MODE_F32) { w.slab = reinterpret_cast<float*>(ws + L.slab); w.slab_bytes = kSlabBytes; }     // k_gemm_tn_f32 (cin % 4 != 0); same stream as every other user
                if (L.XA[bi][l] >= 0) {
                    const bool par = side_on;
                    char* EY = ws + ((par && (seq & 1)) ? L.ey2 : L.ey);
                    float* tail = reinterpret_cast<float*>(ws + L.slab + kSlabGemmBytes + ((par && (seq & 1)) ? (kSlabBytes - kSlabGemmBytes) / 2 : 0));
                    if (par && seq >= 2) TCVN_CHECK(hipStreamWaitEvent(st, ev_done[seq & 1], 0));   // that EY buffer / tail half is free again
                    SlabJob bias_job{};     // bias column sums: reduced together with the weight-gradient slab below
                    EffMatArgs em{e1, M, EY, mid, grad[ls.b1], tail, &bias_job};
                    if ((rc = eff_materialize_bf16(em, st))) return rc;
                    if (par) {
                        TCVN_CHECK(hipEventRecord(ev_fork_b, st));
                        TCVN_CHECK(hipStreamWaitEvent(side_st, ev_fork_b, 0));
                        side_busy = true;
                    }
                    const int cin8 = (int)round_up(ls.cin, 8);
                    GemmTnArgs ga{EY, mid, mid, ws + L.XA[bi][l], cin8, cin8, M, gw_of(ls.w1), ef.Kp, ws + L.zeros,
                                  reinterpret_cast<float*>(ws + L.slab), kSlabGemmBytes, mid, bias_job};
                    if (!xa_materialize()) {      // raw concat buffer as the R operand, transformed tile by tile in LDS
                        ga.R = D; ga.ldr = bg.ld; ga.rsc = sc_of(ls.n1); ga.rsh = sh_of(ls.n1); ga.rsl = data[ls.a1]; ga.Rreal = ls.cin;
                    }
                    if ((rc = gemm_tn_bf16(ga, "k_gemm_tn_bf16<conv1>", par ? side_st : st))) return rc;
                    if (par) TCVN_CHECK(hipEventRecord(ev_done[seq & 1], side_st));
                } else if ((rc = conv_wgrad(w, st))) return rc;
            }
            if (L.XA[bi][l] >= 0) {   // conv1 data gradient on the NT GEMM (A = EY)
                const WkEntry& etf = wk_find(ls.w1, 1, 1);
                GemmNtArgs ga{};
                ga.epi = EPI_DGRAD; ga.A = ws + ((side_on && (seq & 1)) ? L.ey2 : L.ey); ga.lda = mid; ga.K = mid; ga.M = M; ga.N = ls.cin;
                ga.Wfrag = ws + L.wk + etf.off; ga.Kp = etf.Kp; ga.zeros = ws + L.zeros;
                ga.Xin = D; ga.ldxin = bg.ld; ga.sc = sc_of(ls.n1); ga.sh = sh_of(ls.n1); ga.sl = data[ls.a1];
                ga.Gout = G; ga.ldgo = bg.ld; ga.part = part; ga.nblk = gemm_nt_nblk(ga);
                if ((rc = gemm_nt_bf16(ga, "k_gemm_nt_bf16<dgrad1x1>", st))) return rc;
                if ((rc = bwd_link(ls.n1, ga.nblk, bstatD, M, P, Q, 1, ls.a1))) return rc;
                ++seq;
            } else {   // conv1 data gradient -> G[:, 0:cin] += sc1 * dU1, norm1 partials
                const WkEntry& et = wk_find(ls.w1, 1);
                ConvDgradArgs d{};
                d.mode = mode; d.dmode = DG_1X1; d.e = e1; d.M = (int)M; d.N = ls.cin; d.Kp = et.Kp; d.H = bg.H; d.W = bg.W;
                d.Wt = ws + L.wk + et.off; d.Xin = D; d.ldxin = bg.ld; d.sc = sc_of(ls.n1); d.sh = sh_of(ls.n1); d.sl = data[ls.a1];
                d.Gout = G; d.ldgo = bg.ld; d.accumulate = 1; d.part = part; d.nblk = conv_dgrad_nblk(d);
                if ((rc = conv_dgrad(d, st))) return rc;
                if ((rc = bwd_link(ls.n1, d.nblk, bstatD, M, P, Q, 1, ls.a1))) return rc;
            }
        }
    }

    // ---- stem: AvgPool0 - PReLU0 - BN0 - conv0 ----
    if ((rc = drain())) return rc;                // the stem weight gradient uses the slab; k_unpack reads every weight gradient
    if (last_part) {
        const BlockGeom& b0 = blocks[0];
        float* P = reinterpret_cast<float*>(ws + L.pqD[0]);
        float* Q = P + b0.ld;
        float* P0 = reinterpret_cast<float*>(ws + L.pq0);
        float* Q0 = P0 + cfg.init_ch;
        const long M0 = (long)n * Hc * Wc;
        EffSrc e{ws + L.G[0], b0.ld, ws + L.D[0], b0.ld, 0, cfg.init_ch, P, Q, 0.f, 0, 0};
        if (last_sparse_stem) {
            // stem_sparse.hip: pass 0 = pooling / PReLU0 / BN0 backward sums with the conv0 output rebuilt from the hit list per region;
            // pass 1 = conv0 weight gradient from the same regions with (P0, Q0) applied.  No conv0-sized tensor is read or written.
            const WkEntry& ef = wk_find(s_w0, 0);
            StemSparseArgs sa{};
            sa.coords = last_coords; sa.values = last_values; sa.nnz = last_nnz; sa.n_img = n; sa.H = cfg.H; sa.W = cfg.W; sa.Cpix = cfg.in_ch;
            sa.value_mode = last_value_mode; sa.noise_std = last_noise; sa.seed = seed;
                stem_sparse_carve(sa, ws + L.sidx);
            sa.Wk = ws + L.wk + ef.off; sa.Kp = ef.Kp; sa.bias = data[s_b0];
            sa.Hc = Hc; sa.Wc = Wc; sa.Ho = b0.H; sa.Wo = b0.W;
            sa.sc = sc_of(n0); sa.sh = sh_of(n0); sa.sl = data[s_a0]; sa.e = e; sa.part = part;
            if ((rc = stem_sparse_bwd(sa, 0, st))) return rc;
            if ((rc = bwd_link(n0, stem_sparse_bwd_grid(sa), reinterpret_cast<const double*>(ws + L.bstat0), M0, P0, Q0, 0, s_a0))) return rc;
            sa.P0 = P0; sa.Q0 = Q0; sa.slab = reinterpret_cast<float*>(ws + L.slab); sa.slab_bytes = kSlabBytes; sa.dWk = gw_of(s_w0);
            if ((rc = stem_sparse_bwd(sa, 1, st))) return rc;
        } else {
        Pool0BwdArgs a{mode, ws + L.c0, n, Hc, Wc, cfg.init_ch, sc_of(n0), sh_of(n0), data[s_a0], e, b0.H, b0.W, ws + L.du0, part,
                       pool0_bwd_grid(n, Hc, Wc), nullptr, nullptr};
        if (last_stem_act) {                     // the forward skipped the conv0-output rows no hit reaches: read the shared row for them, skip their gradient rows
            a.act = reinterpret_cast<const uint32_t*>(ws + L.sact); a.cline = ws + L.zeros + 512;
        }
        const bool vec = pool0_bwd_vec_ok(a) && conv3x3_tile_enabled();
        if (last_stem_act && !(vec && cfg.init_ch == 64 && mode == MODE_BF16)) { fprintf(stderr, "tcvn: stem activity bitmap without the tile kernel\n"); return -16; }
        if (vec) { a.nblk = pool0_bwd_vec_grid(n, Hc, Wc); rc = pool0_bwd_vec(a, st); }
        else rc = pool0_bwd(a, st);
        if (rc) return rc;
        if ((rc = bwd_link(n0, a.nblk, reinterpret_cast<const double*>(ws + L.bstat0), M0, P0, Q0, 0, s_a0))) return rc;
        EffSrc e0{ws + L.du0, cfg.init_ch, ws + L.c0, cfg.init_ch, 0, cfg.init_ch, P0, Q0, 0.f, 0, 0};
        const WkEntry& ef = wk_find(s_w0, 0);
        ConvWgradArgs w{};
        w.mode = mode; w.e = e0; w.dWk = gw_of(s_w0); w.dbias = grad[s_b0];
        w.fa.mode = mode; w.fa.amode = A_STEM; w.fa.A = ws + L.img; w.fa.lda = cfg.in_ch; w.fa.M = (int)M0; w.fa.N = cfg.init_ch;
        w.fa.K = 49 * cfg.in_ch; w.fa.Kp = ef.Kp; w.fa.C = cfg.in_ch; w.fa.H = Hc; w.fa.W = Wc; w.fa.Hin = cfg.H; w.fa.Win = cfg.W;
        if (conv3x3_tile_enabled() && cfg.in_ch <= 3 && cfg.init_ch <= 64 && last_coords != nullptr) {
            // conv0 weight gradient from the hit list (bias gradient is exactly zero in exact arithmetic: BN0 follows)
            StemWgradArgs sa{last_coords, last_nnz, ws + L.img, n, cfg.H, cfg.W, cfg.in_ch, e0, Hc, Wc, ef.Kp,
                             reinterpret_cast<float*>(ws + L.slab), kSlabBytes, mode};
            if ((rc = stem_wgrad_sparse(sa, gw_of(s_w0), st))) return rc;
        } else if ((rc = conv_wgrad(w, st))) return rc;
        }
    }
    // kernel-layout weight gradients of the blocks just finished -> reference OIHW gradients (conv0 rides with block 0)
    const int u0 = last_part ? 0 : unpack_first[bi_lo], u1 = unpack_first[bi_hi + 1];
    return unpack_wgrads(reinterpret_cast<const UnpackDesc*>(d_undesc) + u0, u1 - u0, st);
}
