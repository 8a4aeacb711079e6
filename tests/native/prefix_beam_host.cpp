// TEST INFRASTRUCTURE ONLY: the device prefix-beam ALGORITHM (end-to-end-asr-pytorch_amd/csrc/prefix_beam.inc,
// the very source the gfx950 kernel is built from) compiled for the host as a single thread, so that
// tests/test_prefix_beam_cpu.py can check it against the reference's golden hypotheses without a GPU.
// The product never links or calls this file.  Built by the test with g++ -O2 -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define PB_HD
#define PB_FOR(i, n) for (int i = 0; i < (n); ++i)
#define PB_SYNC() ((void)0)
#define PB_TID0 true
#define PB_STAMP(k) ((void)0)
static inline double pb_exp(double v) { return std::exp(v); }
static inline double pb_log1p(double v) { return std::log1p(v); }
#include "../../end-to-end-asr-pytorch_amd/csrc/prefix_beam.inc"

namespace {
struct Host {
    PBState s;
    std::vector<unsigned char> mem;
    int cur = 0, T = 0;
    std::vector<unsigned char> allowed;
};

template <typename Tp>
Tp *carve(std::vector<unsigned char> &m, size_t &o, size_t n) {
    o = (o + 15) & ~(size_t)15;
    Tp *p = reinterpret_cast<Tp *>(m.data() + o);
    o += n * sizeof(Tp);
    return p;
}

// candidate ranking of src/ctc.py:296-303 (host twin of pb_rank_row in prefix_beam.hip)
void rank_row(const PBState &s, int i, const float *x, const float *lmrow, float lw, const unsigned char *allowed) {
    float psc = INFINITY;
    int pv = -1;
    for (int c = 0; c < s.C; ++c) {
        float bsc = -INFINITY;
        int bv = 0x7fffffff;
        for (int v = 0; v < s.V; ++v) {
            if (!allowed[v]) continue;
            const float sc = lmrow ? x[v] + lw * lmrow[v] : x[v];
            const bool eligible = sc < psc || (sc == psc && v > pv);
            if (eligible && (sc > bsc || (sc == bsc && v < bv))) { bsc = sc; bv = v; }
        }
        s.cand[i * s.C + c] = bv == 0x7fffffff ? 0 : bv;
        psc = bsc; pv = bv;
    }
}
}  // namespace

extern "C" void *pbh_new(int W, int C, int V, int T, const unsigned char *allowed) {
    Host *h = new Host();
    PBState &s = h->s;
    s.W = W; s.C = C; s.V = V; s.Lcap = T + 1; s.Scap = 5 * (T + 1);
    h->T = T;
    h->allowed.assign(allowed, allowed + V);
    h->mem.assign((size_t)64 << 20, 0);
    size_t o = 0;
    for (int k = 0; k < 2; ++k) {
        s.beam[k].pb = carve<double>(h->mem, o, W); s.beam[k].pnb = carve<double>(h->mem, o, W);
        s.beam[k].tok = carve<int>(h->mem, o, (size_t)W * s.Lcap);
        s.beam[k].str = carve<unsigned char>(h->mem, o, (size_t)W * s.Scap);
        s.beam[k].len = carve<int>(h->mem, o, W); s.beam[k].slen = carve<int>(h->mem, o, W);
        s.beam[k].upd = carve<int>(h->mem, o, W);
        s.beam[k].tail = carve<unsigned long long>(h->mem, o, PB_PAIRS);
        s.beam[k].lcp = carve<int>(h->mem, o, PB_PAIRS);
        s.beam[k].dif = carve<signed char>(h->mem, o, PB_PAIRS);
        s.beam[k].pre = carve<unsigned char>(h->mem, o, PB_PAIRS);
    }
    s.nb = carve<int>(h->mem, o, 4);
    s.e_pb = carve<double>(h->mem, o, PB_MAX_ENTRIES); s.e_pnb = carve<double>(h->mem, o, PB_MAX_ENTRIES);
    s.e_sc = carve<double>(h->mem, o, PB_MAX_ENTRIES);
    s.key = carve<double>(h->mem, o, PB_MAX_ENTRIES);
    s.e_dig = carve<unsigned long long>(h->mem, o, PB_MAX_ENTRIES);
    s.s_pb1 = carve<double>(h->mem, o, PB_MAX_BEAM); s.s_pnb1 = carve<double>(h->mem, o, PB_MAX_BEAM);
    s.s_same = carve<double>(h->mem, o, PB_MAX_BEAM); s.s_diff = carve<double>(h->mem, o, PB_MAX_BEAM);
    s.e_par = carve<int>(h->mem, o, PB_MAX_ENTRIES); s.e_tok = carve<int>(h->mem, o, PB_MAX_ENTRIES);
    s.sorted = carve<int>(h->mem, o, PB_MAX_ENTRIES); s.m_list = carve<int>(h->mem, o, PB_MAX_ENTRIES);
    s.order = carve<int>(h->mem, o, PB_MAX_BEAM); s.off = carve<int>(h->mem, o, PB_MAX_BEAM);
    s.fin = carve<int>(h->mem, o, PB_MAX_BEAM);
    s.r_len = carve<int>(h->mem, o, PB_MAX_BEAM); s.r_slen = carve<int>(h->mem, o, PB_MAX_BEAM);
    s.r_last = carve<int>(h->mem, o, PB_MAX_BEAM);
    s.t_lcp = carve<int>(h->mem, o, PB_PAIRS);
    s.t_tail = carve<unsigned long long>(h->mem, o, PB_PAIRS);
    s.t_tkey = carve<unsigned long long>(h->mem, o, PB_PAIRS);
    s.e_lk = carve<unsigned long long>(h->mem, o, PB_MAX_ENTRIES);
    s.t_dif = carve<signed char>(h->mem, o, PB_PAIRS);
    s.t_pre = carve<unsigned char>(h->mem, o, PB_PAIRS);
    s.cand = carve<int>(h->mem, o, (size_t)W * C);
    s.scal = carve<int>(h->mem, o, 4);
    s.bnd = carve<unsigned char>(h->mem, o, PB_MAX_ENTRIES);
    s.out_parent = carve<int>(h->mem, o, W); s.out_last = carve<int>(h->mem, o, W); s.out_gidx = carve<int>(h->mem, o, W);
    if (o > h->mem.size() || W > PB_MAX_BEAM || (size_t)W * (C + 1) > PB_MAX_ENTRIES) { delete h; return nullptr; }
    const PBBeam &b = s.beam[0];
    b.len[0] = 0; b.slen[0] = 0; b.pb[0] = 0.0; b.pnb[0] = PB_LOG_ZERO; b.upd[0] = 1;
    b.lcp[0] = 0; b.dif[0] = 0; b.pre[0] = 1; b.tail[0] = 0ull;
    s.nb[0] = 1;
    return h;
}

// one frame: x [V]; lm [nb rows][V] or NULL; returns the live-row count of the new beam
extern "C" int pbh_frame(void *hp, const float *x, const float *lm, float lw, int last_frame, int lm_follows) {
    Host *h = reinterpret_cast<Host *>(hp);
    PBState &s = h->s;
    const int nb = s.nb[h->cur];
    const int rows = lm ? nb : 1;
    for (int i = 0; i < rows; ++i) rank_row(s, i, x, lm ? lm + (size_t)i * s.V : nullptr, lw, h->allowed.data());
    if (!lm)
        for (int z = 0; z < (nb - 1) * s.C; ++z) s.cand[s.C + z] = s.cand[z % s.C];
    pb_frame(s, h->cur, x, lm, lw, last_frame, lm_follows);
    h->cur ^= 1;
    return s.nb[h->cur];
}

extern "C" void pbh_get(void *hp, int *len, int *tok /*[W][T+1]*/, int *parent, int *last, int *gidx) {
    Host *h = reinterpret_cast<Host *>(hp);
    const PBState &s = h->s;
    const PBBeam &b = s.beam[h->cur];
    std::memcpy(len, b.len, sizeof(int) * s.W);
    std::memcpy(tok, b.tok, sizeof(int) * (size_t)s.W * s.Lcap);
    if (parent) std::memcpy(parent, s.out_parent, sizeof(int) * s.W);
    if (last) std::memcpy(last, s.out_last, sizeof(int) * s.W);
    if (gidx) std::memcpy(gidx, s.out_gidx, sizeof(int) * s.W);
}

extern "C" void pbh_free(void *hp) { delete reinterpret_cast<Host *>(hp); }
