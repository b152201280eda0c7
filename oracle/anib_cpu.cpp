// oracle/anib_cpu.cpp — CPU statement of fragment mode (ANIb).   TEST / MEASUREMENT INFRASTRUCTURE ONLY.
//
// The scalar functions of pyani_amd/csrc/pg_anib_core.h (anchor choice, X-drop extension, HSP row) compiled for the HOST and fed
// with exactly the seeds the GPU's LDS seeding reports in fragment mode: the maximal exact matches of at least 16 bases (every
// query-strand position is looked up: FRAG_QSTEP = 1; pga_seed.inc), found here with an exhaustive sorted 16-mer table.
// What it restates: pyani's ANIb per ordered pair — fragment the query genome into 1020-nt pieces (anib.py:164-203), blastn
// every piece against the subject genome (anib.py:451-471; BLAST+ is third-party, absent: see pg_anib_core.h for what is
// restated of it and tests/golden/anib for the BLAST+ tables it is calibrated on), keep per fragment the first HSP with
// coverage > 70 % and identity > 30 % (anib.py:641-649).  Tests compare the GPU pipeline with this row for row; bench.py may
// time it as the own-cpu baseline of the fragment workload.  Nothing under pyani_amd/ loads this library.
//   g++ -O2 -std=c++17 -pthread -fPIC -shared -Ipyani_amd/csrc oracle/anib_cpu.cpp -o oracle/libanibcpu.so
#include <algorithm>
#include <cstdio>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "pg_anib_core.h"
using namespace pga;

namespace {
struct Genome {
  std::vector<uint32_t> codes, mask;
  std::vector<int32_t> rec_start;  // stream position of each record's first base; last entry = stream length + 1
  int64_t len = 0;
  SeqView view() const { return SeqView{codes.data(), mask.data(), len}; }
};
Genome pack(const uint8_t* seq, const uint64_t* rec_off, uint32_t n_rec) {
  Genome g;
  int64_t len = 0;
  for (uint32_t r = 0; r < n_rec; ++r) len += (int64_t)(rec_off[r + 1] - rec_off[r]) + (r ? 1 : 0);
  g.len = len;
  g.codes.assign(len / 16 + 2, 0);
  g.mask.assign(len / 32 + 2, 0);
  int64_t p = 0;
  for (uint32_t r = 0; r < n_rec; ++r) {
    if (r) ++p;
    g.rec_start.push_back((int32_t)p);
    for (uint64_t i = rec_off[r]; i < rec_off[r + 1]; ++i, ++p) {
      int c = -1;
      switch (seq[i]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; }
      if (c >= 0) { g.codes[p >> 4] |= (uint32_t)c << (2 * (p & 15)); g.mask[p >> 5] |= 1u << (p & 31); }
    }
  }
  g.rec_start.push_back((int32_t)g.len + 1);
  return g;
}

struct Row {     // = pg_anib_row (include/pyani_gpu.h)
  int32_t frag, length, mismatch, gaps, nident, qlen, qstart, qend, sstart, send, srec, score;
};

constexpr int SEED_K = 16, SEED_STEP = FRAG_QSTEP;   // the engine's fragment-mode seeding: every query-strand position

// the seeds of the GPU pipeline for one query strand: maximal exact matches >= 16 containing a sampled 16-mer
template <typename QV>
void sampled_matches(const Genome& S, const std::vector<std::pair<uint32_t, int32_t>>& tab, const std::vector<uint32_t>& start,
                     const QV& Q, std::vector<Match>& out) {
  const SeqView R = S.view();
  uint32_t v = 0;
  int run = 0;
  for (int64_t e = 0; e < Q.len(); ++e) {
    if (!Q.clean(e)) { run = 0; v = 0; continue; }
    v = (v << 2) | (uint32_t)Q.base(e);
    if (++run < SEED_K) continue;
    const int64_t q = e - SEED_K + 1;
    if (q % SEED_STEP) continue;
    const uint32_t b = v >> 10;                              // 22-bit bucket of the 32-bit 16-mer
    for (uint32_t t = start[b]; t < start[b + 1]; ++t) {
      if (tab[t].first != v) continue;
      const int64_t r = tab[t].second;
      int left = 0;
      while (left < SEED_STEP && R.clean(r - 1 - left) && Q.clean(q - 1 - left) && R.base(r - 1 - left) == Q.base(q - 1 - left)) ++left;
      if (left == SEED_STEP) continue;                       // an earlier sampled position of the same match reports it
      int32_t L = SEED_K;
      while (R.clean(r + L) && Q.clean(q + L) && R.base(r + L) == Q.base(q + L)) ++L;
      out.push_back(Match{(int32_t)(r - left), (int32_t)(q - left), left + L, 0});
    }
  }
}

// ---- the word tier (blastn's word size): fragments the 16-mer seeds leave without a reportable HSP ------------------------------
// pyani asks for `-task blastn`, whose seeds are 11-mers (anib.py:465-471); at 70-80 % identity a 1020-nt fragment holds an exact
// 16-mer with probability 0.6-0.99 but an exact 11-mer almost surely.  For such a fragment every 11-mer of either strand is looked
// up in the subject's word index; a hit becomes a seed if its neighbourhood looks like an alignment — at least WORD_FLANK_MIN of
// the WORD_FLANK bases on its left or on its right match on the same diagonal (chance: 8 +- 2.4 of 32) — and is reported once, as
// the maximal exact match around it.  The pair is only searched this way if the 16-mer tier found it related at all (one
// reportable fragment): unrelated genomes have ~10^4 chance word hits per fragment and nothing to find.
struct WordIndex { std::vector<uint32_t> start; std::vector<int32_t> pos; };
void build_word_index(const SeqView& SV, WordIndex& W) {
  const uint32_t NB = 1u << (2 * WORD_K);
  W.start.assign((size_t)NB + 1, 0);
  uint32_t v = 0; int run = 0;
  for (int64_t p = 0; p < SV.len; ++p) {
    if (!SV.clean(p)) { run = 0; v = 0; continue; }
    v = ((v << 2) | (uint32_t)SV.base(p)) & (NB - 1);
    if (++run >= WORD_K) ++W.start[v + 1];
  }
  for (size_t b = 0; b < NB; ++b) W.start[b + 1] += W.start[b];
  W.pos.assign(W.start[NB], 0);
  std::vector<uint32_t> fill(W.start.begin(), W.start.end() - 1);
  v = 0; run = 0;
  for (int64_t p = 0; p < SV.len; ++p) {
    if (!SV.clean(p)) { run = 0; v = 0; continue; }
    v = ((v << 2) | (uint32_t)SV.base(p)) & (NB - 1);
    if (++run >= WORD_K) W.pos[fill[v]++] = (int32_t)(p - WORD_K + 1);
  }
}
// Seeds of the word tier for one (fragment, strand): q_at(p) = base of the fragment's searched strand (4 outside / dirty),
// s_at(p) = subject base (5 outside / dirty).  Appends FragSeeds (maximal exact matches >= WORD_K clipped to the fragment).
template <typename QA, typename SA>
void word_tier_seeds(QA&& q_at, int32_t qlen, SA&& s_at, const WordIndex& W, std::vector<FragSeed>& out) {
  const uint32_t NB = 1u << (2 * WORD_K);
  uint32_t v = 0; int run = 0;
  for (int32_t e = 0; e < qlen; ++e) {
    const int c = q_at(e);
    if (c > 3) { run = 0; v = 0; continue; }
    v = ((v << 2) | (uint32_t)c) & (NB - 1);
    if (++run < WORD_K) continue;
    const int32_t i = e - WORD_K + 1;
    for (uint32_t t = W.start[v]; t < W.start[v + 1]; ++t) {
      const int64_t p = W.pos[t];
      if (i > 0 && q_at(i - 1) == s_at(p - 1)) continue;          // not the first word of its exact match inside the fragment
      int left = 0, right = 0;
      for (int k = 1; k <= WORD_FLANK; ++k) left += q_at(i - k) == s_at(p - k);
      for (int k = 0; k < WORD_FLANK; ++k) right += q_at(i + WORD_K + k) == s_at(p + WORD_K + k);
      if (left < WORD_FLANK_MIN && right < WORD_FLANK_MIN) continue;
      int32_t L = WORD_K;
      while (i + L < qlen && q_at(i + L) == s_at(p + L)) ++L;
      out.push_back(FragSeed{(int32_t)p, i, L});
    }
  }
}

void run_pair(const Genome& Q, const Genome& S, int32_t fragsize, std::vector<Row>& rows) {
  const SeqView SV = S.view(), QVw = Q.view();
  std::vector<std::pair<uint32_t, int32_t>> tab;
  {
    uint32_t v = 0; int run = 0;
    for (int64_t p = 0; p < SV.len; ++p) {
      if (!SV.clean(p)) { run = 0; v = 0; continue; }
      v = (v << 2) | (uint32_t)SV.base(p);
      if (++run >= SEED_K) tab.push_back({v, (int32_t)(p - SEED_K + 1)});
    }
    std::sort(tab.begin(), tab.end());
  }
  std::vector<uint32_t> start((size_t(1) << 22) + 1, 0);
  for (const auto& e : tab) ++start[(e.first >> 10) + 1];
  for (size_t b = 0; b < (size_t(1) << 22); ++b) start[b + 1] += start[b];
  // fragments of the query: (stream position, length), ids run across the records (anib.py:190-200)
  std::vector<std::pair<int32_t, int32_t>> frags;
  std::vector<int32_t> rec_frag0;
  for (size_t rec = 0; rec + 1 < Q.rec_start.size(); ++rec) {
    rec_frag0.push_back((int32_t)frags.size());
    const int32_t r0 = Q.rec_start[rec], r1 = Q.rec_start[rec + 1] - 1;
    for (int32_t f0 = r0; f0 < r1; f0 += fragsize) frags.push_back({f0, std::min(fragsize, r1 - f0)});
  }
  std::vector<std::vector<FragSeed>> seeds[2], wseeds[2];      // wseeds: what the word tier adds (fragment_rows merges them in)
  for (int strand = 0; strand < 2; ++strand) {
    seeds[strand].assign(frags.size(), {});
    StrandView QS{QVw, strand};
    std::vector<Match> mem;
    sampled_matches(S, tab, start, QS, mem);
    for (const Match& m : mem) {
      const int64_t a = strand ? Q.len - m.q - m.len : m.q, b = a + m.len;           // forward interval of the match in the query
      const int rec = record_of(Q.rec_start.data(), (int)Q.rec_start.size() - 1, (int32_t)a);
      const int32_t r0 = Q.rec_start[rec];
      for (int32_t f = rec_frag0[rec] + (int32_t)((a - r0) / fragsize); f < (int32_t)frags.size() && frags[f].first < b; ++f) {
        const int64_t fp = frags[f].first, fe = fp + frags[f].second;
        if (fp >= Q.rec_start[rec + 1] - 1) break;
        const int64_t lo = std::max(a, fp), hi = std::min(b, fe);
        if (hi - lo < FRAG_MIN_CLIP) continue;
        FragSeed e;
        e.len = (int32_t)(hi - lo);
        if (!strand) { e.q = (int32_t)(lo - fp); e.s = m.r + (int32_t)(lo - a); }
        else { e.q = (int32_t)(fe - hi); e.s = m.r + (int32_t)(b - hi); }
        seeds[strand][f].push_back(e);
      }
    }
  }
  // rows of one fragment from its seed lists (both strands): blastn's start points on the seeds' diagonals, extensions, e-value
  int64_t db_len = 0;
  const int32_t db_seqs = (int32_t)S.rec_start.size() - 1;
  for (int32_t r = 0; r < db_seqs; ++r) db_len += S.rec_start[r + 1] - 1 - S.rec_start[r];
  auto fragment_rows = [&](size_t f, std::vector<Row>& cand) {
    const int32_t fp = frags[f].first, qlen = frags[f].second;
    FragInit init[2][FRAG_MAX_SEEDS], init2[2][FRAG_MAX_SEEDS];
    int64_t dg[2][FRAG_MAX_SEEDS];
    int pick[2][FRAG_MAX_CAND], members[2][FRAG_MAX_CAND] = {}, nc[2] = {0, 0};
    int32_t best_score = 0;
    for (int strand = 0; strand < 2; ++strand) {
      std::vector<FragSeed>& e = seeds[strand][f];
      if (e.empty() && (wseeds[strand].empty() || wseeds[strand][f].empty())) continue;      // (word seeds alone are a list too)
      const auto by_len = [](const FragSeed& x, const FragSeed& y) { return x.len != y.len ? x.len > y.len : (x.q != y.q ? x.q < y.q : x.s < y.s); };
      std::sort(e.begin(), e.end(), by_len);
      if (!wseeds[strand].empty() && !wseeds[strand][f].empty()) {
        // the word tier's seeds come FIRST: each passed the flank test (chance: ~1e-7 per hit), while a fragment with a low-complexity
        // stretch can hold more than FRAG_MAX_SEEDS chance 16-mers, all longer than a real alignment's 11 ... 15-base words
        std::vector<FragSeed> w = wseeds[strand][f];
        std::sort(w.begin(), w.end(), by_len);
        if (w.size() > (size_t)FRAG_WORD_FIRST) w.resize(FRAG_WORD_FIRST);      // (at most half of the list: the own seeds keep the other half)
        w.insert(w.end(), e.begin(), e.end());
        e.swap(w);
        wseeds[strand][f].clear();
      }
      if (e.size() > (size_t)FRAG_MAX_SEEDS) e.resize(FRAG_MAX_SEEDS);
      auto q_at = [&](int64_t p) -> int {
        if (p < 0 || p >= qlen) return 4;
        const int64_t g = strand ? fp + (qlen - 1 - p) : fp + p;
        if (!QVw.clean(g)) return 4;
        return strand ? 3 - QVw.base(g) : QVw.base(g);
      };
      const int n = (int)e.size();
      for (int t = 0; t < n; ++t) {                       // one walk per distinct seed diagonal
        const int64_t diag = (int64_t)e[t].s - e[t].q;
        dg[strand][t] = diag;
        init[strand][t] = init2[strand][t] = FragInit{0, 0, 0, 0};
        bool dup = false;
        for (int u = 0; u < n; ++u) dup = dup || (u != t && (int64_t)e[u].s - e[u].q == diag && frag_seed_before(e[u], e[t]));      // the diagonal's longest seed walks it
        if (dup) continue;
        const int srec = record_of(S.rec_start.data(), (int)S.rec_start.size() - 1, e[t].s);
        const int64_t s_lo = S.rec_start[srec], s_hi = S.rec_start[srec + 1] - 1;
        auto match = [&](int32_t p) -> bool { const int qb = q_at(p); const int64_t sp = p + diag; return qb < 4 && sp >= s_lo && sp < s_hi && SV.clean(sp) && SV.base(sp) == qb; };
        init[strand][t] = frag_diag_best_init(match, qlen, diag, e[t].q, &init2[strand][t]);
      }
      nc[strand] = frag_pick_inits(init[strand], dg[strand], n, pick[strand], members[strand], FRAG_MAX_CAND);
      if (getenv("ANIB_DEBUG_FRAG") && atoi(getenv("ANIB_DEBUG_FRAG")) == (int)f) {
        for (int t = 0; t < n; ++t) fprintf(stderr, "DBG frag %zu strand %d seed %d: s %d q %d len %d diag %lld -> init score %d q_start %d len %d word %d | second %d\n", f, strand, t, e[t].s, e[t].q, e[t].len, (long long)dg[strand][t], init[strand][t].score, init[strand][t].q_start, init[strand][t].len, init[strand][t].q_off, init2[strand][t].score);
        for (int c = 0; c < nc[strand]; ++c) fprintf(stderr, "DBG   cand %d: seed %d members %d\n", c, pick[strand][c], members[strand][c]);
      }
      for (int c = 0; c < nc[strand]; ++c) best_score = std::max(best_score, init[strand][pick[strand][c]].score);
    }
    for (int strand = 0; strand < 2; ++strand) {
      auto q_at = [&](int64_t p) -> int {
        if (p < 0 || p >= qlen) return 4;
        const int64_t g = strand ? fp + (qlen - 1 - p) : fp + p;
        if (!QVw.clean(g)) return 4;
        return strand ? 3 - QVw.base(g) : QVw.base(g);
      };
      // phase A: the preliminary stage of every candidate that is not a lone chance-sized hit (pg_anib_core.h, FragPrelim)
      FragPrelim pbest[FRAG_MAX_CAND];
      bool have[FRAG_MAX_CAND] = {};
      const int n_seeds = (int)seeds[strand][f].size() < FRAG_MAX_SEEDS ? (int)seeds[strand][f].size() : FRAG_MAX_SEEDS;
      auto init_of = [&](int x) -> const FragInit& { return (x & 1) ? init2[strand][x >> 1] : init[strand][x >> 1]; };
      for (int c = 0; c < nc[strand]; ++c) {
        const FragInit& I = init[strand][pick[strand][c]];
        if (!frag_keep_init(I.score, best_score) || frag_init_is_lone_weak(I.score, members[strand][c])) continue;
        const int64_t diag = dg[strand][pick[strand][c]];
        const int srec = record_of(S.rec_start.data(), (int)S.rec_start.size() - 1, (int32_t)(I.q_off + diag));
        const int64_t s_lo = S.rec_start[srec], s_hi = S.rec_start[srec + 1] - 1;
        auto s_at = [&](int64_t p) -> int { return (p >= s_lo && p < s_hi && SV.clean(p)) ? SV.base(p) : 5; };
        struct Ent { FragInit I; int64_t d; };
        std::vector<Ent> order;      // the initial HSPs of the candidate's neighbourhood: per diagonal its best and its second
        for (int t = 0; t < n_seeds; ++t) {
          if (init[strand][t].score <= 0) continue;
          const int64_t dd = dg[strand][t] - diag;
          if (dd >= FRAG_VOTE_FAR || -dd >= FRAG_VOTE_FAR) continue;
          bool taken = false;      // (the neighbourhood of an earlier candidate)
          for (int p = 0; p < c; ++p) { const int64_t d1 = dg[strand][t] - dg[strand][pick[strand][p]]; taken = taken || (d1 < FRAG_VOTE_FAR && -d1 < FRAG_VOTE_FAR); }
          if (taken) continue;
          if (record_of(S.rec_start.data(), (int)S.rec_start.size() - 1, (int32_t)(init[strand][t].q_off + dg[strand][t])) != srec) continue;
          order.push_back(Ent{init[strand][t], dg[strand][t]});
          if (init2[strand][t].score > 0) order.push_back(Ent{init2[strand][t], dg[strand][t]});
        }
        if (getenv("ANIB_ALL_DIAGS")) {
          // EXPERIMENT: blastn's initial HSPs come from EVERY 11-mer diagonal, not only from the diagonals of the product's seeds: the
          // neighbouring diagonals without a seed are walked too
          for (int64_t dd = -(FRAG_VOTE_FAR - 1); dd < FRAG_VOTE_FAR; ++dd) {
            const int64_t d2 = diag + dd;
            bool has_seed = false, taken = false;
            for (int t = 0; t < n_seeds; ++t) has_seed = has_seed || dg[strand][t] == d2;
            for (int p = 0; p < c; ++p) { const int64_t d1 = d2 - dg[strand][pick[strand][p]]; taken = taken || (d1 < FRAG_VOTE_FAR && -d1 < FRAG_VOTE_FAR); }
            if (has_seed || taken) continue;
            auto match2 = [&](int32_t p) -> bool { const int qb = q_at(p); return qb < 4 && s_at(p + d2) == qb; };
            FragInit second{0, 0, 0, 0};
            const FragInit best = frag_diag_walk(match2, qlen, d2, &second);
            if (best.score > 0 && record_of(S.rec_start.data(), (int)S.rec_start.size() - 1, (int32_t)(best.q_off + d2)) == srec) {
              order.push_back(Ent{best, d2});
              if (second.score > 0) order.push_back(Ent{second, d2});
            }
          }
        }
        std::sort(order.begin(), order.end(), [&](const Ent& x, const Ent& y) { return frag_init_before(x.I, x.d, y.I, y.d); });
        FragPrelim pre[BL_MAX_PRELIMS];
        int np = 0, tried = 0;
        FragPrelim first{0, 0, 0, 0, 0, 0, 0};
        for (const Ent& en : order) {
          if (tried == BL_MAX_PRELIMS) break;
          const FragInit& J = en.I;
          const int64_t dj = en.d;
          bool inside = false;
          for (int u = 0; u < np; ++u) inside = inside || frag_init_contained(J, dj, pre[u]);
          if (inside) continue;
          const int32_t g0 = frag_word_start(J.q_off, J.q_off + dj - s_lo, qlen);
          const int64_t as = g0 + dj;
          auto qr = [&](int32_t x2) { return q_at(g0 + 1 + x2); };
          auto sr = [&](int32_t x2) { return s_at(as + 1 + x2); };
          auto ql = [&](int32_t x2) { return q_at(g0 - 1 - x2); };
          auto sl = [&](int32_t x2) { return s_at(as - 1 - x2); };
          const int32_t cap = FRAG_SIZE + FRAG_SLACK;
          const FragExt Rp = frag_extend(qr, qlen - (g0 + 1), sr, (int32_t)std::min<int64_t>(s_hi - (as + 1), cap), FRAG_NEG, FRAG_XDROP_PRELIM);
          const FragExt Lp = frag_extend(ql, g0, sl, (int32_t)std::min<int64_t>(as - s_lo, cap), FRAG_NEG, FRAG_XDROP_PRELIM);
          const FragPrelim P{Rp.score + Lp.score + FRAG_MATCH, g0 - Lp.di, g0 + 1 + Rp.di, g0, as - Lp.dj, as + 1 + Rp.dj, dj};
          if (tried++ == 0) first = P;
          if (frag_evalue_ok_db(P.score, qlen, db_len, db_seqs)) pre[np++] = P;
        }
        const FragPrelim* best = nullptr;
        for (int u = 0; u < np; ++u) if (!best || frag_prelim_before(pre[u], *best)) best = &pre[u];
        if (!best && tried > 0 && first.score >= BL_PRELIM_RESCUE) best = &first;
        if (best) { pbest[c] = *best; have[c] = true; }
      }
      // a repeat family: the final alignments of the FRAG_MAX_FINALS best preliminary ones only (frag_pick_inits)
      {
        int n_have = 0;
        for (int c = 0; c < nc[strand]; ++c) n_have += have[c];
        while (n_have > FRAG_MAX_FINALS) {
          int worst = -1;
          for (int c = 0; c < nc[strand]; ++c) if (have[c] && (worst < 0 || !frag_prelim_before(pbest[c], pbest[worst]))) worst = c;
          have[worst] = false; --n_have;
        }
      }
      // phase B: the final alignments, in the candidates' order
      for (int c = 0; c < nc[strand]; ++c) {
        const FragInit& I = init[strand][pick[strand][c]];
        if (!frag_keep_init(I.score, best_score)) continue;
        const bool lone_weak = frag_init_is_lone_weak(I.score, members[strand][c]);
        if (!lone_weak && !have[c]) continue;
        const int64_t diag = dg[strand][pick[strand][c]];
        const int srec = record_of(S.rec_start.data(), (int)S.rec_start.size() - 1, (int32_t)(I.q_off + diag));
        const int64_t s_lo = S.rec_start[srec], s_hi = S.rec_start[srec + 1] - 1;
        auto s_at = [&](int64_t p) -> int { return (p >= s_lo && p < s_hi && SV.clean(p)) ? SV.base(p) : 5; };
        int32_t g;
        int64_t gdiag = diag;
        if (lone_weak) {
          auto match = [&](int32_t p) -> bool { const int qb = q_at(p); return qb < 4 && s_at(p + diag) == qb; };
          const int32_t lo = (int32_t)std::max<int64_t>(0, s_lo - diag), hi = (int32_t)std::min<int64_t>(qlen, s_hi - diag);
          g = frag_start_point(match, qlen, I.q_off, I.q_off + diag - s_lo, lo, hi, I.score);
        } else {
          const FragPrelim& best = pbest[c];
          gdiag = best.diag;
          auto match_b = [&](int32_t p) -> bool { const int qb = q_at(p); return qb < 4 && s_at(p + gdiag) == qb; };
          const int32_t rec_lo = (int32_t)std::max<int64_t>(0, s_lo - gdiag);
          g = frag_start_point_boxed(match_b, best.g, best.g + gdiag, best.q0, best.q1, best.s0, best.s1, rec_lo);
        }
        const FragHit h = frag_hsp(q_at, qlen, s_at, s_lo, s_hi, g, g + gdiag, 1, [&](int32_t sc) { return frag_evalue_ok_db(sc, qlen, db_len, db_seqs); }, lone_weak);
        if (getenv("ANIB_DUMP_STARTS")) fprintf(stderr, "PSTART %zu %d %d %lld %d init score %d q_start %d len %d word %d -> score %d\n", f, strand, g, (long long)(g + gdiag - s_lo), srec, I.score, I.q_start, I.len, I.q_off, h.score);
        if (!frag_evalue_ok_db(h.score, qlen, db_len, db_seqs)) continue;
        Row r;
        r.frag = (int32_t)f; r.length = h.length; r.mismatch = h.mismatch; r.gaps = h.gaps; r.nident = h.nident; r.qlen = qlen;
        r.srec = srec; r.score = h.score;
        if (!strand) { r.qstart = h.qs + 1; r.qend = h.qe; r.sstart = h.ss - (int32_t)s_lo + 1; r.send = h.se - (int32_t)s_lo; }
        else { r.qstart = qlen - h.qe + 1; r.qend = qlen - h.qs; r.sstart = h.se - (int32_t)s_lo; r.send = h.ss - (int32_t)s_lo + 1; }
        cand.push_back(r);
      }
    }
    std::stable_sort(cand.begin(), cand.end(), [](const Row& x, const Row& y) { return x.score > y.score; });
    if (cand.size() > (size_t)FRAG_KEEP_ROWS) cand.resize(FRAG_KEEP_ROWS);
    {   // HSPs with a common start or end point (frag_rows_share_end): the better one stays
      std::vector<Row> keep;
      for (const Row& r : cand) {
        bool drop = false;
        for (const Row& k : keep) drop = drop || frag_rows_share_end(k.qstart, k.qend, k.sstart, k.send, r.qstart, r.qend, r.sstart, r.send);
        if (!drop) keep.push_back(r);
      }
      cand.swap(keep);
    }
    // -max_target_seqs 1: only the subject record of the fragment's best HSP is reported
    if (!cand.empty()) {
      const int32_t keep = cand[0].srec;
      cand.erase(std::remove_if(cand.begin(), cand.end(), [keep](const Row& x) { return x.srec != keep; }), cand.end());
    }
  };
  auto reportable = [](const std::vector<Row>& cand) {   // parse_blast_tab's test (anib.py:641-649) on any row of the fragment
    for (const Row& r : cand) {
      const int32_t alnlen = r.length - r.gaps;
      if ((double)alnlen / r.qlen > 0.7 && (double)(alnlen - r.mismatch) / r.qlen > 0.3) return true;
    }
    return false;
  };
  std::vector<std::vector<Row>> per_frag(frags.size());
  size_t n_reportable = 0;
  for (size_t f = 0; f < frags.size(); ++f) { fragment_rows(f, per_frag[f]); n_reportable += reportable(per_frag[f]); }
  if (n_reportable > 0 && n_reportable < frags.size() && !getenv("ANIB_NO_WORD_TIER")) {
    WordIndex W;
    build_word_index(SV, W);
    for (size_t f = 0; f < frags.size(); ++f) {
      if (reportable(per_frag[f])) continue;
      const int32_t fp = frags[f].first, qlen = frags[f].second;
      bool added = false;
      for (int strand = 0; strand < 2; ++strand) {
        auto q_at = [&](int64_t p) -> int {
          if (p < 0 || p >= qlen) return 4;
          const int64_t g = strand ? fp + (qlen - 1 - p) : fp + p;
          if (!QVw.clean(g)) return 4;
          return strand ? 3 - QVw.base(g) : QVw.base(g);
        };
        auto s_at = [&](int64_t p) -> int { return (p >= 0 && p < SV.len && SV.clean(p)) ? SV.base(p) : 5; };
        std::vector<FragSeed> extra;
        word_tier_seeds(q_at, qlen, s_at, W, extra);
        std::vector<FragSeed> fresh;     // not a piece of one of the fragment's own (at most FRAG_MAX_SEEDS) seeds
        for (const FragSeed& x : extra) {
          bool dup = false;
          for (const FragSeed& y : seeds[strand][f]) if (y.s - y.q == x.s - x.q && x.q >= y.q && x.q + x.len <= y.q + y.len) dup = true;
          if (!dup) fresh.push_back(x);
        }
        if (getenv("ANIB_DUMP_STARTS")) fprintf(stderr, "WORDTIER frag %zu strand %d: %zu word seeds, %zu fresh\n", f, strand, extra.size(), fresh.size());
        if (fresh.empty() || fresh.size() > (size_t)(WORD_MAX_SEEDS - FRAG_MAX_SEEDS)) continue;   // (a repeat family: left as it is)
        if (wseeds[strand].empty()) wseeds[strand].assign(frags.size(), {});
        wseeds[strand][f] = fresh;
        added = true;
      }
      if (added) { per_frag[f].clear(); fragment_rows(f, per_frag[f]); }
    }
  }
  for (size_t f = 0; f < frags.size(); ++f)
    for (const Row& r : per_frag[f]) rows.push_back(r);
}
}  // namespace

extern "C" {
// One ordered pair: rows of the BLAST-shaped table (at most 4 per fragment: 2 strands x 2 anchors), best score first within a
// fragment.  Returns the number of rows (written up to cap).
int64_t anib_cpu_pair(const uint8_t* qseq, const uint64_t* qrec_off, uint32_t q_nrec, const uint8_t* sseq, const uint64_t* srec_off,
                      uint32_t s_nrec, int32_t fragsize, Row* out, uint64_t cap) {
  const Genome Q = pack(qseq, qrec_off, q_nrec), S = pack(sseq, srec_off, s_nrec);
  std::vector<Row> rows;
  run_pair(Q, S, fragsize, rows);
  for (size_t i = 0; i < rows.size() && i < cap; ++i) out[i] = rows[i];
  return (int64_t)rows.size();
}
}
