// anim_debug.cpp — HOST build of the ANIm per-pair core (pyani_amd/csrc/pg_anim_core.h) for development in the
// GPU-less build container: reads two FASTA files, finds exact matches with a sorted 20-mer table on the CPU, then
// runs the same MUM filter / clustering / extension / 1-to-1 filter / reduction functions the HIP kernels run.
// NOT part of the product and NOT the oracle.   g++ -O2 -std=c++17 -I../../pyani_amd/csrc anim_debug.cpp -o anim_debug
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include "pg_anim_core.h"
#include "pg_nucmer_core.h"
#include "pg_nucmer_diag.h"
#include "pg_anim_trace.h"
using namespace pga;

struct Genome {
  std::vector<uint32_t> codes, mask;
  std::vector<int32_t> rec_start;  // stream position of each record's first base; last entry = stream length + 1
  std::vector<std::string> ids;
  int64_t len = 0;
  SeqView view() const { return SeqView{codes.data(), mask.data(), len}; }
};

static Genome load(const char* path) {
  Genome g;
  std::ifstream in(path);
  std::string line, seq;
  std::vector<std::string> recs;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') { g.ids.push_back(line.substr(1, line.find_first_of(" \t\r") - 1)); recs.emplace_back(); }
    else if (!recs.empty()) for (char c : line) if (c != ' ' && c != '\r' && c != '\n') recs.back().push_back(c);
  }
  std::string stream;
  for (size_t r = 0; r < recs.size(); ++r) { if (r) stream.push_back('#'); g.rec_start.push_back((int32_t)stream.size()); stream += recs[r]; }
  g.len = (int64_t)stream.size();
  g.rec_start.push_back((int32_t)g.len + 1);
  g.codes.assign(g.len / 16 + 2, 0); g.mask.assign(g.len / 32 + 2, 0);
  for (int64_t p = 0; p < g.len; ++p) {
    int c = -1;
    switch (stream[p]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; }
    if (c >= 0) { g.codes[p >> 4] |= (uint32_t)c << (2 * (p & 15)); g.mask[p >> 5] |= 1u << (p & 31); }
  }
  return g;
}

// all maximal exact matches >= MIN_MATCH between ref and one query strand
template <typename QV>
static void find_mems(const Genome& G, const QV& Q, int strand, std::vector<Match>& out) {
  const SeqView R = G.view();
  const int K = MIN_MATCH;
  std::vector<std::pair<uint64_t, int32_t>> tab;
  auto kmer = [&](auto& S, int64_t p, uint64_t& v) { v = 0; for (int t = 0; t < K; ++t) { if (!S.clean(p + t)) return false; v = (v << 2) | (uint64_t)S.base(p + t); } return true; };
  for (int64_t p = 0; p + K <= R.len; ++p) { uint64_t v; if (kmer(R, p, v)) tab.push_back({v, (int32_t)p}); }
  std::sort(tab.begin(), tab.end());
  for (int64_t q = 0; q + K <= Q.len(); ++q) {
    uint64_t v;
    if (!kmer(Q, q, v)) continue;
    auto it = std::lower_bound(tab.begin(), tab.end(), std::make_pair(v, (int32_t)-1));
    for (; it != tab.end() && it->first == v; ++it) {
      const int64_t r = it->second;
      if (R.clean(r - 1) && Q.clean(q - 1) && R.base(r - 1) == Q.base(q - 1)) continue;  // not left-maximal
      int32_t L = K;
      while (R.clean(r + L) && Q.clean(q + L) && R.base(r + L) == Q.base(q + L)) ++L;
      out.push_back(Match{(int32_t)r, (int32_t)q, L, strand});
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: anim_debug ref.fna qry.fna [--dump]\n"); return 2; }
  const bool dump = argc > 3 && !strcmp(argv[3], "--dump");
  // (--exact is accepted and ignored: the postnuc statement, pg_nucmer_core.h, is the only extender since round 5)
  long exact_cells = 0;
  bool want_delta = false;        // --delta (with --exact): every alignment's .delta indel list after its ALN line
  for (int i = 3; i < argc; ++i) if (!strcmp(argv[i], "--delta")) want_delta = true;
  std::vector<std::vector<int64_t>> deltas_all;
  std::vector<int32_t> visit_all;       // per alignment: the PIECE_VISIT key (MUMmer's print order: by visit, forward strand first)
  Genome G = load(argv[1]), H = load(argv[2]);
  const SeqView R = G.view();
  std::vector<Aln> alns;
  std::vector<int32_t> a_rrec, a_qrec;
  // per strand: what seeding and clustering leave for the walk, and what the walk leaves for the output
  struct StrandState {
    std::vector<Chain> chains; std::vector<Match> cm; std::vector<int32_t> co; int n = 0, n_chains = 0, n_cm = 0;
    std::vector<pgn::PnTurn> tlog; std::vector<int32_t> born;
    std::vector<Aln> alns; std::vector<int32_t> a_rrec, a_qrec; std::vector<std::vector<int64_t>> deltas; std::vector<int32_t> visit; long cells = 0;
  } ST[2];
  const int nq = (int)H.rec_start.size() - 1;
  for (int strand = 0; strand < 2; ++strand) {
    StrandView Q{H.view(), strand};
    std::vector<Match> mem;
    find_mems(G, Q, strand, mem);
    int n = (int)mem.size();
    bool maxmatch = false;
    for (int i = 3; i < argc; ++i) if (!strcmp(argv[i], "--maxmatch")) maxmatch = true;
    if (!maxmatch)
      n = mum_filter(mem.data(), n, strand, [&](int32_t q) { return record_of(H.rec_start.data(), nq, strand ? (int32_t)(H.len - 1 - q) : q); });
    else
      std::sort(mem.begin(), mem.end(), [](const Match& a, const Match& b) { return a.q != b.q ? a.q < b.q : a.r < b.r; });      // mgaps' By_Start2: query start, then reference start
    mem.resize(n);
    std::vector<int32_t> rrec(n), qrec(n), parent(n), score(n), from(n), adj(n), order(n);
    for (int i = 0; i < n; ++i) {
      rrec[i] = record_of(G.rec_start.data(), (int)G.rec_start.size() - 1, mem[i].r);
      const int32_t qf = strand ? (int32_t)(H.len - 1 - mem[i].q) : mem[i].q;
      qrec[i] = record_of(H.rec_start.data(), nq, qf);
    }
    StrandState& T = ST[strand];
    T.n = n;
    T.chains.resize(n + 1); T.cm.resize(n + 1);
    std::vector<Chain>& chains = T.chains;
    std::vector<Match>& cm = T.cm;
    int& n_chains = T.n_chains; int& n_cm = T.n_cm;
    mgaps_strand(mem.data(), n, strand, rrec.data(), qrec.data(), parent.data(), score.data(), from.data(), adj.data(),
                 order.data(), chains.data(), n_chains, (int)chains.size(), cm.data(), n_cm, (int)cm.size());
    n_chains = split_chains_by_ref_record(chains.data(), n_chains, cm.data(), [&](int32_t r) { return record_of(G.rec_start.data(), (int)G.rec_start.size() - 1, r); });
    // order chains by first-match ref start; pick forward targets
    std::vector<int32_t>& co = T.co;
    co.resize(n_chains);
    for (int i = 0; i < n_chains; ++i) co[i] = i;
    std::sort(co.begin(), co.end(), [&](int a, int b) { return chain_before(chains.data(), cm.data(), a, b); });
    T.tlog.resize(n_chains + 1); T.born.resize(n_chains + 1);
  }
  // the two strands' walks side by side (pgn::PnPairSync: MUMmer walks the clusters of both strands of a record pair in ONE list;
  // the walks consult each other where that matters), one thread each
  pgn::PnHostShared pair_shared;
  auto walk_strand = [&](int strand) {
    StrandState& T = ST[strand];
    StrandView Q{H.view(), strand};
    std::vector<Chain>& chains = T.chains;
    std::vector<Match>& cm = T.cm;
    std::vector<int32_t>& co = T.co;
    const int n = T.n, n_chains = T.n_chains, n_cm = T.n_cm;
    std::vector<Aln>& alns = T.alns;
    std::vector<int32_t>&a_rrec = T.a_rrec, &a_qrec = T.a_qrec;
    std::vector<std::vector<int64_t>>& deltas_all = T.deltas;
    std::vector<int32_t>& visit_all = T.visit;
    long& exact_cells = T.cells;
    pgn::PnPairSync<pgn::PnHostPrim> sync{pgn::PnHostPrim{&pair_shared, strand}, T.tlog.data(), ST[1 - strand].tlog.data(), T.born.data()};
    {
      const int cap = 1 << 15;
      std::vector<pgn::Cell> d0(cap), d1(cap), d2(cap);
      pgn::ScalarEngine<SeqView, StrandView> eng{R, Q, d0.data(), d1.data(), d2.data(), cap};
      std::vector<uint8_t> fused(n_chains + 1);
      std::vector<pgn::PnAln> al(n_chains + 1);
      pgn::DiagEngine<SeqView, StrandView> deng{pgn::DiagScalarEngine<SeqView, StrandView>{R, Q}, eng};
      auto bounds_of = [&](int c, int32_t& rl, int32_t& rh, int32_t& ql, int32_t& qh) {
            rl = G.rec_start[chains[c].rrec]; rh = G.rec_start[chains[c].rrec + 1] - 1;
            ql = H.rec_start[chains[c].qrec]; qh = H.rec_start[chains[c].qrec + 1] - 1;
            if (strand) { const int32_t a = (int32_t)H.len - qh, b = (int32_t)H.len - ql; ql = a; qh = b; } };
      std::vector<pgn::PnPiece> pieces;
      if (want_delta) { pieces.resize((size_t)3 * n_cm + (size_t)4 * n_chains + 16); eng.pieces = pieces.data(); eng.piece_cap = (int32_t)pieces.size(); }
      // ANIM_HOIST: every cluster's forward extension first (what the GPU's pre-pass does), the walk then only reads them
      std::vector<pgn::PnFwd> fw;
      if (getenv("ANIM_HOIST")) {
        fw.resize(n_chains);
        for (int k = 0; k < n_chains; ++k) fw[k] = pgn::postnuc_forward(eng, chains.data(), cm.data(), co.data(), n_chains, k, bounds_of);
        eng.fwd = fw.data();
        if (getenv("ANIM_FWD_LEN"))      // development: the extent of every forward search in walk order (anti-diagonals ~ bases of A + bases of B, + the break length where the target was not reached)
          for (int k = 0; k < n_chains; ++k) {
            const Chain& C = chains[co[k]];
            const Match ml = cm[C.first + C.count - 1];
            fprintf(stderr, "FWDLEN %d\n", (fw[k].eA - (ml.r + ml.len - 1)) + (fw[k].eB - (ml.q + ml.len - 1)) + (fw[k].reached ? 0 : pgn::BREAK_LEN));
          }
      }
      // ... and every match-to-match alignment (the GPU's gap pre-pass): the walk takes runs of them at once (ScalarEngine::gap_run)
      std::vector<pgn::PnGap> gp;
      if (getenv("ANIM_HOIST") && !want_delta) {
        gp.assign(n_cm + 1, pgn::PnGap{0, 0, 0, -1});
        for (int c = 0; c < n_chains; ++c)
          for (int m = 0; m + 1 < chains[c].count; ++m) {
            const Match a = cm[chains[c].first + m], b = cm[chains[c].first + m + 1];
            int32_t tA = b.r, tB = b.q, err = 0;
            const bool reached = eng.align(a.r + a.len - 1, tA, a.q + a.len - 1, tB, pgn::FORWARD_ALIGN, err);
            gp[chains[c].first + m] = pgn::PnGap{tA, tB, err, (reached && !eng.overflow) ? 1 : 0};
          }
        eng.gaps = gp.data();
      }
      // ANIM_BWD_AHEAD (with ANIM_HOIST): the walk rehearsed without its backward searches, the searches it predicts run ahead, the real
      // walk takes those whose arguments it repeats — what the GPU's backward pre-pass does (results must not change; the rate is printed)
      std::vector<pgn::PnBwd> bwd;
      if (getenv("ANIM_BWD_AHEAD") && !fw.empty()) {
        bwd.assign(n_chains, pgn::PnBwd{0, 0, 0, 0, 0u, 0, 0, 0, 0});
        std::vector<uint8_t> fused2(n_chains + 1);
        std::vector<pgn::PnAln> al2(n_chains + 1);
        pgn::PnRehearsal<pgn::ScalarEngine<SeqView, StrandView>> dry{eng, bwd.data()};
        { pgn::PnNoSync alone; pgn::postnuc_unit(dry, chains.data(), cm.data(), co.data(), n_chains, bounds_of, fused2.data(), al2.data(), (int)al2.size(), alone, strand); }
        long predicted = 0;
        for (auto& b : bwd) if (b.state == 1) {
          int32_t a = b.tA, q = b.tB, err = 0;
          b.reached = eng.align(b.sA, a, b.sB, q, b.m_o, err) ? 1 : 0;
          b.rA = a; b.rB = q; b.state = 2; ++predicted;
        }
        const long ahead_cells = eng.search_cells;
        eng.searches = 0; eng.search_cells = 0;
        eng.bwd = bwd.data();
        fprintf(stderr, "backward searches run ahead: %ld (%ld cells); ", predicted, ahead_cells);
      }
      // ANIM_DIAGWAVE: the diagonal-layout wave engines of the GPU (pg_nucmer_diag.h), emulated lane by lane, under the same walk
      pgd::DiagWaveEngine<SeqView, StrandView> weng(R, Q, d0.data(), d1.data(), d2.data(), cap);
      if (getenv("ANIM_FALLBACK_LOG"))      // the runs no single-wave window holds (on the GPU: the group kernel / the column strips)
        weng.fallback_log = [](int32_t N, int32_t M, unsigned m_o, int32_t band_w) {
          if (m_o & pgn::FORCED_BIT) fprintf(stderr, "FALLBACK forced N %d M %d band %d span %lld\n", N, M, band_w,
                                             band_w >= 0 ? (long long)(N > M ? N - M : M - N) + 2ll * band_w + 6 : (long long)N + M + 6);
          else fprintf(stderr, "FALLBACK search N %d M %d\n", N, M);
        };
      const int na = getenv("ANIM_DIAGWAVE") ? pgn::postnuc_unit(weng, chains.data(), cm.data(), co.data(), n_chains, bounds_of, fused.data(), al.data(), (int)al.size(), sync, strand)
                   : getenv("ANIM_DIAG") ? pgn::postnuc_unit(deng, chains.data(), cm.data(), co.data(), n_chains, bounds_of, fused.data(), al.data(), (int)al.size(), sync, strand)
                                         : pgn::postnuc_unit(eng, chains.data(), cm.data(), co.data(), n_chains,
          [&](int c, int32_t& rl, int32_t& rh, int32_t& ql, int32_t& qh) {
            rl = G.rec_start[chains[c].rrec]; rh = G.rec_start[chains[c].rrec + 1] - 1;
            ql = H.rec_start[chains[c].qrec]; qh = H.rec_start[chains[c].qrec + 1] - 1;
            if (strand) { const int32_t a = (int32_t)H.len - qh, b = (int32_t)H.len - ql; ql = a; qh = b; } },
          fused.data(), al.data(), (int)al.size(), sync, strand);
      if (getenv("ANIM_DIAGWAVE"))
        fprintf(stderr, "diag-wave engines: calls %ld / %ld / %ld / %ld / %ld / %ld / %ld / %ld (128 / 256 / 384 / 512 / 768 / 1024 / 1536 / 2048 diagonals), window moves %ld, did not fit %ld, "
                        "fell back to the scalar engine %ld, cells %ld + %ld\n",
                weng.e2.calls, weng.e4.calls, weng.e6.calls, weng.e8.calls, weng.e12.calls, weng.e16.calls, weng.e24.calls, weng.e32.calls,
                weng.e2.moves + weng.e4.moves + weng.e6.moves + weng.e8.moves + weng.e12.moves + weng.e16.moves + weng.e24.moves + weng.e32.moves,
                weng.e2.fails + weng.e4.fails + weng.e6.fails + weng.e8.fails + weng.e12.fails + weng.e16.fails + weng.e24.fails + weng.e32.fails, weng.fallbacks,
                weng.e2.cells + weng.e4.cells + weng.e6.cells + weng.e8.cells + weng.e12.cells + weng.e16.cells + weng.e24.cells + weng.e32.cells, weng.slow.cells);
      if (na < 0 || eng.overflow || deng.slow.overflow || weng.slow.overflow) { fprintf(stderr, "postnuc statement: capacity exceeded\n"); exit(3); }
      exact_cells += eng.cells + deng.fast.cells + deng.slow.cells;
      if (!bwd.empty()) fprintf(stderr, "the walk still ran %ld itself (%ld cells)\n", eng.searches, eng.search_cells);
      if (want_delta) {
        // the paths of the walk's search / forced pieces, by the scalar engine with its traceback store (on the GPU: anim_trace_kernel)
        if (eng.n_pieces > eng.piece_cap) { fprintf(stderr, "piece list too small\n"); exit(3); }
        eng.pieces = nullptr;
        std::vector<std::vector<uint32_t>> rle((size_t)eng.n_pieces);
        for (int p = 0; p < eng.n_pieces; ++p) {
          const pgn::PnPiece& P = pieces[p];
          if (P.kind == pgn::PIECE_MATCH || P.kind == pgn::PIECE_VISIT) continue;
          const int32_t N = P.kind == pgn::PIECE_FORCED ? P.A1 - P.A0 + 1 : P.tA - P.A0 + 1, M = P.kind == pgn::PIECE_FORCED ? P.B1 - P.B0 + 1 : P.tB - P.B0 + 1;
          std::vector<uint8_t> bp((size_t)P.cells + 1);
          std::vector<uint32_t> doff((size_t)N + M + 4);
          std::vector<int32_t> dlo((size_t)N + M + 4);
          pgn::PnTrace tr{bp.data(), (uint64_t)P.cells, doff.data(), dlo.data(), (int32_t)doff.size(), 0, 0, 0, 0};
          eng.trace = &tr;
          int32_t a = P.kind == pgn::PIECE_FORCED ? P.A1 : P.tA, b = P.kind == pgn::PIECE_FORCED ? P.B1 : P.tB, err = 0;
          eng.align(P.A0, a, P.B0, b, P.m_o, err);
          eng.trace = nullptr;
          if (tr.overflow || eng.overflow || a != P.A1 || b != P.B1) { fprintf(stderr, "trace: piece %d does not repeat (%d %d vs %d %d, overflow %d)\n", p, a, b, P.A1, P.B1, tr.overflow); exit(3); }
          rle[p].resize((size_t)N + M + 4);
          const int32_t cnt = pgn::pn_trace_back(tr, rle[p].data(), (int32_t)rle[p].size());
          if (cnt < 0) { fprintf(stderr, "trace: broken path at piece %d\n", p); exit(3); }
          rle[p].resize((size_t)cnt);
        }
        std::vector<std::vector<int64_t>> deltas;
        std::vector<int32_t> visit;
        std::string why;
        if (!pgt::unit_deltas(pieces.data(), eng.n_pieces, al.data(), na, [&](int32_t p, int32_t& cnt) { cnt = (int32_t)rle[p].size(); return rle[p].data(); }, deltas, visit, &why)) {
          fprintf(stderr, "trace: %s\n", why.c_str()); exit(3); }
        for (auto& d : deltas) deltas_all.push_back(std::move(d));
        for (int i = 0; i < na; ++i) visit_all.push_back(visit[(size_t)i]);
      }
      if (getenv("ANIM_DIAG")) { fprintf(stderr, "calls / cells by class (0 = trimmed, 1.. = forced w 32, 64, ..., 15 = whole):"); for (int t = 0; t < 16; ++t) if (deng.stat_calls[t]) fprintf(stderr, " [%d] %ld / %ld", t, deng.stat_calls[t], deng.stat_cells[t]); fprintf(stderr, "\n"); }
      if (getenv("ANIM_DIAG")) fprintf(stderr, "diagonal-window engine: %ld cells, %ld calls fell back to the general engine (%ld cells)\n", deng.fast.cells, deng.fast.fallbacks, deng.slow.cells);
      for (int i = 0; i < na; ++i) {
        Aln a; a.rs = al[i].sA; a.re = al[i].eA + 1; a.qs = al[i].sB; a.qe = al[i].eB + 1; a.errors = al[i].errors; a.strand = strand; a.keep = 0;
        a_rrec.push_back(record_of(G.rec_start.data(), (int)G.rec_start.size() - 1, a.rs));
        if (strand) { const int32_t qs = (int32_t)H.len - a.qe, qe = (int32_t)H.len - a.qs; a.qs = qs; a.qe = qe; }
        a_qrec.push_back(record_of(H.rec_start.data(), nq, a.qs));
        alns.push_back(a);
      }
      fprintf(stderr, "strand %d: MEMs->MUMs %d, chains %d, alignments %d (postnuc statement, %ld cells); shadow tests that asked for the record pair's current alignment: %ld, answered otherwise than the strand's own current alignment would: %ld\n", strand, n, n_chains, na, exact_cells, sync.asked, sync.differs);
    }
  };
  {
    std::thread other(walk_strand, 1);
    walk_strand(0);
    other.join();
  }
  for (int strand = 0; strand < 2; ++strand) {
    StrandState& T = ST[strand];
    alns.insert(alns.end(), T.alns.begin(), T.alns.end());
    a_rrec.insert(a_rrec.end(), T.a_rrec.begin(), T.a_rrec.end());
    a_qrec.insert(a_qrec.end(), T.a_qrec.begin(), T.a_qrec.end());
    for (auto& d : T.deltas) deltas_all.push_back(std::move(d));
    visit_all.insert(visit_all.end(), T.visit.begin(), T.visit.end());
    exact_cells += T.cells;
  }
  if (pga::g_chain_full_scans) fprintf(stderr, "chain DP: %ld scans beyond the 64-entry window, %ld of them changed the predecessor\n", pga::g_chain_full_scans, pga::g_chain_full_wins);
  const int n = (int)alns.size();
  std::vector<int32_t> idx(n + 1), from(n + 1);
  std::vector<double> sc(n + 1);
  bool nofilter = false;
  for (int i = 3; i < argc; ++i) if (!strcmp(argv[i], "--nofilter")) nofilter = true;
  if (nofilter) for (auto& a : alns) a.keep = 3;
  else { lis_filter(alns.data(), n, 0, a_rrec.data(), a_qrec.data(), idx.data(), sc.data(), from.data()); lis_filter(alns.data(), n, 1, a_qrec.data(), a_rrec.data(), idx.data(), sc.data(), from.data()); }
  PairResult pr = reduce_pair(alns.data(), n, a_rrec.data(), a_qrec.data(), idx.data());
  printf("%lld %lld %.16g %lld %lld\n", (long long)pr.ref_aln_len, (long long)pr.qry_aln_len, (double)pr.weighted / (double)pr.aligned,
         (long long)pr.sim_errors, (long long)pr.n_alignments);
  if (dump)
    for (int i = 0; i < n; ++i) {
      const Aln& a = alns[i];
      const int32_t ro = G.rec_start[a_rrec[i]], qo = H.rec_start[a_qrec[i]];
      printf("ALN %s %s %d %d %d %d %d keep=%d\n", G.ids[a_rrec[i]].c_str(), H.ids[a_qrec[i]].c_str(), a.rs - ro + 1, a.re - ro,
             a.strand ? a.qe - qo : a.qs - qo + 1, a.strand ? a.qs - qo + 1 : a.qe - qo, a.errors, a.keep);
      if (want_delta && (size_t)i < deltas_all.size()) { printf("VISIT %d %d\n", visit_all[i], a.strand); for (int64_t d : deltas_all[i]) printf("%lld\n", (long long)d); printf("0\n"); }
    }
  return 0;
}
