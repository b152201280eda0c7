// oracle/nucmer_oracle.cpp — CPU restatement of `nucmer --mum` (MUMmer 3.23) for ONE ordered genome pair.
// TEST INFRASTRUCTURE ONLY: nothing under pyani_amd/ builds, loads or calls this file.
//
// What pyani runs per ordered pair (pyani/anim.py:240-289): `nucmer --mum -p <out> ref qry` (+ `--maxmatch`), then
// `delta-filter -1`, then its own parse_delta (anim.py:292-411).  MUMmer is a third-party dependency (conda pin mummer=3.23,
// requirements-thirdparty-linux.txt) that is NOT under /root/reference and is not installed in this image, so it can be neither
// built nor run here.  This file restates MUMmer 3.23's PUBLISHED algorithm — the pipeline nucmer drives:
//     mummer -mum -b -l 20 -n   ->   mgaps -l 65 -s 90 -d 5 -f .12   ->   postnuc -b 200
// stage by stage and with its own data structures (1-based inclusive coordinates, clusters, alignment objects, the dynamic
// anti-diagonal band of the extender), independently of the product's engine (pyani_amd/csrc/pg_anim_core.h: fixed 64-diagonal
// band, per-chain searches, packed keys).  It is pinned against every MUMmer output file the reference's tests hold and whose
// genomes are available (tests/golden/anim/**: 25 192 alignment records of 43 nucmer runs, coordinates + error counts, and the
// indel lists of the .delta files) — see tests/test_nucmer_oracle.py and tools/anim_host_fixture_check.py --oracle.
//
//   mummer (maximal unique matches)   Kurtz' maxmat3: maximal matches of length >= l whose string is unique in the reference
//                                     (the MUM candidates), then `mumuniqueinquery`: candidates sorted by reference start
//                                     (longer first), one whose reference interval ends at or before the running right end is
//                                     dropped, two with the same interval drop each other.  Per query RECORD and strand.
//   mgaps                             union-find over the matches sorted by query start (separation <= 90, diagonal difference
//                                     <= max(5, 0.12 * separation)), then per component repeated extraction of the best chain
//                                     (score = sum of lengths - overlap - diagonal drift), printed if the lengths sum to >= 65.
//   postnuc                           extendClusters: clusters in reference order; backward search from a cluster's first match
//                                     towards the closest earlier alignment end (getReverseTargetAlignment), re-aligned forward;
//                                     forward alignment match to match and from the last match towards the closest later cluster
//                                     (getForwardTargetCluster); a reached target fuses the two; shadowed clusters are skipped.
//   sw_align (_alignEngine)           anti-diagonal DP, three states per cell (+3 / -7, gap open -10 ... see SCORES below), the
//                                     band grows by one cell per side and anti-diagonal and is trimmed from its edges where a cell
//                                     has fallen more than breaklen * 3 below the best; the search ends breaklen anti-diagonals
//                                     after the last high score (ties move it forward) or when the target corner is computed.
//
// Output (stdout): one line "ALN <ref id> <qry id> <rs> <re> <qs> <qe> <errors>" per alignment in .delta coordinates (1-based,
// reverse-strand alignments with qs > qe), followed by its .delta indel list when --delta is given.
//   g++ -O2 -std=c++17 oracle/nucmer_oracle.cpp -o oracle/_build/nucmer_oracle
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace nuc {

// ---- parameters (nucmer defaults, as pyani leaves them) ----------------------------------------------------------------
static int MIN_MATCH = 20, MIN_CLUSTER = 65, MAX_GAP = 90, DIAG_DIFF = 5, BREAK_LEN = 200;
static double DIAG_FACTOR = 0.12;
static const long MAX_ALIGNMENT_LENGTH = 10000;
// SCORES: nucleotide matrix of sw_alignscore.hh
static int GOOD_SCORE = 3, BAD_SCORE = -7, OPEN_GAP_SCORE = -10, CONT_GAP_SCORE = -7;
// open questions of the restatement, settled on the fixtures (tools/anim_host_fixture_check.py --oracle):
static long MAX_DIFF = -1;        // trim threshold; -1 = GOOD_SCORE * BREAK_LEN
static int TRIM_STRICT = 1;       // 1: trim when high - value > MAX_DIFF, 0: >=
static int FORCED_TRIM = 0;       // forced alignments (FORCED_BIT) neither break nor trim: with trimming 4 of the 43 fixture runs lose their way
static int STORED_BOUNDS = 1;     // a cell reads neighbours that were computed and then trimmed
static int EXP_TIE = 0;           // EXPERIMENT (not MUMmer): among equal scores prefer fewer errors (the engine's packed-key rule)
static int EXP_BAND = 0;          // EXPERIMENT (not MUMmer): confine the band to +-EXP_BAND diagonals (design studies for the GPU engine)
static int EXP_BAND_SHIFT = 1;    //   ... centred between the start diagonal and the target's (as the engine's shifted band)

static int STATS = 0;
static long stat_maxw = 0, stat_cells = 0, stat_diags = 0, stat_forced_cells = 0, stat_hist[64];
static const long NEG = -(1L << 40);
enum { DELETE = 0, INSERT = 1, MATCH = 2, NONE = 3 };   // DELETE consumes a B base, INSERT an A base
enum { DIRECTION_BIT = 1, SEARCH_BIT = 2, FORCED_BIT = 4, OPTIMAL_BIT = 8 };
static const unsigned FORWARD_ALIGN = 1, FORCED_FORWARD_ALIGN = 5, BACKWARD_SEARCH = 2;

struct Node { long v[3]; uint8_t used[3]; uint8_t mx; };
struct Diagonal { long jlo, jhi; std::vector<Node> I; };   // cells (i = Dct - j, j) for j in [jlo, jhi]

static inline void score_edit(Node& c, int st, long del, long ins, long mat) {
  if (del > ins) { if (del > mat) { c.v[st] = del; c.used[st] = DELETE; } else { c.v[st] = mat; c.used[st] = MATCH; } }
  else if (ins > mat) { c.v[st] = ins; c.used[st] = INSERT; }
  else { c.v[st] = mat; c.used[st] = MATCH; }
  if (c.v[st] < NEG / 2) c.v[st] = NEG;
}
static inline uint8_t max_state(const Node& c) {
  if (c.v[DELETE] > c.v[INSERT]) return c.v[DELETE] > c.v[MATCH] ? DELETE : MATCH;
  return c.v[INSERT] > c.v[MATCH] ? INSERT : MATCH;
}
static inline long plus(long a, long b) { return a < NEG / 2 ? NEG : a + b; }

struct Seq {             // 1-based sequence text, [0] unused
  const char* s; long len;
  char at(long p) const { return (p >= 1 && p <= len) ? s[p] : '\0'; }
};
static inline bool same_base(char a, char b) {
  return a == b && (a == 'A' || a == 'C' || a == 'G' || a == 'T');
}

// The alignment engine.  Aligns A[Astart .. Aend] with B[Bstart .. Bend] (inclusive; backwards when DIRECTION_BIT is clear).
// On return Aend / Bend hold the finish position; `ops` (if not null and not a search) receives the edit path from the start,
// one char per column: 'M' (a base of each), 'I' (a base of A only), 'D' (a base of B only).
static bool align_engine(const Seq& A, long Astart, long& Aend, const Seq& B, long Bstart, long& Bend, std::string* ops, unsigned m_o) {
  const bool fwd = m_o & DIRECTION_BIT;
  const long N = fwd ? Aend - Astart + 1 : Astart - Aend + 1, M = fwd ? Bend - Bstart + 1 : Bstart - Bend + 1;
  auto a_at = [&](long i) { return A.at(fwd ? Astart + i - 1 : Astart - i + 1); };
  auto b_at = [&](long j) { return B.at(fwd ? Bstart + j - 1 : Bstart - j + 1); };
  const long SC = EXP_TIE ? 65536 : 1, ER = EXP_TIE ? 1 : 0;   // EXP_TIE: value = score * 65536 - errors
  const long max_diff = (MAX_DIFF >= 0 ? MAX_DIFF : (long)GOOD_SCORE * BREAK_LEN) * SC;
  const long kGOOD = GOOD_SCORE * SC, kBAD = BAD_SCORE * SC - ER, kOPEN = OPEN_GAP_SCORE * SC - ER, kCONT = CONT_GAP_SCORE * SC - ER;
  auto sc_of = [&](long v) { return EXP_TIE ? ((v + 65535) >> 16) : v; };
  const bool forced = m_o & FORCED_BIT;
  const bool keep_all = !(m_o & SEARCH_BIT);
  std::vector<Diagonal> Diag;
  Diag.reserve(1024);
  Diag.push_back(Diagonal{0, 0, std::vector<Node>(1)});
  Diag[0].I[0] = Node{{NEG, NEG, 0}, {NONE, NONE, NONE}, MATCH};
  long high_score = NEG * 2, FinishCt = 0, FinishJ = 0;
  long jlo = 0, jhi = 1;     // band of the next anti-diagonal, in j
  long Dct;
  for (Dct = 1; Dct <= N + M && (forced || Dct - FinishCt <= BREAK_LEN) && jlo <= jhi; ++Dct) {
    // clip to the matrix
    long lo = std::max(jlo, std::max(0L, Dct - N)), hi = std::min(jhi, std::min(M, Dct));
    if (EXP_BAND > 0 && !forced) {   // diagonal k = j - i = 2j - Dct within [kc - W, kc + W)
      long kc = 0;
      if (EXP_BAND_SHIFT && !(m_o & OPTIMAL_BIT)) { kc = (M - N) / 2; kc = std::max(-(long)EXP_BAND + 2, std::min((long)EXP_BAND - 2, kc)); }
      const long klo = kc - EXP_BAND, khi = kc + EXP_BAND - 1;
      lo = std::max(lo, (Dct + klo + 1) >> 1);            // 2j - Dct >= klo
      hi = std::min(hi, (Dct + khi) >> 1);                // 2j - Dct <= khi
    }
    if (lo > hi) { break; }
    if (STATS) { const long w = hi - lo + 1; if (!forced) { if (w > stat_maxw) stat_maxw = w; stat_cells += w; stat_diags += 1; ++stat_hist[std::min(w / 8, 63L)]; } else stat_forced_cells += w; }
    Diag.push_back(Diagonal{lo, hi, std::vector<Node>((size_t)(hi - lo + 1))});
    Diagonal& cur = Diag[Dct];
    const Diagonal& p1 = Diag[Dct - 1];
    const Diagonal* p2 = Dct >= 2 ? &Diag[Dct - 2] : nullptr;
    for (long j = lo; j <= hi; ++j) {
      const long i = Dct - j;
      Node& c = cur.I[j - lo];
      // DELETE: from (i, j-1) on the previous anti-diagonal
      if (j - 1 >= p1.jlo && j - 1 <= p1.jhi && j >= 1) {
        const Node& p = p1.I[j - 1 - p1.jlo];
        score_edit(c, DELETE, plus(p.v[DELETE], kCONT), plus(p.v[INSERT], kOPEN), plus(p.v[MATCH], kOPEN));
      } else { c.v[DELETE] = NEG; c.used[DELETE] = NONE; }
      // INSERT: from (i-1, j)
      if (j >= p1.jlo && j <= p1.jhi && i >= 1) {
        const Node& p = p1.I[j - p1.jlo];
        score_edit(c, INSERT, plus(p.v[DELETE], kOPEN), plus(p.v[INSERT], kCONT), plus(p.v[MATCH], kOPEN));
      } else { c.v[INSERT] = NEG; c.used[INSERT] = NONE; }
      // MATCH: from (i-1, j-1) two anti-diagonals back
      if (p2 && i >= 1 && j >= 1 && j - 1 >= p2->jlo && j - 1 <= p2->jhi) {
        const Node& p = p2->I[j - 1 - p2->jlo];
        c.v[MATCH] = plus(p.v[p.mx], same_base(a_at(i), b_at(j)) ? kGOOD : kBAD);
        c.used[MATCH] = p.mx;
      } else { c.v[MATCH] = NEG; c.used[MATCH] = NONE; }
      c.mx = max_state(c);
      if (c.v[c.mx] > NEG / 2 && sc_of(c.v[c.mx]) >= sc_of(high_score)) { high_score = c.v[c.mx]; FinishCt = Dct; FinishJ = j; }
    }
    if (!keep_all && Dct >= 2) { std::vector<Node>().swap(Diag[Dct - 2].I); if (!STORED_BOUNDS) {} }
    // trim hopeless cells from the edges
    long tlo = lo, thi = hi;
    if (!forced || FORCED_TRIM) {
      auto hopeless = [&](long j) { const Node& c = cur.I[j - lo]; const long d = high_score - c.v[c.mx]; return TRIM_STRICT ? d > max_diff : d >= max_diff; };
      while (tlo <= thi && hopeless(tlo)) ++tlo;
      while (thi >= tlo && hopeless(thi)) --thi;
    }
    if (!STORED_BOUNDS) { cur.jlo = tlo; /* cells outside [tlo, thi] are not readable */
      if (tlo > lo) cur.I.erase(cur.I.begin(), cur.I.begin() + (tlo - lo));
      cur.jhi = thi; if ((long)cur.I.size() > thi - tlo + 1) cur.I.resize((size_t)std::max(0L, thi - tlo + 1)); }
    // grow: the neighbours (below / right) of the surviving cells
    jlo = tlo; jhi = thi + 1;
    if (tlo > thi) { jlo = 1; jhi = 0; }
  }
  --Dct;
  bool reached = false;
  if (Dct == N + M) {
    if (!(m_o & OPTIMAL_BIT)) { reached = true; FinishCt = N + M; FinishJ = M; }
    else if (FinishCt == Dct) reached = true;
  }
  const long fi = FinishCt - FinishJ, fj = FinishJ;   // bases consumed, counting the start bases
  Aend = fwd ? Astart + fi - 1 : Astart - fi + 1;
  Bend = fwd ? Bstart + fj - 1 : Bstart - fj + 1;
  if (ops && keep_all) {
    std::string rev;
    long d = FinishCt, j = FinishJ;
    const Diagonal* dg = &Diag[d];
    if (j < dg->jlo || j > dg->jhi) { fprintf(stderr, "nucmer_oracle: finish cell outside its band\n"); exit(3); }
    uint8_t st = dg->I[j - dg->jlo].mx;
    while (d > 0) {
      const Node& c = Diag[d].I[j - Diag[d].jlo];
      const uint8_t from = c.used[st];
      if (st == MATCH) { rev.push_back('M'); d -= 2; j -= 1; }
      else if (st == INSERT) { rev.push_back('I'); d -= 1; }
      else { rev.push_back('D'); d -= 1; j -= 1; }
      if (d > 0 && from == NONE) { fprintf(stderr, "nucmer_oracle: broken trace at d=%ld\n", d); exit(3); }
      st = from == NONE ? (uint8_t)MATCH : from;
    }
    ops->append(rev.rbegin(), rev.rend());
  }
  return reached;
}

// ---- postnuc ------------------------------------------------------------------------------------------------------------
struct Match { long sA, sB, len; };
struct Cluster { bool wasFused = false; char dirB = 0; std::vector<Match> matches; };
struct Alignment {
  long sA, sB, eA, eB; char dirB;
  std::string ops;     // edit path from (sA, sB) to (eA, eB): one op per column, the boundary bases included once
};

static bool is_shadowed(const Cluster& C, const std::vector<Alignment>& Al, long upto) {
  const long sA = C.matches.front().sA, eA = C.matches.back().sA + C.matches.back().len - 1;
  const long sB = C.matches.front().sB, eB = C.matches.back().sB + C.matches.back().len - 1;
  for (long k = upto; k >= 0; --k)
    if (Al[k].dirB == C.dirB && Al[k].eA >= eA && Al[k].eB >= eB && Al[k].sA <= sA && Al[k].sB <= sB) return true;
  return false;
}

static bool close_enough(long lesser, long greater) {
  if (lesser > greater) std::swap(lesser, greater);
  return greater < BREAK_LEN || lesser * GOOD_SCORE + (greater - lesser) * CONT_GAP_SCORE >= 0;
}

static long get_forward_target(std::vector<Cluster>& Cl, long cur, long& targetA, long& targetB) {
  const Cluster& C = Cl[cur];
  const long sA = C.matches.back().sA + C.matches.back().len - 1, sB = C.matches.back().sB + C.matches.back().len - 1;
  long dist = std::min(targetA - sA, targetB - sB);
  long best = -1;
  for (long k = cur + 1; k < (long)Cl.size(); ++k) {
    const Cluster& T = Cl[k];
    if (T.dirB != C.dirB) continue;
    long eA = T.matches.front().sA, eB = T.matches.front().sB;
    if ((eA < sA || eB < sB) && T.matches.back().sA >= sA && T.matches.back().sB >= sB)
      for (size_t m = 0; m < T.matches.size() && (eA < sA || eB < sB); ++m) { eA = T.matches[m].sA; eB = T.matches[m].sB; }
    if (eA >= sA && eB >= sB) {
      long lesser = eA - sA, greater = eB - sB;
      if (lesser > greater) std::swap(lesser, greater);
      if (close_enough(lesser, greater)) { best = k; targetA = eA; targetB = eB; break; }
      else if ((greater << 1) - lesser < dist) { best = k; targetA = eA; targetB = eB; dist = (greater << 1) - lesser; }
    }
  }
  return best;
}

static long get_reverse_target(const std::vector<Alignment>& Al, long cur) {
  const long sA = Al[cur].sA, sB = Al[cur].sB;
  long dist = std::min(sA, sB), best = -1;
  for (long k = cur - 1; k >= 0; --k) {
    if (Al[k].dirB != Al[cur].dirB) continue;
    const long eA = Al[k].eA, eB = Al[k].eB;
    if (eA <= sA && eB <= sB) {
      long lesser = sA - eA, greater = sB - eB;
      if (lesser > greater) std::swap(lesser, greater);
      if (close_enough(lesser, greater)) { best = k; break; }
      else if ((greater << 1) - lesser < dist) { best = k; dist = (greater << 1) - lesser; }
    }
  }
  return best;
}

// The edit path of an extension starts ON the alignment's current last base (already represented in ops) — drop that column.
// Two paths that share one boundary base pair (the last column of dst = the first column of piece): keep it once.
static void append_ops(std::string& dst, const std::string& piece) {
  if (piece.empty()) return;
  if (piece[0] == 'M') dst.append(piece, 1, std::string::npos);
  else if (!dst.empty() && dst.back() == 'M') { dst.pop_back(); dst.append(piece); }
  else { fprintf(stderr, "nucmer_oracle: cannot join two paths at a gapped boundary\n"); exit(3); }
}

static bool extend_forward(Alignment& Al, const Seq& A, long targetA, const Seq& B, long targetB, unsigned m_o) {
  bool overflow = false;
  if (targetA - Al.eA + 1 > MAX_ALIGNMENT_LENGTH) { targetA = Al.eA + MAX_ALIGNMENT_LENGTH - 1; overflow = true; m_o |= OPTIMAL_BIT; }
  if (targetB - Al.eB + 1 > MAX_ALIGNMENT_LENGTH) { targetB = Al.eB + MAX_ALIGNMENT_LENGTH - 1; if (!overflow) m_o |= OPTIMAL_BIT; overflow = true; }
  std::string piece;
  bool reached = align_engine(A, Al.eA, targetA, B, Al.eB, targetB, &piece, m_o);
  if (reached && overflow) reached = false;
  append_ops(Al.ops, piece);
  Al.eA = targetA; Al.eB = targetB;
  return reached;
}

// returns true if `cur` (the last alignment, a bare match) was merged into Al[target]
static bool extend_backward(std::vector<Alignment>& Al, long cur, long target, const Seq& A, const Seq& B) {
  unsigned m_o = BACKWARD_SEARCH;
  long targetA, targetB;
  bool overflow = false;
  if (target >= 0) { targetA = Al[target].eA; targetB = Al[target].eB; }
  else { targetA = 1; targetB = 1; m_o |= OPTIMAL_BIT; }
  if (Al[cur].sA - targetA + 1 > MAX_ALIGNMENT_LENGTH) { targetA = Al[cur].sA - MAX_ALIGNMENT_LENGTH + 1; overflow = true; m_o |= OPTIMAL_BIT; }
  if (Al[cur].sB - targetB + 1 > MAX_ALIGNMENT_LENGTH) { targetB = Al[cur].sB - MAX_ALIGNMENT_LENGTH + 1; if (!overflow) m_o |= OPTIMAL_BIT; overflow = true; }
  bool reached = align_engine(A, Al[cur].sA, targetA, B, Al[cur].sB, targetB, nullptr, m_o | SEARCH_BIT);
  if (overflow || target < 0) reached = false;
  if (reached) {
    extend_forward(Al[target], A, Al[cur].sA, B, Al[cur].sB, FORCED_FORWARD_ALIGN);
    append_ops(Al[target].ops, Al[cur].ops);
    Al[target].eA = Al[cur].eA; Al[target].eB = Al[cur].eB;
    Al.pop_back();
  } else {
    std::string piece;
    long eA = Al[cur].sA, eB = Al[cur].sB;
    if (targetA != eA || targetB != eB) {
      align_engine(A, targetA, eA, B, targetB, eB, &piece, FORCED_FORWARD_ALIGN);
      append_ops(piece, Al[cur].ops);         // the piece's last column is the match's first base
      Al[cur].ops.swap(piece);
    }
    Al[cur].sA = targetA; Al[cur].sB = targetB;
  }
  return reached;
}

static void extend_clusters(std::vector<Cluster>& Cl, const Seq& A, const Seq& Bf, const Seq& Br, std::vector<Alignment>& Al) {
  // postnuc: sort (Clusters.begin(), Clusters.end(), by the reference start of the first match) — std::sort, NOT stable, over an
  // input order that depends on mgaps' union-by-size roots: clusters that start on the same reference base (only --maxmatch makes
  // them) come in an order MUMmer does not define.  This restatement makes itself a function of its input: forward strand first,
  // then by the query-strand start of the first match (round 5; before: the order of this file's own mgaps output).
  std::stable_sort(Cl.begin(), Cl.end(), [](const Cluster& x, const Cluster& y) {
    const Match &a = x.matches.front(), &b = y.matches.front();
    if (a.sA != b.sA) return a.sA < b.sA;
    if (x.dirB != y.dirB) return x.dirB == '+';
    return a.sB < b.sB;
  });
  bool target_reached = false;
  long prev = 0, curc = 0, targetc = -1, cura = -1;
  long targetA = 0, targetB = 0;
  const long n = (long)Cl.size();
  while (curc < n) {
    Cluster& C = Cl[curc];
    if (!target_reached)
      if (C.wasFused || is_shadowed(C, Al, cura)) { C.wasFused = true; curc = ++prev; continue; }
    const Seq& B = C.dirB == '+' ? Bf : Br;
    for (size_t m = 0; m < C.matches.size(); ++m) {
      const Match& Mp = C.matches[m];
      if (target_reached) {
        if (Al[cura].eA != Mp.sA || Al[cura].eB != Mp.sB) {
          if (m + 1 >= C.matches.size()) { fprintf(stderr, "nucmer_oracle: target match does not exist\n"); exit(3); }
          continue;
        }
        Al[cura].eA += Mp.len - 1; Al[cura].eB += Mp.len - 1;
        Al[cura].ops.append((size_t)(Mp.len - 1), 'M');
      } else {
        Al.push_back(Alignment{Mp.sA, Mp.sB, Mp.sA + Mp.len - 1, Mp.sB + Mp.len - 1, C.dirB, std::string((size_t)Mp.len, 'M')});
        cura = (long)Al.size() - 1;
        const long t = get_reverse_target(Al, cura);
        if (extend_backward(Al, cura, t, A, B)) cura = t;
      }
      unsigned m_o = FORWARD_ALIGN;
      if (m + 1 < C.matches.size()) {
        targetA = C.matches[m + 1].sA; targetB = C.matches[m + 1].sB;
        target_reached = extend_forward(Al[cura], A, targetA, B, targetB, m_o);
      } else {
        targetA = A.len; targetB = B.len;
        targetc = get_forward_target(Cl, curc, targetA, targetB);
        if (targetc < 0) m_o |= OPTIMAL_BIT;
        target_reached = extend_forward(Al[cura], A, targetA, B, targetB, m_o);
      }
    }
    if (targetc < 0) target_reached = false;
    C.wasFused = true;
    if (!target_reached) curc = ++prev; else curc = targetc;
  }
}

// ---- mummer -mum + mgaps -----------------------------------------------------------------------------------------------
struct Rec { std::string id, seq; };    // seq[0] is a pad so that positions are 1-based
static std::vector<Rec> load_fasta(const char* path) {
  std::vector<Rec> recs;
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') { recs.push_back(Rec{line.substr(1, line.find_first_of(" \t\r") - 1), std::string(1, '\0')}); }
    else if (!recs.empty()) for (char c : line) if (c != ' ' && c != '\r' && c != '\n' && c != '\t') recs.back().seq.push_back((char)toupper((unsigned char)c));
  }
  return recs;
}
static std::string revcomp(const std::string& s) {   // 1-based in, 1-based out
  std::string r(1, '\0');
  for (size_t p = s.size() - 1; p >= 1; --p) {
    char c = s[p];
    switch (c) { case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break; default: break; }
    r.push_back(c);
  }
  return r;
}
static inline int code(char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; } }

struct RefIndex {       // all reference records joined by one separator, as prenuc hands them to mummer
  std::string text;     // 1-based
  std::vector<long> start;   // first position of each record in text; start[n] = text end + 2
  std::vector<std::pair<uint64_t, int32_t>> tab;
  std::vector<uint32_t> bucket;
  static constexpr int IB = 22;
};
static void build_ref(const std::vector<Rec>& recs, RefIndex& R) {
  R.text.assign(1, '\0');
  for (size_t r = 0; r < recs.size(); ++r) {
    if (r) R.text.push_back('x');
    R.start.push_back((long)R.text.size());
    R.text.append(recs[r].seq, 1, std::string::npos);
  }
  R.start.push_back((long)R.text.size() + 1);
  const int K = MIN_MATCH;
  const uint64_t keep = (1ull << (2 * K)) - 1;
  uint64_t v = 0; int run = 0;
  for (long p = 1; p < (long)R.text.size(); ++p) {
    const int c = code(R.text[p]);
    if (c < 0) { run = 0; v = 0; continue; }
    v = ((v << 2) | (uint64_t)c) & keep;
    if (++run >= K) R.tab.push_back({v, (int32_t)(p - K + 1)});
  }
  std::sort(R.tab.begin(), R.tab.end());
  R.bucket.assign((size_t(1) << RefIndex::IB) + 1, 0);
  for (auto& e : R.tab) ++R.bucket[(e.first >> (2 * K - RefIndex::IB)) + 1];
  for (size_t b = 0; b < (size_t(1) << RefIndex::IB); ++b) R.bucket[b + 1] += R.bucket[b];
}

struct Mem { long r, q, len; };   // r in joined-reference coordinates, q in the query record strand, both 1-based
static void find_mems(const RefIndex& R, const std::string& Q, std::vector<Mem>& out) {
  const int K = MIN_MATCH;
  const uint64_t keep = (1ull << (2 * K)) - 1;
  uint64_t v = 0; int run = 0;
  const long qn = (long)Q.size() - 1, rn = (long)R.text.size() - 1;
  for (long e = 1; e <= qn; ++e) {
    const int c = code(Q[e]);
    if (c < 0) { run = 0; v = 0; continue; }
    v = ((v << 2) | (uint64_t)c) & keep;
    if (++run < K) continue;
    const long q = e - K + 1;
    const uint64_t b = v >> (2 * K - RefIndex::IB);
    for (uint32_t t = R.bucket[b]; t < R.bucket[b + 1]; ++t) {
      if (R.tab[t].first != v) continue;
      const long r = R.tab[t].second;
      if (r > 1 && q > 1 && code(R.text[r - 1]) >= 0 && R.text[r - 1] == Q[q - 1]) continue;   // not left-maximal
      long L = K;
      while (r + L <= rn && q + L <= qn && code(R.text[r + L]) >= 0 && R.text[r + L] == Q[q + L]) ++L;
      out.push_back(Mem{r, q, L});
    }
  }
}

// MUM candidates (unique in the reference) then mumuniqueinquery.  maxmatch: all maximal matches.
static void select_mums(std::vector<Mem>& m, bool maxmatch) {
  if (maxmatch) return;
  // unique in the reference: no other match (another reference position) covers its query interval
  std::sort(m.begin(), m.end(), [](const Mem& a, const Mem& b) { return a.q != b.q ? a.q < b.q : a.len > b.len; });
  std::vector<char> drop(m.size(), 0);
  long maxend = -1;
  for (size_t i = 0; i < m.size(); ++i) {
    const long e = m[i].q + m[i].len;
    if (e <= maxend) drop[i] = 1;
    else if (i + 1 < m.size() && m[i + 1].q == m[i].q && m[i + 1].len == m[i].len) drop[i] = 1;
    if (e > maxend) maxend = e;
  }
  std::vector<Mem> cand;
  for (size_t i = 0; i < m.size(); ++i) if (!drop[i]) cand.push_back(m[i]);
  // mumuniqueinquery over the candidates: by reference start, longer first
  std::sort(cand.begin(), cand.end(), [](const Mem& a, const Mem& b) { return a.r != b.r ? a.r < b.r : a.len > b.len; });
  std::vector<char> ign(cand.size(), 0);
  long dbright = 0;
  for (size_t i = 0; i < cand.size(); ++i) {
    const long right = cand[i].r + cand[i].len - 1;
    if (dbright > right) ign[i] = 1;
    else if (dbright == right) { ign[i] = 1; if (i > 0 && cand[i - 1].r == cand[i].r) ign[i - 1] = 1; }
    else dbright = right;
  }
  m.clear();
  for (size_t i = 0; i < cand.size(); ++i) if (!ign[i]) m.push_back(cand[i]);
}

struct MgMatch { long s1, s2, len; long score, adj, from; bool good; int id; };
// mgaps over the matches of one query record strand (reference = the joined text).  Appends (matches, in output order) per cluster.
static void mgaps(std::vector<Mem>& mem, std::vector<std::vector<Match>>& clusters) {
  const long N = (long)mem.size();
  std::sort(mem.begin(), mem.end(), [](const Mem& a, const Mem& b) { return a.q != b.q ? a.q < b.q : a.r < b.r; });
  std::vector<long> uf((size_t)N);
  for (long i = 0; i < N; ++i) uf[i] = i;
  auto find = [&](long x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
  for (long i = 0; i < N; ++i) {
    const long i_end = mem[i].q + mem[i].len, i_diag = mem[i].q - mem[i].r;
    for (long j = i + 1; j < N; ++j) {
      const long sep = mem[j].q - i_end;
      if (sep > MAX_GAP) break;
      const long dd = labs((mem[j].q - mem[j].r) - i_diag);
      if (dd <= std::max((long)DIAG_DIFF, (long)(DIAG_FACTOR * sep))) { const long a = find(i), b = find(j); if (a != b) uf[a] = b; }
    }
  }
  std::vector<long> order((size_t)N);
  for (long i = 0; i < N; ++i) { uf[i] = find(i); order[i] = i; }
  std::stable_sort(order.begin(), order.end(), [&](long a, long b) { return uf[a] < uf[b]; });
  for (long g0 = 0; g0 < N;) {
    long g1 = g0;
    while (g1 < N && uf[order[g1]] == uf[order[g0]]) ++g1;
    std::vector<MgMatch> A;
    for (long k = g0; k < g1; ++k) A.push_back(MgMatch{mem[order[k]].r, mem[order[k]].q, mem[order[k]].len, 0, 0, -1, false, 0});
    while (!A.empty()) {
      const long n = (long)A.size();
      for (long i = 0; i < n; ++i) {
        A[i].score = A[i].len; A[i].adj = 0; A[i].from = -1; A[i].good = false;
        for (long j = 0; j < i; ++j) {
          const long o1 = A[j].s1 + A[j].len - A[i].s1, o2 = A[j].s2 + A[j].len - A[i].s2;
          const long olap = std::max(std::max(0L, o1), o2);
          const long pen = olap + labs((A[i].s2 - A[i].s1) - (A[j].s2 - A[j].s1));
          if (A[j].score + A[i].len - pen > A[i].score) { A[i].from = j; A[i].score = A[j].score + A[i].len - pen; A[i].adj = olap; }
        }
      }
      long best = 0;
      for (long i = 1; i < n; ++i) if (A[i].score > A[best].score) best = i;
      long total = 0;
      for (long i = best; i >= 0; i = A[i].from) { A[i].good = true; total += A[i].len; }
      if (total >= MIN_CLUSTER) {
        std::vector<Match> out;
        bool first = true;
        for (long i = 0; i < n; ++i)
          if (A[i].good) {
            const long adj = first ? 0 : A[i].adj;
            out.push_back(Match{A[i].s1 + adj, A[i].s2 + adj, A[i].len - adj});
            first = false;
          }
        clusters.push_back(out);
      }
      std::vector<MgMatch> rest;
      for (long i = 0; i < n; ++i) if (!A[i].good) rest.push_back(A[i]);
      A.swap(rest);
    }
    g0 = g1;
  }
}

}  // namespace nuc

int main(int argc, char** argv) {
  using namespace nuc;
  if (argc < 3) { fprintf(stderr, "usage: nucmer_oracle ref.fna qry.fna [--maxmatch] [--delta]\n"); return 2; }
  bool maxmatch = false, delta = false;
  for (int i = 3; i < argc; ++i) { if (!strcmp(argv[i], "--maxmatch")) maxmatch = true; if (!strcmp(argv[i], "--delta")) delta = true; }
  if (const char* e = getenv("NUC_MAX_DIFF")) MAX_DIFF = atol(e);
  if (const char* e = getenv("NUC_TRIM_STRICT")) TRIM_STRICT = atoi(e);
  if (const char* e = getenv("NUC_FORCED_TRIM")) FORCED_TRIM = atoi(e);
  if (const char* e = getenv("NUC_STORED_BOUNDS")) STORED_BOUNDS = atoi(e);
  if (getenv("NUC_STATS")) STATS = 1;
  if (const char* e = getenv("NUC_EXP_TIE")) EXP_TIE = atoi(e);
  if (const char* e = getenv("NUC_EXP_BAND")) EXP_BAND = atoi(e);
  if (const char* e = getenv("NUC_EXP_BAND_SHIFT")) EXP_BAND_SHIFT = atoi(e);
  if (const char* e = getenv("NUC_OPEN")) OPEN_GAP_SCORE = atoi(e);
  if (const char* e = getenv("NUC_CONT")) CONT_GAP_SCORE = atoi(e);
  std::vector<Rec> ref = load_fasta(argv[1]), qry = load_fasta(argv[2]);
  RefIndex R;
  build_ref(ref, R);
  for (size_t qi = 0; qi < qry.size(); ++qi) {
    const std::string& fseq = qry[qi].seq;
    const std::string rseq = revcomp(fseq);
    const long qlen = (long)fseq.size() - 1;
    // clusters of this query record per reference record ("synteny"), both directions
    std::vector<std::vector<Cluster>> syn(ref.size());
    for (int dir = 0; dir < 2; ++dir) {
      std::vector<Mem> mem;
      find_mems(R, dir ? rseq : fseq, mem);
      select_mums(mem, maxmatch);
      std::vector<std::vector<Match>> cl;
      mgaps(mem, cl);
      for (auto& c : cl) {
        // re-map joined-reference coordinates to their record; a cluster crossing records is split (postnuc)
        long cur_rec = -1;
        for (auto& m : c) {
          long rec = (long)(std::upper_bound(R.start.begin(), R.start.end(), m.sA) - R.start.begin()) - 1;
          if (rec != cur_rec) { syn[rec].push_back(Cluster{false, dir ? '-' : '+', {}}); cur_rec = rec; }
          syn[rec].back().matches.push_back(Match{m.sA - R.start[rec] + 1, m.sB, m.len});
        }
      }
    }
    for (size_t ri = 0; ri < ref.size(); ++ri) {
      if (syn[ri].empty()) continue;
      const Seq A{ref[ri].seq.data(), (long)ref[ri].seq.size() - 1}, Bf{fseq.data(), qlen}, Br{rseq.data(), qlen};
      std::vector<Alignment> Al;
      extend_clusters(syn[ri], A, Bf, Br, Al);
      for (const Alignment& a : Al) {
        const Seq& B = a.dirB == '+' ? Bf : Br;
        long errors = 0, i = a.sA, j = a.sB;
        std::vector<long> dl;
        long run = 0;
        for (char op : a.ops) {
          ++run;
          if (op == 'M') { if (!same_base(A.at(i), B.at(j))) ++errors; ++i; ++j; }
          else if (op == 'I') { ++errors; ++i; dl.push_back(run); run = 0; }
          else { ++errors; ++j; dl.push_back(-run); run = 0; }
        }
        if (i != a.eA + 1 || j != a.eB + 1) { fprintf(stderr, "nucmer_oracle: path does not end at the alignment end (%ld %ld vs %ld %ld)\n", i - 1, j - 1, a.eA, a.eB); return 3; }
        const long qs = a.dirB == '+' ? a.sB : qlen - a.sB + 1, qe = a.dirB == '+' ? a.eB : qlen - a.eB + 1;
        printf("ALN %s %s %ld %ld %ld %ld %ld\n", ref[ri].id.c_str(), qry[qi].id.c_str(), a.sA, a.eA, qs, qe, errors);
        if (delta) { for (long d : dl) printf("%ld\n", d); printf("0\n"); }
      }
    }
  }
  if (STATS) {
    fprintf(stderr, "STATS max band cells %ld, mean %.1f over %ld anti-diagonals, forced cells %ld (%.1f %% of search cells)\nSTATS width histogram (cells/8):", stat_maxw, (double)stat_cells / (double)std::max(1L, stat_diags), stat_diags, stat_forced_cells, 100.0 * stat_forced_cells / std::max(1L, stat_cells));
    for (int b = 0; b < 64; ++b) if (stat_hist[b]) fprintf(stderr, " %d:%ld", b * 8, stat_hist[b]);
    fprintf(stderr, "\n");
  }
  return 0;
}
