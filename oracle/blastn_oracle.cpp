// oracle/blastn_oracle.cpp — INDEPENDENT CPU restatement of the search pyani's ANIb delegates to BLAST+.
// TEST INFRASTRUCTURE ONLY: nothing under pyani_amd/ builds, loads or calls this file, and this file includes NOTHING from
// pyani_amd/csrc/ (the product's own host statement is oracle/anib_cpu.cpp; this one shares no line with it).
//
// What it restates.  pyani runs (pyani/anib.py:451-471)
//     blastn -task blastn -query <1020-nt fragments of Q> -db <S> -xdrop_gap_final 150 -dust no -evalue 1e-15 -max_target_seqs 1
//            -outfmt '6 qseqid sseqid length mismatch pident nident qlen slen qstart qend sstart send positive ppos gaps'
// and reads the table with parse_blast_tab (anib.py:569-667).  BLAST+ is a third-party dependency (conda pin in the reference:
// blast >= 2.9; NCBI C++ toolkit, algo/blast/core), ABSENT from /root/reference and from the image — it can be neither built nor
// run here.  This file restates the published algorithm of its core for exactly that command line, scalar and slow:
//
//   word finding      every exact word of 11 bases (-task blastn) of either query strand against the subject; one hit per
//                     diagonal stretch: a hit that starts inside the stretch already explored on its diagonal is dropped
//                     (na_ungapped.c, the diagonal table of the one-hit mode: window_size 0)
//   ungapped stage    X-drop extension left of the word and right from its first base, reward 2 / penalty -3,
//                     X = ceil(20 bits * ln2 / lambda_u) = 22 with the ungapped Karlin-Altschul lambda_u = 0.634 (K 0.408) of 2 / -3
//                     on uniform base frequencies; kept as an initial HSP when its score reaches the gap trigger
//                     floor((27 bits * ln2 + ln K_u) / lambda_u) = 28   (blast_parameters.c)
//   preliminary gapped  initial HSPs best ungapped score first; one contained in an earlier gapped HSP of the same strand (both
//                     ends inside its box and one end within 50 diagonals: min_diag_separation of blastn) is skipped; the others
//                     are extended from a point inside their word — (q_off, s_off) moved to the next 4-base boundary of the
//                     subject, 1..4 bases — to the left and to the right with the score-only dynamic programme of
//                     Blast_SemiGappedAlign (blast_gapalign.c): row by row over the query, a window [first_b, b_size) of live
//                     subject columns, a cell is dropped when it falls more than X below the best score seen SO FAR in
//                     row-major order, affine gaps 5 + 2k, X = floor(30 bits * ln2 / 0.625) = 33; kept when the score reaches the
//                     cut-off of e-value 1e-15 on the effective search space
//   traceback         kept HSPs best score first, again skipping contained ones, re-aligned from the same point with
//                     X = floor(150 bits * ln2 / 0.625) = 166 (-xdrop_gap_final 150) by ALIGN_EX: the same programme with a
//                     traceback (ties: substitution, then the gap that consumes a query base, then the gap that consumes a
//                     subject base; an open gap is EXTENDED on ties)
//   culling           HSPs with a common start or a common end on the same strand: the better one stays; e-value
//                     = searchsp * K * exp(-lambda * S) with the gapped parameters of 2 / -3 / 5 / 2 (lambda 0.625, K 0.41,
//                     alpha 0.8, beta -2) and BLAST's length adjustment (BLAST_ComputeLengthAdjustment); <= 1e-15 stays
//   order             e-value, then score descending, subject start ascending, subject end descending, query start ascending
//   -max_target_seqs 1  only the subject sequence (FASTA record) holding the fragment's best HSP is reported
//
// Not restated (documented, their effect is part of the measured agreement — profiles/r06_blastn_oracle_vs_blastplus.json):
// the split of subject sequences longer than 5 000 000 bases into overlapping chunks, the 4-bases-at-a-time approximate ungapped
// pre-extension (an initial HSP whose approximate score is below 0.9 * the trigger is dropped there), ambiguity codes (N scores as
// a mismatch here; the fixtures contain none), the lookup-table width (it changes no result: every exact 11-mer is found).
//
// PINNED on the 12 BLAST+ tables the reference's tests hold (tests/fixtures/anib/blastn/*.blast_tab, committed as data under
// tests/golden/anib/): tests/test_blastn_oracle.py compares rows, the row-level agreement per table is in
// profiles/r06_blastn_oracle_vs_blastplus.json.
//
//   g++ -O2 -std=c++17 -pthread -fPIC -shared oracle/blastn_oracle.cpp -o oracle/libblastnoracle.so
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---- parameters of the command line ------------------------------------------------------------------------------------------
constexpr int WORD = 11;
constexpr int REWARD = 2, PENALTY = -3, GAP_OPEN = 5, GAP_EXTEND = 2;
constexpr double LAMBDA_U = 0.634, K_U = 0.408;                 // ungapped Karlin-Altschul parameters of 2 / -3
constexpr double LAMBDA_G = 0.625, K_G = 0.41, ALPHA_G = 0.8, BETA_G = -2.0;   // gapped: 2 / -3 with gap costs 5 / 2
constexpr double EVALUE = 1e-15;
constexpr int MIN_DIAG_SEPARATION = 50;
constexpr int MININT = -(1 << 30);
const double LN2 = 0.69314718055994530941723212145818;

int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

struct Params {
  int x_ungapped, trigger, x_prelim, x_final;
  Params() {
    x_ungapped = env_int("BLASTN_ORACLE_X_UNGAPPED", (int)std::ceil(20.0 * LN2 / LAMBDA_U));
    trigger = env_int("BLASTN_ORACLE_TRIGGER", (int)((27.0 * LN2 + std::log(K_U)) / LAMBDA_U));
    x_prelim = env_int("BLASTN_ORACLE_X_PRELIM", (int)(30.0 * LN2 / LAMBDA_G));
    x_final = env_int("BLASTN_ORACLE_X_FINAL", (int)(150.0 * LN2 / LAMBDA_G));
  }
};

// ---- sequences -----------------------------------------------------------------------------------------------------------------
inline uint8_t code_of(uint8_t c) {
  switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; }
  return 4;
}
inline int sub_score(uint8_t a, uint8_t b) { return (a < 4 && a == b) ? REWARD : PENALTY; }

struct Subject {                 // all records of the subject genome, coded, with an index of every exact 11-mer
  std::vector<uint8_t> code;     // records back to back
  std::vector<int64_t> rec_off;  // n_rec + 1 offsets into code
  std::vector<uint32_t> start;   // CSR over the 4^11 words
  std::vector<int32_t> pos;      // word start, position in `code`
  int rec_of(int64_t p) const { return (int)(std::upper_bound(rec_off.begin(), rec_off.end(), p) - rec_off.begin()) - 1; }
};

void build_index(Subject& S) {
  const uint32_t NB = 1u << (2 * WORD);
  S.start.assign((size_t)NB + 1, 0);
  for (int pass = 0; pass < 2; ++pass) {
    std::vector<uint32_t> fill;
    if (pass) {
      for (size_t b = 0; b < NB; ++b) S.start[b + 1] += S.start[b];
      S.pos.assign(S.start[NB], 0);
      fill.assign(S.start.begin(), S.start.end() - 1);
    }
    for (size_t r = 0; r + 1 < S.rec_off.size(); ++r) {
      uint32_t v = 0; int run = 0;
      for (int64_t p = S.rec_off[r]; p < S.rec_off[r + 1]; ++p) {
        const uint8_t c = S.code[p];
        if (c > 3) { run = 0; v = 0; continue; }
        v = ((v << 2) | c) & (NB - 1);
        if (++run >= WORD) { if (pass) S.pos[fill[v]++] = (int32_t)(p - WORD + 1); else ++S.start[v + 1]; }
      }
    }
  }
}

// ---- statistics ----------------------------------------------------------------------------------------------------------------
// BLAST_ComputeLengthAdjustment (blast_stat.c): the fixed point of  l = alpha/lambda * ln(K (m - l)(n - N l)) + beta
int length_adjustment(double K, double logK, double alpha_d_lambda, double beta, int query_length, int64_t db_length, int db_num_seqs) {
  const int kMaxIterations = 20;
  const double m = query_length, n = (double)db_length, N = db_num_seqs;
  double ell, ss, ell_min = 0, ell_max, ell_next = 0;
  bool converged = false;
  {
    const double a = N, mb = m * N + n, c = n * m - std::max(m, n) / K;
    if (c < 0) return 0;
    ell_max = 2 * c / (mb + std::sqrt(mb * mb - 4 * a * c));
  }
  for (int i = 1; i <= kMaxIterations; i++) {
    ell = ell_next;
    ss = (m - ell) * (n - N * ell);
    const double ell_bar = alpha_d_lambda * (logK + std::log(ss)) + beta;
    if (ell_bar >= ell) {
      ell_min = ell;
      if (ell_bar - ell_min <= 1.0) { converged = true; break; }
      if (ell_min == ell_max) break;
    } else {
      ell_max = ell;
    }
    if (ell_min <= ell_bar && ell_bar <= ell_max) ell_next = ell_bar;
    else ell_next = (i == 1) ? ell_max : (ell_min + ell_max) / 2;
  }
  int adj = (int)ell_min;
  if (converged) {
    ell = std::ceil(ell_min);
    if (ell <= ell_max) {
      ss = (m - ell) * (n - N * ell);
      if (alpha_d_lambda * (logK + std::log(ss)) + beta >= ell) adj = (int)ell;
    }
  }
  return adj;
}

double search_space(int qlen, int64_t db_length, int db_num_seqs) {
  const int adj = length_adjustment(K_G, std::log(K_G), ALPHA_G / LAMBDA_G, BETA_G, qlen, db_length, db_num_seqs);
  int64_t eff_db = db_length - (int64_t)db_num_seqs * adj;
  if (eff_db <= 0) eff_db = 1;
  int eff_q = qlen - adj;
  if (eff_q <= 0) eff_q = 1;
  return (double)eff_db * (double)eff_q;
}
inline double evalue_of(int score, double searchsp) { return searchsp * std::exp(-LAMBDA_G * score + std::log(K_G)); }
int cutoff_score_of(double searchsp) {      // BLAST_Cutoffs: the smallest score whose e-value is <= EVALUE
  int s = (int)std::ceil(std::log(K_G * searchsp / EVALUE) / LAMBDA_G);
  return s < 1 ? 1 : s;
}

// ---- the gapped X-drop programme (Blast_SemiGappedAlign / ALIGN_EX) -------------------------------------------------------------
// Row a = 1..M consumes A(a), column b = 1..N consumes B(b); cell (0, 0) = the start point, score 0.  Returns the best score and its
// cell.  With `ops` the path from (0, 0) to that cell is appended as one byte per column, in path order FROM THE BEST CELL BACK TO
// THE START: 0 = substitution, 1 = gap consuming a subject base (b only), 2 = gap consuming a query base (a only).
struct GapCell { int best, best_gap; };
enum { SCRIPT_SUB = 0, SCRIPT_GAP_IN_A = 1, SCRIPT_GAP_IN_B = 2, SCRIPT_OP_MASK = 3, SCRIPT_EXTEND_GAP_A = 4, SCRIPT_EXTEND_GAP_B = 8 };

template <typename FA, typename FB>
int semi_gapped_align(FA&& A, int M, FB&& B, int N, int x_dropoff, int* a_offset, int* b_offset, std::vector<uint8_t>* ops,
                      bool stale_gap_quirk) {
  const int gap_open_extend = GAP_OPEN + GAP_EXTEND, gap_extend = GAP_EXTEND;
  *a_offset = 0; *b_offset = 0;
  if (x_dropoff < gap_open_extend) x_dropoff = gap_open_extend;
  if (N <= 0 || M <= 0) return 0;
  // a cell (a, b) scores at most 2 min(a, b) - 5 - 2 |b - a| and dies X below a best that is never negative: columns beyond
  // 2 M + X cannot live, whatever the subject holds there (BLAST grows its window on demand; this is only an allocation bound)
  if (N > 2 * M + x_dropoff + 8) N = 2 * M + x_dropoff + 8;
  std::vector<GapCell> score_array((size_t)N + 4);
  std::vector<std::vector<uint8_t>> script;      // per row: scripts of columns [row_start[a], ...)
  std::vector<int> row_start;
  const bool trace = ops != nullptr;
  int score = -gap_open_extend;
  score_array[0].best = 0;
  score_array[0].best_gap = -gap_open_extend;
  int i;
  for (i = 1; i <= N; i++) {
    if (score < -x_dropoff) break;
    score_array[i].best = score;
    score_array[i].best_gap = score - gap_open_extend;
    score -= gap_extend;
  }
  int b_size = i, best_score = 0, first_b_index = 0;
  if (trace) { script.emplace_back((size_t)b_size, (uint8_t)SCRIPT_GAP_IN_A); row_start.push_back(0); }
  for (int a_index = 1; a_index <= M; a_index++) {
    const uint8_t a_base = A(a_index);
    score = MININT;
    int score_gap_row = MININT, last_b_index = first_b_index;
    if (trace) { script.emplace_back(); row_start.push_back(first_b_index); script.back().reserve((size_t)(b_size - first_b_index) + 8); }
    const int row_first = first_b_index;
    for (int b_index = row_first; b_index < b_size; b_index++) {
      int score_gap_col = score_array[b_index].best_gap;
      const int next_score = score_array[b_index].best + (b_index + 1 <= N ? sub_score(a_base, B(b_index + 1)) : PENALTY);
      uint8_t sc = SCRIPT_SUB;
      if (score < score_gap_col) { sc = SCRIPT_GAP_IN_B; score = score_gap_col; }
      if (score < score_gap_row) { sc = SCRIPT_GAP_IN_A; score = score_gap_row; }
      if (best_score - score > x_dropoff) {
        if (b_index == first_b_index) first_b_index++;
        else {
          score_array[b_index].best = MININT;
          if (!stale_gap_quirk) score_array[b_index].best_gap = MININT;
        }
      } else {
        last_b_index = b_index;
        if (score > best_score) { best_score = score; *a_offset = a_index; *b_offset = b_index; }
        score_gap_row -= gap_extend;
        score_gap_col -= gap_extend;
        if (score_gap_col < score - gap_open_extend) score_array[b_index].best_gap = score - gap_open_extend;
        else { score_array[b_index].best_gap = score_gap_col; sc += SCRIPT_EXTEND_GAP_B; }
        if (score_gap_row < score - gap_open_extend) score_gap_row = score - gap_open_extend;
        else sc += SCRIPT_EXTEND_GAP_A;
        score_array[b_index].best = score;
      }
      score = next_score;
      if (trace) script.back().push_back(sc);
    }
    if (first_b_index == b_size) break;
    if (last_b_index < b_size - 1) {
      b_size = last_b_index + 1;
    } else {
      while (score_gap_row >= best_score - x_dropoff && b_size <= N) {
        score_array[b_size].best = score_gap_row;
        score_array[b_size].best_gap = score_gap_row - gap_open_extend;
        score_gap_row -= gap_extend;
        if (trace) script.back().push_back((uint8_t)SCRIPT_GAP_IN_A);
        b_size++;
      }
    }
    if (b_size <= N) {
      score_array[b_size].best = MININT;
      score_array[b_size].best_gap = MININT;
      b_size++;
    }
  }
  if (trace) {
    int a_index = *a_offset, b_index = *b_offset;
    uint8_t sc = SCRIPT_SUB;
    while (a_index > 0 || b_index > 0) {
      const uint8_t next = script[a_index][b_index - row_start[a_index]];
      switch (sc) {
        case SCRIPT_GAP_IN_A: sc = next & SCRIPT_OP_MASK; if (next & SCRIPT_EXTEND_GAP_A) sc = SCRIPT_GAP_IN_A; break;
        case SCRIPT_GAP_IN_B: sc = next & SCRIPT_OP_MASK; if (next & SCRIPT_EXTEND_GAP_B) sc = SCRIPT_GAP_IN_B; break;
        default: sc = next & SCRIPT_OP_MASK; break;
      }
      if (sc == SCRIPT_GAP_IN_A) b_index--;
      else if (sc == SCRIPT_GAP_IN_B) a_index--;
      else { a_index--; b_index--; }
      ops->push_back(sc);
    }
  }
  return best_score;
}

// ---- one fragment ----------------------------------------------------------------------------------------------------------------
struct InitHsp { int ctx; int q_off, s_off; int q_start, s_start, length, score; };   // s_* relative to the subject record
struct Hsp {
  int ctx, score;
  int q0, q1, s0, s1;                 // half-open, query on the searched strand, subject relative to its record
  int gq, gs;                         // the point the gapped alignment was grown from
  int length, mismatch, gaps, nident;
  double evalue;
  std::vector<uint8_t> ops;           // edit script, one byte per column from (q0, s0) to (q1, s1): SCRIPT_SUB / SCRIPT_GAP_IN_A / SCRIPT_GAP_IN_B
};
struct Row { int32_t frag, length, mismatch, gaps, nident, qlen, qstart, qend, sstart, send, srec, score; };

inline bool contained_in(int q, int s, const Hsp& t) { return q >= t.q0 && q <= t.q1 && s >= t.s0 && s <= t.s1; }
// s_HSPIsContained of blast_itree.c: both ends of `in` inside the box of the better `t`, and one end on a near diagonal
bool hsp_contained(int ctx, int score, int q0, int q1, int s0, int s1, const Hsp& t) {
  if (ctx != t.ctx || score > t.score) return false;
  if (!contained_in(q0, s0, t) || !contained_in(q1, s1, t)) return false;
  const auto close = [](int qa, int sa, int qb, int sb) { return std::abs((qa - sa) - (qb - sb)) < MIN_DIAG_SEPARATION; };
  return close(t.q0, t.s0, q0, s0) || close(t.q1, t.s1, q1, s1);
}

// BlastGetStartForGappedAlignmentNucl (blast_gapalign.c): the traceback keeps the preliminary start point when it sits in a run of
// more than RUN_OK identities; otherwise it moves to the first run of more than 1.5 RUN_OK identities on the start point's diagonal
// inside the preliminary HSP (to its middle), or to the middle of the longest run there.
void nucl_gapped_start(const uint8_t* q, const uint8_t* sb, const Hsp& h, int* gq, int* gs) {
  static const int RUN_OK = env_int("BLASTN_ORACLE_RUN_OK", 20);
  int max_run = RUN_OK;
  int score = -1;
  for (int qi = *gq, si = *gs; qi < h.q1 && q[qi] < 4 && q[qi] == sb[si]; ++qi, ++si) { if (++score > max_run) return; }
  for (int qi = *gq, si = *gs; qi >= 0 && si >= 0 && q[qi] < 4 && q[qi] == sb[si]; --qi, --si) { if (++score > max_run) return; }
  static const int RUN2 = env_int("BLASTN_ORACLE_RUN2", 20);
  max_run = RUN2;
  const int offset = std::min(*gs - h.s0, *gq - h.q0);
  const int q_start = *gq - offset, s_start = *gs - offset;
  const int q_len = std::min(h.s1 - s_start, h.q1 - q_start);
  int max_score = 0, max_offset = q_start;
  score = 0;
  bool match = false, prev_match = false;
  int index;
  for (index = q_start; index < q_start + q_len; index++) {
    match = q[index] < 4 && q[index] == sb[s_start + (index - q_start)];
    if (match != prev_match) {
      prev_match = match;
      if (match) score = 1;
      else if (score > max_score) { max_score = score; max_offset = index - score / 2; }
    } else if (match) {
      if (++score > max_run) { max_offset = index - max_run / 2; *gq = max_offset; *gs = max_offset + s_start - q_start; return; }
    }
  }
  if (match && score > max_score) { max_score = score; max_offset = index - score / 2; }
  if (max_score > 0) { *gq = max_offset; *gs = max_offset + s_start - q_start; }
}

struct Options { bool stale_gap_quirk = true; bool approx_prefilter = false; bool all_hsps = true; bool reevaluate = false; bool cut_common = true; bool recheck_contained = true; };

void count_columns(Hsp& h, const uint8_t* q, const uint8_t* sb) {
  h.length = h.mismatch = h.gaps = h.nident = 0;
  int a = h.q0, b = h.s0;
  for (uint8_t op : h.ops) {
    if (op == SCRIPT_SUB) { if (q[a] < 4 && q[a] == sb[b]) ++h.nident; else ++h.mismatch; ++a; ++b; }
    else if (op == SCRIPT_GAP_IN_A) { ++h.gaps; ++b; }
    else { ++h.gaps; ++a; }
    ++h.length;
  }
}

// Blast_HSPReevaluateWithAmbiguitiesGapped (blast_hits.c), run on every blastn HSP after its traceback: one pass over the edit script
// from the left, substitutions one at a time and a gap as a whole; when the running sum falls below zero the alignment restarts behind
// that point (what was found before is forgotten unless it had reached the cut-off), the best-scoring stretch is kept and then grown
// over exact matches at both ends.  false = the HSP is dropped (its score is below the cut-off).
bool reevaluate(Hsp& h, const uint8_t* q, int qlen, const uint8_t* sb, int slen, int cutoff) {
  int sum = 0, score = 0;
  int qa = h.q0, sa = h.s0;                                  // running position
  size_t cur_start = 0, best_start = 0, best_end = 0;        // indices into ops: [best_start, best_end)
  int cur_q = qa, cur_s = sa, best_q0 = qa, best_s0 = sa, best_q1 = qa, best_s1 = sa;
  size_t i = 0;
  const size_t n = h.ops.size();
  while (i < n) {
    const uint8_t op = h.ops[i];
    if (op == SCRIPT_SUB) { sum += sub_score(q[qa], sb[sa]); ++qa; ++sa; ++i; }
    else {
      size_t j = i;
      while (j < n && h.ops[j] == op) ++j;
      const int len = (int)(j - i);
      sum -= GAP_OPEN + GAP_EXTEND * len;
      if (op == SCRIPT_GAP_IN_A) sa += len; else qa += len;
      i = j;
    }
    if (sum < 0) {
      sum = 0;
      cur_start = i; cur_q = qa; cur_s = sa;
      if (score < cutoff) { best_start = best_end = i; best_q0 = best_q1 = qa; best_s0 = best_s1 = sa; score = 0; }
    } else if (sum > score) {
      score = sum;
      best_start = cur_start; best_q0 = cur_q; best_s0 = cur_s;
      best_end = i; best_q1 = qa; best_s1 = sa;
    }
  }
  if (best_end <= best_start) return false;
  std::vector<uint8_t> ops(h.ops.begin() + best_start, h.ops.begin() + best_end);
  // "try to extend further": exact matches beyond both ends
  int ext = 0;
  while (best_q0 - ext > 0 && best_s0 - ext > 0 && q[best_q0 - ext - 1] < 4 && q[best_q0 - ext - 1] == sb[best_s0 - ext - 1]) ++ext;
  if (ext) { ops.insert(ops.begin(), (size_t)ext, (uint8_t)SCRIPT_SUB); best_q0 -= ext; best_s0 -= ext; score += ext * REWARD; }
  ext = 0;
  while (best_q1 + ext < qlen && best_s1 + ext < slen && q[best_q1 + ext] < 4 && q[best_q1 + ext] == sb[best_s1 + ext]) ++ext;
  if (ext) { ops.insert(ops.end(), (size_t)ext, (uint8_t)SCRIPT_SUB); best_q1 += ext; best_s1 += ext; score += ext * REWARD; }
  h.ops.swap(ops);
  h.q0 = best_q0; h.s0 = best_s0; h.q1 = best_q1; h.s1 = best_s1; h.score = score;
  return score >= cutoff;
}

// s_CutOffGapEditScript (blast_hits.c): walk the script from the left, substitutions one at a time and a gap as a whole, to the
// first point that has consumed q_cut query AND s_cut subject bases; cut_begin: the HSP starts there, else it ends there.
bool cut_script(Hsp& h, int q_cut_abs, int s_cut_abs, bool cut_begin) {
  const int q_cut = q_cut_abs - h.q0, s_cut = s_cut_abs - h.s0;
  int qid = 0, sid = 0;
  size_t i = 0;
  const size_t n = h.ops.size();
  bool found = false;
  while (i < n) {
    const uint8_t op = h.ops[i];
    if (op == SCRIPT_SUB) { ++qid; ++sid; ++i; }
    else {
      size_t j = i;
      while (j < n && h.ops[j] == op) ++j;
      if (op == SCRIPT_GAP_IN_A) sid += (int)(j - i); else qid += (int)(j - i);
      i = j;
    }
    if (qid >= q_cut && sid >= s_cut) { found = true; break; }
  }
  if (!found) return true;                 // (left as it is)
  if (cut_begin) { h.ops.erase(h.ops.begin(), h.ops.begin() + i); h.q0 += qid; h.s0 += sid; }
  else { h.ops.resize(i); h.q1 = h.q0 + qid; h.s1 = h.s0 + sid; }
  return !h.ops.empty();
}

// HSPs of one fragment (both strands) against ONE subject record, in BLAST's output order
void search_record(const uint8_t* frag_fwd, int qlen, const Subject& S, int rec, const std::vector<std::pair<int, int32_t>>* hits,
                   const Params& P, const Options& opt, double searchsp, std::vector<Hsp>& out) {
  const int64_t r0 = S.rec_off[rec];
  const int slen = (int)(S.rec_off[rec + 1] - r0);
  const uint8_t* sb = S.code.data() + r0;
  std::vector<uint8_t> qctx[2];
  qctx[0].assign(frag_fwd, frag_fwd + qlen);
  qctx[1].resize(qlen);
  for (int p = 0; p < qlen; ++p) { const uint8_t c = frag_fwd[qlen - 1 - p]; qctx[1][p] = c < 4 ? 3 - c : 4; }
  const int cutoff = cutoff_score_of(searchsp);

  // --- word hits -> ungapped initial HSPs (per strand, per diagonal, left to right)
  std::vector<InitHsp> init;
  for (int ctx = 0; ctx < 2; ++ctx) {
    const uint8_t* q = qctx[ctx].data();
    std::vector<std::pair<int, int>> h;                 // (diagonal s - q, s) of every exact 11-mer
    for (const auto& e : hits[ctx]) {
      const int64_t p = e.second;
      if (p < r0 || p + WORD > r0 + slen) continue;
      const int s = (int)(p - r0);
      h.push_back({s - e.first, s});
    }
    std::sort(h.begin(), h.end());
    size_t k = 0;
    while (k < h.size()) {
      const int diag = h[k].first;
      int last_hit = -1;
      while (k < h.size() && h[k].first == diag) {
        // a maximal exact run: consecutive words on this diagonal
        const int s_a = h[k].second;
        size_t e = k + 1;
        while (e < h.size() && h[e].first == diag && h[e].second == h[e - 1].second + 1) ++e;
        k = e;
        if (s_a < last_hit) continue;
        const int q_a = s_a - diag;
        // s_NuclUngappedExtendExact: left of the word, then right from its first base
        const int X = -P.x_ungapped;
        int score = 0, sum = 0, q_beg = q_a;
        for (int qi = q_a - 1, si = s_a - 1; qi >= 0 && si >= 0; --qi, --si) {
          sum += sub_score(q[qi], sb[si]);
          if (sum > 0) { q_beg = qi; score += sum; sum = 0; }
          else if (sum < X) break;
        }
        int q_end = q_a;
        sum = 0;
        for (int qi = q_a, si = s_a; qi < qlen && si < slen; ++qi, ++si) {
          sum += sub_score(q[qi], sb[si]);
          if (sum > 0) { q_end = qi + 1; score += sum; sum = 0; }
          else if (sum < X) break;
        }
        const int len = q_end - q_beg, s_beg = s_a - (q_a - q_beg);
        last_hit = s_beg + len;
        if (last_hit < s_a + WORD) last_hit = s_a + WORD;
        static const int MIN_RUN = env_int("BLASTN_ORACLE_EXP_MIN_RUN", 0);
        bool has_run = MIN_RUN == 0;
        if (MIN_RUN) { int run = 0; for (int t = 0; t < len; ++t) { run = (q[q_beg + t] < 4 && q[q_beg + t] == sb[s_beg + t]) ? run + 1 : 0; if (run >= MIN_RUN) has_run = true; } }
        if (score >= P.trigger && has_run) init.push_back(InitHsp{ctx, q_a, s_a, q_beg, s_beg, len, score});
      }
    }
  }
  if (init.empty()) return;
  std::sort(init.begin(), init.end(), [qlen](const InitHsp& x, const InitHsp& y) {
    if (x.score != y.score) return x.score > y.score;
    if (x.s_start != y.s_start) return x.s_start < y.s_start;
    if (x.length != y.length) return x.length > y.length;
    return x.ctx * qlen + x.q_start < y.ctx * qlen + y.q_start;
  });

  const bool debug = getenv("BLASTN_ORACLE_DEBUG") != nullptr;
  if (debug) {
    for (const InitHsp& ih : init) {
      const uint8_t* q = qctx[ih.ctx].data();
      const int scan = env_int("BLASTN_ORACLE_DEBUG_SCAN", 0);
      for (int adjv = 0; adjv <= (scan ? ih.length : 4); ++adjv) {
        const int adj = scan ? (ih.q_start - ih.q_off) + adjv : (adjv ? adjv : 4 - (ih.s_off % 4));
        const int gq = ih.q_off + adj, gs = ih.s_off + adj;
        if (scan && (gq >= qlen || q[gq] != sb[gs])) continue;
        int la = 0, lb = 0, ra = 0, rb = 0;
        std::vector<uint8_t> lops, rops;
        const int sl = semi_gapped_align([&](int a) { return q[gq - a]; }, gq, [&](int b) { return sb[gs - b]; }, gs, P.x_final, &la, &lb, &lops, opt.stale_gap_quirk);
        const int sr = semi_gapped_align([&](int a) { return q[gq + a - 1]; }, qlen - gq, [&](int b) { return sb[gs + b - 1]; }, slen - gs, P.x_final, &ra, &rb, &rops, opt.stale_gap_quirk);
        int mm = 0, gp = 0, len = 0;
        { int a = la, b = lb; for (uint8_t op : lops) { if (op == SCRIPT_SUB) { mm += !(q[gq - a] == sb[gs - b]); --a; --b; } else if (op == SCRIPT_GAP_IN_A) { ++gp; --b; } else { ++gp; --a; } ++len; }
          a = ra; b = rb; for (uint8_t op : rops) { if (op == SCRIPT_SUB) { mm += !(q[gq + a - 1] == sb[gs + b - 1]); --a; --b; } else if (op == SCRIPT_GAP_IN_A) { ++gp; --b; } else { ++gp; --a; } ++len; } }
        fprintf(stderr, "init ctx %d word (%d,%d) ungapped [%d,+%d) score %d | adj %d%s -> q [%d,%d) s [%d,%d) score %d len %d mm %d gaps %d\n", ih.ctx, ih.q_off,
                ih.s_off, ih.q_start, ih.length, ih.score, adj, adjv ? "" : "*", gq - la, gq + ra, gs - lb, gs + rb, sl + sr, len, mm, gp);
      }
    }
  }
  // --- preliminary gapped extension
  std::vector<Hsp> prelim;
  for (const InitHsp& ih : init) {
    bool skip = false;
    for (const Hsp& t : prelim)
      if (hsp_contained(ih.ctx, ih.score, ih.q_start, ih.q_start + ih.length, ih.s_start, ih.s_start + ih.length, t)) { skip = true; break; }
    if (skip) continue;
    const uint8_t* q = qctx[ih.ctx].data();
    const int adj = 4 - (ih.s_off % 4);
    int gq = ih.q_off + adj, gs = ih.s_off + adj;
    static const int START_MODE = env_int("BLASTN_ORACLE_START_MODE", 0);
    if (START_MODE == 1) { gq = ih.q_start + ih.length / 2; gs = ih.s_start + ih.length / 2; }
    if (START_MODE == 2) { gq = ih.q_off; gs = ih.s_off; }
    if (START_MODE == 3) { const int m = ih.length / 2; const int a2 = 4 - ((ih.s_start + m) % 4); gq = ih.q_start + m + a2; gs = ih.s_start + m + a2; }
    if (gq > qlen || gs > slen) { gq = ih.q_off; gs = ih.s_off; }
    int la = 0, lb = 0, ra = 0, rb = 0;
    const int sl = semi_gapped_align([&](int a) { return q[gq - a]; }, gq, [&](int b) { return sb[gs - b]; }, gs, P.x_prelim, &la, &lb, nullptr,
                                     opt.stale_gap_quirk);
    int sr = 0;
    if (gq < qlen && gs < slen)
      sr = semi_gapped_align([&](int a) { return q[gq + a - 1]; }, qlen - gq, [&](int b) { return sb[gs + b - 1]; }, slen - gs, P.x_prelim, &ra, &rb,
                             nullptr, opt.stale_gap_quirk);
    Hsp hs{};
    hs.ctx = ih.ctx; hs.score = sl + sr;
    hs.q0 = gq - la; hs.s0 = gs - lb; hs.q1 = gq + ra; hs.s1 = gs + rb;
    hs.gq = gq; hs.gs = gs;
    if (debug) fprintf(stderr, "prelim ctx %d word (%d,%d) start (%d,%d) -> q [%d,%d) s [%d,%d) score %d (cutoff %d)\n", ih.ctx, ih.q_off, ih.s_off, gq, gs, hs.q0, hs.q1, hs.s0, hs.s1, hs.score, cutoff);
    if (hs.score >= cutoff) prelim.push_back(hs);
  }
  if (prelim.empty()) return;
  auto score_order = [qlen](const Hsp& x, const Hsp& y) {
    if (x.score != y.score) return x.score > y.score;
    if (x.s0 != y.s0) return x.s0 < y.s0;
    if (x.s1 != y.s1) return x.s1 > y.s1;
    const int xq = x.ctx * qlen + x.q0, yq = y.ctx * qlen + y.q0;
    if (xq != yq) return xq < yq;
    return x.q1 > y.q1;
  };
  // purge HSPs with common start / end points (same strand): the better one stays
  auto purge = [&](std::vector<Hsp>& v) {
    for (int pass = 0; pass < 2; ++pass) {
      std::stable_sort(v.begin(), v.end(), [&](const Hsp& x, const Hsp& y) {
        if (x.ctx != y.ctx) return x.ctx < y.ctx;
        const int xq = pass ? x.q1 : x.q0, yq = pass ? y.q1 : y.q0, xs = pass ? x.s1 : x.s0, ys = pass ? y.s1 : y.s0;
        if (xq != yq) return xq < yq;
        if (xs != ys) return xs < ys;
        return x.score > y.score;
      });
      std::vector<Hsp> keep;
      for (const Hsp& h : v) {
        if (!keep.empty()) {
          const Hsp& p = keep.back();
          if (p.ctx == h.ctx && (pass ? (p.q1 == h.q1 && p.s1 == h.s1) : (p.q0 == h.q0 && p.s0 == h.s0))) continue;
        }
        keep.push_back(h);
      }
      v.swap(keep);
    }
  };
  purge(prelim);
  std::stable_sort(prelim.begin(), prelim.end(), score_order);

  // --- traceback with the final X
  std::vector<Hsp> fin;
  for (const Hsp& ph : prelim) {
    bool skip = false;
    for (const Hsp& t : fin)
      if (hsp_contained(ph.ctx, ph.score, ph.q0, ph.q1, ph.s0, ph.s1, t)) { skip = true; break; }
    if (skip) continue;
    const uint8_t* q = qctx[ph.ctx].data();
    int gq = ph.gq, gs = ph.gs;
    static const int NO_PRELIM_EXTENTS = env_int("BLASTN_ORACLE_EXP_NO_PRELIM_EXTENTS", 0);
    if (NO_PRELIM_EXTENTS) { Hsp whole = ph; const int off = std::min(gq, gs); whole.q0 = gq - off; whole.s0 = gs - off; const int room = std::min(qlen - gq, slen - gs); whole.q1 = gq + room; whole.s1 = gs + room; nucl_gapped_start(q, sb, whole, &gq, &gs); }
    else nucl_gapped_start(q, sb, ph, &gq, &gs);
    if (debug && env_int("BLASTN_ORACLE_DEBUG_SCAN", 0) == 2 && ph.score > 300) {
      const int off = std::min(ph.gs - ph.s0, ph.gq - ph.q0);
      const int qs0 = ph.gq - off, ss0 = ph.gs - off, n = std::min(ph.s1 - ss0, ph.q1 - qs0);
      fprintf(stderr, "SCAN prelim ctx %d q [%d,%d) score %d prelim start %d chosen %d; runs>=8 on the diagonal (q:len):", ph.ctx, ph.q0, ph.q1, ph.score, ph.gq, gq);
      for (int i = 0; i < n;) { int j = i; while (j < n && q[qs0 + j] == sb[ss0 + j]) ++j; if (j - i >= 8) fprintf(stderr, " %d:%d", qs0 + i, j - i); i = j + 1; }
      fprintf(stderr, "\n");
      int last_mm = -1, last_gp = -1, last_len = -1, from = -1;
      for (int i = 0; i <= n; ++i) {
        int mm = -2, gp = -2, len = -2;
        if (i < n && q[qs0 + i] == sb[ss0 + i]) {
          const int tq = qs0 + i, ts = ss0 + i;
          int la2 = 0, lb2 = 0, ra2 = 0, rb2 = 0;
          std::vector<uint8_t> lo, ro;
          semi_gapped_align([&](int a) { return q[tq - a]; }, tq, [&](int b) { return sb[ts - b]; }, ts, P.x_final, &la2, &lb2, &lo, opt.stale_gap_quirk);
          semi_gapped_align([&](int a) { return q[tq + a - 1]; }, qlen - tq, [&](int b) { return sb[ts + b - 1]; }, slen - ts, P.x_final, &ra2, &rb2, &ro, opt.stale_gap_quirk);
          Hsp t{}; t.q0 = tq - la2; t.s0 = ts - lb2; t.q1 = tq + ra2; t.s1 = ts + rb2; t.ops = lo; t.ops.insert(t.ops.end(), ro.rbegin(), ro.rend());
          count_columns(t, q, sb);
          mm = t.mismatch; gp = t.gaps; len = t.length;
        } else if (i < n) continue;
        if (mm != last_mm || gp != last_gp || len != last_len) {
          if (from >= 0) fprintf(stderr, "   starts %d..%d -> len %d mm %d gaps %d\n", from, qs0 + i - 1, last_len, last_mm, last_gp);
          from = qs0 + i; last_mm = mm; last_gp = gp; last_len = len;
        }
      }
    }
    int la = 0, lb = 0, ra = 0, rb = 0;
    std::vector<uint8_t> lops, rops;
    const int sl = semi_gapped_align([&](int a) { return q[gq - a]; }, gq, [&](int b) { return sb[gs - b]; }, gs, P.x_final, &la, &lb, &lops,
                                     opt.stale_gap_quirk);
    int sr = 0;
    if (gq < qlen && gs < slen)
      sr = semi_gapped_align([&](int a) { return q[gq + a - 1]; }, qlen - gq, [&](int b) { return sb[gs + b - 1]; }, slen - gs, P.x_final, &ra, &rb,
                             &rops, opt.stale_gap_quirk);
    Hsp hs{};
    hs.ctx = ph.ctx; hs.score = sl + sr;
    hs.q0 = gq - la; hs.s0 = gs - lb; hs.q1 = gq + ra; hs.s1 = gs + rb;
    hs.gq = gq; hs.gs = gs;
    // the edit script from the left end to the right end: the left part's ops run from its far end to the start point already,
    // the right part's from its far end back to the start point
    hs.ops = lops;
    hs.ops.insert(hs.ops.end(), rops.rbegin(), rops.rend());
    if (debug) fprintf(stderr, "traceback ctx %d start (%d,%d) -> q [%d,%d) s [%d,%d) score %d\n", hs.ctx, gq, gs, hs.q0, hs.q1, hs.s0, hs.s1, hs.score);
    if (opt.reevaluate && !reevaluate(hs, q, qlen, sb, slen, cutoff)) continue;
    if (debug) fprintf(stderr, "   re-evaluated -> q [%d,%d) s [%d,%d) score %d\n", hs.q0, hs.q1, hs.s0, hs.s1, hs.score);
    if (hs.score < cutoff) continue;
    bool dup = false;      // (the tree is asked once more after the traceback)
    for (const Hsp& t : fin)
      if (hsp_contained(hs.ctx, hs.score, hs.q0, hs.q1, hs.s0, hs.s1, t)) { dup = true; break; }
    if (!dup || !opt.recheck_contained) fin.push_back(std::move(hs));
  }
  // --- HSPs with a common start or a common end (same strand).  blastn does not simply drop the weaker one: when it reaches beyond
  // the better one it is CUT where the better one ends (begins) and what is left is re-evaluated (blast_hits.c,
  // Blast_HSPListPurgeHSPsWithCommonEndpoints with purge = FALSE, s_CutOffGapEditScript; blast_traceback.c re-evaluates the cut ones)
  std::stable_sort(fin.begin(), fin.end(), score_order);
  if (!opt.cut_common) {
    purge(fin);
  } else {
    std::vector<Hsp> cut;                       // removed from the passes once cut
    for (int pass = 0; pass < 2; ++pass) {
      std::stable_sort(fin.begin(), fin.end(), [&](const Hsp& x, const Hsp& y) {
        if (x.ctx != y.ctx) return x.ctx < y.ctx;
        if (!pass) {
          if (x.q0 != y.q0) return x.q0 < y.q0;
          if (x.s0 != y.s0) return x.s0 < y.s0;
          if (x.score != y.score) return x.score > y.score;
          if (x.q1 != y.q1) return x.q1 > y.q1;
          return x.s1 > y.s1;
        }
        if (x.q1 != y.q1) return x.q1 < y.q1;
        if (x.s1 != y.s1) return x.s1 < y.s1;
        if (x.score != y.score) return x.score > y.score;
        if (x.q0 != y.q0) return x.q0 > y.q0;
        return x.s0 > y.s0;
      });
      std::vector<Hsp> keep;
      for (Hsp& h : fin) {
        if (!keep.empty()) {
          const Hsp& lead = keep.back();
          const bool same = lead.ctx == h.ctx && (pass ? (lead.q1 == h.q1 && lead.s1 == h.s1) : (lead.q0 == h.q0 && lead.s0 == h.s0));
          if (same) {
            if (!pass && h.q1 > lead.q1) { if (cut_script(h, lead.q1, lead.s1, true)) cut.push_back(std::move(h)); }
            else if (pass && h.q0 < lead.q0) { if (cut_script(h, lead.q0, lead.s0, false)) cut.push_back(std::move(h)); }
            continue;
          }
        }
        keep.push_back(std::move(h));
      }
      fin.swap(keep);
    }
    for (Hsp& h : cut) {
      const uint8_t* q = qctx[h.ctx].data();
      if (debug) fprintf(stderr, "cut hsp ctx %d q [%d,%d) s [%d,%d) ops %zu\n", h.ctx, h.q0, h.q1, h.s0, h.s1, h.ops.size());
      const bool ok = reevaluate(h, q, qlen, sb, slen, cutoff);
      if (debug) fprintf(stderr, "   re-evaluated -> q [%d,%d) s [%d,%d) score %d %s\n", h.q0, h.q1, h.s0, h.s1, h.score, ok ? "kept" : "dropped");
      if (!ok) continue;
      fin.push_back(std::move(h));
    }
  }
  for (Hsp& h : fin) {
    const uint8_t* q = qctx[h.ctx].data();
    count_columns(h, q, sb);
    h.evalue = evalue_of(h.score, searchsp);
  }
  std::vector<Hsp> kept;
  for (Hsp& h : fin) if (h.score >= cutoff && h.evalue <= EVALUE) kept.push_back(std::move(h));
  std::stable_sort(kept.begin(), kept.end(), score_order);
  out.swap(kept);
}

void search_fragment(int frag_id, const uint8_t* frag_fwd, int qlen, const Subject& S, const Params& P, const Options& opt, std::vector<Row>& rows) {
  if (qlen < WORD) return;
  // every exact 11-mer of either strand, looked up once for all records
  std::vector<std::pair<int, int32_t>> hits[2];
  const uint32_t NB = 1u << (2 * WORD);
  for (int ctx = 0; ctx < 2; ++ctx) {
    uint32_t v = 0; int run = 0;
    for (int e = 0; e < qlen; ++e) {
      uint8_t c = ctx ? frag_fwd[qlen - 1 - e] : frag_fwd[e];
      if (c > 3) { run = 0; v = 0; continue; }
      if (ctx) c = 3 - c;
      v = ((v << 2) | c) & (NB - 1);
      if (++run < WORD) continue;
      const int qpos = e - WORD + 1;
      for (uint32_t t = S.start[v]; t < S.start[v + 1]; ++t) hits[ctx].push_back({qpos, S.pos[t]});
    }
  }
  const int n_rec = (int)S.rec_off.size() - 1;
  const int64_t db_len = S.rec_off[n_rec];
  const double searchsp = search_space(qlen, db_len, n_rec);
  std::vector<Hsp> best;
  int best_rec = -1;
  std::vector<char> touched(n_rec, 0);
  for (int ctx = 0; ctx < 2; ++ctx) for (const auto& e : hits[ctx]) touched[S.rec_of(e.second)] = 1;
  for (int rec = 0; rec < n_rec; ++rec) {
    if (!touched[rec]) continue;
    std::vector<Hsp> hs;
    search_record(frag_fwd, qlen, S, rec, hits, P, opt, searchsp, hs);
    if (hs.empty()) continue;
    if (best_rec < 0 || hs[0].score > best[0].score) { best.swap(hs); best_rec = rec; }   // -max_target_seqs 1
  }
  if (const char* dump = getenv("BLASTN_ORACLE_DUMP_STARTS")) {
    static FILE* fh = fopen(dump, "w");
    if (fh && !best.empty()) { const Hsp& h = best[0]; fprintf(fh, "%d %d %d %d %d\n", frag_id, h.ctx, h.gq, h.gs, best_rec); fflush(fh); }
  }
  for (const Hsp& h : best) {
    Row r;
    r.frag = frag_id; r.length = h.length; r.mismatch = h.mismatch; r.gaps = h.gaps; r.nident = h.nident; r.qlen = qlen;
    r.srec = best_rec; r.score = h.score;
    if (h.ctx == 0) { r.qstart = h.q0 + 1; r.qend = h.q1; r.sstart = h.s0 + 1; r.send = h.s1; }
    else { r.qstart = qlen - h.q1 + 1; r.qend = qlen - h.q0; r.sstart = h.s1; r.send = h.s0 + 1; }
    rows.push_back(r);
    if (!opt.all_hsps) break;
  }
}

}  // namespace

extern "C" {
// One ordered pair: the rows BLAST+ would print for the fragments of `query` against the database made of `subject`
// (all HSPs of the reported subject sequence, BLAST's order).  frag = 0-based fragment number (frag%05d - 1).  Returns the number
// of rows (written up to cap).  flags: bit 0 = first HSP of a fragment only.
int64_t blastn_oracle_pair(const uint8_t* qseq, const uint64_t* qrec_off, uint32_t q_nrec, const uint8_t* sseq, const uint64_t* srec_off,
                           uint32_t s_nrec, int32_t fragsize, Row* out, uint64_t cap, uint32_t flags, int32_t n_threads) {
  Subject S;
  S.rec_off.push_back(0);
  for (uint32_t r = 0; r < s_nrec; ++r) {
    for (uint64_t i = srec_off[r]; i < srec_off[r + 1]; ++i) S.code.push_back(code_of(sseq[i]));
    S.rec_off.push_back((int64_t)S.code.size());
  }
  build_index(S);
  std::vector<std::pair<std::vector<uint8_t>, int>> frags;
  for (uint32_t r = 0; r < q_nrec; ++r)
    for (uint64_t i = qrec_off[r]; i < qrec_off[r + 1]; i += (uint64_t)fragsize) {
      const uint64_t e = std::min<uint64_t>(i + (uint64_t)fragsize, qrec_off[r + 1]);
      std::vector<uint8_t> f;
      for (uint64_t p = i; p < e; ++p) f.push_back(code_of(qseq[p]));
      frags.push_back({std::move(f), (int)(e - i)});
    }
  const Params P;
  Options opt;
  opt.all_hsps = !(flags & 1u);
  opt.stale_gap_quirk = env_int("BLASTN_ORACLE_STALE_GAP", 1) != 0;
  opt.reevaluate = env_int("BLASTN_ORACLE_REEVALUATE", 0) != 0;
  opt.cut_common = env_int("BLASTN_ORACLE_CUT", 1) != 0;
  opt.recheck_contained = env_int("BLASTN_ORACLE_RECHECK", 1) != 0;
  std::vector<std::vector<Row>> per(frags.size());
  std::atomic<size_t> next{0};
  if (n_threads < 1) n_threads = 1;
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&]() {
      for (size_t f; (f = next.fetch_add(1)) < frags.size();) search_fragment((int)f, frags[f].first.data(), frags[f].second, S, P, opt, per[f]);
    });
  for (auto& t : th) t.join();
  uint64_t n = 0;
  for (const auto& v : per)
    for (const Row& r : v) { if (n < cap) out[n] = r; ++n; }
  return (int64_t)n;
}
}
