// forced_check.cpp — HOST unit check of the forced-run band loop on rectangles whose OPTIMAL path dips far below zero (DESIGN §4,
// deviation (3): "scores below the floor of the packed words saturate to unreachable").  MUMmer's sw_align works on plain long
// ints; the engine's state words hold 15 bits of score with a bias (pg_nucmer_core.h, SCORE_BIAS).  Such rectangles cannot come out
// of a walk (a backward search breaks 200 anti-diagonals after its last best cell, so no forced rectangle spans 200 mismatches in
// a row), hence no genome pair reaches this through the C ABI: the check drives the engines directly — pgn::ScalarEngine (the
// definition the GPU engines follow cell for cell, and what anim_trace_kernel runs) and pgd::DiagWaveEngine (the host emulation
// of the GPU's diagonal-window wave engines, same header the kernels compile).
// The referee is a plain-integer statement of sw_align's forced alignment written here (whole rectangle, int64 scores, MUMmer's
// tie order MATCH > INSERT > DELETE on the state of origin, errors riding along): no packed words, no band, no floor.
//   case A  20 matching bases, 700 unrelated ones, 1200 matching: the optimal path's prefix falls to about -1 650 before it recovers
//           (rounds 3-4, floor -1024: its cells saturated and the run came back with another path's error count)
//   case B  20 matching, 1500 unrelated, 2500 matching: the prefix minimum lies below the floor (-2700) on EVERY path to the corner:
//           the run must FAIL LOUDLY (engine overflow -> PG_E_CAPACITY on the pair), not return the count of an unreachable word
//   case C  1500 bases with 30 scattered mismatches: an ordinary rectangle
// Prints one line per case and engine; exit code 0 iff all are as stated.   g++ -O2 -std=c++17 -Ipyani_amd/csrc forced_check.cpp
#include <cstdio>
#include <string>
#include <thread>
#include <vector>
#include "pg_nucmer_diag.h"
using namespace pga;

struct Packed {
  std::vector<uint32_t> codes, mask;
  int64_t len;
  explicit Packed(const std::string& s) : codes(s.size() / 16 + 2, 0), mask(s.size() / 32 + 2, 0), len((int64_t)s.size()) {
    for (size_t p = 0; p < s.size(); ++p) {
      const int c = s[p] == 'A' ? 0 : s[p] == 'C' ? 1 : s[p] == 'G' ? 2 : 3;
      codes[p >> 4] |= (uint32_t)c << (2 * (p & 15));
      mask[p >> 5] |= 1u << (p & 31);
    }
  }
  SeqView view() const { return SeqView{codes.data(), mask.data(), len}; }
};

static uint64_t rng_state = 12345;
static int rnd4() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (int)((rng_state >> 33) & 3); }

// a / b: equal where `same`, a fixed substitution elsewhere; a's bases never repeat the neighbour's, so that a shifted diagonal
// (a gap) does not line up by chance more than random
static void build(const std::vector<int>& runs, std::string& a, std::string& b) {      // runs: +n = n matching bases, -n = n mismatching ones
  const char* B = "ACGT";
  for (int r : runs)
    for (int i = 0; i < (r < 0 ? -r : r); ++i) {
      const int c = rnd4();
      a.push_back(B[c]);
      b.push_back(r < 0 ? B[(c + 1 + rnd4() % 3) & 3] : B[c]);
    }
}

// sw_align's forced (global, untrimmed) alignment of a[0 .. N) with b[0 .. M) on plain integers: score and errors of the corner
struct St { long long s; int e; };
static const long long NEG = -(1ll << 50);
static void reference(const std::string& a, const std::string& b, long long& score, int& errors, long long& prefix_min) {
  const int N = (int)a.size(), M = (int)b.size();
  // states 0 = DELETE (a B base alone: from the left), 1 = INSERT (an A base alone: from above), 2 = MATCH column
  std::vector<St> prev(3 * (M + 1)), cur(3 * (M + 1));
  std::vector<long long> pmin_prev(3 * (M + 1)), pmin_cur(3 * (M + 1));      // the lowest score along the chosen path to each state
  auto better = [](const St& x, int sx, const St& y, int sy) { return x.s != y.s ? x.s > y.s : sx > sy; };      // ties: MATCH > INSERT > DELETE
  for (int i = 0; i <= N; ++i) {
    for (int j = 0; j <= M; ++j) {
      St D{NEG, 0}, I{NEG, 0}, Mm{NEG, 0};
      long long pD = 0, pI = 0, pM = 0;
      if (i == 0 && j == 0) { Mm = St{0, 0}; }
      else {
        if (j >= 1) {      // DELETE from the left cell's states
          const St* L = &cur[3 * (j - 1)]; const long long* pl = &pmin_cur[3 * (j - 1)];
          int bs = -1; St bv{NEG, 0};
          for (int st = 0; st < 3; ++st) { if (L[st].s <= NEG / 2) continue; const St c{L[st].s + (st == 0 ? -7 : -10), L[st].e + 1}; if (bs < 0 || better(c, st, bv, bs)) { bv = c; bs = st; } }
          if (bs >= 0) { D = bv; pD = pl[bs] < bv.s ? pl[bs] : bv.s; }
        }
        if (i >= 1) {      // INSERT from the cell above
          const St* U = &prev[3 * j]; const long long* pu = &pmin_prev[3 * j];
          int bs = -1; St bv{NEG, 0};
          for (int st = 0; st < 3; ++st) { if (U[st].s <= NEG / 2) continue; const St c{U[st].s + (st == 1 ? -7 : -10), U[st].e + 1}; if (bs < 0 || better(c, st, bv, bs)) { bv = c; bs = st; } }
          if (bs >= 0) { I = bv; pI = pu[bs] < bv.s ? pu[bs] : bv.s; }
        }
        if (i >= 1 && j >= 1) {      // MATCH column from the best state of the diagonal cell
          const St* G = &prev[3 * (j - 1)]; const long long* pg = &pmin_prev[3 * (j - 1)];
          int bs = -1; St bv{NEG, 0};
          for (int st = 0; st < 3; ++st) { if (G[st].s <= NEG / 2) continue; if (bs < 0 || better(G[st], st, bv, bs)) { bv = G[st]; bs = st; } }
          if (bs >= 0) { const bool same = a[i - 1] == b[j - 1]; Mm = St{bv.s + (same ? 3 : -7), bv.e + (same ? 0 : 1)}; pM = pg[bs] < Mm.s ? pg[bs] : Mm.s; }
        }
      }
      cur[3 * j] = D; cur[3 * j + 1] = I; cur[3 * j + 2] = Mm;
      pmin_cur[3 * j] = pD; pmin_cur[3 * j + 1] = pI; pmin_cur[3 * j + 2] = pM;
    }
    prev.swap(cur); pmin_prev.swap(pmin_cur);
  }
  const St* C = &prev[3 * M];
  int bs = -1; St bv{NEG, 0};
  for (int st = 0; st < 3; ++st) { if (C[st].s <= NEG / 2) continue; if (bs < 0 || better(C[st], st, bv, bs)) { bv = C[st]; bs = st; } }
  score = bv.s; errors = bv.e; prefix_min = pmin_prev[3 * M + bs];
}

int main() {
  struct Case { const char* name; std::vector<int> runs; int want_errors; bool want_fail; };
  std::vector<int> scattered;
  for (int i = 0; i < 30; ++i) { scattered.push_back(49); scattered.push_back(-1); }
  Case cases[] = {{"A deep dip", {20, -700, 1200}, 0, false}, {"B below floor", {20, -1500, 2500}, 0, true}, {"C ordinary", scattered, 0, false}};
  int bad = 0;
  // the cases are independent and the emulated window engines are slow (64 lanes in software): one thread per case
  std::string report[3];
  int bad_of[3] = {0, 0, 0};
  auto run_case = [&](int ci) {
    Case& c = cases[ci];
    char line[512];
    std::string a, b;
    build(c.runs, a, b);
    long long ref_score, ref_min; int ref_err;
    // (the engine's rectangle includes the base pair the path already stands on: cell (1, 1) = a[0] / b[0])
    reference(a, b, ref_score, ref_err, ref_min);
    c.want_errors = ref_err;
    snprintf(line, sizeof line, "%-14s plain-integer statement: score %lld, errors %d, lowest prefix of the optimal path %lld\n", c.name, ref_score, ref_err, ref_min);
    report[ci] += line;
    if (c.want_fail ? ref_min >= -(long long)pgn::SCORE_BIAS : (ref_min < -(long long)pgn::SCORE_BIAS || (c.name[0] == 'A' && ref_min > -1100))) { report[ci] += "  the case is not what it claims to be\n"; ++bad_of[ci]; }
    const Packed PA(a), PB(b);
    const SeqView R = PA.view();
    const StrandView Q{PB.view(), 0};
    const int cap = 1 << 14;
    std::vector<pgn::Cell> d0(cap), d1(cap), d2(cap);
    for (int which = 0; which < 2; ++which) {
      int32_t A1 = (int32_t)a.size() - 1, B1 = (int32_t)b.size() - 1, err = -1;
      bool reached, overflow;
      if (which == 0) {
        pgn::ScalarEngine<SeqView, StrandView> eng{R, Q, d0.data(), d1.data(), d2.data(), cap};
        reached = eng.align(0, A1, 0, B1, pgn::FORCED_FORWARD_ALIGN, err);
        overflow = eng.overflow != 0;
      } else {
        pgd::DiagWaveEngine<SeqView, StrandView> eng(R, Q, d0.data(), d1.data(), d2.data(), cap);
        reached = eng.align(0, A1, 0, B1, pgn::FORCED_FORWARD_ALIGN, err);
        overflow = eng.slow.overflow != 0;
      }
      const bool ok = c.want_fail ? (overflow && !reached) : (!overflow && reached && err == c.want_errors && A1 == (int32_t)a.size() - 1 && B1 == (int32_t)b.size() - 1);
      snprintf(line, sizeof line, "%-14s %-16s reached %d overflow %d errors %d (want %s%d) %s\n", c.name, which ? "DiagWaveEngine" : "ScalarEngine", (int)reached, (int)overflow, err,
               c.want_fail ? "failure, not " : "", c.want_errors, ok ? "ok" : "WRONG");
      report[ci] += line;
      bad_of[ci] += !ok;
    }
  };
  {
    std::thread t0(run_case, 0), t1(run_case, 1), t2(run_case, 2);
    t0.join(); t1.join(); t2.join();
  }
  for (int ci = 0; ci < 3; ++ci) { fputs(report[ci].c_str(), stdout); bad += bad_of[ci]; }
  return bad ? 1 : 0;
}
