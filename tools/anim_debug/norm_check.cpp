// norm_check.cpp — HOST unit check of the NORMALISED score frame of the wave engines' trimmed searches (pg_nucmer_diag.h
// diag_lane_step<NORM>, pg_nucmer_core.h norm_offset / w_step_norm).  In that frame the score field of a word
// on anti-diagonal d holds score - 3 floor(d / 2) + 32 767, a match keeps the word and the offset slides the best / threshold words
// every other step; the plain frame (pgn::ScalarEngine: score + 2 700) is the definition.  The MUMmer fixtures and C4 pairs already
// run both (tests/test_nucmer_oracle.py, ANIM_DIAGWAVE); this check drives the two engines DIRECTLY on searches built to sit at the
// corners of the frame's range, where a fixture rarely goes:
//   long-low     10 000 x 10 000 bases in blocks of 30 matching + 16 substituted ones: the search just survives, a few points per
//                block — the score creeps while the offset runs to 30 000: the live words sit in the BOTTOM quarter of the field
//                for thousands of anti-diagonals
//   long-exact   10 000 matching bases: score 30 000, the word stays at the TOP of the field (bias 32 767) all the way
//   indels       a gap every ~40 bases, both kinds: gap steps on odd and even anti-diagonals (where the offset moves)
//   unrelated    random against random: the search breaks 200 anti-diagonals after its only best cell
//   tail-drop    2 000 good bases, then unrelated ones: the band is trimmed away far from the start
//   tiny         N, M in 1 .. 6 and a 1 x 400 strip: the matrix's far sides clip from the first steps on
// each forwards (FORWARD_ALIGN with and without OPTIMAL_BIT, i.e. target reached or best cell) and backwards (BACKWARD_SEARCH).
// Equal means: end coordinates, errors, score and `reached`, and the window must have held the band.  Exit code 0 iff all equal.
//   g++ -O2 -std=c++17 -Ipyani_amd/csrc tools/anim_debug/norm_check.cpp -o tools/anim_debug/norm_check
#include <cstdio>
#include <string>
#include <vector>
#include "pg_nucmer_diag.h"
using namespace pga;

struct Packed {
  std::vector<uint32_t> codes, mask;
  int64_t len;
  explicit Packed(const std::string& s) : codes(s.size() / 16 + 2, 0), mask(s.size() / 32 + 2, 0), len((int64_t)s.size()) {
    for (size_t p = 0; p < s.size(); ++p) {
      if (s[p] == 'N') continue;      // (an unclean base: code 0, clean bit off)
      const int c = s[p] == 'A' ? 0 : s[p] == 'C' ? 1 : s[p] == 'G' ? 2 : 3;
      codes[p >> 4] |= (uint32_t)c << (2 * (p & 15));
      mask[p >> 5] |= 1u << (p & 31);
    }
  }
  SeqView view() const { return SeqView{codes.data(), mask.data(), len}; }
};

static uint64_t rng_state = 20250301;
static unsigned rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng_state >> 33); }
static const char* BASES = "ACGT";

// b = a copy of a with substitutions at rate sub (per mille), single-base insertions / deletions at rate indel (per mille), and an
// N every n_every bases (0: none)
static void mutate(const std::string& a, int sub, int indel, int n_every, std::string& b) {
  b.clear();
  for (size_t i = 0; i < a.size(); ++i) {
    const unsigned r = rnd() % 1000;
    if ((int)r < indel) { if (rnd() & 1) continue; b.push_back(BASES[rnd() & 3]); }
    char c = a[i];
    if ((int)(rnd() % 1000) < sub) c = BASES[((c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3) + 1 + rnd() % 3) & 3];
    if (n_every && i % n_every == (size_t)n_every - 1) c = 'N';
    b.push_back(c);
  }
}
static std::string random_seq(size_t n) { std::string s; for (size_t i = 0; i < n; ++i) s.push_back(BASES[rnd() & 3]); return s; }

int main() {
  struct Case { const char* name; std::string a, b; };
  std::vector<Case> cases;
  {   // 30 matching bases, 16 substituted ones, 217 times: the search just keeps finding a new best cell within 200 anti-diagonals
    Case c{"long-low", random_seq(10000), ""};
    c.b = c.a;
    for (size_t i = 0; i < c.b.size(); ++i) if (i % 46 >= 30) c.b[i] = BASES[((c.b[i] == 'A' ? 0 : c.b[i] == 'C' ? 1 : c.b[i] == 'G' ? 2 : 3) + 1 + rnd() % 3) & 3];
    cases.push_back(c);
  }
  { Case c{"long-exact", random_seq(10000), ""}; c.b = c.a; cases.push_back(c); }
  { Case c{"indels", random_seq(6000), ""}; mutate(c.a, 20, 25, 0, c.b); cases.push_back(c); }
  { Case c{"unrelated", random_seq(3000), random_seq(3000)}; cases.push_back(c); }
  { Case c{"tail-drop", random_seq(4000), ""}; mutate(c.a.substr(0, 2000), 30, 3, 0, c.b); c.b += random_seq(2000); cases.push_back(c); }
  { Case c{"with-N", random_seq(5000), ""}; mutate(c.a, 40, 2, 97, c.b); cases.push_back(c); }
  { Case c{"strip-1x400", random_seq(1), random_seq(400)}; cases.push_back(c); }
  for (int n = 1; n <= 6; ++n) for (int m = 1; m <= 6; m += 2) { Case c{"tiny", random_seq((size_t)n), ""}; c.b = (c.a + random_seq(6)).substr(0, (size_t)m); cases.push_back(c); }
  int bad = 0, runs = 0;
  long emu_calls = 0;
  const int cap = 1 << 15;
  std::vector<pgn::Cell> d0(cap), d1(cap), d2(cap);
  for (const Case& c : cases) {
    const Packed PA(c.a), PB(c.b);
    const SeqView R = PA.view();
    const StrandView Q{PB.view(), 0};
    const int32_t N = (int32_t)c.a.size(), M = (int32_t)c.b.size();
    struct Mode { const char* name; unsigned m_o; bool fwd; };
    const Mode modes[] = {{"forward to the target", pgn::FORWARD_ALIGN, true}, {"forward, best cell", pgn::FORWARD_ALIGN | pgn::OPTIMAL_BIT, true},
                          {"backward search", pgn::BACKWARD_SEARCH, false}, {"backward search, best cell", pgn::BACKWARD_SEARCH | pgn::OPTIMAL_BIT, false}};
    for (const Mode& md : modes) {
      // forwards: from (0, 0) towards (N - 1, M - 1); backwards: from (N - 1, M - 1) towards (0, 0)
      const int32_t As = md.fwd ? 0 : N - 1, Bs = md.fwd ? 0 : M - 1;
      int32_t a1 = md.fwd ? N - 1 : 0, b1 = md.fwd ? M - 1 : 0, e1 = -1, s1 = 0;
      int32_t a2 = a1, b2 = b1, e2 = -1, s2 = 0;
      pgn::ScalarEngine<SeqView, StrandView> ref{R, Q, d0.data(), d1.data(), d2.data(), cap};
      const bool r1 = ref.run(As, a1, Bs, b1, md.m_o, -1, e1, &s1);
      pgd::DiagWaveEmu<4, SeqView, StrandView> emu{R, Q};
      bool r2 = false;
      const bool fit = emu.run(As, a2, Bs, b2, md.m_o, -1, e2, s2, r2);
      emu_calls += emu.calls;
      ++runs;
      const bool ok = !ref.overflow && fit && r1 == r2 && a1 == a2 && b1 == b2 && e1 == e2 && s1 == s2;
      if (!ok || std::string(c.name) != "tiny")
        printf("%-12s %4d x %-5d %-28s plain: end (%d, %d) errors %d score %d reached %d | normalised: end (%d, %d) errors %d score %d reached %d fit %d  %s\n", c.name, N, M, md.name,
               a1, b1, e1, s1, (int)r1, a2, b2, e2, s2, (int)r2, (int)fit, ok ? "ok" : "WRONG");
      bad += !ok;
    }
  }
  printf("%d searches on both engines (%ld on the emulated wave engine), %d differ\n", runs, emu_calls, bad);
  return bad ? 1 : 0;
}
