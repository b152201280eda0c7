// pg_nucmer_diag.h — MUMmer's alignment engine (pg_nucmer_core.h: ScalarEngine / DiagScalarEngine) laid out BY DIAGONAL for a
// 64-lane wave: lane l holds DPL consecutive diagonals (slots g = DPL l .. DPL l + DPL - 1 of a window of W = 64 DPL diagonals),
// per slot the best-state word X of the latest cell on that diagonal and its gap states D / I.  Anti-diagonal Dct has cells on the
// diagonals of its own parity, so a lane computes DPL / 2 cells per step; a cell's left / up neighbours are the adjacent slots
// (the cells of anti-diagonal Dct - 1), its diagonal neighbour is the slot's own previous word — everything but one slot per step
// is in the lane's own registers (the column layout of round 3, pga_postnuc.inc pn_align_regs, needs five cross-lane moves per
// cell and slides its window every other step: ~190 wave instructions per anti-diagonal of ~100 live cells against ~70 here).
// The base comparisons come from per-slot MATCH WINDOWS: along a diagonal the cells compare a[i + t] with b[j + t], so one XOR
// of two packed 16-base windows yields the next 16 cells' match bits; they are refilled every 32 anti-diagonals.
// The band may drift (net indels): when it comes near the window's edge the window is re-centred by whole lanes.
//
// This header holds everything that is not a cross-lane operation — the per-lane cell code, the per-step control (ranges, best
// cell, trimming, finish) and the window bookkeeping — as plain C++ for the device (pga_postnuc_diag.inc: DPP / ballots around
// it) AND for the host (DiagWaveEmu below: the same code over an array of 64 emulated lanes; tools/anim_debug --diagwave and
// tests/test_anim_cpu.py hold it against ScalarEngine on the MUMmer fixtures, so the layout is checked before it meets a GPU).
// Results are those of pgn::ScalarEngine word for word: same cells (MUMmer's dynamic band: grows by one cell per side and
// anti-diagonal, trimmed at MAX_DIFF below the best, trimmed cells stay readable), same tie order, same riding error counts.
// Two score FRAMES (round 6): forced runs keep ScalarEngine's words (score + SCORE_BIAS: their floor is part of the definition);
// trimmed searches hold score - GOOD_SCORE * floor(Dct / 2) + NORM_BIAS (diag_lane_step<NORM>), in which a match leaves the word as
// it is — the same maxima, ties and errors, cheaper instructions; DiagCtl::finish2<NORM> hands back the plain score.
#pragma once
#include <stdint.h>
#include "pg_nucmer_core.h"

namespace pgd {
using namespace pgn;

template <int DPL> struct DiagRegs { uint32_t X[DPL], D[DPL], I[DPL], mw[DPL]; };   // one lane; mw: match bits of the next 16 cells (bit 2 t)

PG_HD uint32_t rev16_fields(uint32_t x) {      // the sixteen 2-bit fields of x in reverse order
#if defined(__HIP_DEVICE_COMPILE__)
  x = __brev(x);
#else
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  x = (x >> 16) | (x << 16);
#endif
  return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}
PG_HD uint32_t spread16(uint32_t m) {      // bit t (t < 16) -> bit 2 t
  m &= 0xFFFFu;
  m = (m | (m << 8)) & 0x00FF00FFu;
  m = (m | (m << 4)) & 0x0F0F0F0Fu;
  m = (m | (m << 2)) & 0x33333333u;
  m = (m | (m << 1)) & 0x55555555u;
  return m;
}

// ---- per-lane code -------------------------------------------------------------------------------------------------------
// The cells of one lane on an anti-diagonal of parity PAR: slots s = PAR, PAR + 2, ...  nbM: the edge word of the neighbouring lane (diag_lane_edge
// below — PAR = 0: the left neighbour of slot 0 is slot DPL - 1 of the lane below; PAR = 1: the up neighbour of slot DPL - 1 is
// slot 0 of the lane above).  g0 = DPL * lane; slots in [glo, glo + gspan] are computed, the others of this parity are zeroed
// ("not computed": what their later readers must see).  key / keyw: the lane's best cell as (score field | slot) and its word —
// ties go to the larger slot = larger column, as MUMmer's ">=" scan does (TRACK = false: forced runs track nothing).  A slot
// outside the range has score field 0, i.e. key = its slot number alone: below every reachable cell's key.
// rel = {~W_STATE, ST_INSERT, ST_MATCH} handed in as values: on the device they sit in vector registers (the mask is selected per
// slot: in range or 0), so that a re-labelling is ONE v_and_or_b32 (a VOP3 instruction of this ISA takes no literal and reads the
// constant bus once; as literals the compiler needs v_and + v_or, and re-materialises the mask inside the loop).
// The match bit of a slot's next cell is the TOP bit of its window (tested as a sign), the window moves up two bits per cell.
//
// NORM (trimmed searches; round 6): the score field holds score - GOOD_SCORE * floor(Dct / 2) + NORM_BIAS instead of score +
// SCORE_BIAS (pg_nucmer_core.h: norm_offset).  Every candidate of a cell sits on the same anti-diagonal, so maxima, ties and the
// riding errors are untouched; what changes is the PRICE LIST: a diagonal step spans two anti-diagonals, so a match costs 0 — the
// word is simply kept, and an unreachable word (field 0) stays one without the `w >= W_ONE` test — and a mismatch BAD - GOOD; a gap
// step costs GOOD more on the even anti-diagonals (where floor(Dct / 2) moves), compile-time constants per parity.  The window
// then holds MISMATCH bits (diag_lane_refill<NORM>), and the diagonal step is sign-extend, and, saturating subtract — three
// full-rate instructions of gfx950 (tools/ubench/valu_issue.hip: 2.3 cycles per SIMD against 4.2 for compare / select) where
// the plain form needs add, saturating subtract, two compares and two selects.
struct DiagRelabel { uint32_t mask, st_insert, st_match; };
// What a lane's EDGE slot offers the neighbouring lane on an anti-diagonal of parity PAR: the better of its two gap candidates,
// already priced (PAR = 0: slot DPL - 1 as the LEFT neighbour of the next lane's slot 0, i.e. its delete candidates; PAR = 1: slot
// 0 as the UP neighbour of the lane below's slot DPL - 1, its insert candidates).  Priced on the sending side, ONE word crosses
// the lanes per step instead of two, and the receiving side's first use of it is a plain `and` the DPP move folds into.
template <int DPL, int PAR, bool NORM = false>
PG_HD uint32_t diag_lane_edge(const DiagRegs<DPL>& T) {
  constexpr int32_t RISE = NORM && PAR == 0 ? GOOD_SCORE : 0;
  const uint32_t c = w_gap(PAR == 0 ? T.D[DPL - 1] : T.I[0], CONT_GAP_SCORE - RISE), x = w_gap(PAR == 0 ? T.X[DPL - 1] : T.X[0], OPEN_GAP_SCORE - RISE);
  return c > x ? c : x;
}
template <int DPL, int PAR, bool TRACK, bool NORM = false>
PG_HD void diag_lane_step(DiagRegs<DPL>& T, uint32_t nbM, uint32_t g0, uint32_t glo, uint32_t gspan, const DiagRelabel& rel,
                          uint32_t& key, uint32_t& keyw) {
  key = 0u; keyw = 0u;
  const uint32_t gofs = g0 - glo;
  constexpr int32_t RISE = NORM && PAR == 0 ? GOOD_SCORE : 0;      // (NORM: the offset moves on the even anti-diagonals)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int s = PAR; s < DPL; s += 2) {
    // the slot's delete candidates come from slot s - 1 (s = 0: the lane below's edge word nbM), its insert candidates from slot
    // s + 1 (s = DPL - 1: the lane above's edge word)
    uint32_t dm, im;
    if (s == 0) dm = nbM;
    else { const uint32_t dc = w_gap(T.D[s - 1], CONT_GAP_SCORE - RISE), dx = w_gap(T.X[s - 1], OPEN_GAP_SCORE - RISE); dm = dc > dx ? dc : dx; }
    if (s == DPL - 1) im = nbM;
    else { const uint32_t ic = w_gap(T.I[s + 1], CONT_GAP_SCORE - RISE), ix = w_gap(T.X[s + 1], OPEN_GAP_SCORE - RISE); im = ic > ix ? ic : ix; }
    // a slot outside the range keeps nothing but its state label: a word with score field 0 is unreachable whatever its low bits
    // (gaps and mismatches saturate it to 0, a match step tests the field, trimming and the best-cell key look at the field), so
    // ONE select — of the re-labelling mask — takes the place of three on the results
    const bool in = (uint32_t)(gofs + (uint32_t)s) <= gspan;
    const uint32_t mk = in ? rel.mask : 0u;
    // (the edge word's insert label is ADDED — the mask has cleared those bits, so it is the same word — because `and` then `add`
    // are two VOP2 instructions the DPP move folds into, where the fused and-or is a VOP3 one that needs the move in front)
    const uint32_t d = dm & mk /* | ST_DELETE = 0 */, i = s == DPL - 1 ? (im & mk) + rel.st_insert : (im & mk) | rel.st_insert;
    const uint32_t m = ((NORM ? w_step_norm(T.X[s], T.mw[s]) : w_step_window(T.X[s], T.mw[s])) & mk) | rel.st_match;
    T.mw[s] <<= 2;
    T.X[s] = w_max3(d, i, m); T.D[s] = d; T.I[s] = i;
    if (TRACK) {
      const uint32_t k = (T.X[s] & ~(W_ONE - 1u)) | (g0 + (uint32_t)s);
      if (k >= key) { key = k; keyw = T.X[s]; }
    }
  }
}
// the lane's best cell of this parity, as diag_lane_step<TRACK = true> leaves it (for callers that look at it only on the
// anti-diagonals where some cell reaches the running best); top: the largest X word of the parity (what decides that)
template <int DPL, int PAR>
PG_HD void diag_lane_key(const DiagRegs<DPL>& T, uint32_t g0, uint32_t& key, uint32_t& keyw) {
  key = 0u; keyw = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int s = PAR; s < DPL; s += 2) {
    const uint32_t k = (T.X[s] & ~(W_ONE - 1u)) | (g0 + (uint32_t)s);
    if (k >= key) { key = k; keyw = T.X[s]; }
  }
}
template <int DPL, int PAR>
PG_HD uint32_t diag_lane_top(const DiagRegs<DPL>& T) {
  uint32_t top = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int s = PAR; s < DPL; s += 2) top = T.X[s] > top ? T.X[s] : top;
  return top;
}
// which of the lane's cells of this parity survive the trimming (bit s): X >= thr, thr = the word of score high - MAX_DIFF
template <int DPL, int PAR>
PG_HD uint32_t diag_lane_alive(const DiagRegs<DPL>& T, uint32_t thr) {
  uint32_t bits = 0u;
  for (int s = PAR; s < DPL; s += 2) bits |= (T.X[s] >= thr ? 1u : 0u) << s;
  return bits;
}
// The match windows of one lane, refilled at an anti-diagonal of parity PAR.  Slot s is next computed on anti-diagonal
// Dct + ((s ^ PAR) & 1); with u = Dct - k0 (k0 = the lane's first diagonal: even) its next cell is row i_s = (u - s + e_s) / 2,
// e_s = (s ^ PAR) & 1, column j_s = i_s + k0 + s.  ca / oka: the codes (2 bits each) / clean bits of the 32 A rows from row
// i_{DPL-1} on (field f = row i_{DPL-1} + f: the caller has undone direction and strand), cb / okb: the 32 B columns from column
// j_0 on.  The offsets of slot s inside the two windows are compile-time constants.
// CLEAN: every base of both windows is a clean one (oka & okb all ones — the caller's test, wave-wide on the device): the sixteen
// clean bits per slot need not be spread over the 2-bit fields, which is half of a refill's instructions.
template <int DPL, int PAR, bool NORM = false, bool CLEAN = false>
PG_HD void diag_lane_refill(DiagRegs<DPL>& T, uint64_t ca, uint32_t oka, uint64_t cb, uint32_t okb) {
  constexpr int e_last = ((DPL - 1) ^ PAR) & 1, e_0 = PAR & 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int s = 0; s < DPL; ++s) {
    const int e_s = (s ^ PAR) & 1;
    const int oa = (DPL - 1 - s + e_s - e_last) / 2, ob = (s + e_s - e_0) / 2;      // both numerators are even and >= 0
    const uint32_t a16 = (uint32_t)(ca >> (2 * oa)), b16 = (uint32_t)(cb >> (2 * ob));
    const uint32_t x = a16 ^ b16;
    const uint32_t eq = ~(x | (x >> 1)) & 0x55555555u;
    // cell t of the slot at bit 31 - 2 t: the fields in reverse order, one bit up
    // (NORM: the MISMATCH bits — unclean bases included — so that the step can AND the mismatch price with the window's sign)
    const uint32_t hit = CLEAN ? eq : eq & spread16((oka >> oa) & (okb >> ob));
    T.mw[s] = rev16_fields(NORM ? hit ^ 0x55555555u : hit) << 1;
  }
}
// first A row / first B column of the windows diag_lane_refill wants, for a lane whose first diagonal is k0
template <int DPL, int PAR>
PG_HD void diag_refill_origin(int32_t Dct, int32_t k0, int32_t& ia0, int32_t& jb0) {
  constexpr int e_last = ((DPL - 1) ^ PAR) & 1, e_0 = PAR & 1;
  const int32_t u = Dct - k0;
  ia0 = (u - (DPL - 1) + e_last) >> 1;      // (even numerators: exact for negative values too)
  jb0 = ((u + e_0) >> 1) + k0;
}

// ---- per-step control (wave-uniform) ---------------------------------------------------------------------------------------
// One engine call: its uniform state and the decisions of every anti-diagonal, as pgn::DiagScalarEngine::run makes them.  This is
// SCALAR code on the GPU, and a CU has one scalar unit for its four SIMDs: with the cells down to ~80 vector instructions per
// anti-diagonal, ~100 scalar ones per step were the bound (round 4, SQ_INSTS_SALU > SQ_INSTS_VALU).  Hence: everything in SLOT
// coordinates of the window (no conversion per step; a window move shifts the few values that live in them), the matrix clips
// kept as two counters, one comparison for "is the run over" (Dend = the last anti-diagonal the break rule and the matrix allow,
// updated when the best cell moves; -1 once the band has been trimmed away), FORCED a template argument (a search carries none
// of the band arithmetic).  The band grows by one diagonal per side and step from the single cell of anti-diagonal 0, so it can
// never overtake the matrix's near sides (k >= -Dct, k <= Dct hold by themselves): only the far sides clip, at Dct - 2 N and
// 2 M - Dct — which have the parity of Dct, as ka - 1 and kb + 1 do: a search's range needs no parity fix; a forced band's does.
struct DiagCtl {
  int32_t N, M, NM;
  bool fwd, forced, optimal, banded;
  int32_t Dct, Dend;
  int32_t ga, gb;                   // the latest anti-diagonal's lowest survivor - 1 / highest + 1 (slots): the next one's band before the clips —
                                    // kept WITH the +-1 so that it folds into the constants the survivors are built from (two scalar adds per step)
  int32_t lo, hi;                   // range of the current one (slots)
  int32_t c1g, c2g;                 // the matrix's far sides on the current anti-diagonal: diagonals Dct - 2 N and 2 M - Dct, as slots
  int32_t kming, kmaxg;             // a forced run's band (slots)
  int32_t shiftk;                   // diagonal of slot g: k = g - HALF + shiftk
  uint32_t high_f, high_w;          // best score so far as a score FIELD (score + SCORE_BIAS) and its word
  uint32_t high_fw, thr_w;          // high_f << SCORE_SHIFT; the trimming threshold word (both change only when the best does)
  int32_t FinishCt, FinishG, FinishShift;
  uint32_t span_sum;                // sum of (hi - lo) over the steps: cells = span_sum / 2 + steps
  int32_t wmax;
  int32_t next_refill;
  template <int DPL>
  PG_HD void init(int32_t N_, int32_t M_, unsigned m_o, int32_t band_w) {
    constexpr int32_t HALF = 32 * DPL;
    N = N_; M = M_; NM = N_ + M_;
    fwd = m_o & DIRECTION_BIT; forced = m_o & FORCED_BIT; optimal = m_o & OPTIMAL_BIT; banded = band_w >= 0;
    kming = (M - N < 0 ? M - N : 0) - band_w + HALF; kmaxg = (M - N > 0 ? M - N : 0) + band_w + HALF;
    Dct = 1; Dend = forced ? NM : (NM < BREAK_LEN ? NM : BREAK_LEN);      // (FinishCt = 0: the break rule allows BREAK_LEN steps)
    ga = HALF - 1; gb = HALF + 1; lo = HALF; hi = HALF; shiftk = 0;
    c1g = 1 - 2 * N + HALF; c2g = 2 * M - 1 + HALF;
    high_f = 0u; high_w = 0u; FinishCt = 0; FinishG = HALF; FinishShift = 0;
    high_fw = 0u; thr_w = (0u - (uint32_t)MAX_DIFF) << SCORE_SHIFT;
    span_sum = 0u; wmax = 0; next_refill = 1;
  }
  // 0: compute anti-diagonal Dct (lo / hi set); 1: the run is over (end of the matrix, break length, band trimmed away);
  // 2: the band is empty after clipping
  template <bool FORCED>
  PG_HD int begin_step() {
    if (Dct > Dend) return 1;
    lo = ga > c1g ? ga : c1g; hi = gb < c2g ? gb : c2g;
    if (FORCED && banded) {
      if (lo < kming) lo = kming;
      if (hi > kmaxg) hi = kmaxg;
      if ((lo + Dct) & 1) ++lo;
      if ((hi + Dct) & 1) --hi;
    }
    return lo > hi ? 2 : 0;
  }
  // begin_step as ONE decision (the device's step loop: a scalar compare + branch less per anti-diagonal): true = compute anti-diagonal Dct
  template <bool FORCED>
  PG_HD bool begin_step_go() {
    lo = ga > c1g ? ga : c1g; hi = gb < c2g ? gb : c2g;
    if (FORCED && banded) {
      if (lo < kming) lo = kming;
      if (hi > kmaxg) hi = kmaxg;
      if ((lo + Dct) & 1) ++lo;
      if ((hi + Dct) & 1) --hi;
    }
    return ((Dend - Dct) | (hi - lo)) >= 0;      // Dct <= Dend && lo <= hi (the misfit mark Dend = INT32_MIN is set when the loop is left)
  }
  // How many LANES the window has to move before this step (0: fine; INT32_MIN: the band does not fit the window).  The slots
  // lo - 1 and hi + 1 are read, so two slots of margin are kept on either side.
  template <int DPL>
  PG_HD int32_t window_check() const {
    constexpr int32_t W = 64 * DPL, HALF = W / 2;
    if (lo >= 2 && hi <= W - 3) return 0;
    if (hi - lo + 5 > W) return INT32_MIN;
    int32_t n = ((lo + hi) / 2 - HALF) / DPL;
    if (n == 0) n = lo < 2 ? -1 : 1;
    if (lo - DPL * n < 2 || hi - DPL * n > W - 3) return INT32_MIN;
    return n;
  }
  template <int DPL>
  PG_HD void window_move(int32_t n) { window_shift(DPL * n); }      // the lanes' registers have moved n lanes down
  PG_HD void window_shift(int32_t d) {      // slot g now holds diagonal g - HALF + shiftk + d (d even): every slot coordinate follows
    shiftk += d; lo -= d; hi -= d; ga -= d; gb -= d; c1g -= d; c2g -= d; kming -= d; kmaxg -= d;
    next_refill = Dct;
  }
  template <bool WIDEST>
  PG_HD void note_cells() {
    span_sum += (uint32_t)(hi - lo);
    if (WIDEST) { const int32_t w = (hi - lo) / 2 + 1; if (w > wmax) wmax = w; }
  }
  PG_HD int32_t steps() const { return Dct - 1; }      // anti-diagonals computed (Dct starts at 1 and moves on after each)
  PG_HD unsigned long long cells_total() const { return (unsigned long long)(span_sum / 2u) + (unsigned long long)steps(); }
  // the wave's best cell of this anti-diagonal: gk = (score field | slot), gw its word; ties move the finish forward (">=")
  PG_HD void update_best(uint32_t gk, uint32_t gw) {
    const uint32_t f = gk >> SCORE_SHIFT;
    if (f >= (high_fw >> SCORE_SHIFT)) {      // (high_fw, not high_f: in the normalised frame it is the one kept in step with Dct)
      high_f = f; high_w = gw; high_fw = f << SCORE_SHIFT; thr_w = (f - (uint32_t)MAX_DIFF) << SCORE_SHIFT; FinishCt = Dct; FinishG = (int32_t)(gk & (W_ONE - 1u)); FinishShift = shiftk;
      Dend = Dct + BREAK_LEN < NM ? Dct + BREAK_LEN : NM;
    }
  }
  // the word a cell must reach to survive the trimming (cells more than MAX_DIFF below the best score go).  high >= -10 after the
  // first anti-diagonal of a search (its cells are one step from the origin) and never falls: the threshold is a positive word, so
  // a zero word — a slot outside the computed range — never counts as a survivor.
  PG_HD uint32_t trim_threshold() const { return thr_w; }
  // survivors: the lowest / highest surviving slot (any = false: none)
  // NORM_EVEN: a trimmed search in the normalised frame (diag_lane_step<NORM>) moves on to an EVEN anti-diagonal, where the
  // frame's offset rises by GOOD_SCORE: the two words its cells are compared with follow (the best score itself stays)
  template <bool FORCED, bool NORM_EVEN = false>
  PG_HD void end_step(bool any, uint32_t gmin, uint32_t gmax) {
    if (FORCED) { ga = lo - 1; gb = hi + 1; }
    else if (any) { ga = (int32_t)gmin - 1; gb = (int32_t)gmax + 1; }
    else { ga = 0; gb = 1; Dend = -1; }
    ++Dct; ++c1g; --c2g;
    if (NORM_EVEN) { high_fw -= (uint32_t)GOOD_SCORE << SCORE_SHIFT; thr_w -= (uint32_t)GOOD_SCORE << SCORE_SHIFT; }
  }
  // after the loop: where the call finished.  corner_slot: the slot of the target corner (its X word is needed when the corner
  // counts as reached); returns whether the caller has to deliver that word (finish2) or the best cell's word stands.
  template <int DPL>
  PG_HD bool finish1(bool& reached, uint32_t& corner_slot, int32_t& FinishK) {
    constexpr int32_t HALF = 32 * DPL;
    reached = false;
    const int32_t last = Dct - 1;
    corner_slot = 0u;
    FinishK = FinishG - HALF + FinishShift;
    if (last == NM) {
      if (!optimal) { reached = true; FinishCt = NM; FinishK = M - N; corner_slot = (uint32_t)(M - N - shiftk + HALF); return true; }
      if (FinishCt == last) reached = true;
    }
    return false;
  }
  template <bool NORM = false>      // (NORM: fin_w is a word of anti-diagonal FinishCt in the normalised frame)
  PG_HD void finish2(uint32_t fin_w, int32_t FinishK, int32_t Astart, int32_t Bstart, int32_t& Aend, int32_t& Bend, int32_t& errors, int32_t& score) const {
    const int32_t fi = (FinishCt - FinishK) / 2, fj = (FinishCt + FinishK) / 2;
    Aend = fwd ? Astart + fi - 1 : Astart - fi + 1;
    Bend = fwd ? Bstart + fj - 1 : Bstart - fj + 1;
    errors = (int32_t)w_errors(fin_w);
    score = NORM ? (int32_t)(fin_w >> SCORE_SHIFT) - (int32_t)NORM_BIAS + norm_offset(FinishCt) : w_score(fin_w);
  }
};

// ---- host emulation of the wave (the statement of pga_postnuc_diag.inc) ---------------------------------------------------------
#if !defined(__HIP_DEVICE_COMPILE__)
template <int DPL, typename RefT, typename QryT>
struct DiagWaveEmu {
  static constexpr int W = 64 * DPL, HALF = W / 2;
  const RefT& R;
  const QryT& Q;
  long cells = 0, calls = 0, moves = 0, fails = 0;
  uint32_t last_cells = 0, last_wmax = 0;
  DiagRegs<DPL> T[64];
  template <int PAR, bool NORM>
  void refill(const DiagCtl& C, int32_t Astart, int32_t Bstart) {
    for (int l = 0; l < 64; ++l) {
      const int32_t k0 = DPL * l - HALF + C.shiftk;
      int32_t ia0, jb0;
      diag_refill_origin<DPL, PAR>(C.Dct, k0, ia0, jb0);
      uint64_t ca = 0, cb = 0; uint32_t oka = 0, okb = 0;
      for (int f = 0; f < 32; ++f) {
        const int64_t pa = C.fwd ? (int64_t)Astart + (ia0 + f) - 1 : (int64_t)Astart - (ia0 + f) + 1;
        const int64_t pb = C.fwd ? (int64_t)Bstart + (jb0 + f) - 1 : (int64_t)Bstart - (jb0 + f) + 1;
        if (R.clean(pa)) { ca |= (uint64_t)R.base(pa) << (2 * f); oka |= 1u << f; }
        if (Q.clean(pb)) { cb |= (uint64_t)Q.base(pb) << (2 * f); okb |= 1u << f; }
      }
      if ((oka & okb) == 0xFFFFFFFFu) diag_lane_refill<DPL, PAR, NORM, true>(T[l], ca, oka, cb, okb);      // (the device decides wave-wide; same words either way)
      else diag_lane_refill<DPL, PAR, NORM, false>(T[l], ca, oka, cb, okb);
    }
  }
  template <int PAR, bool FORCED>
  void step(DiagCtl& C) {
    uint32_t nbM[64], key[64], keyw[64];
    for (int l = 0; l < 64; ++l) {
      if (PAR == 0) nbM[l] = l > 0 ? diag_lane_edge<DPL, PAR, !FORCED>(T[l - 1]) : 0u;
      else nbM[l] = l < 63 ? diag_lane_edge<DPL, PAR, !FORCED>(T[l + 1]) : 0u;
    }
    for (int l = 0; l < 64; ++l)
      diag_lane_step<DPL, PAR, !FORCED, !FORCED>(T[l], nbM[l], (uint32_t)(DPL * l), (uint32_t)C.lo, (uint32_t)(C.hi - C.lo), DiagRelabel{~W_STATE, ST_INSERT, ST_MATCH}, key[l], keyw[l]);
    bool any = false; uint32_t gmin = 0, gmax = 0;
    if (!FORCED) {
      uint32_t gk = 0, gw = 0;
      for (int l = 0; l < 64; ++l) if (key[l] > gk) { gk = key[l]; gw = keyw[l]; }      // (keys are unique: they carry the slot)
      C.update_best(gk, gw);
      const uint32_t thr = C.trim_threshold();
      for (int l = 0; l < 64; ++l) {
        const uint32_t bits = diag_lane_alive<DPL, PAR>(T[l], thr);
        for (int s = PAR; s < DPL; s += 2)
          if ((bits >> s) & 1u) { const uint32_t g = (uint32_t)(DPL * l + s); if (!any) { gmin = g; any = true; } gmax = g; }
      }
    }
    C.template end_step<FORCED, !FORCED && PAR == 1>(any, gmin, gmax);
  }
  template <bool FORCED>
  bool run_(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t band_w, int32_t& errors, int32_t& score, bool& reached) {
    DiagCtl C;
    const bool fwd = m_o & DIRECTION_BIT;
    C.template init<DPL>(fwd ? Aend - Astart + 1 : Astart - Aend + 1, fwd ? Bend - Bstart + 1 : Bstart - Bend + 1, m_o, band_w);
    for (int l = 0; l < 64; ++l) for (int s = 0; s < DPL; ++s) { T[l].X[s] = 0u; T[l].D[s] = 0u; T[l].I[s] = 0u; T[l].mw[s] = 0u; }
    T[HALF / DPL].X[0] = FORCED ? w_make(0, 0, ST_MATCH) : (NORM_BIAS << SCORE_SHIFT) | ST_MATCH;      // (trimmed searches: the normalised frame)
    ++calls;
    for (;;) {
      if (C.template begin_step<FORCED>()) break;
      const int32_t n = C.template window_check<DPL>();
      if (n == INT32_MIN) { ++fails; return false; }
      if (n != 0) {
        DiagRegs<DPL> Z;
        for (int s = 0; s < DPL; ++s) { Z.X[s] = 0u; Z.D[s] = 0u; Z.I[s] = 0u; Z.mw[s] = 0u; }
        DiagRegs<DPL> U[64];
        for (int l = 0; l < 64; ++l) U[l] = (l + n >= 0 && l + n < 64) ? T[l + n] : Z;
        for (int l = 0; l < 64; ++l) T[l] = U[l];
        C.template window_move<DPL>(n);
        ++moves;
      }
      if (C.Dct == C.next_refill) { if (C.Dct & 1) refill<1, !FORCED>(C, Astart, Bstart); else refill<0, !FORCED>(C, Astart, Bstart); C.next_refill = C.Dct + 32; }
      C.template note_cells<true>();
      if (C.Dct & 1) step<1, FORCED>(C); else step<0, FORCED>(C);
    }
    uint32_t corner = 0, fin_w = C.high_w;
    int32_t FinishK = 0;
    if (C.template finish1<DPL>(reached, corner, FinishK)) fin_w = T[corner / DPL].X[corner % DPL];
    C.template finish2<!FORCED>(fin_w, FinishK, Astart, Bstart, Aend, Bend, errors, score);
    const unsigned long long nc = C.cells_total();
    cells += (long)nc;
    last_cells = nc > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nc; last_wmax = (uint32_t)C.wmax;
    return true;
  }
  // as pgn::ScalarEngine::run (band_w < 0: MUMmer's own band).  false: the band did not fit the window (nothing is returned).
  bool run(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t band_w, int32_t& errors, int32_t& score, bool& reached) {
    return (m_o & FORCED_BIT) ? run_<true>(Astart, Aend, Bstart, Bend, m_o, band_w, errors, score, reached)
                              : run_<false>(Astart, Aend, Bstart, Bend, m_o, band_w, errors, score, reached);
  }
};
#endif

// A fixed band of `span` diagonals (its 5 margin slots included) is sure to fit a window of 64 DPL diagonals: the window moves by
// whole lanes towards the band's centre (DiagCtl::window_check: the quotient is truncated), so up to DPL - 1 diagonals of either
// margin are lost to the placement.
PG_HD bool diag_window_holds(int64_t span, int dpl) { return span <= 64 * (int64_t)dpl - 2 * (int64_t)dpl + 2; }
#if !defined(__HIP_DEVICE_COMPILE__)
// The engine postnuc_unit is given when the host statement runs on the emulated wave engines: trimmed searches / alignments on
// the 256-diagonal window, forced runs on the smallest window that holds their certified band (256, 384, 512 ... 2048 diagonals), anything that
// does not fit on pgn::ScalarEngine — the dispatch of the GPU's PnWaveEngine (pga_postnuc.inc).
template <typename RefT, typename QryT>
struct DiagWaveEngine {
  ScalarEngine<RefT, QryT> slow;
  DiagWaveEmu<2, RefT, QryT> e2;      // (forced runs only)
  DiagWaveEmu<4, RefT, QryT> e4;
  DiagWaveEmu<6, RefT, QryT> e6;
  DiagWaveEmu<8, RefT, QryT> e8;
  DiagWaveEmu<12, RefT, QryT> e12;
  DiagWaveEmu<16, RefT, QryT> e16;
  DiagWaveEmu<24, RefT, QryT> e24;
  DiagWaveEmu<32, RefT, QryT> e32;
  long fallbacks = 0;
  void (*fallback_log)(int32_t N, int32_t M, unsigned m_o, int32_t band_w) = nullptr;      // development: what did not fit
  DiagWaveEngine(const RefT& R, const QryT& Q, Cell* d0, Cell* d1, Cell* d2, int32_t cap)
      : slow{R, Q, d0, d1, d2, cap}, e2{R, Q}, e4{R, Q}, e6{R, Q}, e8{R, Q}, e12{R, Q}, e16{R, Q}, e24{R, Q}, e32{R, Q} {}
  bool gap_ready(int32_t, PnGap&) const { return false; }
  void piece(uint32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, unsigned) {}
  bool bwd_ready(int, PnBwd&) const { return false; }
  void bwd_key(int, int32_t, int32_t, int32_t, int32_t, unsigned) {}
  bool fwd_ready(int, PnFwd&) const { return false; }
  int32_t forced_errors(int32_t A0, int32_t A1, int32_t B0, int32_t B1, PnAln*) {
    int32_t err = 0, a = A1, b = B1;
    align(A0, a, B0, b, FORCED_FORWARD_ALIGN, err);
    return err;
  }
  bool shadowed(const Chain* chains, const PnAln* al, int from, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return PnScalarScans().shadowed(chains, al, from, c, sA, eA, sB, eB); }
  int shadow_first(const Chain* chains, const PnAln* al, int n_al, int c, int32_t sA, int32_t eA, int32_t sB, int32_t eB) const {
    return PnScalarScans().shadow_first(chains, al, n_al, c, sA, eA, sB, eB); }
  int reverse_target(const Chain* chains, const PnAln* al, int cura, int c, int32_t sA, int32_t sB, int32_t dist) const {
    return PnScalarScans().reverse_target(chains, al, cura, c, sA, sB, dist); }
  int forward_target(const Chain* chains, const Match* cm, const int32_t* order, int n, int curk, int c, int32_t sA, int32_t sB,
                     int32_t dist, int32_t& targetA, int32_t& targetB) const {
    return PnScalarScans().forward_target(chains, cm, order, n, curk, c, sA, sB, dist, targetA, targetB); }
  // diagonals a run needs in its window (band_w < 0: MUMmer's own band -> the narrow window)
  static int window_for(int32_t N, int32_t M, unsigned m_o, int32_t band_w) {
    if (!(m_o & FORCED_BIT)) return 256;
    const int32_t df = N < M ? M - N : N - M;
    const int64_t span = band_w >= 0 ? (int64_t)df + 2 * (int64_t)band_w + 1 : (int64_t)N + M + 1;
    for (int w : {128, 256, 384, 512, 768, 1024, 1536, 2048}) if (diag_window_holds(span + 5, w / 64)) return w;
    return 0;
  }
  bool run(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t band_w, int32_t& errors, int32_t& score) {
    const bool fwd = m_o & DIRECTION_BIT;
    const int32_t N = fwd ? Aend - Astart + 1 : Astart - Aend + 1, M = fwd ? Bend - Bstart + 1 : Bstart - Bend + 1;
    int32_t a = Aend, b = Bend;
    bool reached = false, done = false;
    switch (window_for(N, M, m_o, band_w)) {
      case 128: done = e2.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      case 256: done = e4.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      case 384: done = e6.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      case 512: done = e8.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      case 768: done = e12.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      case 1024: done = e16.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      case 1536: done = e24.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      case 2048: done = e32.run(Astart, a, Bstart, b, m_o, band_w, errors, score, reached); break;
      default: break;
    }
    if (done) { Aend = a; Bend = b; return reached; }
    ++fallbacks;
    if (fallback_log) fallback_log(N, M, m_o, band_w);
    return slow.run(Astart, Aend, Bstart, Bend, m_o, band_w, errors, &score);
  }
  bool align(int32_t Astart, int32_t& Aend, int32_t Bstart, int32_t& Bend, unsigned m_o, int32_t& errors) {
    int32_t score = 0;
    if (!(m_o & FORCED_BIT)) return run(Astart, Aend, Bstart, Bend, m_o, -1, errors, score);
    const bool fwd = m_o & DIRECTION_BIT;
    const int32_t N = fwd ? Aend - Astart + 1 : Astart - Aend + 1, M = fwd ? Bend - Bstart + 1 : Bstart - Bend + 1;
    for (int32_t w = FORCED_BAND_FIRST;;) {
      int32_t a = Aend, b = Bend;
      const bool whole = w >= (N > M ? N : M);
      const bool reached = run(Astart, a, Bstart, b, m_o, whole ? -1 : w, errors, score);
      const int v = slow.overflow ? 0 : forced_verdict(reached, score, whole, N, M, w);
      if (v == 2) { slow.overflow = 1; Aend = a; Bend = b; errors = 0; return false; }
      if (v == 0) { Aend = a; Bend = b; return reached; }
      w = forced_band_after(w, N, M, score);
    }
  }
};
#endif

}  // namespace pgd
