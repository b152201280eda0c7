// pg_anim_trace.h — host side of the optional traceback (pg_anim_alignments_batch with indel lists): from the pieces a unit's
// walk laid down (pg_nucmer_core.h: PnPiece, in walk order) and the run-length coded paths of its search / forced pieces
// (pgn::pn_trace_back, computed on the GPU by anim_trace_kernel; by the scalar engine in tools/anim_debug) to the indel offset
// lists of MUMmer's .delta format (pyani/nucmer.py:170-290 reads them; `show-aligns` needs them).  List management only: every
// DP cell is computed by an engine.
//
// .delta body: after each alignment header, one signed number per indel — the distance (in alignment columns, the indel's own
// column included) from the previous indel; positive = a reference base facing a gap, negative = a query base facing a gap —
// and a terminating 0.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "pg_nucmer_core.h"

namespace pgt {

// ops of one alignment, run-length coded in path order: (op, count), op as in pn_trace_back (0 D, 1 I, 2 M)
struct OpRuns {
  std::vector<std::pair<uint32_t, uint64_t>> runs;
  void push(uint32_t op, uint64_t n) {
    if (!n) return;
    if (!runs.empty() && runs.back().first == op) runs.back().second += n; else runs.emplace_back(op, n);
  }
  bool pop_match() {   // removes one trailing MATCH column; false if the path does not end on one
    if (runs.empty() || runs.back().first != 2u) return false;
    if (--runs.back().second == 0) runs.pop_back();
    return true;
  }
  bool empty() const { return runs.empty(); }
};

// Appends an engine piece (reverse RLE of pn_trace_back: n entries) to `dst`.  Its first column is the base pair the path so far
// already ends on (the last base of a match, or the start corner): kept once — if the piece starts with a gap op instead, the
// pair's MATCH column is taken back and the piece stands as it is (MUMmer joins its alignment pieces the same way).
inline bool append_piece(OpRuns& dst, const uint32_t* rle, int32_t n, bool first_of_alignment) {
  bool first = true;
  for (int32_t k = n - 1; k >= 0; --k) {
    const uint32_t op = rle[k] >> 28;
    uint64_t cnt = rle[k] & 0x0FFFFFFFu;
    if (first) {
      first = false;
      if (!first_of_alignment) {
        if (op == 2u) --cnt;
        else if (!dst.pop_match()) return false;
      }
    }
    dst.push(op, cnt);
  }
  return true;
}

// the .delta indel list of a path (without the terminating 0) and its column / base counts
inline void delta_of(const OpRuns& ops, std::vector<int64_t>& out, int64_t& a_bases, int64_t& b_bases) {
  out.clear();
  a_bases = b_bases = 0;
  int64_t run = 0;
  for (const auto& r : ops.runs) {
    if (r.first == 2u) { run += (int64_t)r.second; a_bases += (int64_t)r.second; b_bases += (int64_t)r.second; continue; }
    for (uint64_t k = 0; k < r.second; ++k) {
      ++run;
      if (r.first == 1u) { out.push_back(run); ++a_bases; } else { out.push_back(-run); ++b_bases; }
      run = 0;
    }
  }
}

// One unit: pieces in walk order (n of them), rle_of(p, n_out) -> the reverse RLE of piece p (search / forced pieces).
// Fills, for alignment a of the unit (index in al[]), its indel list.  Returns false on an inconsistent plan (a piece that does
// not start where its alignment ends, a path whose base counts differ from the alignment's extent).
// visit_key[a] = the reference start of the cluster the walk had arrived at when alignment a was created (PIECE_VISIT).
template <typename RLE>
bool unit_deltas(const pgn::PnPiece* pieces, int32_t n, const pgn::PnAln* al, int32_t n_al, RLE&& rle_of,
                 std::vector<std::vector<int64_t>>& deltas, std::vector<int32_t>& visit_key, std::string* why = nullptr) {
  std::vector<OpRuns> ops((size_t)n_al);
  visit_key.assign((size_t)n_al, -1);
  int32_t visit = -1;
  const auto fail = [&](const char* w, int32_t p) { if (why) *why = std::string(w) + " at piece " + std::to_string(p); return false; };
  for (int32_t p = 0; p < n; ++p) {
    const pgn::PnPiece& P = pieces[p];
    if (P.kind == pgn::PIECE_VISIT) { visit = P.A0; continue; }
    if (P.aln < 0 || P.aln >= n_al) continue;      // (an alignment that was merged away later never gets here: see postnuc_unit)
    OpRuns& O = ops[(size_t)P.aln];
    if (O.empty() && visit_key[(size_t)P.aln] < 0) visit_key[(size_t)P.aln] = visit;
    if (P.kind == pgn::PIECE_MATCH) {
      // a match: its first base pair may already be the path's last column (the target of the piece before it)
      uint64_t len = (uint64_t)(P.A1 - P.A0 + 1);
      if (!O.empty()) --len;
      O.push(2u, len);
      continue;
    }
    int32_t cnt = 0;
    const uint32_t* rle = rle_of(p, cnt);
    if (!rle || cnt < 0) return fail("no path", p);
    if (P.kind == pgn::PIECE_FORCED && O.empty()) {
      // the forced run in front of an alignment's first match: its own first column opens the path
      if (!append_piece(O, rle, cnt, true)) return fail("join", p);
    } else if (!append_piece(O, rle, cnt, false)) return fail("join", p);
  }
  deltas.assign((size_t)n_al, {});
  for (int32_t a = 0; a < n_al; ++a) {
    int64_t na = 0, nb = 0;
    delta_of(ops[(size_t)a], deltas[(size_t)a], na, nb);
    if (na != (int64_t)al[a].eA - al[a].sA + 1 || nb != (int64_t)al[a].eB - al[a].sB + 1) return fail("path length differs from the alignment's extent, alignment", a);
  }
  return true;
}

}  // namespace pgt
