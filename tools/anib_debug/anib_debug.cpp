// anib_debug.cpp — HOST prototype / checker of fragment mode (pyani_amd/csrc/pg_anib_core.h): fragments of the query FASTA
// against the subject FASTA, one BLAST-table-shaped row per fragment hit.  Development harness in the GPU-less container.
//   g++ -O2 -std=c++17 -I../../pyani_amd/csrc anib_debug.cpp -o anib_debug ;  anib_debug query.fna subject.fna [K] [top]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>
#include "pg_anib_core.h"
using namespace pga;

struct Genome {
  std::vector<uint32_t> codes, mask;
  std::vector<int32_t> rec_start;
  std::vector<std::string> ids;
  int64_t len = 0;
  SeqView view() const { return SeqView{codes.data(), mask.data(), len}; }
};
static Genome load(const char* path) {
  Genome g;
  std::ifstream in(path);
  std::string line;
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

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: anib_debug query.fna subject.fna [K=16] [top=2]\n"); return 2; }
  const int K = argc > 3 ? atoi(argv[3]) : 16, TOP = argc > 4 ? atoi(argv[4]) : 2, STEP = argc > 5 ? atoi(argv[5]) : 1;
  Genome Q = load(argv[1]), S = load(argv[2]);
  const SeqView SV = S.view(), QV = Q.view();
  // subject K-mer table (forward strand)
  std::vector<std::pair<uint64_t, int32_t>> tab;
  {
    uint64_t v = 0; int run = 0; const uint64_t keep = (1ull << (2 * K)) - 1;
    for (int64_t p = 0; p < SV.len; ++p) {
      if (!SV.clean(p)) { run = 0; v = 0; continue; }
      v = ((v << 2) | (uint64_t)SV.base(p)) & keep;
      if (++run >= K) tab.push_back({v, (int32_t)(p - K + 1)});
    }
    std::sort(tab.begin(), tab.end());
  }
  int frag_no = 0;
  for (size_t rec = 0; rec + 1 < Q.rec_start.size(); ++rec) {
    const int32_t r0 = Q.rec_start[rec], r1 = Q.rec_start[rec + 1] - 1;   // [r0, r1) stream positions of the record
    for (int32_t f0 = r0; f0 < r1; f0 += FRAG_SIZE) {
      ++frag_no;
      const int32_t qlen = std::min<int32_t>(FRAG_SIZE, r1 - f0);
      std::vector<FragHit> hits; std::vector<int> hit_strand;
      for (int strand = 0; strand < 2; ++strand) {
        // fragment on this strand: base(i) for i in [0, qlen)
        auto qbase = [&](int32_t i) -> int {
          const int64_t p = strand ? f0 + (qlen - 1 - i) : f0 + i;
          if (!QV.clean(p)) return 4;
          return strand ? 3 - QV.base(p) : QV.base(p);
        };
        // seeds: K-mer hits (fragment position q, subject position r); votes per diagonal D = r - q
        std::vector<std::pair<int32_t, int32_t>> seeds;   // (D, q)
        uint64_t v = 0; int run = 0; const uint64_t keep = (1ull << (2 * K)) - 1;
        for (int32_t e = 0; e < qlen; ++e) {
          const int b = qbase(e);
          if (b >= 4) { run = 0; v = 0; continue; }
          v = ((v << 2) | (uint64_t)b) & keep;
          if (++run < K) continue;
          const int32_t q = e - K + 1;
          // the engine samples the query genome's STRAND position (every STEP-th), not the fragment's
          const int64_t gp = strand ? (QV.len - 1 - (f0 + (qlen - 1 - q))) : f0 + q;
          if (gp % STEP) continue;
          auto it = std::lower_bound(tab.begin(), tab.end(), std::make_pair(v, (int32_t)-1));
          for (int cnt = 0; it != tab.end() && it->first == v && cnt < 64; ++it, ++cnt) seeds.push_back({it->second - q, q});
        }
        if (seeds.empty()) continue;
        std::sort(seeds.begin(), seeds.end());
        // best windows of 32 diagonals (votes = seed hits inside), up to TOP of them at least 48 diagonals apart
        struct Win { int votes; size_t lo, hi; };
        std::vector<Win> wins;
        size_t lo = 0;
        for (size_t hi = 0; hi < seeds.size(); ++hi) {
          while (seeds[hi].first - seeds[lo].first > 32) ++lo;
          wins.push_back(Win{(int)(hi - lo + 1), lo, hi});
        }
        std::stable_sort(wins.begin(), wins.end(), [](const Win& a, const Win& b) { return a.votes > b.votes; });
        std::vector<int32_t> chosen;
        for (auto& w : wins) {
          const int32_t centre = seeds[w.lo].first + (seeds[w.hi].first - seeds[w.lo].first) / 2;
          bool far = true;
          for (int32_t c : chosen) if (std::abs(c - centre) < 48) far = false;
          if (!far) continue;
          chosen.push_back(centre);
          // anchor: the longest exact match through a seed of the window (ties: smallest q)
          auto q_at = [&](int64_t p) -> int { return (p >= 0 && p < qlen) ? qbase((int32_t)p) : 4; };
          const int srec0 = record_of(S.rec_start.data(), (int)S.rec_start.size() - 1, std::max<int64_t>(0, std::min<int64_t>(SV.len - 1, seeds[w.lo].first + seeds[w.lo].second)));
          const int64_t s_lo = S.rec_start[srec0], s_hi = S.rec_start[srec0 + 1] - 1;
          auto s_at = [&](int64_t p) -> int { return (p >= s_lo && p < s_hi && SV.clean(p)) ? SV.base(p) : 5; };
          int32_t best_len = 0, best_q = 0; int64_t best_s = 0;
          for (size_t t = w.lo; t <= w.hi; ++t) {
            int32_t q = seeds[t].second; int64_t r = (int64_t)seeds[t].first + q;
            int32_t len = K;
            while (q > 0 && q_at(q - 1) < 4 && q_at(q - 1) == s_at(r - 1)) { --q; --r; ++len; }
            while (q_at(q + len) < 4 && q_at(q + len) == s_at(r + len)) ++len;
            if (len > best_len || (len == best_len && q < best_q)) { best_len = len; best_q = q; best_s = r; }
          }
          FragHit h = frag_hsp(q_at, qlen, s_at, s_lo, s_hi, best_q, best_s, best_len, [](int32_t) { return true; });
          hits.push_back(h); hit_strand.push_back(strand);
          if ((int)chosen.size() >= TOP) break;
        }
      }
      // table order: best score first
      std::vector<int> ord(hits.size());
      for (size_t i = 0; i < ord.size(); ++i) ord[i] = (int)i;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return hits[a].score > hits[b].score; });
      for (int o : ord) {
        const FragHit& h = hits[o];
        const int strand = hit_strand[o];
        const int srec = record_of(S.rec_start.data(), (int)S.rec_start.size() - 1, h.ss);
        const int32_t so = S.rec_start[srec];
        // BLAST reports the query on its plus strand; a minus-strand hit has sstart > send
        int qs1, qe1, ss1, se1;
        if (!strand) { qs1 = h.qs + 1; qe1 = h.qe; ss1 = h.ss - so + 1; se1 = h.se - so; }
        else { qs1 = qlen - h.qe + 1; qe1 = qlen - h.qs; ss1 = h.se - so; se1 = h.ss - so + 1; }
        printf("frag%05d\t%s\t%d\t%d\t%.3f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%.2f\t%d\t%d\n", frag_no, S.ids[srec].c_str(), h.length, h.mismatch,
               100.0 * h.nident / h.length, h.nident, qlen, S.rec_start[srec + 1] - 1 - so, qs1, qe1, ss1, se1, h.nident, 100.0 * h.nident / h.length,
               h.gaps, h.score);
      }
    }
  }
  return 0;
}
