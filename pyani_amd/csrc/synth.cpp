// synth.cpp — deterministic synthetic bacterial-genome generator (test/bench DATA, not part of the ANI engine).
//
// Implements SURVEY.md §8(d) "Synthetic inputs": K = ceil(N/25) random ancestors with a per-ancestor GC
// fraction, and descendants derived by substitutions, short indels, 4 inversions + 4 translocations,
// split into 1..3 records, every 10th genome sprinkled with runs of 'N'.  Everything is a pure function
// of 64-bit integer hashes (splitmix64 finaliser), so the same (seed, index) gives the same genome on
// every machine and from every language that calls this library through ctypes.
//
// Built into libpgsynth.so (host-only, no HIP).  C ABI:
//   pgs_genome(set_seed, n_genomes, g, L, out, cap, rec_off, max_rec, n_rec) -> total bases or <0
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>

namespace {

inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t h3(uint64_t seed, uint64_t stream, uint64_t i) {
  return mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) + i);
}

const char BASES[4] = {'A', 'C', 'G', 'T'};
inline int code_of(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

// ancestor k of a set: i.i.d. bases with GC fraction in [0.35, 0.68]
void make_ancestor(uint64_t seed, uint64_t L, std::vector<char>& out) {
  const uint64_t gc_ppm = 350000ull + (h3(seed, 0, 0) % 330001ull);
  const uint64_t T = (gc_ppm << 32) / 1000000ull;  // threshold on a 32-bit uniform
  out.resize(L);
  for (uint64_t i = 0; i < L; ++i) {
    const uint64_t h = h3(seed, 1, i);
    const uint64_t r = h >> 32;
    const int bit = (int)(h & 1);
    out[i] = (r < T) ? (bit ? 'G' : 'C') : (bit ? 'A' : 'T');
  }
}

const uint32_t RATE_PPM[6] = {1000, 5000, 20000, 50000, 100000, 150000};

void revcomp_inplace(std::vector<char>& s, size_t a, size_t b) {  // [a,b)
  std::reverse(s.begin() + a, s.begin() + b);
  for (size_t i = a; i < b; ++i) {
    const int c = code_of(s[i]);
    if (c >= 0) s[i] = BASES[3 - c];
  }
}

}  // namespace

extern "C" {

// Number of ancestors used for a set of n genomes.
uint32_t pgs_n_ancestors(uint32_t n_genomes) { return (n_genomes + 24) / 25; }

// Generate genome g of a synthetic set.  `out` receives the concatenated record sequences (ASCII, no
// newlines); rec_off[0..n_rec] are the record boundaries (rec_off[0]=0, rec_off[n_rec]=total).
// Returns total length, or -(needed capacity) if cap is too small, or -1 on bad arguments.
int64_t pgs_genome(uint64_t set_seed, uint32_t n_genomes, uint32_t g, uint64_t L, char* out, uint64_t cap,
                   uint64_t* rec_off, uint32_t max_rec, uint32_t* n_rec_out) {
  if (!out || !rec_off || !n_rec_out || max_rec < 3 || n_genomes == 0 || g >= n_genomes || L < 64) return -1;
  const uint32_t K = pgs_n_ancestors(n_genomes);
  const uint32_t k = g % K;
  std::vector<char> anc;
  make_ancestor(set_seed + k, L, anc);

  const uint64_t gs = set_seed + 1000 + g;
  const uint64_t p_ppm = RATE_PPM[(g / K) % 6];
  const uint64_t Ts = (p_ppm << 32) / 1000000ull;
  const uint64_t Ti = Ts / 10;
  std::vector<char> s;
  s.reserve(L + L / 8 + 64);
  for (uint64_t i = 0; i < L; ++i) {
    const uint64_t h = h3(gs, 2, i);
    const uint64_t r = h >> 32;
    const int c = code_of(anc[i]);
    if (r < Ts) {  // substitution to one of the 3 other bases
      s.push_back(BASES[(c + 1 + (int)((h >> 8) % 3)) & 3]);
    } else if (r < Ts + Ti / 2) {
      // deletion: emit nothing
    } else if (r < Ts + Ti) {  // insertion after this base, length 1 + Geom(2/3), capped at 8
      s.push_back(anc[i]);
      int len = 1;
      uint64_t hb = h >> 8;
      while (len < 8 && (hb & 0xFF) < 171) { ++len; hb >>= 8; }
      for (int j = 0; j < len; ++j) s.push_back(BASES[h3(gs, 3, i * 8 + (uint64_t)j) & 3]);
    } else {
      s.push_back(anc[i]);
    }
  }
  // structural rearrangements: 4 inversions then 4 translocations, 20–200 kb (scaled down for small L)
  {
    const uint64_t n = s.size();
    const uint64_t lo = std::min<uint64_t>(20000, n / 20), hi = std::min<uint64_t>(200000, n / 8);
    for (int j = 0; j < 8 && hi > lo && n > 4 * hi; ++j) {
      const uint64_t hl = h3(gs, 4, 2 * j), hp = h3(gs, 4, 2 * j + 1);
      const uint64_t len = lo + hl % (hi - lo + 1);
      const uint64_t pos = hp % (s.size() - len);
      if (j < 4) {
        revcomp_inplace(s, pos, pos + len);
      } else {
        std::vector<char> seg(s.begin() + pos, s.begin() + pos + len);
        s.erase(s.begin() + pos, s.begin() + pos + len);
        const uint64_t dst = h3(gs, 5, j) % (s.size() + 1);
        s.insert(s.begin() + dst, seg.begin(), seg.end());
      }
    }
  }
  // every 10th genome: 20 runs of 1..50 'N'
  if (g % 10 == 9) {
    for (int j = 0; j < 20; ++j) {
      const uint64_t len = 1 + h3(gs, 6, 2 * j) % 50;
      if (s.size() <= len + 1) break;
      const uint64_t pos = h3(gs, 6, 2 * j + 1) % (s.size() - len);
      for (uint64_t t = 0; t < len; ++t) s[pos + t] = 'N';
    }
  }
  const uint64_t total = s.size();
  if (total > cap) return -(int64_t)total;
  // 1..3 records
  const uint32_t nrec = 1 + (uint32_t)(h3(gs, 7, 0) % 3);
  std::vector<uint64_t> cuts;
  cuts.push_back(0);
  for (uint32_t j = 1; j < nrec; ++j) cuts.push_back(1 + h3(gs, 7, j) % (total - 1));
  cuts.push_back(total);
  std::sort(cuts.begin(), cuts.end());
  cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
  std::memcpy(out, s.data(), total);
  *n_rec_out = (uint32_t)cuts.size() - 1;
  for (size_t j = 0; j < cuts.size(); ++j) rec_off[j] = cuts[j];
  return (int64_t)total;
}

}  // extern "C"
