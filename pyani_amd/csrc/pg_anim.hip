// pg_anim.hip — gfx950 kernels of the ANIm engine (kernel family 3 of BASELINE.json's north star): replaces the
// `nucmer --mum` + `delta-filter -1` processes pyani shells out to (pyani/anim.py:240-289) and the parse_delta
// reduction (anim.py:292-411) with an in-process pipeline over the 2-bit/1-bit packed genomes already resident in HBM.
//
//   A1 anim_index_kernel     20-mer hash table of the reference genome (open addressing, 8 B slots: kmer40 | pos24)
//   A2 anim_seed_kernel      one thread per query-strand position: table probe, left-maximality test, right extension
//                            -> maximal exact matches >= 20 (the only O(genome) stage; HBM/MALL-latency bound)
//   A3 anim_cluster_kernel   per (pair, strand): MUM filter, mgaps clustering, chain extraction      (pg_anim_core.h)
//   A4 anim_extend_kernel    per chain: gap fills + free forward extension, then backward extension towards the
//                            previous chain's end (banded affine DP, band in registers/scratch)
//   A5 anim_finish_kernel    per pair: stitch/fuse chains, 1-to-1 filter, parse_delta reduction -> pg_anim_result
//
// Round-1 state: correctness first — A3/A5 run one thread per unit (thousands of pairs give the parallelism), A2
// extends base by base.  The DESIGN.md section "ANIm" lists what is measured and what comes next.
#include "pg_internal.h"
#include "pg_anim_core.h"

using namespace pga;

namespace {

constexpr uint64_t SLOT_EMPTY = ~0ull;
constexpr int MAX_HITS = 64;  // probes per lookup; 20-mers with more copies than this in one genome are skipped

struct RefDesc {
  const uint32_t* codes;
  const uint32_t* mask;
  int32_t len;
  const int32_t* rec_start;  // n_rec + 1 entries
  int32_t n_rec;
  uint64_t* table;
  uint32_t table_mask;
};

struct UnitDesc {   // one (pair, query strand)
  const uint32_t* codes;
  const uint32_t* mask;
  int32_t len;
  const int32_t* rec_start;
  int32_t n_rec;
  int32_t strand;
  int32_t pair;  // index into the batch's pair list
  int32_t ref;   // index into the batch's reference list
};

__device__ __forceinline__ uint64_t mix40(uint64_t k) {
  k *= 0x9E3779B97F4A7C15ull;
  return k >> 24;
}

// 40-bit k-mer (20 bases, first base in the low bits) at stream position p, or false if a base is dirty
__device__ __forceinline__ bool kmer_at(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask, int32_t len,
                                        int32_t p, uint64_t& out) {
  if (p < 0 || p + MIN_MATCH > len) return false;
  const uint32_t mw = p >> 5, ms = p & 31;
  const uint64_t m = ((uint64_t)mask[mw] | ((uint64_t)mask[mw + 1] << 32)) >> ms;
  if ((m & 0xFFFFFull) != 0xFFFFFull) return false;
  const uint32_t cw = p >> 4, cs = 2 * (p & 15);
  const uint64_t lo = (uint64_t)codes[cw] | ((uint64_t)codes[cw + 1] << 32);
  uint64_t v = lo >> cs;
  if (cs > 24) v |= (uint64_t)codes[cw + 2] << (64 - cs);
  out = v & 0xFFFFFFFFFFull;
  return true;
}

// 16 bases starting at stream position p (p + 16 <= len): codes in 32 bits (first base low), clean bits in 16
__device__ __forceinline__ void get16(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask, int32_t p,
                                      uint32_t& c, uint32_t& m) {
  const uint32_t cw = p >> 4, cs = 2 * (p & 15);
  const uint64_t lo = (uint64_t)codes[cw] | ((uint64_t)codes[cw + 1] << 32);
  c = (uint32_t)(lo >> cs);
  const uint32_t mw = p >> 5, ms = p & 31;
  const uint64_t ml = (uint64_t)mask[mw] | ((uint64_t)mask[mw + 1] << 32);
  m = (uint32_t)(ml >> ms) & 0xFFFFu;
}

__device__ __forceinline__ uint64_t revcomp40(uint64_t k) {
  uint64_t x = __brevll(k);                                                   // pair order reversed, bits in pairs swapped
  x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);  // un-swap inside each pair
  return (~(x >> 24)) & 0xFFFFFFFFFFull;
}

__global__ __launch_bounds__(256) void anim_index_kernel(const RefDesc* __restrict__ refs) {
  const RefDesc R = refs[blockIdx.y];
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t k;
  if (!kmer_at(R.codes, R.mask, R.len, p, k)) return;
  const uint64_t val = (k << 24) | (uint32_t)p;
  uint32_t slot = (uint32_t)mix40(k) & R.table_mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&R.table[slot]), SLOT_EMPTY, val);
    if (prev == SLOT_EMPTY) break;
    slot = (slot + 1) & R.table_mask;
  }
}

// moff[u] .. moff[u+1]: this unit's slice of every per-match array (exact size, from a first count-only pass)
__global__ __launch_bounds__(256) void anim_seed_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                        Match* __restrict__ mem, uint32_t* __restrict__ mem_count,
                                                        const uint32_t* __restrict__ moff, int count_only) {
  const UnitDesc U = units[blockIdx.y];
  const RefDesc R = refs[U.ref];
  const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;  // strand position
  if (q + MIN_MATCH > U.len) return;
  uint64_t k;
  if (U.strand == 0) {
    if (!kmer_at(U.codes, U.mask, U.len, q, k)) return;
  } else {
    if (!kmer_at(U.codes, U.mask, U.len, U.len - MIN_MATCH - q, k)) return;  // forward window of the same bases
    k = revcomp40(k);
  }
  const SeqView RV{R.codes, R.mask, R.len};
  const StrandView QV{SeqView{U.codes, U.mask, U.len}, U.strand};
  uint32_t slot = (uint32_t)mix40(k) & R.table_mask;
  for (int probe = 0; probe < MAX_HITS; ++probe) {
    const uint64_t v = R.table[slot];
    if (v == SLOT_EMPTY) break;
    slot = (slot + 1) & R.table_mask;
    if ((v >> 24) != k) continue;
    const int32_t r = (int32_t)(v & 0xFFFFFFu);
    if (RV.clean(r - 1) && QV.clean(q - 1) && RV.base(r - 1) == QV.base(q - 1)) continue;  // not left-maximal
    if (count_only) { atomicAdd(&mem_count[blockIdx.y], 1u); continue; }
    int32_t L = MIN_MATCH;
    // right extension, 16 bases per step (word compare of the packed codes and masks), then base by base
    for (;;) {
      if (r + L + 16 > R.len || q + L + 16 > U.len) break;
      uint32_t rc_, rm_, qc_, qm_;
      get16(R.codes, R.mask, r + L, rc_, rm_);
      if (U.strand == 0) {
        get16(U.codes, U.mask, q + L, qc_, qm_);
      } else {
        uint32_t fc, fm;
        get16(U.codes, U.mask, U.len - 16 - (q + L), fc, fm);
        uint32_t x = __brev(fc);
        x = ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
        qc_ = ~x;
        qm_ = __brev(fm) >> 16;
      }
      const uint32_t x = rc_ ^ qc_;
      const uint32_t diff = (x | (x >> 1)) & 0x55555555u;
      const uint32_t bad = ~(rm_ & qm_) & 0xFFFFu;
      const int nd = diff ? (__ffs(diff) - 1) >> 1 : 16;
      const int nb = bad ? __ffs(bad) - 1 : 16;
      const int n = nd < nb ? nd : nb;
      L += n;
      if (n < 16) break;
    }
    while (RV.clean(r + L) && QV.clean(q + L) && RV.base(r + L) == QV.base(q + L)) ++L;
    const uint32_t at = atomicAdd(&mem_count[blockIdx.y], 1u);
    if (at < moff[blockIdx.y + 1] - moff[blockIdx.y]) mem[(size_t)moff[blockIdx.y] + at] = Match{r, q, L, U.strand};
  }
}

struct ClusterOut {   // per-match arrays are sliced by moff[] (a chain has >= 1 match, so chains fit the same slices)
  const uint32_t* moff; // [U + 1]
  Match* cm;
  Chain* chains;
  int32_t* n_chains;    // [U]
  int32_t* order;       // chains sorted by first-match ref start
  int32_t* prev_of;
  int32_t* next_of;
  int32_t* status;      // [P]
};

__global__ __launch_bounds__(64) void anim_cluster_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units, uint32_t n_units,
                                                          Match* __restrict__ mem, const uint32_t* __restrict__ mem_count,
                                                          int32_t* __restrict__ iscratch, ClusterOut O) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_units) return;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  O.n_chains[u] = 0;
  const size_t off = O.moff[u];
  const uint32_t cap = O.moff[u + 1] - O.moff[u];
  uint32_t n0 = mem_count[u];
  if (n0 > cap) { atomicOr(&O.status[U.pair], 1); n0 = cap; }
  if (n0 == 0) return;
  Match* m = mem + off;
  const int n = mum_filter(m, (int)n0, U.strand);
  int32_t* s = iscratch + off * 7;
  int32_t *rrec = s, *qrec = s + cap, *parent = s + 2 * (size_t)cap, *score = s + 3 * (size_t)cap, *from = s + 4 * (size_t)cap,
          *adj = s + 5 * (size_t)cap, *order = s + 6 * (size_t)cap;
  for (int i = 0; i < n; ++i) {
    rrec[i] = record_of(R.rec_start, R.n_rec, m[i].r);
    const int32_t qf = U.strand ? U.len - 1 - m[i].q : m[i].q;
    qrec[i] = record_of(U.rec_start, U.n_rec, qf);
  }
  int n_chains = 0, n_cm = 0;
  Chain* chains = O.chains + off;
  Match* cm = O.cm + off;
  mgaps_strand(m, n, U.strand, rrec, qrec, parent, score, from, adj, order, chains, n_chains, (int)cap, cm, n_cm, (int)cap);
  int32_t* co = O.order + off;
  for (int i = 0; i < n_chains; ++i) co[i] = i;
  heapsort(co, n_chains, [&](int a, int b) { return cm[chains[a].first].r < cm[chains[b].first].r; });
  chain_neighbours(chains, co, n_chains, O.prev_of + off, O.next_of + off);
  O.n_chains[u] = n_chains;
}

__device__ __forceinline__ int32_t from_lane_above(int32_t v, int32_t fill) {  // lane l <- lane l+1 (lane 63 <- fill)
  return __builtin_amdgcn_update_dpp(fill, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ int32_t from_lane_below(int32_t v, int32_t fill) {  // lane l <- lane l-1 (lane 0 <- fill)
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ long long wave_max64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// =====================================================================================================================
// A3, wave-cooperative: one WAVE per (pair, strand) unit.  Same results as the scalar statement (pga::mum_filter +
// pga::mgaps_strand, which the one-thread kernel above runs): stable LSD radix sorts instead of heapsorts, wave scans
// for the containment flags, a lock-free union-find, and a chain DP whose 64-deep look-back lives in the 64 lanes.
// =====================================================================================================================
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// Stable LSD radix sort of (key, val) pairs by `passes` 8-bit digits.  Result ends in (k0, v0) if passes is even,
// else in (k1, v1).  hist: 256 words of LDS.
__device__ void wave_radix_sort(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, int n, int passes, uint32_t* hist) {
  const int lane = threadIdx.x & 63;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 8 * pass;
    const uint32_t* ki = (pass & 1) ? k1 : k0;
    const uint32_t* vi = (pass & 1) ? v1 : v0;
    uint32_t* ko = (pass & 1) ? k0 : k1;
    uint32_t* vo = (pass & 1) ? v0 : v1;
    __syncthreads();
    for (int b = lane; b < 256; b += 64) hist[b] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      if (i < n) atomicAdd(&hist[(ki[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {  // exclusive scan of the 256 bins: 4 bins per lane
      uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      const uint32_t tot = c0 + c1 + c2 + c3;
      uint32_t incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      uint32_t ex = incl - tot;
      __syncthreads();
      hist[4 * lane] = ex; ex += c0;
      hist[4 * lane + 1] = ex; ex += c1;
      hist[4 * lane + 2] = ex; ex += c2;
      hist[4 * lane + 3] = ex;
    }
    __syncthreads();
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const bool act = i < n;
      const uint32_t key = act ? ki[i] : 0u, val = act ? vi[i] : 0u;
      const uint32_t d = (key >> shift) & 255u;
      uint64_t peers = __ballot(act);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint64_t vote = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? vote : ~vote;
      }
      const uint32_t rank = (uint32_t)__popcll(peers & lanemask_lt());
      uint32_t pos = 0;
      if (act) pos = hist[d] + rank;
      __syncthreads();
      if (act && rank == 0) hist[d] += (uint32_t)__popcll(peers);
      __syncthreads();
      if (act) { ko[pos] = key; vo[pos] = val; }
    }
  }
  __syncthreads();
}

// containment flags over elements in sorted order (ascending start, ties: longer first): flag[idx] |= 1 if an earlier
// element reaches at least as far, or if the next element has the same start and length.
__device__ void wave_containment_flags(const uint32_t* order, const int32_t* start, const int32_t* len, int n, int32_t* flag) {
  const int lane = threadIdx.x & 63;
  int32_t carry = -1;
  for (int base = 0; base < n; base += 64) {
    const int t = base + lane;
    const bool act = t < n;
    const uint32_t idx = act ? order[t] : 0u;
    const int32_t st = act ? start[idx] : 0, ln = act ? len[idx] : 0;
    const int32_t e = act ? st + ln : -1;
    int32_t incl = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(incl, o, 64); if (lane >= o && v > incl) incl = v; }
    int32_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = -1;
    const int32_t prevmax = excl > carry ? excl : carry;
    if (act) {
      bool f = e <= prevmax;
      if (!f && t + 1 < n) { const uint32_t nx = order[t + 1]; f = start[nx] == st && len[nx] == ln; }
      if (f) flag[idx] = 1;
    }
    const int32_t last = __shfl(incl, 63, 64);
    if (last > carry) carry = last;
  }
}

__device__ __forceinline__ int uf_find(int32_t* parent, int x) {
  for (;;) {
    const int p = parent[x];
    if (p == x) return x;
    x = p;
  }
}
__device__ __forceinline__ void uf_union(int32_t* parent, int a, int b) {   // larger root -> smaller root (deterministic roots)
  for (;;) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    if (atomicCAS(&parent[a], a, b) == a) return;
  }
}

__global__ __launch_bounds__(64) void anim_cluster_wave_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                               Match* __restrict__ mem, const uint32_t* __restrict__ mem_count,
                                                               int32_t* __restrict__ iscratch, ClusterOut O) {
  __shared__ uint32_t hist[256];
  const uint32_t u = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  const size_t off = O.moff[u];
  const uint32_t cap = O.moff[u + 1] - O.moff[u];
  uint32_t n0 = mem_count[u];
  if (n0 > cap) { if (lane == 0) atomicOr(&O.status[U.pair], 1); n0 = cap; }
  if (n0 == 0) { if (lane == 0) O.n_chains[u] = 0; return; }
  Match* m = mem + off;
  Match* cm = O.cm + off;
  // scratch slices (7 x cap ints): a..g
  int32_t* sa = iscratch + off * 7;
  int32_t *sb = sa + cap, *sc = sa + 2 * (size_t)cap, *sd = sa + 3 * (size_t)cap, *se = sa + 4 * (size_t)cap,
          *sf = sa + 5 * (size_t)cap, *sg = sa + 6 * (size_t)cap;
  const int n_in = (int)n0;
  // ---- MUM filter -----------------------------------------------------------------------------------------------
  // SoA copies: se = r, sf = q, sg = len; flags in sd
  for (int i = lane; i < n_in; i += 64) { const Match t = m[i]; se[i] = t.r; sf[i] = t.q; sg[i] = t.len; sd[i] = 0; }
  __syncthreads();
  uint32_t *k0 = (uint32_t*)sa, *v0 = (uint32_t*)sb, *k1 = (uint32_t*)sc;
  uint32_t* v1 = (uint32_t*)cm;   // cm is free until the chains are written (16 B per entry >= 4 B)
  for (int side = 0; side < 2; ++side) {
    const int32_t* start = side == 0 ? sf : se;   // query intervals first, then reference intervals
    // sort by (start asc, len desc): LSD = len-desc key first, then start
    for (int i = lane; i < n_in; i += 64) { const uint32_t l = (uint32_t)sg[i]; k0[i] = 0xFFFFFFu - (l > 0xFFFFFFu ? 0xFFFFFFu : l); v0[i] = (uint32_t)i; }
    wave_radix_sort(k0, v0, k1, v1, n_in, 3, hist);            // result in (k1, v1)
    for (int i = lane; i < n_in; i += 64) k1[i] = (uint32_t)start[v1[i]];
    __syncthreads();
    wave_radix_sort(k1, v1, k0, v0, n_in, 4, hist);            // 4 passes (even): result back in (k1, v1)
    wave_containment_flags(v1, start, sg, n_in, sd);
    __syncthreads();
  }
  // survivors in q order (distinct q among survivors): sort indices by q once more, compact
  for (int i = lane; i < n_in; i += 64) { k0[i] = (uint32_t)sf[i]; v0[i] = (uint32_t)i; }
  wave_radix_sort(k0, v0, k1, v1, n_in, 4, hist);              // result in (k0, v0)
  int n = 0;
  {
    // compact into sb? v0 aliases sb; write survivors' Match into cm-temp is not possible (v1 lives there) -> use m itself
    // two-step: first the survivor index list into k1 (sc), then gather through registers chunk by chunk into cm, copy back
    for (int base = 0; base < n_in; base += 64) {
      const int t = base + lane;
      const bool keep = t < n_in && sd[v0[t]] == 0;
      const uint64_t b = __ballot(keep);
      if (keep) k1[n + __popcll(b & lanemask_lt())] = v0[t];
      n += (int)__popcll(b);
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) { const uint32_t idx = k1[i]; cm[i] = Match{se[idx], sf[idx], sg[idx], U.strand}; }
    __syncthreads();
    for (int i = lane; i < n; i += 64) m[i] = cm[i];
    __syncthreads();
  }
  // ---- clustering (mgaps) ------------------------------------------------------------------------------------------
  int32_t *rrec = sa, *qrec = sb, *parent = sc, *score = sd, *from = se, *adj = sf, *order = sg;
  for (int i = lane; i < n; i += 64) {
    rrec[i] = record_of(R.rec_start, R.n_rec, m[i].r);
    const int32_t qf = U.strand ? U.len - 1 - m[i].q : m[i].q;
    qrec[i] = record_of(U.rec_start, U.n_rec, qf);
    parent[i] = i;
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    const Match mi = m[i];
    const int32_t iend = mi.q + mi.len, idiag = mi.q - mi.r;
    for (int j = i + 1; j < n; ++j) {
      const Match mj = m[j];
      const int32_t sep = mj.q - iend;
      if (sep > MAX_GAP) break;
      if (rrec[i] != rrec[j] || qrec[i] != qrec[j]) continue;
      int32_t dd = (mj.q - mj.r) - idiag;
      if (dd < 0) dd = -dd;
      int32_t lim = (int32_t)(DIAG_FACTOR * sep);
      if (lim < DIAG_DIFF) lim = DIAG_DIFF;
      if (dd <= lim) uf_union(parent, i, j);
    }
  }
  __threadfence_block();
  __syncthreads();
  {  // group by root (stable: q order inside a cluster): radix sort of (root, index)
    uint32_t *rk0 = (uint32_t*)score, *rv0 = (uint32_t*)from, *rk1 = (uint32_t*)adj, *rv1 = (uint32_t*)order;
    for (int i = lane; i < n; i += 64) { rk0[i] = (uint32_t)uf_find(parent, i); rv0[i] = (uint32_t)i; }
    __syncthreads();
    wave_radix_sort(rk0, rv0, rk1, rv1, n, 4, hist);           // result in (rk0, rv0) = (score, from) slices
    for (int i = lane; i < n; i += 64) { parent[i] = (int32_t)rk0[i]; }   // parent[] now = root id of the i-th element in grouped order
    __syncthreads();
    for (int i = lane; i < n; i += 64) order[i] = (int32_t)rv0[i];
    __syncthreads();
  }
  // ---- chain extraction per cluster ----------------------------------------------------------------------------------
  // grouped list: order[t] = match index, parent[t] = its root.  score/from/adj are indexed by LIST POSITION here.
  // `lst` (compacted working list of the current cluster) lives in adj's slice after use... keep it simple: a cluster's
  // live entries are kept contiguous in order[g0 .. g0+live).
  int n_chains = 0, n_cm = 0;
  Chain* chains = O.chains + off;
  int g0 = 0;
  while (g0 < n) {
    int g1 = g0 + 1;
    {  // cluster end: first position whose root differs (wave search)
      const int32_t root = parent[g0];
      for (;;) {
        const int t = g1 + lane;
        const uint64_t diff = __ballot(t >= n || parent[t] != root);
        if (diff) { g1 += __ffsll((long long)diff) - 1; break; }
        g1 += 64;
      }
    }
    int live = g1 - g0;
    while (live > 0) {
      // DP over the live entries order[g0 .. g0+live): lane l holds the entry at position k-1-l (sliding window)
      int32_t wr = 0, wq = 0, wl = 0, wsc = NEG_INF;   // window registers: r, q, len, score of predecessor k-1-lane
      int32_t best_sc = NEG_INF, best_k = -1;
      for (int k = 0; k < live; ++k) {
        const Match mi = m[order[g0 + k]];
        // candidate through my predecessor
        int32_t cand = NEG_INF, ol = 0;
        if (lane < k && wsc > NEG_INF / 2) {
          ol = wr + wl - mi.r;
          if (ol < 0) ol = 0;
          const int32_t ol2 = wq + wl - mi.q;
          if (ol2 > ol) ol = ol2;
          int32_t dd = (mi.q - mi.r) - (wq - wr);
          if (dd < 0) dd = -dd;
          cand = wsc + mi.len - (ol + dd);
        }
        // best candidate: max cand, ties -> nearest predecessor (smallest lane)
        long long key = (((long long)cand + (1ll << 30)) << 8) | (long long)(63 - lane);
        key = wave_max64(key);
        const int32_t bc = (int32_t)((key >> 8) - (1ll << 30));
        const int bl = 63 - (int)(key & 63);
        int32_t sc_k = mi.len, fr_k = -1, ad_k = 0;
        if (bc > sc_k) { sc_k = bc; fr_k = k - 1 - bl; ad_k = __shfl(ol, bl, 64); }
        if (lane == 0) { score[g0 + k] = sc_k; from[g0 + k] = fr_k; adj[g0 + k] = ad_k; }
        if (sc_k > best_sc) { best_sc = sc_k; best_k = k; }
        // slide the window: lane l <- lane l-1, lane 0 <- entry k
        wr = from_lane_below(wr, mi.r); wq = from_lane_below(wq, mi.q); wl = from_lane_below(wl, mi.len);
        wsc = from_lane_below(wsc, sc_k);
      }
      __threadfence_block();
      __syncthreads();
      // walk the best chain (lane 0), emit if long enough, mark removed (from = -2)
      int32_t total = 0, cnt = 0;
      if (lane == 0) {
        for (int k = best_k; k >= 0; k = from[g0 + k]) { total += m[order[g0 + k]].len; ++cnt; }
      }
      total = __shfl(total, 0, 64); cnt = __shfl(cnt, 0, 64);
      const bool emit = total >= MIN_CLUSTER && n_chains < (int)cap && n_cm + cnt <= (int)cap;
      if (lane == 0) {
        if (emit) {
          const int first_idx = order[g0 + best_k];
          Chain c;
          c.first = n_cm; c.count = cnt; c.strand = U.strand; c.rrec = rrec[first_idx]; c.qrec = qrec[first_idx];
          chains[n_chains] = c;
        }
        int pos = n_cm + cnt;
        for (int k = best_k; k >= 0;) {
          const int nx = from[g0 + k];
          if (emit) {
            Match t = m[order[g0 + k]];
            const int32_t a = adj[g0 + k];
            t.r += a; t.q += a; t.len -= a;
            cm[--pos] = t;
          }
          from[g0 + k] = -2;
          k = nx;
        }
      }
      if (emit) { n_chains += 1; n_cm += cnt; }
      __threadfence_block();
      __syncthreads();
      // compact the live list (drop removed entries), preserving order
      int kept = 0;
      for (int base = 0; base < live; base += 64) {
        const int k = base + lane;
        const bool keep = k < live && from[g0 + k] != -2;
        const int32_t idx = k < live ? order[g0 + k] : 0;
        const uint64_t b = __ballot(keep);
        __syncthreads();
        if (keep) order[g0 + kept + __popcll(b & lanemask_lt())] = idx;
        kept += (int)__popcll(b);
        __syncthreads();
      }
      live = kept;
    }
    g0 = g1;
  }
  // ---- chains in reference order + neighbours ------------------------------------------------------------------------
  int32_t* co = O.order + off;
  {
    uint32_t *ck0 = (uint32_t*)score, *cv0 = (uint32_t*)from, *ck1 = (uint32_t*)adj, *cv1 = (uint32_t*)order;
    __syncthreads();
    for (int i = lane; i < n_chains; i += 64) { ck0[i] = (uint32_t)cm[chains[i].first].r; cv0[i] = (uint32_t)i; }
    __syncthreads();
    wave_radix_sort(ck0, cv0, ck1, cv1, n_chains, 4, hist);    // result in (ck0, cv0)
    for (int i = lane; i < n_chains; i += 64) co[i] = (int32_t)cv0[i];
    __threadfence_block();
    __syncthreads();
  }
  int32_t* prev_of = O.prev_of + off;
  int32_t* next_of = O.next_of + off;
  for (int k = lane; k < n_chains; k += 64) {
    const int c = co[k];
    int p = -1, q = -1;
    for (int kk = k - 1; kk >= 0 && kk >= k - 8 && p < 0; --kk)
      if (chains[co[kk]].rrec == chains[c].rrec && chains[co[kk]].qrec == chains[c].qrec) p = co[kk];
    for (int kk = k + 1; kk < n_chains && kk <= k + 8 && q < 0; ++kk)
      if (chains[co[kk]].rrec == chains[c].rrec && chains[co[kk]].qrec == chains[c].qrec) q = co[kk];
    prev_of[c] = p;
    next_of[c] = q;
  }
  if (lane == 0) O.n_chains[u] = n_chains;
}

__device__ __forceinline__ void chain_bounds(const RefDesc& R, const UnitDesc& U, const Chain& c, int32_t& r_lo, int32_t& r_hi,
                                             int32_t& q_lo, int32_t& q_hi) {
  r_lo = R.rec_start[c.rrec]; r_hi = R.rec_start[c.rrec + 1] - 1;
  q_lo = U.rec_start[c.qrec]; q_hi = U.rec_start[c.qrec + 1] - 1;
  if (U.strand) { const int32_t a = U.len - q_hi, b = U.len - q_lo; q_lo = a; q_hi = b; }
}

// ---- wave-cooperative banded DP: the 64 lanes of a wave ARE the 64 diagonals of the band ----------------------------
// Same cells, checks and tie-breaks as pga::extend_banded (pg_anim_core.h); neighbours' cells arrive through DPP
// wave shifts, so a step costs a handful of VALU ops per lane and no LDS.  All lanes return the same result.
// LDS staging of the two sequences for one wave: base codes (0-3, 4 = dirty / out of range) of consumed indices
// t = 0, 1, 2, ... in a 256-entry ring.  Cell (i, j) compares ring_r[i-1] with ring_q[j-1]; on anti-diagonal d every lane
// needs indices within [d/2 - 17, d/2 + 15], so the ring is topped up 64 entries at a time, one base per lane.
struct WaveSeq {
  uint8_t* ring_r;
  uint8_t* ring_q;
  int32_t loaded;  // indices [0, loaded) have been staged (uniform)
};

__device__ __forceinline__ void wave_seq_fill(WaveSeq& ws, const SeqView& R, const StrandView& Q, int64_t r0, int64_t q0, int dir,
                                              int32_t rmax, int32_t qmax, int lane) {
  const int32_t t = ws.loaded + lane;
  uint8_t rb = 4, qb = 4;
  if (t < rmax) { const int64_t rp = dir > 0 ? r0 + t : r0 - 1 - t; if (R.clean(rp)) rb = (uint8_t)R.base(rp); }
  if (t < qmax) { const int64_t qp = dir > 0 ? q0 + t : q0 - 1 - t; if (Q.clean(qp)) qb = (uint8_t)Q.base(qp); }
  ws.ring_r[t & 255] = rb;
  ws.ring_q[t & 255] = qb;
  ws.loaded += 64;
}

__device__ ExtResult extend_wave(const SeqView& R, const StrandView& Q, int64_t r0, int64_t q0, int dir, int32_t rmax,
                                 int32_t qmax, int32_t tr, int32_t tq) {
  constexpr int W = BAND / 2;
  static_assert(BAND == 64, "one lane per diagonal");
  __shared__ uint8_t s_ring[2][256];
  const int lane = threadIdx.x & 63;
  ExtResult res{0, 0, 0, 0, 0};
  bool targeted = tr >= 0;
  int koff = 0;   // band placement, see pga::extend_banded
  if (targeted) {
    koff = (tq - tr) / 2;
    if (koff > W - 2) koff = W - 2;
    if (koff < -(W - 2)) koff = -(W - 2);
    const int lt = (tq - tr) - koff + W;
    if (lt < 0 || lt >= BAND || tr > rmax || tq > qmax) { targeted = false; koff = 0; }
  }
  if (targeted && tr == 0 && tq == 0) { res.reached = 1; return res; }
  const int k = lane - W + koff;
  DpCell cur{NEG_INF, 0, NEG_INF, 0, NEG_INF, 0};
  int32_t bs = NEG_INF, bd = 0, be = 0;
  if (lane == W - koff) { cur.h = 0; bs = 0; }
  const int32_t d_end = targeted ? tr + tq : rmax + qmax;
  constexpr long long BIAS = 1ll << 30;
  WaveSeq ws{s_ring[0], s_ring[1], 0};
  __syncthreads();  // previous user of the ring (same wave) is done
  wave_seq_fill(ws, R, Q, r0, q0, dir, rmax, qmax, lane);
  wave_seq_fill(ws, R, Q, r0, q0, dir, rmax, qmax, lane);
  __syncthreads();
  // Break rule with PER-STEP semantics (as the scalar code) at the price of one wave reduction every CHECK steps:
  // g_known / t_prev = global best score and its anti-diagonal as of the last check; every lane remembers the first
  // step since then at which it matched or beat g_known (fimp) and a snapshot of its best as of the last check.
  constexpr int CHECK = 16;
  int32_t g_known = 0, t_prev = 0, fimp = 0x7FFFFFFF;
  int32_t sbs = bs, sbd = bd, sbe = be;
  for (int32_t d = 1; d <= d_end; ++d) {
    if ((d >> 1) + 36 > ws.loaded) {   // uniform; covers the diagonals of a shifted band (|koff| <= 30)
      wave_seq_fill(ws, R, Q, r0, q0, dir, rmax, qmax, lane);
      __syncthreads();
    }
    const int32_t up_h = from_lane_above(cur.h, NEG_INF), up_he = from_lane_above(cur.he, 0);
    const int32_t up_x = from_lane_above(cur.x, NEG_INF), up_xe = from_lane_above(cur.xe, 0);
    const int32_t lf_h = from_lane_below(cur.h, NEG_INF), lf_he = from_lane_below(cur.he, 0);
    const int32_t lf_y = from_lane_below(cur.y, NEG_INF), lf_ye = from_lane_below(cur.ye, 0);
    if (!((d + k) & 1)) {
      const int32_t i = (d - k) / 2, j = (d + k) / 2;
      if (i < 0 || j < 0 || i > rmax || j > qmax) {
        cur = DpCell{NEG_INF, 0, NEG_INF, 0, NEG_INF, 0};
      } else {
        const bool has_up = i >= 1 && lane + 1 < BAND, has_left = j >= 1 && lane >= 1, has_diag = i >= 1 && j >= 1;
        bool ok = false;
        if (has_diag) {
          const uint8_t rb = ws.ring_r[(i - 1) & 255], qb = ws.ring_q[(j - 1) & 255];
          ok = rb == qb && rb < 4;
        }
        cur = dp_cell(has_up, up_h, up_he, up_x, up_xe, has_left, lf_h, lf_he, lf_y, lf_ye, has_diag, cur.h, cur.he, ok);
        if (cur.h > NEG_INF / 2) {
          if (cur.h > bs || (cur.h == bs && d >= bd)) { bs = cur.h; bd = d; be = cur.he; }
          if (cur.h >= g_known && fimp == 0x7FFFFFFF) fimp = d;
        }
      }
    }
    if ((d % CHECK) == 0 || d == d_end) {
      const long long key = wave_max64((((long long)bs + BIAS) << 32) | (uint32_t)bd);  // max score, ties: larger d
      const int32_t g = (int32_t)((key >> 32) - BIAS), t = (int32_t)(key & 0xFFFFFFFFll);
      const int32_t b = t_prev + BREAK_LEN + 1;      // step at which the per-step rule fires without an improvement
      int32_t d1 = fimp;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const int32_t v = __shfl_xor(d1, o, 64); d1 = v < d1 ? v : d1; }
      if (b <= d && d1 > b) {                        // it fired before the first improvement of this interval
        bs = sbs; bd = sbd; be = sbe;                // results as of the last check (nothing global changed until b)
        break;
      }
      if (!__any(cur.h > NEG_INF / 2)) break;
      g_known = g; t_prev = t; fimp = 0x7FFFFFFF;
      sbs = bs; sbd = bd; sbe = be;
    }
    if (targeted && d == d_end) {
      const int lt = (tq - tr) - koff + W;
      const int32_t th = __shfl(cur.h, lt, 64), the = __shfl(cur.he, lt, 64);
      if (th > NEG_INF / 2) { res.di = tr; res.dj = tq; res.score = th; res.errors = the; res.reached = 1; return res; }
    }
  }
  // best cell: max score, ties -> larger d, then larger diagonal
  const long long key = wave_max64((((long long)bs + BIAS) << 32) | ((long long)(uint32_t)bd << 6) | (long long)lane);
  const int bl = (int)(key & 63);
  const int32_t gd = (int32_t)((key >> 6) & 0x3FFFFFF);
  res.score = (int32_t)((key >> 32) - BIAS);
  res.errors = __shfl(be, bl, 64);
  const int kk = bl - W + koff;
  res.di = (gd - kk) / 2; res.dj = (gd + kk) / 2;
  return res;
}

// Wave version of pga::thin_rect_errors: full DP of a small rectangle whose SHORT side (<= 63) is spread over the
// lanes; skewed wavefront: lane t works on line t of the short side, step s handles the cells with long-side index
// s - t, so the three predecessors are the lane's own previous cell and the lower neighbour's last two cells.
__device__ int32_t thin_rect_errors_wave(const SeqView& R, const StrandView& Q, int64_t r0, int32_t n, int64_t q0, int32_t m) {
  if (n < 0 || m < 0 || n > THIN_LONG || m > THIN_LONG || (n > THIN_MAX && m > THIN_MAX)) return -1;
  __shared__ uint8_t s_long[THIN_LONG + 1];
  const int lane = threadIdx.x & 63;
  const bool lanes_q = m <= THIN_MAX;            // lanes over query columns (rows = ref) or over ref rows
  const int32_t n_short = lanes_q ? m : n, n_long = lanes_q ? n : m;
  __syncthreads();
  for (int32_t t = lane; t < n_long; t += 64) {  // stage the long side's bases
    uint8_t b = 4;
    if (lanes_q) { if (R.clean(r0 + t)) b = (uint8_t)R.base(r0 + t); }
    else { if (Q.clean(q0 + t)) b = (uint8_t)Q.base(q0 + t); }
    s_long[t] = b;
  }
  uint8_t mine = 4;                              // this lane's base on the short side (line `lane`, 1-based)
  if (lane >= 1 && lane <= n_short) {
    if (lanes_q) { if (Q.clean(q0 + lane - 1)) mine = (uint8_t)Q.base(q0 + lane - 1); }
    else { if (R.clean(r0 + lane - 1)) mine = (uint8_t)R.base(r0 + lane - 1); }
  }
  __syncthreads();
  const DpCell dead{NEG_INF, 0, NEG_INF, 0, NEG_INF, 0};
  DpCell cur = dead, prev = dead;
  for (int32_t s = 0; s <= n_short + n_long; ++s) {
    // neighbour (lane - 1): its cell of step s-1 (same long index) and of step s-2 (long index - 1)
    DpCell nb1, nb2;
    nb1.h = from_lane_below(cur.h, NEG_INF); nb1.he = from_lane_below(cur.he, 0);
    nb1.x = from_lane_below(cur.x, NEG_INF); nb1.xe = from_lane_below(cur.xe, 0);
    nb1.y = from_lane_below(cur.y, NEG_INF); nb1.ye = from_lane_below(cur.ye, 0);
    nb2.h = from_lane_below(prev.h, NEG_INF); nb2.he = from_lane_below(prev.he, 0);
    const int32_t u = s - lane;                  // long-side index of this lane's cell
    if (lane <= n_short && u >= 0 && u <= n_long) {
      DpCell c;
      if (lane == 0 && u == 0) {
        c = DpCell{0, 0, NEG_INF, 0, NEG_INF, 0};
      } else {
        const bool ok = lane >= 1 && u >= 1 && mine < 4 && s_long[u - 1] == mine;
        if (lanes_q) {   // i = u (rows, long), j = lane: up = own previous cell, left = neighbour (step s-1), diag = neighbour (s-2)
          c = dp_cell(u >= 1, cur.h, cur.he, cur.x, cur.xe, lane >= 1, nb1.h, nb1.he, nb1.y, nb1.ye, u >= 1 && lane >= 1, nb2.h, nb2.he, ok);
        } else {         // i = lane (rows, short), j = u: up = neighbour (step s-1), left = own previous cell, diag = neighbour (s-2)
          c = dp_cell(lane >= 1, nb1.h, nb1.he, nb1.x, nb1.xe, u >= 1, cur.h, cur.he, cur.y, cur.ye, u >= 1 && lane >= 1, nb2.h, nb2.he, ok);
        }
      }
      prev = cur;
      cur = c;
    }
  }
  return __shfl(cur.he, n_short, 64);            // cell (n, m) lives in lane n_short after the last step
}

__device__ int32_t gap_errors_wave(const SeqView& R, const StrandView& Q, int64_t r0, int32_t n, int64_t q0, int32_t m) {
  if (n == 0) return m;
  if (m == 0) return n;
  if (n == m && n <= 2) {
    int32_t err = 0;
    for (int32_t t = 0; t < n; ++t) err += (R.clean(r0 + t) && Q.clean(q0 + t) && R.base(r0 + t) == Q.base(q0 + t)) ? 0 : 1;
    return err;
  }
  const ExtResult e = extend_wave(R, Q, r0, q0, +1, n, m, n, m);
  if (e.reached) return e.errors;
  int32_t kq = n < m ? n : m, err = (n > m ? n - m : m - n);
  for (int32_t t = 0; t < kq; ++t) err += (R.clean(r0 + t) && Q.clean(q0 + t) && R.base(r0 + t) == Q.base(q0 + t)) ? 0 : 1;
  return err;
}

// One WAVE per chain (work list wl: unit, chain).  phase 0: gap fills + free forward extension (extend_chain_fwd);
// phase 1: backward extension towards the previous chain's forward end (extend_chain_bwd).
__global__ __launch_bounds__(64) void anim_extend_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units,
                                                         ClusterOut O, const uint2* __restrict__ wl,
                                                         ChainFwd* __restrict__ fw, ChainBwd* __restrict__ bw, int phase) {
  const uint32_t u = wl[blockIdx.x].x;
  const int32_t c = (int32_t)wl[blockIdx.x].y;
  const UnitDesc U = units[u];
  const RefDesc R = refs[U.ref];
  const SeqView RV{R.codes, R.mask, R.len};
  const StrandView QV{SeqView{U.codes, U.mask, U.len}, U.strand};
  const size_t off = O.moff[u];
  const Chain ch = O.chains[off + c];
  int32_t r_lo, r_hi, q_lo, q_hi;
  chain_bounds(R, U, ch, r_lo, r_hi, q_lo, q_hi);
  ChainFwd* fwu = fw + off;
  const Match* cm = O.cm + off;
  if (phase == 0) {
    ChainFwd e;
    const Match f = cm[ch.first];
    e.first_r = f.r; e.first_q = f.q;
    int32_t inner = 0, er = f.r + f.len, eq = f.q + f.len;
    for (int kq = 1; kq < ch.count; ++kq) {
      Match t = cm[ch.first + kq];
      int32_t trim = er - t.r;
      if (eq - t.q > trim) trim = eq - t.q;
      if (trim > 0) { t.r += trim; t.q += trim; t.len -= trim; }
      if (t.len <= 0) continue;
      inner += gap_errors_wave(RV, QV, er, t.r - er, eq, t.q - eq);
      er = t.r + t.len; eq = t.q + t.len;
    }
    e.inner_err = inner;
    e.lr = er; e.lq = eq;
    int32_t nr, nq;
    e.target = pick_forward_target(O.chains + off, cm, O.next_of + off, c, er, eq, nr, nq);
    forward_extension([&](int32_t cr, int32_t cq, int32_t rmax, int32_t qmax, int32_t tr, int32_t tq) {
                        return extend_wave(RV, QV, cr, cq, +1, rmax, qmax, tr, tq); },
                      er, eq, r_hi, q_hi, nr, nq, e.re, e.qe, e.err_fwd, e.reached);
    if ((threadIdx.x & 63) == 0) fwu[c] = e;
  } else {
    const int32_t p = O.prev_of[off + c];
    const int32_t first_r = fwu[c].first_r, first_q = fwu[c].first_q;
    const int32_t prev_re = p >= 0 ? fwu[p].re : -1, prev_qe = p >= 0 ? fwu[p].qe : -1;
    if (p >= 0 && ((fwu[p].reached && fwu[p].target == c) ||
                   (fwu[p].first_r <= first_r && fwu[p].first_q <= first_q && prev_re >= fwu[c].lr && prev_qe >= fwu[c].lq))) {
      if ((threadIdx.x & 63) == 0) bw[off + c] = ChainBwd{first_r, first_q, 0, 0};  // will be shadowed
      return;
    }
    int32_t tr = -1, tq = -1;
    if (prev_re >= 0 && first_r >= prev_re && first_q >= prev_qe) { tr = first_r - prev_re; tq = first_q - prev_qe; }
    if (p >= 0) {   // never search into the previous chain's matches (same rule as pga::extend_chain_bwd)
      const int32_t plr = fwu[p].lr, plq = fwu[p].lq;
      if (plr <= first_r && plq <= first_q) {   // collinear predecessor only
        if (plr > r_lo) r_lo = plr;
        if (plq > q_lo) q_lo = plq;
      }
    }
    const ExtResult b = extend_wave(RV, QV, first_r, first_q, -1, cap_ext(first_r - r_lo, MAX_EXT_BWD), cap_ext(first_q - q_lo, MAX_EXT_BWD), tr, tq);
    ChainBwd e;
    e.rs = first_r - b.di; e.qs = first_q - b.dj; e.err_back = b.errors;
    e.reached = (tr >= 0 && b.reached) ? 1 : 0;
    bridge_junction(e, prev_re, prev_qe, tr, tq, [&](int32_t r0, int32_t n, int32_t q0, int32_t m) {
      return thin_rect_errors_wave(RV, QV, r0, n, q0, m); });
    if ((threadIdx.x & 63) == 0) bw[off + c] = e;
  }
}

struct FinishScratch {   // per-alignment arrays: pair p owns the slice [moff[2p], moff[2p+2])
  Aln* alns;
  int32_t* a_rrec;
  int32_t* a_qrec;
  int32_t* idx;
  int32_t* from;
  double* sc;
  int32_t* aln_of;  // per chain
};

__global__ __launch_bounds__(64) void anim_finish_kernel(const RefDesc* __restrict__ refs, const UnitDesc* __restrict__ units, uint32_t n_pairs,
                                                         ClusterOut O, const ChainFwd* __restrict__ fw, const ChainBwd* __restrict__ bw,
                                                         FinishScratch S, int filter_1to1, pg_anim_result* __restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const RefDesc R = refs[units[2 * p].ref];
  const size_t poff = O.moff[2 * p];
  const int cap_a = (int)(O.moff[2 * p + 2] - O.moff[2 * p]);
  Aln* alns = S.alns + poff;
  int32_t* a_rrec = S.a_rrec + poff;
  int32_t* a_qrec = S.a_qrec + poff;
  int32_t* idx = S.idx + poff;
  int32_t* from = S.from + poff;
  double* sc = S.sc + poff;
  int n = 0;
  for (int strand = 0; strand < 2; ++strand) {
    const uint32_t u = 2 * p + strand;  // units are laid out pair-major: (pair, fwd), (pair, rev)
    const UnitDesc U = units[u];
    const size_t off = O.moff[u];
    const int before = n;
    n = stitch_chains(fw + off, bw + off, O.cm + off, O.chains + off, O.order + off, O.prev_of + off, O.next_of + off,
                      O.n_chains[u], strand, S.aln_of + off, alns, n, cap_a);
    for (int i = before; i < n; ++i) {
      Aln& a = alns[i];
      a_rrec[i] = record_of(R.rec_start, R.n_rec, a.rs);
      if (strand) { const int32_t qs = U.len - a.qe, qe = U.len - a.qs; a.qs = qs; a.qe = qe; }  // forward coordinates
      a_qrec[i] = record_of(U.rec_start, U.n_rec, a.qs);
    }
  }
  if (filter_1to1) {
    lis_filter(alns, n, 0, a_rrec, idx, sc, from);
    lis_filter(alns, n, 1, a_qrec, idx, sc, from);
  } else {
    for (int i = 0; i < n; ++i) alns[i].keep = 3;
  }
  const PairResult r = reduce_pair(alns, n, a_rrec, a_qrec, idx);
  pg_anim_result o;
  o.ref_aln_len = r.ref_aln_len;
  o.qry_aln_len = r.qry_aln_len;
  o.sim_errors = r.sim_errors;
  o.n_alignments = r.n_alignments;
  o.identity = r.aligned > 0 ? (double)r.weighted / (double)r.aligned : 0.0;  // int/int true division (anim.py:396)
  o.status = O.status[p] ? PG_E_CAPACITY : (r.n_alignments == 0 ? PG_ANIM_NO_ALIGNMENT : 0);
  o.reserved = 0;
  out[p] = o;
}

// reduction of caller-supplied alignment records (pg_anim_reduce): one thread per pair
__global__ __launch_bounds__(64) void anim_reduce_kernel(uint32_t n_pairs, const uint64_t* __restrict__ offsets, Aln* alns,
                                                         const int32_t* __restrict__ rgrp, const int32_t* __restrict__ qgrp,
                                                         int32_t* idx, int32_t* from, double* sc, int apply_filter,
                                                         pg_anim_result* __restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const uint64_t o = offsets[p];
  const int n = (int)(offsets[p + 1] - o);
  if (apply_filter) {
    lis_filter(alns + o, n, 0, rgrp + o, idx + o, sc + o, from + o);
    lis_filter(alns + o, n, 1, qgrp + o, idx + o, sc + o, from + o);
  }
  const PairResult r = reduce_pair(alns + o, n, rgrp + o, qgrp + o, idx + o);
  pg_anim_result res;
  res.ref_aln_len = r.ref_aln_len; res.qry_aln_len = r.qry_aln_len; res.sim_errors = r.sim_errors;
  res.n_alignments = r.n_alignments;
  res.identity = r.aligned > 0 ? (double)r.weighted / (double)r.aligned : 0.0;
  res.status = r.n_alignments == 0 ? PG_ANIM_NO_ALIGNMENT : 0;
  res.reserved = 0;
  out[p] = res;
}

template <typename T>
int anim_alloc(pg_ctx* ctx, T*& p, size_t n) {
  PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
  return PG_OK;
}

}  // namespace

// ---- host driver ---------------------------------------------------------------------------------------------------
// One batch of ordered pairs (any mix of references): ref_ids[i] = nucmer's reference (pyani's query genome, anim.py:280).
// Scratch lives in the context and only grows.  The per-unit kernels are latency-bound single-thread code, so the
// batch should be as large as memory allows: thousands of units in flight are what fills the GPU.
namespace {
struct AnimScratch {
  size_t units = 0, pairs = 0, refs = 0, table_slots = 0, recs = 0, wl = 0, matches = 0;
  uint64_t* table = nullptr;
  int32_t* recs_d = nullptr;
  RefDesc* refs_d = nullptr;
  UnitDesc* units_d = nullptr;
  uint32_t *mem_count = nullptr, *moff = nullptr;
  int32_t *nch = nullptr, *status = nullptr;
  pg_anim_result* out = nullptr;
  // per-match arrays (sliced by moff)
  Match *mem = nullptr, *cm = nullptr;
  int32_t *iscratch = nullptr, *order = nullptr, *prev = nullptr, *next = nullptr, *alnof = nullptr;
  Chain* chains = nullptr;
  ChainFwd* fw = nullptr;
  ChainBwd* bw = nullptr;
  FinishScratch S{};
  uint2* wl_d = nullptr;
};

template <typename T>
int regrow(pg_ctx* ctx, T*& p, size_t n) {
  if (p) PG_HIP(ctx, hipFree(p));
  p = nullptr;
  PG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T)));
  return PG_OK;
}
}  // namespace

static AnimScratch* anim_scratch(pg_ctx* ctx) {
  if (!ctx->anim_scratch) ctx->anim_scratch = new AnimScratch();
  return static_cast<AnimScratch*>(ctx->anim_scratch);
}

void pg_anim_free_scratch(pg_ctx* ctx) {
  AnimScratch* A = static_cast<AnimScratch*>(ctx->anim_scratch);
  if (!A) return;
  void* ptrs[] = {A->table, A->recs_d, A->refs_d, A->units_d, A->mem_count, A->moff, A->nch, A->status, A->out, A->mem, A->cm,
                  A->iscratch, A->order, A->prev, A->next, A->alnof, A->chains, A->fw, A->bw, A->S.alns, A->S.a_rrec,
                  A->S.a_qrec, A->S.idx, A->S.from, A->S.sc, A->wl_d};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete A;
  ctx->anim_scratch = nullptr;
}

// One batch of ordered pairs (ref_ids grouped).  Two seeding passes: the first only counts the maximal matches of every
// (pair, strand) unit, so that every per-match array gets exactly the slice it needs; this is what lets thousands of
// units — whose single-thread cluster kernels are latency-bound — be in flight at once within the HBM budget.
// If the batch needs more than max_matches, only its first n_done pairs are processed (the caller continues from there).
int pg_anim_run_batch(pg_ctx* ctx, const int32_t* ref_ids, const int32_t* qry_ids, uint32_t n_pairs, int filter_1to1,
                      uint64_t max_matches, pg_anim_result* out_host, uint32_t* n_done) {
  AnimScratch* A = anim_scratch(ctx);
  uint32_t n_units = 2 * n_pairs;
  int rc;
  std::vector<int32_t> ref_list;
  std::vector<uint32_t> ref_of_pair(n_pairs);
  for (uint32_t p = 0; p < n_pairs; ++p) {
    if (ref_list.empty() || ref_list.back() != ref_ids[p]) ref_list.push_back(ref_ids[p]);
    ref_of_pair[p] = (uint32_t)ref_list.size() - 1;
  }
  const uint32_t n_refs = (uint32_t)ref_list.size();
  std::vector<RefDesc> refs(n_refs);
  std::vector<int32_t> recs;
  std::vector<uint32_t> ref_rec_off(n_refs), qry_rec_off(n_pairs);
  std::vector<size_t> table_off(n_refs);
  size_t slots = 0;
  int32_t max_rlen = 0, max_qlen = 0;
  for (uint32_t r = 0; r < n_refs; ++r) {
    const PgGenome& G = ctx->genomes[ref_list[r]];
    uint32_t tbits = 10;
    while ((1ull << tbits) < (uint64_t)(G.stream_len + G.stream_len / 2 + 16)) ++tbits;
    refs[r].codes = ctx->d_codes + G.arena_start / 16;
    refs[r].mask = ctx->d_mask + G.arena_start / 32;
    refs[r].len = (int32_t)G.stream_len;
    refs[r].n_rec = (int32_t)G.n_rec;
    refs[r].table_mask = (uint32_t)((1ull << tbits) - 1);
    table_off[r] = slots;
    slots += (size_t)1 << tbits;
    ref_rec_off[r] = (uint32_t)recs.size();
    recs.insert(recs.end(), G.rec_start.begin(), G.rec_start.end());
    if ((int32_t)G.stream_len > max_rlen) max_rlen = (int32_t)G.stream_len;
  }
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const PgGenome& Q = ctx->genomes[qry_ids[p]];
    qry_rec_off[p] = (uint32_t)recs.size();
    recs.insert(recs.end(), Q.rec_start.begin(), Q.rec_start.end());
    if ((int32_t)Q.stream_len > max_qlen) max_qlen = (int32_t)Q.stream_len;
  }
  if (slots > A->table_slots) { if ((rc = regrow(ctx, A->table, slots))) return rc; A->table_slots = slots; }
  if (recs.size() > A->recs) { if ((rc = regrow(ctx, A->recs_d, recs.size()))) return rc; A->recs = recs.size(); }
  if (n_refs > A->refs) { if ((rc = regrow(ctx, A->refs_d, n_refs))) return rc; A->refs = n_refs; }
  if (n_units > A->units) {
    if ((rc = regrow(ctx, A->units_d, n_units))) return rc;
    if ((rc = regrow(ctx, A->mem_count, n_units))) return rc;
    if ((rc = regrow(ctx, A->moff, (size_t)n_units + 1))) return rc;
    if ((rc = regrow(ctx, A->nch, n_units))) return rc;
    A->units = n_units;
  }
  if (n_pairs > A->pairs) {
    if ((rc = regrow(ctx, A->status, n_pairs))) return rc;
    if ((rc = regrow(ctx, A->out, n_pairs))) return rc;
    A->pairs = n_pairs;
  }
  for (uint32_t r = 0; r < n_refs; ++r) { refs[r].rec_start = A->recs_d + ref_rec_off[r]; refs[r].table = A->table + table_off[r]; }
  std::vector<UnitDesc> units(n_units);
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const PgGenome& Q = ctx->genomes[qry_ids[p]];
    for (int s = 0; s < 2; ++s) {
      UnitDesc& U = units[2 * p + s];
      U.codes = ctx->d_codes + Q.arena_start / 16;
      U.mask = ctx->d_mask + Q.arena_start / 32;
      U.len = (int32_t)Q.stream_len;
      U.rec_start = A->recs_d + qry_rec_off[p];
      U.n_rec = (int32_t)Q.n_rec;
      U.strand = s;
      U.pair = (int32_t)p;
      U.ref = (int32_t)ref_of_pair[p];
    }
  }
  PG_HIP(ctx, hipMemcpyAsync(A->recs_d, recs.data(), recs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemcpyAsync(A->refs_d, refs.data(), n_refs * sizeof(RefDesc), hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemcpyAsync(A->units_d, units.data(), n_units * sizeof(UnitDesc), hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(A->table, 0xFF, slots * 8, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(A->mem_count, 0, n_units * 4, ctx->stream));
  hipLaunchKernelGGL(anim_index_kernel, dim3((max_rlen + 255) / 256, n_refs), dim3(256), 0, ctx->stream, A->refs_d);
  // pass 1: count
  hipLaunchKernelGGL(anim_seed_kernel, dim3((max_qlen + 255) / 256, n_units), dim3(256), 0, ctx->stream, A->refs_d, A->units_d,
                     (Match*)nullptr, A->mem_count, (const uint32_t*)nullptr, 1);
  std::vector<uint32_t> cnt(n_units), moff(n_units + 1, 0);
  PG_HIP(ctx, hipMemcpyAsync(cnt.data(), A->mem_count, n_units * 4, hipMemcpyDeviceToHost, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint32_t pairs_fit = 0;
  uint64_t tot = 0;
  for (uint32_t p = 0; p < n_pairs; ++p) {
    const uint64_t need = (uint64_t)cnt[2 * p] + cnt[2 * p + 1] + 2;   // +1 per unit: never a zero-size slice
    if (p > 0 && tot + need > max_matches) break;
    tot += need;
    pairs_fit = p + 1;
  }
  n_pairs = pairs_fit;
  n_units = 2 * n_pairs;
  *n_done = n_pairs;
  for (uint32_t u = 0; u < n_units; ++u) moff[u + 1] = moff[u] + cnt[u] + 1;
  const size_t M = moff[n_units];
  if (M > A->matches) {
    const size_t cap = M + M / 4;
    if ((rc = regrow(ctx, A->mem, cap))) return rc;
    if ((rc = regrow(ctx, A->cm, cap))) return rc;
    if ((rc = regrow(ctx, A->iscratch, cap * 7))) return rc;
    if ((rc = regrow(ctx, A->chains, cap))) return rc;
    if ((rc = regrow(ctx, A->order, cap))) return rc;
    if ((rc = regrow(ctx, A->prev, cap))) return rc;
    if ((rc = regrow(ctx, A->next, cap))) return rc;
    if ((rc = regrow(ctx, A->alnof, cap))) return rc;
    if ((rc = regrow(ctx, A->fw, cap))) return rc;
    if ((rc = regrow(ctx, A->bw, cap))) return rc;
    if ((rc = regrow(ctx, A->S.alns, cap))) return rc;
    if ((rc = regrow(ctx, A->S.a_rrec, cap))) return rc;
    if ((rc = regrow(ctx, A->S.a_qrec, cap))) return rc;
    if ((rc = regrow(ctx, A->S.idx, cap))) return rc;
    if ((rc = regrow(ctx, A->S.from, cap))) return rc;
    if ((rc = regrow(ctx, A->S.sc, cap))) return rc;
    A->matches = cap;
  }
  A->S.aln_of = A->alnof;
  PG_HIP(ctx, hipMemcpyAsync(A->moff, moff.data(), ((size_t)n_units + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(A->mem_count, 0, n_units * 4, ctx->stream));
  PG_HIP(ctx, hipMemsetAsync(A->status, 0, n_pairs * 4, ctx->stream));
  ClusterOut O{A->moff, A->cm, A->chains, A->nch, A->order, A->prev, A->next, A->status};
  // pass 2: write the matches
  hipLaunchKernelGGL(anim_seed_kernel, dim3((max_qlen + 255) / 256, n_units), dim3(256), 0, ctx->stream, A->refs_d, A->units_d,
                     A->mem, A->mem_count, A->moff, 0);
  if (getenv("PYANI_ANIM_SCALAR_CLUSTER"))   // debugging aid: the one-thread-per-unit statement of the same algorithm
    hipLaunchKernelGGL(anim_cluster_kernel, dim3((n_units + 63) / 64), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, n_units,
                       A->mem, A->mem_count, A->iscratch, O);
  else
    hipLaunchKernelGGL(anim_cluster_wave_kernel, dim3(n_units), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, A->mem,
                       A->mem_count, A->iscratch, O);
  // work list of (unit, chain): one wave each
  std::vector<int32_t> nch(n_units);
  PG_HIP(ctx, hipMemcpyAsync(nch.data(), A->nch, n_units * 4, hipMemcpyDeviceToHost, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<uint2> wl;
  for (uint32_t u = 0; u < n_units; ++u)
    for (int32_t c = 0; c < nch[u]; ++c) wl.push_back(make_uint2(u, (uint32_t)c));
  if (!wl.empty()) {
    if (wl.size() > A->wl) { if ((rc = regrow(ctx, A->wl_d, wl.size() + wl.size() / 2))) return rc; A->wl = wl.size() + wl.size() / 2; }
    PG_HIP(ctx, hipMemcpyAsync(A->wl_d, wl.data(), wl.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    for (int phase = 0; phase < 2; ++phase)
      hipLaunchKernelGGL(anim_extend_kernel, dim3((uint32_t)wl.size()), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, O,
                         A->wl_d, A->fw, A->bw, phase);
  }
  hipLaunchKernelGGL(anim_finish_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, ctx->stream, A->refs_d, A->units_d, n_pairs,
                     O, A->fw, A->bw, A->S, filter_1to1, A->out);
  PG_HIP(ctx, hipGetLastError());
  PG_HIP(ctx, hipMemcpyAsync(out_host, A->out, n_pairs * sizeof(pg_anim_result), hipMemcpyDeviceToHost, ctx->stream));
  PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PG_OK;
}

int pg_anim_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const int32_t* rseq, const int32_t* qseq,
                       const int32_t* rs, const int32_t* re, const int32_t* qs, const int32_t* qe, const int32_t* errors,
                       int apply_filter, pg_anim_result* out) {
  const uint64_t n = offsets[n_pairs];
  std::vector<Aln> h(n);
  for (uint64_t i = 0; i < n; ++i) {
    Aln a;
    a.strand = qs[i] > qe[i];
    a.rs = (rs[i] < re[i] ? rs[i] : re[i]) - 1; a.re = rs[i] < re[i] ? re[i] : rs[i];
    a.qs = (qs[i] < qe[i] ? qs[i] : qe[i]) - 1; a.qe = qs[i] < qe[i] ? qe[i] : qs[i];
    a.errors = errors[i];
    a.keep = apply_filter ? 0 : 3;
    h[i] = a;
  }
  uint64_t* d_off = nullptr; Aln* d_a = nullptr; int32_t *d_rg = nullptr, *d_qg = nullptr, *d_idx = nullptr, *d_from = nullptr;
  double* d_sc = nullptr; pg_anim_result* d_out = nullptr;
  std::vector<void*> to_free;
  auto cleanup = [&]() { for (void* p : to_free) if (p) (void)hipFree(p); };
  int rc;
#define AA(ptr, cnt) do { if ((rc = anim_alloc(ctx, ptr, (cnt)))) { cleanup(); return rc; } to_free.push_back(ptr); } while (0)
  AA(d_off, n_pairs + 1); AA(d_a, n + 1); AA(d_rg, n + 1); AA(d_qg, n + 1); AA(d_idx, n + 1); AA(d_from, n + 1); AA(d_sc, n + 1);
  AA(d_out, n_pairs + 1);
#undef AA
  hipError_t e = hipMemcpyAsync(d_off, offsets, (n_pairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_a, h.data(), n * sizeof(Aln), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_rg, rseq, n * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_qg, qseq, n * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(anim_reduce_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, ctx->stream, n_pairs, d_off, d_a, d_rg, d_qg,
                       d_idx, d_from, d_sc, apply_filter, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n_pairs * sizeof(pg_anim_result), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  cleanup();
  if (e != hipSuccess) return pg_fail(ctx, PG_E_HIP, std::string("anim reduce: ") + hipGetErrorString(e));
  return PG_OK;
}

// ---- ANIb: parse_blast_tab reduction (pyani/anib.py:641-665), one thread per ordered pair ---------------------------
namespace {
__global__ __launch_bounds__(64) void anib_reduce_kernel(uint32_t n_pairs, const uint64_t* __restrict__ offsets,
                                                         const uint64_t* __restrict__ foff, const int32_t* __restrict__ frag,
                                                         const int32_t* __restrict__ length, const int32_t* __restrict__ mismatch,
                                                         const int32_t* __restrict__ gaps, const int32_t* __restrict__ qlen,
                                                         const double* __restrict__ pident, int64_t* first_row,
                                                         int64_t* __restrict__ aln_out, int64_t* __restrict__ err_out,
                                                         double* __restrict__ pid_out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  int64_t* first = first_row + foff[p];
  const uint64_t nf = foff[p + 1] - foff[p];
  for (uint64_t f = 0; f < nf; ++f) first[f] = -1;
  for (uint64_t i = offsets[p]; i < offsets[p + 1]; ++i) {
    const int32_t alnlen = length[i] - gaps[i], alnids = alnlen - mismatch[i];
    const double cov = (double)alnlen / (double)qlen[i], pid = (double)alnids / (double)qlen[i];
    if (cov > 0.7 && pid > 0.3 && (uint64_t)frag[i] < nf && first[frag[i]] < 0) first[frag[i]] = (int64_t)i;
  }
  int64_t aln = 0, err = 0, cnt = 0;
  double sum = 0.0;
  for (uint64_t f = 0; f < nf; ++f) {
    const int64_t i = first[f];
    if (i < 0) continue;
    aln += length[i] - gaps[i];
    err += (int64_t)mismatch[i] + gaps[i];
    sum = sum + pident[i];
    ++cnt;
  }
  aln_out[p] = aln;
  err_out[p] = err;
  pid_out[p] = cnt ? sum / (double)cnt : 0.0;
}
}  // namespace

int pg_anib_reduce_run(pg_ctx* ctx, uint32_t n_pairs, const uint64_t* offsets, const uint32_t* n_frags, const int32_t* frag,
                       const int32_t* length, const int32_t* mismatch, const int32_t* gaps, const int32_t* qlen,
                       const double* pident, int64_t* aln_out, int64_t* err_out, double* pid_out) {
  const uint64_t n = offsets[n_pairs];
  std::vector<uint64_t> foff(n_pairs + 1, 0);
  for (uint32_t p = 0; p < n_pairs; ++p) foff[p + 1] = foff[p] + n_frags[p];
  std::vector<void*> to_free;
  auto cleanup = [&]() { for (void* q : to_free) if (q) (void)hipFree(q); };
  uint64_t *d_off = nullptr, *d_foff = nullptr;
  int32_t *d_frag = nullptr, *d_len = nullptr, *d_mm = nullptr, *d_gap = nullptr, *d_ql = nullptr;
  double *d_pid = nullptr, *d_pout = nullptr;
  int64_t *d_first = nullptr, *d_aln = nullptr, *d_err = nullptr;
  int rc;
#define AA(ptr, cnt) do { if ((rc = anim_alloc(ctx, ptr, (cnt)))) { cleanup(); return rc; } to_free.push_back(ptr); } while (0)
  AA(d_off, n_pairs + 1); AA(d_foff, n_pairs + 1); AA(d_frag, n + 1); AA(d_len, n + 1); AA(d_mm, n + 1); AA(d_gap, n + 1);
  AA(d_ql, n + 1); AA(d_pid, n + 1); AA(d_first, foff[n_pairs] + 1); AA(d_aln, n_pairs); AA(d_err, n_pairs); AA(d_pout, n_pairs);
#undef AA
  hipError_t e = hipMemcpyAsync(d_off, offsets, (n_pairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_foff, foff.data(), (n_pairs + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
  const struct { void* d; const void* h; size_t b; } cp[] = {{d_frag, frag, n * 4}, {d_len, length, n * 4}, {d_mm, mismatch, n * 4},
                                                           {d_gap, gaps, n * 4}, {d_ql, qlen, n * 4}, {d_pid, pident, n * 8}};
  for (const auto& c : cp)
    if (e == hipSuccess && c.b) e = hipMemcpyAsync(c.d, c.h, c.b, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(anib_reduce_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, ctx->stream, n_pairs, d_off, d_foff, d_frag, d_len,
                       d_mm, d_gap, d_ql, d_pid, d_first, d_aln, d_err, d_pout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(aln_out, d_aln, n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(err_out, d_err, n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(pid_out, d_pout, n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  cleanup();
  if (e != hipSuccess) return pg_fail(ctx, PG_E_HIP, std::string("anib reduce: ") + hipGetErrorString(e));
  return PG_OK;
}
