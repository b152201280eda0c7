// pg_tetra_count.h — K0, the HBM-streaming k-mer histogram kernel (device code; included inside an anonymous
// namespace by pg_tetra.hip and by tools/microbench/count_bench.hip, which times its variants).
//
// What bounds it (measured on MI355X, profiles/archive/r01_*): the stream itself runs at the HBM copy peak (~6.1 TB/s) when the
// LDS atomics are removed, and the LDS atomic pipe sustains ~32 ds_add_u32 lanes/ns/CU conflict-free (~20 with
// random banks).  One atomic per base would be 5x too slow, so the kernel counts HEPTAmers at stride 4 — one
// atomic per 4 bases — into a 16384-bin u32 histogram (64 KiB of LDS) and folds each heptamer bin into the four
// tetramers it contains when a block leaves a genome (a cascade of marginal sums, ~30 LDS ops per thread).
//
// Other design points:
//   * di-/tri-nucleotide counts are NOT histogrammed: they are marginals of the tetramer counts plus the rare
//     windows that end at a dirty base / record end (E2/E3), produced by a slow path that only waves seeing a
//     dirty base take.  The reverse strand is never scanned: c_k[x] = F_k[x] + F_k[rc(x)] (K1).
//   * each lane streams 64 bases per tile: one 16 B code load + one 8 B mask load, coalesced (1 KiB + 512 B per
//     wave); the 3-base look-ahead comes from the neighbour lane through DPP wave_shl:1, and for lane 63 from a
//     wave-uniform dword load.  No branches around loads, no scalar loads in the hot loop: the compiler can count
//     vmcnt exactly, so a PF-deep register prefetch ring really keeps PF tiles in flight.
//   * blocks own contiguous ranges of super-tiles; tile indices inside a genome segment are arithmetic.
//
// Template knobs (the product instantiates ONE configuration; the microbenchmark times the others):
//   PF    register prefetch depth in super-tiles
//   MODE  0 = full kernel, 1 = loads only (no LDS atomics), 2 = atomics only (no global loads in the loop)
//   BLOCK workgroup size (1024 or 512); two workgroups share a CU (73 KiB of LDS each), so one group's flush /
//         prologue overlaps the other's streaming
#pragma once

constexpr int K0_H7_BINS = 16384;
constexpr uint32_t K0_FORCE_FLUSH_TILES = 16384;  // <= 2^30 bases: keeps every u32 bin far from overflow

// LDS layout in words: H7[16384] | S_hi2[1024] | S_lo2[1024] | F4[256] | E3[64] | E2[16] | DUMMY[64]   = 73.6 KiB: two workgroups per CU
struct K0Lds {
  static constexpr int H7 = 0;
  static constexpr int S_HI2 = H7 + K0_H7_BINS;
  static constexpr int S_LO2 = S_HI2 + 1024;
  static constexpr int F4 = S_LO2 + 1024;
  static constexpr int E3 = F4 + 256;
  static constexpr int E2 = E3 + 64;
  static constexpr int DUMMY = E2 + 16;    // 64 words, one per lane (its own bank): where the atomic of a dirty heptamer lands
  static constexpr int WORDS = DUMMY + 64;
  static_assert(2 * WORDS * 4 <= 160 * 1024, "two workgroups must fit the 160 KiB of one gfx950 CU");
};

__device__ __forceinline__ uint32_t nat4_from_lowfirst(uint32_t r) {
  // r = b0 | b1<<2 | b2<<4 | b3<<6 (first base in the low bits)  ->  b0<<6 | b1<<4 | b2<<2 | b3
  return ((r & 3u) << 6) | ((r & 0xCu) << 2) | ((r & 0x30u) >> 2) | ((r & 0xC0u) >> 6);
}

__device__ __forceinline__ void lds_inc(uint32_t* p) {
  __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct LaneData {
  uint4 c;          // 64 bases of codes
  uint2 m;          // 64 mask bits
  uint32_t nc, nm;  // first code / mask word after the WAVE's span (same value in all lanes)
};

typedef uint32_t k0_v4u __attribute__((ext_vector_type(4)));
typedef uint32_t k0_v2u __attribute__((ext_vector_type(2)));

// Loads of one lane for the tile whose first 64-base span index is span0 (= tile * 1024).  Branch-free.
__device__ __forceinline__ LaneData k0_load(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask,
                                            uint64_t span0, uint32_t tid) {
  LaneData d;
  const uint64_t lane_span = span0 + tid;
  const k0_v4u cv = __builtin_nontemporal_load(reinterpret_cast<const k0_v4u*>(codes + lane_span * 4));  // streamed once
  const k0_v2u mv = __builtin_nontemporal_load(reinterpret_cast<const k0_v2u*>(mask + lane_span * 2));
  d.c = make_uint4(cv.x, cv.y, cv.z, cv.w);
  d.m = make_uint2(mv.x, mv.y);
  const uint64_t next_wave_span = (lane_span | 63u) + 1u;  // same address in all 64 lanes: one broadcast fetch
  d.nc = codes[next_wave_span * 4];
  d.nm = mask[next_wave_span * 2];
  return d;
}

// One heptamer whose 7 mask bits are not all clean: its four window starts q = 0..3 are classified one by one.
// m7: mask bits of bases p..p+6; h: the 14-bit heptamer code (dirty bases are 0).  Rare (record ends, N runs).
__device__ __forceinline__ void k0_dirty_heptamer(uint32_t m7, uint32_t h, uint32_t* F4, uint32_t* E3, uint32_t* E2) {
  const uint32_t v2 = m7 & (m7 >> 1), v3 = v2 & (m7 >> 2), v4 = v3 & (m7 >> 3);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t r = h >> (2 * q);
    if ((v4 >> q) & 1u) lds_inc(&F4[nat4_from_lowfirst(r & 0xFFu)]);
    else if ((v3 >> q) & 1u) lds_inc(&E3[((r & 3u) << 4) | (r & 0xCu) | ((r >> 4) & 3u)]);   // trimer, first base MSB
    else if ((v2 >> q) & 1u) lds_inc(&E2[((r & 3u) << 2) | ((r >> 2) & 3u)]);                 // dinucleotide
  }
}

// 16 heptamers at offsets 0,4,..,60 of the lane's 64 bases (+3 look-ahead), one LDS atomic each.
// CHECK = false: the whole wave is known clean (no mask work at all).  CHECK = true (the wave saw a dirty base): still sixteen
// unconditional atomics per lane, no branch in the hot part — whether a heptamer is clean comes from three shift-and-AND steps
// per mask word (c7 bit p = mask bits p..p+6 all set) and a dirty heptamer's atomic is redirected to the lane's own dummy word;
// the dirty heptamers that still hold clean window starts are then classified one by one in a loop only the lanes that have
// some enter (r01: every heptamer of such a wave tested its mask bits and branched, 3.7 x the clean cost at 3000 N runs per
// genome — scaffolded drafts look like that).
template <bool CHECK>
__device__ __forceinline__ void k0_count_lane(const LaneData& d, uint32_t* lds_h7, uint32_t* F4, uint32_t* E3, uint32_t* E2,
                                              uint32_t dummy_byte_off = 0) {
  const uint32_t w[5] = {d.c.x, d.c.y, d.c.z, d.c.w, d.nc};
  // mask bits starting at base 0, 16, 32, 48 (each word holds >= 19 valid bits from there)
  const uint32_t mw[4] = {d.m.x, __builtin_amdgcn_alignbit(d.m.y, d.m.x, 16), d.m.y, __builtin_amdgcn_alignbit(d.nm, d.m.y, 16)};
  char* h7b = reinterpret_cast<char*>(lds_h7);
  uint32_t need = 0, need_lo = 0;   // CHECK: dirty heptamers that still hold clean window starts (bit 4t + 16 (j & 1), words (0,1) | (2,3))
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t lo = w[j], hi = w[j + 1];
    // byte offsets (heptamer << 2) of the four heptamers that start in this code word
    const uint32_t off[4] = {lo << 2, lo >> 6, lo >> 14, __builtin_amdgcn_alignbit(hi, lo, 22)};
    uint32_t c7 = 0;
    if (CHECK) {
      const uint32_t c2 = mw[j] & (mw[j] >> 1), c4 = c2 & (c2 >> 2);
      c7 = c4 & (c4 >> 3);
      // heptamers (stride 4: bits 0, 4, 8, 12) that are dirty but whose first four bases hold a clean one
      const uint32_t a2 = mw[j] | (mw[j] >> 1), a4 = a2 | (a2 >> 2);
      need |= (a4 & ~c7 & 0x1111u) << (16 * (j & 1));   // words 0 / 1 -> low / high half of a 32-bit word ...
      if (j == 1) { need_lo = need; need = 0; }                    // ... two such words: (need_lo, need)
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t o = off[t] & 0xFFFCu;
      if (!CHECK) {
        lds_inc(reinterpret_cast<uint32_t*>(h7b + o));
      } else {
        const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe((int)c7, 4 * t, 1);   // 0 or ~0: one bit-field extract + one v_bfi
        lds_inc(reinterpret_cast<uint32_t*>(h7b + ((o & keep) | (dummy_byte_off & ~keep))));
      }
    }
  }
  if (CHECK) {
    // Rare: heptamers that touch the edge of a dirty run.  Not a loop per lane (r02 first form: every lane with such a heptamer
    // walked its own bits, the whole wave waiting through ~300 divergent instructions per dirty base) but wave-wide: the words
    // of a lane that has some are broadcast, and its <= 16 heptamers x 4 window starts are 64 work items, one per lane.
    unsigned long long lanes_todo = __ballot((need | need_lo) != 0u);
    const int lane = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    while (lanes_todo) {   // uniform
      const int src = __ffsll(lanes_todo) - 1;
      lanes_todo &= lanes_todo - 1;
      const uint32_t W0 = __builtin_amdgcn_readlane(w[0], src), W1 = __builtin_amdgcn_readlane(w[1], src),
                     W2 = __builtin_amdgcn_readlane(w[2], src), W3 = __builtin_amdgcn_readlane(w[3], src),
                     W4 = __builtin_amdgcn_readlane(w[4], src);
      const uint32_t M0 = __builtin_amdgcn_readlane(mw[0], src), M1 = __builtin_amdgcn_readlane(mw[1], src),
                     M2 = __builtin_amdgcn_readlane(mw[2], src), M3 = __builtin_amdgcn_readlane(mw[3], src);
      unsigned long long T = ((unsigned long long)__builtin_amdgcn_readlane(need, src) << 32) | __builtin_amdgcn_readlane(need_lo, src);
      int myb = -1;          // bit 16 j + 4 t of the heptamer this lane works on: the (lane >> 2)-th set bit
      for (int k = 0; T; T &= T - 1, ++k) {   // uniform
        const int b = __ffsll(T) - 1;
        myb = (lane >> 2) == k ? b : myb;
      }
      if (myb >= 0) {
        const int j = myb >> 4, tt = (myb >> 2) & 3, q = lane & 3;
        const uint32_t lo = j == 0 ? W0 : j == 1 ? W1 : j == 2 ? W2 : W3;
        const uint32_t hi = j == 0 ? W1 : j == 1 ? W2 : j == 2 ? W3 : W4;
        const uint32_t mj = j == 0 ? M0 : j == 1 ? M1 : j == 2 ? M2 : M3;
        const uint32_t h = (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * tt)) & 0x3FFFu;
        const uint32_t m7 = (mj >> (4 * tt)) & 0x7Fu;
        const uint32_t v2 = m7 & (m7 >> 1), v3 = v2 & (m7 >> 2), v4 = v3 & (m7 >> 3);
        const uint32_t r = h >> (2 * q);
        // (k0_dirty_heptamer's classification of window start q)
        if ((v4 >> q) & 1u) lds_inc(&F4[nat4_from_lowfirst(r & 0xFFu)]);
        else if ((v3 >> q) & 1u) lds_inc(&E3[((r & 3u) << 4) | (r & 0xCu) | ((r >> 4) & 3u)]);
        else if ((v2 >> q) & 1u) lds_inc(&E2[((r & 3u) << 2) | ((r >> 2) & 3u)]);
      }
    }
  }
}

template <int MODE>
__device__ __forceinline__ void k0_process(LaneData d, uint32_t* lds, uint32_t tid, uint32_t& sink) {
  // look-ahead from the next lane (lane 63 keeps the wave-uniform word it loaded)
  d.nc = (uint32_t)__builtin_amdgcn_update_dpp((int)d.nc, (int)d.c.x, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
  d.nm = (uint32_t)__builtin_amdgcn_update_dpp((int)d.nm, (int)d.m.x, 0x130, 0xf, 0xf, false);
  if (MODE == 1) {  // loads only: consume the data so the loads are not dead
    sink ^= d.c.x ^ d.c.y ^ d.c.z ^ d.c.w ^ d.m.x ^ d.m.y ^ d.nc ^ d.nm;
    return;
  }
  using L = K0Lds;
  const bool all_clean = (d.m.x & d.m.y) == 0xFFFFFFFFu && (d.nm & 7u) == 7u;
  const bool none_clean = (d.m.x | d.m.y) == 0u;
  uint32_t* h7 = lds + L::H7;
  if (__all(all_clean)) {
    k0_count_lane<false>(d, h7, nullptr, nullptr, nullptr);
  } else if (!__all(none_clean)) {
    // every lane takes part (the rare path hands work items to ALL 64 lanes); a lane without a clean base sends its sixteen
    // atomics to its dummy word and asks for nothing
    k0_count_lane<true>(d, h7, lds + L::F4, lds + L::E3, lds + L::E2, (uint32_t)(L::DUMMY + (tid & 63u)) * 4u);
  }
}

// Fold the heptamer histogram into the tetramers at its four offsets and push the block's partial counts to the
// genome's accumulator.  Low-first encoding: heptamer f = b0 | b1<<2 | ... | b6<<12.
//   S_hi2[b0..b4] = sum_{b5,b6} H7 (16 strided reads)     T0[b0..b3] = sum_b4 S_hi2   T1[b1..b4] = sum_b0 S_hi2
//   S_lo2[b2..b6] = sum_{b0,b1} H7 (16 contiguous words)  T2[b2..b5] = sum_b6 S_lo2   T3[b3..b6] = sum_b2 S_lo2
template <int BLOCK>
__device__ __forceinline__ void k0_flush(uint32_t* lds, unsigned long long* __restrict__ acc_g, uint32_t tid) {
  using L = K0Lds;
  uint32_t* H7 = lds + L::H7;
  uint32_t* S_hi2 = lds + L::S_HI2;
  uint32_t* S_lo2 = lds + L::S_LO2;
  uint32_t* F4 = lds + L::F4;
  uint32_t* E3 = lds + L::E3;
  uint32_t* E2 = lds + L::E2;
  for (uint32_t o = tid; o < 1024; o += BLOCK) {
    uint32_t hi = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) hi += H7[o + 1024 * t];
    uint32_t lo = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint4 q = *reinterpret_cast<const uint4*>(H7 + 16 * o + 4 * t);
      lo += q.x + q.y + q.z + q.w;
    }
    S_hi2[o] = hi;
    S_lo2[o] = lo;
  }
  __syncthreads();
  {  // zero H7 for the next genome (all reads of H7 are done)
    uint4* z = reinterpret_cast<uint4*>(H7);
    for (uint32_t i = tid; i < K0_H7_BINS / 4; i += BLOCK) z[i] = make_uint4(0, 0, 0, 0);
  }
  if (tid < 256) {
    const uint32_t t0 = S_hi2[tid] + S_hi2[tid + 256] + S_hi2[tid + 512] + S_hi2[tid + 768];
    const uint4 q1 = *reinterpret_cast<const uint4*>(S_hi2 + 4 * tid);
    const uint32_t t2 = S_lo2[tid] + S_lo2[tid + 256] + S_lo2[tid + 512] + S_lo2[tid + 768];
    const uint4 q3 = *reinterpret_cast<const uint4*>(S_lo2 + 4 * tid);
    const uint32_t tot = t0 + (q1.x + q1.y + q1.z + q1.w) + t2 + (q3.x + q3.y + q3.z + q3.w);
    if (tot) atomicAdd(&F4[nat4_from_lowfirst(tid)], tot);  // the dirty path may have counted into F4 as well
  }
  __syncthreads();
  if (tid < PG_ACC_WORDS) {
    // acc layout: E2[16] | E3[64] | F4[256]
    uint32_t* src = tid < 16 ? &E2[tid] : tid < 80 ? &E3[tid - 16] : &F4[tid - 80];
    const uint32_t v = *src;
    if (v) atomicAdd(&acc_g[tid], (unsigned long long)v);
    *src = 0;
  }
  __syncthreads();
}

// One contiguous run of `n` tiles (BLOCK x 64 bases each) of ONE genome starting at arena tile `tile0`.
template <int PF, int MODE, int BLOCK>
__device__ __forceinline__ void k0_segment(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask,
                                           uint32_t tile0, uint32_t n, uint32_t* lds, uint32_t tid, uint32_t& sink) {
  const uint32_t last = tile0 + n - 1;
  LaneData ring[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) ring[s] = k0_load(codes, mask, (uint64_t)min(tile0 + s, last) * BLOCK, tid);
  for (uint32_t i = 0; i < n; i += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      if (i + s < n) {  // uniform
        const LaneData d = ring[s];
        // refill this slot PF tiles ahead; past the end it re-reads the last tile (in bounds, result unused)
        if (MODE != 2) ring[s] = k0_load(codes, mask, (uint64_t)min(tile0 + i + s + PF, last) * BLOCK, tid);
        k0_process<MODE>(d, lds, tid, sink);
      }
    }
  }
}

// seg_prefix[0..n_batch]: cumulative SUPER-tile counts (65536 bases) of the batch genomes; seg_tile0[b]: first arena
// super-tile of genome b.  A workgroup of BLOCK threads works in tiles of BLOCK x 64 bases (1024 / BLOCK per super-tile).
template <int PF, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void tetra_count_kernel(const uint32_t* __restrict__ codes,
                                                            const uint32_t* __restrict__ mask,
                                                            const uint32_t* __restrict__ seg_tile0,
                                                            const uint32_t* __restrict__ seg_prefix, uint32_t n_batch,
                                                            unsigned long long* __restrict__ acc) {
  // static (not `extern`): the histogram's LDS offsets become instruction immediates; with a dynamic allocation the
  // compiler emits one `v_add 0, idx` per atomic for the unknown base (97 instructions of the hot loop)
  __shared__ __attribute__((aligned(16))) uint32_t lds[K0Lds::WORDS];
  using L = K0Lds;
  constexpr uint32_t TPS = 1024 / BLOCK;   // tiles per super-tile
  const uint32_t tid = threadIdx.x;
  const uint32_t n_work = seg_prefix[n_batch] * TPS;
  const uint32_t w0 = (uint32_t)(((uint64_t)blockIdx.x * n_work) / gridDim.x);
  const uint32_t w1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * n_work) / gridDim.x);
  if (w0 >= w1) return;
  {
    uint4* z = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = tid; i < L::WORDS / 4; i += BLOCK) z[i] = make_uint4(0, 0, 0, 0);
  }
  // genome containing w0: largest b with seg_prefix[b] * TPS <= w0 (uniform binary search)
  uint32_t lo = 0, hi = n_batch;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (seg_prefix[mid] * TPS <= w0) lo = mid; else hi = mid;
  }
  __syncthreads();
  uint32_t sink = 0;
  uint32_t b = lo, w = w0;
  while (w < w1) {
    const uint32_t p0 = seg_prefix[b] * TPS, p1 = seg_prefix[b + 1] * TPS;
    if (p1 > w) {
      const uint32_t end = min(w1, p1);
      uint32_t tile = seg_tile0[b] * TPS + (w - p0);
      while (w < end) {  // chunked only to bound the u32 bins
        const uint32_t n = min(end - w, K0_FORCE_FLUSH_TILES);
        k0_segment<PF, MODE, BLOCK>(codes, mask, tile, n, lds, tid, sink);
        __syncthreads();
        k0_flush<BLOCK>(lds, acc + (size_t)b * PG_ACC_WORDS, tid);
        w += n;
        tile += n;
      }
    }
    ++b;
  }
  if (MODE == 1 && sink == 0x12345678u) acc[0] = sink;  // keeps the loads alive, practically never true
}
