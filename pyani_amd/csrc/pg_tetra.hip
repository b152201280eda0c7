// pg_tetra.hip — hand-written gfx950 (CDNA4, wave64) kernels for pyani's TETRA path.
//
//   K0 tetra_count     HBM-streaming k-mer histogram over the 2-bit/1-bit packed genome arena   (HBM roofline)
//   K1 tetra_finalize  marginals + reverse-complement fold + quirk -> c2/c3/c4 -> 256 Z-scores   (tiny)
//   K2 tetra_stats     per-genome mean / deviations / sum of squares, sequential order           (tiny)
//   K3 tetra_pairs     all-vs-all Pearson, sequential 256-term dot products in fp64              (latency bound)
//
// Reference semantics: pyani/tetra.py:98-138 (counts, Z) and :158-194 (Pearson); closed form in SURVEY.md App. A.
// Built with -ffp-contract=off: the fp64 operation ORDER is part of the contract (bit-exact results).
//
// K0 design (integer/byte work — no MFMA on purpose):
//   * one 1024-thread workgroup per CU (LDS-limited), each lane streams 64 bases per iteration with one
//     16-byte code load + one 8-byte mask load (fully coalesced: a wave reads 1 KiB + 512 B contiguous);
//   * LDS ds_add_u32 is the scarce resource (~16 lanes/clk/CU), so the kernel counts PENTAmers at stride 2
//     (one atomic per 2 bases) into a 1024-bin histogram that is replicated 32x so that lane l always hits
//     bank l%32: conflict-free by construction.  At flush each 5-mer bin is folded into its two tetramers.
//   * di-/tri-nucleotide counts are NOT histogrammed: they are marginals of the tetramer counts plus the rare
//     windows that end at a dirty base / record end (E2/E3), handled on a slow path taken only by waves that
//     see a dirty base.  The reverse strand is never scanned: c_k[x] = F_k[x] + F_k[rc(x)].
//   * neighbouring lanes exchange their 3-base look-ahead with one DPP wave_shl:1 (no LDS traffic).
#include "pg_internal.h"

namespace {

constexpr int K0_BLOCK = 1024;
constexpr int K0_REPL = 32;
constexpr int K0_H5_WORDS = 1024 * K0_REPL;              // 128 KiB
constexpr int K0_LDS_WORDS = K0_H5_WORDS + 256 + 64 + 16;  // + F4 | E3 | E2
constexpr uint32_t K0_FORCE_FLUSH_TILES = 16384;         // 2^30 bases: keeps every u32 bin far from overflow

__device__ __forceinline__ uint32_t nat4_from_lowfirst(uint32_t r) {
  // r = b0 | b1<<2 | b2<<4 | b3<<6 (first base in the low bits)  ->  b0<<6 | b1<<4 | b2<<2 | b3
  return ((r & 3u) << 6) | ((r & 0xCu) << 2) | ((r & 0x30u) >> 2) | ((r & 0xC0u) >> 6);
}

__device__ __forceinline__ void lds_inc(uint32_t* p) {
  __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct LaneData {
  uint4 c;   // 64 bases of codes
  uint2 m;   // 64 mask bits
  uint32_t nc, nm;  // first code / mask word of the NEXT lane's span (valid in lane 63 only before the DPP)
};

__device__ __forceinline__ LaneData k0_load(const uint32_t* __restrict__ codes, const uint32_t* __restrict__ mask,
                                            uint32_t tile, uint32_t tid) {
  LaneData d;
  const uint64_t lane_span = (uint64_t)tile * K0_BLOCK + tid;  // index of this lane's 64-base span in the arena
  d.c = *reinterpret_cast<const uint4*>(codes + lane_span * 4);
  d.m = *reinterpret_cast<const uint2*>(mask + lane_span * 2);
  d.nc = 0;
  d.nm = 0;
  if ((tid & 63u) == 63u) {  // the wave's last lane looks into the next wave's / next tile's first words
    d.nc = codes[(lane_span + 1) * 4];
    d.nm = mask[(lane_span + 1) * 2];
  }
  return d;
}

// generic path: every window checked against the mask (taken by a wave only if one of its lanes sees a dirty base)
__device__ __noinline__ void k0_slow_lane(const LaneData& d, uint32_t* F4, uint32_t* E3, uint32_t* E2) {
  const uint64_t M = (uint64_t)d.m.x | ((uint64_t)d.m.y << 32);
  const uint32_t w[5] = {d.c.x, d.c.y, d.c.z, d.c.w, d.nc};
  auto clean = [&](int p) -> bool { return p < 64 ? ((M >> p) & 1ull) != 0 : ((d.nm >> (p - 64)) & 1u) != 0; };
  auto base = [&](int p) -> uint32_t { return (w[p >> 4] >> (2 * (p & 15))) & 3u; };
  for (int p = 0; p < 64; ++p) {
    if (!clean(p) || !clean(p + 1)) continue;
    const uint32_t di = base(p) * 4 + base(p + 1);
    if (!clean(p + 2)) { lds_inc(&E2[di]); continue; }
    const uint32_t tri = di * 4 + base(p + 2);
    if (!clean(p + 3)) { lds_inc(&E3[tri]); continue; }
    lds_inc(&F4[tri * 4 + base(p + 3)]);
  }
}

// fast path: all 64 + 3 look-ahead bases clean -> 32 pentamers at even offsets, one conflict-free LDS atomic each
__device__ __forceinline__ void k0_fast_lane(const LaneData& d, uint32_t* h5_lane) {
  const uint32_t w[5] = {d.c.x, d.c.y, d.c.z, d.c.w, d.nc};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t lo = w[j], hi = w[j + 1];
#pragma unroll
    for (int t = 0; t < 6; ++t) lds_inc(h5_lane + (((lo >> (4 * t)) & 0x3FFu) << 5));
    const uint32_t x6 = __builtin_amdgcn_alignbit(hi, lo, 24);  // bits 24.. of hi:lo
    lds_inc(h5_lane + ((x6 & 0x3FFu) << 5));
    lds_inc(h5_lane + (((x6 >> 4) & 0x3FFu) << 5));
  }
}

__device__ __forceinline__ void k0_process(LaneData d, uint32_t* lds, uint32_t tid) {
  // look-ahead from the next lane (lane 63 keeps what it loaded itself)
  d.nc = (uint32_t)__builtin_amdgcn_update_dpp((int)d.nc, (int)d.c.x, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
  d.nm = (uint32_t)__builtin_amdgcn_update_dpp((int)d.nm, (int)d.m.x, 0x130, 0xf, 0xf, false);
  const bool all_clean = (d.m.x & d.m.y) == 0xFFFFFFFFu && (d.nm & 7u) == 7u;
  const bool none_clean = (d.m.x | d.m.y) == 0u;
  if (__all(all_clean)) {
    k0_fast_lane(d, lds + (tid & 31u));
  } else if (!__all(none_clean)) {
    if (all_clean) k0_fast_lane(d, lds + (tid & 31u));
    else if (!none_clean) k0_slow_lane(d, lds + K0_H5_WORDS, lds + K0_H5_WORDS + 256, lds + K0_H5_WORDS + 320);
  }
}

// fold the pentamer histogram into tetramers and push the block's partial counts to the genome's accumulator
__device__ void k0_flush(uint32_t* lds, unsigned long long* __restrict__ acc_g, uint32_t tid) {
  uint32_t* F4 = lds + K0_H5_WORDS;
  uint32_t* E3 = F4 + 256;
  uint32_t* E2 = E3 + 64;
  uint32_t sum = 0;
  uint32_t* bin = lds + tid * K0_REPL;
#pragma unroll 8
  for (int r = 0; r < K0_REPL; ++r) {
    const uint32_t rr = (r + tid) & (K0_REPL - 1);  // rotate: consecutive lanes read consecutive banks
    sum += bin[rr];
    bin[rr] = 0;
  }
  if (sum) {
    atomicAdd(&F4[nat4_from_lowfirst(tid & 255u)], sum);  // tetramer at the even position
    atomicAdd(&F4[nat4_from_lowfirst(tid >> 2)], sum);    // tetramer at the following odd position
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

__global__ __launch_bounds__(K0_BLOCK) void tetra_count_kernel(const uint32_t* __restrict__ codes,
                                                               const uint32_t* __restrict__ mask,
                                                               const uint32_t* __restrict__ w_tile,
                                                               const uint32_t* __restrict__ w_batch, uint32_t n_work,
                                                               unsigned long long* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t w0 = (uint32_t)(((uint64_t)blockIdx.x * n_work) / gridDim.x);
  const uint32_t w1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * n_work) / gridDim.x);
  if (w0 >= w1) return;
  {
    uint4* z = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = tid; i < K0_LDS_WORDS / 4; i += K0_BLOCK) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  uint32_t cur = w_batch[w0];
  uint32_t since_flush = 0;
  LaneData nxt = k0_load(codes, mask, w_tile[w0], tid);
  for (uint32_t w = w0; w < w1; ++w) {
    const uint32_t b = w_batch[w];
    if (b != cur || since_flush >= K0_FORCE_FLUSH_TILES) {
      __syncthreads();
      k0_flush(lds, acc + (size_t)cur * PG_ACC_WORDS, tid);
      cur = b;
      since_flush = 0;
    }
    const LaneData d = nxt;
    if (w + 1 < w1) nxt = k0_load(codes, mask, w_tile[w + 1], tid);
    k0_process(d, lds, tid);
    ++since_flush;
  }
  __syncthreads();
  k0_flush(lds, acc + (size_t)cur * PG_ACC_WORDS, tid);
}

// ---- K1: counts + Z-scores --------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rc_index(uint32_t x, int k) {
  // reverse complement of a k-mer index (first base most significant): complement = 3 - digit, then reverse digits
  uint32_t c = ((1u << (2 * k)) - 1u) - x, r = 0;
  for (int i = 0; i < k; ++i) { r = (r << 2) | (c & 3u); c >>= 2; }
  return r;
}

// One 256-thread block per genome.  If acc != nullptr: build c2/c3/c4 from the K0 accumulators (and store them
// to counts); else read c2/c3/c4 from counts.  Then the Z-scores in the reference's operation order.
__global__ __launch_bounds__(256) void tetra_finalize_kernel(const unsigned long long* __restrict__ acc,
                                                             const uint32_t* __restrict__ quirk,
                                                             const uint32_t* __restrict__ batch_gid,
                                                             unsigned long long* __restrict__ counts,
                                                             double* __restrict__ z, uint8_t* __restrict__ present) {
  __shared__ unsigned long long F4[256], F3[64], F2[16], c4[256], c3[64], c2[16];
  const uint32_t g = blockIdx.x, t = threadIdx.x;
  unsigned long long* cg = counts + (size_t)g * PG_ACC_WORDS;
  if (acc) {
    const unsigned long long* a = acc + (size_t)g * PG_ACC_WORDS;
    F4[t] = a[80 + t];
    __syncthreads();
    if (t < 64) F3[t] = F4[4 * t] + F4[4 * t + 1] + F4[4 * t + 2] + F4[4 * t + 3] + a[16 + t];
    __syncthreads();
    if (t < 16) F2[t] = F3[4 * t] + F3[4 * t + 1] + F3[4 * t + 2] + F3[4 * t + 3] + a[t];
    __syncthreads();
    c4[t] = F4[t] + F4[rc_index(t, 4)] - (unsigned long long)quirk[(size_t)batch_gid[g] * 256 + t];
    if (t < 64) c3[t] = F3[t] + F3[rc_index(t, 3)];
    if (t < 16) c2[t] = F2[t] + F2[rc_index(t, 2)];
    __syncthreads();
    cg[80 + t] = c4[t];
    if (t < 64) cg[16 + t] = c3[t];
    if (t < 16) cg[t] = c2[t];
  } else {
    c4[t] = cg[80 + t];
    if (t < 64) c3[t] = cg[16 + t];
    if (t < 16) c2[t] = cg[t];
    __syncthreads();
  }
  if (!z) return;
  const unsigned long long obs = c4[t];
  double zv = 0.0;
  uint8_t pv = 0;
  if (obs != 0) {
    const unsigned long long a = c3[t >> 2], b = c3[t & 63u], den = c2[(t >> 2) & 15u];
    const double e = ((1.0 * (double)a) * (double)b) / (double)den;                                   // tetra.py:121-123
    const double sd = sqrt(((e * (double)(den - a)) * (double)(den - b)) / (double)(den * den));      // tetra.py:129-132
    if (sd != 0.0) zv = ((double)obs - e) / sd;                                                        // tetra.py:134
    else zv = 1.0 / (double)(den * den);                                                               // tetra.py:135-138
    pv = 1;
  }
  z[(size_t)g * 256 + t] = zv;
  present[(size_t)g * 256 + t] = pv;
}

// ---- K2: per-genome statistics (sequential sums, CPython-3.10 sum() order) ------------------------------------
// One thread per genome.  dev[g][0..cnt) = z - mean over the present keys in tetramer order; ss[g] = sum(dev^2).
// flags[0] |= 1 if present[g] differs from present[0] (AssertionError in the reference); flags[1] = cnt.
__global__ __launch_bounds__(64) void tetra_stats_kernel(const double* __restrict__ z, const uint8_t* __restrict__ present,
                                                         uint32_t n, double* __restrict__ dev, double* __restrict__ ss,
                                                         int32_t* __restrict__ flags) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const double* zg = z + (size_t)g * 256;
  const uint8_t* pg = present + (size_t)g * 256;
  const uint8_t* p0 = present;
  double s = 0.0;
  int cnt = 0;
  bool same = true;
  for (int t = 0; t < 256; ++t) {
    const uint8_t p = pg[t];
    same = same && (p == p0[t]);
    if (p) { s = s + zg[t]; ++cnt; }
  }
  if (!same) atomicOr(&flags[0], 1);
  if (g == 0) flags[1] = cnt;
  double* dg = dev + (size_t)g * 256;
  double acc = 0.0;
  if (cnt > 0) {
    const double m = s / (double)cnt;
    int k = 0;
    for (int t = 0; t < 256; ++t) {
      if (pg[t]) {
        const double d = zg[t] - m;
        dg[k++] = d;
        acc = acc + d * d;
      }
    }
  }
  ss[g] = acc;
}

// ---- K3: Pearson matrix ---------------------------------------------------------------------------------------
// 32x32 tile of pairs per 256-thread block, 2x2 pairs per thread; the 256-term dot product of each pair is
// accumulated strictly in k order (each product rounded, then added: no FMA), as tetra.py:186-188 does.
constexpr int K3_T = 32, K3_KC = 64;

__global__ __launch_bounds__(256) void tetra_pairs_kernel(const double* __restrict__ dev, const double* __restrict__ ss,
                                                          const int32_t* __restrict__ flags, uint32_t n, uint32_t row0,
                                                          uint32_t nrows, double* __restrict__ out, int mirror) {
  __shared__ double A[K3_T][K3_KC + 1], B[K3_T][K3_KC + 1];
  const uint32_t ti = blockIdx.y, tj = blockIdx.x;
  if (mirror && tj < ti) return;  // upper-triangle tiles only; results are mirrored (r(i,j) == r(j,i) bitwise)
  const int cnt = flags[1];
  const uint32_t i0 = row0 + ti * K3_T, j0 = tj * K3_T;
  const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  for (int k0 = 0; k0 < cnt; k0 += K3_KC) {
    const int kc = min(K3_KC, cnt - k0);
    for (int e = threadIdx.x; e < K3_T * K3_KC; e += 256) {
      const int r = e / K3_KC, k = e % K3_KC;
      const uint32_t gi = i0 + r, gj = j0 + r;
      A[r][k] = (gi < row0 + nrows && gi < n && k < kc) ? dev[(size_t)gi * 256 + k0 + k] : 0.0;
      B[r][k] = (gj < n && k < kc) ? dev[(size_t)gj * 256 + k0 + k] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      const double a0 = A[ty][k], a1 = A[ty + 16][k], b0 = B[tx][k], b1 = B[tx + 16][k];
      acc[0][0] = acc[0][0] + a0 * b0;
      acc[0][1] = acc[0][1] + a0 * b1;
      acc[1][0] = acc[1][0] + a1 * b0;
      acc[1][1] = acc[1][1] + a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const uint32_t i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      if (i >= row0 + nrows || i >= n || j >= n) continue;
      const double r = (i == j) ? 1.0 : acc[a][b] / sqrt(ss[i] * ss[j]);   // tetra.py:171 (diag), :190-192
      if (!mirror) {
        out[(size_t)(i - row0) * n + j] = r;
      } else if (j >= i) {
        out[(size_t)i * n + j] = r;
        out[(size_t)j * n + i] = r;
      }
    }
}

}  // namespace

// ---- host launchers ---------------------------------------------------------------------------------------------
int pg_launch_tetra_count(pg_ctx* ctx, uint32_t n_batch) {
  static bool attr_set = false;
  const size_t lds_bytes = K0_LDS_WORDS * sizeof(uint32_t);
  if (!attr_set) {
    PG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tetra_count_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  PG_HIP(ctx, hipMemsetAsync(ctx->d_acc, 0, (size_t)n_batch * PG_ACC_WORDS * sizeof(unsigned long long), ctx->stream));
  if (ctx->n_work == 0) return PG_OK;
  const uint32_t grid = ctx->n_work < (uint32_t)ctx->num_cu ? ctx->n_work : (uint32_t)ctx->num_cu;
  pg_prof_begin(ctx, PG_K_TETRA_COUNT);
  hipLaunchKernelGGL(tetra_count_kernel, dim3(grid), dim3(K0_BLOCK), lds_bytes, ctx->stream, ctx->d_codes, ctx->d_mask,
                     ctx->d_w_tile, ctx->d_w_batch, ctx->n_work, ctx->d_acc);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_finalize(pg_ctx* ctx, uint32_t n_batch, const unsigned long long* d_acc_in) {
  if (n_batch == 0) return PG_OK;
  pg_prof_begin(ctx, PG_K_TETRA_FINALIZE);
  hipLaunchKernelGGL(tetra_finalize_kernel, dim3(n_batch), dim3(256), 0, ctx->stream, d_acc_in, ctx->d_quirk,
                     ctx->d_batch_gid, ctx->d_counts, ctx->d_z, ctx->d_present);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_stats(pg_ctx* ctx, const double* d_z, const uint8_t* d_present, uint32_t n) {
  PG_HIP(ctx, hipMemsetAsync(ctx->d_flags, 0, 2 * sizeof(int32_t), ctx->stream));
  if (n == 0) return PG_OK;
  pg_prof_begin(ctx, PG_K_TETRA_STATS);
  hipLaunchKernelGGL(tetra_stats_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_z, d_present, n, ctx->d_dev,
                     ctx->d_ss, ctx->d_flags);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}

int pg_launch_tetra_pairs(pg_ctx* ctx, uint32_t n, uint32_t row0, uint32_t nrows, double* d_out, bool mirror) {
  if (n == 0 || nrows == 0) return PG_OK;
  const dim3 grid((n + K3_T - 1) / K3_T, (nrows + K3_T - 1) / K3_T);
  pg_prof_begin(ctx, PG_K_TETRA_PAIRS);
  hipLaunchKernelGGL(tetra_pairs_kernel, grid, dim3(256), 0, ctx->stream, ctx->d_dev, ctx->d_ss, ctx->d_flags, n, row0,
                     nrows, d_out, mirror ? 1 : 0);
  pg_prof_end(ctx);
  PG_HIP(ctx, hipGetLastError());
  return PG_OK;
}
