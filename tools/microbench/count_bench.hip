// count_bench.hip — times variants of the TETRA count kernel (pyani_amd/csrc/pg_tetra_count.h) on random all-clean
// data shaped like BASELINE config C2 (200 genomes x 77 super-tiles ~ 1.0e9 bases), to attribute time to the memory
// pipe vs the LDS atomics and to pick the prefetch depth.  hipcc --offload-arch=gfx950 -O3 -I../../pyani_amd/csrc -I../../include
#include "pg_internal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace {
#include "pg_tetra_count.h"
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int PF, int MODE, int BLOCK = 1024>
void run(const char* name, const uint32_t* codes, const uint32_t* mask, const uint32_t* wt, const uint32_t* wb, uint32_t nw,
         unsigned long long* acc, uint32_t nb, int grid, double bytes) {
  auto k = tetra_count_kernel<PF, MODE, BLOCK>;
  const size_t lds = 0;   // the kernel's LDS is static
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, 0, codes, mask, wt, wb, nb, acc);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, 0, codes, mask, wt, wb, nb, acc);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps;
  printf("%-28s BLOCK=%d PF=%d grid=%d  %8.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)\n", name, BLOCK, PF, grid, us, bytes / us * 1e-3, bytes / us * 1e-3 / 80.0);
}

int main(int argc, char** argv) {
  const uint32_t n_gen = 200, tiles_per = 77;
  const uint32_t nw = n_gen * tiles_per;
  const uint64_t bases = (uint64_t)nw * PG_SUPER;
  std::vector<uint32_t> h_codes(bases / 16 + PG_SUPER / 16), h_mask(bases / 32 + PG_SUPER / 32, 0xFFFFFFFFu);
  uint64_t s = 88172645463325252ull;
  const double gc = argc > 1 ? atof(argv[1]) : 0.5;        // GC fraction of the random sequence
  const int dirty_per_genome = argc > 2 ? atoi(argv[2]) : 0;  // isolated dirty bases sprinkled into every genome
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (auto& c : h_codes) {
    uint32_t w = 0;
    for (int k = 0; k < 16; ++k) {
      const uint64_t r = rnd();
      const bool is_gc = (double)(r >> 11) * (1.0 / 9007199254740992.0) < gc;
      const uint32_t b = is_gc ? ((r & 1) ? 1u : 2u) : ((r & 1) ? 0u : 3u);
      w |= b << (2 * k);
    }
    c = w;
  }
  for (uint32_t g = 0; g < n_gen; ++g)
    for (int d = 0; d < dirty_per_genome; ++d) {
      const uint64_t pos = (uint64_t)g * tiles_per * PG_SUPER + rnd() % ((uint64_t)tiles_per * PG_SUPER);
      h_mask[pos >> 5] &= ~(1u << (pos & 31));
    }
  printf("gc=%.2f dirty/genome=%d\n", gc, dirty_per_genome);
  std::vector<uint32_t> h_wt(n_gen), h_wb(n_gen + 1);   // seg_tile0, seg_prefix
  for (uint32_t i = 0; i < n_gen; ++i) { h_wt[i] = i * tiles_per; h_wb[i] = i * tiles_per; }
  h_wb[n_gen] = nw;
  uint32_t *codes, *mask, *wt, *wb; unsigned long long* acc;
  CK(hipMalloc(&codes, h_codes.size() * 4)); CK(hipMalloc(&mask, h_mask.size() * 4));
  CK(hipMalloc(&wt, n_gen * 4)); CK(hipMalloc(&wb, (n_gen + 1) * 4)); CK(hipMalloc(&acc, (size_t)n_gen * PG_ACC_WORDS * 8));
  CK(hipMemcpy(codes, h_codes.data(), h_codes.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(mask, h_mask.data(), h_mask.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wt, h_wt.data(), n_gen * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wb, h_wb.data(), (n_gen + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(acc, 0, (size_t)n_gen * PG_ACC_WORDS * 8));
  const double bytes = (double)bases * 0.375;
  printf("bases %.3e, bytes %.1f MB\n", (double)bases, bytes / 1e6);
  run<1, 0, 1024>("full", codes, mask, wt, wb, nw, acc, n_gen, 256, bytes);
  run<2, 0, 1024>("full", codes, mask, wt, wb, nw, acc, n_gen, 256, bytes);
  run<1, 0, 1024>("full", codes, mask, wt, wb, nw, acc, n_gen, 512, bytes);
  run<2, 0, 1024>("full", codes, mask, wt, wb, nw, acc, n_gen, 512, bytes);
  run<1, 0, 512>("full", codes, mask, wt, wb, nw, acc, n_gen, 512, bytes);
  run<2, 0, 512>("full", codes, mask, wt, wb, nw, acc, n_gen, 512, bytes);
  run<3, 0, 512>("full", codes, mask, wt, wb, nw, acc, n_gen, 512, bytes);
  run<2, 0, 512>("full", codes, mask, wt, wb, nw, acc, n_gen, 1024, bytes);
  run<1, 1, 1024>("loads only", codes, mask, wt, wb, nw, acc, n_gen, 512, bytes);
  run<1, 2, 1024>("atomics only", codes, mask, wt, wb, nw, acc, n_gen, 512, bytes);
  return 0;
}
