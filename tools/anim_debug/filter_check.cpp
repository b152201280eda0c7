// filter_check.cpp — HOST check of pga::lis_filter (the delta-filter -1 emulation) against real MUMmer output:
// reads alignments "rid qid rs re qs qe err" (1-based closed, qs>qe for reverse) from stdin, prints the kept ones.
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include "pg_anim_core.h"
using namespace pga;
int main() {
  std::vector<Aln> a; std::vector<int32_t> rg, qg; std::map<std::string,int> rid, qid;
  char r[512], q[512]; int rs, re, qs, qe, err;
  while (scanf("%511s %511s %d %d %d %d %d", r, q, &rs, &re, &qs, &qe, &err) == 7) {
    Aln x; x.strand = qs > qe; x.rs = rs - 1; x.re = re; x.qs = (qs < qe ? qs : qe) - 1; x.qe = qs < qe ? qe : qs; x.errors = err; x.keep = 0;
    a.push_back(x);
    rg.push_back(rid.emplace(r, (int)rid.size()).first->second); qg.push_back(qid.emplace(q, (int)qid.size()).first->second);
  }
  const int n = (int)a.size();
  std::vector<int32_t> idx(n + 1), from(n + 1); std::vector<double> sc(n + 1);
  lis_filter(a.data(), n, 0, rg.data(), qg.data(), idx.data(), sc.data(), from.data());
  lis_filter(a.data(), n, 1, qg.data(), rg.data(), idx.data(), sc.data(), from.data());
  for (int i = 0; i < n; ++i) if (a[i].keep == 3) printf("%d %d %d %d %d\n", a[i].rs + 1, a[i].re, a[i].strand ? a[i].qe : a[i].qs + 1, a[i].strand ? a[i].qs + 1 : a[i].qe, a[i].errors);
  return 0;
}
