// Host-only logic of the engine that needs no HIP: per-gene count histograms (the sufficient statistic of every
// lgamma / digamma term of the negative binomial), their compaction from the dense device tables,
// and the task list of the histogram kernel.  Header-only and free of HIP types so that tests/host_logic_sanitize.cpp
// can run it under AddressSanitizer / UBSan on the CPU (sanitizers are not available on the GPU pool).
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <thread>
#include <utility>
#include <vector>

#define VC_HIST_CAP 2048          // counts 1 .. VC_HIST_CAP-1 have a dense per-gene bin; everything else is "overflow"

// a count the negative binomial / Poisson likelihood accepts: finite and >= 0
static inline bool vc_count_ok(float v) { return v >= 0.f && v <= 3.0e38f; }
// does the value own a dense bin?  (no float -> int conversion of an out-of-range value: that is undefined behaviour)
static inline bool vc_count_dense(float v) { return v > 0.f && v < (float)VC_HIST_CAP && (float)(int)v == v; }

// Histogram CSR of one matrix from the dense per-gene tables filled on the device (tab[g * VC_HIST_CAP + k] = number of
// cells with count k) plus the overflow entries (gene, value): distinct values in increasing order, dense bins first.
// Appends to ptr / val / cnt; returns sum over all entries of lgamma(k + 1) (the constant of the log-likelihood).
static inline double vc_compact_hist(const unsigned* tab, int Ng, const std::vector<std::pair<int, float>>& ovf,
                                     std::vector<int>& ptr, std::vector<float>& val, std::vector<float>& cnt) {
  std::vector<std::map<float, unsigned>> over(Ng);
  for (const auto& e : ovf)
    if (e.first >= 0 && e.first < Ng) over[e.first][e.second]++;
  double tot = 0.0;
  for (int g = 0; g < Ng; ++g) {
    ptr.push_back((int)val.size());
    const unsigned* row = tab + (size_t)g * VC_HIST_CAP;
    for (int k = 1; k < VC_HIST_CAP; ++k)
      if (row[k]) {
        val.push_back((float)k);
        cnt.push_back((float)row[k]);
        tot += (double)row[k] * std::lgamma((double)k + 1.0);
      }
    for (const auto& kv : over[g]) {
      val.push_back(kv.first);
      cnt.push_back((float)kv.second);
      tot += (double)kv.second * std::lgamma((double)kv.first + 1.0);
    }
  }
  return tot;
}

// The same histograms straight from a host copy of the matrix (element (g, c) at M[g * gs + c * cs]): the checker of the
// device path (VC_HOST_HIST=1) and the reference for tests.  *bad is set when a value is negative / NaN / infinite.
static inline double vc_build_hist_host(const float* M, long long gs, long long cs, int Ng, int Nc, std::vector<int>& ptr_out,
                                        std::vector<float>& val, std::vector<float>& cnt, bool* bad) {
  std::vector<std::vector<std::pair<float, float>>> per_gene(Ng);
  std::vector<double> lg(Ng, 0.0);
  std::vector<char> bad_t;
  unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  nt = std::min<unsigned>(nt, (unsigned)std::max(Ng, 1));
  bad_t.assign(nt, 0);
  auto work = [&](int ga, int gb, int tid) {
    const int n = gb - ga;
    std::vector<unsigned> dense((size_t)n * VC_HIST_CAP, 0u);
    std::vector<std::map<float, unsigned>> over(n);
    auto put = [&](int gi, float v) {
      if (!vc_count_ok(v)) { bad_t[tid] = 1; return; }
      if (v == 0.f) return;
      if (vc_count_dense(v)) dense[(size_t)gi * VC_HIST_CAP + (int)v]++;
      else over[gi][v]++;
    };
    if (gs <= cs) {
      for (int c = 0; c < Nc; ++c) {
        const float* row = M + (long long)c * cs;
        for (int g = ga; g < gb; ++g) put(g - ga, row[(long long)g * gs]);
      }
    } else {
      for (int g = ga; g < gb; ++g) {
        const float* row = M + (long long)g * gs;
        for (int c = 0; c < Nc; ++c) put(g - ga, row[(long long)c * cs]);
      }
    }
    for (int gi = 0; gi < n; ++gi) {
      auto& out = per_gene[ga + gi];
      double s = 0.0;
      for (int k = 1; k < VC_HIST_CAP; ++k) {
        const unsigned m = dense[(size_t)gi * VC_HIST_CAP + k];
        if (m) { out.emplace_back((float)k, (float)m); s += (double)m * std::lgamma((double)k + 1.0); }
      }
      for (auto& kv : over[gi]) {
        out.emplace_back(kv.first, (float)kv.second);
        s += (double)kv.second * std::lgamma((double)kv.first + 1.0);
      }
      lg[ga + gi] = s;
    }
  };
  std::vector<std::thread> th;
  const int per = (Ng + (int)nt - 1) / (int)nt;
  for (unsigned t = 0; t < nt; ++t) {
    const int ga = (int)t * per, gb = std::min(Ng, ga + per);
    if (ga < gb) th.emplace_back(work, ga, gb, (int)t);
  }
  for (auto& t : th) t.join();
  double tot = 0.0;
  for (int g = 0; g < Ng; ++g) {
    ptr_out.push_back((int)val.size());
    for (auto& kv : per_gene[g]) { val.push_back(kv.first); cnt.push_back(kv.second); }
    tot += lg[g];
  }
  if (bad)
    for (char b : bad_t) if (b) *bad = true;
  return tot;
}

// Tasks of the histogram kernel: runs of <= 64 histogram entries of one gene and matrix {gene, matrix, begin, end},
// sorted by gene; tptr[g] = first task of gene g.  ptr is the CSR of [S genes..., U genes..., end].
static inline void vc_build_hist_tasks(const std::vector<int>& ptr, int Ng, std::vector<int>& task, std::vector<int>& tptr) {
  for (int g = 0; g < Ng; ++g) {
    tptr.push_back((int)task.size() / 4);
    for (int m = 0; m < 2; ++m)
      for (int beg = ptr[(size_t)m * Ng + g], end = ptr[(size_t)m * Ng + g + 1]; beg < end; beg += 64)
        task.insert(task.end(), {g, m, beg, std::min(end, beg + 64)});
  }
  tptr.push_back((int)task.size() / 4);
}
