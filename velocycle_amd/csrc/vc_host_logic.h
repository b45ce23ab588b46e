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

#if defined(__HIPCC__)
#define VC_HD __host__ __device__
#else
#define VC_HD
#endif

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
// device path (tuning.host_hist) and the reference for tests.  *bad is set when a value is negative / NaN / infinite.
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

// Dense form of the same sufficient statistic (round 4), for matrices whose non-zero counts are all integers < VC_HIST_CAP:
//   sum_k cnt_k [lgamma(r + k) - lgamma(r)] = sum_{j >= 0} C_j log(r + j),   C_j = number of cells with count > j
//   sum_k cnt_k [psi(r + k)    - psi(r)]    = sum_{j >= 0} C_j / (r + j)
// (lgamma(r + k) - lgamma(r) = sum_{j < k} log(r + j) for integer k): a logarithm and a reciprocal per (gene, j) instead of a
// Stirling-series evaluation per distinct count value, and -- laid out [gene block of 64][j][gene] -- 64 genes per wave
// instruction.  Appends the rows of one matrix: rows[gb] = the largest count of the block's genes (0: all zero; row j = C_j),
// off[gb] = index of the block's first row in HC (rows of 64 floats).  Counts are exact in float up to 2^24 cells.
static inline void vc_build_dense_hist(const unsigned* tab, int Ng, int Ng_pad, std::vector<float>& HC, std::vector<int>& off,
                                       std::vector<int>& rows) {
  const int nblk = Ng_pad / 64;
  for (int gb = 0; gb < nblk; ++gb) {
    int kmax = 0;
    for (int l = 0; l < 64; ++l) {
      const int g = gb * 64 + l;
      if (g >= Ng) break;
      const unsigned* row = tab + (size_t)g * VC_HIST_CAP;
      for (int k = VC_HIST_CAP - 1; k > kmax; --k)
        if (row[k]) { kmax = k; break; }
    }
    off.push_back((int)(HC.size() / 64));
    rows.push_back(kmax);
    const size_t base = HC.size();
    HC.resize(base + (size_t)kmax * 64, 0.f);
    for (int l = 0; l < 64; ++l) {
      const int g = gb * 64 + l;
      if (g >= Ng) break;
      const unsigned* row = tab + (size_t)g * VC_HIST_CAP;
      unsigned long long tail = 0;                       // cells with count > j, built from the top
      for (int j = kmax - 1; j >= 0; --j) {
        tail += row[j + 1];
        HC[base + (size_t)j * 64 + l] = (float)tail;
      }
    }
  }
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


// ---------------------------------------------------------------------------------------------
// Cell tiling of the likelihood kernel.  Workgroup (chunk, gene block gb) has the linear index
// chunk * nGB + gb and runs in dispatch pass index / pass_wgs (pass_wgs = CUs: the dispatcher places one workgroup per CU
// per pass); the waves of pass p take pass_cw[p] consecutive cells each (passes >= 3: pass_cw[3]).  Chunks of one gene
// block are contiguous per pass: first chunk of pass p = ceil((p * pass_wgs - gb) / nGB).
// ---------------------------------------------------------------------------------------------
// first cell and cells-per-wave of wave `wave` (0..waves-1) of workgroup (chunk, gb); the caller clamps to the cell count
VC_HD static inline long long vc_wave_first_cell(int chunk, int gb, int wave, int nGB, int pass_wgs, const int* pass_cw,
                                                 int waves, int* my_cw) {
  long long cbeg = 0;
  int cw = pass_cw[0];
  int c_lo = 0;
  for (int p = 0; p < 4; ++p) {
    const long long nxt = (long long)(p + 1) * pass_wgs - gb;
    const int c_hi = (p == 3) ? 0x7fffffff : (nxt > 0 ? (int)((nxt + nGB - 1) / nGB) : 0);
    if (chunk >= c_hi) cbeg += (long long)(c_hi - c_lo) * (waves * pass_cw[p]);
    else if (chunk >= c_lo) { cbeg += (long long)(chunk - c_lo) * (waves * pass_cw[p]); cw = pass_cw[p]; }
    c_lo = c_hi;
  }
  *my_cw = cw;
  return cbeg + (long long)wave * cw;
}

struct VcTiling { int n_chunks, cw, pass_cw[4]; };

// Balanced tiling (every wave `cw` cells) or, when `share` is given, cells per wave proportional to share[p] in pass p,
// widened until every gene block covers all Nc cells.  slots = workgroups resident at once (occupancy x CUs).
static inline VcTiling vc_tile_cells(long long Nc, int nGB, int n_cu, int blocks_per_cu, int waves, long long cw_override,
                                     const double* share, int min_cw) {
  VcTiling t;
  const long long slots = (long long)blocks_per_cu * n_cu;
  long long chunks = slots / nGB;
  if (chunks < 1) chunks = 1;
  long long cw = (Nc + waves * chunks - 1) / (waves * chunks);
  if (cw < 8) cw = 8;       // keep the per-gene prologue/epilogue amortised (measured: 8 beats 16 for Nc <= 6250, r01)
  if (cw_override > 0) cw = cw_override;
  t.cw = (int)cw;
  const long long per_wg = (long long)waves * t.cw;
  t.n_chunks = (int)((Nc + per_wg - 1) / per_wg);
  for (int p = 0; p < 4; ++p) t.pass_cw[p] = t.cw;
  const int P = blocks_per_cu > 4 ? 4 : blocks_per_cu;
  if (!share || blocks_per_cu > 4 || P < 2 || cw_override > 0 || t.cw < min_cw ||
      (long long)nGB * t.n_chunks <= (long long)(P - 1) * n_cu)
    return t;
  for (int p = 0; p < P; ++p) if (!(share[p] > 0.0)) return t;
  auto first = [&](int p, int gb) -> long long {
    const long long x = (long long)p * n_cu - gb;
    return x > 0 ? (x + nGB - 1) / nGB : 0;
  };
  const long long n_ch = (long long)P * n_cu / nGB;              // chunks per gene block of the full grid
  double ssum = 0;
  for (int p = 0; p < P; ++p) ssum += share[p];
  int pcw[4];
  for (int p = 0; p < 4; ++p) {
    const int q = p < P ? p : P - 1;
    pcw[p] = (int)std::ceil(share[q] / ssum * (double)Nc / (waves * ((double)n_ch / P)));
    if (pcw[p] < 8) pcw[p] = 8;
  }
  auto covered = [&](int gb) -> long long {
    long long c = 0;
    for (int p = 0; p < P; ++p) {
      long long lo = first(p, gb), hi = (p == P - 1) ? n_ch : first(p + 1, gb);
      if (hi > n_ch) hi = n_ch;
      if (hi > lo) c += (hi - lo) * waves * pcw[p];
    }
    return c;
  };
  for (int it = 0; it < 1000000; ++it) {                         // widen every pass by one cell until every gene block is covered
    long long mn = covered(0);
    for (int gb = 1; gb < nGB; ++gb) { const long long c = covered(gb); if (c < mn) mn = c; }
    if (mn >= Nc) break;
    for (int p = 0; p < 4; ++p) ++pcw[p];
  }
  int mx = 0;
  for (int p = 0; p < 4; ++p) { t.pass_cw[p] = pcw[p]; if (pcw[p] > mx) mx = pcw[p]; }
  t.cw = mx;
  t.n_chunks = (int)n_ch;
  return t;
}

// ---------------------------------------------------------------------------------------------
// One-hot batch design ("batches"): the reference builds Db as one indicator column per unique id
// (preprocessing.py:65-93), so sum_b Db[b,c] dnu[b,g] = dnu[b(c),g].  When every workgroup of the likelihood kernel lies inside one
// batch, that offset joins the constant harmonic once per wave and the kernel without batch terms runs for any number of batches.
// ---------------------------------------------------------------------------------------------
// Batch id of every cell of a (Nb, Nc) row-major design matrix, or false when some cell does not have exactly one 1 and Nb - 1 zeros.
static inline bool vc_onehot_batches(const float* Db, int Nb, long long Nc, std::vector<int>& bid) {
  bid.assign((size_t)Nc, -1);
  for (int q = 0; q < Nb; ++q) {
    const float* row = Db + (size_t)q * Nc;
    for (long long c = 0; c < Nc; ++c) {
      const float v = row[c];
      if (v == 0.f) continue;
      if (v != 1.f || bid[(size_t)c] >= 0) return false;
      bid[(size_t)c] = q;
    }
  }
  for (long long c = 0; c < Nc; ++c) if (bid[(size_t)c] < 0) return false;
  return true;
}

// Cells ordered by batch (stable): pos[c] = position of cell c, ord[p] = cell at position p, len[q] = cells of batch q.
// Returns true when the cells are in that order already (pos is then the identity).
static inline bool vc_order_by_batch(const std::vector<int>& bid, int Nb, std::vector<int>& pos, std::vector<int>& ord, std::vector<int>& len) {
  const size_t n = bid.size();
  len.assign((size_t)Nb, 0);
  bool sorted = true;
  for (size_t c = 0; c < n; ++c) { len[(size_t)bid[c]]++; if (c && bid[c] < bid[c - 1]) sorted = false; }
  std::vector<long long> start((size_t)Nb + 1, 0);
  for (int q = 0; q < Nb; ++q) start[(size_t)q + 1] = start[(size_t)q] + len[(size_t)q];
  pos.resize(n); ord.resize(n);
  std::vector<long long> cur(start.begin(), start.end() - 1);
  for (size_t c = 0; c < n; ++c) { const long long p = cur[(size_t)bid[c]]++; pos[c] = (int)p; ord[(size_t)p] = (int)c; }
  return sorted;
}

// The workgroup table of the likelihood kernel {first cell, cells per wave, batch, end} with every workgroup inside one batch
// (cells = positions; batch q holds positions [sum len[<q], + len[q])).  The chunks of a gene block keep their nominal weights
// (pass_cw of their dispatch pass: the unequal shares of vc_tile_cells); consecutive groups of chunks are given to the batches
// in proportion to their cells, every non-empty batch at least one chunk, and inside a group the cells are split in proportion
// to the weights -- exactly: the groups tile the batch without gaps or overlap.  bat_chunk[gb][q] = first chunk of batch q.
// Needs n_chunks >= number of non-empty batches.  Returns the largest cells-per-wave of the table.
static inline int vc_tile_batches(long long Nc, int nGB, int n_chunks, int pass_wgs, const int* pass_cw, int waves,
                                  const std::vector<int>& len, std::vector<int>& tile, std::vector<int>& bat_chunk) {
  const int Nb = (int)len.size();
  tile.assign(4 * (size_t)nGB * n_chunks, 0);
  bat_chunk.assign((size_t)nGB * (Nb + 1), 0);
  int nonempty = 0;
  for (int q = 0; q < Nb; ++q) nonempty += len[(size_t)q] > 0;
  int cw_max = 0;
  std::vector<double> pre((size_t)n_chunks + 1);
  for (int gb = 0; gb < nGB; ++gb) {
    pre[0] = 0.0;
    for (int k = 0; k < n_chunks; ++k) {
      long long pass = ((long long)k * nGB + gb) / (pass_wgs > 0 ? pass_wgs : 1);
      if (pass > 3) pass = 3;
      pre[(size_t)k + 1] = pre[(size_t)k] + (double)(pass_cw[pass] > 0 ? pass_cw[pass] : 1);
    }
    const double Wtot = pre[(size_t)n_chunks];
    int k = 0, left = nonempty;
    long long seg = 0, cum = 0;
    for (int q = 0; q < Nb; ++q) {
      bat_chunk[(size_t)gb * (Nb + 1) + q] = k;
      const long long L = len[(size_t)q];
      if (L == 0) continue;
      --left;
      cum += L;
      int k_end = k + 1;
      if (left == 0) k_end = n_chunks;
      else {
        const double target = Wtot * (double)cum / (double)Nc;
        while (k_end < n_chunks - left && std::fabs(pre[(size_t)k_end + 1] - target) <= std::fabs(pre[(size_t)k_end] - target)) ++k_end;
      }
      const double W0 = pre[(size_t)k], Wg = pre[(size_t)k_end] - W0;
      for (int j = k; j < k_end; ++j) {
        const long long a = (long long)std::floor((double)L * ((pre[(size_t)j] - W0) / Wg));
        const long long bnd = (j + 1 == k_end) ? L : (long long)std::floor((double)L * ((pre[(size_t)j + 1] - W0) / Wg));
        const long long cells = bnd > a ? bnd - a : 0;
        const int cw = (int)((cells + waves - 1) / waves);
        int* t = &tile[4 * ((size_t)j * nGB + gb)];
        t[0] = (int)(seg + a); t[1] = cw; t[2] = q; t[3] = (int)(seg + a + cells);
        if (cw > cw_max) cw_max = cw;
      }
      k = k_end;
      seg += L;
    }
    bat_chunk[(size_t)gb * (Nb + 1) + Nb] = n_chunks;
    // (chunks behind the last non-empty batch do not exist: the last group ends at n_chunks)
  }
  return cw_max;
}

