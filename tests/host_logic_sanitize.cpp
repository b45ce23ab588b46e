// CPU-only exercise of the engine's HIP-free host logic (velocycle_amd/csrc/vc_host_logic.h) for AddressSanitizer / UBSan
// (tests/test_host_logic_sanitize_cpu.py builds this with g++ -fsanitize=address,undefined: sanitizers are not
// available on the GPU pool, so the host half is checked here).  Exits non-zero on any mismatch.
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <utility>
#include <vector>

#include "../velocycle_amd/csrc/vc_host_logic.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

// the device pass, restated on the host: dense per-gene tables + overflow list, in an arbitrary (shuffled) order
static void device_like(const std::vector<float>& M, long long gs, long long cs, int Ng, int Nc, std::vector<unsigned>& tab,
                        std::vector<std::pair<int, float>>& ovf, bool& bad, std::mt19937& rng) {
  tab.assign((size_t)Ng * VC_HIST_CAP, 0u);
  std::vector<std::pair<int, int>> order;
  for (int g = 0; g < Ng; ++g) for (int c = 0; c < Nc; ++c) order.push_back({g, c});
  std::shuffle(order.begin(), order.end(), rng);
  for (auto gc : order) {
    const float v = M[(size_t)gc.first * gs + (size_t)gc.second * cs];
    if (!vc_count_ok(v)) { bad = true; continue; }
    if (v == 0.f) continue;
    if (vc_count_dense(v)) tab[(size_t)gc.first * VC_HIST_CAP + (int)v]++;
    else ovf.push_back({gc.first, v});
  }
}

static void one_case(int Ng, int Nc, bool gene_major, bool spiky, unsigned seed) {
  std::mt19937 rng(seed);
  std::poisson_distribution<int> pois(1.3);
  const long long gs = gene_major ? Nc : 1, cs = gene_major ? 1 : Ng;
  std::vector<float> S((size_t)Ng * Nc), U((size_t)Ng * Nc);
  for (auto* M : {&S, &U})
    for (auto& v : *M) {
      v = (float)pois(rng);
      if (spiky && rng() % 97 == 0) v = (float)(2000 + rng() % 100000);
      if (spiky && rng() % 101 == 0) v = 0.25f + (float)(rng() % 64) * 0.5f;
    }
  std::vector<int> ptr_h, ptr_d;
  std::vector<float> val_h, cnt_h, val_d, cnt_d;
  bool bad_h = false, bad_d = false;
  const double lgS = vc_build_hist_host(S.data(), gs, cs, Ng, Nc, ptr_h, val_h, cnt_h, &bad_h);
  const double lgU = vc_build_hist_host(U.data(), gs, cs, Ng, Nc, ptr_h, val_h, cnt_h, &bad_h);
  ptr_h.push_back((int)val_h.size());
  double lgd[2];
  int m = 0;
  for (auto* M : {&S, &U}) {
    std::vector<unsigned> tab;
    std::vector<std::pair<int, float>> ovf;
    device_like(*M, gs, cs, Ng, Nc, tab, ovf, bad_d, rng);
    lgd[m++] = vc_compact_hist(tab.data(), Ng, ovf, ptr_d, val_d, cnt_d);
  }
  ptr_d.push_back((int)val_d.size());
  CHECK(!bad_h && !bad_d);
  CHECK(ptr_h == ptr_d && val_h == val_d && cnt_h == cnt_d);
  CHECK(std::fabs(lgS - lgd[0]) <= 1e-9 * (1.0 + std::fabs(lgS)) && std::fabs(lgU - lgd[1]) <= 1e-9 * (1.0 + std::fabs(lgU)));
  CHECK((int)ptr_h.size() == 2 * Ng + 1);
  // multiplicities add up to the non-zero entries; values strictly increase inside the dense part of a gene
  double total = 0;
  for (float c : cnt_h) total += c;
  double nz = 0;
  for (auto* M : {&S, &U}) for (float v : *M) nz += v != 0.f;
  CHECK(total == nz);
  std::vector<int> task, tptr;
  vc_build_hist_tasks(ptr_h, Ng, task, tptr);
  CHECK((int)tptr.size() == Ng + 1 && tptr.back() == (int)task.size() / 4);
  long long covered = 0;
  for (size_t t = 0; t + 3 < task.size(); t += 4) {
    CHECK(task[t] >= 0 && task[t] < Ng && (task[t + 1] == 0 || task[t + 1] == 1));
    CHECK(task[t + 3] - task[t + 2] >= 1 && task[t + 3] - task[t + 2] <= 64);
    CHECK(task[t + 2] >= ptr_h[(size_t)task[t + 1] * Ng + task[t]] && task[t + 3] <= ptr_h[(size_t)task[t + 1] * Ng + task[t] + 1]);
    covered += task[t + 3] - task[t + 2];
  }
  CHECK(covered == (long long)val_h.size());
}

// The cell tiling of the likelihood kernel: for every gene block the waves' ranges [first, first + cw) clamped to Nc, taken in
// the order (chunk, wave), tile [0, Nc) exactly -- no gap, no overlap -- whatever the shares, the number of gene blocks
// (also when it does not divide the CUs), the occupancy; and the grid never exceeds the resident slots when shares apply.
static void tiling_case(long long Nc, int nGB, int n_cu, int bpc, const double* share, int min_cw, long long cw_override) {
  const int waves = 4;
  const VcTiling t = vc_tile_cells(Nc, nGB, n_cu, bpc, waves, cw_override, share, min_cw);
  CHECK(t.n_chunks >= 1 && t.cw >= 1);
  bool unequal = false;
  for (int p = 1; p < 4; ++p) unequal |= t.pass_cw[p] != t.pass_cw[0];
  if (unequal) CHECK((long long)nGB * t.n_chunks <= (long long)bpc * n_cu);
  for (int p = 0; p < 4; ++p) CHECK(t.pass_cw[p] >= 1 && t.pass_cw[p] <= t.cw);
  for (int gb = 0; gb < nGB; ++gb) {
    long long next = 0;                    // first cell nobody has taken yet
    for (int chunk = 0; chunk < t.n_chunks; ++chunk)
      for (int w = 0; w < waves; ++w) {
        int cw = 0;
        long long b = vc_wave_first_cell(chunk, gb, w, nGB, n_cu, t.pass_cw, waves, &cw);
        long long e = b + cw;
        CHECK(cw >= 1);
        if (b > Nc) b = Nc;
        if (e > Nc) e = Nc;
        if (b < Nc) CHECK(b == next);
        if (e > next) next = e;
      }
    CHECK(next == Nc);
  }
}

// one-hot batches: order by batch, batch-aligned workgroup table -- every position of every gene block is covered exactly once,
// every workgroup lies inside the batch the table names, the batches' chunk ranges are consecutive and complete
static void batch_case(long long Nc, int Nb, int nGB, int n_cu, int bpc, bool shuffled, unsigned seed) {
  std::mt19937 rng(seed);
  std::vector<float> Db((size_t)Nb * Nc, 0.f);
  std::vector<int> want((size_t)Nc);
  for (long long c = 0; c < Nc; ++c) {
    int q = shuffled ? (int)(rng() % Nb) : (int)((c * Nb) / Nc);
    if (Nb > 2 && q == 1) q = 0;                 // an empty batch
    want[(size_t)c] = q;
    Db[(size_t)q * Nc + c] = 1.f;
  }
  std::vector<int> bid, pos, ord, len;
  CHECK(vc_onehot_batches(Db.data(), Nb, Nc, bid));
  CHECK(bid == want);
  const bool sorted = vc_order_by_batch(bid, Nb, pos, ord, len);
  if (!shuffled) CHECK(sorted);
  long long tot = 0;
  for (int q = 0; q < Nb; ++q) tot += len[(size_t)q];
  CHECK(tot == Nc);
  for (long long c = 0; c < Nc; ++c) CHECK(ord[(size_t)pos[(size_t)c]] == (int)c);
  for (long long p = 1; p < Nc; ++p) {
    CHECK(bid[(size_t)ord[(size_t)p]] >= bid[(size_t)ord[(size_t)p - 1]]);
    if (bid[(size_t)ord[(size_t)p]] == bid[(size_t)ord[(size_t)p - 1]]) CHECK(ord[(size_t)p] > ord[(size_t)p - 1]);      // stable
  }
  const double half[4] = {1.0, 0.5, 0.25, 0.125};
  VcTiling t = vc_tile_cells(Nc, nGB, n_cu, bpc, 4, 0, half, 12);
  int nonempty = 0;
  for (int v : len) nonempty += v > 0;
  if (t.n_chunks < nonempty) t.n_chunks = nonempty;
  std::vector<int> tile, bc;
  const int cwm = vc_tile_batches(Nc, nGB, t.n_chunks, n_cu, t.pass_cw, 4, len, tile, bc);
  std::vector<long long> start((size_t)Nb + 1, 0);
  for (int q = 0; q < Nb; ++q) start[(size_t)q + 1] = start[(size_t)q] + len[(size_t)q];
  for (int gb = 0; gb < nGB; ++gb) {
    long long next = 0;
    for (int k = 0; k < t.n_chunks; ++k) {
      const int* w = &tile[4 * ((size_t)k * nGB + gb)];
      const int q = w[2];
      CHECK(q >= 0 && q < Nb && w[1] <= cwm && w[1] >= 0);
      CHECK(k >= bc[(size_t)gb * (Nb + 1) + q] && k < bc[(size_t)gb * (Nb + 1) + q + 1]);
      CHECK(w[0] >= start[(size_t)q] && w[3] <= start[(size_t)q + 1] && w[0] <= w[3]);
      CHECK(w[0] == next || w[0] == w[3]);
      CHECK((long long)w[1] * 4 >= w[3] - w[0]);            // four waves of cw cells reach the end
      if (w[3] > w[0]) next = w[3];
    }
    CHECK(next == Nc);
    CHECK(bc[(size_t)gb * (Nb + 1)] == 0 && bc[(size_t)gb * (Nb + 1) + Nb] == t.n_chunks);
    for (int q = 0; q < Nb; ++q) CHECK(bc[(size_t)gb * (Nb + 1) + q] <= bc[(size_t)gb * (Nb + 1) + q + 1]);
  }
  // not one-hot: a fractional entry, two entries in one column, an empty column
  if (Nc >= 2) {
    std::vector<float> bad = Db;
    bad[0] = bad[0] == 1.f ? 0.5f : bad[0];
    for (int q = 0; q < Nb; ++q) if (bad[(size_t)q * Nc] == 1.f) bad[(size_t)q * Nc] = 0.5f;
    CHECK(!vc_onehot_batches(bad.data(), Nb, Nc, bid));
    bad = Db;
    for (int q = 0; q < Nb; ++q) bad[(size_t)q * Nc + 1] = 0.f;
    CHECK(!vc_onehot_batches(bad.data(), Nb, Nc, bid));
    if (Nb >= 2) { bad = Db; bad[1] = 1.f; bad[(size_t)Nc + 1] = 1.f; CHECK(!vc_onehot_batches(bad.data(), Nb, Nc, bid)); }
  }
}

int main() {
  {
    std::mt19937 rng(5);
    for (int it = 0; it < 120; ++it)
      batch_case(1 + (long long)(rng() % 60000), 1 + (int)(rng() % 13), 1 + (int)(rng() % 8), (it % 4 == 0) ? 1 + (int)(rng() % 300) : 256,
                 1 + (int)(rng() % 4), it % 2 == 1, 100 + it);
    batch_case(50000, 2, 4, 256, 2, false, 1);
    batch_case(50000, 8, 4, 256, 2, true, 2);
    batch_case(23, 2, 1, 256, 2, false, 3);
    batch_case(5, 5, 1, 256, 3, true, 4);
  }
  {
    const double half[4] = {1.0, 0.5, 0.25, 0.125}, flat[4] = {1, 1, 1, 1}, odd[4] = {0.5, 0.3, 0.2, 0.2}, steep[4] = {0.85, 0.15, 0.15, 0.15};
    std::mt19937 rng(11);
    for (int it = 0; it < 400; ++it) {
      const long long Nc = 1 + (long long)(rng() % 200000) * (it % 7 == 0 ? 13 : 1);
      const int nGB = 1 + (int)(rng() % 9), n_cu = (it % 5 == 0) ? 1 + (int)(rng() % 300) : 256, bpc = 1 + (int)(rng() % 5);
      const double* sh[] = {half, flat, odd, steep, nullptr};
      tiling_case(Nc, nGB, n_cu, bpc, sh[rng() % 5], (it % 3 == 0) ? 1 : 12, (it % 11 == 0) ? 1 + (long long)(rng() % 300) : 0);
    }
    tiling_case(50000, 4, 256, 2, half, 12, 0);
    tiling_case(50000, 4, 256, 3, half, 12, 0);
    tiling_case(9000, 3, 256, 2, half, 12, 0);
    tiling_case(1, 1, 256, 2, half, 12, 0);
    const VcTiling t = vc_tile_cells(50000, 4, 256, 2, 4, 0, half, 12);
    CHECK(t.n_chunks == 128 && t.pass_cw[0] == 131 && t.pass_cw[1] == 66);
  }
  one_case(7, 33, false, false, 1);
  one_case(7, 33, true, true, 2);
  one_case(64, 301, false, true, 3);
  one_case(1, 1, true, false, 4);
  one_case(130, 77, true, true, 5);
  // invalid values are flagged, never converted to int (UB) or used as map keys
  {
    std::vector<float> M = {1.f, std::numeric_limits<float>::quiet_NaN(), 3.f, -2.f, std::numeric_limits<float>::infinity(), 1e30f};
    std::vector<int> ptr;
    std::vector<float> val, cnt;
    bool bad = false;
    vc_build_hist_host(M.data(), 3, 1, 2, 3, ptr, val, cnt, &bad);
    CHECK(bad);
    CHECK(!vc_count_ok(M[1]) && !vc_count_ok(M[3]) && !vc_count_ok(M[4]) && vc_count_ok(M[5]) && !vc_count_dense(M[5]));
  }
  if (fails) { std::printf("%d check(s) failed\n", fails); return 1; }
  std::printf("host logic ok\n");
  return 0;
}
