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

int main() {
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
