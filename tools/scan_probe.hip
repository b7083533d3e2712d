// Diagnostic only: runs the scan on random unit vectors and prints slow-path statistics.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DSSKD_PROBE -Iinclude -Isemantic-search-kd_amd/csrc \
//         tools/scan_probe.hip semantic-search-kd_amd/csrc/capi_common.hip -o tools/scan_probe.bin
#include "../semantic-search-kd_amd/csrc/search.hip"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const int nq = 10000, k = 10;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> h((size_t)n * 384);
  for (auto& v : h) v = nd(rng);
  float *rows, *tiled, *q, *os;
  int64_t* oi;
  hipMalloc(&rows, (size_t)n * 384 * 4);
  hipMemcpy(rows, h.data(), (size_t)n * 384 * 4, hipMemcpyHostToDevice);
  hipMalloc(&tiled, sskd_index_tiled_bytes(n));
  sskd_index_add_rows(rows, n, 1, tiled, 0, nullptr);
  hipMalloc(&q, (size_t)nq * 384 * 4);
  for (size_t i = 0; i < (size_t)nq * 384; ++i) h[i] = nd(rng);
  hipMemcpy(q, h.data(), (size_t)nq * 384 * 4, hipMemcpyHostToDevice);
  sskd_l2_normalize_rows(q, nq, 384, nullptr);
  hipMalloc(&os, (size_t)nq * k * 4);
  hipMalloc(&oi, (size_t)nq * k * 8);
  const size_t wsb = sskd_index_search_workspace_bytes(n, nq, k);
  void* ws;
  hipMalloc(&ws, wsb);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_scan_probe), z, sizeof(z));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int rc = sskd_index_search_profiled(tiled, n, q, nq, k, 0, os, oi, ws, wsb, nullptr, a, b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c[8];
    hipMemcpyFromSymbol(c, HIP_SYMBOL(g_scan_probe), sizeof(c));
    printf("rc=%d scan %.2f ms | wave-tiles(x QB) %llu slow-path %llu (%.1f%%) reg-blocks %llu lane-inserts %llu publishes %llu\n",
           rc, ms, c[0], c[1], 100.0 * c[1] / c[0], c[2], c[3], c[4]);
  }
  return 0;
}
