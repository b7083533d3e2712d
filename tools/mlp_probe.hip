// Diagnostic only: runs fused_mlp_ln_kernel on random data with s_memtime stamps (SSKD_PROBE)
// and prints where producer / consumer waves of workgroup 0 spend an iteration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSKD_PROBE -Iinclude -Isemantic-search-kd_amd/csrc \
//         tools/mlp_probe.hip semantic-search-kd_amd/csrc/capi_common.hip semantic-search-kd_amd/csrc/pool.hip -o /tmp/mlp_probe
#include "../semantic-search-kd_amd/csrc/encoder.hip"

#include <cstdio>
#include <vector>

int main() {
  const int T = 131072;
  std::vector<unsigned short> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 22);  // ~[0.008, 0.03]
  auto dalloc = [&](size_t bytes) {
    void* p = nullptr;
    hipMalloc(&p, bytes);
    for (size_t off = 0; off < bytes; off += h.size() * 2)
      hipMemcpy((char*)p + off, h.data(), std::min(h.size() * 2, bytes - off), hipMemcpyHostToDevice);
    return p;
  };
  MlpParams m{};
  m.x1 = (const bf16x8*)dalloc((size_t)T * 384 * 2);
  m.w1 = (const bf16x8*)dalloc(1536 * 384 * 2);
  m.w2c = (const bf16x8*)dalloc(1536 * 384 * 2);
  float* f = (float*)dalloc(8192 * 4);
  hipMemset(f, 0, 8192 * 4);
  m.b1 = f; m.b2 = f + 2048; m.gamma = f + 3072; m.beta = f + 4096;
  m.eps = 1e-12f;
  m.out = (__bf16*)dalloc((size_t)T * 384 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(fused_mlp_ln_kernel<false>, dim3(T / 128), dim3(512), 0, 0, m);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("launch %d: %.1f us (%s)\n", rep, ms * 1e3, hipGetErrorString(hipGetLastError()));
  }
  unsigned long long pr[2][64][4];
  hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr));
  printf("it   | producer: stage  compute  barrier | consumer: stage  compute  barrier | iter(P)\n");
  for (int it = 8; it < 20; ++it) {
    printf("%3d  | %8llu %8llu %8llu | %8llu %8llu %8llu | %8llu\n", it,
           pr[0][it][1] - pr[0][it][0], pr[0][it][2] - pr[0][it][1], pr[0][it][3] - pr[0][it][2],
           pr[1][it][1] - pr[1][it][0], pr[1][it][2] - pr[1][it][1], pr[1][it][3] - pr[1][it][2],
           pr[0][it + 1][0] - pr[0][it][0]);
  }
  return 0;
}
