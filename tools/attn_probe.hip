// Diagnostic only: times qkv_attention_kernel on random data and prints the phases of one head.
#include "../semantic-search-kd_amd/csrc/encoder.hip"

#include <cstdio>
#include <vector>

int main() {
  const int B = 512, S = 256, nkt = 8, T = B * S;
  std::vector<unsigned short> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 22);
  auto dalloc = [&](size_t bytes) {
    void* p = nullptr;
    hipMalloc(&p, bytes);
    for (size_t off = 0; off < bytes; off += h.size() * 2)
      hipMemcpy((char*)p + off, h.data(), std::min(h.size() * 2, bytes - off), hipMemcpyHostToDevice);
    return p;
  };
  QkvAttnParams qa{};
  qa.x = (const bf16x8*)dalloc((size_t)T * 384 * 2);
  qa.wqkv = (const bf16x8*)dalloc(1152 * 384 * 2);
  float* f = (float*)dalloc(4096 * 4);
  hipMemset(f, 0, 4096 * 4);
  qa.bqkv = f;
  int* mask;
  hipMalloc(&mask, (size_t)B * S * 4);
  std::vector<int> ones((size_t)B * S, 1);
  hipMemcpy(mask, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
  qa.mask = mask;
  qa.B = B; qa.S = S; qa.nkt = nkt; qa.hpw = 12; qa.q_scale = 0.25f;
  qa.ctx = (__bf16*)dalloc((size_t)T * 384 * 2);
  const size_t lds = 3 * WTILE_VEC * sizeof(bf16x8) + (size_t)nkt * (4 * 128 * sizeof(bf16x8) + 32 * sizeof(float)) + 2 * 96 * sizeof(float);
  auto kern = qkv_attention_kernel<false>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(B), dim3(512), lds, 0, qa);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("launch %d: %.1f us (%s)\n", rep, ms * 1e3, hipGetErrorString(hipGetLastError()));
  }
  unsigned long long pr[16][8];
  hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe_qa), sizeof(pr));
  printf("step | bar0  wwrite  bar1  wload-issue  first-phase  second-phase   step-total\n");
  for (int hi = 2; hi < 8; ++hi)
    printf("%3d  | %5llu %6llu %5llu %8llu %9llu %5llu %9llu\n", hi, pr[hi][1] - pr[hi][0], pr[hi][2] - pr[hi][1],
           pr[hi][3] - pr[hi][2], pr[hi][4] - pr[hi][3], pr[hi][5] - pr[hi][4], pr[hi][6] - pr[hi][5],
           pr[hi + 1][0] - pr[hi][0]);
  return 0;
}
