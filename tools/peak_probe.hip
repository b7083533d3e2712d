// What this box actually delivers, next to the nominal peaks the rooflines are priced against:
//   - device-to-device copy bandwidth (hipMemcpyAsync and a plain float4 copy kernel), 2 GiB buffers
//   - bare bf16 MFMA rate: register operands (random data), 4 independent accumulators per wave, two waves
//     per SIMD, for v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/peak_probe.hip -o tools/ab/peak_probe ; run it on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } \
  } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

__global__ __launch_bounds__(512) void mfma32_kernel(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  const bf16x8 a = src[threadIdx.x], b = src[512 + threadIdx.x];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (s == 12345.678f) sink[blockIdx.x] = s;  // never true: keeps the loop alive
}

__global__ __launch_bounds__(512) void mfma16_kernel(const bf16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  const bf16x8 a = src[threadIdx.x], b = src[512 + threadIdx.x];
  f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c7, 0, 0, 0);
  }
  float s = 0.f;
  for (int e = 0; e < 4; ++e) s += c0[e] + c1[e] + c2[e] + c3[e] + c4[e] + c5[e] + c6[e] + c7[e];
  if (s == 12345.678f) sink[blockIdx.x] = s;
}

template <typename F>
static float time_ms(F&& f, int reps) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  f();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  void *x = nullptr, *y = nullptr;
  CHECK(hipMalloc(&x, bytes));
  CHECK(hipMalloc(&y, bytes));
  CHECK(hipMemset(x, 1, bytes));
  float ms = time_ms([&] { CHECK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
  std::printf("hipMemcpy d2d      : %7.1f GB/s read + the same written (%.2f TB/s of traffic)\n", bytes / ms / 1e6, 2.0 * bytes / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const float4*)x, (float4*)y, bytes / 16); }, 10);
  std::printf("float4 copy kernel : %7.1f GB/s read + the same written (%.2f TB/s of traffic)\n", bytes / ms / 1e6, 2.0 * bytes / ms / 1e9);

  std::vector<unsigned short> h(1024 * 8);
  unsigned int r = 12345u;
  for (auto& v : h) {
    r = r * 1664525u + 1013904223u;
    v = (unsigned short)(0x3C00u + ((r >> 16) & 0x03FFu) + ((r >> 9) & 0x8000u));  // random bf16 of magnitude ~ 0.01
  }
  void* src = nullptr;
  float* sink = nullptr;
  CHECK(hipMalloc(&src, h.size() * 2));
  CHECK(hipMalloc(&sink, 4096 * sizeof(float)));
  CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int iters = 20000, blocks = 256;  // one 8-wave workgroup per CU: two waves per SIMD
  ms = time_ms([&] { hipLaunchKernelGGL(mfma32_kernel, dim3(blocks), dim3(512), 0, 0, (const bf16x8*)src, sink, iters); }, 5);
  double fl = (double)blocks * 8 * iters * 4 * (2.0 * 32 * 32 * 16);
  std::printf("bare 32x32x16 bf16 : %7.1f TFLOP/s (nominal dense peak 2500)\n", fl / ms / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(mfma16_kernel, dim3(blocks), dim3(512), 0, 0, (const bf16x8*)src, sink, iters); }, 5);
  fl = (double)blocks * 8 * iters * 8 * (2.0 * 16 * 16 * 32);
  std::printf("bare 16x16x32 bf16 : %7.1f TFLOP/s\n", fl / ms / 1e9);
  return 0;
}
