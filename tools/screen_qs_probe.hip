// Diagnostic only (round 4, for round 5): the SCORING CORE of a QUERY-STATIONARY screening kernel.
//
// Today (csrc/search.hip screen_append_kernel): a workgroup holds 160 queries as B fragments in LDS, every one of its 12
// waves streams ITS OWN corpus tiles global -> registers (24 KiB per 32 rows and wave) and reads one query fragment from LDS
// per MFMA.  60 GB from L2 per call at the bench shape; the tile loads are 1.6 ms of the 6.75 ms kernel.
// Here: the QUERIES are stationary in registers - 12 waves x 32 queries = 384 per workgroup, 96 registers each - and the
// corpus tiles pass ONCE per workgroup through LDS by LDS-DMA (no registers, no ds_write); every wave reads one TILE fragment
// from LDS per MFMA (the same LDS read rate as today).  A stage = TPS tiles; NST stages in LDS; one workgroup barrier per
// stage (each wave waits for its own DMA pieces first).  2.4 x fewer L2 bytes per MFMA; the question this probe answers is
// what the barriers and the DMA issue cost.  No pruning pools, no appends: every accumulator is folded into a running
// maximum so that nothing is optimised away.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/screen_qs_probe.hip -o tools/screen_qs_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KSTEPS = 24;            // 384 dims / 16
constexpr int TILE_VEC = KSTEPS * 64; // 16-byte vectors of a 32-row tile (24 KiB)
constexpr int WAVES = 12;
#ifndef QS_TPS
#define QS_TPS 2   // tiles per stage
#endif
#ifndef QS_NST
#define QS_NST 3   // stages in LDS
#endif
#ifndef QS_RING
#define QS_RING 4   // LDS fragment reads in flight per wave + 1
#endif
constexpr int TPS = QS_TPS, NST = QS_NST;
static_assert(TPS * TILE_VEC % (WAVES * 64) == 0, "a stage is dealt evenly to the waves");
constexpr int PIECES = TPS * TILE_VEC / 64 / WAVES;   // 1-KiB DMA pieces per wave and stage

__device__ inline void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(WAVES * 64) void qs_kernel(const bf16x8* __restrict__ tiles, const bf16x8* __restrict__ queries,
                                                       int tiles_per_slice, int n_slices, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bf16x8* const stage = reinterpret_cast<bf16x8*>(lds_raw);   // [NST][TPS][KSTEPS][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int qblock = blockIdx.x / n_slices, slice = blockIdx.x - qblock * n_slices;
  // this wave's 32 queries: B fragments [24][64] of query tile (qblock * 12 + wave)
  bf16x8 qf[KSTEPS];
  {
    const bf16x8* q = queries + (int64_t)(qblock * WAVES + wave) * TILE_VEC + lane;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qf[s] = q[s * 64];
  }
  const bf16x8* src = tiles + (int64_t)slice * tiles_per_slice * TILE_VEC;
  const int n_stages = tiles_per_slice / TPS;
  auto request = [&](int st) {   // this wave's pieces of stage st
    const bf16x8* g = src + (int64_t)st * TPS * TILE_VEC + lane;
    bf16x8* l = stage + (st % NST) * TPS * TILE_VEC;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int piece = wave + WAVES * i;
#ifndef QS_ABL_NODMA
      glds16(g + piece * 64, l + piece * 64);
#endif
    }
  };
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < n_stages) request(st);
  float best = -1e30f;
  for (int st = 0; st < n_stages; ++st) {
    // the pieces of stage st are the OLDEST of this wave's outstanding requests (NST - 1 stages in flight)
    if (st + NST - 2 < n_stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef QS_ABL_NOBARRIER
    asm volatile("s_barrier" ::: "memory");   // stage st complete for everybody; everybody has left stage st - 1
#endif
    if (st + NST - 1 < n_stages) request(st + NST - 1);   // into the buffer stage st - 1 has just left
    const bf16x8* a = stage + (st % NST) * TPS * TILE_VEC + lane;
#ifdef QS_PAIR   // two tiles at a time: two independent accumulator chains, fragments of both through one ring
    static_assert(TPS % 2 == 0, "pairs");
#pragma unroll
    for (int t = 0; t < TPS; t += 2) {
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
      constexpr int R = QS_RING, NS = 2 * KSTEPS;   // slot n = 2 s + u: fragment s of tile t + u
      auto at = [&](int n) { return ((t + (n & 1)) * KSTEPS + (n >> 1)) * 64; };
      bf16x8 ar[R];
#pragma unroll
      for (int i = 0; i < R - 1; ++i) ar[i] = a[at(i)];
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        if (n + R - 1 < NS) ar[(n + R - 1) % R] = a[at(n + R - 1)];
        if (n & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[n % R], qf[n >> 1], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[n % R], qf[n >> 1], acc0, 0, 0, 0);
      }
      float m = fmaxf(acc0[0], acc1[0]);
#pragma unroll
      for (int i = 1; i < 16; ++i) m = fmaxf(m, fmaxf(acc0[i], acc1[i]));
      best = fmaxf(best, m);
    }
#else
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      constexpr int R = QS_RING;
      bf16x8 ar[R];
#pragma unroll
      for (int i = 0; i < R - 1; ++i) ar[i] = a[(t * KSTEPS + i) * 64];
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
#ifndef QS_ABL_NOREAD
        if (s + R - 1 < KSTEPS) ar[(s + R - 1) % R] = a[(t * KSTEPS + s + R - 1) * 64];
#endif
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[s % R], qf[s], acc, 0, 0, 0);
      }
      float m = acc[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[i]);
      best = fmaxf(best, m);   // stands in for the pruning compare (one v_max3 chain per tile)
    }
#endif
  }
  out[(int64_t)blockIdx.x * (WAVES * 64) + threadIdx.x] = best;
}

int main(int argc, char** argv) {
  const int n_rows = 1000000, n_queries = 9984;               // 26 blocks of 384 queries
  const int n_tiles = n_rows / 32 / (TPS * 10) * (TPS * 10);  // 31 240 tiles
  const int q_blocks = n_queries / (WAVES * 32);
  const int n_slices = argc > 1 ? atoi(argv[1]) : 10;         // 26 x 10 = 260 workgroups
  const int tiles_per_slice = n_tiles / n_slices / TPS * TPS;
  std::vector<unsigned short> h((size_t)1 << 22);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + ((i * 2654435761u) >> 23) + ((i & 1) << 15));
  auto dalloc = [&](size_t bytes) {
    void* p = nullptr;
    (void)hipMalloc(&p, bytes);
    for (size_t off = 0; off < bytes; off += h.size() * 2)
      (void)hipMemcpy((char*)p + off, h.data(), std::min(h.size() * 2, bytes - off), hipMemcpyHostToDevice);
    return p;
  };
  const bf16x8* tiles = (const bf16x8*)dalloc((size_t)n_tiles * TILE_VEC * 16);
  const bf16x8* queries = (const bf16x8*)dalloc((size_t)q_blocks * WAVES * TILE_VEC * 16);
  float* out = (float*)dalloc((size_t)q_blocks * n_slices * WAVES * 64 * 4);
  const size_t lds = (size_t)NST * TPS * TILE_VEC * 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const double flops = 2.0 * (double)q_blocks * WAVES * 32 * (double)n_slices * tiles_per_slice * 32 * 384;
  printf("%d query blocks x %d slices = %d workgroups, %d tiles per slice, %d tiles per stage, %d stages in %zu KiB of LDS\n",
         q_blocks, n_slices, q_blocks * n_slices, tiles_per_slice, TPS, NST, lds / 1024);
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(qs_kernel, dim3(q_blocks * n_slices), dim3(WAVES * 64), lds, 0, tiles, queries, tiles_per_slice, n_slices, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("launch %d: %.3f ms, %.0f TFLOP/s = %.3f of 2 500 (%s)\n", rep, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0,
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
