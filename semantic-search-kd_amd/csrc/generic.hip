// Dimension-generic bf16 kernels (row-major activations): NT GEMM on v_mfma_f32_32x32x16_bf16,
// transpose, LayerNorm / softmax / GELU forward + backward, embeddings, mean-pool.  See generic.h.
#include "generic.h"

namespace sskd_generic {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
// erf-GELU of the GEMM epilogues (inference: the teacher's FFN1).  sskd::gelu_erf (common.h): x * sigmoid(cubic in x^2),
// |error| <= 2.6e-5 - 80 x below the bf16 half-ulp of the value produced - in 9 instructions; libm's erff is ~40, all of it
// un-overlapped vector time (128 values per lane and tile while the matrix pipe idles).  -DSSKD_GELU_LIBM: the old form (A/B).
__device__ inline float epilogue_gelu(float x) {
#ifdef SSKD_GELU_LIBM
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
#else
  return sskd::gelu_erf(x);
#endif
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ------------------------------------------------------------------------- //
// NT GEMM: 128 x 128 block tile, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles.
// (A 128 x 256 block tile - wave tile 64 x 128, 6 fragment reads per 8 MFMAs - was built and measured:
// its 66 KiB of LDS leave 2 workgroups per CU instead of 4, and this two-barrier loop lives on
// occupancy: teacher 3 390 -> 2 860 pairs/s, KD step 60 -> 209 ms.  Bigger tiles need the counted-wait
// double-buffered structure first.)
// Operand tiles go global -> registers -> LDS (rows padded by 16 B: conflict-free ds_read_b128),
// the next tile's global loads are in flight while the current one is multiplied.
// ------------------------------------------------------------------------- //
template <int BK>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
  constexpr int LDT = BK + 8;
  constexpr int CPR = BK / 8;
  constexpr int NCH = 128 * CPR / 256;
  constexpr int LDC = 128 + 8;  // staged output tile row (bf16 elements)
  constexpr int LDS_ELEMS = 2 * 128 * LDT > 128 * LDC ? 2 * 128 * LDT : 128 * LDC;
  __shared__ __attribute__((aligned(16))) bf16_t lds[LDS_ELEMS];
  bf16_t* const As = lds;
  bf16_t* const Bs = lds + 128 * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const bool split = p.split_k > 1;
  const int z = blockIdx.z, b1 = split ? 0 : z / p.batch2, b2 = split ? 0 : z - b1 * p.batch2;
  const int nk_all = p.K / BK;
  const int k_per = split ? (nk_all + p.split_k - 1) / p.split_k : nk_all;
  const int kt0 = split ? z * k_per : 0;
  const int nk = split ? max(0, min(k_per, nk_all - kt0)) : nk_all;
  const bf16_t* A = p.A + b1 * p.sA1 + b2 * p.sA2 + (int64_t)kt0 * BK;
  const bf16_t* B = p.B + b1 * p.sB1 + b2 * p.sB2 + (int64_t)kt0 * BK;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  if (nk == 0) return;

  u32x4 ra[NCH], rb[NCH];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + 256 * i, row = idx / CPR, c8 = idx - row * CPR;
      const int gm = min(m0 + row, p.M - 1), gn = min(n0 + row, p.N - 1);
      ra[i] = *reinterpret_cast<const u32x4*>(A + (int64_t)gm * p.lda + kt * BK + c8 * 8);
      rb[i] = *reinterpret_cast<const u32x4*>(B + (int64_t)gn * p.ldb + kt * BK + c8 * 8);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + 256 * i, row = idx / CPR, c8 = idx - row * CPR;
      *reinterpret_cast<u32x4*>(As + row * LDT + c8 * 8) = ra[i];
      *reinterpret_cast<u32x4*>(Bs + row * LDT + c8 * 8) = rb[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    sstore();
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
    const bf16_t* as = As + (wm * 64 + (lane & 31)) * LDT + 8 * (lane >> 5);
    const bf16_t* bs = Bs + (wn * 64 + (lane & 31)) * LDT + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = *reinterpret_cast<const bf16x8*>(as + t * 32 * LDT + ks * 16);
        b[t] = *reinterpret_cast<const bf16x8*>(bs + t * 32 * LDT + ks * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // D[m][n]: lane holds column n = lane & 31, registers hold rows 8g + 4h + e
  const int h = lane >> 5;
  char* Cb = static_cast<char*>(p.C);
  const int64_t cbase = b1 * p.sC1 + b2 * p.sC2;
  if (!p.c_is_f32) {
    // bf16 output: the tile meets in LDS (the operand buffers are dead) and leaves in 16-byte row
    // segments instead of 64 two-byte stores per lane
    bf16_t* const Ct = lds;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nl = wn * 64 + j * 32 + (lane & 31);
        const int n = n0 + nl;
        const float bias = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e)
          {
            float v = p.alpha * acc[i][j][4 * g + e] + bias;
            if (p.act == 1) v = epilogue_gelu(v);
            Ct[(wm * 64 + i * 32 + 8 * g + 4 * h + e) * LDC + nl] = (bf16_t)v;
          }
      }
    __syncthreads();
    bf16_t* Cg = reinterpret_cast<bf16_t*>(Cb) + cbase;
    const bool vec_ok = (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(Cg) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, row = idx >> 4, c8 = (idx & 15) * 8;
      const int m = m0 + row, n = n0 + c8;
      if (m >= p.M || n >= p.N) continue;
      if (vec_ok && n + 8 <= p.N) {
        *reinterpret_cast<u32x4*>(Cg + (int64_t)m * p.ldc + n) = *reinterpret_cast<const u32x4*>(Ct + row * LDC + c8);
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) Cg[(int64_t)m * p.ldc + n + e] = Ct[row * LDC + c8 + e];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + (lane & 31);
      if (n >= p.N) continue;
      const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * 64 + i * 32 + 8 * g + 4 * h + e;
          if (m >= p.M) continue;
          const float v = p.alpha * acc[i][j][4 * g + e] + bias;
          const int64_t off = cbase + (int64_t)m * p.ldc + n;
          if (split) {
            atomicAdd(reinterpret_cast<float*>(Cb) + off, v);
          } else if (p.c_is_f32) {
            float* c = reinterpret_cast<float*>(Cb) + off;
            *c = p.accumulate ? *c + v : v;
          } else {
            reinterpret_cast<bf16_t*>(Cb)[off] = (bf16_t)v;
          }
        }
    }
}

// ------------------------------------------------------------------------- //
// NT GEMM, 256 x 256 block tile (large unbatched products: M, N multiples of 256, K of 64, bf16 out).
// 8 waves as 2 (M) x 4 (N), each 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator
// registers).  Operand tiles go global -> LDS directly (global_load_lds, 16 B per lane, no staging
// registers); two 64 KiB stages; ONE barrier per 64-deep K tile: wait for tile t, barrier, request tile
// t + 1 into the other stage, multiply tile t.  The LDS image of a tile is row-linear (a wave's DMA
// writes 1 KiB = 8 rows of 128 B); bank conflicts are avoided by swizzling on the SOURCE side: LDS chunk c
// of row r holds global chunk c ^ (r & 7), and the fragment reads apply the same XOR.
// One workgroup per CU (128 KiB of LDS): against the 128 x 128 kernel above, 4x the flops per operand byte
// staged and half the barriers per flop.
// ------------------------------------------------------------------------- //
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int GEMM256_CUS = 256;  // persistent grid: one workgroup per CU of an MI355X

__device__ inline void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int BN>   // 256; 192 for the 384-wide student's outputs (384, 1152: 128 x 48 per wave); 128 for other multiples of 128
__global__ __launch_bounds__(512) void gemm_nt256_kernel(GemmArgs p) {
  constexpr int BK = 64;
  constexpr int NREP = BN / 64;              // 16-column MFMA tiles per wave (4 waves across N)
  constexpr int STAGE = (256 + BN) * BK;     // bf16 elements of one stage: A tile then B tile
  constexpr int WN = BN / 4;                 // output columns per wave
  constexpr int LDW = WN + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char g256_lds[];
  bf16_t* const lds = reinterpret_cast<bf16_t*>(g256_lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  bf16_t* const wt = lds + 2 * STAGE + wave * (16 * LDW);  // epilogue slice of this wave, outside the stages
  const int tiles_n = p.N / BN;
  const int total = tiles_n * (p.M / 256);
  const int nk = p.K / BK;
  // PERSISTENT workgroups (one per CU): workgroup b multiplies output tiles b, b + grid, ...; the first K tile of
  // the next output tile is requested before the current one's epilogue, so the DMA pipeline never restarts
  // (short K: 384 is 6 K tiles per output tile).
  // Tile order: workgroups are dealt to the 8 XCDs round-robin (grid is a multiple of 8 or the whole problem).  For
  // narrow outputs (<= 12 tiles across N) an XCD gets a contiguous run of tiles (N fastest), so the A row panel and the
  // B column panels it re-reads stay in ITS L2 (bijective): +4..40 % on the 384- to 3072-wide products; wide outputs
  // (8192: 32 tiles across) lose 7 % to it and keep the dealt order.
  auto coords = [&](int v, int& m0, int& n0) {
    const int xcd = v & 7;
    const int t = tiles_n <= 12 ? xcd * (total >> 3) + min(xcd, total & 7) + (v >> 3) : v;
    m0 = (t / tiles_n) * 256;
    n0 = (t % tiles_n) * BN;
  };
  // DMA geometry: instruction i of wave w fills LDS bytes [(8 i + w) KiB, + 1 KiB) of a tile = rows 8 (8 i + w) .. + 7
  const int drow = lane >> 3, dchunk = (lane & 7) ^ drow;  // (8 (8 i + w) + drow) & 7 == drow
  auto request = [&](int stage, int m0, int n0, int kt) {
    bf16_t* const sa = lds + stage * STAGE;
    bf16_t* const sb = sa + 256 * BK;
    const bf16_t* ga = p.A + (int64_t)(m0 + 8 * wave + drow) * p.lda + dchunk * 8 + kt * BK;
    const bf16_t* gb = p.B + (int64_t)(n0 + 8 * wave + drow) * p.ldb + dchunk * 8 + kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(ga + (int64_t)(64 * i) * p.lda, sa + (8 * i + wave) * 512);
#pragma unroll
    for (int i = 0; i < NREP; ++i) glds16(gb + (int64_t)(64 * i) * p.ldb, sb + (8 * i + wave) * 512);
  };

  f32x4 acc[8][NREP];
  // fragment reads: lane -> row (lane & 15) of a 16-row block, k chunk 4 kk + (lane >> 4), XOR-swizzled by row & 7
  const int fr = lane & 15, fq = lane >> 4;
  bf16_t* const Cg = static_cast<bf16_t*>(p.C);
  int v = blockIdx.x, m0, n0, step = 0;
  if (v >= total) return;
  coords(v, m0, n0);
  request(0, m0, n0, 0);
  for (; v < total; v += gridDim.x) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int m1 = 0, n1 = 0;
    const bool more_tiles = v + (int)gridDim.x < total;
    if (more_tiles) coords(v + gridDim.x, m1, n1);
    for (int kt = 0; kt < nk; ++kt, ++step) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the current K tile have landed
      __syncthreads();                     // everybody's have; everybody is done with the other stage
      if (kt + 1 < nk) request((step + 1) & 1, m0, n0, kt + 1);
      else if (more_tiles) request((step + 1) & 1, m1, n1, 0);
      const bf16_t* const sa = lds + (step & 1) * STAGE + (wr * 128) * BK;
      const bf16_t* const sb = lds + (step & 1) * STAGE + 256 * BK + (wc * WN) * BK;
#ifndef SSKD_GEMM256_FRAGS_PER_STEP
      // both k-steps' fragments requested up front (the second step's reads land behind the first step's MFMAs) and the
      // MFMA block at raised priority: +5..10 % on deep-K shapes (8192^3: 1 287 -> 1 359 TFLOP/s), -2 % on K = 384
      bf16x8 a[2][8], b[2][NREP];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = i * 16 + fr;
          a[kk][i] = *reinterpret_cast<const bf16x8*>(sa + row * BK + (((4 * kk + fq) ^ (row & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
          const int row = j * 16 + fr;
          b[kk][j] = *reinterpret_cast<const bf16x8*>(sb + row * BK + (((4 * kk + fq) ^ (row & 7)) << 3));
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < NREP; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);   // C^T tile: see the epilogue
      __builtin_amdgcn_s_setprio(0);
#else
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[8], b[NREP];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = i * 16 + fr;
          a[i] = *reinterpret_cast<const bf16x8*>(sa + row * BK + (((4 * kk + fq) ^ (row & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
          const int row = j * 16 + fr;
          b[j] = *reinterpret_cast<const bf16x8*>(sb + row * BK + (((4 * kk + fq) ^ (row & 7)) << 3));
        }
#ifdef SSKD_GEMM256_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < NREP; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
#ifdef SSKD_GEMM256_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      }
#endif
    }
#ifdef SSKD_GEMM256_ABL_NOEPI   // timing ablation (tools/gemm_probe.py): no epilogue
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j) asm volatile("" ::"v"(acc[i][j]));
    m0 = m1;
    n0 = n1;
    continue;
#endif
    // epilogue.  The MFMAs above take the B fragments AS THE A OPERAND (they compute the transposed 16 x 16 tile), so
    // acc[i][j][e] = C[m0 + 128 wr + 16 i + fr][n0 + WN wc + 16 j + 4 fq + e]: a lane holds FOUR CONSECUTIVE COLUMNS of one
    // row - one 8-byte LDS store per tile instead of four 2-byte ones (+3..9 % on the K = 384 products).  Each wave turns
    // 16 rows at a time through its own slice of LDS into 16-byte row segments (the next tile's DMA is already in
    // flight).  Measured and NOT kept (tools/gemm_probe.py, gpurun_out/r03_gemm3.log): the same epilogue straight from
    // registers (v_permlane32_swap pairs -> 16-byte stores, no LDS, no waits) is no faster - the epilogue is 28-33 % of
    // the K = 384 products and 16-20 % of the K = 1024 ones (-DSSKD_GEMM256_ABL_NOEPI) because the matrix pipe idles
    // while 128 accumulators per lane are biased, converted and stored, not because of the LDS round trip: hiding it
    // needs a second accumulator set or a second workgroup per CU, i.e. another tile shape.
    f32x4 bias4[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j)
      bias4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + wc * WN + j * 16 + 4 * fq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = p.alpha * acc[i][j][e] + bias4[j][e];
          if (p.act == 1) x = epilogue_gelu(x);
          o[e] = (bf16_t)x;
        }
        *reinterpret_cast<bf16x4*>(wt + fr * LDW + j * 16 + 4 * fq) = o;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS writes (wave-private slice)
      __builtin_amdgcn_wave_barrier();
      constexpr int CPR = WN / 8;          // 16-byte segments per row: 8, 6 or 4
#pragma unroll
      for (int h2 = 0; h2 < (16 * CPR + 63) / 64; ++h2) {
        const int seg = lane + 64 * h2, row = seg / CPR, c8 = (seg % CPR) * 8;
        if ((16 * CPR) % 64 != 0 && seg >= 16 * CPR) continue;   // (192-wide tiles: 96 segments)
        const u32x4 val = *reinterpret_cast<const u32x4*>(wt + row * LDW + c8);
        *reinterpret_cast<u32x4*>(Cg + (int64_t)(m0 + wr * 128 + i * 16 + row) * p.ldc + n0 + wc * WN + c8) = val;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
    }
    m0 = m1;
    n0 = n1;
  }
}

// ------------------------------------------------------------------------- //
// TN product for weight gradients: C[M, N] += A[T, M]^T B[T, N] (fp32 atomics), 384 x 128 output tile.
// Both operands are row-major with the REDUCTION index (token) as the row, so a K tile of 64 tokens is DMA'd
// as it lies in memory ([token][feature], global_load_lds) and the MFMA fragments - 8 consecutive tokens of one
// feature per lane - are gathered by ds_read_b64_tr_b16, gfx950's transposing LDS read (a 16-lane group reads
// 4 token rows x 16 feature columns and each lane receives one column).  That removes the two operand transposes
// per product the NT kernel needs.  LDS chunk c of token row r holds global chunk c ^ swz(r),
// swz(r) = 2 (r & 3) ^ 8 ((r >> 3) & 1): the 4 rows of a read and the two groups of a 32-lane half land on
// different banks.  8 waves as 4 (M) x 2 (N), each 96 x 64 = 6 x 4 tiles of v_mfma_f32_16x16x32_bf16.
// grid = output tiles x K slices, one-dimensional (the order is XCD-aware: see the kernel).
// ------------------------------------------------------------------------- //
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct GemmTnArgs {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  int64_t lda, ldb, ldc, T;
  int M, N;
  int k_per;   // K tiles (of 64 tokens) per slice
  int xcd_map; // workgroup order: all tiles of a K slice on one XCD (see the kernel)
};

__device__ inline int tn_swz(int row) { return (2 * (row & 3)) ^ (8 * ((row >> 3) & 1)); }

__global__ __launch_bounds__(512) void gemm_tn384_kernel(GemmTnArgs p) {
  constexpr int BK = 64, TM = 384, TN = 128;
  constexpr int CA = TM / 8, CB = TN / 8;        // 16-byte chunks per token row: 48, 16
  constexpr int STAGE = (TM + TN) * BK;          // bf16 elements per stage (64 KiB)
  extern __shared__ __attribute__((aligned(16))) unsigned char gtn_lds[];
  bf16_t* const lds = reinterpret_cast<bf16_t*>(gtn_lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;        // 4 (M) x 2 (N) waves, each 96 x 64 = 6 x 4 MFMA tiles: 20 KiB of fragment
  constexpr int NI = 6, NJ = 4;                   // reads per wave and K tile (2 x 4 waves of 192 x 32 read 28 KiB and were LDS-bound)
  constexpr int WMF = NI * 16, WNF = NJ * 16;     // features per wave along M and N
  const int tiles_n = p.N / TN;
  const int tiles = tiles_n * (p.M / TM);
  // Workgroup -> (K slice, output tile).  xcd_map: the hardware deals consecutive workgroup ids to the 8 XCDs round-robin;
  // ALL tiles of a K slice go to ONE XCD (id = 8 j + xcd: slice = xcd + 8 (j / tiles), tile = j % tiles), so the token
  // rows the tiles of a slice share (every tile of an M row reads the same A panel, every tile of an N column the same
  // B panel) come out of that XCD's L2 once.  Otherwise: tile-major (id = tile + tiles * slice), any slice count.
  int slice, tile;
  if (p.xcd_map) {
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    slice = xcd + 8 * (jj / tiles);
    tile = jj % tiles;
  } else {
    slice = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
  const int nk_all = (int)(p.T / BK);
  const int kt0 = slice * p.k_per;
  const int nk = min(p.k_per, nk_all - kt0);
  if (nk <= 0) return;

  // DMA: the image of a K tile is [64 tokens][chunks] linear; wave-instruction (8 i + w) writes chunks 64 (8 i + w) .. + 63
  int a_off[6], b_off[2];   // element offsets of this lane's source chunk, relative to the K tile's first token row
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int L = (8 * i + wave) * 64 + lane, row = L / CA, c = L - row * CA;
    a_off[i] = row * (int)p.lda + m0 + ((c ^ tn_swz(row)) << 3);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int L = (8 * i + wave) * 64 + lane, row = L / CB, c = L - row * CB;
    b_off[i] = row * (int)p.ldb + n0 + ((c ^ tn_swz(row)) << 3);
  }
  auto request = [&](int stage, int kt) {
    bf16_t* const sa = lds + stage * STAGE;
    bf16_t* const sb = sa + TM * BK;
    const bf16_t* ga = p.A + (int64_t)(kt0 + kt) * BK * p.lda;
    const bf16_t* gb = p.B + (int64_t)(kt0 + kt) * BK * p.ldb;
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16(ga + a_off[i], sa + (8 * i + wave) * 512);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(gb + b_off[i], sb + (8 * i + wave) * 512);
  };

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed fragment reads: lane 4 q + pp of a 16-lane group supplies the address of token row q, columns 4 pp .. + 3
  const int fr = lane & 15, fq = lane >> 4, q = fr >> 2, pp = fr & 3;
  request(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (kt + 1 < nk) request((kt + 1) & 1, kt + 1);
    const bf16_t* const sa = lds + (kt & 1) * STAGE;
    const bf16_t* const sb = sa + TM * BK;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      s16x8 a[NI], b[NJ];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int row = 32 * kk + 8 * fq + 4 * hh + q, sw = tn_swz(row);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int chunk = wr * (WMF / 8) + 2 * i + (pp >> 1);
          const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sa + row * TM + ((chunk ^ sw) << 3) + 4 * (pp & 1)));
          a[i][4 * hh + 0] = v[0]; a[i][4 * hh + 1] = v[1]; a[i][4 * hh + 2] = v[2]; a[i][4 * hh + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int chunk = wc * (WNF / 8) + 2 * j + (pp >> 1);
          const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(sb + row * TN + ((chunk ^ sw) << 3) + 4 * (pp & 1)));
          b[j][4 * hh + 0] = v[0]; b[j][4 * hh + 1] = v[1]; b[j][4 * hh + 2] = v[2]; b[j][4 * hh + 3] = v[3];
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                              acc[i][j], 0, 0, 0);
    }
  }
  // acc[i][j][e] = C[m0 + WMF wr + 16 i + 4 fq + e][n0 + WNF wc + 16 j + fr]
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        atomicAdd(p.C + (int64_t)(m0 + wr * WMF + i * 16 + 4 * fq + e) * p.ldc + n0 + wc * WNF + j * 16 + fr, acc[i][j][e]);
}

// ------------------------------------------------------------------------- //
// Fused inference attention (teacher cross-encoder): one workgroup = one (batch row, head) and up to 8
// query tiles of 32 (one per wave); K and V of the head are staged once in LDS in MFMA fragment order.
//   S^T = K Q^T      A = K fragments [key, dim] (16 contiguous bytes of a qkv row), B = Q^T fragments
//                    (registers, 16 contiguous bytes of a qkv row): accumulators [key, query], the
//                    QUERY on the lane, so the online softmax is lane-local (+ one exchange with lane ^ 32)
//   O^T = V^T P^T    B = P^T: the score accumulators, exponentiated and packed, ARE the B operand
//                    (k index 8h + e' <-> key 16 s2 + 8 (e' >> 2) + 4h + (e' & 3)); A = V^T fragments
//                    [dim, key] in that same key order.  V is staged ROW-MAJOR ([key][dim], 16-byte stores) and
//                    transposed by the READ: ds_read_b64_tr_b16 hands a lane one dim column of four key rows,
//                    and the four rows are whatever the lanes' addresses say - here keys base + 0..3 of the packed
//                    order, two reads per fragment.  (Rounds 1-2 transposed while staging, with 8 two-byte LDS
//                    stores per 16 bytes loaded: 8-way bank conflicts, about a third of the kernel.)
// Same operand trick as the hidden-384 inference kernel (encoder.hip), with operands from memory.
// ------------------------------------------------------------------------- //
constexpr float ATT_NEG = -1.0e30f;

struct AttnArgs {
  const bf16_t* qkv;
  const int32_t* mask;
  bf16_t* ctx;
  float* lse;   // optional [B, NH, S]
  int B, S, NH, H;
  int nkt;      // S / 32
  int qsplit;   // workgroups per (row, head): ceil(nkt / 8)
  float scale2; // scale * log2(e)
};

// exchange with lane ^ 32 in the VALU (v_permlane32_swap) instead of __shfl_xor's ds_bpermute: that is an LDS round trip
// in the middle of the per-key-tile dependency chain (score MFMA -> max -> exchange -> exp -> PV MFMA)
__device__ inline float pair_max32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float pair_sum32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Row-major [key][DH] bf16 image of a head's V in LDS, read TRANSPOSED.  Byte offset of 16-byte chunk c of key k:
// k * 2 DH + (16 c ^ att_vswz(k)).  The XOR moves whole 64-byte groups of a row so that the four key rows of one
// transposing read (consecutive keys, same dims) fall into different banks: rows are 64 / 128 / 256 bytes for head
// widths 32 / 64 / 128 and a ds_read_b64_tr_b16 half-wave reads 64 contiguous bytes of each of its 4 rows (bank =
// (address / 4) % 64): at 64-byte rows the four rows already tile the 64 banks; at 128 bytes rows k and k + 2 collide
// (swap the row's two 64-byte halves when bit 1 of k is set); at 256 bytes all four collide (rotate by k & 3).
template <int DH>
__device__ inline int att_vswz(int key) {
  return DH == 128 ? (key & 3) << 6 : DH == 64 ? ((key >> 1) & 1) << 6 : 0;
}

// A-operand fragment of O^T = V^T P^T for key tile kt, dim tile t, 16-key half s2: lane l holds V[key][dim 32 t + (l & 31)]
// for the 8 keys 32 kt + 16 s2 + 8 (e' >> 2) + 4 (l >> 5) + (e' & 3), e' = 0 .. 7 - two transposing reads of 4 keys each.
// Lane 4 q + pidx of a 16-lane group supplies the address of key row q, dims 4 pidx .. + 3 of the group's 16 dims.
// EXEC must be all ones (whole waves only call this).
template <int DH>
__device__ inline bf16x8 att_vt_frag(const bf16_t* vl, int kt, int t, int s2, int lane) {
  const int grp = lane >> 4, q = (lane & 15) >> 2, pidx = lane & 3, h = lane >> 5;
  const int dim0 = 32 * t + 16 * (grp & 1) + 4 * pidx;       // first of this lane's 4 address dims
  const unsigned char* base = reinterpret_cast<const unsigned char*>(vl);
  s16x8 out;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const int key = 32 * kt + 16 * s2 + 8 * which + 4 * h + q;
    const int off = key * (DH * 2) + (((dim0 >> 3) << 4) ^ att_vswz<DH>(key)) + 2 * (dim0 & 7);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
    out[4 * which + 0] = v[0]; out[4 * which + 1] = v[1]; out[4 * which + 2] = v[2]; out[4 * which + 3] = v[3];
  }
  return __builtin_bit_cast(bf16x8, out);
}

template <int DH>
__global__ __launch_bounds__(512) void attention_fwd_kernel(AttnArgs p) {
  constexpr int KS = DH / 16;   // k-steps of a score tile
  constexpr int DT = DH / 32;   // 32-wide tiles of the head dimension
  extern __shared__ __attribute__((aligned(16))) unsigned char att_lds[];
  bf16x8* const kl = reinterpret_cast<bf16x8*>(att_lds);                        // [nkt][KS][64]
  bf16_t* const vl = reinterpret_cast<bf16_t*>(kl + (size_t)p.nkt * KS * 64);   // [nkt][DT][2][64][8]
  float* const mb = reinterpret_cast<float*>(vl + (size_t)p.S * DH);            // [S] additive key bias
  __shared__ int s_kmax;
  __shared__ int s_partial[16];   // key tile holds a masked key (S <= 512)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int qs = blockIdx.x % p.qsplit;
  const int bh = blockIdx.x / p.qsplit;
  const int head = bh % p.NH, b = bh / p.NH;
  const int64_t row0 = (int64_t)b * p.S;
  const int64_t ld = 3 * (int64_t)p.H;
  const bf16_t* base = p.qkv + row0 * ld + (int64_t)head * DH;

  if (tid == 0) s_kmax = 0;
  if (tid < 16) s_partial[tid] = 0;
  __syncthreads();
  for (int i = tid; i < p.S; i += 512) {
    const bool on = p.mask[row0 + i] != 0;
    mb[i] = on ? 0.f : ATT_NEG;
    if (on) atomicMax(&s_kmax, i / 32 + 1);
    else s_partial[i >> 5] = 1;
  }
  // K: 16-byte pieces straight into fragment order
  constexpr int CPK = DH / 8;  // 8-element chunks per key
  for (int v = tid; v < p.S * CPK; v += 512) {
    const int key = v / CPK, c = v - key * CPK;
    const bf16x8 kv = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + p.H + 8 * c);
    kl[((key >> 5) * KS + (c >> 1)) * 64 + (key & 31) + 32 * (c & 1)] = kv;
    // V: the same piece of the V block, row-major [key][dim] (swizzled against the transposing reads' bank conflicts)
    const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + 2 * p.H + 8 * c);
    *reinterpret_cast<bf16x8*>(reinterpret_cast<unsigned char*>(vl) + key * (DH * 2) + ((16 * c) ^ att_vswz<DH>(key))) = vv;
  }
  const int qt = qs * 8 + wave;
  const bool active = qt < p.nkt;
  bf16x8 qf[KS];
  {
    const bf16_t* qrow = base + (int64_t)((active ? qt : 0) * 32 + j) * ld + 8 * h;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
  }
  __syncthreads();
  if (!active) return;
  const int kmax = s_kmax;

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m = ATT_NEG, l = 0.f;
  for (int kt = 0; kt < kmax; ++kt) {
    f32x16 sc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[(kt * KS + s) * 64 + lane], qf[s], sc, 0, 0, 0);
    // sc[4g + e] = score(key 32 kt + 8g + 4h + e, query = lane & 31)
    // Key tiles without a masked key (all of them on full-length rows) skip the bias: the maximum is taken on the RAW
    // scores (scale2 > 0 is monotone) and the scale rides in the exponent's fma, exp2(sc scale2 - m): 16 multiplies and
    // four LDS reads fewer per tile and lane.  m, l and the saved log-sum-exp stay in scaled (log2) units.
    float mt = ATT_NEG;
    const bool clean = s_partial[kt] == 0;   // workgroup-uniform
    if (clean) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        mt = fmaxf(fmaxf(fmaxf(mt, sc[4 * g + 0]), fmaxf(sc[4 * g + 1], sc[4 * g + 2])), sc[4 * g + 3]);
      mt *= p.scale2;
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bias = *reinterpret_cast<const float4*>(&mb[kt * 32 + 8 * g + 4 * h]);
        sc[4 * g + 0] = fmaf(sc[4 * g + 0], p.scale2, bias.x);
        sc[4 * g + 1] = fmaf(sc[4 * g + 1], p.scale2, bias.y);
        sc[4 * g + 2] = fmaf(sc[4 * g + 2], p.scale2, bias.z);
        sc[4 * g + 3] = fmaf(sc[4 * g + 3], p.scale2, bias.w);
        mt = fmaxf(fmaxf(fmaxf(mt, sc[4 * g + 0]), fmaxf(sc[4 * g + 1], sc[4 * g + 2])), sc[4 * g + 3]);
      }
    }
    mt = pair_max32(mt);
    if (__any(mt > m)) {
      const float m_new = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
      m = m_new;
    }
    bf16x8 pf[2];
    float ps = 0.f;
    const float es = clean ? p.scale2 : 1.0f;   // scores still raw on clean tiles
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = __builtin_amdgcn_exp2f(fmaf(sc[i], es, -m));
      ps += e;
      pf[i >> 3][i & 7] = (bf16_t)e;
    }
    l += ps;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(vl, kt, t, 0, lane), pf[0], o[t], 0, 0, 0);
      o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(vl, kt, t, 1, lane), pf[1], o[t], 0, 0, 0);
    }
  }
  l = pair_sum32(l);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (p.lse && h == 0) p.lse[((int64_t)b * p.NH + head) * p.S + qt * 32 + j] = m + log2f(l);
  // o[t][4g + e] = O[query = lane & 31][dim 32t + 8g + 4h + e]: 4 consecutive dims = one 8-byte store
  bf16_t* out = p.ctx + (row0 + qt * 32 + j) * (int64_t)p.H + (int64_t)head * DH;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(o[t][4 * g + e] * inv);
      *reinterpret_cast<bf16x4*>(out + 32 * t + 8 * g + 4 * h) = w;
    }
}

// ------------------------------------------------------------------------- //
// Fused attention backward.  One workgroup = one (batch row, head), S <= 256 (8 tiles, one per wave).
// The probabilities are rebuilt from Q, K and the saved log-sum-exp, in two passes so that every
// accumulation is wave-local (no atomics):
//   pass A  wave = QUERY tile, loop over key tiles (the forward's geometry: query on the lane)
//     S^T  = K Q^T, dP^T = V dO^T           A = row fragments of K / V (LDS), B = Q^T / dO^T (registers)
//     dS^T = P^T o (dP^T - D) scale         lse and D = rowsum(dO o O) are lane-local
//     dQ^T += K^T dS^T                      B = the packed dS^T accumulators, A = K^T in their key order
//   pass B  wave = KEY tile, loop over query tiles (key on the lane)
//     S = Q K^T, dP = dO V^T                A = row fragments of Q / dO (LDS), B = K^T / V^T (registers)
//     dV^T += dO^T P, dK^T += Q^T dS        B = packed P / dS accumulators, A = dO^T / Q^T in their query order
// ------------------------------------------------------------------------- //
struct AttnBwdArgs {
  const bf16_t* qkv;
  const int32_t* mask;
  const bf16_t* ctx;
  const bf16_t* dctx;
  const float* lse;
  bf16_t* dqkv;
  int B, S, NH, H, nkt;
  float scale, scale2;
};

// row-major [rows][DH] block of a head -> MFMA row fragments (A operand rows = block rows, B operand columns = block rows)
template <int DH>
__device__ inline void stage_row_frags(bf16x8* dst, const bf16_t* src, int64_t ld, int S, int tid) {
  constexpr int CPK = DH / 8, KS = DH / 16;
  for (int v = tid; v < S * CPK; v += 512) {
    const int row = v / CPK, c = v - row * CPK;
    dst[((row >> 5) * KS + (c >> 1)) * 64 + (row & 31) + 32 * (c & 1)] =
        *reinterpret_cast<const bf16x8*>(src + (int64_t)row * ld + 8 * c);
  }
}
// the same block ROW-MAJOR ([row][DH], swizzled like the forward's V): the transposed A-operand fragments [dim][row] of
// K, Q and dO - rows of a 32-tile in the order the packed accumulators present them - are gathered by the READ
// (att_vt_frag: ds_read_b64_tr_b16).  Rounds 2-3 staged them transposed with eight 2-byte LDS stores per 16 bytes
// loaded: 45 % of this kernel's LDS cycles were bank conflicts of those stores.
template <int DH>
__device__ inline void stage_row_major(bf16_t* dst, const bf16_t* src, int64_t ld, int S, int tid) {
  constexpr int CPK = DH / 8;
  for (int v = tid; v < S * CPK; v += 512) {
    const int row = v / CPK, c = v - row * CPK;
    *reinterpret_cast<bf16x8*>(reinterpret_cast<unsigned char*>(dst) + row * (DH * 2) + ((16 * c) ^ att_vswz<DH>(row))) =
        *reinterpret_cast<const bf16x8*>(src + (int64_t)row * ld + 8 * c);
  }
}

// PASS 0 = pass A (dQ: needs K and V as row fragments, K row-major: 3 blocks of LDS), PASS 1 = pass B (dK, dV: Q and dO
// as row fragments and row-major: 4 blocks).  One kernel doing both held all 7 blocks (112 KiB at S = 256, DH = 32: ONE
// workgroup per CU, and each workgroup is a chain of short dependent MFMA groups); split, three and two fit.
template <int DH, int PASS>
__global__ __launch_bounds__(512) void attention_bwd_kernel(AttnBwdArgs p) {
  constexpr int KS = DH / 16, DT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_lds[];
  const int SD8 = p.S * DH / 8;  // bf16x8 vectors of one [S][DH] block
  bf16x8* const blk0 = reinterpret_cast<bf16x8*>(att_lds);
  bf16x8* const kA = blk0;                                 // pass A: kA, vA, kT      pass B: qA, doA, qT, doT
  bf16x8* const vA = blk0 + SD8;
  bf16x8* const kT = blk0 + 2 * SD8;
  bf16x8* const qA = blk0;
  bf16x8* const doA = blk0 + SD8;
  bf16x8* const qT = blk0 + 2 * SD8;
  bf16x8* const doT = blk0 + 3 * SD8;
  float* const lse = reinterpret_cast<float*>(blk0 + (PASS == 0 ? 3 : 4) * SD8);
  float* const Dq = lse + p.S;
  float* const mb = Dq + p.S;
  __shared__ int s_kmax;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int head = blockIdx.x % p.NH, b = blockIdx.x / p.NH;
  const int64_t row0 = (int64_t)b * p.S;
  const int64_t ld = 3 * (int64_t)p.H;
  const bf16_t* qbase = p.qkv + row0 * ld + (int64_t)head * DH;
  const bf16_t* obase = p.ctx + row0 * p.H + (int64_t)head * DH;
  const bf16_t* dobase = p.dctx + row0 * p.H + (int64_t)head * DH;

  if (tid == 0) s_kmax = 0;
  __syncthreads();
  for (int i = tid; i < p.S; i += 512) {
    const bool on = p.mask[row0 + i] != 0;
    mb[i] = on ? 0.f : ATT_NEG;
    if (on) atomicMax(&s_kmax, i / 32 + 1);
    lse[i] = p.lse[((int64_t)b * p.NH + head) * p.S + i];
    float dsum = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
      const bf16x8 o = *reinterpret_cast<const bf16x8*>(obase + (int64_t)i * p.H + 8 * c);
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(dobase + (int64_t)i * p.H + 8 * c);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum = fmaf((float)o[e], (float)g[e], dsum);
    }
    Dq[i] = dsum;
  }
  const bf16_t* const kR = reinterpret_cast<const bf16_t*>(kT);
  const bf16_t* const qR = reinterpret_cast<const bf16_t*>(qT);
  const bf16_t* const doR = reinterpret_cast<const bf16_t*>(doT);
  if constexpr (PASS == 0) {
    stage_row_frags<DH>(kA, qbase + p.H, ld, p.S, tid);
    stage_row_frags<DH>(vA, qbase + 2 * p.H, ld, p.S, tid);
    stage_row_major<DH>(reinterpret_cast<bf16_t*>(kT), qbase + p.H, ld, p.S, tid);
  } else {
    stage_row_frags<DH>(qA, qbase, ld, p.S, tid);
    stage_row_frags<DH>(doA, dobase, p.H, p.S, tid);
    stage_row_major<DH>(reinterpret_cast<bf16_t*>(qT), qbase, ld, p.S, tid);
    stage_row_major<DH>(reinterpret_cast<bf16_t*>(doT), dobase, p.H, p.S, tid);
  }
  __syncthreads();
  if (wave >= p.nkt) return;
  const int kmax = s_kmax;
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16_t* const drow = p.dqkv + (row0 + wave * 32 + j) * ld + (int64_t)head * DH;  // this lane's row of dQ | dK | dV

  // ---- pass A: dQ of query tile `wave` ----
  if constexpr (PASS == 0) {
    bf16x8 qB[KS], doB[KS];
    {   // B operand of a row block = its row fragment: row j of the tile, 8 features from 8 h of every 16-feature step
      const bf16_t* qrow = qbase + (int64_t)(wave * 32 + j) * ld + 8 * h;
      const bf16_t* dorow = dobase + (int64_t)(wave * 32 + j) * p.H + 8 * h;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        qB[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
        doB[s] = *reinterpret_cast<const bf16x8*>(dorow + 16 * s);
      }
    }
    const float lse_q = lse[wave * 32 + j], d_q = Dq[wave * 32 + j];
    f32x16 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) dq[t][i] = 0.f;
    for (int kt = 0; kt < kmax; ++kt) {
      f32x16 st, dpt;
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = dpt[i] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kA[(kt * KS + s) * 64 + lane], qB[s], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vA[(kt * KS + s) * 64 + lane], doB[s], dpt, 0, 0, 0);
      }
      bf16x8 dsf[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bias = *reinterpret_cast<const float4*>(&mb[kt * 32 + 8 * g + 4 * h]);
        const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          const float pr = __builtin_amdgcn_exp2f(fmaf(st[i], p.scale2, bb[e]) - lse_q);
          dsf[i >> 3][i & 7] = (bf16_t)(pr * (dpt[i] - d_q) * p.scale);
        }
      }
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(kR, kt, t, 0, lane), dsf[0], dq[t], 0, 0, 0);
        dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(kR, kt, t, 1, lane), dsf[1], dq[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (bf16_t)dq[t][4 * g + e];
        *reinterpret_cast<bf16x4*>(drow + 32 * t + 8 * g + 4 * h) = w;
      }
  }

  // ---- pass B: dK, dV of key tile `wave` ----
  if constexpr (PASS == 1) {
    bf16x8 kB[KS], vB[KS];
    {
      const bf16_t* krow = qbase + p.H + (int64_t)(wave * 32 + j) * ld + 8 * h;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        kB[s] = *reinterpret_cast<const bf16x8*>(krow + 16 * s);
        vB[s] = *reinterpret_cast<const bf16x8*>(krow + p.H + 16 * s);
      }
    }
    const float mbk = mb[wave * 32 + j];
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) dk[t][i] = dv[t][i] = 0.f;
    const int qmax = wave < kmax ? kmax : 0;  // a fully masked key tile gets zero gradients
    for (int qt = 0; qt < qmax; ++qt) {
      f32x16 sq, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) sq[i] = dp[i] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qA[(qt * KS + s) * 64 + lane], kB[s], sq, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doA[(qt * KS + s) * 64 + lane], vB[s], dp, 0, 0, 0);
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l4 = *reinterpret_cast<const float4*>(&lse[qt * 32 + 8 * g + 4 * h]);
        const float4 d4 = *reinterpret_cast<const float4*>(&Dq[qt * 32 + 8 * g + 4 * h]);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          const float pr = __builtin_amdgcn_exp2f(fmaf(sq[i], p.scale2, mbk) - ll[e]);
          pf[i >> 3][i & 7] = (bf16_t)pr;
          dsf[i >> 3][i & 7] = (bf16_t)(pr * (dp[i] - dd[e]) * p.scale);
        }
      }
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(doR, qt, t, 0, lane), pf[0], dv[t], 0, 0, 0);
        dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(doR, qt, t, 1, lane), pf[1], dv[t], 0, 0, 0);
        dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(qR, qt, t, 0, lane), dsf[0], dk[t], 0, 0, 0);
        dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(att_vt_frag<DH>(qR, qt, t, 1, lane), dsf[1], dk[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 wk, wv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          wk[e] = (bf16_t)dk[t][4 * g + e];
          wv[e] = (bf16_t)dv[t][4 * g + e];
        }
        *reinterpret_cast<bf16x4*>(drow + p.H + 32 * t + 8 * g + 4 * h) = wk;
        *reinterpret_cast<bf16x4*>(drow + 2 * p.H + 32 * t + 8 * g + 4 * h) = wv;
      }
  }
}

// 64 x 64 tiles through LDS; both the loads and the stores move 8 bytes per thread in 128-byte row
// segments (R, C, leading dimensions and batch strides are multiples of 4 on this path)
__global__ __launch_bounds__(256) void transpose_kernel(TransposeArgs p) {
  __shared__ unsigned short tile[64][66];
  const int z = blockIdx.z, b1 = z / p.batch2, b2 = z - b1 * p.batch2;
  const unsigned short* in = reinterpret_cast<const unsigned short*>(p.in) + b1 * p.sI1 + b2 * p.sI2;
  unsigned short* out = reinterpret_cast<unsigned short*>(p.out) + b1 * p.sO1 + b2 * p.sO2;
  __shared__ float colacc[64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (p.colsum && threadIdx.x < 64) colacc[threadIdx.x] = 0.f;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};  // this thread's 4 columns over its 4 rows
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
    unsigned int lo = 0, hi = 0;
    if (r < p.R && c < p.C) {  // C % 4 == 0: the four columns are valid together
      const uint2 v = *reinterpret_cast<const uint2*>(in + (int64_t)r * p.ld_in + c);
      lo = v.x;
      hi = v.y;
    }
    unsigned int* dst = reinterpret_cast<unsigned int*>(&tile[ty + 16 * i][tx * 4]);
    dst[0] = lo;
    dst[1] = hi;
    cs[0] += __uint_as_float(lo << 16);
    cs[1] += __uint_as_float(lo & 0xffff0000u);
    cs[2] += __uint_as_float(hi << 16);
    cs[3] += __uint_as_float(hi & 0xffff0000u);
  }
  __syncthreads();
  if (p.colsum) {
    // lanes tx + 16 k of a wave hold the same columns: fold them (2 steps), then one LDS atomic per wave and column
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs[e] += __shfl_xor(cs[e], 16);
      cs[e] += __shfl_xor(cs[e], 32);
    }
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&colacc[tx * 4 + e], cs[e]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 16 * i, r = r0 + tx * 4;
    if (c < p.C && r < p.R) {
      const int lc = ty + 16 * i, lr = tx * 4;
      uint2 v;
      v.x = (unsigned int)tile[lr][lc] | ((unsigned int)tile[lr + 1][lc] << 16);
      v.y = (unsigned int)tile[lr + 2][lc] | ((unsigned int)tile[lr + 3][lc] << 16);
      *reinterpret_cast<uint2*>(out + (int64_t)c * p.ld_out + r) = v;
    }
  }
  if (p.colsum) {
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < p.C) atomicAdd(p.colsum + c0 + threadIdx.x, colacc[threadIdx.x]);
  }
}

// ------------------------------------------------------------------------- //
// LayerNorm: one wave per row, H <= 1024 (two 8-wide chunks per lane)
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, int64_t M, int H, bf16_t* __restrict__ y,
                                                         bf16_t* __restrict__ z_save, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[2][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (lane + 64 * i) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    if (c < H) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(a + row * H + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = (float)av[j];
      if (b) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b + row * H + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += (float)bv[j];
      }
      if (z_save) {
        bf16x8 zv;
#pragma unroll
        for (int j = 0; j < 8; ++j) zv[j] = (bf16_t)v[i][j];
        *reinterpret_cast<bf16x8*>(z_save + row * H + c) = zv;
        // the backward pass normalises the SAVED (bf16) z: use the same values here
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = (float)zv[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  }
  const float mean = wave_sum(s) / H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if ((lane + 64 * i) * 8 < H)
#pragma unroll
      for (int j = 0; j < 8; ++j) q += (v[i][j] - mean) * (v[i][j] - mean);
  const float rstd = rsqrtf(wave_sum(q) / H + eps);
  if (lane == 0 && mean_out) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((v[i][j] - mean) * rstd * gamma[c + j] + beta[c + j]);
      *reinterpret_cast<bf16x8*>(y + row * H + c) = o;
    }
  }
}

constexpr int LN_BWD_ROWS = 64;  // rows per workgroup (16 per wave)
// dy2 (optional): the incoming gradient is dy + dy2 (the residual branch's share), added in fp32 on the fly - the
// separate add kernel and its 3 passes over [tokens, hidden] are gone.  dz may alias dy or dy2: a wave reads its whole
// row into registers before it writes.
// NI = 16-byte pieces per lane and row (1: hidden <= 512, 2: <= 1024).  A wave takes RU = 2 rows per trip (4: no better) and requests
// every row's pieces before it reduces any: with one row in flight the kernel ran at 65 % of the copy rate (every row
// is a load -> wave reduction -> store chain).
template <int NI, int RU>   // RU rows per trip
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* dy, const bf16_t* dy2, const bf16_t* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, int64_t M, int H,
                                                     bf16_t* dz, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dz_colsum) {
  __shared__ float red[3][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[NI][8], db[NI][8], gm[NI][8], dzs[NI][8];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = (lane + 64 * i) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dg[i][j] = db[i][j] = dzs[i][j] = 0.f;
      gm[i][j] = c < H ? gamma[c + j] : 0.f;
    }
  }
  const int64_t row0 = (int64_t)blockIdx.x * LN_BWD_ROWS + wave * (LN_BWD_ROWS / 4);
  for (int r = 0; r < LN_BWD_ROWS / 4; r += RU) {
    if (row0 + r >= M) break;
    bf16x8 dv[RU][NI], zv[RU][NI], d2[RU][NI];
    float mu[RU], rs[RU];
    bool live[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {   // every row's loads first
      const int64_t row = row0 + r + u;
      live[u] = row < M;
      mu[u] = live[u] ? mean[row] : 0.f;
      rs[u] = live[u] ? rstd[row] : 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = (lane + 64 * i) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[u][i][j] = zv[u][i][j] = d2[u][i][j] = (bf16_t)0.f;
        if (live[u] && c < H) {
          dv[u][i] = *reinterpret_cast<const bf16x8*>(dy + row * H + c);
          zv[u][i] = *reinterpret_cast<const bf16x8*>(z + row * H + c);
          if (dy2) d2[u][i] = *reinterpret_cast<const bf16x8*>(dy2 + row * H + c);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (!live[u]) continue;
      const int64_t row = row0 + r + u;
      float g[NI][8], xh[NI][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const bool on = (lane + 64 * i) * 8 < H;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)dv[u][i][j] + (float)d2[u][i][j];
          xh[i][j] = on ? ((float)zv[u][i][j] - mu[u]) * rs[u] : 0.f;
          g[i][j] = d * gm[i][j];
          dg[i][j] += d * xh[i][j];
          db[i][j] += d;
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
        }
      }
      s1 = wave_sum(s1) / H;
      s2 = wave_sum(s2) / H;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < H) {
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            o[j] = (bf16_t)(rs[u] * (g[i][j] - s1 - xh[i][j] * s2));
            dzs[i][j] += (float)o[j];  // column sums of dz as stored: the bias gradient of the product that made z
          }
          *reinterpret_cast<bf16x8*>(dz + row * H + c) = o;
        }
      }
    }
  }
  // the four waves' partial parameter gradients meet in LDS; one atomic per column per workgroup
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < H)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            red[0][c + j] = (w ? red[0][c + j] : 0.f) + dg[i][j];
            red[1][c + j] = (w ? red[1][c + j] : 0.f) + db[i][j];
            red[2][c + j] = (w ? red[2][c + j] : 0.f) + dzs[i][j];
          }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(dgamma + c, red[0][c]);
    atomicAdd(dbeta + c, red[1][c]);
    if (dz_colsum) atomicAdd(dz_colsum + c, red[2][c]);
  }
}

// ------------------------------------------------------------------------- //
// softmax over rows of S <= 512 scores, one wave per row
// ------------------------------------------------------------------------- //
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void softmax_fwd_kernel(bf16_t* __restrict__ sc, const int32_t* __restrict__ key_mask,
                                                          int64_t rows, int heads, int S, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t b = row / ((int64_t)heads * S);
  bf16_t* p = sc + row * S;
  float v[2][4];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = (lane + 64 * i) * 4;  // S % 4 == 0: four keys are valid together
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = -INFINITY;
    if (j < S) {
      const bf16x4v x = *reinterpret_cast<const bf16x4v*>(p + j);
      const int4 km = *reinterpret_cast<const int4*>(key_mask + b * S + j);
      const int k[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (k[e] != 0) v[i][e] = (float)x[e] * scale;
        mx = fmaxf(mx, v[i][e]);
      }
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = mx == -INFINITY ? 0.f : __expf(v[i][e] - mx);
      sum += v[i][e];
    }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = (lane + 64 * i) * 4;
    if (j < S) {
      bf16x4v o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(v[i][e] * inv);
      *reinterpret_cast<bf16x4v*>(p + j) = o;
    }
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(bf16_t* __restrict__ dP, const bf16_t* __restrict__ P,
                                                          int64_t rows, int S, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  bf16_t* d = dP + row * S;
  const bf16_t* p = P + row * S;
  float dv[2][4], pv[2][4];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = (lane + 64 * i) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) dv[i][e] = pv[i][e] = 0.f;
    if (j < S) {
      const bf16x4v a = *reinterpret_cast<const bf16x4v*>(d + j);
      const bf16x4v b = *reinterpret_cast<const bf16x4v*>(p + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dv[i][e] = (float)a[e];
        pv[i][e] = (float)b[e];
        dot += dv[i][e] * pv[i][e];
      }
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = (lane + 64 * i) * 4;
    if (j < S) {
      bf16x4v o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(scale * pv[i][e] * (dv[i][e] - dot));
      *reinterpret_cast<bf16x4v*>(d + j) = o;
    }
  }
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ h, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const bf16x8 x = reinterpret_cast<const bf16x8*>(u)[i];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)x[j];
      o[j] = (bf16_t)(0.5f * f * (1.0f + erff(f * 0.70710678118654752f)));
    }
    reinterpret_cast<bf16x8*>(h)[i] = o;
  }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ dh,
                                                       bf16_t* __restrict__ du, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const bf16x8 x = reinterpret_cast<const bf16x8*>(u)[i];
    const bf16x8 g = reinterpret_cast<const bf16x8*>(dh)[i];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)x[j];
      const float cdf = 0.5f * (1.0f + erff(f * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * f * f);
      o[j] = (bf16_t)((float)g[j] * (cdf + f * pdf));
    }
    reinterpret_cast<bf16x8*>(du)[i] = o;
  }
}

// du = dh * gelu'(u) over a [M, F] matrix AND db[c] += sum_r du[r][c] (the FFN1 bias gradient: the separate column-sum
// pass re-read all of du).  Workgroup = RY row-lanes x F / 8 column chunks (a row's chunks are 16-byte loads by
// consecutive threads); a thread walks rows blockIdx.x RY + ry, + gridDim.x RY, ... two at a time, keeps the column sums
// of du AS STORED (bf16) in registers; the row-lanes meet in LDS and the workgroup adds ONCE per column: 512
// workgroups x F atomics per call (2 048 x F on the same F addresses ran 5x slower than the unfused pair).
constexpr int GELU_CS_BLOCKS = 512;
__global__ __launch_bounds__(1024) void gelu_bwd_colsum_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ dh,
                                                               bf16_t* __restrict__ du, float* __restrict__ db, int64_t M,
                                                               int F8, int RY) {
  extern __shared__ float gcs_part[];   // [F8 * 8]
  const int c = threadIdx.x % F8, ry = threadIdx.x / F8;
  for (int i = threadIdx.x; i < F8 * 8; i += blockDim.x) gcs_part[i] = 0.f;
  __syncthreads();
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto one = [&](const bf16x8& x, const bf16x8& g, int64_t i) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)x[j];
      const float cdf = 0.5f * (1.0f + erff(f * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * f * f);
      o[j] = (bf16_t)((float)g[j] * (cdf + f * pdf));
      s8[j] += (float)o[j];
    }
    reinterpret_cast<bf16x8*>(du)[i] = o;
  };
  if (ry < RY) {
    const int64_t step = (int64_t)gridDim.x * RY;
    int64_t r = (int64_t)blockIdx.x * RY + ry;
    for (; r + step < M; r += 2 * step) {   // two rows in flight
      const int64_t i0 = r * F8 + c, i1 = (r + step) * F8 + c;
      const bf16x8 x0 = reinterpret_cast<const bf16x8*>(u)[i0], g0 = reinterpret_cast<const bf16x8*>(dh)[i0];
      const bf16x8 x1 = reinterpret_cast<const bf16x8*>(u)[i1], g1 = reinterpret_cast<const bf16x8*>(dh)[i1];
      one(x0, g0, i0);
      one(x1, g1, i1);
    }
    if (r < M) {
      const int64_t i0 = r * F8 + c;
      one(reinterpret_cast<const bf16x8*>(u)[i0], reinterpret_cast<const bf16x8*>(dh)[i0], i0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&gcs_part[8 * c + j], s8[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < F8 * 8; i += blockDim.x) atomicAdd(db + i, gcs_part[i]);
}

__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                  bf16_t* __restrict__ c, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const bf16x8 x = reinterpret_cast<const bf16x8*>(a)[i], y = reinterpret_cast<const bf16x8*>(b)[i];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)x[j] + (float)y[j]);
    reinterpret_cast<bf16x8*>(c)[i] = o;
  }
}

constexpr int COLSUM_ROWS = 128;  // rows per workgroup
// db[c] += sum_r dY[r][c].  Thread = one 8-column chunk (16-byte loads, whole rows coalesced) of every (256 / chunks)-th
// row of a 256-row slab; the row-threads of a chunk meet through LDS atomics, one global atomic per column and slab.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ dY, int64_t M, int N, int64_t ld,
                                                     float* __restrict__ db) {
  __shared__ float part[1024];  // N <= 1024 per workgroup column window
  const int chunks = (N + 7) / 8;                 // N % 8 == 0 here
  const int lanes_r = 256 / chunks > 0 ? 256 / chunks : 1;
  const int c = threadIdx.x % chunks, rr = threadIdx.x / chunks;
  for (int i = threadIdx.x; i < N; i += 256) part[i] = 0.f;
  __syncthreads();
  if (rr < lanes_r) {
    const int64_t r0 = (int64_t)blockIdx.x * COLSUM_ROWS;
    const int64_t r1 = r0 + COLSUM_ROWS < M ? r0 + COLSUM_ROWS : M;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t r = r0 + rr;
    for (; r + 3 * lanes_r < r1; r += 4 * lanes_r) {  // four loads in flight per thread
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(dY + (r + u * lanes_r) * ld + 8 * c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] += (float)v[u][e];
    }
    for (; r < r1; r += lanes_r) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dY + r * ld + 8 * c);
#pragma unroll
      for (int e = 0; e < 8; ++e) s8[e] += (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&part[8 * c + e], s8[e]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256) atomicAdd(db + i, part[i]);
}

// ------------------------------------------------------------------------- //
// embeddings
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int32_t* __restrict__ ids, const bf16_t* __restrict__ word,
                                                        const bf16_t* __restrict__ pos, const bf16_t* __restrict__ type0,
                                                        int64_t M, int S, int H, int vocab, int pos_offset,
                                                        bf16_t* __restrict__ z) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const int t = (int)(row % S) + pos_offset;
  for (int c = lane * 8; c < H; c += 512) {
    const bf16x8 w = *reinterpret_cast<const bf16x8*>(word + (int64_t)id * H + c);
    const bf16x8 p = *reinterpret_cast<const bf16x8*>(pos + (int64_t)t * H + c);
    const bf16x8 ty = *reinterpret_cast<const bf16x8*>(type0 + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)w[j] + (float)p[j] + (float)ty[j]);
    *reinterpret_cast<bf16x8*>(z + row * H + c) = o;
  }
}

// word-embedding rows: one wave per token, fp32 atomics (ids repeat rarely inside a step)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ mask,
                                                        const bf16_t* __restrict__ dz, int64_t M, int H, int vocab,
                                                        float* __restrict__ dword) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M || mask[row] == 0) return;  // padding positions carry exactly zero gradient
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  for (int c = lane; c < H; c += 64) atomicAdd(dword + (int64_t)id * H + c, (float)dz[row * H + c]);
}

// position / token-type rows: every sequence adds into the SAME S position rows and the one type row, so atomics
// per token would queue B (resp. B * S) deep on each address (950 us at 65 k tokens).  One workgroup per position
// sums its B rows in registers instead: one atomic per (position, column) and one more for the type row.
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(const int32_t* __restrict__ mask, const bf16_t* __restrict__ dz,
                                                            int B, int S, int H, int pos_offset, float* __restrict__ dpos,
                                                            float* __restrict__ dtype0) {
  const int t = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += 256) {
    float sum = 0.f;
    for (int b = 0; b < B; ++b) {
      const int64_t row = (int64_t)b * S + t;
      if (mask[row] != 0) sum += (float)dz[row * H + c];
    }
    atomicAdd(dpos + (int64_t)(t + pos_offset) * H + c, sum);
    atomicAdd(dtype0 + c, sum);
  }
}

// ------------------------------------------------------------------------- //
// masked mean pool (+ L2 normalise), one workgroup per sequence
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void pool_fwd_kernel(const bf16_t* __restrict__ hidden, const int32_t* __restrict__ mask,
                                                       int S, int H, int normalize, float* __restrict__ out,
                                                       float* __restrict__ pooled_save) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.f;
  for (int t = 0; t < S; ++t) {
    if (mask[(int64_t)b * S + t] == 0) continue;
    cnt += 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if (c < H) acc[i] += (float)hidden[((int64_t)b * S + t) * H + c];
    }
  }
  const float n = fmaxf(cnt, 1e-9f);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc[i] /= n;
    if (tid + 256 * i < H) {
      ss += acc[i] * acc[i];
      if (pooled_save) pooled_save[(int64_t)b * H + tid + 256 * i] = acc[i];
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float norm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    if (c < H) out[(int64_t)b * H + c] = normalize ? acc[i] / norm : acc[i];
  }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ pooled,
                                                       const int32_t* __restrict__ mask, int S, int H, int normalize,
                                                       bf16_t* __restrict__ dhidden) {
  __shared__ float red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float e[4], g[4];
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    e[i] = c < H ? pooled[(int64_t)b * H + c] : 0.f;
    g[i] = c < H ? dout[(int64_t)b * H + c] : 0.f;
    ss += e[i] * e[i];
    dot += e[i] * g[i];
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = ss;
    red[1][tid >> 6] = dot;
  }
  __syncthreads();
  float cnt = 0.f;
  for (int t = 0; t < S; ++t) cnt += mask[(int64_t)b * S + t] != 0 ? 1.f : 0.f;
  const float n = fmaxf(cnt, 1e-9f);
  if (normalize) {
    const float nrm2 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float nrm = fmaxf(sqrtf(nrm2), 1e-12f);
    const float ed = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (nrm * nrm);  // (e_hat . g) / ||e||
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = (g[i] - e[i] * ed) / nrm;  // de = (g - e_hat (e_hat . g)) / ||e||
  }
  for (int t = 0; t < S; ++t) {
    const float m = mask[(int64_t)b * S + t] != 0 ? 1.0f / n : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if (c < H) dhidden[((int64_t)b * S + t) * H + c] = (bf16_t)(g[i] * m);
    }
  }
}

inline unsigned grid1d(int64_t n) {
  int64_t b = sskd::ceil_div(n, 256);
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

// ------------------------------------------------------------------------- //
// launchers
// ------------------------------------------------------------------------- //
int launch_gemm_nt(const GemmArgs& a, hipStream_t st) {
  SSKD_REQUIRE(a.M >= 0 && a.N >= 0 && a.K > 0 && a.batch1 >= 1 && a.batch2 >= 1, "gemm_nt: bad shape");
  if (a.M == 0 || a.N == 0) return SSKD_OK;
  SSKD_REQUIRE(a.K % 32 == 0, "gemm_nt: K=%d must be a multiple of 32", a.K);
  SSKD_REQUIRE(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.sA1 % 8 == 0 && a.sA2 % 8 == 0 && a.sB1 % 8 == 0 && a.sB2 % 8 == 0,
               "gemm_nt: operand strides must be multiples of 8 elements");
  SSKD_REQUIRE(a.A && a.B && a.C, "gemm_nt: null pointer");
  SSKD_REQUIRE((reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0,
               "gemm_nt: operands must be 16-byte aligned");
  SSKD_REQUIRE(!a.accumulate || a.c_is_f32, "gemm_nt: accumulate needs an fp32 output");
  SSKD_REQUIRE(a.split_k <= 1 || (a.accumulate && a.c_is_f32 && a.batch1 * a.batch2 == 1 && !a.bias),
               "gemm_nt: split-K needs an unbatched fp32 accumulating output without bias");
  {
    int rc = SSKD_OK;
    if (blaslt_gemm_nt(a, st, &rc)) return rc;   // plain large-K product: the vendor library's kernel (blaslt.hip)
  }
  const dim3 grid((unsigned)sskd::ceil_div(a.N, 128), (unsigned)sskd::ceil_div(a.M, 128),
                  (unsigned)(a.split_k > 1 ? a.split_k : a.batch1 * a.batch2));
  const bool big = a.batch1 * a.batch2 == 1 && a.split_k <= 1 && !a.c_is_f32 && a.M % 256 == 0 && a.N % 128 == 0 &&
                   a.K % 64 == 0 && a.M >= 1024 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc % 8 == 0 &&
                   (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.C) & 15) == 0;
#ifndef SSKD_NO_GEMM256
  if (big) {
    auto go = [&](auto kern, int bn) {
      const int lds256 = 2 * (256 + bn) * 64 * 2 + 8 * 16 * (bn / 4 + 8) * 2;  // two stages + the waves' epilogue slices
      const int tiles = (a.N / bn) * (a.M / 256);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
      hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < GEMM256_CUS ? tiles : GEMM256_CUS)), dim3(512), lds256, st, a);
    };
    if (a.N % 256 == 0) go(gemm_nt256_kernel<256>, 256);
#ifndef SSKD_GEMM_NO_BN192
    // a 128-column tile leaves a wave 128 x 32: 10 KiB of fragment reads per 16 MFMAs, more than the LDS delivers in
    // their time; 192 columns (128 x 48 per wave: 11 KiB per 24 MFMAs) fit under it
    else if (a.N % 192 == 0) go(gemm_nt256_kernel<192>, 192);
#endif
    else go(gemm_nt256_kernel<128>, 128);
    return sskd::check_launch("gemm_nt256_kernel");
  }
#endif
  if (a.K % 64 == 0) hipLaunchKernelGGL(gemm_nt_kernel<64>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gemm_nt_kernel<32>, grid, dim3(256), 0, st, a);
  return sskd::check_launch("gemm_nt_kernel");
}

int launch_attention_fwd(const bf16_t* qkv, const int32_t* key_mask, int B, int S, int heads, int DH, float scale,
                         bf16_t* ctx, float* lse, hipStream_t st) {
  if (B == 0) return SSKD_OK;
  SSKD_REQUIRE(qkv && key_mask && ctx, "attention_fwd: null pointer");
  SSKD_REQUIRE(S >= 32 && S % 32 == 0 && S <= 512, "attention_fwd: S=%d must be a multiple of 32 in [32, 512]", S);
  SSKD_REQUIRE(DH == 32 || DH == 64 || DH == 128, "attention_fwd: head width %d not in {32, 64, 128}", DH);
  AttnArgs a{};
  a.qkv = qkv;
  a.mask = key_mask;
  a.ctx = ctx;
  a.lse = lse;
  a.B = B;
  a.S = S;
  a.NH = heads;
  a.H = heads * DH;
  a.nkt = S / 32;
  a.qsplit = (a.nkt + 7) / 8;
  a.scale2 = scale * 1.4426950408889634f;
  const size_t lds = (size_t)S * DH * 2 * 2 + (size_t)S * sizeof(float);
  const dim3 grid((unsigned)(B * heads * a.qsplit));
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);
  };
  if (DH == 32) go(attention_fwd_kernel<32>);
  else if (DH == 64) go(attention_fwd_kernel<64>);
  else go(attention_fwd_kernel<128>);
  return sskd::check_launch("attention_fwd_kernel");
}

bool attention_bwd_supported(int S, int DH) {
  return (DH == 32 || DH == 64) && S >= 32 && S % 32 == 0 && S <= 256 && S * DH <= 8192;
}

int launch_attention_bwd(const bf16_t* qkv, const int32_t* key_mask, const bf16_t* ctx, const bf16_t* dctx,
                         const float* lse, int B, int S, int heads, int DH, float scale, bf16_t* dqkv, hipStream_t st) {
  if (B == 0) return SSKD_OK;
  SSKD_REQUIRE(qkv && key_mask && ctx && dctx && lse && dqkv, "attention_bwd: null pointer");
  SSKD_REQUIRE(attention_bwd_supported(S, DH), "attention_bwd: S=%d, head width %d not served (S * DH <= 8192)", S, DH);
  AttnBwdArgs a{};
  a.qkv = qkv;
  a.mask = key_mask;
  a.ctx = ctx;
  a.dctx = dctx;
  a.lse = lse;
  a.dqkv = dqkv;
  a.B = B;
  a.S = S;
  a.NH = heads;
  a.H = heads * DH;
  a.nkt = S / 32;
  a.scale = scale;
  a.scale2 = scale * 1.4426950408889634f;
  auto go = [&](auto kern, int blocks) {
    const size_t lds = (size_t)blocks * S * DH * 2 + (size_t)3 * S * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * heads)), dim3(512), lds, st, a);
  };
  if (DH == 32) {
    go(attention_bwd_kernel<32, 0>, 3);
    go(attention_bwd_kernel<32, 1>, 4);
  } else {
    go(attention_bwd_kernel<64, 0>, 3);
    go(attention_bwd_kernel<64, 1>, 4);
  }
  return sskd::check_launch("attention_bwd_kernel");
}

bool gemm_tn_supported(int64_t T, int M, int N, int64_t lda, int64_t ldb) {
  return T >= 64 && T % 64 == 0 && M > 0 && M % 384 == 0 && N > 0 && N % 128 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
         T * lda < ((int64_t)1 << 31) && T * ldb < ((int64_t)1 << 31);
}

int launch_gemm_tn(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t T, int M,
                   int N, hipStream_t st) {
  SSKD_REQUIRE(A && B && C, "gemm_tn: null pointer");
  SSKD_REQUIRE(gemm_tn_supported(T, M, N, lda, ldb), "gemm_tn: shape T=%lld M=%d N=%d not served", (long long)T, M, N);
  SSKD_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
               "gemm_tn: operands must be 16-byte aligned");
  GemmTnArgs a{};
  a.A = A;
  a.B = B;
  a.C = C;
  a.lda = lda;
  a.ldb = ldb;
  a.ldc = ldc;
  a.T = T;
  a.M = M;
  a.N = N;
  const int tiles = (M / 384) * (N / 128);
  const int nk = (int)(T / 64);
  // K slices, at least 4 K tiles each, every workgroup of the launch resident at once (one per CU).  The tiles of a
  // slice share operand panels: with slices s, s + 8, ... on XCD s & 7 they come out of ONE L2 (dWo 66 -> 61 us,
  // dWqkv 117 -> 102, dW1 141 -> 130 at 65 536 tokens) - except for a single row of 12 tiles (dW2: the shared A panel
  // is small, and whole slices per XCD leave a quarter of the CUs idle: 125 -> 131 us), which keeps the dealt order.
  a.xcd_map = !(M / 384 == 1 && tiles >= 12);
  int split;
  if (a.xcd_map) {
    int per_xcd = 32 / tiles;
    if (per_xcd < 1) per_xcd = 1;
    split = 8 * per_xcd;
  } else {
    split = GEMM256_CUS / tiles;
    if (split < 1) split = 1;
  }
  if (split > nk / 4) split = nk / 4 > 0 ? nk / 4 : 1;
  a.k_per = (int)sskd::ceil_div(nk, split);
  split = (int)sskd::ceil_div(nk, a.k_per);
  const unsigned grid = a.xcd_map ? 8u * (unsigned)(tiles * (int)sskd::ceil_div(split, 8)) : (unsigned)(tiles * split);
  constexpr int lds_bytes = 2 * (384 + 128) * 64 * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn384_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(gemm_tn384_kernel, dim3(grid), dim3(512), lds_bytes, st, a);
  return sskd::check_launch("gemm_tn384_kernel");
}

int launch_transpose(const TransposeArgs& a, hipStream_t st) {
  if (a.R == 0 || a.C == 0) return SSKD_OK;
  SSKD_REQUIRE(a.R % 4 == 0 && a.C % 4 == 0 && a.ld_in % 4 == 0 && a.ld_out % 4 == 0 && a.sI1 % 4 == 0 && a.sI2 % 4 == 0 &&
                   a.sO1 % 4 == 0 && a.sO2 % 4 == 0,
               "transpose: shapes and strides must be multiples of 4");
  SSKD_REQUIRE(!a.colsum || a.batch1 * a.batch2 == 1, "transpose: the column-sum side output is for unbatched calls");
  const dim3 grid((unsigned)sskd::ceil_div(a.C, 64), (unsigned)sskd::ceil_div(a.R, 64), (unsigned)(a.batch1 * a.batch2));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, a);
  return sskd::check_launch("transpose_kernel");
}

int launch_add_ln_fwd(const bf16_t* a, const bf16_t* b, const float* gamma, const float* beta, float eps,
                      int64_t M, int H, bf16_t* y, bf16_t* z_save, float* mean, float* rstd, hipStream_t st) {
  SSKD_REQUIRE(H % 8 == 0 && H <= 1024, "layernorm: hidden=%d must be a multiple of 8, at most 1024", H);
  if (M == 0) return SSKD_OK;
  hipLaunchKernelGGL(add_ln_fwd_kernel, dim3((unsigned)sskd::ceil_div(M, 4)), dim3(256), 0, st, a, b, gamma, beta, eps,
                     M, H, y, z_save, mean, rstd);
  return sskd::check_launch("add_ln_fwd_kernel");
}

int launch_ln_bwd(const bf16_t* dy, const bf16_t* z, const float* mean, const float* rstd, const float* gamma,
                  int64_t M, int H, bf16_t* dz, float* dgamma, float* dbeta, hipStream_t st, float* dz_colsum,
                  const bf16_t* dy2) {
  SSKD_REQUIRE(H % 8 == 0 && H <= 1024, "layernorm: hidden=%d must be a multiple of 8, at most 1024", H);
  if (M == 0) return SSKD_OK;
  if (H <= 512)
    hipLaunchKernelGGL((ln_bwd_kernel<1, 2>), dim3((unsigned)sskd::ceil_div(M, LN_BWD_ROWS)), dim3(256), 0, st, dy, dy2, z, mean,
                       rstd, gamma, M, H, dz, dgamma, dbeta, dz_colsum);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<2, 2>), dim3((unsigned)sskd::ceil_div(M, LN_BWD_ROWS)), dim3(256), 0, st, dy, dy2, z, mean,
                       rstd, gamma, M, H, dz, dgamma, dbeta, dz_colsum);
  return sskd::check_launch("ln_bwd_kernel");
}

int launch_softmax_fwd(bf16_t* scores, const int32_t* key_mask, int B, int heads, int S, float scale, hipStream_t st) {
  SSKD_REQUIRE(S >= 4 && S <= 512 && S % 4 == 0, "softmax: S=%d must be a multiple of 4 in [4, 512]", S);
  const int64_t rows = (int64_t)B * heads * S;
  if (rows == 0) return SSKD_OK;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)sskd::ceil_div(rows, 4)), dim3(256), 0, st, scores, key_mask,
                     rows, heads, S, scale);
  return sskd::check_launch("softmax_fwd_kernel");
}

int launch_softmax_bwd(bf16_t* dP, const bf16_t* P, int64_t rows, int S, float scale, hipStream_t st) {
  SSKD_REQUIRE(S >= 4 && S <= 512 && S % 4 == 0, "softmax: S=%d must be a multiple of 4 in [4, 512]", S);
  if (rows == 0) return SSKD_OK;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)sskd::ceil_div(rows, 4)), dim3(256), 0, st, dP, P, rows, S, scale);
  return sskd::check_launch("softmax_bwd_kernel");
}

int launch_gelu_fwd(const bf16_t* u, bf16_t* h, int64_t n, hipStream_t st) {
  SSKD_REQUIRE(n % 8 == 0, "gelu: element count must be a multiple of 8");
  if (n == 0) return SSKD_OK;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid1d(n / 8)), dim3(256), 0, st, u, h, n / 8);
  return sskd::check_launch("gelu_fwd_kernel");
}

int launch_gelu_bwd(const bf16_t* u, const bf16_t* dh, bf16_t* du, int64_t n, hipStream_t st) {
  SSKD_REQUIRE(n % 8 == 0, "gelu: element count must be a multiple of 8");
  if (n == 0) return SSKD_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid1d(n / 8)), dim3(256), 0, st, u, dh, du, n / 8);
  return sskd::check_launch("gelu_bwd_kernel");
}

int launch_gelu_bwd_colsum(const bf16_t* u, const bf16_t* dh, bf16_t* du, float* db, int64_t M, int F, hipStream_t st) {
  SSKD_REQUIRE(F % 8 == 0, "gelu: width must be a multiple of 8");
  if (M == 0 || F == 0) return SSKD_OK;
  const int F8 = F / 8;
  if (F8 > 512) {   // wide FFNs: the plain pair
    int rc = launch_gelu_bwd(u, dh, du, M * F, st);
    return rc != SSKD_OK ? rc : launch_colsum(du, M, F, F, db, st);
  }
  const int RY = 1024 / F8 > 0 ? 1024 / F8 : 1;   // 5 row-lanes at F = 1536 (960 threads)
  const int threads = ((F8 * RY + 63) / 64) * 64;
  int64_t blocks = sskd::ceil_div(M, (int64_t)RY);
  if (blocks > GELU_CS_BLOCKS) blocks = GELU_CS_BLOCKS;
  hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3((unsigned)blocks), dim3(threads), (size_t)F * sizeof(float), st, u, dh, du, db,
                     M, F8, RY);
  return sskd::check_launch("gelu_bwd_colsum_kernel");
}

int launch_colsum(const bf16_t* dY, int64_t M, int N, int64_t ld, float* db, hipStream_t st) {
  if (M == 0 || N == 0) return SSKD_OK;
  SSKD_REQUIRE(N % 8 == 0 && ld % 8 == 0, "colsum: N and ld must be multiples of 8");
  // column windows of at most 1024 (256 threads x ... chunks): wider matrices take several launches
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int w = N - n0 < 1024 ? N - n0 : 1024;
    // 2048-wide rows would need > 256 chunks per row: windows keep chunks <= 128
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)sskd::ceil_div(M, COLSUM_ROWS)), dim3(256), 0, st, dY + n0, M, w, ld,
                       db + n0);
  }
  return sskd::check_launch("colsum_kernel");
}

int launch_add(const bf16_t* a, const bf16_t* b, bf16_t* c, int64_t n, hipStream_t st) {
  SSKD_REQUIRE(n % 8 == 0, "add: element count must be a multiple of 8");
  if (n == 0) return SSKD_OK;
  hipLaunchKernelGGL(add_kernel, dim3(grid1d(n / 8)), dim3(256), 0, st, a, b, c, n / 8);
  return sskd::check_launch("add_kernel");
}

int launch_embed_fwd(const int32_t* ids, const int32_t* mask, const bf16_t* word, const bf16_t* pos, const bf16_t* type0,
                     int B, int S, int H, int vocab, int pos_offset, bf16_t* z, hipStream_t st) {
  (void)mask;
  SSKD_REQUIRE(H % 8 == 0, "embed: hidden must be a multiple of 8");
  const int64_t M = (int64_t)B * S;
  if (M == 0) return SSKD_OK;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)sskd::ceil_div(M, 4)), dim3(256), 0, st, ids, word, pos, type0, M,
                     S, H, vocab, pos_offset, z);
  return sskd::check_launch("embed_fwd_kernel");
}

int launch_embed_bwd(const int32_t* ids, const int32_t* mask, const bf16_t* dz, int B, int S, int H, int vocab,
                     int pos_offset, float* dword, float* dpos, float* dtype0, hipStream_t st) {
  const int64_t M = (int64_t)B * S;
  if (M == 0) return SSKD_OK;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)sskd::ceil_div(M, 4)), dim3(256), 0, st, ids, mask, dz, M, H, vocab,
                     dword);
  hipLaunchKernelGGL(embed_pos_bwd_kernel, dim3((unsigned)S), dim3(256), 0, st, mask, dz, B, S, H, pos_offset, dpos, dtype0);
  return sskd::check_launch("embed_bwd_kernel");
}

int launch_pool_fwd(const bf16_t* hidden, const int32_t* mask, int B, int S, int H, int normalize, float* out,
                    float* pooled_save, hipStream_t st) {
  SSKD_REQUIRE(H <= 1024, "pool: hidden=%d > 1024", H);
  if (B == 0) return SSKD_OK;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(B), dim3(256), 0, st, hidden, mask, S, H, normalize, out, pooled_save);
  return sskd::check_launch("pool_fwd_kernel");
}

int launch_pool_bwd(const float* dout, const float* pooled, const int32_t* mask, int B, int S, int H, int normalize,
                    bf16_t* dhidden, hipStream_t st) {
  SSKD_REQUIRE(H <= 1024, "pool: hidden=%d > 1024", H);
  if (B == 0) return SSKD_OK;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(B), dim3(256), 0, st, dout, pooled, mask, S, H, normalize, dhidden);
  return sskd::check_launch("pool_bwd_kernel");
}

}  // namespace sskd_generic
