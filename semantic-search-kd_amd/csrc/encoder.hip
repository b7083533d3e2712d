// e5-small-v2 shaped BERT encoder forward on gfx950 (bf16 MFMA, fp32 accumulate).
//
// Replaces what the reference's StudentModel delegates to
// sentence_transformers.SentenceTransformer.encode -> transformers.BertModel
// (reference: src/serve/app.py:287,385-389; tests/test_model_validation.py:80-89;
// architecture constants SURVEY.md App. B / configs/kd.yaml:13-19).
//
// Formulation.  Every GEMM is computed TRANSPOSED, D^T[feature, token] = W[feature, k] * X^T[k, token]
// with v_mfma_f32_32x32x16_bf16: weights are the A operand (pre-tiled on the host into
// fragment order, staged through LDS and shared by the workgroup's waves), activations are the
// B operand with the TOKEN ON THE LANE.  A 32x32 result then has the token on the lane and 16
// features in registers, so bias / GELU / residual / LayerNorm / softmax are lane-local and an
// accumulator tile can feed the next MFMA as its B operand (attention P*V).
//
// Activation layout ("fragment order").  A [T, K] activation lives in HBM as
//     [T/32 token tiles][K/16 k-steps][64 lanes][8 bf16]
// where lane l of fragment (tt, s) holds X[32 tt + (l & 31)][16 s + 8 (l >> 5) + j], j = 0..7 -
// exactly one B operand of the MFMA.  Consequences: (1) a wave loads its activations with
// contiguous 1 KiB reads, (2) the epilogue of a 32-feature tile - lane (r, h) holds features
// 8g + 4h + e of token r - stores 4 x 8 bytes per lane that land as 512 contiguous bytes per
// wave-instruction, with no shuffles.  Sequences are padded to a multiple of 32 tokens
// (S_pad) so that a token tile never straddles two sequences; token buffers are padded to a
// multiple of 256 rows so that no kernel needs a bounds check.
//
// Kernels per layer (hidden 384, 12 heads x 32, FFN 1536):
//   qkv_attention_kernel    ctx = softmax((X Wq^T)(X Wk^T)^T / sqrt(32) + mask)(X Wv^T)   (Q, K, V stay on chip)
//   fused_mlp_ln_kernel     X1 = LN(X  + ctx Wo^T + bo)                      (prologue; X1 stays on chip)
//                           X2 = LN(X1 + gelu(X1 W1^T + b1) W2^T + b2)       (hidden 1536 stays on chip)
#include "common.h"

#include <algorithm>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int H = 384;
constexpr int NH = 12;
constexpr int DH = 32;
constexpr int FF = 1536;
constexpr int KSTEPS = H / 16;           // 24 MFMA k-steps per 384-wide K chunk
constexpr int WTILE_VEC = 32 * H / 8;    // bf16x8 vectors in one [32 x 384] weight tile (24 KiB)
constexpr float LOG2E = 1.4426950408889634f;

__device__ inline f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ inline bf16x8 zero_bf8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f;
  return z;
}

__device__ inline float bf2f(__bf16 v) { return (float)v; }

using sskd::gelu_erf;   // common.h: erf-GELU as x * sigmoid(cubic-in-x^2), 7 VALU + 2 transcendental instructions

// Exchange with lane ^ 32 in the VALU (v_permlane32_swap) instead of __shfl_xor's ds_bpermute,
// which is an LDS round trip in the middle of a dependency chain.  After the swap r[0] / r[1]
// hold {own, partner} in some order on every lane, so symmetric reductions need no select.
__device__ inline float pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float pair_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// first vector (lane 0) of fragment (token tile tt, k-step s) of an activation with KS k-steps
__device__ inline int64_t frag_base(int64_t tt, int s, int KS) { return (tt * KS + s) * 64; }

// Store the 16 accumulator values of a finished 32-feature tile in fragment order.  Lane (r, h)
// holds features 8g + 4h + e (g, e = 0..3) of token r, i.e. HALF of each of the four 16-byte
// fragment slots of its token.  One v_permlane32_swap per dword trades halves inside the lane
// pair (r, 0) <-> (r, 1): afterwards lane l owns the complete slot l of k-step 2 nt (from groups
// g = 0, 1) and of k-step 2 nt + 1 (g = 2, 3), so the tile leaves as two fully linear 1 KiB
// dwordx4 stores.  `dst` points at fragment (tt, 2 nt) lane 0.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ inline void store_tile_frag(__bf16* __restrict__ dst, const f32x4 (&v)[4], int lane) {
  u32x2 d[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const bf16x4 o = __builtin_convertvector(v[g], bf16x4);
    d[g] = __builtin_bit_cast(u32x2, o);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    // swap(a = d[2s], b = d[2s+1]): lanes < 32 end up with (own a, partner's a) = features 0..7 of
    // the step, lanes >= 32 with (partner's b, own b) = features 8..15
    const auto w0 = __builtin_amdgcn_permlane32_swap(d[2 * s][0], d[2 * s + 1][0], false, false);
    const auto w1 = __builtin_amdgcn_permlane32_swap(d[2 * s][1], d[2 * s + 1][1], false, false);
    u32x4 out;
    out[0] = w0[0];
    out[1] = w1[0];
    out[2] = w0[1];
    out[3] = w1[1];
#if defined(SSKD_PROBE) && SSKD_PROBE_NOSTORE
    asm volatile("" ::"v"(out));
#else
    *reinterpret_cast<u32x4*>(dst + ((int64_t)(s * 64 + lane)) * 8) = out;
#endif
  }
}

// ------------------------------------------------------------------------- //
// embeddings + LayerNorm -> fragment-order X [T_pad, 384]
// one workgroup per 32-token tile; each wave normalises 8 tokens (lanes 0..47 own 8 columns)
// into an LDS image of the tile, which is then copied out linearly.
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void embed_ln_kernel(
    const int* __restrict__ ids, const bf16x8* __restrict__ word, const bf16x8* __restrict__ pos,
    const bf16x8* __restrict__ type0, const float* __restrict__ gamma,
    const float* __restrict__ beta, int B, int S, int S_pad, int vocab, float eps,
    bf16x8* __restrict__ out, const int* __restrict__ seg) {
  __shared__ bf16x8 tile[KSTEPS * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tt = blockIdx.x;
  const bool act = lane < H / 8;
  f32x4 g0, g1, b0, b1;
  if (act) {
    g0 = reinterpret_cast<const f32x4*>(gamma)[lane * 2];
    g1 = reinterpret_cast<const f32x4*>(gamma)[lane * 2 + 1];
    b0 = reinterpret_cast<const f32x4*>(beta)[lane * 2];
    b1 = reinterpret_cast<const f32x4*>(beta)[lane * 2 + 1];
  }
  for (int i = 0; i < 8; ++i) {
    const int r = wave * 8 + i;
    const int tok = tt * 32 + r;
    const int b = tok / S_pad, t = tok - b * S_pad;
    bool real = b < B && t < S;  // wave-uniform
    int tpos = t;
    if (seg && real) {
      // packed rows: the token's segment [lo, hi) inside its row; position ids restart per segment
      const int sw = seg[(int64_t)b * S + t];
      const int lo = sw & 0xffff, hi = (sw >> 16) & 0xffff;
      real = hi > lo;
      tpos = t - lo;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (real && act) {
      int id = ids[(int64_t)b * S + t];
      id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
      const bf16x8 w = word[(int64_t)id * (H / 8) + lane];
      const bf16x8 p = pos[(int64_t)tpos * (H / 8) + lane];
      const bf16x8 ty = type0[lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bf2f(w[j]) + bf2f(ty[j]) + bf2f(p[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / H);
    float q = 0.f;
    if (act) {
#pragma unroll
      for (int j = 0; j < 8; ++j) q += (v[j] - mean) * (v[j] - mean);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * (1.0f / H) + eps);
    if (act) {
      bf16x8 o = zero_bf8();  // padding positions stay exactly zero (finite keys / values)
      if (real) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = (__bf16)((v[j] - mean) * rstd * g0[j] + b0[j]);
          o[4 + j] = (__bf16)((v[4 + j] - mean) * rstd * g1[j] + b1[j]);
        }
      }
      // columns 8 lane .. 8 lane + 7  ->  k-step lane >> 1, lane half lane & 1
      tile[(lane >> 1) * 64 + r + 32 * (lane & 1)] = o;
    }
  }
  __syncthreads();
  bf16x8* dst = out + frag_base(tt, 0, KSTEPS);
  for (int i = threadIdx.x; i < KSTEPS * 64; i += 256) dst[i] = tile[i];
}

// ------------------------------------------------------------------------- //
// shared pieces of the GEMM kernels
// ------------------------------------------------------------------------- //

struct GemmN384Params {
  const bf16x8* x;        // fragment-order [T_pad, 384]: the attention context
  const bf16x8* w;        // tiled [12][24][64] fragments of Wo [384, 384]
  const float* bias;      // [384]
  const __bf16* resid;    // fragment-order [T_pad, 384]
  const float* gamma;
  const float* beta;
  float eps;
};

// LDS-DMA: 16 B per lane, global -> LDS, no registers (LDS address = wave-uniform base + 16 lane)
__device__ inline void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// 16 B per lane through a buffer resource: wave-uniform fragment offset in an SGPR, one 32-bit lane offset - no 64-bit
// address arithmetic and no address registers inside the MFMA bursts
__device__ inline bf16x8 buffer_frag(__amdgpu_buffer_rsrc_t rs, unsigned lane16, unsigned byte_off) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, byte_off, 0));
}
__device__ inline __amdgpu_buffer_rsrc_t weight_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}

// The context rows of token tiles tt0 .. tt0 + 3 -> `img` by LDS-DMA: 96 pieces of 1 KiB, dealt to `n_waves` waves
// (this one is number `w`).  The caller waits (vmcnt) and synchronises the workgroup before anybody reads the image.
__device__ inline void request_context_image(const GemmN384Params& p, int64_t tt0, bf16x8 (*img)[KSTEPS][64], int w,
                                             int n_waves, int lane) {
  const bf16x8* cs = p.x + frag_base(tt0, 0, KSTEPS) + lane;
  bf16x8* flat = &img[0][0][0];
  for (int piece = w; piece < 4 * KSTEPS; piece += n_waves) glds16(cs + piece * 64, flat + piece * 64);
}

// Prologue of the fused MLP: X1 = LN(resid + ctx Wo^T + bo) for the 4 token tiles tt0 .. tt0 + 3 of a 512-thread
// workgroup, left as the fragment-order image `img` (96 KiB of LDS).  Weight-stationary like the MLP itself (round 4):
//   - the context rows arrive as a fragment-order image by LDS-DMA (96 pieces of 1 KiB, no registers), the residual rows
//     and the first Wo fragments are requested in the same breath: ONE memory round trip before the first MFMA (the round-2/3
//     block staged Wo through LDS in six barrier-separated steps and waited a load latency in each: 31-41 k cycles
//     for 9 k cycles of MFMAs, tools/mlp_probe.hip);
//   - wave = (feature group fg: output tiles 3 fg .. 3 fg + 2, token half th: tiles 2 th, 2 th + 1), 6 accumulators; its
//     Wo fragments stream global -> registers through a 12-deep window (each used for both token tiles), the context
//     fragments come from the image (each read feeds 3 MFMAs);
//   - bias + residual + LayerNorm: a token's 384 features live in 4 waves x 2 lanes; the partial sums meet through
//     `stats` ([128 tokens][4 groups], 4 KiB), behind the barrier that also ends everybody's reads of the context image;
//     the normalised rows then overwrite it.
// Every wave of the workgroup must call this (two workgroup barriers inside); the caller adds the one that completes
// the image.  `ctx_requested`: the DMA pieces were requested earlier (the persistent workgroup's previous tail);
// `lane`: the caller's (per-group, opaque) lane id.
__device__ inline void outproj_ln_image(const GemmN384Params& p, int64_t tt0, bf16x8 (*img)[KSTEPS][64], float2* stats,
                                        bool ctx_requested, int lane) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int fg = wave & 3, th = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  const unsigned lane16 = (unsigned)lane * 16u;

  if (!ctx_requested) request_context_image(p, tt0, img, wave, 8, lane);
  bf16x4 rr[3][2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const __bf16* res = p.resid + frag_base(tt0 + 2 * th + t, 0, KSTEPS) * 8;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        rr[j][t][g] = *reinterpret_cast<const bf16x4*>(
            res + ((int64_t)((2 * (3 * fg + j) + (g >> 1)) * 64 + r + 32 * (g & 1))) * 8 + 4 * h);
  }
  // fragment f = 3 s + j (k-step s, tile 3 fg + j) lives in register f % WA
  const __amdgpu_buffer_rsrc_t wrs = weight_rsrc(p.w);
  constexpr int WA = 12, NFR = 3 * KSTEPS;
  auto wfrag = [&](int f) {
    const int s = f / 3, j = f - 3 * s;
    return buffer_frag(wrs, lane16, (unsigned)(((3 * fg + j) * KSTEPS + s) * 1024));
  };
  bf16x8 a[WA];
#pragma unroll
  for (int i = 0; i < WA; ++i) a[i] = wfrag(i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces have landed (and everything requested with them)
  __syncthreads();

  f32x16 acc[3][2];
  {
    const bf16x8* xl = &img[2 * th][0][lane];
    constexpr int BR = 4, NS = 2 * KSTEPS;   // slot n = 2 s + t: context fragment (2 th + t, s)
    auto xat = [](int n) { return ((n & 1) * KSTEPS + (n >> 1)) * 64; };
    bf16x8 xb[BR];
#pragma unroll
    for (int i = 0; i < BR - 1; ++i) xb[i] = xl[xat(i)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int s = n >> 1, t = n & 1;
      if (n + BR - 1 < NS) xb[(n + BR - 1) % BR] = xl[xat(n + BR - 1)];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(3 * s + j) % WA], xb[n % BR], s == 0 ? zero16() : acc[j][t], 0, 0, 0);
      if (t == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (3 * s + j + WA < NFR) a[(3 * s + j) % WA] = wfrag(3 * s + j + WA);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // v = acc + bias + residual; LayerNorm over the token's 384 features: 48 in this lane, 48 in lane ^ 32, 96 per wave
  float sum[2] = {0.f, 0.f}, sq[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + (3 * fg + j) * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[j][t][4 * g + e] + b[e] + bf2f(rr[j][t][g][e]);
          acc[j][t][4 * g + e] = v;
          sum[t] += v;
          sq[t] = fmaf(v, v, sq[t]);
        }
    }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float s1 = pair_sum(sum[t]), s2 = pair_sum(sq[t]);
    if (h == 0) stats[((2 * th + t) * 32 + r) * 4 + fg] = make_float2(s1, s2);
  }
  __syncthreads();   // the statistics are complete, and nobody reads the context image any more
  float mean[2], rstd[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float2 st = stats[((2 * th + t) * 32 + r) * 4 + k];
      s1 += st.x;
      s2 += st.y;
    }
    mean[t] = s1 * (1.0f / H);
    const float var = fmaxf(s2 * (1.0f / H) - mean[t] * mean[t], 0.f);
    rstd[t] = rsqrtf(var + p.eps);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int nt = 3 * fg + j;
    f32x4 ga[4], be[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ga[g] = *reinterpret_cast<const f32x4*>(p.gamma + nt * 32 + 8 * g + 4 * h);
      be[g] = *reinterpret_cast<const f32x4*>(p.beta + nt * 32 + 8 * g + 4 * h);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[g][e] = (acc[j][t][4 * g + e] - mean[t]) * rstd[t] * ga[g][e] + be[g][e];
      store_tile_frag(reinterpret_cast<__bf16*>(&img[2 * th + t][2 * nt][0]), v, lane);
    }
  }
}

// ------------------------------------------------------------------------- //
// fused MLP: X2 = LN(X1 + gelu(X1 W1^T + b1) W2^T + b2) without the [T, 1536] round trip
// ------------------------------------------------------------------------- //
struct MlpParams {
  const bf16x8* x1;      // fragment-order [T_pad, 384]: GEMM input and residual
  const bf16x8* w1;      // tiled [48][24][64]: chunk c = features 32c..32c+31 of W1 [1536, 384]
  const float* b1;       // [1536]
  const bf16x8* w2p;     // chunk-major [48][12][2][64]: lane l, slot j of fragment (c, nt, s2) =
                         // W2[32 nt + (l & 31)][32 c + 16 s2 + 8 (j >> 2) + 4 (l >> 5) + (j & 3)]  (weights.tile_w2_chunked)
  const float* b2;       // [384]
  const float* gamma;
  const float* beta;
  float eps;
  __bf16* out;           // fragment-order [T_pad, 384]
  GemmN384Params outp;   // FUSE_OUTPROJ: X1 = LN(X + ctx Wo^T + bo) is computed in the prologue
  int n_groups;          // T_pad / 128: groups of 128 tokens, walked by gridDim.x persistent workgroups
};

constexpr int MLP_SUPER = FF / 128;  // 12 super-chunks of 128 hidden units

#ifndef SSKD_MLP_XR
#define SSKD_MLP_XR 6   // ring of X1 fragments in the producers' burst (XR - 1 LDS reads in flight)
#endif

#ifdef SSKD_PROBE
// diagnostic build only (tools/mlp_probe.hip): s_memtime stamps of workgroup 0, waves 0 and 4
__device__ unsigned long long g_probe[2][64][4];
#define SSKD_STAMP(role, it, slot)                                                         \
  do {                                                                                     \
    if (blockIdx.x == 0 && pq == 0 && (it) < 64) {                                         \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                                  \
      if (lane == 0) g_probe[role][it][slot] = t_;                                         \
      __builtin_amdgcn_sched_barrier(0);                                                   \
    }                                                                                      \
  } while (0)
#else
#define SSKD_STAMP(role, it, slot) do {} while (0)
#endif

// Workgroup = 8 waves = 128 tokens, WEIGHT-STATIONARY (round 4).  Every weight byte of a 128-hidden-unit super-chunk is
// needed by exactly one wave and goes global -> registers; nothing is staged through LDS (the round-3 kernel kept X1 in
// the producers' registers and walked both weight matrices through LDS: its ds_write_b128 staging was 16.6 % of the step).
//
//   producer p (waves 0-3)  owns hidden tile p of the super-chunk: its W1 tile [32 x 384] is 24 A fragments in registers
//                           (each reloaded for the next super-chunk right after its last use); the B operands are the
//                           X1 fragments of ALL FOUR token tiles, read from a resident 96-KiB LDS image.
//   consumer q (waves 4-7)  owns output features 96 q .. 96 q + 95 of all 128 tokens (192 accumulators): its W2 slice
//                           streams through a 9-fragment register window, the h fragments come from LDS (one read per
//                           three MFMAs).
//
// The two waves of a SIMD share its matrix pipe and its vector issue port, and an in-order wave cannot slip an MFMA
// into the gaps of its partner's MFMA stream (both waves mixing MFMAs and VALU work: 64 cycles per MFMA, round 2).  So
// every super-chunk has two phases, separated by workgroup barriers, in which exactly one wave of each SIMD runs a dense
// burst of 96 MFMAs while its partner does the vector work:
//
//   phase 1   producer: hT[sc] = W1[tile] X1^T, 96 MFMAs (first k-step on the constant 0: no accumulator set-up)
//             consumer: GELU of ITS quarter of the raw half of super-chunk sc - 1 -> h fragments in LDS
//   phase 2   consumer: Y^T += W2[:, sc - 1] hT[sc - 1], 96 MFMAs
//             producer: bias + GELU of half its accumulators -> registers; the other half + bias -> LDS as fp32
//                       (no value is rounded twice); next bias, first X1 fragments of the next burst
//
// Hand-over.  Accumulator element 4g + e of producer lane (r, hp) is hidden unit 8g + 4hp + e of token r.  W2's image
// is permuted on the host so that slot j of the consumer's k-step s2 IS that element with g = 2 s2 + (j >> 2),
// e = j & 3: a producer lane's elements 0..7 / 8..15 are, packed to bf16, exactly ITS OWN lane of the B fragments
// s2 = 0 / 1 - no cross-lane traffic.  The producer finishes s2 = 0 itself and keeps it in registers until the start
// of the next phase 1 (the consumers' burst that reads the buffer has ended by then: single buffer), the s2 = 1
// half crosses as fp32 and consumer q finishes hidden tile q of it.  LDS: X1 image 96 KiB + h fragments 32 KiB + raw half
// 32 KiB = all 160 KiB; biases and LayerNorm parameters come from global memory (L2).
//
// Latencies hidden across the barriers: the producers issue the first XR - 1 image reads of a burst BEFORE the barrier
// that opens it (the image never changes), the consumers start their burst with the fragments they wrote themselves
// (group order rotated by wave) and read those before the barrier too.
//
// FUSE_OUTPROJ: the attention output projection + residual + LayerNorm (X1) of the workgroup's 128 tokens runs as a
// PROLOGUE (outproj_ln_image: the context image takes the X1 image's place first, its scratch is the raw buffer) and
// leaves X1 as the fragment-order image: X1 never reaches HBM.
template <bool FUSE_OUTPROJ>
__global__ __launch_bounds__(512) void fused_mlp_ln_kernel(MlpParams p) {
  __shared__ bf16x8 ximg[4][KSTEPS][64];          // X1 of the workgroup's 128 tokens, fragment order: 96 KiB
  __shared__ bf16x8 hh[2][4][4][64];              // h fragments [k-step s2][hidden tile][token tile][lane]: 32 KiB
  __shared__ f32x4 hraw[4][4][2][64];             // the consumers' half of the GELU work, fp32 (bias added): 32 KiB
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const bool producer = wave < 4;  // wave-uniform
  const int pq = wave & 3;

#ifdef SSKD_PROBE   // [60] = {start, image complete, loop entered, loop left}, [61][0] = end
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  // PERSISTENT workgroup: groups blockIdx.x, blockIdx.x + gridDim.x, ... of 128 tokens.  A workgroup's first and last
  // ~30 % are memory bursts that nothing else on its CU overlaps (one workgroup per CU); walking several groups lets the
  // producers, idle while the consumers finish a group, request the NEXT group's context image under that tail.
  // The two roles never share code, not even the loop over groups: the producers' 96 registers of W1 fragments and the
  // consumers' 192 accumulators must not be live in one block, and a loop body without role branches is what the register
  // allocator keeps spill-free.  Both branches execute the same workgroup barriers per group: 2 inside the prologue, 1 that
  // completes the image, 2 per super-chunk (the consumers run one super-chunk behind: their first two are bare, the producers'
  // last two sit in their tail), 1 for the LayerNorm statistics.
  if (producer) {
    for (int grp = blockIdx.x; grp < p.n_groups; grp += gridDim.x) {
      // the lane id is re-derived per group through an opaque copy: otherwise every per-lane address of the body (a dozen
      // 64-bit pointers) is hoisted out of this loop and lives - spilled - across all of it
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, r = lane & 31, h = lane >> 5;
      const unsigned lane16 = (unsigned)lane * 16u;
      const int64_t tile0 = (int64_t)grp * 4;
      const bool has_next = grp + (int)gridDim.x < p.n_groups;
      if constexpr (FUSE_OUTPROJ) {
        static_assert(sizeof(hraw) >= 128 * 4 * sizeof(float2), "prologue scratch");
        outproj_ln_image(p.outp, tile0, ximg, reinterpret_cast<float2*>(&hraw[0][0][0][0]), grp != (int)blockIdx.x, lane);
      } else {
        const bf16x8* xs = p.x1 + frag_base(tile0, 0, KSTEPS);
        bf16x8* xd = &ximg[0][0][0];
        for (int i = tid; i < 4 * KSTEPS * 64; i += 512) xd[i] = xs[i];
      }
      __syncthreads();  // the X1 image is complete
      const int hp = pq;
      SSKD_STAMP(0, 60, 1);
#ifdef SSKD_PROBE
      if (blockIdx.x == 0 && pq == 0 && lane == 0) g_probe[0][60][0] = t_begin;
#endif
      const __amdgpu_buffer_rsrc_t w1rs = weight_rsrc(p.w1);
      bf16x8 w[KSTEPS];
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) w[s] = buffer_frag(w1rs, lane16, (unsigned)((hp * KSTEPS + s) * 1024));
      f32x4 bias[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(p.b1 + hp * 32 + 8 * g + 4 * h);
      bf16x8 fd[4];   // this wave's finished half of the previous super-chunk
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) fd[tt] = zero_bf8();
      const bf16x8* xl = &ximg[0][0][lane];
      // slot n = 4 s + tt of a burst multiplies fragment (tt, s) of the image
      constexpr int XR = SSKD_MLP_XR, NS = 4 * KSTEPS;
      auto xat = [](int n) { return ((n & 3) * KSTEPS + (n >> 2)) * 64; };
      bf16x8 xb[XR];
#pragma unroll
      for (int i = 0; i < XR - 1; ++i) xb[i] = xl[xat(i)];
      SSKD_STAMP(0, 60, 2);
      for (int it = 0; it < MLP_SUPER; ++it) {   // the consumers run one iteration behind (their last one is peeled below)
        f32x16 acc[4];
        SSKD_STAMP(0, it, 0);
        {
          const unsigned wnext = (unsigned)(((it + 1 < MLP_SUPER ? it + 1 : it) * 4 + hp) * KSTEPS * 1024);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(3);
#pragma unroll
          for (int n = 0; n < NS; ++n) {
            const int s = n >> 2, tt = n & 3;
            if (n + XR - 1 < NS) xb[(n + XR - 1) % XR] = xl[xat(n + XR - 1)];
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], xb[n % XR], s == 0 ? zero16() : acc[tt], 0, 0, 0);
            if (n < 4) hh[0][hp][n][lane] = fd[n];   // the finished half of the previous super-chunk, behind the first MFMAs
            if (tt == 3) w[s] = buffer_frag(w1rs, lane16, wnext + s * 1024);   // last use: fetch the next super-chunk's
            __builtin_amdgcn_sched_barrier(0);   // one slot = one MFMA: keeps every request where it is written
          }
          __builtin_amdgcn_s_setprio(0);
        }
        SSKD_STAMP(0, it, 1);
        __builtin_amdgcn_sched_barrier(0);   // the GELU belongs behind the barrier (the consumers' burst waits on it)
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        SSKD_STAMP(0, it, 2);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
          for (int i = 0; i < 8; ++i) fd[tt][i] = (__bf16)gelu_erf(acc[tt][i] + bias[i >> 2][i & 3]);
          f32x4 r0, r1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            r0[e] = acc[tt][8 + e] + bias[2][e];
            r1[e] = acc[tt][12 + e] + bias[3][e];
          }
          hraw[hp][tt][0][lane] = r0;
          hraw[hp][tt][1][lane] = r1;
        }
        // pin the finished values here, or the compiler sinks their whole computation past the barrier into the burst
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          u32x4 t = __builtin_bit_cast(u32x4, fd[tt]);
          asm volatile("" : "+v"(t));
          fd[tt] = __builtin_bit_cast(bf16x8, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // the next super-chunk's bias and the ring's first fragments: both a whole phase ahead of their use
          const int nx = it + 1 < MLP_SUPER ? it + 1 : it;
#pragma unroll
          for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const f32x4*>(p.b1 + nx * 128 + hp * 32 + 8 * g + 4 * h);
#pragma unroll
          for (int i = 0; i < XR - 1; ++i) xb[i] = xl[xat(i)];
        }
        __builtin_amdgcn_sched_barrier(0);
        SSKD_STAMP(0, it, 3);
        __syncthreads();
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) hh[0][hp][tt][lane] = fd[tt];
      SSKD_STAMP(0, 60, 3);
      // nobody reads the X1 image any more (the consumers took the residual at their set-up): the next group's context rows
      // can land in its place while the consumers run their last burst and the LayerNorm
      if (FUSE_OUTPROJ && has_next)
        request_context_image(p.outp, (int64_t)(grp + (int)gridDim.x) * 4, ximg, hp, 4, lane);
      // Bare barriers: __syncthreads() would make this wave wait for its DMA pieces (vmcnt(0): they write LDS) BEFORE it
      // signals - and the consumers' last burst sits behind that barrier.  The producers' own LDS stores (h fragments)
      // are what the first barrier publishes: lgkmcnt(0) alone.  The pieces are waited for in the next prologue.
      // (Requesting the producers' residual rows and first Wo fragments here as well measured neutral: the consumers'
      // half of those requests still opens the prologue.)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");   // the consumers' LayerNorm statistics meet behind this barrier
      SSKD_STAMP(0, 61, 0);
      if (!FUSE_OUTPROJ && has_next) __syncthreads();   // the consumers are done with the image before the next one is copied in
    }   // groups
  } else {
    for (int grp = blockIdx.x; grp < p.n_groups; grp += gridDim.x) {
      // the lane id is re-derived per group through an opaque copy: otherwise every per-lane address of the body (a dozen
      // 64-bit pointers) is hoisted out of this loop and lives - spilled - across all of it
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, r = lane & 31, h = lane >> 5;
      const unsigned lane16 = (unsigned)lane * 16u;
      const int64_t tile0 = (int64_t)grp * 4;
      const bool has_next = grp + (int)gridDim.x < p.n_groups;
      if constexpr (FUSE_OUTPROJ) {
        static_assert(sizeof(hraw) >= 128 * 4 * sizeof(float2), "prologue scratch");
        outproj_ln_image(p.outp, tile0, ximg, reinterpret_cast<float2*>(&hraw[0][0][0][0]), grp != (int)blockIdx.x, lane);
      } else {
        const bf16x8* xs = p.x1 + frag_base(tile0, 0, KSTEPS);
        bf16x8* xd = &ximg[0][0][0];
        for (int i = tid; i < 4 * KSTEPS * 64; i += 512) xd[i] = xs[i];
      }
      __syncthreads();  // the X1 image is complete
      const int fq = pq;
      SSKD_STAMP(1, 60, 1);
#ifdef SSKD_PROBE
      if (blockIdx.x == 0 && pq == 0 && lane == 0) g_probe[1][60][0] = t_begin;
#endif
      // y starts as bias + residual (the image is X1): the epilogue is the LayerNorm alone
      f32x16 y[3][4];
      f32x4 bb[2][4];   // b2 of feature tile j, requested one tile ahead (one L2 round trip per tile, not per accumulator)
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[0][g] = *reinterpret_cast<const f32x4*>(p.b2 + fq * 96 + 8 * g + 4 * h);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int nt = fq * 3 + j;
        if (j + 1 < 3) {
#pragma unroll
          for (int g = 0; g < 4; ++g) bb[(j + 1) & 1][g] = *reinterpret_cast<const f32x4*>(p.b2 + (nt + 1) * 32 + 8 * g + 4 * h);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const __bf16* res = reinterpret_cast<const __bf16*>(&ximg[tt][0][0]);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(res + ((2 * nt + (g >> 1)) * 64 + r + 32 * (g & 1)) * 8 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[j][tt][4 * g + e] = bb[j & 1][g][e] + bf2f(rr[e]);
          }
          __builtin_amdgcn_sched_barrier(0);   // one tile at a time: hoisted loads would spill the accumulators for good
        }                                      // (reading the residual rows one tile ahead: measured neutral)
      }
      // Group order of this wave: position cs -> (hidden tile cc, k-step s2) = (cs + 2 fq + 1) mod 8, i.e. it starts with
      // the fragments (fq, 1) this wave wrote itself in phase 1.  Fragment f = 3 cs + j of super-chunk sc is
      // w2p[(((4 sc + cc) * 12 + 3 fq + j) * 2 + s2) * 64 + lane]; 24 fragments per super-chunk walk through 9 registers,
      // the sequence padded to 27 so that fragment f always lives in register f % 9.
      const __amdgpu_buffer_rsrc_t w2rs = weight_rsrc(p.w2p);
      const int rot = 2 * fq + 1;
      auto frag_ld = [&](int sc, int f) {
        const int cs = f / 3, j = f - cs * 3, csr = (cs + rot) & 7, cc = csr >> 1, s2 = csr & 1;
        return buffer_frag(w2rs, lane16, (unsigned)((((4 * sc + cc) * 12 + 3 * fq + j) * 2 + s2) * 1024));
      };
      constexpr int NF = 24;
      bf16x8 a[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) a[i] = frag_ld(0, i);
      __syncthreads();   // iteration 0: nothing to consume yet
      __syncthreads();
      for (int it = 1; it <= MLP_SUPER; ++it) {
        SSKD_STAMP(1, it, 0);
        // phase 1: GELU of this wave's share (hidden tile fq, k-step 1) of the previous super-chunk's raw half
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {   // four values at a time: 192 accumulators leave ~25 registers for this
          bf16x8 f;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const f32x4 rw = hraw[fq][tt][hf][lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * hf + e] = (__bf16)gelu_erf(rw[e]);
            __builtin_amdgcn_sched_barrier(0);
          }
          hh[1][fq][tt][lane] = f;
          __builtin_amdgcn_sched_barrier(0);
        }
        // the burst walks (group cs, token tile tt); a group's three W2 fragments stay in registers while the four token
        // tiles' h fragments stream through a ring of HR (one LDS read per three MFMAs, two reads ahead)
        constexpr int HR = 3;
        const bf16x8* hl = &hh[0][0][0][lane];
        auto hidx = [&](int g) {
          const int csr = ((g >> 2) + rot) & 7, tt = g & 3;
          return ((csr & 1) * 16 + (csr >> 1) * 4 + tt) * 64;
        };
        bf16x8 hb[HR];
#pragma unroll
        for (int i = 0; i < HR - 1; ++i) hb[i] = hl[hidx(i)];   // own fragments (same wave: ordered behind the writes above)
        __builtin_amdgcn_sched_barrier(0);
        SSKD_STAMP(1, it, 3);
        __syncthreads();
        SSKD_STAMP(1, it, 1);
        {
          const int sc = it - 1;
          const int scn = sc + 1 < MLP_SUPER ? sc + 1 : sc;
          __builtin_amdgcn_s_setprio(3);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < 32; ++g) {
            const int cs = g >> 2, tt = g & 3;
            if (g + HR - 1 < 32) hb[(g + HR - 1) % HR] = hl[hidx(g + HR - 1)];
#pragma unroll
            for (int j = 0; j < 3; ++j)
              y[j][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(cs * 3 + j) % 9], hb[g % HR], y[j][tt], 0, 0, 0);
            if (tt == 3) {   // the group is finished: refill its registers (f + 9 < 24: this super-chunk's fragment f + 9;
                             // f + 9 >= 27: the next one's fragment f - 18)
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                const int f = cs * 3 + j, fn = f + 9;
                if (fn < NF) a[f % 9] = frag_ld(sc, fn);
                else if (fn >= 27) a[f % 9] = frag_ld(scn, fn - 27);
              }
              if (cs == 7) {   // the pad: fragments 6 .. 8 of the next super-chunk
#pragma unroll
                for (int j = 6; j < 9; ++j) a[j] = frag_ld(scn, j);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          __builtin_amdgcn_s_setprio(0);
        }
        SSKD_STAMP(1, it, 2);
        __syncthreads();
      }
      // epilogue: LayerNorm over the token's 384 features - 48 in this lane, 48 in lane ^ 32, 96 per consumer
      SSKD_STAMP(1, 60, 3);
      float2* const stats = reinterpret_cast<float2*>(&hraw[0][0][0][0]);   // dead by now: [128 tokens][4 consumers]
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            sum += y[j][tt][i];
            sq = fmaf(y[j][tt][i], y[j][tt][i], sq);
          }
        sum = pair_sum(sum);
        sq = pair_sum(sq);
        if (h == 0) stats[(tt * 32 + r) * 4 + fq] = make_float2(sum, sq);
      }
      __syncthreads();
      float mean[4], rstd[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float2 t = stats[(tt * 32 + r) * 4 + k];
          sum += t.x;
          sq += t.y;
        }
        mean[tt] = sum * (1.0f / H);
        const float var = fmaxf(sq * (1.0f / H) - mean[tt] * mean[tt], 0.f);
        rstd[tt] = rsqrtf(var + p.eps);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {   // feature tile outermost: gamma / beta are fetched once per tile, not once per token tile
        const int nt = fq * 3 + j;
        f32x4 ga[4], be[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          ga[g] = *reinterpret_cast<const f32x4*>(p.gamma + nt * 32 + 8 * g + 4 * h);
          be[g] = *reinterpret_cast<const f32x4*>(p.beta + nt * 32 + 8 * g + 4 * h);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          f32x4 v[4];
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[g][e] = (y[j][tt][4 * g + e] - mean[tt]) * rstd[tt] * ga[g][e] + be[g][e];
          store_tile_frag(p.out + frag_base(tile0 + tt, 2 * nt, KSTEPS) * 8, v, lane);
        }
      }
      SSKD_STAMP(1, 61, 0);
      if (!FUSE_OUTPROJ && has_next) __syncthreads();
    }   // groups
  }
}

constexpr int ATT_MAX_S = 512;
constexpr float MASK_NEG = -1.0e30f;

// ------------------------------------------------------------------------- //
// fused QKV projection + attention: ctx = softmax((X Wq^T)(X Wk^T)^T / sqrt(32) + mask) (X Wv^T)
// one workgroup per (batch row, head); Q, K, V never reach HBM
// ------------------------------------------------------------------------- //
struct QkvAttnParams {
  const bf16x8* x;      // fragment-order [T_pad, 384]
  const bf16x8* wqkv;   // tiled [36][24][64]: tile h = Wq head h, 12 + h = Wk, 24 + h = Wv
  const float* bqkv;    // [1152]
  const int* mask;      // [B, S] (unused when PACKED)
  const int* seg;       // PACKED: [B, S] words lo | hi << 16: the token's segment [lo, hi) inside its row
  int B;
  int S;
  int nkt;              // S_pad / 32
  int hpw;              // heads per workgroup (divides 12)
  int spw;              // sequences per workgroup: 8 / nkt when nkt <= 4 (short sequences), else 1
  float q_scale;        // log2(e) / sqrt(32)
  __bf16* ctx;          // fragment-order [T_pad, 384]
};

// Phase 1 (projection), wave = one 32-token tile at a time, activations in 96 registers:
//   Q^T = Wq X^T   accumulators [dim, query]: packed, they ARE the B operand of S^T = K Q^T
//   K^T = Wk X^T   accumulators [dim, key], key on the lane: packed, they ARE K's A-operand
//                  fragments (rows = keys, k = dims in the same accumulator order as Q^T)
//   V   = X Wv^T   (operands swapped) accumulators [key, dim], dim on the lane: packed, they ARE
//                  the A-operand fragments of O^T = V^T P^T (rows = dims, k = keys in accumulator
//                  order, the order P^T's accumulators have) - no transpose anywhere.
// K and V fragments go to LDS (every wave needs every key), Q stays in registers.
// Phase 2: flash-style attention per query tile, online softmax lane-local (query on the lane).
#ifdef SSKD_PROBE
__device__ unsigned long long g_probe_qa[16][8];
#define SSKD_QA_STAMP(hi, slot)                                                            \
  do {                                                                                     \
    if (blockIdx.x == 0 && wave == 0 && (hi) < 16) {                                       \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                                  \
      if (lane == 0) g_probe_qa[hi][slot] = t_;                                            \
      __builtin_amdgcn_sched_barrier(0);                                                   \
    }                                                                                      \
  } while (0)
#else
#define SSKD_QA_STAMP(hi, slot) do {} while (0)
#endif

// TWO_TILES = sequences longer than 256 tokens: every wave owns two query tiles and re-reads their
// activations per head (192 resident registers would not fit); otherwise one tile, read once.
//
// Software pipeline over heads with a role stagger: in step k every wave runs the projection of
// head k (MFMA-bound) and the attention of head k-1 (VALU-bound: exp / max / sum), waves 0-3 in
// that order, waves 4-7 in the opposite order.  Wave w and w + 4 share a SIMD, so each SIMD
// always has one wave on the matrix pipe and one on the vector pipe.  K / V fragments are
// double-buffered in LDS (head k is written while head k-1 is read).
//
// PACKED (rows of <= 8 tiles): a row is a concatenation of whole sequences ("segments") and
// attention is block-diagonal: a query sees the keys of its own segment only.  Per query tile the
// key tiles outside the union of its queries' segments are skipped, tiles inside every query's
// segment need no masking, the others get a per-element range test (key position - lo < length).
template <bool TWO_TILES, bool PACKED = false>
__global__ __launch_bounds__(512) void qkv_attention_kernel(QkvAttnParams p) {
  static_assert(!(TWO_TILES && PACKED), "packed rows hold at most 8 token tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char qa_lds[];
  // T = spw * nkt token tiles live in the workgroup (<= 8, or nkt <= 16 in the TWO_TILES case)
  // [3 weight tiles: 72 KiB][2 x (K frags T*2 KiB, V frags T*2 KiB)][mask bias T*32 f32][bias 2 x 96 f32]
  const int T = p.spw * p.nkt;
  bf16x8* const wlds = reinterpret_cast<bf16x8*>(qa_lds);
  bf16x8* const kvlds = wlds + 3 * WTILE_VEC;  // buffer i: K at i * T * 256, V at + T * 128
  float* const mbias = reinterpret_cast<float*>(kvlds + (TWO_TILES ? 1 : 2) * T * 256);
  float* const bias_lds = mbias + T * 32;
  __shared__ int s_kmax[8];                    // per sequence of the workgroup
  __shared__ int s_partial[ATT_MAX_S / 32];    // per token tile of the workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool late = wave >= 4;
  const int r = lane & 31, h = lane >> 5;
  // One workgroup = `spw` batch rows x `hpw` consecutive heads (hpw = 12 when there are enough
  // rows to fill the chip: the rows' activations are then read once and stay in registers).
  // Short sequences (nkt <= 4 tiles) are packed spw = 8 / nkt to a workgroup so that every wave
  // has a query tile: wave = (sequence sl, tile tq_w) and its LDS tile slot is `wave` itself.
  const int groups = NH / p.hpw;
  const int bg = blockIdx.x / groups, head0 = (blockIdx.x - bg * groups) * p.hpw;
  const int S = p.S, nkt = p.nkt;
  const int sl = TWO_TILES ? 0 : wave / nkt;
  const int tq_w = wave - sl * nkt;
  const bool active = TWO_TILES || (sl < p.spw && bg * p.spw + sl < p.B);
  const int b = bg * p.spw + (active ? sl : 0);

  if (tid < 8) s_kmax[tid] = 0;
  if (tid < ATT_MAX_S / 32) s_partial[tid] = 0;
  __syncthreads();
  for (int i = tid; i < T * 32; i += 512) {  // at most one position per thread (T * 32 <= 512)
    const int si = i / (nkt * 32), pos = i - si * (nkt * 32);
    const int bi = bg * p.spw + si;
    if (PACKED) {
      reinterpret_cast<int*>(mbias)[i] = (bi < p.B && pos < S) ? p.seg[(int64_t)bi * S + pos] : 0;
    } else {
      const bool on = bi < p.B && pos < S && p.mask[(int64_t)bi * S + pos] != 0;
      mbias[i] = on ? 0.f : MASK_NEG;
      if (on) atomicMax(&s_kmax[si], pos / 32 + 1);
      else s_partial[i / 32] = 1;
    }
  }

  // activations of this wave's token tile: resident for every head (single-tile case)
  bf16x8 x0[KSTEPS];
  if (!TWO_TILES) {
    const int tq = active ? tq_w : 0;
    const bf16x8* xs = p.x + frag_base((int64_t)b * nkt + tq, 0, KSTEPS) + lane;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) x0[s] = xs[s * 64];
  }

  // the three weight tiles of a head = 4608 vectors = 9 per thread; staged through registers so
  // that head + 1 is already in flight while head is being processed
  bf16x8 wstage[9];
  auto load_weights = [&](int head) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int v = tid + 512 * i;
      const int which = v / WTILE_VEC, off = v - which * WTILE_VEC;
      wstage[i] = p.wqkv[(int64_t)(which * NH + head) * WTILE_VEC + off];
    }
  };
  load_weights(head0);

  constexpr int NU = TWO_TILES ? 2 : 1;
  bf16x8 qf_new[NU][2], qf_old[NU][2];  // Q^T fragments of head k / head k-1

  // ---- projection of one token tile of head (step k): Q^T -> registers, K / V -> LDS buffer k & 1 ----
  auto project = [&](const bf16x8 (&x)[KSTEPS], int u, int tq, int buf) {
    // three independent accumulator chains per k-step (Q^T, K^T and - operands swapped: A =
    // activations, B = weights - V), weight fragments read PF k-steps ahead through a ring
    f32x16 aq = zero16(), ak = zero16(), av = zero16();
    {
      const bf16x8* wq = wlds + lane;
      const bf16x8* wk = wlds + WTILE_VEC + lane;
      const bf16x8* wv = wlds + 2 * WTILE_VEC + lane;
      constexpr int PF = 3, RING = PF + 1;
      bf16x8 fq[RING], fk[RING], fv[RING];
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        fq[i] = wq[i * 64];
        fk[i] = wk[i * 64];
        fv[i] = wv[i * 64];
      }
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        if (s + PF < KSTEPS) {
          fq[(s + PF) % RING] = wq[(s + PF) * 64];
          fk[(s + PF) % RING] = wk[(s + PF) * 64];
          fv[(s + PF) % RING] = wv[(s + PF) * 64];
        }
        aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s % RING], x[s], aq, 0, 0, 0);
        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[s % RING], x[s], ak, 0, 0, 0);
        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[s], fv[s % RING], av, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // Q^T / K^T: lane = token, acc[4g + e] = dim 8g + 4h + e.  V: lane = dim, acc = keys.
    const float* bl = bias_lds + 96 * buf;
    bf16x8 kf[2], vf[2];
    const float bv = bl[64 + r];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(&bl[8 * g + 4 * h]);
      const f32x4 bk = *reinterpret_cast<const f32x4*>(&bl[32 + 8 * g + 4 * h]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g + e;
        qf_new[u][i >> 3][i & 7] = (__bf16)((aq[i] + bq[e]) * p.q_scale);
        kf[i >> 3][i & 7] = (__bf16)(ak[i] + bk[e]);
        vf[i >> 3][i & 7] = (__bf16)(av[i] + bv);
      }
    }
    bf16x8* kl = kvlds + buf * T * 256;
    bf16x8* vl = kl + T * 128;
    kl[(tq * 2 + 0) * 64 + lane] = kf[0];
    kl[(tq * 2 + 1) * 64 + lane] = kf[1];
    vl[(tq * 2 + 0) * 64 + lane] = vf[0];
    vl[(tq * 2 + 1) * 64 + lane] = vf[1];
  };
  auto project_all = [&](int buf) {
    if (!TWO_TILES) {
      if (active) project(x0, 0, wave, buf);  // LDS tile slot = sl * nkt + tq_w = wave
    } else {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int tq = wave + 8 * u;
        if (tq < nkt) {
          const bf16x8* xs = p.x + frag_base((int64_t)b * nkt + tq, 0, KSTEPS) + lane;
#pragma unroll
          for (int s = 0; s < KSTEPS; ++s) x0[s] = xs[s * 64];
          project(x0, u, tq, buf);
        }
      }
    }
  };

  // ---- attention of this wave's query tiles for `head`, K / V from LDS buffer `buf` ------------
  auto attend = [&](int head, int buf) {
    // K / V fragments, mask bias and flags of this wave's own sequence
    const bf16x8* kl = kvlds + buf * T * 256 + sl * nkt * 128 + lane;
    const bf16x8* vl = kl + T * 128;
    const float* mb_seq = mbias + sl * nkt * 32;
    const int* partial_seq = s_partial + sl * nkt;
    int kmin = 0, kmax = s_kmax[sl];
    int qlo = 0, qlen = 0;  // PACKED: this lane's query segment inside the row
    if (PACKED) {
      const int sw = reinterpret_cast<const int*>(mb_seq)[tq_w * 32 + r];
      qlo = sw & 0xffff;
      qlen = ((sw >> 16) & 0xffff) - qlo;
      // key tiles covered by the segments of this tile's queries (padding queries have none)
      int t_lo = qlen > 0 ? (qlo >> 5) : ATT_MAX_S, t_hi = qlen > 0 ? ((qlo + qlen + 31) >> 5) : 0;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        t_lo = min(t_lo, __shfl_xor(t_lo, o));
        t_hi = max(t_hi, __shfl_xor(t_hi, o));
      }
      kmin = __builtin_amdgcn_readfirstlane(t_lo);
      kmax = __builtin_amdgcn_readfirstlane(t_hi);
      if (kmin > kmax) kmin = kmax = 0;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int qt = (TWO_TILES ? wave : tq_w) + 8 * u;
      if (TWO_TILES ? qt >= nkt : !active) break;
      const bf16x8 qf0 = qf_old[u][0], qf1 = qf_old[u][1];
      f32x16 o = zero16();
      float m = MASK_NEG, l = 0.f;
      // NT key tiles per iteration: their score MFMAs are independent, and one max / rescale
      // decision covers all of them - the per-tile work is a long dependent chain (LDS -> MFMA ->
      // max -> half-wave exchange -> branch -> exp -> convert -> MFMA), so a wave needs this ILP.
      // (Explicitly software-pipelining the next tile's score MFMAs instead measured slower.  So
      // did, on the real workload where the maximum moves often: feeding -m into the score MFMA
      // as its C operand to drop the subtraction, summing the bf16 weights with v_dot2c, reading
      // the K fragments one iteration ahead, and a static s_setprio for the late half - the two
      // waves of a SIMD are latency-bound here and what one gains the other loses.)
      auto tiles = [&](int kt, auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        f32x16 sc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          sc[t] = zero16();
          sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[((kt + t) * 2 + 0) * 64], qf0, sc[t], 0, 0, 0);
          sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[((kt + t) * 2 + 1) * 64], qf1, sc[t], 0, 0, 0);
        }
        // sc[t][4g + e] = score(key 32(kt + t) + 8g + 4h + e, query = lane), in log2 units
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (PACKED) {
            // tiles lying inside every query's segment need no test (padding queries: anything goes)
            const int k0 = (kt + t) * 32;
            const bool inside = qlen <= 0 || (qlo <= k0 && qlo + qlen >= k0 + 32);
            if (!__all(inside)) {
              const unsigned base = (unsigned)(k0 + 4 * h - qlo);
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (base + (unsigned)(8 * g + e) >= (unsigned)qlen) sc[t][4 * g + e] = MASK_NEG;
            }
          } else if (partial_seq[kt + t]) {  // only tiles with masked / padding keys pay for the bias
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 mb = *reinterpret_cast<const f32x4*>(&mb_seq[(kt + t) * 32 + 8 * g + 4 * h]);
#pragma unroll
              for (int e = 0; e < 4; ++e) sc[t][4 * g + e] += mb[e];
            }
          }
        }
        float mt = fmaxf(fmaxf(sc[0][0], sc[0][1]), sc[0][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mt = fmaxf(fmaxf(mt, sc[0][i]), sc[0][i + 1]);
        mt = fmaxf(mt, sc[0][15]);
#pragma unroll
        for (int t = 1; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; i += 2) mt = fmaxf(fmaxf(mt, sc[t][i]), sc[t][i + 1]);
        mt = pair_max(mt);
        if (__any(mt > m)) {
          // the running maximum moves (rare after the first tiles): rescale what is accumulated
          const float m_new = fmaxf(m, mt);
          const float alpha = __builtin_amdgcn_exp2f(m - m_new);
          l *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) o[i] *= alpha;
          m = m_new;
        }
        float ps = 0.f;
        bf16x8 pf[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float e = __builtin_amdgcn_exp2f(sc[t][i] - m);
            ps += e;
            pf[t][i >> 3][i & 7] = (__bf16)e;
          }
        l += ps;
        // O^T[dim, query] += V^T[dim, key] P^T[key, query]; P^T is the accumulator as B operand
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[((kt + t) * 2 + 0) * 64], pf[t][0], o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[((kt + t) * 2 + 1) * 64], pf[t][1], o, 0, 0, 0);
        }
      };
      int kt = kmin;
      for (; kt + 2 <= kmax; kt += 2) tiles(kt, std::integral_constant<int, 2>{});
      if (kt < kmax) tiles(kt, std::integral_constant<int, 1>{});
      l = pair_sum(l);
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[g][e] = o[4 * g + e] * inv;
      store_tile_frag(p.ctx + frag_base((int64_t)b * nkt + qt, 2 * head, KSTEPS) * 8, v, lane);
    }
  };

  for (int k = 0; k <= p.hpw; ++k) {
    const bool has_p1 = k < p.hpw, has_p2 = k >= 1;
    SSKD_QA_STAMP(k, 0);
    __syncthreads();  // step k-1 complete: weights free, K / V of head k-1 complete
    SSKD_QA_STAMP(k, 1);
    if (has_p1) {
#ifdef SSKD_QA_ABL_NOHANDOVER   // timing ablation: only the first head's weights ever reach LDS (results wrong)
      if (k == 0)
#endif
#pragma unroll
      for (int i = 0; i < 9; ++i) wlds[tid + 512 * i] = wstage[i];
      if (tid < 96) bias_lds[96 * (k & 1) + tid] = p.bqkv[(tid >> 5) * H + (head0 + k) * DH + (tid & 31)];
    }
    SSKD_QA_STAMP(k, 2);
    __syncthreads();
    SSKD_QA_STAMP(k, 3);
#ifndef SSKD_QA_ABL_NOLOADS
    if (k + 1 < p.hpw) load_weights(head0 + k + 1);
#endif
    SSKD_QA_STAMP(k, 4);
    if (TWO_TILES) {
      // S > 256: double-buffered K / V would not fit LDS -> plain schedule, one buffer
      if (has_p1) {
        project_all(0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          qf_old[u][0] = qf_new[u][0];
          qf_old[u][1] = qf_new[u][1];
        }
        attend(head0 + k, 0);
      }
      continue;
    }
    if (!late) {
      if (has_p1) project_all(k & 1);
      SSKD_QA_STAMP(k, 5);
      if (has_p2) attend(head0 + k - 1, (k - 1) & 1);
    } else {
      if (has_p2) attend(head0 + k - 1, (k - 1) & 1);
      SSKD_QA_STAMP(k, 5);
      if (has_p1) project_all(k & 1);
    }
    SSKD_QA_STAMP(k, 6);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      qf_old[u][0] = qf_new[u][0];
      qf_old[u][1] = qf_new[u][1];
    }
  }
}

// ------------------------------------------------------------------------- //
// tail: masked mean-pool + L2 normalise straight from the fragment-order hidden states,
// and the row-major un-tiling used by the hidden-state test hook
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(512) void pool_normalize_frag_kernel(const bf16x8* __restrict__ hidden,
                                                                  const int* __restrict__ mask, int S,
                                                                  int nkt, int normalize,
                                                                  float* __restrict__ out) {
  __shared__ float e_lds[H];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.x;
  // wave w owns k-steps w, w + 8, w + 16; lane (r, h) accumulates over the tokens r, r + 32, ...
  float acc[3][8];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[a][j] = 0.f;
  float cnt = 0.f;
  for (int kt = 0; kt < nkt; ++kt) {
    const int t = kt * 32 + r;
    const float w = (t < S && mask[(int64_t)b * S + t] != 0) ? 1.0f : 0.0f;
    cnt += w;
    if (__any(w != 0.f)) {
      const bf16x8* src = hidden + frag_base((int64_t)b * nkt + kt, 0, KSTEPS) + lane;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const bf16x8 v = src[(wave + 8 * a) * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[a][j] = fmaf(w, bf2f(v[j]), acc[a][j]);
      }
    }
  }
  // sum over the 32 tokens of a lane half (xor < 32 stays inside the half)
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[a][j] += __shfl_xor(acc[a][j], o);
  }
  const float n = fmaxf(cnt, 1e-9f);
  if (r == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) e_lds[16 * (wave + 8 * a) + 8 * h + j] = acc[a][j] / n;
  }
  __syncthreads();
  float ss = 0.f;
  if (tid < H) ss = e_lds[tid] * e_lds[tid];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  if (tid < H) {
    float e = e_lds[tid];
    if (normalize) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 6; ++w) tot += red[w];
      e /= fmaxf(sqrtf(tot), 1e-12f);
    }
    out[(int64_t)b * H + tid] = e;
  }
}

// packed rows: one workgroup per SEQUENCE; table entry = { row, lo, hi, output row }
__global__ __launch_bounds__(512) void pool_normalize_packed_kernel(const bf16x8* __restrict__ hidden,
                                                                    const int4* __restrict__ table, int nkt,
                                                                    int normalize, float* __restrict__ out) {
  __shared__ float e_lds[H];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int4 ent = table[blockIdx.x];
  const int row = ent.x, lo = ent.y, hi = ent.z;
  float acc[3][8];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[a][j] = 0.f;
  for (int kt = lo >> 5; kt < ((hi + 31) >> 5); ++kt) {
    const int t = kt * 32 + r;
    const float w = (t >= lo && t < hi) ? 1.0f : 0.0f;
    const bf16x8* src = hidden + frag_base((int64_t)row * nkt + kt, 0, KSTEPS) + lane;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bf16x8 v = src[(wave + 8 * a) * 64];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[a][j] = fmaf(w, bf2f(v[j]), acc[a][j]);
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[a][j] += __shfl_xor(acc[a][j], o);
  }
  const float n = fmaxf((float)(hi - lo), 1e-9f);
  if (r == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) e_lds[16 * (wave + 8 * a) + 8 * h + j] = acc[a][j] / n;
  }
  __syncthreads();
  float ss = 0.f;
  if (tid < H) ss = e_lds[tid] * e_lds[tid];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  if (tid < H) {
    float e = e_lds[tid];
    if (normalize) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 6; ++w) tot += red[w];
      e /= fmaxf(sqrtf(tot), 1e-12f);
    }
    out[(int64_t)ent.w * H + tid] = e;
  }
}

// flat token stream + per-sequence placement -> padded packed rows: ids [R, C] and segment words
// [R, C] (lo | hi << 16), both zero-filled beforehand.  One workgroup per sequence.
__global__ __launch_bounds__(256) void pack_tokens_kernel(const int* __restrict__ flat_ids,
                                                          const int* __restrict__ cu_seqlens,
                                                          const int4* __restrict__ table, int C,
                                                          int* __restrict__ ids, int* __restrict__ seg) {
  const int4 ent = table[blockIdx.x];
  const int src0 = cu_seqlens[blockIdx.x];
  const int len = ent.z - ent.y;
  const int word = ent.y | (ent.z << 16);
  for (int i = threadIdx.x; i < len; i += 256) {
    const int64_t dst = (int64_t)ent.x * C + ent.y + i;
    ids[dst] = flat_ids[src0 + i];
    seg[dst] = word;
  }
}

__global__ __launch_bounds__(256) void untile_hidden_kernel(const __bf16* __restrict__ frag, int B, int S,
                                                            int nkt, __bf16* __restrict__ rows) {
  const int64_t total = (int64_t)B * S * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % H);
    const int64_t bt = i / H;
    const int t = (int)(bt % S), b = (int)(bt / S);
    const int64_t vec = frag_base((int64_t)b * nkt + (t >> 5), col >> 4, KSTEPS) + (t & 31) + 32 * ((col >> 3) & 1);
    rows[i] = frag[vec * 8 + (col & 7)];
  }
}

// ------------------------------------------------------------------------- //
// workspace carve-up
// ------------------------------------------------------------------------- //
struct Workspace {
  __bf16 *xa, *ctx;
  size_t bytes;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline int s_pad_of(int S) { return (S + 31) / 32 * 32; }
inline int64_t t_pad_of(int B, int S) { return ((int64_t)B * s_pad_of(S) + 255) / 256 * 256; }

// Token buffers hold T_pad = roundup(B * S_pad, 256) rows (no kernel bounds-checks).
Workspace carve(void* base, int B, int S) {
  const size_t T = (size_t)t_pad_of(B, S);
  char* pch = static_cast<char*>(base);
  Workspace w{};
  auto take = [&](size_t elems) {
    __bf16* ptr = reinterpret_cast<__bf16*>(pch);
    pch += align256(elems * sizeof(__bf16));
    return ptr;
  };
  w.xa = take(T * H);
  w.ctx = take(T * H);
  w.bytes = (size_t)(pch - static_cast<char*>(base));
  return w;
}

int check_cfg(const sskd_encoder_config* cfg, const sskd_encoder_weights* w, int B, int S) {
  SSKD_REQUIRE(cfg && w, "encoder: null config / weights");
  if (cfg->hidden != H || cfg->heads != NH || cfg->intermediate != FF)
    return sskd::fail(SSKD_ERR_UNSUPPORTED,
                      "encoder: kernels are specialised for hidden=384, heads=12, intermediate=1536 "
                      "(got %d, %d, %d)", cfg->hidden, cfg->heads, cfg->intermediate);
  SSKD_REQUIRE(cfg->layers >= 0 && cfg->vocab_size > 0, "encoder: bad layers / vocab");
  SSKD_REQUIRE(B >= 0 && S >= 1, "encoder: bad shape B=%d S=%d", B, S);
  if (S > ATT_MAX_S || S > cfg->max_positions)
    return sskd::fail(SSKD_ERR_UNSUPPORTED, "encoder: S=%d exceeds max %d", S,
                      ATT_MAX_S < cfg->max_positions ? ATT_MAX_S : cfg->max_positions);
  SSKD_REQUIRE(w->word_emb && w->pos_emb && w->type_emb && w->emb_ln_g && w->emb_ln_b &&
                   (cfg->layers == 0 || w->layers),
               "encoder: null weight pointer");
  return SSKD_OK;
}

// runs embeddings + all layers; returns the (fragment-order) buffer holding the final hidden states
int run_layers(const sskd_encoder_config* cfg, const sskd_encoder_weights* w, const int32_t* d_ids,
               const int32_t* d_mask, int B, int S, const Workspace& ws, hipStream_t st,
               __bf16** final_hidden, const int32_t* d_seg = nullptr) {
  const int Sp = s_pad_of(S), nkt = Sp / 32;
  const int Tpad = (int)t_pad_of(B, S);
  const int n_cus = sskd::cu_count();   // the persistent fused MLP launches one workgroup per CU
  hipLaunchKernelGGL(embed_ln_kernel, dim3(Tpad / 32), dim3(256), 0, st, d_ids,
                     static_cast<const bf16x8*>(w->word_emb), static_cast<const bf16x8*>(w->pos_emb),
                     static_cast<const bf16x8*>(w->type_emb), w->emb_ln_g, w->emb_ln_b, B, S, Sp,
                     cfg->vocab_size, cfg->layer_norm_eps, reinterpret_cast<bf16x8*>(ws.xa), d_seg);
  int rc = sskd::check_launch("embed_ln_kernel");
  if (rc != SSKD_OK) return rc;

  __bf16* x = ws.xa;  // layer input; the fused MLP writes the layer output over it (row-local, in place)
  for (int li = 0; li < cfg->layers; ++li) {
    const sskd_encoder_layer_weights& lw = w->layers[li];
    SSKD_REQUIRE(lw.wqkv && lw.bqkv && lw.wo && lw.bo && lw.ln1_g && lw.ln1_b && lw.w1 && lw.b1 &&
                     lw.w2 && lw.b2 && lw.ln2_g && lw.ln2_b,
                 "encoder: layer %d has a null weight pointer", li);
    QkvAttnParams qa{};
    qa.x = reinterpret_cast<const bf16x8*>(x);
    qa.wqkv = static_cast<const bf16x8*>(lw.wqkv);
    qa.bqkv = lw.bqkv;
    qa.mask = d_mask;
    qa.seg = d_seg;
    qa.B = B;
    qa.S = S;
    qa.nkt = nkt;
    qa.q_scale = LOG2E / sqrtf((float)DH);
    qa.ctx = ws.ctx;
    // short sequences: several to a workgroup, so that all 8 waves own a query tile
    qa.spw = nkt <= 4 ? 8 / nkt : 1;
    const int qa_tiles = qa.spw * nkt;
    const int qa_rows = (B + qa.spw - 1) / qa.spw;  // workgroup rows
    const size_t qa_lds_bytes = 3 * WTILE_VEC * sizeof(bf16x8) +
                          (size_t)qa_tiles * ((nkt > 8 ? 2 : 4) * 128 * sizeof(bf16x8) + 32 * sizeof(float)) + 2 * 96 * sizeof(float);
    // all 12 heads per workgroup once there are enough batch rows to fill the chip; fewer heads
    // per workgroup (more workgroups) for small batches
    qa.hpw = qa_rows >= 256 ? 12 : (qa_rows >= 64 ? 4 : 1);   // (12 stays best under the two-branch forward too: 6 / 4 / 3 heads measured -1.8 / -4 / -6 %)
    GemmN384Params o{};
    o.x = reinterpret_cast<const bf16x8*>(ws.ctx);
    o.w = static_cast<const bf16x8*>(lw.wo);
    o.bias = lw.bo;
    o.resid = x;
    o.gamma = lw.ln1_g;
    o.beta = lw.ln1_b;
    o.eps = cfg->layer_norm_eps;
    auto qa_kernel = nkt > 8 ? qkv_attention_kernel<true, false>
                             : (d_seg ? qkv_attention_kernel<false, true> : qkv_attention_kernel<false, false>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qa_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)qa_lds_bytes);
    hipLaunchKernelGGL(qa_kernel, dim3(qa_rows * (NH / qa.hpw)), dim3(512), qa_lds_bytes, st, qa);
    if ((rc = sskd::check_launch("qkv_attention_kernel")) != SSKD_OK) return rc;

    // (Running this GEMM in the attention workgroup's TAIL - it owns all heads of its 256 tokens -
    // was built and measured: 5.5 % slower end to end on one box (7.92 vs 7.51 ms): every workgroup
    // reaches its tail at the same time and the tail then runs the same bytes with 1 workgroup per
    // CU.  It runs as the PROLOGUE of the fused MLP instead: see fused_mlp_ln_kernel.)

    MlpParams m{};
    m.x1 = nullptr;
    m.w1 = static_cast<const bf16x8*>(lw.w1);
    m.b1 = lw.b1;
    m.w2p = static_cast<const bf16x8*>(lw.w2);
    m.b2 = lw.b2;
    m.gamma = lw.ln2_g;
    m.beta = lw.ln2_b;
    m.eps = cfg->layer_norm_eps;
    m.out = x;
    m.outp = o;
    m.n_groups = (int)(Tpad / 128);
    hipLaunchKernelGGL(fused_mlp_ln_kernel<true>, dim3(std::min(m.n_groups, n_cus)), dim3(512), 0, st, m);
    if ((rc = sskd::check_launch("fused_mlp_ln_kernel")) != SSKD_OK) return rc;
  }
  *final_hidden = x;
  return SSKD_OK;
}

}  // namespace

extern "C" {

size_t sskd_encoder_workspace_bytes(const sskd_encoder_config* cfg, int B, int S) {
  (void)cfg;
  if (B <= 0 || S <= 0) return 0;
  return carve(nullptr, B, S).bytes;
}

}  // extern "C"

namespace {

int prepare(const sskd_encoder_config* cfg, const sskd_encoder_weights* w, int B, int S,
            void* d_workspace, size_t workspace_bytes, Workspace* ws) {
  int rc = check_cfg(cfg, w, B, S);
  if (rc != SSKD_OK) return rc;
  const size_t need = sskd_encoder_workspace_bytes(cfg, B, S);
  if (B > 0 && (!d_workspace || workspace_bytes < need))
    return sskd::fail(SSKD_ERR_WORKSPACE, "encoder: workspace %zu B < required %zu B",
                      workspace_bytes, need);
  if (B > 0) *ws = carve(d_workspace, B, S);
  return SSKD_OK;
}

}  // namespace

extern "C" {

int sskd_encoder_hidden(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                        const int32_t* d_ids, const int32_t* d_mask, int B, int S,
                        void* d_hidden_bf16, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
  Workspace ws{};
  int rc = prepare(cfg, w, B, S, d_workspace, workspace_bytes, &ws);
  if (rc != SSKD_OK || B == 0) return rc;
  SSKD_REQUIRE(d_ids && d_mask && d_hidden_bf16, "encoder_hidden: null pointer");
  hipStream_t st = sskd::as_stream(stream);
  __bf16* fin = nullptr;
  rc = run_layers(cfg, w, d_ids, d_mask, B, S, ws, st, &fin);
  if (rc != SSKD_OK) return rc;
  const int64_t total = (int64_t)B * S * H;
  int64_t blocks = sskd::ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(untile_hidden_kernel, dim3((unsigned)blocks), dim3(256), 0, st, fin, B, S,
                     s_pad_of(S) / 32, static_cast<__bf16*>(d_hidden_bf16));
  return sskd::check_launch("untile_hidden_kernel");
}

namespace {
int forward_rows(const sskd_encoder_config* cfg, const sskd_encoder_weights* w, const int32_t* d_ids,
                 const int32_t* d_mask, int B, int S, int normalize, float* d_out, const Workspace& ws, hipStream_t st) {
  __bf16* fin = nullptr;
  int rc = run_layers(cfg, w, d_ids, d_mask, B, S, ws, st, &fin);
  if (rc != SSKD_OK) return rc;
  hipLaunchKernelGGL(pool_normalize_frag_kernel, dim3(B), dim3(512), 0, st,
                     reinterpret_cast<const bf16x8*>(fin), d_mask, S, s_pad_of(S) / 32, normalize,
                     d_out);
  return sskd::check_launch("pool_normalize_frag_kernel");
}

}  // namespace

// Large batches run as TWO (SSKD_FORWARD_STREAMS: 1 .. 4) parts on as many streams (the caller's and side streams forked
// from / joined back into it with events; under stream capture the side streams join the capture, the graph gets
// branches).  Every kernel of a part fills the chip by itself; what the branches buy is that the parts drift apart: the fused MLP's memory bursts at
// both ends of every 128-token group (one workgroup per CU, nothing else resident) then meet the other half's attention
// or MLP compute instead of 255 other CUs in the same phase (tools/two_stream_probe.py: 6.33 -> 6.17 ms at 512 x 256).
// Rows do not interact, so the result is bit-identical to the one-stream forward.
int sskd_encoder_forward(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                         const int32_t* d_ids, const int32_t* d_mask, int B, int S, int normalize,
                         float* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
  Workspace ws{};
  int rc = prepare(cfg, w, B, S, d_workspace, workspace_bytes, &ws);
  if (rc != SSKD_OK || B == 0) return rc;
  SSKD_REQUIRE(d_ids && d_mask && d_out, "encoder_forward: null pointer");
  hipStream_t st = sskd::as_stream(stream);
  // every part must fill the chip twice over (>= 2 groups of 128 tokens per CU: the measured configuration; four parts of one
  // round each measured +1.5 % where two parts gave +4.8 %) and start on a 256-row boundary
  int parts = sskd::forward_stream_parts();
  while (parts > 1 && !(B % parts == 0 && ((int64_t)(B / parts) * s_pad_of(S)) % 256 == 0 &&
                        (int64_t)(B / parts) * s_pad_of(S) / 128 >= 2 * sskd::cu_count()))
    --parts;
  const int Bp = B / parts;
  const int64_t Tp = (int64_t)Bp * s_pad_of(S);   // rows of one part in the token buffers
  return sskd::run_parts_on_streams(parts, st, [&](int i, hipStream_t s) {
    Workspace wp = ws;
    wp.xa = ws.xa + i * Tp * H;
    wp.ctx = ws.ctx + i * Tp * H;
    return forward_rows(cfg, w, d_ids + (int64_t)i * Bp * S, d_mask + (int64_t)i * Bp * S, Bp, S, normalize,
                        d_out + (int64_t)i * Bp * H, wp, s);
  });
}


// ---- sequence packing ("cu_seqlens" varlen): whole sequences concatenated into rows of C tokens ----

int sskd_pack_plan(const int32_t* lengths, int n_seq, int capacity, int32_t* table, int* n_rows) {
  SSKD_REQUIRE(n_seq >= 0 && lengths && table && n_rows, "pack_plan: null pointer / negative count");
  SSKD_REQUIRE(capacity >= 32 && capacity <= 256 && capacity % 32 == 0,
               "pack_plan: capacity=%d must be a multiple of 32 in [32, 256]", capacity);
  // best-fit decreasing: sequences by decreasing length (counting sort), each into the row whose
  // free space is the smallest that still fits (rows bucketed by free space), else a new row
  std::vector<int> order(n_seq), start(capacity + 2, 0);
  for (int i = 0; i < n_seq; ++i) {
    SSKD_REQUIRE(lengths[i] >= 1 && lengths[i] <= capacity, "pack_plan: length[%d]=%d outside [1, %d]", i,
                 lengths[i], capacity);
    ++start[capacity - lengths[i] + 1];
  }
  for (int f = 0; f <= capacity; ++f) start[f + 1] += start[f];
  for (int i = 0; i < n_seq; ++i) order[start[capacity - lengths[i]]++] = i;
  std::vector<std::vector<int>> by_free(capacity + 1);
  int rows = 0;
  for (int oi = 0; oi < n_seq; ++oi) {
    const int i = order[oi], len = lengths[i];
    int f = len;
    while (f <= capacity && by_free[f].empty()) ++f;
    int row;
    if (f > capacity) {
      row = rows++;
      f = capacity;
    } else {
      row = by_free[f].back();
      by_free[f].pop_back();
    }
    table[4 * i + 0] = row;
    table[4 * i + 1] = capacity - f;
    table[4 * i + 2] = capacity - f + len;
    table[4 * i + 3] = i;
    if (f - len > 0) by_free[f - len].push_back(row);
  }
  *n_rows = rows;
  return SSKD_OK;
}

int sskd_pack_tokens(const int32_t* d_flat_ids, const int32_t* d_cu_seqlens, const int32_t* d_table,
                     int n_seq, int n_rows, int capacity, int32_t* d_ids, int32_t* d_seg, void* stream) {
  SSKD_REQUIRE(n_seq >= 0 && n_rows >= 0 && capacity >= 32 && capacity % 32 == 0, "pack_tokens: bad shape");
  if (n_rows == 0) return SSKD_OK;
  SSKD_REQUIRE(d_ids && d_seg, "pack_tokens: null output");
  hipStream_t st = sskd::as_stream(stream);
  const size_t bytes = (size_t)n_rows * capacity * sizeof(int32_t);
  if (hipMemsetAsync(d_ids, 0, bytes, st) != hipSuccess || hipMemsetAsync(d_seg, 0, bytes, st) != hipSuccess)
    return sskd::fail(SSKD_ERR_HIP, "pack_tokens: hipMemsetAsync failed");
  if (n_seq == 0) return SSKD_OK;
  SSKD_REQUIRE(d_flat_ids && d_cu_seqlens && d_table, "pack_tokens: null input");
  hipLaunchKernelGGL(pack_tokens_kernel, dim3(n_seq), dim3(256), 0, st, d_flat_ids, d_cu_seqlens,
                     reinterpret_cast<const int4*>(d_table), capacity, d_ids, d_seg);
  return sskd::check_launch("pack_tokens_kernel");
}

int sskd_encoder_forward_packed(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                                const int32_t* d_ids, const int32_t* d_seg, int n_rows, int capacity,
                                const int32_t* d_table, int n_seq, int normalize, float* d_out,
                                void* d_workspace, size_t workspace_bytes, void* stream) {
  SSKD_REQUIRE(capacity >= 32 && capacity <= 256 && capacity % 32 == 0,
               "encoder_forward_packed: capacity=%d must be a multiple of 32 in [32, 256]", capacity);
  Workspace ws{};
  int rc = prepare(cfg, w, n_rows, capacity, d_workspace, workspace_bytes, &ws);
  if (rc != SSKD_OK || n_rows == 0 || n_seq == 0) return rc;
  SSKD_REQUIRE(n_seq > 0 && d_ids && d_seg && d_table && d_out, "encoder_forward_packed: null pointer");
  hipStream_t st = sskd::as_stream(stream);
  // the layers in two branches like sskd_encoder_forward (packed rows do not interact either); the pooling reads rows of
  // both parts and runs behind the join
  int parts = sskd::forward_stream_parts();
  while (parts > 1 && !(n_rows % parts == 0 && ((int64_t)(n_rows / parts) * capacity) % 256 == 0 &&
                        (int64_t)(n_rows / parts) * capacity / 128 >= 2 * sskd::cu_count()))
    --parts;
  const int Rp = n_rows / parts;
  const int64_t Tp = (int64_t)Rp * capacity;
  __bf16* fins[sskd::MAX_STREAM_PARTS] = {};
  rc = sskd::run_parts_on_streams(parts, st, [&](int i, hipStream_t s) {
    Workspace wp = ws;
    wp.xa = ws.xa + i * Tp * H;
    wp.ctx = ws.ctx + i * Tp * H;
    return run_layers(cfg, w, d_ids + (int64_t)i * Rp * capacity, nullptr, Rp, capacity, wp, s, &fins[i],
                      d_seg + (int64_t)i * Rp * capacity);
  });
  if (rc != SSKD_OK) return rc;
  __bf16* fin = fins[0];   // the parts' final buffers are consecutive slices of one buffer
  hipLaunchKernelGGL(pool_normalize_packed_kernel, dim3(n_seq), dim3(512), 0, st,
                     reinterpret_cast<const bf16x8*>(fin), reinterpret_cast<const int4*>(d_table),
                     capacity / 32, normalize, d_out);
  return sskd::check_launch("pool_normalize_packed_kernel");
}

}  // extern "C"
