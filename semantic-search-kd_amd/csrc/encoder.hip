// e5-small-v2 shaped BERT encoder forward on gfx950 (bf16 MFMA, fp32 accumulate).
//
// Replaces what the reference's StudentModel delegates to
// sentence_transformers.SentenceTransformer.encode -> transformers.BertModel
// (reference: src/serve/app.py:287,385-389; tests/test_model_validation.py:80-89;
// architecture constants SURVEY.md App. B / configs/kd.yaml:13-19).
//
// All GEMMs are computed TRANSPOSED, D^T[feature, token] = W[feature, k] * X^T[k, token]
// with v_mfma_f32_32x32x16_bf16: weights are the A operand (pre-tiled on the host into
// fragment order, staged through LDS and shared by the workgroup's waves), activations
// are the B operand (token on the lane, loaded once into registers straight from the
// row-major [T, 384] activation matrix).  The 32x32 result then has the token on the
// lane and 16 features in registers, so bias / GELU / residual / LayerNorm / softmax
// are lane-local and an accumulator tile can feed the next MFMA as its B operand
// without leaving registers (attention P*V).
//
// Kernels per layer (hidden H = 384, 12 heads x 32, FFN 1536):
//   gemm_k384_kernel<QKV>   X -> Q (pre-scaled), K head-major [B,12,S,32], V^T [B,12,32,S]
//   attention_kernel        softmax(QK^T/sqrt(32) + mask) V  -> ctx [T,384]
//   gemm_n384_ln_kernel<1>  X1 = LN(X  + ctx Wo^T + bo)
//   gemm_k384_kernel<GELU>  Hh = gelu(X1 W1^T + b1)           [T,1536]
//   gemm_n384_ln_kernel<4>  X2 = LN(X1 + Hh W2^T + b2)
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

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

__device__ inline float bf2f(__bf16 v) { return (float)v; }

// exact-erf GELU (HF "gelu") with erf by Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7
__device__ inline float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-z * z * LOG2E);
  const float erf_abs = 1.0f - p * e;
  const float erf = x < 0.f ? -erf_abs : erf_abs;
  return 0.5f * x * (1.0f + erf);
}

// ------------------------------------------------------------------------- //
// embeddings + LayerNorm: one wave per token, lanes 0..47 own 8 columns each
// ------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void embed_ln_kernel(
    const int* __restrict__ ids, const bf16x8* __restrict__ word, const bf16x8* __restrict__ pos,
    const bf16x8* __restrict__ type0, const float* __restrict__ gamma,
    const float* __restrict__ beta, int T, int S, int vocab, float eps, bf16x8* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= T) return;
  const bool act = lane < H / 8;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (act) {
    int id = ids[tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const bf16x8 w = word[(int64_t)id * (H / 8) + lane];
    const bf16x8 p = pos[(int64_t)(tok % S) * (H / 8) + lane];
    const bf16x8 ty = type0[lane];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(w[i]) + bf2f(ty[i]) + bf2f(p[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.0f / H);
  float q = 0.f;
  if (act) {
#pragma unroll
    for (int i = 0; i < 8; ++i) q += (v[i] - mean) * (v[i] - mean);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q * (1.0f / H) + eps);
  if (act) {
    f32x4 g0 = reinterpret_cast<const f32x4*>(gamma)[lane * 2], g1 = reinterpret_cast<const f32x4*>(gamma)[lane * 2 + 1];
    f32x4 b0 = reinterpret_cast<const f32x4*>(beta)[lane * 2], b1 = reinterpret_cast<const f32x4*>(beta)[lane * 2 + 1];
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = (__bf16)((v[i] - mean) * rstd * g0[i] + b0[i]);
      o[4 + i] = (__bf16)((v[4 + i] - mean) * rstd * g1[i] + b1[i]);
    }
    out[(int64_t)tok * (H / 8) + lane] = o;
  }
}

// ------------------------------------------------------------------------- //
// shared pieces of the two GEMM kernels
// ------------------------------------------------------------------------- //

// B-operand fragments of one token row: x[s] = X[tok][16s + 8h .. +7]
__device__ inline void load_x_frags(bf16x8 (&x)[KSTEPS], const __bf16* __restrict__ row, int h,
                                    bool valid) {
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    if (valid) {
      x[s] = *reinterpret_cast<const bf16x8*>(row + 16 * s + 8 * h);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[s][i] = (__bf16)0.f;
    }
  }
}

// one [32 features x 384 k] weight tile: 24 MFMAs against the register-resident x
__device__ inline f32x16 tile_mfma(const bf16x8* __restrict__ wlds, const bf16x8 (&x)[KSTEPS],
                                   f32x16 acc, int lane) {
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlds[s * 64 + lane], x[s], acc, 0, 0, 0);
  return acc;
}

enum { EPI_QKV = 0, EPI_GELU = 1 };

struct GemmK384Params {
  const __bf16* x;       // [T, 384] row-major
  const bf16x8* w;       // tiled [N/32][24][64] fragments
  const float* bias;     // [N]
  int T;
  int N;                 // multiple of 32
  int S;                 // sequence length (QKV epilogue)
  float q_scale;         // folded into Q: log2(e) / sqrt(32)
  __bf16* out;           // GELU: [T, N] row-major
  __bf16* q;             // QKV: [B, 12, S, 32]
  __bf16* k;             //      [B, 12, S, 32]
  __bf16* vt;            //      [B, 12, 32, S]
};

// K = 384 GEMM, activations stationary in registers, weight tiles streamed through LDS.
// Workgroup = 8 waves = 256 tokens; wave w owns tokens 32w..32w+31 of the block.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_k384_kernel(GemmK384Params p) {
  __shared__ bf16x8 wlds[2][WTILE_VEC];  // 2 x 24 KiB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tok = blockIdx.x * 256 + wave * 32 + r;
  const bool valid = tok < p.T;

  bf16x8 x[KSTEPS];
  load_x_frags(x, p.x + (int64_t)(valid ? tok : 0) * H, h, valid);

  const int n_tiles = p.N / 32;
  // prologue: stage tile 0
  {
    const bf16x8* src = p.w;
#pragma unroll
    for (int i = 0; i < 3; ++i) wlds[0][tid + 512 * i] = src[tid + 512 * i];
  }
  __syncthreads();

  for (int nt = 0; nt < n_tiles; ++nt) {
    const int cur = nt & 1;
    bf16x8 stage[3];
    const bool more = nt + 1 < n_tiles;
    if (more) {
      const bf16x8* src = p.w + (int64_t)(nt + 1) * WTILE_VEC;
#pragma unroll
      for (int i = 0; i < 3; ++i) stage[i] = src[tid + 512 * i];
    }
    f32x16 acc = tile_mfma(wlds[cur], x, zero16(), lane);
    if (more) {
#pragma unroll
      for (int i = 0; i < 3; ++i) wlds[cur ^ 1][tid + 512 * i] = stage[i];
    }

    // epilogue: lane = token, acc[4g + e] = feature 32nt + 8g + 4h + e
    if (valid) {
      if (EPI == EPI_GELU) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nt * 32 + 8 * g + 4 * h;
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(acc[4 * g + e] + b[e]);
          *reinterpret_cast<bf16x4*>(p.out + (int64_t)tok * p.N + n) = __builtin_convertvector(v, bf16x4);
        }
      } else {
        const int which = nt / NH;  // 0 = Q, 1 = K, 2 = V; one tile == one head (32 dims)
        const int head = nt - which * NH;
        const int b_idx = tok / p.S, s_idx = tok - b_idx * p.S;
        const int64_t bh = (int64_t)b_idx * NH + head;
        if (which < 2) {
          __bf16* dst = (which == 0 ? p.q : p.k) + (bh * p.S + s_idx) * DH;
          const float sc = which == 0 ? p.q_scale : 1.0f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * h;
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + nt * 32 + d);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc[4 * g + e] + b[e]) * sc;
            *reinterpret_cast<bf16x4*>(dst + d) = __builtin_convertvector(v, bf16x4);
          }
        } else {
          __bf16* dst = p.vt + bh * DH * p.S + s_idx;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int d = 8 * g + 4 * h + e;
              dst[(int64_t)d * p.S] = (__bf16)(acc[4 * g + e] + p.bias[nt * 32 + d]);
            }
        }
      }
    }
    __syncthreads();
  }
}

struct GemmN384Params {
  const __bf16* x;        // [T, 384 * KC] row-major
  const bf16x8* w;        // tiled [12][24 * KC][64] fragments
  const float* bias;      // [384]
  const __bf16* resid;    // [T, 384]
  const float* gamma;
  const float* beta;
  float eps;
  int T;
  __bf16* out;            // [T, 384]
};

// N = 384 GEMM (K = 384 * KC) with fused bias + residual + LayerNorm epilogue.
// Workgroup = 4 waves = 128 tokens; every wave keeps all 12 output tiles (its 32 tokens'
// complete rows) in 192 accumulator registers, so LayerNorm never leaves the lane pair.
template <int KC>
__global__ __launch_bounds__(256) void gemm_n384_ln_kernel(GemmN384Params p) {
  __shared__ bf16x8 wlds[2][WTILE_VEC];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tok = blockIdx.x * 128 + wave * 32 + r;
  const bool valid = tok < p.T;
  constexpr int KTOT = KSTEPS * KC;  // k-steps per output tile
  constexpr int NT = H / 32;         // 12

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();

  // weight tile (nt, kc) = fragments [nt][24kc .. 24kc+23]
  auto tile_src = [&](int it) {
    const int kc = it / NT, nt = it - kc * NT;
    return p.w + ((int64_t)nt * KTOT + KSTEPS * kc) * 64;
  };
  {
    const bf16x8* src = tile_src(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) wlds[0][tid + 256 * i] = src[tid + 256 * i];
  }
  __syncthreads();

#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    // compiler-only barrier: keep the next chunk's 96 registers of activation loads from
    // being hoisted above the previous chunk's MFMAs (the file is already full)
    asm volatile("" ::: "memory");
    bf16x8 x[KSTEPS];
    load_x_frags(x, p.x + (int64_t)(valid ? tok : 0) * (H * KC) + H * kc, h, valid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int it = kc * NT + nt;
      const int cur = it & 1;
      bf16x8 stage[6];
      const bool more = it + 1 < KC * NT;
      if (more) {
        const bf16x8* src = tile_src(it + 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) stage[i] = src[tid + 256 * i];
      }
      acc[nt] = tile_mfma(wlds[cur], x, acc[nt], lane);
      if (more) {
#pragma unroll
        for (int i = 0; i < 6; ++i) wlds[cur ^ 1][tid + 256 * i] = stage[i];
      }
      __syncthreads();
    }
  }

  // epilogue: v = acc + bias + residual; LayerNorm over the 384 features of the token,
  // 192 of which live in this lane and 192 in lane ^ 32
  asm volatile("" ::: "memory");
  const __bf16* res = p.resid + (int64_t)(valid ? tok : 0) * H;
  float sum = 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = nt * 32 + 8 * g + 4 * h;
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
      const bf16x4 rr = *reinterpret_cast<const bf16x4*>(res + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[nt][4 * g + e] + b[e] + bf2f(rr[e]);
        acc[nt][4 * g + e] = v;
        sum += v;
      }
    }
  sum += __shfl_xor(sum, 32);
  const float mean = sum * (1.0f / H);
  float sq = 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float d = acc[nt][i] - mean;
      sq += d * d;
    }
  sq += __shfl_xor(sq, 32);
  const float rstd = rsqrtf(sq * (1.0f / H) + p.eps);
  if (valid) {
    __bf16* dst = p.out + (int64_t)tok * H;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nt * 32 + 8 * g + 4 * h;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + n);
        const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + n);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[nt][4 * g + e] - mean) * rstd * ga[e] + be[e];
        *reinterpret_cast<bf16x4*>(dst + n) = __builtin_convertvector(v, bf16x4);
      }
  }
}

// ------------------------------------------------------------------------- //
// attention: one workgroup per (batch row, head, block of 256 queries)
// ------------------------------------------------------------------------- //
struct AttnParams {
  const __bf16* q;    // [B, 12, S, 32], pre-scaled by log2(e)/sqrt(32)
  const __bf16* k;    // [B, 12, S, 32]
  const __bf16* vt;   // [B, 12, 32, S]
  const int* mask;    // [B, S] (1 = attend)
  int S;
  __bf16* ctx;        // [B*S, 384]
};

constexpr int ATT_MAX_S = 512;
constexpr float MASK_NEG = -1.0e30f;

__global__ __launch_bounds__(512) void attention_kernel(AttnParams p) {
  // fragment-ordered K and V^T of this (b, head): per 32-key tile 2 + 2 fragments of 1 KiB
  __shared__ bf16x8 klds[ATT_MAX_S / 32 * 2 * 64];
  __shared__ bf16x8 vlds[ATT_MAX_S / 32 * 2 * 64];
  __shared__ __attribute__((aligned(16))) float mbias[ATT_MAX_S];
  __shared__ int s_kmax;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x;  // b * 12 + head
  const int b = bh / NH, head = bh - b * NH;
  const int S = p.S;
  const int n_ktiles = (S + 31) / 32;
  const __bf16* kg = p.k + (int64_t)bh * S * DH;
  const __bf16* vg = p.vt + (int64_t)bh * DH * S;

  if (tid == 0) s_kmax = 0;
  __syncthreads();
  // mask bias + last tile that holds an attended key
  int local_max = 0;
  for (int i = tid; i < n_ktiles * 32; i += 512) {
    const bool on = i < S && p.mask[(int64_t)b * S + i] != 0;
    mbias[i] = on ? 0.f : MASK_NEG;
    if (on) local_max = i / 32 + 1;
  }
  if (local_max) atomicMax(&s_kmax, local_max);
  // stage K fragments: A[row = key][k = dim]: lane (r, h) of step s holds K[32kt + r][16s + 8h ..+7]
  for (int i = tid; i < n_ktiles * 2 * 64; i += 512) {
    const int l = i & 63, s = (i >> 6) & 1, kt = i >> 7;
    const int key = kt * 32 + (l & 31);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    if (key < S) v = *reinterpret_cast<const bf16x8*>(kg + (int64_t)key * DH + 16 * s + 8 * (l >> 5));
    klds[i] = v;
  }
  // stage V^T fragments: A[row = dim][k = key] in accumulator order:
  // element j of lane (r, h), step s2 <- V^T[dim r][key 32kt + 16 s2 + 8 (j >> 2) + 4h + (j & 3)]
  for (int i = tid; i < n_ktiles * 2 * 64; i += 512) {
    const int l = i & 63, s2 = (i >> 6) & 1, kt = i >> 7;
    const int d = l & 31, hh = l >> 5;
    const int key0 = kt * 32 + 16 * s2 + 4 * hh;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = key0 + 8 * (e >> 2) + (e & 3);
      v[e] = key < S ? vg[(int64_t)d * S + key] : (__bf16)0.f;
    }
    vlds[i] = v;
  }
  __syncthreads();
  const int kmax = s_kmax;

  for (int qb = 0; qb * 256 < S; ++qb) {
    const int qrow = qb * 256 + wave * 32 + r;
    const bool valid = qrow < S;
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (valid) {
        qf[s] = *reinterpret_cast<const bf16x8*>(p.q + ((int64_t)bh * S + qrow) * DH + 16 * s + 8 * h);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)0.f;
      }
    }
    f32x16 o = zero16();
    float m = MASK_NEG, l = 0.f;
    for (int kt = 0; kt < kmax; ++kt) {
      f32x16 sc = zero16();
      sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(klds[(kt * 2 + 0) * 64 + lane], qf[0], sc, 0, 0, 0);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(klds[(kt * 2 + 1) * 64 + lane], qf[1], sc, 0, 0, 0);
      // sc[4g + e] = score(key 32kt + 8g + 4h + e, query = lane), in log2 units
      float mt = MASK_NEG;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 mb = *reinterpret_cast<const f32x4*>(&mbias[kt * 32 + 8 * g + 4 * h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sc[4 * g + e] += mb[e];
          mt = fmaxf(mt, sc[4 * g + e]);
        }
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32));
      const float m_new = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      float ps = 0.f;
      bf16x8 pf[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = __builtin_amdgcn_exp2f(sc[i] - m_new);
        ps += e;
        pf[i >> 3][i & 7] = (__bf16)e;
      }
      l = l * alpha + ps;
      m = m_new;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] *= alpha;
      // O^T[dim, query] += V^T[dim, key] P^T[key, query]; P^T is the accumulator as B operand
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlds[(kt * 2 + 0) * 64 + lane], pf[0], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlds[(kt * 2 + 1) * 64 + lane], pf[1], o, 0, 0, 0);
    }
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (valid) {
      __bf16* dst = p.ctx + ((int64_t)b * S + qrow) * H + head * DH;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[4 * g + e] * inv;
        *reinterpret_cast<bf16x4*>(dst + 8 * g + 4 * h) = __builtin_convertvector(v, bf16x4);
      }
    }
  }
}

// ------------------------------------------------------------------------- //
// workspace carve-up
// ------------------------------------------------------------------------- //
struct Workspace {
  __bf16 *xa, *xb, *q, *k, *vt, *ctx, *ffn;
  size_t bytes;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

Workspace carve(void* base, int B, int S) {
  const size_t T = (size_t)B * S;
  char* pch = static_cast<char*>(base);
  Workspace w{};
  auto take = [&](size_t elems) {
    __bf16* ptr = reinterpret_cast<__bf16*>(pch);
    pch += align256(elems * sizeof(__bf16));
    return ptr;
  };
  w.xa = take(T * H);
  w.xb = take(T * H);
  w.q = take(T * H);
  w.k = take(T * H);
  w.vt = take(T * H);
  w.ctx = take(T * H);
  w.ffn = take(T * FF);
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

// runs embeddings + all layers; returns the buffer holding the final hidden states
int run_layers(const sskd_encoder_config* cfg, const sskd_encoder_weights* w, const int32_t* d_ids,
               const int32_t* d_mask, int B, int S, const Workspace& ws, hipStream_t st,
               __bf16** final_hidden) {
  const int T = B * S;
  hipLaunchKernelGGL(embed_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, st, d_ids,
                     static_cast<const bf16x8*>(w->word_emb), static_cast<const bf16x8*>(w->pos_emb),
                     static_cast<const bf16x8*>(w->type_emb), w->emb_ln_g, w->emb_ln_b, T, S,
                     cfg->vocab_size, cfg->layer_norm_eps, reinterpret_cast<bf16x8*>(ws.xa));
  int rc = sskd::check_launch("embed_ln_kernel");
  if (rc != SSKD_OK) return rc;

  __bf16* x = ws.xa;
  __bf16* x1 = ws.xb;
  for (int li = 0; li < cfg->layers; ++li) {
    const sskd_encoder_layer_weights& lw = w->layers[li];
    SSKD_REQUIRE(lw.wqkv && lw.bqkv && lw.wo && lw.bo && lw.ln1_g && lw.ln1_b && lw.w1 && lw.b1 &&
                     lw.w2 && lw.b2 && lw.ln2_g && lw.ln2_b,
                 "encoder: layer %d has a null weight pointer", li);
    GemmK384Params g{};
    g.x = x;
    g.w = static_cast<const bf16x8*>(lw.wqkv);
    g.bias = lw.bqkv;
    g.T = T;
    g.N = 3 * H;
    g.S = S;
    g.q_scale = LOG2E / sqrtf((float)DH);
    g.q = ws.q;
    g.k = ws.k;
    g.vt = ws.vt;
    hipLaunchKernelGGL(gemm_k384_kernel<EPI_QKV>, dim3((T + 255) / 256), dim3(512), 0, st, g);
    if ((rc = sskd::check_launch("gemm_k384_kernel<QKV>")) != SSKD_OK) return rc;

    AttnParams a{};
    a.q = ws.q;
    a.k = ws.k;
    a.vt = ws.vt;
    a.mask = d_mask;
    a.S = S;
    a.ctx = ws.ctx;
    hipLaunchKernelGGL(attention_kernel, dim3(B * NH), dim3(512), 0, st, a);
    if ((rc = sskd::check_launch("attention_kernel")) != SSKD_OK) return rc;

    GemmN384Params o{};
    o.x = ws.ctx;
    o.w = static_cast<const bf16x8*>(lw.wo);
    o.bias = lw.bo;
    o.resid = x;
    o.gamma = lw.ln1_g;
    o.beta = lw.ln1_b;
    o.eps = cfg->layer_norm_eps;
    o.T = T;
    o.out = x1;
    hipLaunchKernelGGL(gemm_n384_ln_kernel<1>, dim3((T + 127) / 128), dim3(256), 0, st, o);
    if ((rc = sskd::check_launch("gemm_n384_ln_kernel<1>")) != SSKD_OK) return rc;

    GemmK384Params f{};
    f.x = x1;
    f.w = static_cast<const bf16x8*>(lw.w1);
    f.bias = lw.b1;
    f.T = T;
    f.N = FF;
    f.S = S;
    f.out = ws.ffn;
    hipLaunchKernelGGL(gemm_k384_kernel<EPI_GELU>, dim3((T + 255) / 256), dim3(512), 0, st, f);
    if ((rc = sskd::check_launch("gemm_k384_kernel<GELU>")) != SSKD_OK) return rc;

    GemmN384Params d{};
    d.x = ws.ffn;
    d.w = static_cast<const bf16x8*>(lw.w2);
    d.bias = lw.b2;
    d.resid = x1;
    d.gamma = lw.ln2_g;
    d.beta = lw.ln2_b;
    d.eps = cfg->layer_norm_eps;
    d.T = T;
    d.out = x;
    hipLaunchKernelGGL(gemm_n384_ln_kernel<4>, dim3((T + 127) / 128), dim3(256), 0, st, d);
    if ((rc = sskd::check_launch("gemm_n384_ln_kernel<4>")) != SSKD_OK) return rc;
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

int sskd_encoder_hidden(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                        const int32_t* d_ids, const int32_t* d_mask, int B, int S,
                        void* d_hidden_bf16, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
  int rc = check_cfg(cfg, w, B, S);
  if (rc != SSKD_OK) return rc;
  if (B == 0) return SSKD_OK;
  SSKD_REQUIRE(d_ids && d_mask && d_hidden_bf16, "encoder_hidden: null pointer");
  const size_t need = sskd_encoder_workspace_bytes(cfg, B, S);
  if (!d_workspace || workspace_bytes < need)
    return sskd::fail(SSKD_ERR_WORKSPACE, "encoder: workspace %zu B < required %zu B",
                      workspace_bytes, need);
  const Workspace ws = carve(d_workspace, B, S);
  hipStream_t st = sskd::as_stream(stream);
  __bf16* fin = nullptr;
  rc = run_layers(cfg, w, d_ids, d_mask, B, S, ws, st, &fin);
  if (rc != SSKD_OK) return rc;
  if (hipMemcpyAsync(d_hidden_bf16, fin, (size_t)B * S * H * sizeof(__bf16),
                     hipMemcpyDeviceToDevice, st) != hipSuccess)
    return sskd::fail(SSKD_ERR_HIP, "encoder_hidden: copy failed");
  return SSKD_OK;
}

int sskd_encoder_forward(const sskd_encoder_config* cfg, const sskd_encoder_weights* w,
                         const int32_t* d_ids, const int32_t* d_mask, int B, int S, int normalize,
                         float* d_out, void* d_workspace, size_t workspace_bytes, void* stream) {
  int rc = check_cfg(cfg, w, B, S);
  if (rc != SSKD_OK) return rc;
  if (B == 0) return SSKD_OK;
  SSKD_REQUIRE(d_ids && d_mask && d_out, "encoder_forward: null pointer");
  const size_t need = sskd_encoder_workspace_bytes(cfg, B, S);
  if (!d_workspace || workspace_bytes < need)
    return sskd::fail(SSKD_ERR_WORKSPACE, "encoder: workspace %zu B < required %zu B",
                      workspace_bytes, need);
  const Workspace ws = carve(d_workspace, B, S);
  hipStream_t st = sskd::as_stream(stream);
  __bf16* fin = nullptr;
  rc = run_layers(cfg, w, d_ids, d_mask, B, S, ws, st, &fin);
  if (rc != SSKD_OK) return rc;
  return sskd_pool_normalize(fin, 1, d_mask, B, S, normalize, d_out, stream);
}

}  // extern "C"
