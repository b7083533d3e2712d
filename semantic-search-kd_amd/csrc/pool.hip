// K7: masked mean-pool + L2 normalise, the tail of SentenceTransformer.encode()
// (sentence_transformers.models.Pooling(mean) + Normalize; reference pins:
// tests/test_model_validation.py:80-89 unit norm, configs/kd.yaml:18-19).
//   e = sum_t m_t h_t / max(sum_t m_t, 1e-9);  e /= max(||e||_2, 1e-12)
#include "common.h"

namespace {

constexpr int H = SSKD_DIM;      // 384
constexpr int CG = H / 4;        // 96 column groups of 4
constexpr int TG = 4;            // token groups
constexpr int THREADS = CG * TG; // 384 = 6 waves

__device__ inline float bf16_to_f32(unsigned short v) {
  return __uint_as_float(((unsigned int)v) << 16);
}

template <bool BF16>
__global__ __launch_bounds__(THREADS) void pool_normalize_kernel(const void* __restrict__ hidden,
                                                                 const int* __restrict__ mask,
                                                                 int S, int normalize,
                                                                 float* __restrict__ out) {
  __shared__ float4 part[TG][CG];
  __shared__ float red[12];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int cg = tid % CG, g = tid / CG;
  const int* m = mask + (int64_t)b * S;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float cnt = 0.f;
  for (int t = g; t < S; t += TG) {
    const int mt = m[t];
    if (mt != 0) {
      const float w = (float)mt;
      float4 v;
      if (BF16) {
        const ushort4 r = reinterpret_cast<const ushort4*>(hidden)[((int64_t)b * S + t) * CG + cg];
        v = make_float4(bf16_to_f32(r.x), bf16_to_f32(r.y), bf16_to_f32(r.z), bf16_to_f32(r.w));
      } else {
        v = reinterpret_cast<const float4*>(hidden)[((int64_t)b * S + t) * CG + cg];
      }
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
      cnt += w;
    }
  }
  part[g][cg] = acc;
  if (cg == 0) red[g] = cnt;
  __syncthreads();
  float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
  float ss = 0.f;
  if (g == 0) {
    const float n = fmaxf(red[0] + red[1] + red[2] + red[3], 1e-9f);
    e = part[0][cg];
#pragma unroll
    for (int k = 1; k < TG; ++k) {
      const float4 o = part[k][cg];
      e.x += o.x; e.y += o.y; e.z += o.z; e.w += o.w;
    }
    e.x /= n; e.y /= n; e.z /= n; e.w /= n;
    ss = e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w;
  }
  __syncthreads();  // red[] is reused below
  // block sum of ss over the first 96 threads (waves 0 and 1)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = ss;  // waves 0..5 -> red[4..9]; only waves 0,1 hold g == 0
  __syncthreads();
  if (g == 0) {
    if (normalize) {
      const float nrm = fmaxf(sqrtf(red[4] + red[5]), 1e-12f);
      e.x /= nrm; e.y /= nrm; e.z /= nrm; e.w /= nrm;
    }
    reinterpret_cast<float4*>(out)[(int64_t)b * CG + cg] = e;
  }
}

}  // namespace

extern "C" int sskd_pool_normalize(const void* d_hidden, int hidden_is_bf16, const int32_t* d_mask,
                                   int B, int S, int normalize, float* d_out, void* stream) {
  SSKD_REQUIRE(B >= 0 && S >= 1, "pool_normalize: bad shape B=%d S=%d", B, S);
  if (B == 0) return SSKD_OK;
  SSKD_REQUIRE(d_hidden && d_mask && d_out, "pool_normalize: null pointer");
  if (hidden_is_bf16)
    hipLaunchKernelGGL(pool_normalize_kernel<true>, dim3(B), dim3(THREADS), 0,
                       sskd::as_stream(stream), d_hidden, d_mask, S, normalize, d_out);
  else
    hipLaunchKernelGGL(pool_normalize_kernel<false>, dim3(B), dim3(THREADS), 0,
                       sskd::as_stream(stream), d_hidden, d_mask, S, normalize, d_out);
  return sskd::check_launch("pool_normalize_kernel");
}
