// Knowledge-distillation losses of the reference (src/kd/losses.py) and their gradient with respect
// to the student scores, for [B, D] score matrices (D <= 64: the reference trains on D = 9, one
// positive + hard negatives per query).  One wave per row keeps a row in registers (lane = document);
// a second, single-workgroup kernel adds the per-row terms in a fixed order, so results are
// bit-reproducible.  All arithmetic is fp32, like the reference's torch code.
//
//   margin-MSE   :35-60    mean_{b,j} ( (s - max_j s) - (t/T - max_j t/T) )^2
//   listwise KD  :81-106   T^2 / B * sum_b KL( softmax(t/T) || softmax(s/T) )
//   contrastive  :127-149  - 1/B sum_b log_softmax(s / tau)[b, 0]
//   combined     :219-252  w_mm * MM + w_lk * LK + w_c * C
#include "common.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>

#include "sskd_amd.h"

namespace {

struct KdParams {
  const float* s;
  const float* t;
  int B;
  int D;
  float T;
  float tau;
  float w_mm, w_lk, w_c;
  float* rows;    // [B][3]: sum_j r^2, KL row, -log p(positive)
  float* losses;  // [4]: total, margin-MSE, listwise, contrastive
  float* grad;    // [B][D] d total / d s, or null
};

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

__global__ __launch_bounds__(256) void kd_loss_rows_kernel(KdParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.B) return;
  const bool valid = lane < p.D;
  const float s = valid ? p.s[(int64_t)row * p.D + lane] : -INFINITY;
  const float ts = valid ? p.t[(int64_t)row * p.D + lane] / p.T : -INFINITY;

  const float smax = wave_max(s), tmax = wave_max(ts);
  // torch.max(dim) hands the maximum's gradient to one index: the first maximal one
  const unsigned long long at_max = __ballot(valid && s == smax);
  const int arg = __ffsll((long long)at_max) - 1;

  // margin-MSE
  const float r = valid ? (s - smax) - (ts - tmax) : 0.f;
  const float sum_r2 = wave_sum(r * r), sum_r = wave_sum(r);

  // listwise: log-softmax of s / T and t / T
  const float a = s / p.T, amax = smax / p.T;
  const float ea = valid ? expf(a - amax) : 0.f;
  const float log_za = logf(wave_sum(ea));
  const float ls = a - amax - log_za;
  const float et = valid ? expf(ts - tmax) : 0.f;
  const float zt = wave_sum(et);
  const float lt = ts - tmax - logf(zt);
  const float pt = et / zt;
  const float kl = wave_sum(valid ? pt * (lt - ls) : 0.f);

  // contrastive: positive = document 0
  const float c = s / p.tau, cmax = smax / p.tau;
  const float ec = valid ? expf(c - cmax) : 0.f;
  const float zc = wave_sum(ec);
  const float lp = c - cmax - logf(zc);
  const float nll = -__shfl(lp, 0);

  if (lane == 0) {
    p.rows[row * 3 + 0] = sum_r2;
    p.rows[row * 3 + 1] = kl;
    p.rows[row * 3 + 2] = nll;
  }
  if (p.grad && valid) {
    const float inv_bd = 1.0f / ((float)p.B * (float)p.D), inv_b = 1.0f / (float)p.B;
    float g = p.w_mm * 2.0f * inv_bd * (r - (lane == arg ? sum_r : 0.f));
    g += p.w_lk * p.T * inv_b * (expf(ls) - pt);
    g += p.w_c * inv_b / p.tau * (ec / zc - (lane == 0 ? 1.0f : 0.f));
    p.grad[(int64_t)row * p.D + lane] = g;
  }
}

__global__ __launch_bounds__(1024) void kd_loss_finish_kernel(KdParams p) {
  __shared__ float part[16][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int r = tid; r < p.B; r += 1024) {  // fixed strides and a fixed tree: reproducible
    a0 += p.rows[r * 3 + 0];
    a1 += p.rows[r * 3 + 1];
    a2 += p.rows[r * 3 + 2];
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) {
    part[wave][0] = a0;
    part[wave][1] = a1;
    part[wave][2] = a2;
  }
  __syncthreads();
  if (tid == 0) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < 16; ++w) {
      s0 += part[w][0];
      s1 += part[w][1];
      s2 += part[w][2];
    }
    const float mm = s0 / ((float)p.B * (float)p.D);
    const float lk = s1 / (float)p.B * (p.T * p.T);
    const float c = s2 / (float)p.B;
    p.losses[0] = p.w_mm * mm + p.w_lk * lk + p.w_c * c;
    p.losses[1] = mm;
    p.losses[2] = lk;
    p.losses[3] = c;
  }
}

}  // namespace

extern "C" int sskd_kd_loss(const float* d_student, const float* d_teacher, int batch, int n_docs,
                            float temperature, float contrastive_temperature, float w_margin_mse,
                            float w_listwise, float w_contrastive, float* d_losses, float* d_grad,
                            float* d_row_workspace, void* stream) {
  SSKD_REQUIRE(batch >= 1, "kd_loss: batch=%d < 1", batch);
  SSKD_REQUIRE(n_docs >= 1 && n_docs <= 64, "kd_loss: n_docs=%d outside [1, 64]", n_docs);
  SSKD_REQUIRE(temperature > 0.f && contrastive_temperature > 0.f, "kd_loss: temperatures must be positive");
  SSKD_REQUIRE(d_student && d_teacher && d_losses && d_row_workspace, "kd_loss: null pointer");
  KdParams p{};
  p.s = d_student;
  p.t = d_teacher;
  p.B = batch;
  p.D = n_docs;
  p.T = temperature;
  p.tau = contrastive_temperature;
  p.w_mm = w_margin_mse;
  p.w_lk = w_listwise;
  p.w_c = w_contrastive;
  p.rows = d_row_workspace;
  p.losses = d_losses;
  p.grad = d_grad;
  hipStream_t st = sskd::as_stream(stream);
  hipLaunchKernelGGL(kd_loss_rows_kernel, dim3((unsigned)sskd::ceil_div(batch, 4)), dim3(256), 0, st, p);
  int rc = sskd::check_launch("kd_loss_rows_kernel");
  if (rc != SSKD_OK) return rc;
  hipLaunchKernelGGL(kd_loss_finish_kernel, dim3(1), dim3(1024), 0, st, p);
  return sskd::check_launch("kd_loss_finish_kernel");
}
