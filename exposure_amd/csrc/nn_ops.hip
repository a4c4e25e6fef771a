// nn_ops.hip -- the activation of the convnets that drive the filter path: lrelu (util.py:225-229) fused with the
// bias add in front of it, and its gradient (feature_extractor agent.py:21-32, cnn critics.py:13-35, the FC heads
// filters.py:31-42, agent.py:87-99, critics.py:94-97).  The GEMMs / convolutions themselves stay with MIOpen /
// hipBLASLt (MFMA); what sits between them is element-wise fp32 work that torch issues as 3 launches forward
// (bias add, leaky_relu) and 4 per backward (sign, mul, add, mul) -- at the 64x64 proxy resolution every one of
// them is launch-bound, and a training iteration runs ~100 of each.  One launch each here.
//   z  = lrelu(y + bias[c])              channel = fastest dimension (NHWC conv outputs, (N, C) FC outputs)
//   dy = dz * (z > 0 ? 1 : z < 0 ? leak : (1 + leak) / 2)
// The slope is read from the OUTPUT z: lrelu keeps the sign (and the zero) of its argument.  At exactly 0 the
// reference's formula f1 x + f2 |x| has TF's sub-gradient f1 = (1 + leak) / 2 (tf.abs has gradient 0 there), not
// leaky_relu's `leak`.  The gradient is linear in dz, so the same kernel is its own double backward (the WGAN-GP
// term differentiates the critic's input gradient again, net.py:174-194).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exposure_hip.h"
#include "host_common.h"

namespace expo {

__device__ __forceinline__ float lrelu1(float v, float leak) { return v > 0.f ? v : v * leak; }
__device__ __forceinline__ float lrelu_slope(float z, float leak) {
  return z > 0.f ? 1.0f : (z < 0.f ? leak : 0.5f * (1.0f + leak));
}

// VEC: count % 4 == 0, channels % 4 == 0 (or no bias), 16-byte aligned pointers -> one float4 per thread iteration
template <bool VEC, bool BIAS>
__global__ __launch_bounds__(256) void bias_lrelu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                             float* __restrict__ z, size_t count, int channels,
                                                             float leak) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  if constexpr (VEC) {
    const size_t n4 = count / 4;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 v = reinterpret_cast<const float4*>(y)[i];
      if constexpr (BIAS) {
        const float4 b = *reinterpret_cast<const float4*>(bias + (i * 4) % size_t(channels));
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      v.x = lrelu1(v.x, leak); v.y = lrelu1(v.y, leak); v.z = lrelu1(v.z, leak); v.w = lrelu1(v.w, leak);
      reinterpret_cast<float4*>(z)[i] = v;
    }
  } else {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
      float v = y[i];
      if constexpr (BIAS) v += bias[i % size_t(channels)];
      z[i] = lrelu1(v, leak);
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dz,
                                                        float* __restrict__ dy, size_t count, float leak) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  if constexpr (VEC) {
    const size_t n4 = count / 4;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 a = reinterpret_cast<const float4*>(z)[i];
      float4 g = reinterpret_cast<const float4*>(dz)[i];
      g.x *= lrelu_slope(a.x, leak); g.y *= lrelu_slope(a.y, leak);
      g.z *= lrelu_slope(a.z, leak); g.w *= lrelu_slope(a.w, leak);
      reinterpret_cast<float4*>(dy)[i] = g;
    }
  } else {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
      dy[i] = dz[i] * lrelu_slope(z[i], leak);
  }
}

// dy = dz * slope(z) AND the bias gradient db[c] = sum_rows dy[row][c] in the same pass (round 4: the separate column
// reduction re-read dy -- 8 MB for the first layer at batch 64 -- and was one more launch per layer and backward).
// 256 threads, one float4 (4 consecutive channels) per thread and trip; gridDim.x * 256 is a multiple of C / 4 (C / 4 is a
// power of two <= 64), so a thread keeps its channel group and accumulates it in registers; per block the threads of one
// channel group are added in a fixed order into partial[block][C]; bias_grad_finish_kernel adds the blocks in order.
// No atomics, no zero fill, bit-reproducible.
constexpr int kBgMaxBlocks = 2048;  // enough waves to stream at full bandwidth; partial sums: 2048 x C floats
__global__ __launch_bounds__(256) void lrelu_bwd_bias_kernel(const float* __restrict__ z, const float* __restrict__ dz,
                                                             float* __restrict__ dy, float* __restrict__ partial,
                                                             size_t count, int channels, float leak) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  const size_t n4 = count / 4;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(z)[i];
    float4 g = reinterpret_cast<const float4*>(dz)[i];
    g.x *= lrelu_slope(a.x, leak); g.y *= lrelu_slope(a.y, leak);
    g.z *= lrelu_slope(a.z, leak); g.w *= lrelu_slope(a.w, leak);
    reinterpret_cast<float4*>(dy)[i] = g;
    acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
  }
  __shared__ float4 part[256];
  part[threadIdx.x] = acc;
  __syncthreads();
  const int groups = channels / 4;  // threads t, t + groups, t + 2 groups, ... share channel group t
  if (threadIdx.x < unsigned(groups)) {
    float4 t = {0.f, 0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < 256; k += groups) {
      const float4 p = part[k];
      t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
    }
    reinterpret_cast<float4*>(partial + size_t(blockIdx.x) * channels)[threadIdx.x] = t;
  }
}
// One block per 4 channels: thread (g, c) -- c = 0..3, g = 0..63 -- adds the blocks b = g, g + 64, ... of its channel
// (independent loads), then the 64 partial sums of a channel are added in order: fixed order, blocks / 64 loads deep.
__global__ __launch_bounds__(256) void bias_grad_finish_kernel(const float* __restrict__ partial, float* __restrict__ db,
                                                               int blocks, int channels) {
  __shared__ float part[256];
  const int c = blockIdx.x * 4 + (threadIdx.x & 3), g = threadIdx.x >> 2;
  float t = 0.f;
  int b = g;
  for (; b + 3 * 64 < blocks; b += 4 * 64) {
    const float a0 = partial[size_t(b) * channels + c], a1 = partial[size_t(b + 64) * channels + c];
    const float a2 = partial[size_t(b + 128) * channels + c], a3 = partial[size_t(b + 192) * channels + c];
    t += a0;
    t += a1;
    t += a2;
    t += a3;
  }
  for (; b < blocks; b += 64) t += partial[size_t(b) * channels + c];
  part[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x < 4) {
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += part[k * 4 + threadIdx.x];
    db[blockIdx.x * 4 + threadIdx.x] = s;
  }
}

// ---- the critic step's loss glue (net.py:126-194), three launches instead of ~20 element-wise / reduction launches ----
// (1) both critic inputs from the replayed batch in ONE pass: cat[0:n] = real, cat[n:2n] = fake (float32), and the
//     gradient penalty's interpolation x^ = real + alpha (fake - real), alpha per image (net.py:170-172)
template <typename T>
__global__ __launch_bounds__(256) void gp_inputs_kernel(const T* __restrict__ real, const T* __restrict__ fake,
                                                        const float* __restrict__ alpha, float* __restrict__ cat_out,
                                                        float* __restrict__ interp, int n, size_t m,
                                                        const long long* __restrict__ real_rows,
                                                        const long long* __restrict__ fake_rows) {
  // real_rows / fake_rows (nullable): image `img` of the batch is row rows[img] of the tensor -- the critic batches of a
  // planned iteration are read straight out of the resident data set and the replay memory's pool (no gather launches)
  const int img = blockIdx.y;
  const float a = interp ? alpha[img] : 0.f;  // (interp == NULL: conversion + concatenation only -- the G step's image pairs)
  const size_t base = size_t(img) * m;
  const T* const rsrc = real + size_t(real_rows ? real_rows[img] : img) * m;
  const T* const fsrc = fake + size_t(fake_rows ? fake_rows[img] : img) * m;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < m; i += size_t(gridDim.x) * blockDim.x) {
    const float r = float(rsrc[i]), f = float(fsrc[i]);
    cat_out[base + i] = r;
    cat_out[size_t(n) * m + base + i] = f;
    if (interp) interp[base + i] = r + a * (f - r);
  }
}
// (2) per image: norm = sqrt(1e-6 + sum g^2), term = max(norm - 1, 0)^2 (net.py:185-187: the one-sided penalty);
//     one block per image, fixed summation order
__global__ __launch_bounds__(256) void grad_penalty_fwd_kernel(const float* __restrict__ g, float* __restrict__ norm,
                                                               float* __restrict__ term, size_t m) {
  const size_t base = size_t(blockIdx.x) * m;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < m; i += 256) {
    const float v = g[base + i];
    s = fmaf(v, v, s);
  }
  __shared__ float part[256];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < unsigned(w)) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float nm = sqrtf(1e-6f + part[0]);
    const float over = fmaxf(nm - 1.0f, 0.0f);
    norm[blockIdx.x] = nm;
    term[blockIdx.x] = over * over;
  }
}
// (3) its gradient: d term / d g = 2 max(norm - 1, 0) / norm * g, times the upstream gradient of the image's term
__global__ __launch_bounds__(256) void grad_penalty_bwd_kernel(const float* __restrict__ g, const float* __restrict__ norm,
                                                               const float* __restrict__ dterm, float* __restrict__ dg,
                                                               size_t m) {
  const int img = blockIdx.y;
  const float nm = norm[img];
  const float c = dterm[img] * 2.0f * fmaxf(nm - 1.0f, 0.0f) / nm;
  const size_t base = size_t(img) * m;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < m; i += size_t(gridDim.x) * blockDim.x)
    dg[base + i] = g[base + i] * c;
}

// ---- the eight FC heads' tail: regressors + one-hot gather (agent.py:58-77, 119-125; filters.py:177-179, 201-203, 224-235,
// 256-262, 306-310, 411-413, 435-436, 481-482) ------------------------------------------------------------------------
// Every filter's second FC emits P_j raw features (+ 6 mask features); `filter_param_regressor` squashes them into the
// filter's parameter range, and the agent keeps, per image, the parameters of the ONE filter it selected (a one-hot
// product over the stacked results).  In torch that is ~45 element-wise launches for the regressors, 24 for the gather
// and ~75 in the backward; here one launch each way: thread (image, slot k < 24) regresses feature k of the selected
// head only.  tanh01(x) = tanh(x) / 2 + 1 / 2, tanh_range(l, r)(x) = tanh01(x + bias) (r - l) + l (util.py:277-294; every
// bias of the shipped ranges is atanh(0) = 0 and is passed in anyway).
struct HeadsArgs {
  const float* raw[EXPO_MAX_HEADS];  // head j: [n][width_j] row-major (width_j = P_j + mask features)
  float* draw[EXPO_MAX_HEADS];       // backward: same shapes
  int width[EXPO_MAX_HEADS];
  int abi[EXPO_MAX_HEADS];           // C-ABI filter id of head j (cfg.filters may be a subset / reorder)
  int heads;
  float exposure_range, log_gamma_range, tone_lo, tone_hi, tone_bias, color_lo, color_hi, color_bias, exposure_bias;
};
__device__ __forceinline__ float tanh01f(float x) { return tanhf(x) * 0.5f + 0.5f; }
__device__ __forceinline__ float dtanh01f(float x) { const float t = tanhf(x); return 0.5f * (1.0f - t * t); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
constexpr int kHeadParams[EXPO_NUM_FILTERS] = {1, 1, 3, 1, 8, 1, 1, 24, 2};

__global__ __launch_bounds__(256) void heads_regress_fwd_kernel(const HeadsArgs a, const int32_t* __restrict__ sel,
                                                               float* __restrict__ params, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * EXPO_MAX_PARAMS) return;
  const int img = t / EXPO_MAX_PARAMS, k = t % EXPO_MAX_PARAMS;
  const int j = sel[img];
  float out = 0.f;
  if (j >= 0 && j < a.heads) {
    const int fid = a.abi[j];
    const int P = (fid >= 0 && fid < EXPO_NUM_FILTERS) ? kHeadParams[fid] : 0;
    if (k < P) {
      const float* f = a.raw[j] + size_t(img) * a.width[j];
      const float x = f[k];
      switch (fid) {
        case 0: out = tanh01f(x + a.exposure_bias) * (2.0f * a.exposure_range) - a.exposure_range; break;
        case 1: out = expf(tanh01f(x) * (2.0f * a.log_gamma_range) - a.log_gamma_range); break;
        case 2: {  // features * (0, 1, 1); exp(tanh_range(-.5, .5)); normalised by the luminance of the scaling
          const float s0 = expf(tanh01f(0.0f) - 0.5f), s1 = expf(tanh01f(f[1]) - 0.5f), s2 = expf(tanh01f(f[2]) - 0.5f);
          const float inv = 1.0f / (1e-5f + 0.27f * s0 + 0.67f * s1 + 0.06f * s2);
          out = (k == 0 ? s0 : (k == 1 ? s1 : s2)) * inv;
        } break;
        case 3: case 6: case 8: out = sigmoidf_(x); break;
        case 4: out = tanh01f(x + a.tone_bias) * (a.tone_hi - a.tone_lo) + a.tone_lo; break;
        case 5: out = tanhf(x); break;
        case 7: out = tanh01f(x + a.color_bias) * (a.color_hi - a.color_lo) + a.color_lo; break;
        default: break;
      }
    }
  }
  params[t] = out;
}

// d raw of EVERY head (zero outside the selected head's parameter slice)
__global__ __launch_bounds__(256) void heads_regress_bwd_kernel(const HeadsArgs a, const int32_t* __restrict__ sel,
                                                               const float* __restrict__ dparams, int n, int total_width) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * total_width) return;
  const int img = t / total_width;
  int col = t % total_width, j = 0;
  while (col >= a.width[j]) col -= a.width[j++];
  float g = 0.f;
  if (j == sel[img]) {
    const int fid = a.abi[j];
    const int P = (fid >= 0 && fid < EXPO_NUM_FILTERS) ? kHeadParams[fid] : 0;
    if (col < P) {
      const float* f = a.raw[j] + size_t(img) * a.width[j];
      const float* dp = dparams + size_t(img) * EXPO_MAX_PARAMS;
      const float x = f[col];
      switch (fid) {
        case 0: g = dp[col] * dtanh01f(x + a.exposure_bias) * (2.0f * a.exposure_range); break;
        case 1: {
          const float y = expf(tanh01f(x) * (2.0f * a.log_gamma_range) - a.log_gamma_range);
          g = dp[col] * y * dtanh01f(x) * (2.0f * a.log_gamma_range);
        } break;
        case 2:
          if (col > 0) {  // feature 0 is masked out (features * (0, 1, 1))
            const float s0 = expf(tanh01f(0.0f) - 0.5f), s1 = expf(tanh01f(f[1]) - 0.5f), s2 = expf(tanh01f(f[2]) - 0.5f);
            const float inv = 1.0f / (1e-5f + 0.27f * s0 + 0.67f * s1 + 0.06f * s2);
            const float sk = col == 1 ? s1 : s2, wk = col == 1 ? 0.67f : 0.06f;
            const float dsk = sk * dtanh01f(f[col]);  // d s_k / d f_k
            // out_c = s_c inv:  d out_c / d s_k = delta_ck inv - s_c inv^2 w_k
            const float dot = dp[0] * s0 + dp[1] * s1 + dp[2] * s2;
            g = dsk * (dp[col] * inv - dot * inv * inv * wk);
          }
          break;
        case 3: case 6: case 8: { const float y = sigmoidf_(x); g = dp[col] * y * (1.0f - y); } break;
        case 4: g = dp[col] * dtanh01f(x + a.tone_bias) * (a.tone_hi - a.tone_lo); break;
        case 5: { const float y = tanhf(x); g = dp[col] * (1.0f - y * y); } break;
        case 7: g = dp[col] * dtanh01f(x + a.color_bias) * (a.color_hi - a.color_lo); break;
        default: break;
      }
    }
  }
  a.draw[j][size_t(img) * a.width[j] + col] = g;
}

// ---- action selection, state update and the non-image part of the penalty (agent.py:87-125, 207-252;
// pdf_sample_layer.py:5-10) -- one thread per image, one launch each way instead of ~65 + ~35 tiny torch launches ------
//   pdf = softmax(logits) + 1e-37;  pdf = pdf (1 - eps) + eps / K;  pdf /= sum(pdf) + 1e-30;  H = -sum pdf log pdf
//   random id = #(exclusive cumsum(pdf / (rowsum(pdf) + 1e-36)) < z) - 1   (explicit association orders: the integer
//   result must not depend on a reduction tree; K = 8 uses the packet order ((p0+p4)+(p2+p6))+((p1+p5)+(p3+p7)))
//   id = is_train ? random id : argmax;  one_hot;  surrogate = log(pdf[id] + 1e-10) (0 for id = -1)
//   new_states = [submitted, submitted, step + 1, max(usage, one_hot)],  submitted = |step + 1 - test_steps| < 1e-4
//   penalty_base = (1 - progress) c_e (log K - H) + <usage, one_hot> c_u + (1 - submitted) submitted c_s
// Backward: d logits from d surrogate and d penalty_base (through H); everything else is integer-valued or data.
struct SelectArgs {
  int k, state_dim, is_train;
  float exploration, exploration_penalty, usage_penalty, early_stop_penalty, test_steps;
};
constexpr int kSelMaxK = 16;

__device__ __forceinline__ void select_pdf(const float* __restrict__ l, int K, float eps, float* p, float* s_out) {
  float mx = l[0];
  for (int i = 1; i < K; ++i) mx = fmaxf(mx, l[i]);
  float den = 0.f;
  for (int i = 0; i < K; ++i) {
    p[i] = expf(l[i] - mx);
    den += p[i];
  }
  float tot = 0.f;
  for (int i = 0; i < K; ++i) {
    const float sm = p[i] / den;
    if (s_out) s_out[i] = sm;
    p[i] = (sm + 1e-37f) * (1.0f - eps) + eps * 1.0f / float(K);
    tot += p[i];
  }
  tot += 1e-30f;
  for (int i = 0; i < K; ++i) p[i] = p[i] / tot;
}

__global__ __launch_bounds__(64) void agent_select_fwd_kernel(const SelectArgs a, const float* __restrict__ logits,
                                                              const float* __restrict__ noise, int noise_stride,
                                                              const float* __restrict__ states,
                                                              const float* __restrict__ progress, float* __restrict__ pdf,
                                                              float* __restrict__ entropy, int32_t* __restrict__ selected,
                                                              float* __restrict__ onehot, float* __restrict__ surrogate,
                                                              float* __restrict__ new_states, float* __restrict__ pen_base,
                                                              int n) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= n) return;
  const int K = a.k;
  float p[kSelMaxK];
  select_pdf(logits + size_t(img) * K, K, a.exploration, p, nullptr);
  float H = 0.f;
  for (int i = 0; i < K; ++i) {
    pdf[size_t(img) * K + i] = p[i];
    H += -p[i] * logf(p[i]);
  }
  entropy[img] = H;
  // --- sampling (pdf_sample_layer.py:5-10), explicit association orders
  float rs;
  if (K == 8) rs = ((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7]));
  else {
    rs = p[0];
    for (int i = 1; i < K; ++i) rs = rs + p[i];
  }
  rs += 1e-36f;
  const float z = noise[size_t(img) * noise_stride];
  float cdf = 0.f;
  int cnt = 0, amax = 0;
  for (int i = 0; i < K; ++i) {
    if (i > 0) cdf = cdf + p[i - 1] / rs;
    cnt += (cdf < z) ? 1 : 0;
    if (p[i] > p[amax]) amax = i;  // torch.argmax: the first maximum
  }
  const int id = a.is_train ? cnt - 1 : amax;
  selected[img] = id;
  const float* st = states + size_t(img) * a.state_dim;
  float* ns = new_states + size_t(img) * a.state_dim;
  const float step = st[2];
  const float submitted = (fabsf(step + 1.0f - a.test_steps) < 1e-4f) ? 1.0f : 0.0f;
  ns[0] = submitted;
  ns[1] = submitted;
  ns[2] = step + 1.0f;
  float usage_pen = 0.f;
  for (int i = 0; i < K; ++i) {
    const float oh = (i == id) ? 1.0f : 0.0f;
    onehot[size_t(img) * K + i] = oh;
    const float u = st[3 + i];
    usage_pen += u * oh;
    ns[3 + i] = fmaxf(u, oh);
  }
  surrogate[img] = (id >= 0 && id < K) ? logf(p[id] + 1e-10f) : 0.0f;
  const float ent_pen = (1.0f - progress[0]) * a.exploration_penalty * (-H + logf(float(K)));
  pen_base[img] = ent_pen + usage_pen * a.usage_penalty + (1.0f - submitted) * submitted * a.early_stop_penalty;
}

__global__ __launch_bounds__(64) void agent_select_bwd_kernel(const SelectArgs a, const float* __restrict__ logits,
                                                              const int32_t* __restrict__ selected,
                                                              const float* __restrict__ progress,
                                                              const float* __restrict__ d_surrogate,
                                                              const float* __restrict__ d_pen_base,
                                                              float* __restrict__ d_logits, int n) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= n) return;
  const int K = a.k;
  float p[kSelMaxK], sm[kSelMaxK], gp[kSelMaxK];
  select_pdf(logits + size_t(img) * K, K, a.exploration, p, sm);
  const int id = selected[img];
  const float gH = d_pen_base[img] * (1.0f - progress[0]) * a.exploration_penalty * -1.0f;
  const float gs = d_surrogate[img];
  float dot = 0.f, tot = 0.f;
  for (int i = 0; i < K; ++i) {  // recompute the renormalisation's denominator (b = p * tot)
    gp[i] = gH * (-(logf(p[i]) + 1.0f)) + ((i == id) ? gs / (p[i] + 1e-10f) : 0.0f);
    dot += gp[i] * p[i];
    tot += (sm[i] + 1e-37f) * (1.0f - a.exploration) + a.exploration * 1.0f / float(K);
  }
  tot += 1e-30f;
  // p = b / tot: dL/db_j = (gp_j - sum_i gp_i p_i) / tot;  b = (s + 1e-37)(1 - eps) + eps / K;  s = softmax(l)
  float gsm[kSelMaxK], dot2 = 0.f;
  for (int i = 0; i < K; ++i) {
    gsm[i] = (gp[i] - dot) / tot * (1.0f - a.exploration);
    dot2 += gsm[i] * sm[i];
  }
  for (int i = 0; i < K; ++i) d_logits[size_t(img) * K + i] = sm[i] * (gsm[i] - dot2);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int grid_for(size_t items) {
  const size_t blocks = (items + 255) / 256;
  return int(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));  // grid-stride beyond 1 M items
}


// ---- Adam over a list of tensors, one launch (net.py:222-251: the three AdamOptimizers of a training iteration) -------
// torch's fused multi-tensor Adam walks chunks of 65 536 elements with one block each: 20-75 blocks for the 1-5 M
// parameters of one of these networks, 43 us per step on 256 CUs (profiles/r04_final_kernel_stats_train.csv).  Here a
// block takes 1 024 elements (one float4 per thread), tensors are addressed through a table passed by value, and the
// step counter lives on the device (read by every block, advanced by a one-thread launch behind the update), so a step is
// capturable and replays without host involvement.  Update rule = torch.optim.Adam (no weight decay, no amsgrad):
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
constexpr int kAdamMaxTensors = EXPO_ADAM_MAX_TENSORS;
constexpr unsigned kAdamBlockElems = 1024;
struct AdamArgs {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  unsigned n[kAdamMaxTensors];
  unsigned first_block[kAdamMaxTensors + 1];
  int count;
  unsigned char vec[kAdamMaxTensors];  // the tensor's four pointers are 16-byte aligned
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                         float step_size, float inv_sqrt_bc2) {
  m = m + (g - m) * (1.0f - b1);
  v = b2 * v + (1.0f - b2) * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a, const float* __restrict__ lr, const float* __restrict__ step,
                                                   float b1, float b2, float eps, float step_add) {
  int t = 0;
  while (t + 1 < a.count && blockIdx.x >= a.first_block[t + 1]) ++t;  // uniform: scalar compares
  const float st = step[0] + step_add;  // (1: the counter holds the steps taken so far; 0: the caller's previous kernel advanced it)
  const float bc1 = 1.0f - powf(b1, st), bc2 = 1.0f - powf(b2, st);
  const float step_size = lr[0] / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  const unsigned n = a.n[t];
  const unsigned i = (blockIdx.x - a.first_block[t]) * kAdamBlockElems + threadIdx.x * 4;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  if (a.vec[t] && i + 4 <= n) {
    float4 pp = *reinterpret_cast<const float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i);
    float4 mm = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
    adam_one(pp.x, gg.x, mm.x, vv.x, b1, b2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.y, gg.y, mm.y, vv.y, b1, b2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.z, gg.z, mm.z, vv.z, b1, b2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.w, gg.w, mm.w, vv.w, b1, b2, eps, step_size, inv_sqrt_bc2);
    *reinterpret_cast<float4*>(p + i) = pp;
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(v + i) = vv;
  } else {
    for (unsigned k = i; k < i + 4 && k < n; ++k) {
      float pp = p[k], mm = m[k], vv = v[k];
      adam_one(pp, g[k], mm, vv, b1, b2, eps, step_size, inv_sqrt_bc2);
      p[k] = pp;
      m[k] = mm;
      v[k] = vv;
    }
  }
}

// t <- t + 1, as a launch of its own BEHIND the update (round 6).  Rounds 4-5 let the last of the update's blocks advance
// the counter, found through one ticket atomic per block on a single word: ~12 ns each, 1 200 to 6 000 of them per step --
// 14 of the 19 us of a critic step's update and 72 of the generator's 78 (profiles/r06_experiments.md).
__global__ void adam_advance_kernel(float* __restrict__ step) { step[0] = step[0] + 1.0f; }

// ---- the input of a convnet: image channels + per-image values broadcast as constant planes, centred ------------------
// critics.py:64-76 (`tf.concat([images, states planes, statistics planes]) - 0.5` before `cnn`) and agent.py:17-19, 47-53
// (`enrich_image_input`, then `net - 0.5` in feature_extractor): float conversion, concatenation(s) and subtraction as one
// launch; out[n, p, c] = (c < 3 ? image[n, p, c] : vec[n, c - 3]) - offset, float32 NHWC with 3 + V channels.
template <typename T>
__global__ __launch_bounds__(256) void planes_concat_kernel(const T* __restrict__ img, const float* __restrict__ vec,
                                                            float* __restrict__ out, size_t total, unsigned hw, unsigned v,
                                                            float offset, unsigned magic) {
  // A block owns 256 consecutive pixels of one image (blockIdx.y: the planes' values are block-uniform) = 256 C
  // CONSECUTIVE output floats; thread t writes elements t, t + 256, ...: every store instruction covers whole cache lines.
  // element -> (pixel, channel) by a multiply-high with magic = ceil(2^32 / C) (exact for elements below 2^16).
  // (History: an element per thread with a 64-bit division each, 18 us for the critic update's 192 x 64 x 64 x 6 input; a
  // pixel per thread, 13 us -- but its C stores per thread are C x 4 bytes apart between lanes: 26 us for the value net's
  // 128 x 64 x 64 x 17.)
  const unsigned C = 3 + v;
  const unsigned n = blockIdx.y;
  __shared__ float vs[64];
  __shared__ float px[256 * 3];  // the block's image values, centred: loaded with coalesced reads, all in flight at once
  if (threadIdx.x < v) vs[threadIdx.x] = vec[size_t(n) * v + threadIdx.x] - offset;
  (void)total;
  for (unsigned p0 = blockIdx.x * 256; p0 < hw; p0 += gridDim.x * 256) {
    const unsigned pixels = min(256u, hw - p0);
    const size_t pix0 = size_t(n) * hw + p0;
    __syncthreads();  // (vs on the first pass; px free again on later ones)
#pragma unroll
    for (unsigned k = 0; k < 3; ++k) {
      const unsigned i = threadIdx.x + 256 * k;
      if (i < pixels * 3) px[i] = img ? float(img[pix0 * 3 + i]) - offset : -offset;
    }
    __syncthreads();
    float* o = out + pix0 * C;
    const unsigned count = pixels * C;
    for (unsigned e = threadIdx.x; e < count; e += 256) {
      const unsigned p = __umulhi(e, magic), c = e - p * C;
      o[e] = c < 3 ? px[p * 3 + c] : vs[c - 3];
    }
  }
}

// ---- the generator step's loss glue (net.py:92-160 as restated in gan.py::generator_losses, cfg.gan == 'w') -----------
// per image: stopped, step from new_states;  nv = new_value [step <= max_len];  gate = a + (1 - a) stopped;
//   reward = gate (fake_logit - fake_input_logit) m - penalty;  q = reward + (1 - stopped) gamma nv;  adv = q - old_value;
//   TD:  g term = -q plm + surrogate (-adv);   otherwise  g term = -reward + surrogate (-reward);   v term = adv^2
// losses = (mean g term, mean v term).  coef[5][N] holds d g_loss / d (fake_logit, new_value, surrogate, penalty) and
// d v_loss / d old_value per image (already divided by N): the backward is coef times the upstream scalar.
struct GLossArgs {
  float all_reward, mult, discount, plm, max_len;
  int use_penalty, use_td, state_dim, stopped_col, step_col;
};
__global__ __launch_bounds__(256) void generator_losses_kernel(GLossArgs a, const float* __restrict__ fake_logit,
                                                               const float* __restrict__ fake_input_logit,
                                                               const float* __restrict__ new_value,
                                                               const float* __restrict__ old_value,
                                                               const float* __restrict__ new_states,
                                                               const float* __restrict__ penalty,
                                                               const float* __restrict__ surrogate, float* __restrict__ losses,
                                                               float* __restrict__ reward_out, float* __restrict__ q_out,
                                                               float* __restrict__ coef, int n, float* __restrict__ adam_step_a,
                                                               float* __restrict__ adam_step_b) {
  float gs = 0.f, vs = 0.f;
  const float inv_n = 1.0f / float(n);
  for (int i = threadIdx.x; i < n; i += 256) {
    const float stopped = new_states[size_t(i) * a.state_dim + a.stopped_col];
    const float keep = new_states[size_t(i) * a.state_dim + a.step_col] > a.max_len ? 0.0f : 1.0f;
    const float nv = new_value[i] * keep;
    const float gate = a.all_reward + (1.0f - a.all_reward) * stopped;
    const float raw = gate * (fake_logit[i] - fake_input_logit[i]) * a.mult;
    const float reward = a.use_penalty ? raw - penalty[i] : raw;
    const float cont = (1.0f - stopped) * a.discount;
    const float q = reward + cont * nv;
    const float adv = q - old_value[i];
    const float sur = surrogate[i];
    float routine, weight, dq;  // dq = d g term / d q (TD) or / d reward
    if (a.use_td) { routine = -q * a.plm; weight = -adv; dq = -a.plm; }
    else { routine = -reward; weight = -reward; dq = -1.0f; }
    gs += routine + sur * weight;
    vs += adv * adv;
    reward_out[i] = reward;
    q_out[i] = q;
    coef[i] = dq * gate * a.mult * inv_n;                            // d g_loss / d fake_logit
    coef[n + i] = a.use_td ? dq * cont * keep * inv_n : 0.0f;        // d g_loss / d new_value
    coef[2 * n + i] = weight * inv_n;                                // d g_loss / d surrogate
    coef[3 * n + i] = a.use_penalty ? -dq * inv_n : 0.0f;            // d g_loss / d penalty
    coef[4 * n + i] = -2.0f * adv * inv_n;                           // d v_loss / d old_value
  }
  __shared__ float pg[256], pv[256];
  pg[threadIdx.x] = gs;
  pv[threadIdx.x] = vs;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < unsigned(w)) {
      pg[threadIdx.x] += pg[threadIdx.x + w];
      pv[threadIdx.x] += pv[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    losses[0] = pg[0] * inv_n;
    losses[1] = pv[0] * inv_n;
    // (nullable) Adam's step counters of the two optimisers this step's updates belong to: t <- t + 1 here, in front of
    // the updates, instead of one one-thread launch behind each (expo_adam_step's step_advanced)
    if (adam_step_a) adam_step_a[0] = adam_step_a[0] + 1.0f;
    if (adam_step_b) adam_step_b[0] = adam_step_b[0] + 1.0f;
  }
}
}  // namespace expo

using namespace expo;

extern "C" {

int expo_bias_lrelu_fwd(const float* y, const float* bias, float* z, size_t count, int channels, float leak,
                        void* stream) {
  if (count == 0) return EXPO_OK;
  if (!y || !z) return fail(EXPO_E_BADARG, "null pointer");
  if (bias && (channels < 1 || count % size_t(channels) != 0))
    return fail(EXPO_E_BADARG, "count must be a multiple of channels");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = count % 4 == 0 && aligned16(y) && aligned16(z) && (!bias || (channels % 4 == 0 && aligned16(bias)));
  const int grid = grid_for(vec ? count / 4 : count);
#define EXPO_L(VEC, BIAS) \
  hipLaunchKernelGGL((bias_lrelu_fwd_kernel<VEC, BIAS>), dim3(grid), dim3(256), 0, s, y, bias, z, count, channels, leak)
  if (vec) { if (bias) EXPO_L(true, true); else EXPO_L(true, false); }
  else { if (bias) EXPO_L(false, true); else EXPO_L(false, false); }
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "bias_lrelu_fwd launch");
  return EXPO_OK;
}

int expo_lrelu_bwd(const float* z, const float* dz, float* dy, size_t count, float leak, void* stream) {
  if (count == 0) return EXPO_OK;
  if (!z || !dz || !dy) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = count % 4 == 0 && aligned16(z) && aligned16(dz) && aligned16(dy);
  const int grid = grid_for(vec ? count / 4 : count);
  if (vec) hipLaunchKernelGGL((lrelu_bwd_kernel<true>), dim3(grid), dim3(256), 0, s, z, dz, dy, count, leak);
  else hipLaunchKernelGGL((lrelu_bwd_kernel<false>), dim3(grid), dim3(256), 0, s, z, dz, dy, count, leak);
  HIP_TRY(hipGetLastError(), "lrelu_bwd launch");
  return EXPO_OK;
}

static int heads_args(HeadsArgs* a, const float* const* raw, float* const* draw, const int* widths, const int* abi_ids,
                      int heads, const float* ranges) {
  if (heads < 1 || heads > EXPO_MAX_HEADS || !raw || !widths || !abi_ids || !ranges)
    return fail(EXPO_E_BADARG, "bad heads arguments");
  for (int j = 0; j < heads; ++j) {
    if (!raw[j] || (draw && !draw[j])) return fail(EXPO_E_BADARG, "null pointer");
    if (abi_ids[j] < 0 || abi_ids[j] >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
    if (widths[j] < kHeadParams[abi_ids[j]]) return fail(EXPO_E_BADARG, "head narrower than its filter's parameter count");
    a->raw[j] = raw[j];
    a->draw[j] = draw ? draw[j] : nullptr;
    a->width[j] = widths[j];
    a->abi[j] = abi_ids[j];
  }
  a->heads = heads;
  a->exposure_range = ranges[0]; a->log_gamma_range = ranges[1];
  a->tone_lo = ranges[2]; a->tone_hi = ranges[3]; a->tone_bias = ranges[4];
  a->color_lo = ranges[5]; a->color_hi = ranges[6]; a->color_bias = ranges[7]; a->exposure_bias = ranges[8];
  return EXPO_OK;
}

int expo_heads_regress_fwd(const float* const* raw, const int* widths, const int* abi_ids, int heads, const float* ranges,
                           const int32_t* selected, float* params, int n, void* stream) {
  if (n < 0) return fail(EXPO_E_BADARG, "n >= 0 required");
  HeadsArgs a{};
  if (int rc = heads_args(&a, raw, nullptr, widths, abi_ids, heads, ranges)) return rc;
  if (n == 0) return EXPO_OK;
  if (!selected || !params) return fail(EXPO_E_BADARG, "null pointer");
  const int total = n * EXPO_MAX_PARAMS;
  hipLaunchKernelGGL(heads_regress_fwd_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     selected, params, n);
  HIP_TRY(hipGetLastError(), "heads_regress_fwd launch");
  return EXPO_OK;
}

int expo_heads_regress_bwd(const float* const* raw, float* const* draw, const int* widths, const int* abi_ids, int heads,
                           const float* ranges, const int32_t* selected, const float* dparams, int n, void* stream) {
  if (n < 0) return fail(EXPO_E_BADARG, "n >= 0 required");
  HeadsArgs a{};
  if (!draw) return fail(EXPO_E_BADARG, "null pointer");
  if (int rc = heads_args(&a, raw, draw, widths, abi_ids, heads, ranges)) return rc;
  if (n == 0) return EXPO_OK;
  if (!selected || !dparams) return fail(EXPO_E_BADARG, "null pointer");
  int total_width = 0;
  for (int j = 0; j < heads; ++j) total_width += widths[j];
  const int total = n * total_width;
  hipLaunchKernelGGL(heads_regress_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     selected, dparams, n, total_width);
  HIP_TRY(hipGetLastError(), "heads_regress_bwd launch");
  return EXPO_OK;
}

static int select_args(SelectArgs* a, int k, int state_dim, int is_train, const float* consts) {
  if (k < 1 || k > kSelMaxK) return fail(EXPO_E_BADARG, "number of filters must be in [1, 16]");
  if (state_dim < 3 + k) return fail(EXPO_E_BADARG, "state rows must hold 3 + K values (reward, stopped, step, usage)");
  if (!consts) return fail(EXPO_E_BADARG, "null pointer");
  *a = SelectArgs{k, state_dim, is_train ? 1 : 0, consts[0], consts[1], consts[2], consts[3], consts[4]};
  return EXPO_OK;
}

int expo_agent_select_fwd(const float* logits, const float* noise, int noise_stride, const float* states,
                          const float* progress, const float* consts, int k, int state_dim, int is_train, float* pdf,
                          float* entropy, int32_t* selected, float* onehot, float* surrogate, float* new_states,
                          float* penalty_base, int n, void* stream) {
  SelectArgs a;
  if (int rc = select_args(&a, k, state_dim, is_train, consts)) return rc;
  if (n < 0 || noise_stride < 1) return fail(EXPO_E_BADARG, "n >= 0 and noise_stride >= 1 required");
  if (n == 0) return EXPO_OK;
  if (!logits || !noise || !states || !progress || !pdf || !entropy || !selected || !onehot || !surrogate || !new_states ||
      !penalty_base)
    return fail(EXPO_E_BADARG, "null pointer");
  hipLaunchKernelGGL(agent_select_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), a, logits,
                     noise, noise_stride, states, progress, pdf, entropy, selected, onehot, surrogate, new_states,
                     penalty_base, n);
  HIP_TRY(hipGetLastError(), "agent_select_fwd launch");
  return EXPO_OK;
}

int expo_agent_select_bwd(const float* logits, const int32_t* selected, const float* progress, const float* consts, int k,
                          int state_dim, const float* d_surrogate, const float* d_penalty_base, float* d_logits, int n,
                          void* stream) {
  SelectArgs a;
  if (int rc = select_args(&a, k, state_dim, 1, consts)) return rc;
  if (n < 0) return fail(EXPO_E_BADARG, "n >= 0 required");
  if (n == 0) return EXPO_OK;
  if (!logits || !selected || !progress || !d_surrogate || !d_penalty_base || !d_logits)
    return fail(EXPO_E_BADARG, "null pointer");
  hipLaunchKernelGGL(agent_select_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), a, logits,
                     selected, progress, d_surrogate, d_penalty_base, d_logits, n);
  HIP_TRY(hipGetLastError(), "agent_select_bwd launch");
  return EXPO_OK;
}

static int gp_inputs_impl(const void* real, const long long* real_rows, const void* fake, const long long* fake_rows,
                          const float* alpha, float* cat_out, float* interp, int n, size_t elems_per_image, int dtype,
                          void* stream) {
  if (n < 0) return fail(EXPO_E_BADARG, "n >= 0 required");
  if (n == 0 || elems_per_image == 0) return EXPO_OK;
  if (!real || !fake || !cat_out || (interp && !alpha)) return fail(EXPO_E_BADARG, "null pointer");
  if (dtype != EXPO_F16 && dtype != EXPO_F32) return fail(EXPO_E_BADDTYPE, "dtype must be EXPO_F16 or EXPO_F32");
  if (n > 65535) return fail(EXPO_E_BADARG, "n > 65535 not supported (grid.y)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t bx = (elems_per_image + 255) / 256;
  if (bx > 64) bx = 64;
  const dim3 grid(unsigned(bx), n), block(256);
  if (dtype == EXPO_F16)
    hipLaunchKernelGGL(gp_inputs_kernel<_Float16>, grid, block, 0, s, (const _Float16*)real, (const _Float16*)fake, alpha,
                       cat_out, interp, n, elems_per_image, real_rows, fake_rows);
  else
    hipLaunchKernelGGL(gp_inputs_kernel<float>, grid, block, 0, s, (const float*)real, (const float*)fake, alpha, cat_out,
                       interp, n, elems_per_image, real_rows, fake_rows);
  HIP_TRY(hipGetLastError(), "gp_inputs launch");
  return EXPO_OK;
}

int expo_gp_inputs(const void* real, const void* fake, const float* alpha, float* cat_out, float* interp, int n,
                   size_t elems_per_image, int dtype, void* stream) {
  return gp_inputs_impl(real, nullptr, fake, nullptr, alpha, cat_out, interp, n, elems_per_image, dtype, stream);
}

int expo_gp_inputs_rows(const void* real, const int64_t* real_rows, const void* fake, const int64_t* fake_rows,
                        const float* alpha, float* cat_out, float* interp, int n, size_t elems_per_image, int dtype,
                        void* stream) {
  return gp_inputs_impl(real, reinterpret_cast<const long long*>(real_rows), fake,
                        reinterpret_cast<const long long*>(fake_rows), alpha, cat_out, interp, n, elems_per_image, dtype, stream);
}

int expo_grad_penalty_fwd(const float* g, float* norm, float* term, int n, size_t elems_per_image, void* stream) {
  if (n < 0) return fail(EXPO_E_BADARG, "n >= 0 required");
  if (n == 0) return EXPO_OK;
  if (!g || !norm || !term) return fail(EXPO_E_BADARG, "null pointer");
  hipLaunchKernelGGL(grad_penalty_fwd_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), g, norm, term,
                     elems_per_image);
  HIP_TRY(hipGetLastError(), "grad_penalty_fwd launch");
  return EXPO_OK;
}

int expo_grad_penalty_bwd(const float* g, const float* norm, const float* dterm, float* dg, int n, size_t elems_per_image,
                          void* stream) {
  if (n < 0) return fail(EXPO_E_BADARG, "n >= 0 required");
  if (n == 0 || elems_per_image == 0) return EXPO_OK;
  if (!g || !norm || !dterm || !dg) return fail(EXPO_E_BADARG, "null pointer");
  if (n > 65535) return fail(EXPO_E_BADARG, "n > 65535 not supported (grid.y)");
  size_t bx = (elems_per_image + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(grad_penalty_bwd_kernel, dim3(unsigned(bx), n), dim3(256), 0, static_cast<hipStream_t>(stream), g, norm,
                     dterm, dg, elems_per_image);
  HIP_TRY(hipGetLastError(), "grad_penalty_bwd launch");
  return EXPO_OK;
}

int expo_planes_concat(const void* images, const float* vec, float* out, int n, size_t pixels_per_image, int v, int dtype,
                       float offset, void* stream) {
  if (n < 0 || v < 0) return fail(EXPO_E_BADARG, "n >= 0 and v >= 0 required");
  if (n == 0 || pixels_per_image == 0) return EXPO_OK;
  if (!out || (v > 0 && !vec)) return fail(EXPO_E_BADARG, "null pointer");
  if (dtype != EXPO_F16 && dtype != EXPO_F32) return fail(EXPO_E_BADDTYPE, "dtype must be EXPO_F16 or EXPO_F32");
  if (pixels_per_image > 0xffffffffull) return fail(EXPO_E_BADARG, "image too large");
  if (n > 65535) return fail(EXPO_E_BADARG, "n > 65535 not supported (grid.y)");
  if (v > 61) return fail(EXPO_E_BADARG, "at most 61 planes (64 channels) supported");
  const unsigned magic = unsigned((0x100000000ull + (3 + v) - 1) / (3 + v));  // ceil(2^32 / C): exact for e < 2^16
  const size_t total = size_t(n) * pixels_per_image * size_t(3 + v);
  size_t bx = (pixels_per_image + 255) / 256;
  if (bx > 1024) bx = 1024;
  const dim3 grid{unsigned(bx), unsigned(n), 1u};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == EXPO_F16)
    hipLaunchKernelGGL(planes_concat_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)images, vec, out, total,
                       unsigned(pixels_per_image), unsigned(v), offset, magic);
  else
    hipLaunchKernelGGL(planes_concat_kernel<float>, grid, dim3(256), 0, s, (const float*)images, vec, out, total,
                       unsigned(pixels_per_image), unsigned(v), offset, magic);
  HIP_TRY(hipGetLastError(), "planes_concat launch");
  return EXPO_OK;
}

int expo_generator_losses(const float* fake_logit, const float* fake_input_logit, const float* new_value,
                          const float* old_value, const float* new_states, int state_dim, const float* penalty,
                          const float* surrogate, const float* consts, int use_td, float* losses, float* reward, float* q_value,
                          float* coef, int n, float* adam_step_a, float* adam_step_b, void* stream) {
  if (n <= 0) return fail(EXPO_E_BADARG, "n >= 1 required");
  if (!fake_logit || !fake_input_logit || !new_value || !old_value || !new_states || !surrogate || !consts || !losses ||
      !reward || !q_value || !coef)
    return fail(EXPO_E_BADARG, "null pointer");
  if (state_dim < 3) return fail(EXPO_E_BADARG, "state rows must hold reward, stopped, step");
  GLossArgs a{consts[0], consts[1], consts[2], consts[3], consts[4], penalty ? 1 : 0, use_td ? 1 : 0, state_dim, 1, 2};
  hipLaunchKernelGGL(generator_losses_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a, fake_logit,
                     fake_input_logit, new_value, old_value, new_states, penalty, surrogate, losses, reward, q_value, coef, n,
                     adam_step_a, adam_step_b);
  HIP_TRY(hipGetLastError(), "generator_losses launch");
  return EXPO_OK;
}

int expo_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const size_t* numel, const float* lr, float* step, void* ticket, float beta1,
                   float beta2, float eps, int step_advanced, void* stream) {
  if (count < 0) return fail(EXPO_E_BADARG, "count >= 0 required");
  if (count == 0) return EXPO_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !step || !ticket)
    return fail(EXPO_E_BADARG, "null pointer");
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f))
    return fail(EXPO_E_BADARG, "betas in [0, 1) and eps >= 0 required");
  for (int j = 0; j < count; ++j) {
    if (!params[j] || !grads[j] || !exp_avg[j] || !exp_avg_sq[j]) return fail(EXPO_E_BADARG, "null tensor pointer");
    if (numel[j] == 0 || numel[j] > 0xffffffffull - kAdamBlockElems)
      return fail(EXPO_E_BADARG, "tensor sizes in [1, 2^32 - 1024) required");
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // tables of at most EXPO_ADAM_MAX_TENSORS tensors per launch; only the last launch of a call advances the step
  for (int base = 0; base < count; base += kAdamMaxTensors) {
    AdamArgs a;
    a.count = count - base < kAdamMaxTensors ? count - base : kAdamMaxTensors;
    unsigned blocks = 0;
    for (int j = 0; j < a.count; ++j) {
      a.p[j] = params[base + j];
      a.g[j] = grads[base + j];
      a.m[j] = exp_avg[base + j];
      a.v[j] = exp_avg_sq[base + j];
      a.n[j] = unsigned(numel[base + j]);
      a.first_block[j] = blocks;
      const unsigned nb = (a.n[j] + kAdamBlockElems - 1) / kAdamBlockElems;
      if (blocks + nb < blocks) return fail(EXPO_E_BADARG, "too many elements for one call");
      blocks += nb;
      a.vec[j] = aligned16(a.p[j]) && aligned16(a.g[j]) && aligned16(a.m[j]) && aligned16(a.v[j]);
    }
    a.first_block[a.count] = blocks;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, a, lr, step, beta1, beta2, eps, step_advanced ? 0.0f : 1.0f);
    HIP_TRY(hipGetLastError(), "adam launch");
  }
  // every launch above computed with t = step + 1; the counter moves behind them (stream order)
  (void)ticket;  // (ABI 4-5: the word the blocks took tickets on; still accepted, no longer touched)
  // (step_advanced: a kernel the caller launched in FRONT of this update -- expo_critic_report, expo_generator_losses -- has
  // moved the counter already: one launch less per update)
  if (!step_advanced) {
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, step);
    HIP_TRY(hipGetLastError(), "adam advance launch");
  }
  return EXPO_OK;
}

size_t expo_lrelu_bwd_bias_workspace_bytes(int channels) {
  return channels > 0 ? size_t(kBgMaxBlocks) * size_t(channels) * sizeof(float) : 0;
}

int expo_lrelu_bwd_bias(const float* z, const float* dz, float* dy, float* dbias, size_t count, int channels, float leak,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!z || !dz || !dy || !dbias) return fail(EXPO_E_BADARG, "null pointer");
  if (channels < 4 || channels > 256 || (channels & (channels - 1)) != 0)
    return fail(EXPO_E_BADARG, "channels must be a power of two in [4, 256]");
  if (count == 0 || count % size_t(channels) != 0) return fail(EXPO_E_BADARG, "count must be a positive multiple of channels");
  if (!aligned16(z) || !aligned16(dz) || !aligned16(dy)) return fail(EXPO_E_BADARG, "pointers must be 16-byte aligned");
  if (!workspace || workspace_bytes < expo_lrelu_bwd_bias_workspace_bytes(channels) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return fail(EXPO_E_BADARG, "workspace missing, misaligned or too small (expo_lrelu_bwd_bias_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t blocks = (count / 4 + 255) / 256;
  if (blocks > size_t(kBgMaxBlocks)) blocks = kBgMaxBlocks;
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(lrelu_bwd_bias_kernel, dim3(unsigned(blocks)), dim3(256), 0, s, z, dz, dy, partial, count, channels, leak);
  HIP_TRY(hipGetLastError(), "lrelu_bwd_bias launch");
  hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(channels / 4), dim3(256), 0, s, partial, dbias, int(blocks), channels);
  HIP_TRY(hipGetLastError(), "bias_grad_finish launch");
  return EXPO_OK;
}

}  // extern "C"
