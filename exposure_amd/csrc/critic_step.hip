// critic_step.hip -- the small kernels between the convolutions and GEMMs of the hand-scheduled WGAN-GP critic update
// (net.py:126-199, 245-251; critics.py:42-98): exposure_amd/critic_direct.py runs the real, fake and interpolated
// images as ONE batch through the critic, forms every first- and second-order quantity of
//     c_loss = mean(D(fake) - D(real)) + lambda mean(max(||grad_x^ D(x^)|| - 1, 0)^2)
// explicitly (no autograd graph), and needs a handful of per-row / per-image reductions on the way.  Each is one launch
// with a fixed summation order.
//
//   expo_critic_head_fwd    fc1's pre-activation [M][hidden] -> h = lrelu(.), logit = h . w2 + b2, the upstream gradient
//                           of every row's pre-activation (real rows -1/N, fake rows +1/N, interpolated rows 1: the
//                           inner gradient d D(x^) / d x^ starts from ones, net.py:174-183)
//   expo_critic_report      the update's reported scalars (net.py:188-199) and the logit centre's moving average (net.py:165-168)
//   expo_critic_head_bwd    bias / fc2 gradients from the rows of the loss and from the penalty's tangent
//   expo_plane_sums         per-image sums of the trailing planes of an NHWC tensor (the gradient that reaches the
//                           statistics planes the critic appends to its input, critics.py:64-76)
//   expo_gp_direct          g = u[..., :3] + ds; norm = sqrt(1e-6 + sum g^2); term = max(norm - 1, 0)^2 (net.py:185-187)
//                           and the penalty's gradient with respect to g in the same launch
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exposure_hip.h"
#include "host_common.h"

namespace expo {

__device__ __forceinline__ float cs_lrelu(float v, float leak) { return v > 0.f ? v : v * leak; }
__device__ __forceinline__ float cs_slope(float z, float leak) {
  return z > 0.f ? 1.0f : (z < 0.f ? leak : 0.5f * (1.0f + leak));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// 16 waves per block, one row each.  Row order of the batch: [real | fake | interpolated].
// hpre arrives as `slabs` partial sums [slabs][rows][hidden] (expo_fc_fwd_slabs: fc1 with its K dimension split) that are
// added here in slab order, then b1 (nullable: already inside a single slab, the library GEMM's addmm).
__global__ __launch_bounds__(1024) void critic_head_fwd_kernel(const float* __restrict__ hpre, const float* __restrict__ b1,
                                                               int slabs, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, int n_real, int n_fake,
                                                               int n_interp, int hidden, float inv_n, float leak,
                                                               float* __restrict__ logits, float* __restrict__ h,
                                                               float* __restrict__ dh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = n_real + n_fake + n_interp;
  const int m = blockIdx.x * 4 + wave;
  if (m >= rows) return;
  const float dl = m < n_real ? -inv_n : (m < n_real + n_fake ? inv_n : 1.0f);
  float dot = 0.f;
  const size_t slab = size_t(rows) * hidden;
  for (int j = lane; j < hidden; j += 64) {
    float pre = hpre[size_t(m) * hidden + j];
#pragma unroll 8
    for (int s = 1; s < slabs; ++s) pre += hpre[s * slab + size_t(m) * hidden + j];
    if (b1) pre += b1[j];
    const float z = cs_lrelu(pre, leak);
    const float wj = w2[j];
    h[size_t(m) * hidden + j] = z;
    dh[size_t(m) * hidden + j] = dl * wj * cs_slope(z, leak);
    dot = fmaf(z, wj, dot);
  }
  dot = wave_sum(dot);
  if (lane == 0) logits[m] = dot + b2[0];
}

// The reported scalars of a critic update (net.py:188-199) and the moving average of the logit centre (net.py:165-168)
// in one launch: out = {c_loss, emd, mean gradient norm, gradient penalty, c_average}; ema (nullable) <- ema + (1 - decay)
// (c_average - ema).  One block; sums in index order per lane, lanes in a fixed tree.
__global__ __launch_bounds__(64) void critic_report_kernel(const float* __restrict__ logits, const float* __restrict__ norm,
                                                           const float* __restrict__ term, int n_real, int n_fake,
                                                           int n_interp, float lambda, float decay, float* __restrict__ out,
                                                           float* __restrict__ ema, float* __restrict__ adam_step) {
  const int lane = threadIdx.x;
  float sr = 0.f, sf = 0.f, sn = 0.f, st = 0.f;
  for (int m = lane; m < n_real; m += 64) sr += logits[m];
  for (int m = lane; m < n_fake; m += 64) sf += logits[n_real + m];
  for (int m = lane; m < n_interp; m += 64) { sn += norm[m]; st += term[m]; }
  sr = wave_sum(sr); sf = wave_sum(sf); sn = wave_sum(sn); st = wave_sum(st);
  if (lane == 0) {
    const float mr = n_real > 0 ? sr / float(n_real) : 0.f, mf = n_fake > 0 ? sf / float(n_fake) : 0.f;
    const float gp = n_interp > 0 ? lambda * st / float(n_interp) : 0.f;
    const float ca = 0.5f * (mf + mr);
    out[0] = (mf - mr) + gp;
    out[1] = mr - mf;
    out[2] = n_interp > 0 ? sn / float(n_interp) : 0.f;
    out[3] = gp;
    out[4] = ca;
    if (ema) ema[0] = ema[0] + (1.0f - decay) * (ca - ema[0]);
    if (adam_step) adam_step[0] = adam_step[0] + 1.0f;  // (the update behind this launch: expo_adam_step's step_advanced)
  }
}

// gb1[j] = sum over the loss rows of dh; gw2[j] = sum over the loss rows of dlogit h + sum over the interpolated rows of
// thpre slope(h); gb2 = sum of dlogit.  thpre arrives as `th_slabs` partial sums [th_slabs][n_interp][hidden]
// (expo_fc_fwd_slabs), added in slab order.  A block = 16 columns x G row groups (G <= 64), the groups' sums added in a
// fixed tree through LDS (round 6: one block of 8 row groups x 128 columns walked 24 dependent rows and added the groups one
// after the other -- 9 us for 192 rows).  G is the largest power of two that divides the real block when the real and fake
// blocks are equally long: group g then holds real row g + i G next to fake row g + i G, whose gradients -w2 s / n and
// +w2 s' / n cancel EXACTLY where the two slopes agree -- the bias gradient of a unit is often exactly zero, and Adam turns
// a rounding-level residue there into a full +-lr step.
__global__ __launch_bounds__(1024) void critic_head_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                               const float* __restrict__ thpre, int th_slabs, int n_real,
                                                               int n_fake, int n_interp, int hidden, int groups, float inv_n,
                                                               float leak, float* __restrict__ gb1, float* __restrict__ gw2,
                                                               float* __restrict__ gb2) {
  __shared__ float p1[64][17], p2[64][17];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int n_loss = n_real + n_fake;
  const int j = blockIdx.x * 16 + col;
  const size_t slab = size_t(n_interp) * hidden;
  float s1 = 0.f, s2 = 0.f;
  if (j < hidden && grp < groups) {
    for (int m = grp; m < n_loss; m += groups) {
      const float dl = m < n_real ? -inv_n : inv_n;
      s1 += dh[size_t(m) * hidden + j];
      s2 = fmaf(dl, h[size_t(m) * hidden + j], s2);
    }
    for (int m = grp; m < n_interp; m += groups) {
      float t = thpre[size_t(m) * hidden + j];
#pragma unroll 8
      for (int s = 1; s < th_slabs; ++s) t += thpre[s * slab + size_t(m) * hidden + j];
      s2 = fmaf(t, cs_slope(h[size_t(n_loss + m) * hidden + j], leak), s2);
    }
  }
  p1[grp][col] = s1;
  p2[grp][col] = s2;
  __syncthreads();
  for (int stride = 32; stride > 0; stride >>= 1) {  // (groups beyond `groups` hold zeros)
    if (grp < stride) {
      p1[grp][col] += p1[grp + stride][col];
      p2[grp][col] += p2[grp + stride][col];
    }
    __syncthreads();
  }
  if (grp == 0 && j < hidden) {
    gb1[j] = p1[0][col];
    gw2[j] = p2[0][col];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) gb2[0] = float(n_fake) * inv_n - float(n_real) * inv_n;
}

// sums[n][c - c0] = sum over the pixels of x[n][p][c], c0 <= c < ct: one block per image, block-reduced in a fixed order
__global__ __launch_bounds__(1024) void plane_sums_kernel(const float* __restrict__ x, float* __restrict__ sums,
                                                          unsigned pixels, int ct, int c0) {
  __shared__ float part[16][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* xi = x + size_t(blockIdx.x) * pixels * ct;
  const int v = ct - c0;  // <= 16
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  for (unsigned p = threadIdx.x; p < pixels; p += 1024) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < v) acc[c] += xi[size_t(p) * ct + c0 + c];
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < v) {
      const float s = wave_sum(acc[c]);
      if (lane == 0) part[wave][c] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < unsigned(v)) {
    float s = part[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 16; ++w) s += part[w][threadIdx.x];
    sums[size_t(blockIdx.x) * v + threadIdx.x] = s;
  }
}

// One block per image: pass 1 the squared norm of g = u[..., :3] + ds, pass 2 the penalty's gradient
//   v = scale 2 max(norm - 1, 0) / norm g       (scale = lambda / N: the mean over the GLOBAL batch)
__global__ __launch_bounds__(1024) void gp_direct_kernel(const float* __restrict__ u, int ct, const float* __restrict__ ds,
                                                         float scale, float* __restrict__ v, float* __restrict__ norm,
                                                         float* __restrict__ term, unsigned pixels) {
  __shared__ float part[16];
  __shared__ float coef;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ui = u + size_t(blockIdx.x) * pixels * ct;
  const float* di = ds + size_t(blockIdx.x) * pixels * 3;
  float* vi = v + size_t(blockIdx.x) * pixels * 3;
  float s = 0.f;
  for (unsigned p = threadIdx.x; p < pixels; p += 1024) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float g = ui[size_t(p) * ct + c] + di[size_t(p) * 3 + c];
      s = fmaf(g, g, s);
    }
  }
  s = wave_sum(s);
  if (lane == 0) part[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = part[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += part[w];
    const float nm = sqrtf(1e-6f + t);
    const float over = fmaxf(nm - 1.0f, 0.0f);
    norm[blockIdx.x] = nm;
    term[blockIdx.x] = over * over;
    coef = scale * 2.0f * over / nm;
  }
  __syncthreads();
  const float c = coef;
  for (unsigned p = threadIdx.x; p < pixels; p += 1024) {
#pragma unroll
    for (int k = 0; k < 3; ++k) vi[size_t(p) * 3 + k] = (ui[size_t(p) * ct + k] + di[size_t(p) * 3 + k]) * c;
  }
}

}  // namespace expo

using namespace expo;

extern "C" {

int expo_critic_head_fwd(const float* hpre, const float* b1, int slabs, const float* w2, const float* b2, int n_real, int n_fake,
                         int n_interp, int hidden, float inv_n, float leak, float* logits, float* h, float* dh, void* stream) {
  if (n_real < 0 || n_fake < 0 || n_interp < 0 || hidden < 1) return fail(EXPO_E_BADARG, "row counts >= 0, hidden >= 1 required");
  if (slabs < 1 || slabs > 64) return fail(EXPO_E_BADARG, "critic_head_fwd: 1 <= slabs <= 64");
  const int rows = n_real + n_fake + n_interp;
  if (rows == 0) return EXPO_OK;
  if (!hpre || !w2 || !b2 || !logits || !h || !dh) return fail(EXPO_E_BADARG, "null pointer");
  hipLaunchKernelGGL(critic_head_fwd_kernel, dim3(unsigned((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     hpre, b1, slabs, w2, b2, n_real, n_fake, n_interp, hidden, inv_n, leak, logits, h, dh);
  HIP_TRY(hipGetLastError(), "critic_head_fwd launch");
  return EXPO_OK;
}

int expo_critic_report(const float* logits, const float* norm, const float* term, int n_real, int n_fake, int n_interp,
                       float lambda, float decay, float* out, float* ema, float* adam_step, void* stream) {
  if (n_real < 0 || n_fake < 0 || n_interp < 0) return fail(EXPO_E_BADARG, "row counts >= 0 required");
  if (!logits || !out || (n_interp > 0 && (!norm || !term))) return fail(EXPO_E_BADARG, "null pointer");
  hipLaunchKernelGGL(critic_report_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), logits, norm, term, n_real,
                     n_fake, n_interp, lambda, decay, out, ema, adam_step);
  HIP_TRY(hipGetLastError(), "critic_report launch");
  return EXPO_OK;
}

int expo_critic_head_bwd(const float* dh, const float* h, const float* thpre, int th_slabs, int n_real, int n_fake, int n_interp,
                         int hidden, float inv_n, float leak, float* gb1, float* gw2, float* gb2, void* stream) {
  if (n_real < 0 || n_fake < 0 || n_interp < 0 || hidden < 1) return fail(EXPO_E_BADARG, "row counts >= 0, hidden >= 1 required");
  if (th_slabs < 1 || th_slabs > 64) return fail(EXPO_E_BADARG, "critic_head_bwd: 1 <= th_slabs <= 64");
  if (!dh || !h || (n_interp > 0 && !thpre) || !gb1 || !gw2 || !gb2) return fail(EXPO_E_BADARG, "null pointer");
  int groups = 64;  // row groups per block: see the kernel
  if (n_real > 0 && n_real == n_fake)
    while (groups > 1 && n_real % groups != 0) groups >>= 1;
  hipLaunchKernelGGL(critic_head_bwd_kernel, dim3(unsigned((hidden + 15) / 16)), dim3(1024), 0, static_cast<hipStream_t>(stream),
                     dh, h, thpre, th_slabs, n_real, n_fake, n_interp, hidden, groups, inv_n, leak, gb1, gw2, gb2);
  HIP_TRY(hipGetLastError(), "critic_head_bwd launch");
  return EXPO_OK;
}

int expo_plane_sums(const float* x, float* sums, int n, size_t pixels_per_image, int channels, int first, void* stream) {
  if (n < 0 || channels < 1 || first < 0 || first >= channels || channels - first > 16)
    return fail(EXPO_E_BADARG, "plane_sums: n >= 0, 0 <= first < channels, at most 16 planes");
  if (n == 0) return EXPO_OK;
  if (!x || !sums) return fail(EXPO_E_BADARG, "null pointer");
  if (pixels_per_image > 0xffffffffull) return fail(EXPO_E_BADARG, "image too large");
  hipLaunchKernelGGL(plane_sums_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), x, sums,
                     unsigned(pixels_per_image), channels, first);
  HIP_TRY(hipGetLastError(), "plane_sums launch");
  return EXPO_OK;
}

int expo_gp_direct(const float* u, int u_channels, const float* ds, float scale, float* v, float* norm, float* term, int n,
                   size_t pixels_per_image, void* stream) {
  if (n < 0 || u_channels < 3) return fail(EXPO_E_BADARG, "gp_direct: n >= 0, u_channels >= 3 required");
  if (n == 0 || pixels_per_image == 0) return EXPO_OK;
  if (!u || !ds || !v || !norm || !term) return fail(EXPO_E_BADARG, "null pointer");
  if (pixels_per_image > 0xffffffffull) return fail(EXPO_E_BADARG, "image too large");
  hipLaunchKernelGGL(gp_direct_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), u, u_channels, ds, scale, v,
                     norm, term, unsigned(pixels_per_image));
  HIP_TRY(hipGetLastError(), "gp_direct launch");
  return EXPO_OK;
}

}  // extern "C"
