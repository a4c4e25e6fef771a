// exposure_hip.hip -- gfx950 kernels + C-ABI (include/exposure_hip.h) for Exposure's
// differentiable per-pixel filter stack (reference: /root/reference/filters.py).
//
// Kernel plan (HBM-bound pixel maps; no contraction => no MFMA here):
//   filter_fwd_kernel<F,T>   one pass:  read x (6 B/px fp16), write y (6 B/px)
//   filter_bwd_kernel<F,T>   one pass:  read x, dy, write dx (18 B/px); y is recomputed;
//                            per-image parameter gradients: per-thread fp32 partials ->
//                            wave shuffle reduce -> LDS across the 4 waves -> one record per
//                            block in the caller's workspace -> finish_kernel (one tiny launch
//                            per call / per chain) sums an image's records in a fixed order and
//                            writes dparams: no float atomics, no zero-fill launch, and the
//                            result is bit-reproducible (block_reduce_record / finish_kernel)
//   dispatch_{fwd,bwd}       same bodies behind a block-uniform switch on filter_ids[n]
//                            (the reference's one-hot select, agent.py:119-125) with the
//                            over-exposure penalty (agent.py:249-251) fused in
//   apply_{fwd,bwd}          Filter.apply with the spatial mask (cfg.masking = True) in one pass
//   chain_fused_fwd          all steps of a per-image filter sequence in registers (inference)
//   stats / penalty          per-image reductions (critics.py:48-62, agent.py:249-251)
// Grid: blockIdx.y = image (so parameters are block-uniform -> SGPRs), blockIdx.x =
// chunk of the image; each thread walks 48-byte pixel groups with a block stride.
// Cache policy: tensors >= EXPO_STREAM_MIN_BYTES run the IoStream instantiations (nt loads,
// write-through stores), smaller ones IoCached -- pixel_io.h.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include <string>
#include <type_traits>

#include "../../include/exposure_hip.h"
#include "filter_math.h"
#include "pixel_io.h"
#include "kernel_common.h"
#include "host_common.h"

// Compile-time structure of the streaming kernels (each alternative was measured and rejected, see
// profiles/r02_experiments.md): forward waves own exactly one chunk and do not prefetch; backward waves walk
// block-strided chunks with a one-deep software prefetch; the forward's per-image constants are fetched after
// the first chunk's loads have been issued.
#ifndef EXPO_BWD_MIN_WAVES
#define EXPO_BWD_MIN_WAVES  // e.g. -DEXPO_BWD_MIN_WAVES=,5 : a register budget for 5 waves per SIMD (probe builds)
#endif
#ifndef EXPO_CURVE_BWD_PREFETCH
#define EXPO_CURVE_BWD_PREFETCH 1  // (probe builds: the curve backward without its one-deep software prefetch)
#endif
constexpr bool kFwdPrefetch = false;
constexpr bool kBwdPrefetch = true;
constexpr int kAccParts = 4;  // partial sums per thread in the element-wise backward (1 / 2 / 4 measured, r02p27)


namespace expo {

// --------------------------------------------------------------------------- forward
template <class F, typename T, bool VEC, bool PEN, class IO = IoCached>
__device__ __forceinline__ void fwd_body(const T* __restrict__ xi, T* __restrict__ yi,
                                         const float* __restrict__ prm, float* __restrict__ rec,
                                         int hw, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  // On the vector path the per-image constants are fetched AFTER the first chunk's loads have been issued
  // (stream_groups' prologue): the dependent scalar loads (kernel argument -> parameter -> exp2) otherwise sit
  // in front of the image loads of every wave, and a forward wave lives for exactly one chunk.
  typename F::Prm q;
  if constexpr (!VEC) q = F::load(prm);
  float pen = 0.f;
  const int stride = gridDim.x * kThreads;
  // Tone / Color on the vector path (all lanes of a wave alive): segment table instead of the
  // telescoped min/fma chain; built once per wave behind the first loads (stream_groups' prologue)
  constexpr bool kCurveTab = VEC && F::kLutFloats > 0;
  constexpr int kNC = kCurveTab ? F::NP / kCurveSteps : 1;
  __shared__ float2_lut ftab[kCurveTab ? kWaves : 1][32];
  float2_lut* const tab = ftab[kCurveTab ? (threadIdx.x >> 6) : 0];
  // per-group work, shared by the prefetching (VEC) and the element-wise loop
  auto compute = [&](float* v, int g) {
    if constexpr (kCurveTab) {
      curve_lut_apply<kNC, PPL>(v, tab);
      if constexpr (PEN) {
#pragma unroll
        for (int j = 0; j < PPL * 3; ++j) {
          const float o = fmaxf(v[j] - 1.0f, 0.0f);
          pen = fmaf(o, o, pen);
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      float y[3];
      F::fwd(q, v + 3 * k, y);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[3 * k + c] = y[c];
        if constexpr (PEN) {
          // padding pixels are x = 0, and f(0) <= 1 for every filter, so they add nothing
          const float o = fmaxf(y[c] - 1.0f, 0.0f);
          pen = fmaf(o, o, pen);
        }
      }
    }
  };
  if constexpr (VEC) {
    const T* const ins[1] = {xi};
    stream_groups<T, 1, true, kFwdPrefetch, IO>(
        ins, yi, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
        [&](float (&v)[1][PPL * 3], int g) { compute(v[0], g); },
        [&]() {
          q = F::load(prm);
          if constexpr (kCurveTab) curve_lut_build<kNC>(prm[(threadIdx.x & 63) % F::NP], tab);
        });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3];
      load_slow<T>(xi, g, hw, v);
      compute(v, g);
      store_slow<T>(yi, g, hw, v);
    }
  }
  if constexpr (PEN) {
    float a[1] = {pen};
    block_reduce_record<1>(a, rec);  // finish_kernel: penalty[n] = sum * inv_count
  }
}

template <class F, typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads) void filter_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                              const float* __restrict__ params,
                                                              int hw, int groups, int reverse) {
  // reverse: walk the images from the last to the first (expo_chain_*: consecutive launches alternate
  // direction so each starts on the images the previous one touched last -- Infinity-Cache reuse)
  const int n = reverse ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  fwd_body<F, T, VEC, false, IO>(x + off, y + off, params + n * F::NP, nullptr, hw, groups);
}

// -------------------------------------------------------------------------- backward
template <class F, typename T, bool VEC, bool HAS_DX, bool PEN, int MODE, class IO = IoCached>
__device__ __forceinline__ void bwd_body(const T* __restrict__ xi, const T* __restrict__ dyi,
                                         T* __restrict__ dxi, const float* __restrict__ prm,
                                         float* __restrict__ rec, int hw, int groups, float pen_scale) {
  constexpr int PPL = PixTraits<T>::PPL;
  const typename F::Prm q = F::load(prm);
  // per-image curve LUT (Tone / Color backward) staged in LDS once per block
  __shared__ __attribute__((aligned(16))) float lut[F::kLutFloats > 0 ? F::kLutFloats : 4];
  // fp16-exact fast path of the curve filters (bit-pattern LUT + packed accumulators).  With the fused penalty
  // dy stays an fp16 value and the penalty's addend travels beside it (CurveF::bwd_group<.., HAS_PEN>); the
  // masked path scales dy -> generic path there.
  constexpr bool kF16X = std::is_same<T, half_t>::value;
  // staged AFTER the first chunk's loads are in flight on the vector path (stream_groups' prologue)
  // fp16 vector path: the table is built from a per-lane copy of the parameters, fetched HERE -- before the first
  // image loads are issued -- so that nothing in the staging waits behind them (CurveF::stage16_lanes)
  constexpr bool kLaneStage = F::kLutFloats > 0 && kF16X && VEC;
  float klane = 0.f;
  if constexpr (kLaneStage) klane = prm[(threadIdx.x & 63) % F::NP];
  auto stage_lut = [&]() {
    if constexpr (F::kLutFloats > 0) {
      if constexpr (kLaneStage) F::stage16_lanes(klane, lut);
      else F::template stage_for<kF16X>(prm, lut);
      __syncthreads();
    }
  };
  if constexpr (!VEC) stage_lut();
  // fused penalty of the curve filters on the vector path: the forward is re-evaluated through the per-wave
  // segment table of the forward kernels (5 VALU + one LDS read per element instead of 16 VALU), and not at all
  // for an image whose curve cannot exceed 1 (curve_can_exceed_one -- every image under the reference's ranges)
  constexpr bool kPenTab = PEN && VEC && F::kLutFloats > 0;
  constexpr int kNC = kPenTab ? F::NP / kCurveSteps : 1;
  __shared__ float2_lut ptab[kPenTab ? kWaves : 1][32];
  float2_lut* const tab = ptab[kPenTab ? (threadIdx.x >> 6) : 0];
  bool pen_live = PEN;
  if constexpr (kPenTab) {
    pen_live = curve_can_exceed_one<kNC>(prm);
    if (pen_live) curve_lut_build<kNC>(prm[(threadIdx.x & 63) % F::NP], tab);
  }
  float acc[F::NACC];
#pragma unroll
  for (int j = 0; j < F::NACC; ++j) acc[j] = 0.f;
  // element-wise filters: the pixels of a group feed kAccParts independent partial sums, so a group's 24
  // accumulator updates are not one dependent chain (Exposure / Gamma / S+ / Contrast / WNB keep ONE accumulator)
  constexpr int kParts = F::kHasGroupBwd ? 1 : (kAccParts < PPL ? kAccParts : PPL);
  float part[kParts > 1 ? kParts - 1 : 1][F::NACC];
#pragma unroll
  for (int p = 0; p < (kParts > 1 ? kParts - 1 : 1); ++p)
#pragma unroll
    for (int j = 0; j < F::NACC; ++j) part[p][j] = 0.f;
  const int stride = gridDim.x * kThreads;
  auto compute = [&](float* v, float* d, int g) {
    if constexpr (kPenTab) {
      // plain group backward on dy, then the penalty's share as a fix-up (block-uniform: only for an image whose
      // curve can exceed 1; element-wise: only where y > 1)
      F::template bwd_group<PPL, kF16X, false>(q, lut, v, d, acc, nullptr);
      if (pen_live) F::template bwd_pen_fixup<PPL, kF16X>(lut, tab, v, d, acc, pen_scale);
      return;
    }
    float pen[PEN ? PPL * 3 : 1];
    if constexpr (PEN) {
      // fused over-exposure penalty: dy += 2 max(y-1,0) * dpen / (H W 3)
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        float y[3];
        F::fwd(q, v + 3 * k, y);
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // padding pixels: y = f(0) <= 1 -> no contribution
          pen[3 * k + c] = fmaxf(y[c] - 1.0f, 0.0f) * pen_scale;
          if constexpr (!F::kHasGroupBwd) d[3 * k + c] += pen[3 * k + c];
        }
      }
    }
    if constexpr (F::kHasGroupBwd) {
      F::template bwd_group<PPL, kF16X, PEN>(q, lut, v, d, acc, pen);
    } else {
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        float dx[3];
        F::bwd(q, lut, v + 3 * k, d + 3 * k, dx, (k % kParts == 0) ? acc : part[(k % kParts + kParts - 1) % kParts], MODE);
#pragma unroll
        for (int c = 0; c < 3; ++c) d[3 * k + c] = dx[c];
      }
    }
  };
  if constexpr (VEC) {
    const T* const ins[2] = {xi, dyi};
    constexpr bool kPf = (F::kLutFloats > 0 && !PEN) ? bool(EXPO_CURVE_BWD_PREFETCH) : kBwdPrefetch;
    stream_groups<T, 2, HAS_DX, kPf, IO>(ins, dxi, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                         [&](float (&v)[2][PPL * 3], int g) { compute(v[0], v[1], g); }, stage_lut);
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3], d[PPL * 3];
      load_slow<T>(xi, g, hw, v);
      load_slow<T>(dyi, g, hw, d);
      compute(v, d, g);
      if constexpr (HAS_DX) store_slow<T>(dxi, g, hw, d);
    }
  }
  if constexpr (kParts > 1) {
#pragma unroll
    for (int p = 0; p < kParts - 1; ++p)
#pragma unroll
      for (int j = 0; j < F::NACC; ++j) acc[j] += part[p][j];
  }
  block_reduce_record<F::NACC>(acc, rec);  // finish_kernel applies F::finish_one to the image totals
}

template <class F, typename T, bool VEC, bool HAS_DX, int MODE, class IO>
__global__ __launch_bounds__(kThreads EXPO_BWD_MIN_WAVES) void filter_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                              T* __restrict__ dx,
                                                              const float* __restrict__ params,
                                                              float* __restrict__ records, int hw, int groups,
                                                              int reverse) {
  const int n = reverse ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  bwd_body<F, T, VEC, HAS_DX, false, MODE, IO>(x + off, dy + off, HAS_DX ? dx + off : nullptr, params + n * F::NP,
                                               records + size_t(n) * gridDim.x * kWsSlots, hw, groups, 0.f);
}

// y = 0 for a whole image (filter id -1 in the per-image dispatch entry points)
template <typename T, bool VEC, class IO = IoCached>
__device__ __forceinline__ void zero_image(T* __restrict__ yi, int hw, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  float z[PPL * 3];
#pragma unroll
  for (int j = 0; j < PPL * 3; ++j) z[j] = 0.f;
  const int stride = gridDim.x * kThreads;
  if constexpr (VEC) {
    const RawGroup rz = pack<T>(z);
    const __amdgpu_buffer_rsrc_t ry = make_image_rsrc(yi, hw);
    for (int gw = blockIdx.x * kThreads + (threadIdx.x & ~63); gw * PPL < hw; gw += stride)
      store_raw<IO::kStore>(ry, chunk_byte_offset<T>(gw, threadIdx.x & 63), rz);
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) store_slow<T>(yi, g, hw, z);
  }
}

// ------------------------------------------------ Filter.apply with a spatial mask (masking on)
// out = (1 - mask) x + mask process(x), mask per pixel from MaskPrm (filters.py:86-88, 110-148).
template <class F, typename T, bool VEC, class IO = IoCached>
__device__ __forceinline__ void apply_fwd_body(const T* __restrict__ xi, T* __restrict__ yi, const float* __restrict__ prm,
                                               const float* __restrict__ mprm, float sharp, float min_strength, int h,
                                               int w, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int hw = h * w;
  const typename F::Prm q = F::load(prm);
  const MaskPrm mk = MaskPrm::load(mprm, sharp, min_strength, h, w, pixel_step_a<T, VEC>(), pixel_step_b<T, VEC>());
  const int stride = gridDim.x * kThreads;
  // curve filters on the vector path: per-wave segment table, as in filter_fwd_kernel
  constexpr bool kCurveTab = VEC && F::kLutFloats > 0;
  constexpr int kNC = kCurveTab ? F::NP / kCurveSteps : 1;
  __shared__ float2_lut ftab[kCurveTab ? kWaves : 1][32];
  float2_lut* const tab = ftab[kCurveTab ? (threadIdx.x >> 6) : 0];
  if constexpr (kCurveTab) curve_lut_build<kNC>(prm[(threadIdx.x & 63) % F::NP], tab);
  auto compute = [&](float* v, int g) {
    const int lane = threadIdx.x & 63;
    // (row, column) of the group's first pixel, the others by constant steps (PixelWalk) -- in the forward only for
    // the curve filters: 64x512x512 fp16, gpurun r03p42: C 53.0 -> 43.9 us, but E 33.1 -> 34.0 and Ct 38.1 -> 41.8
    // (the walk's ordering anchor costs the light bodies more than the integer multiplies it removes)
    constexpr bool kWalk = F::kLutFloats > 0;
    int row = 0, col = 0;
    if constexpr (kWalk) mk.pw.start(pixel_index<T, VEC>(g, 0, lane), row, col);
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      float p[3];
      if constexpr (kCurveTab) curve_lut_pixel<kNC>(tab, v + 3 * k, p);
      else F::fwd(q, v + 3 * k, p);
      MaskPrm::Eval e;
      if constexpr (kWalk) {
        if (k > 0) mk.pw.step(pixel_step_is_b<T, VEC>(k), row, col);
        e = mk.eval_rc(row, col, v + 3 * k);
      } else {
        e = mk.eval(pixel_index<T, VEC>(g, k, lane), v + 3 * k);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) v[3 * k + c] = fmaf(e.m, p[c] - v[3 * k + c], v[3 * k + c]);
      if constexpr (kWalk) PixelWalk::after(row, col, v[3 * k]);
    }
  };
  if constexpr (VEC) {
    const T* const ins[1] = {xi};
    stream_groups<T, 1, true, true, IO>(ins, yi, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                        [&](float (&v)[1][PPL * 3], int g) { compute(v[0], g); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3];
      load_slow<T>(xi, g, hw, v);
      compute(v, g);
      store_slow<T>(yi, g, hw, v);
    }
  }
}

template <class F, typename T, bool VEC, class IO = IoCached>
__global__ __launch_bounds__(kThreads) void apply_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             const float* __restrict__ params,
                                                             const float* __restrict__ mask_params, float sharp,
                                                             float min_strength, int h, int w, int groups) {
  const int n = blockIdx.y;
  const size_t off = size_t(n) * h * w * 3;
  apply_fwd_body<F, T, VEC, IO>(x + off, y + off, params + n * F::NP, mask_params + n * 6, sharp, min_strength, h, w, groups);
}

// Masked apply with a per-image filter choice: the agent's step with cfg.masking = True.  The reference applies EVERY
// filter's masked apply to the whole batch and reduces with the one-hot (agent.py:58-77, 119-125); the one-hot zeroes the
// seven others, so only the selected filter's kernel body runs per image here (block-uniform switch, as in
// dispatch_fwd_kernel).  params: [N][EXPO_MAX_PARAMS], mask_params: [N][6] of the SELECTED filter; id -1 -> y = 0.
template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void apply_dispatch_fwd_kernel(const int32_t* __restrict__ ids,
                                                                      const T* __restrict__ x, T* __restrict__ y,
                                                                      const float* __restrict__ params,
                                                                      const float* __restrict__ mask_params, float sharp,
                                                                      float min_strength, int h, int w, int groups) {
  const int n = blockIdx.y;
  const size_t off = size_t(n) * h * w * 3;
  const float* prm = params + n * EXPO_MAX_PARAMS;
  const float* mp = mask_params + n * 6;
#define EXPO_CASE(ID, F) \
  case ID: apply_fwd_body<F, T, VEC>(x + off, y + off, prm, mp, sharp, min_strength, h, w, groups); break;
  switch (ids[n]) {
    EXPO_CASE(0, ExposureF)
    EXPO_CASE(1, GammaF)
    EXPO_CASE(2, WhiteBalanceF)
    EXPO_CASE(3, SatPlusF)
    EXPO_CASE(4, ToneF)
    EXPO_CASE(5, ContrastF)
    EXPO_CASE(6, WnbF)
    EXPO_CASE(7, ColorF)
    EXPO_CASE(8, LevelF)
    default: zero_image<T, VEC, IoCached>(y + off, h * w, groups); break;
  }
#undef EXPO_CASE
}

template <class F, typename T, bool VEC, bool HAS_DX, int MODE, class IO = IoCached>
__device__ __forceinline__ void apply_bwd_body(const T* __restrict__ xi, const T* __restrict__ dyi, T* __restrict__ dxi,
                                               const float* __restrict__ prm, const float* __restrict__ mprm,
                                               float* __restrict__ rec, float sharp, float min_strength, int h, int w,
                                               int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int hw = h * w;
  const typename F::Prm q = F::load(prm);
  const MaskPrm mk = MaskPrm::load(mprm, sharp, min_strength, h, w, pixel_step_a<T, VEC>(), pixel_step_b<T, VEC>());
  __shared__ __attribute__((aligned(16))) float lut[F::kLutFloats > 0 ? F::kLutFloats : 4];
  if constexpr (F::kLutFloats > 0) {
    F::stage(prm, lut);
    __syncthreads();
  }
  // curve filters on the vector path: the forward re-evaluation goes through the per-wave segment table
  constexpr bool kCurveTab = VEC && F::kLutFloats > 0;
  constexpr int kNC = kCurveTab ? F::NP / kCurveSteps : 1;
  __shared__ float2_lut ftab[kCurveTab ? kWaves : 1][32];
  float2_lut* const tab = ftab[kCurveTab ? (threadIdx.x >> 6) : 0];
  if constexpr (kCurveTab) curve_lut_build<kNC>(prm[(threadIdx.x & 63) % F::NP], tab);
  // filter accumulators and the 6 mask-parameter accumulators share ONE per-block record
  float acc[F::NACC + 6];
  float* const macc = acc + F::NACC;
#pragma unroll
  for (int j = 0; j < F::NACC + 6; ++j) acc[j] = 0.f;
  const int stride = gridDim.x * kThreads;
  auto compute = [&](float* v, float* d, int g) {
    const int lane = threadIdx.x & 63;
    int row, col;
    mk.pw.start(pixel_index<T, VEC>(g, 0, lane), row, col);
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const float* xv = v + 3 * k;
      float* dv = d + 3 * k;
      float p[3], gp[3], dxf[3];
      if constexpr (kCurveTab) curve_lut_pixel<kNC>(tab, xv, p);
      else F::fwd(q, xv, p);
      if (k > 0) mk.pw.step(pixel_step_is_b<T, VEC>(k), row, col);
      const MaskPrm::Eval e = mk.eval_rc(row, col, xv);
      const float dm = dv[0] * (p[0] - xv[0]) + dv[1] * (p[1] - xv[1]) + dv[2] * (p[2] - xv[2]);
      const float gsig = dm * mk.S * e.sg * (1.0f - e.sg);  // dL/d(inp)
      const float graw = gsig * mk.k;                       // dL/d(inp_raw)
      macc[0] = fmaf(graw, e.gx, macc[0]);
      macc[1] = fmaf(graw, e.gy, macc[1]);
      macc[2] = fmaf(graw, e.lumc, macc[2]);
      macc[3] = fmaf(graw, 2.0f, macc[3]);
      macc[4] = fmaf(gsig * e.inp_raw, mk.k_over_mp4, macc[4]);
      macc[5] = fmaf(dm * e.sg, mk.dS, macc[5]);
#pragma unroll
      for (int c = 0; c < 3; ++c) gp[c] = e.m * dv[c];
      F::bwd(q, lut, xv, gp, dxf, acc, MODE);
      const float gl = graw * mk.c;  // through lum(x) inside the mask
      dv[0] = fmaf(1.0f - e.m, dv[0], dxf[0]) + kLumR * gl;
      dv[1] = fmaf(1.0f - e.m, dv[1], dxf[1]) + kLumG * gl;
      dv[2] = fmaf(1.0f - e.m, dv[2], dxf[2]) + kLumB * gl;
      PixelWalk::after(row, col, dv[2]);
    }
  };
  if constexpr (VEC) {
    const T* const ins[2] = {xi, dyi};
    stream_groups<T, 2, HAS_DX, true, IO>(ins, dxi, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                          [&](float (&v)[2][PPL * 3], int g) { compute(v[0], v[1], g); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3], d[PPL * 3];
      load_slow<T>(xi, g, hw, v);
      load_slow<T>(dyi, g, hw, d);
      compute(v, d, g);
      if constexpr (HAS_DX) store_slow<T>(dxi, g, hw, d);
    }
  }
  // record = [filter accumulators | 6 mask accumulators]; finish_kernel splits them
  block_reduce_record<F::NACC + 6>(acc, rec);
}

template <class F, typename T, bool VEC, bool HAS_DX, int MODE, class IO = IoCached>
__global__ __launch_bounds__(kThreads) void apply_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             T* __restrict__ dx, const float* __restrict__ params,
                                                             const float* __restrict__ mask_params,
                                                             float* __restrict__ records, float sharp,
                                                             float min_strength, int h, int w, int groups) {
  const int n = blockIdx.y;
  const size_t off = size_t(n) * h * w * 3;
  apply_bwd_body<F, T, VEC, HAS_DX, MODE, IO>(x + off, dy + off, HAS_DX ? dx + off : nullptr, params + n * F::NP,
                                              mask_params + n * 6, records + size_t(n) * gridDim.x * kWsSlots, sharp,
                                              min_strength, h, w, groups);
}

// Backward of apply_dispatch_fwd_kernel: the selected filter's masked-apply backward per image.  One launch for all
// filters (it runs at the 64x64 proxy resolution, where every launch is latency-bound; the curve bodies set the register
// budget).  Records: [filter accumulators | 6 mask accumulators] of the image's filter; id -1 writes none (finish_kernel
// emits zero rows).
template <typename T, bool VEC, bool HAS_DX, int MODE>
__global__ __launch_bounds__(kThreads) void apply_dispatch_bwd_kernel(const int32_t* __restrict__ ids,
                                                                      const T* __restrict__ x, const T* __restrict__ dy,
                                                                      T* __restrict__ dx, const float* __restrict__ params,
                                                                      const float* __restrict__ mask_params,
                                                                      float* __restrict__ records, float sharp,
                                                                      float min_strength, int h, int w, int groups) {
  const int n = blockIdx.y;
  const size_t off = size_t(n) * h * w * 3;
  const float* prm = params + n * EXPO_MAX_PARAMS;
  const float* mp = mask_params + n * 6;
  float* rec = records + size_t(n) * gridDim.x * kWsSlots;
  T* dxi = HAS_DX ? dx + off : nullptr;
#define EXPO_CASE(ID, F)                                                                                              \
  case ID:                                                                                                            \
    apply_bwd_body<F, T, VEC, HAS_DX, MODE>(x + off, dy + off, dxi, prm, mp, rec, sharp, min_strength, h, w, groups); \
    break;
  switch (ids[n]) {
    EXPO_CASE(0, ExposureF)
    EXPO_CASE(1, GammaF)
    EXPO_CASE(2, WhiteBalanceF)
    EXPO_CASE(3, SatPlusF)
    EXPO_CASE(4, ToneF)
    EXPO_CASE(5, ContrastF)
    EXPO_CASE(6, WnbF)
    EXPO_CASE(7, ColorF)
    EXPO_CASE(8, LevelF)
    default:
      if constexpr (HAS_DX) zero_image<T, VEC, IoCached>(dxi, h * w, groups);
      break;
  }
#undef EXPO_CASE
}

// ------------------------------------------------------------ VignetFilter.apply (filters.py:341-396)
template <typename T, bool VEC, class IO = IoCached>
__global__ __launch_bounds__(kThreads) void vignet_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                              const float* __restrict__ mask_params, float sharp,
                                                              int masking, int h, int w, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y, hw = h * w;
  const size_t off = size_t(n) * hw * 3;
  const VignetPrm mk = VignetPrm::load(mask_params + n * 5, sharp, masking, h, w);
  const int stride = gridDim.x * kThreads;
  auto compute = [&](float* v, int g) {
    const int lane = threadIdx.x & 63;
    // (a row / column per pixel: the incremental walk of the masked apply kernels buys nothing here and costs 12
    // registers = an occupancy step, gpurun r03p42)
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const float keep = 1.0f - mk.eval(pixel_index<T, VEC>(g, k, lane)).m;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[3 * k + c] *= keep;
    }
  };
  if constexpr (VEC) {
    const T* const ins[1] = {x + off};
    stream_groups<T, 1, true, true, IO>(ins, y + off, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                        [&](float (&v)[1][PPL * 3], int g) { compute(v[0], g); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3];
      load_slow<T>(x + off, g, hw, v);
      compute(v, g);
      store_slow<T>(y + off, g, hw, v);
    }
  }
}

template <typename T, bool VEC, bool HAS_DX, class IO = IoCached>
__global__ __launch_bounds__(kThreads) void vignet_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                              T* __restrict__ dx, const float* __restrict__ mask_params,
                                                              float* __restrict__ records, float sharp, int masking,
                                                              int h, int w, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y, hw = h * w;
  const size_t off = size_t(n) * hw * 3;
  const VignetPrm mk = VignetPrm::load(mask_params + n * 5, sharp, masking, h, w);
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int stride = gridDim.x * kThreads;
  auto compute = [&](const float* xv, float* d, int g) {  // padding pixels: x = dy = 0 -> t = 0
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const VignetPrm::Eval e = mk.eval(pixel_index<T, VEC>(g, k, lane));
      const float t = -(xv[3 * k] * d[3 * k] + xv[3 * k + 1] * d[3 * k + 1] + xv[3 * k + 2] * d[3 * k + 2]);
      const float ts = t * e.sg * (1.0f - e.sg);
      acc[0] = fmaf(ts, e.gx2, acc[0]);
      acc[1] = fmaf(ts, e.gy2, acc[1]);
      acc[2] += ts;
      acc[3] = fmaf(ts, e.u, acc[3]);
      acc[4] = fmaf(t, e.sg, acc[4]);
      const float keep = 1.0f - e.m;
#pragma unroll
      for (int c = 0; c < 3; ++c) d[3 * k + c] *= keep;
    }
  };
  if constexpr (VEC) {
    const T* const ins[2] = {x + off, dy + off};
    stream_groups<T, 2, HAS_DX, true, IO>(ins, HAS_DX ? dx + off : nullptr, hw,
                                          blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                          [&](float (&v)[2][PPL * 3], int g) { compute(v[0], v[1], g); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3], d[PPL * 3];
      load_slow<T>(x + off, g, hw, v);
      load_slow<T>(dy + off, g, hw, d);
      compute(v, d, g);
      if constexpr (HAS_DX) store_slow<T>(dx + off, g, hw, d);
    }
  }
  block_reduce_record<5>(acc, records + size_t(n) * gridDim.x * kWsSlots);
}

// --------------------------------------------------- per-image dispatch (one-hot select)
// SET is a bit mask of the filter ids this launch handles (bit 15 = id -1).  The BACKWARD dispatch
// is issued as two launches -- light filters and the register-heavy curve filters -- so each gets
// its own VGPR budget / occupancy; blocks whose image selected a filter outside SET exit at once.
// The forward needs no split (its curve bodies are light): one launch with kSetAll.
constexpr int kSetLight = 0x8000 | 0x100 | 0x6F;  // -1, E, G, W, S+, Ct, BW, Le
constexpr int kSetCurves = 0x90;         // T, C
constexpr int kSetAll = kSetLight | kSetCurves;

template <typename T, bool VEC, bool PEN, int SET, class IO>
__global__ __launch_bounds__(kThreads) void dispatch_fwd_kernel(const int32_t* __restrict__ ids,
                                                                const T* __restrict__ x, T* __restrict__ y,
                                                                const float* __restrict__ params,
                                                                float* __restrict__ records, int hw,
                                                                int groups) {
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const float* prm = params + n * EXPO_MAX_PARAMS;
  float* rec = PEN ? records + size_t(n) * gridDim.x * kWsSlots : nullptr;
  const int id = ids[n];  // block-uniform
#define EXPO_CASE(ID, F)                                                                                   \
  case ID:                                                                                                 \
    if constexpr ((SET >> ID) & 1) fwd_body<F, T, VEC, PEN, IO>(x + off, y + off, prm, rec, hw, groups);   \
    break;
  switch (id) {
    EXPO_CASE(0, ExposureF)
    EXPO_CASE(1, GammaF)
    EXPO_CASE(2, WhiteBalanceF)
    EXPO_CASE(3, SatPlusF)
    EXPO_CASE(4, ToneF)
    EXPO_CASE(5, ContrastF)
    EXPO_CASE(6, WnbF)
    EXPO_CASE(7, ColorF)
    EXPO_CASE(8, LevelF)
    default:  // id -1: all-zero one-hot -> y = 0 (finish_kernel writes penalty = 0 without reading records)
      if constexpr ((SET >> 15) & 1) zero_image<T, VEC, IO>(y + off, hw, groups);
      break;
  }
#undef EXPO_CASE
}

template <typename T, bool VEC, bool HAS_DX, bool PEN, int MODE, int SET, class IO>
__global__ __launch_bounds__(kThreads) void dispatch_bwd_kernel(const int32_t* __restrict__ ids,
                                                                const T* __restrict__ x, const T* __restrict__ dy,
                                                                T* __restrict__ dx, const float* __restrict__ params,
                                                                float* __restrict__ records,
                                                                const float* __restrict__ dpenalty, int hw,
                                                                int groups, float inv_count) {
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const float* prm = params + n * EXPO_MAX_PARAMS;
  float* rec = records + size_t(n) * gridDim.x * kWsSlots;
  const float ps = PEN ? 2.0f * inv_count * dpenalty[n] : 0.f;
  T* dxi = HAS_DX ? dx + off : nullptr;
  const int id = ids[n];
#define EXPO_CASE(ID, F)                                                                                  \
  case ID:                                                                                                \
    if constexpr ((SET >> ID) & 1)                                                                        \
      bwd_body<F, T, VEC, HAS_DX, PEN, MODE, IO>(x + off, dy + off, dxi, prm, rec, hw, groups, ps);       \
    break;
  switch (id) {
    EXPO_CASE(0, ExposureF)
    EXPO_CASE(1, GammaF)
    EXPO_CASE(2, WhiteBalanceF)
    EXPO_CASE(3, SatPlusF)
    EXPO_CASE(4, ToneF)
    EXPO_CASE(5, ContrastF)
    EXPO_CASE(6, WnbF)
    EXPO_CASE(7, ColorF)
    EXPO_CASE(8, LevelF)
    default:  // id -1 selects nothing: dx = 0 (finish_kernel writes the all-zero dparams row)
      if constexpr (HAS_DX && ((SET >> 15) & 1)) zero_image<T, VEC, IO>(dxi, hw, groups);
      break;
  }
#undef EXPO_CASE
}

// ------------------------------------------------------------- per-image reductions
// critics.py:48-62.  Raw sums {sum(l-1/2), sum (l-1/2)^2, sum sat} are accumulated (shifted to tame
// the E[l^2]-E[l]^2 cancellation); finish_kernel turns them into {mean, variance, mean sat}.
template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads) void stats_kernel(const T* __restrict__ x, float* __restrict__ records,
                                                         int hw, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y;
  const T* xi = x + size_t(n) * hw * 3;
  float acc[3] = {0.f, 0.f, 0.f};
  const int stride = gridDim.x * kThreads;
  auto compute = [&](const float* v, int g) {
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      if (live_pixel<T, VEC>(g, k, threadIdx.x & 63, hw)) {
        const float* p = v + 3 * k;
        // critics.py:48-49: r*.27 + g*.67 + b*.06 + 1e-5
        const float l = ((p[0] * kLumR + p[1] * kLumG) + p[2] * kLumB + 1e-5f) - 0.5f;
        acc[0] += l;
        acc[1] = fmaf(l, l, acc[1]);
        const float c0 = clamp01x(p[0], 0.f, 1.f), c1 = clamp01x(p[1], 0.f, 1.f), c2 = clamp01x(p[2], 0.f, 1.f);
        const float mx = fmaxf(fmaxf(c0, c1), c2), mn = fminf(fminf(c0, c1), c2);
        // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the denominator is >= 1e-2
        acc[2] = fmaf(mx - mn, fast_rcp(fminf(mx + mn, 2.0f - mx - mn) + 1e-2f), acc[2]);
      }
    }
  };
  if constexpr (VEC) {
    const T* const ins[1] = {xi};
    stream_groups<T, 1, false, true, IO>(ins, (T*)nullptr, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                      [&](float (&v)[1][PPL * 3], int g) { compute(v[0], g); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3];
      load_slow<T>(xi, g, hw, v);
      compute(v, g);
    }
  }
  block_reduce_record<3>(acc, records + size_t(n) * gridDim.x * kWsSlots);
}

template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads) void penalty_kernel(const T* __restrict__ y, float* __restrict__ records,
                                                           int hw, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y;
  const T* yi = y + size_t(n) * hw * 3;
  float acc[1] = {0.f};
  const int stride = gridDim.x * kThreads;
  auto compute = [&](const float* v) {
#pragma unroll
    for (int j = 0; j < PPL * 3; ++j) {
      const float o = fmaxf(v[j] - 1.0f, 0.0f);  // padding lanes hold 0 -> contribute 0
      acc[0] = fmaf(o, o, acc[0]);
    }
  };
  if constexpr (VEC) {
    const T* const ins[1] = {yi};
    stream_groups<T, 1, false, true, IO>(ins, (T*)nullptr, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                      [&](float (&v)[1][PPL * 3], int) { compute(v[0]); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3];
      load_slow<T>(yi, g, hw, v);
      compute(v);
    }
  }
  block_reduce_record<1>(acc, records + size_t(n) * gridDim.x * kWsSlots);
}

// d penalty / d y (agent.py:249-251): dy = 2 max(y - 1, 0) dpenalty[n] / (H W 3).  A map, 12 B/px.
template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads) void penalty_bwd_kernel(const T* __restrict__ y, const float* __restrict__ dpen,
                                                               T* __restrict__ dy, int hw, int groups, float inv_count) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const float sc = 2.0f * inv_count * dpen[n];
  const int stride = gridDim.x * kThreads;
  auto compute = [&](float* v) {
#pragma unroll
    for (int j = 0; j < PPL * 3; ++j) v[j] = fmaxf(v[j] - 1.0f, 0.0f) * sc;
  };
  if constexpr (VEC) {
    const T* const ins[1] = {y + off};
    stream_groups<T, 1, true, false, IO>(ins, dy + off, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                         [&](float (&v)[1][PPL * 3], int) { compute(v[0]); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3];
      load_slow<T>(y + off, g, hw, v);
      compute(v);
      store_slow<T>(dy + off, g, hw, v);
    }
  }
}

// ---------------------------------------------- critic statistics: first and second derivatives
// The critic's three statistics sit INSIDE the training graph: the generator's reward flows through
// critic(fake_output) (net.py:68-90), and the WGAN-GP term differentiates D(x^) with respect to x^ and then that
// gradient with respect to theta_c (net.py:174-194) -- a double backward through everything the critic does to its
// input, the statistics planes included.  Three kernels cover it (stats = S(x), J = dS/dx, per image):
//   stats_bwd   dx      = J^T g                      (g = dL/dstats, [N][3])              map, 12 B/px
//   stats_jvp   jv      = J v                        (v an image; = d<dx, v>/dg)          read-only reduction, 12 B/px
//   stats_hvp   out     = d<J^T g, v>/dx             (the second-order term in x)         map, 18 B/px
// Conventions (oracle/nets_np.py::stat_features_backward): tf.nn.moments is the population variance (its mean is
// a stop_gradient in TF, which leaves the first derivative unchanged); reduce_max / reduce_min split the gradient
// evenly between tied channels; clip_by_value passes on 0 <= x <= 1 inclusive; tf.minimum(x=a, y=b) sends ties to a.
struct SatPix {
  float a[3], b[3];       // d max / d x_j and d min / d x_j
  float fmx, fmn;         // d sat / d max, d sat / d min
  float hxx, hxn, hnn;    // second derivatives of sat in (max, min)   (HESS only)
};
template <bool HESS>
__device__ __forceinline__ SatPix sat_pix(const float p[3]) {
  SatPix s;
  float c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) c[j] = clamp01x(p[j], 0.f, 1.f);
  const float mx = fmaxf(fmaxf(c[0], c[1]), c[2]), mn = fminf(fminf(c[0], c[1]), c[2]);
  const int nmax = (c[0] == mx) + (c[1] == mx) + (c[2] == mx), nmin = (c[0] == mn) + (c[1] == mn) + (c[2] == mn);
  const float imax = nmax == 1 ? 1.0f : (nmax == 2 ? 0.5f : 1.0f / 3.0f);
  const float imin = nmin == 1 ? 1.0f : (nmin == 2 ? 0.5f : 1.0f / 3.0f);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const bool in = p[j] >= 0.f && p[j] <= 1.f;
    s.a[j] = (in && c[j] == mx) ? imax : 0.f;
    s.b[j] = (in && c[j] == mn) ? imin : 0.f;
  }
  const float sa = mx + mn, sb = 2.0f - mx - mn;
  const bool use_a = sa <= sb;
  const float sg = use_a ? 1.0f : -1.0f;
  const float r = fast_rcp((use_a ? sa : sb) + 1e-2f);  // denominator >= 1e-2
  const float num = mx - mn;
  const float t = num * sg * r * r;
  s.fmx = r - t;
  s.fmn = -r - t;
  if constexpr (HESS) {
    const float q = 2.0f * num * r * r * r, e = 2.0f * sg * r * r;
    s.hxx = q - e;
    s.hxn = q;
    s.hnn = q + e;
  }
  return s;
}

template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads) void stats_bwd_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                             const float* __restrict__ dstats, T* __restrict__ dx,
                                                             int hw, int groups, float inv_hw) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const float mean = stats[n * 3];
  const float g0 = dstats[n * 3] * inv_hw, g1 = dstats[n * 3 + 1] * 2.0f * inv_hw, g2 = dstats[n * 3 + 2] * inv_hw;
  const int stride = gridDim.x * kThreads;
  auto compute = [&](float* v) {
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      float* p = v + 3 * k;
      const float l = (p[0] * kLumR + p[1] * kLumG) + p[2] * kLumB + 1e-5f;
      const float dl = fmaf(g1, l - mean, g0);
      const SatPix s = sat_pix<false>(p);
      p[0] = fmaf(kLumR, dl, g2 * (s.fmx * s.a[0] + s.fmn * s.b[0]));
      p[1] = fmaf(kLumG, dl, g2 * (s.fmx * s.a[1] + s.fmn * s.b[1]));
      p[2] = fmaf(kLumB, dl, g2 * (s.fmx * s.a[2] + s.fmn * s.b[2]));
    }
  };
  if constexpr (VEC) {
    const T* const ins[1] = {x + off};
    stream_groups<T, 1, true, false, IO>(ins, dx + off, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                         [&](float (&v)[1][PPL * 3], int) { compute(v[0]); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3];
      load_slow<T>(x + off, g, hw, v);
      compute(v);
      store_slow<T>(dx + off, g, hw, v);
    }
  }
}

template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads) void stats_jvp_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                             const T* __restrict__ vimg, float* __restrict__ records,
                                                             int hw, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const float mean = stats[n * 3];
  float acc[3] = {0.f, 0.f, 0.f};
  const int stride = gridDim.x * kThreads;
  auto compute = [&](const float* xv, const float* vv) {  // padding pixels carry v = 0 and add nothing
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const float* p = xv + 3 * k;
      const float* d = vv + 3 * k;
      const float l = (p[0] * kLumR + p[1] * kLumG) + p[2] * kLumB + 1e-5f;
      const float wv = (d[0] * kLumR + d[1] * kLumG) + d[2] * kLumB;
      acc[0] += wv;
      acc[1] = fmaf(l - mean, wv, acc[1]);
      const SatPix s = sat_pix<false>(p);
      const float vm = s.a[0] * d[0] + s.a[1] * d[1] + s.a[2] * d[2];
      const float vn = s.b[0] * d[0] + s.b[1] * d[1] + s.b[2] * d[2];
      acc[2] += s.fmx * vm + s.fmn * vn;
    }
  };
  if constexpr (VEC) {
    const T* const ins[2] = {x + off, vimg + off};
    stream_groups<T, 2, false, true, IO>(ins, (T*)nullptr, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                         [&](float (&v)[2][PPL * 3], int) { compute(v[0], v[1]); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3], d[PPL * 3];
      load_slow<T>(x + off, g, hw, v);
      load_slow<T>(vimg + off, g, hw, d);
      compute(v, d);
    }
  }
  acc[1] *= 2.0f;
  block_reduce_record<3>(acc, records + size_t(n) * gridDim.x * kWsSlots);
}

template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads) void stats_hvp_kernel(const T* __restrict__ x, const float* __restrict__ dstats,
                                                             const float* __restrict__ jv, const T* __restrict__ vimg,
                                                             T* __restrict__ out, int hw, int groups, float inv_hw) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const float jv0 = jv[n * 3];  // mean over the image of w . v
  const float g1 = dstats[n * 3 + 1] * 2.0f * inv_hw, g2 = dstats[n * 3 + 2] * inv_hw;
  const int stride = gridDim.x * kThreads;
  auto compute = [&](const float* xv, float* d) {
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const float* p = xv + 3 * k;
      float* q = d + 3 * k;
      const float wv = (q[0] * kLumR + q[1] * kLumG) + q[2] * kLumB;
      const float dv = g1 * (wv - jv0);  // variance: d/dx of (2/HW) sum (l - mean)(w . v)
      const SatPix s = sat_pix<true>(p);
      const float vm = s.a[0] * q[0] + s.a[1] * q[1] + s.a[2] * q[2];
      const float vn = s.b[0] * q[0] + s.b[1] * q[1] + s.b[2] * q[2];
      const float dfx = g2 * (s.hxx * vm + s.hxn * vn), dfn = g2 * (s.hxn * vm + s.hnn * vn);
      q[0] = fmaf(kLumR, dv, s.a[0] * dfx + s.b[0] * dfn);
      q[1] = fmaf(kLumG, dv, s.a[1] * dfx + s.b[1] * dfn);
      q[2] = fmaf(kLumB, dv, s.a[2] * dfx + s.b[2] * dfn);
    }
  };
  if constexpr (VEC) {
    const T* const ins[2] = {x + off, vimg + off};
    stream_groups<T, 2, true, false, IO>(ins, out + off, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                         [&](float (&v)[2][PPL * 3], int) { compute(v[0], v[1]); });
  } else {
    for (int g = blockIdx.x * kThreads + threadIdx.x; g < groups; g += stride) {
      float v[PPL * 3], d[PPL * 3];
      load_slow<T>(x + off, g, hw, v);
      load_slow<T>(vimg + off, g, hw, d);
      compute(v, d);
      store_slow<T>(out + off, g, hw, d);
    }
  }
}

// ---- the middle of the hand-scheduled critic update in ONE launch (round 6, exposure_amd/critic_direct.py) --------------
// Between the first layer's data gradient u (6 planes: image + statistics planes, critics.py:64-76) and the tangent pass of
// the gradient penalty's double backward (net.py:174-194) sit, per interpolated image: the gradient reaching the
// statistics (plane sums), J^T of the statistics (stats_bwd), g = u[..., :3] + J^T gs, ||g||, the one-sided term and its
// gradient v = scale 2 max(||g|| - 1, 0) / ||g|| g (net.py:185-187), J v (stats_jvp) and the tangent's 6-plane input
// [v | J v broadcast].  Six launches of 5-7 us each as separate kernels; here one block per image walks its pixels three
// times (u and x stay in L2) with a fixed-order block reduction after each walk.  x is float32 (the critic's input).
template <int WAVES = 16>
__device__ __forceinline__ void block_sum3(float (&a)[3], float (*part)[3]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = a[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) part[wave][k] = v;
  }
  __syncthreads();
  // lane w holds wave w's partial; a butterfly leaves the same total (one fixed order) in every lane.  (The first version let
  // every thread add the 16 partials one after the other: a rolled loop of dependent LDS reads, ~2.5 us per call.)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = lane < WAVES ? part[lane][k] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    a[k] = v;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void critic_penalty_tangent_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                      const float* __restrict__ stats, float scale,
                                                                      float* __restrict__ t0, float* __restrict__ norm,
                                                                      float* __restrict__ term, int hw) {
  __shared__ float part[16][3];
  const int n = blockIdx.x;
  const float* un = u + size_t(n) * hw * 6;
  const float* xn = x + size_t(n) * hw * 3;
  float* tn = t0 + size_t(n) * hw * 6;
  const float inv_hw = 1.0f / float(hw), mean = stats[n * 3];
  // walk 1: the gradient reaching the three statistics = the sums of u's statistics planes
  float gs[3] = {0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < hw; p += 1024) {
    gs[0] += un[p * 6 + 3];
    gs[1] += un[p * 6 + 4];
    gs[2] += un[p * 6 + 5];
  }
  block_sum3(gs, part);
  const float g0 = gs[0] * inv_hw, g1 = gs[1] * 2.0f * inv_hw, g2 = gs[2] * inv_hw;
  auto grad_at = [&](int p, float (&g)[3], float& lm, SatPix& sp) {  // g = u[:3] + J^T gs at pixel p (stats_bwd_kernel)
    const float px[3] = {xn[p * 3], xn[p * 3 + 1], xn[p * 3 + 2]};
    const float l = (px[0] * kLumR + px[1] * kLumG) + px[2] * kLumB + 1e-5f;
    lm = l - mean;
    const float dl = fmaf(g1, lm, g0);
    sp = sat_pix<false>(px);
    g[0] = un[p * 6 + 0] + fmaf(kLumR, dl, g2 * (sp.fmx * sp.a[0] + sp.fmn * sp.b[0]));
    g[1] = un[p * 6 + 1] + fmaf(kLumG, dl, g2 * (sp.fmx * sp.a[1] + sp.fmn * sp.b[1]));
    g[2] = un[p * 6 + 2] + fmaf(kLumB, dl, g2 * (sp.fmx * sp.a[2] + sp.fmn * sp.b[2]));
  };
  // walk 2: ||g||^2
  float sq[3] = {0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < hw; p += 1024) {
    float g[3], lm;
    SatPix sp;
    grad_at(p, g, lm, sp);
    sq[0] = fmaf(g[0], g[0], sq[0]);
    sq[1] = fmaf(g[1], g[1], sq[1]);
    sq[2] = fmaf(g[2], g[2], sq[2]);
  }
  block_sum3(sq, part);
  const float nm = sqrtf(1e-6f + ((sq[0] + sq[1]) + sq[2]));
  const float over = fmaxf(nm - 1.0f, 0.0f);
  const float coef = scale * 2.0f * over / nm;
  if (threadIdx.x == 0) {
    norm[n] = nm;
    term[n] = over * over;
  }
  // walk 3: v = coef g (the tangent's image planes) and J v (stats_jvp_kernel)
  float jv[3] = {0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < hw; p += 1024) {
    float g[3], lm;
    SatPix sp;
    grad_at(p, g, lm, sp);
    const float v[3] = {g[0] * coef, g[1] * coef, g[2] * coef};
    tn[p * 6 + 0] = v[0];
    tn[p * 6 + 1] = v[1];
    tn[p * 6 + 2] = v[2];
    const float wv = (v[0] * kLumR + v[1] * kLumG) + v[2] * kLumB;
    jv[0] += wv;
    jv[1] = fmaf(lm, wv, jv[1]);
    jv[2] += sp.fmx * (sp.a[0] * v[0] + sp.a[1] * v[1] + sp.a[2] * v[2]) + sp.fmn * (sp.b[0] * v[0] + sp.b[1] * v[1] + sp.b[2] * v[2]);
  }
  block_sum3(jv, part);
  const float j0 = jv[0] * inv_hw, j1 = jv[1] * 2.0f * inv_hw, j2 = jv[2] * inv_hw;
  for (int p = threadIdx.x; p < hw; p += 1024) {
    tn[p * 6 + 3] = j0;
    tn[p * 6 + 4] = j1;
    tn[p * 6 + 5] = j2;
  }
}

// The same for images of at most 4096 pixels (the 64 x 64 proxies of a training iteration): a thread OWNS four pixels and
// keeps what the three walks share -- u[..., :3], x, then g -- in registers, so u and x are read once (the three-walk kernel
// above reads them three times from L2; with one block per image the launch is bound by what ONE CU can stream) and the
// six planes of the tangent's input leave as 24 contiguous bytes per pixel after the last reduction.
__global__ __launch_bounds__(1024) void critic_penalty_tangent_reg_kernel(const float* __restrict__ u, const float* __restrict__ x,
                                                                          const float* __restrict__ stats, float scale,
                                                                          float* __restrict__ t0, float* __restrict__ norm,
                                                                          float* __restrict__ term, int hw) {
  __shared__ float part[16][3];
  const int n = blockIdx.x;
  const float* un = u + size_t(n) * hw * 6;
  const float* xn = x + size_t(n) * hw * 3;
  float* tn = t0 + size_t(n) * hw * 6;
  const float inv_hw = 1.0f / float(hw), mean = stats[n * 3];
  float g[4][3], px[4][3];
  float gs[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = threadIdx.x + 1024 * i;
    if (p < hw) {
      const float2 u01 = *reinterpret_cast<const float2*>(un + p * 6), u23 = *reinterpret_cast<const float2*>(un + p * 6 + 2),
                   u45 = *reinterpret_cast<const float2*>(un + p * 6 + 4);
      g[i][0] = u01.x, g[i][1] = u01.y, g[i][2] = u23.x;
      gs[0] += u23.y, gs[1] += u45.x, gs[2] += u45.y;
      px[i][0] = xn[p * 3], px[i][1] = xn[p * 3 + 1], px[i][2] = xn[p * 3 + 2];
    }
  }
  block_sum3(gs, part);
  const float g0 = gs[0] * inv_hw, g1 = gs[1] * 2.0f * inv_hw, g2 = gs[2] * inv_hw;
  float sq[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (threadIdx.x + 1024 * i < hw) {  // g = u[:3] + J^T gs (stats_bwd_kernel)
      const float l = (px[i][0] * kLumR + px[i][1] * kLumG) + px[i][2] * kLumB + 1e-5f;
      const float dl = fmaf(g1, l - mean, g0);
      const SatPix sp = sat_pix<false>(px[i]);
      g[i][0] += fmaf(kLumR, dl, g2 * (sp.fmx * sp.a[0] + sp.fmn * sp.b[0]));
      g[i][1] += fmaf(kLumG, dl, g2 * (sp.fmx * sp.a[1] + sp.fmn * sp.b[1]));
      g[i][2] += fmaf(kLumB, dl, g2 * (sp.fmx * sp.a[2] + sp.fmn * sp.b[2]));
      sq[0] = fmaf(g[i][0], g[i][0], sq[0]);
      sq[1] = fmaf(g[i][1], g[i][1], sq[1]);
      sq[2] = fmaf(g[i][2], g[i][2], sq[2]);
    }
  }
  block_sum3(sq, part);
  const float nm = sqrtf(1e-6f + ((sq[0] + sq[1]) + sq[2]));
  const float over = fmaxf(nm - 1.0f, 0.0f);
  const float coef = scale * 2.0f * over / nm;
  if (threadIdx.x == 0) {
    norm[n] = nm;
    term[n] = over * over;
  }
  float jv[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (threadIdx.x + 1024 * i < hw) {  // v = coef g and J v (stats_jvp_kernel)
      g[i][0] *= coef, g[i][1] *= coef, g[i][2] *= coef;
      const float* v = g[i];
      const float l = (px[i][0] * kLumR + px[i][1] * kLumG) + px[i][2] * kLumB + 1e-5f;
      const SatPix sp = sat_pix<false>(px[i]);
      const float wv = (v[0] * kLumR + v[1] * kLumG) + v[2] * kLumB;
      jv[0] += wv;
      jv[1] = fmaf(l - mean, wv, jv[1]);
      jv[2] += sp.fmx * (sp.a[0] * v[0] + sp.a[1] * v[1] + sp.a[2] * v[2]) + sp.fmn * (sp.b[0] * v[0] + sp.b[1] * v[1] + sp.b[2] * v[2]);
    }
  }
  block_sum3(jv, part);
  const float j0 = jv[0] * inv_hw, j1 = jv[1] * 2.0f * inv_hw, j2 = jv[2] * inv_hw;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = threadIdx.x + 1024 * i;
    if (p < hw) {
      *reinterpret_cast<float2*>(tn + p * 6) = make_float2(g[i][0], g[i][1]);
      *reinterpret_cast<float2*>(tn + p * 6 + 2) = make_float2(g[i][2], j0);
      *reinterpret_cast<float2*>(tn + p * 6 + 4) = make_float2(j1, j2);
    }
  }
}

// ---- the input side of a hand-scheduled net pass in ONE launch (round 6, exposure_amd/critic_direct.py, generator_direct.py)
// critics.py:42-76 in front of `cnn`: the batch [a | b (| a + alpha (b - a))] as float32, its statistics {mean luminance,
// luminance variance, mean saturation} and `concat([images, states planes, statistics planes]) - 0.5`.  As separate launches
// that was expo_gp_inputs -> expo_critic_stats (streaming pass + finish launch) -> (torch.cat of the states) ->
// expo_planes_concat: 25 us and four dependent launches for 64 x 64 proxies, of which the statistics' two launches exist only
// because a block sees part of an image.  Here a block holds its WHOLE image in LDS (<= 4096 pixels = 48 KB): it reads the
// source rows once (16-byte loads straight from the data set / the replay memory's pool: `rows`), forms the three sums in a
// fixed order, and writes its share (blockIdx.y of gridDim.y: the loads are cheap, the stores are the traffic) of the
// (3 + V + 3)-plane tensor with whole-line stores -- and of the float32 image for the rows whose backward needs it
// (x_first .. x_first + x_count: the interpolated rows of a critic update, the retouched rows of a G / V step).
constexpr int kNetInputsMaxPixels = 4096;
struct NetInputsArgs {
  const void *a, *b;                  // images [.][hw][3] in T
  const long long *a_rows, *b_rows;   // nullable: image j of the block is row rows[j]
  const float* alpha;                 // nullable: [n] -> a third block of rows a + alpha (b - a) (net.py:170-172)
  const float *vec_a, *vec_b;         // nullable: [n][v0] per-image values of the two blocks (the value net's states)
  float *planes, *stats, *x_out;      // [m][hw][3 + v0 + 3]; [m][3]; nullable [x_count][hw][3]
  int n, hw, v0, x_first, x_count;
  float offset;
  unsigned magic;                     // ceil(2^32 / (3 + v0 + 3)): element -> pixel by a multiply-high (exact below 2^32 / C)
};
template <typename T, int NT>
__global__ __launch_bounds__(NT) void net_inputs_kernel(const NetInputsArgs g) {
  typedef T Vec4 __attribute__((ext_vector_type(4)));
  __shared__ float px[kNetInputsMaxPixels * 3];
  __shared__ float part[NT / 64][3];
  __shared__ float vs[64];
  const int img = blockIdx.x, blk = img / g.n, j = img - blk * g.n;
  const int elems = g.hw * 3;
  const T* pa = static_cast<const T*>(g.a) + size_t(g.a_rows ? g.a_rows[j] : j) * elems;
  const T* pb = static_cast<const T*>(g.b) + size_t(g.b_rows ? g.b_rows[j] : j) * elems;
  const bool vec4 = (elems & 3) == 0 && ((uintptr_t(g.a) | uintptr_t(g.b)) & (sizeof(Vec4) - 1)) == 0;
  // ---- the image, float32, into LDS
  if (blk == 2) {
    const float al = g.alpha[j];
    if (vec4) {
      for (int e = threadIdx.x * 4; e < elems; e += NT * 4) {
        const Vec4 r = *reinterpret_cast<const Vec4*>(pa + e), f = *reinterpret_cast<const Vec4*>(pb + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) px[e + k] = float(r[k]) + al * (float(f[k]) - float(r[k]));
      }
    } else {
      for (int e = threadIdx.x; e < elems; e += NT) px[e] = float(pa[e]) + al * (float(pb[e]) - float(pa[e]));
    }
  } else {
    const T* src = blk ? pb : pa;
    if (vec4) {
      for (int e = threadIdx.x * 4; e < elems; e += NT * 4) {
        const Vec4 r = *reinterpret_cast<const Vec4*>(src + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) px[e + k] = float(r[k]);
      }
    } else {
      for (int e = threadIdx.x; e < elems; e += NT) px[e] = float(src[e]);
    }
  }
  __syncthreads();
  // ---- statistics (stats_kernel's arithmetic; finish_kernel's kFinStats): shifted sums, one fixed-order block reduction
  float acc[3] = {0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < g.hw; p += NT) {
    const float* q = px + 3 * p;
    const float l = ((q[0] * kLumR + q[1] * kLumG) + q[2] * kLumB + 1e-5f) - 0.5f;
    acc[0] += l;
    acc[1] = fmaf(l, l, acc[1]);
    const float c0 = clamp01x(q[0], 0.f, 1.f), c1 = clamp01x(q[1], 0.f, 1.f), c2 = clamp01x(q[2], 0.f, 1.f);
    const float mx = fmaxf(fmaxf(c0, c1), c2), mn = fminf(fminf(c0, c1), c2);
    acc[2] = fmaf(mx - mn, fast_rcp(fminf(mx + mn, 2.0f - mx - mn) + 1e-2f), acc[2]);
  }
  block_sum3<NT / 64>(acc, part);
  const float inv_hw = 1.0f / float(g.hw);
  const float m1 = acc[0] * inv_hw, m2 = acc[1] * inv_hw;
  const float st[3] = {m1 + 0.5f, m2 - m1 * m1, acc[2] * inv_hw};
  if (threadIdx.x < 3) {
    const float s = threadIdx.x == 0 ? st[0] : (threadIdx.x == 1 ? st[1] : st[2]);
    if (blockIdx.y == 0) g.stats[size_t(img) * 3 + threadIdx.x] = s;
    vs[g.v0 + threadIdx.x] = s - g.offset;
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + unsigned(g.v0)) {
    const int k = threadIdx.x - 64;
    vs[k] = (blk ? g.vec_b : g.vec_a)[size_t(j) * g.v0 + k] - g.offset;
  }
  __syncthreads();
  // ---- this block's share of the planes: consecutive output floats, element e -> (pixel, channel)
  const unsigned C = 3 + g.v0 + 3;
  const unsigned count = unsigned(g.hw) * C;
  // (16 bytes per lane and store: the epilogue is store-ISSUE-bound, a dword per lane runs at a third of the rate)
  const unsigned per = ((count + gridDim.y - 1) / gridDim.y + 255) & ~255u;
  const unsigned e0 = blockIdx.y * per, e1 = min(count, e0 + per);
  float* o = g.planes + size_t(img) * count;
  auto value = [&](unsigned p, unsigned c) { return c < 3 ? px[p * 3 + c] - g.offset : vs[c - 3]; };
  if ((count & 3) == 0) {
    for (unsigned e = e0 + threadIdx.x * 4; e < e1; e += NT * 4) {
      unsigned p = __umulhi(e, g.magic), c = e - p * C;
      float4 v;
      v.x = value(p, c);
      if (++c == C) { c = 0; ++p; }
      v.y = value(p, c);
      if (++c == C) { c = 0; ++p; }
      v.z = value(p, c);
      if (++c == C) { c = 0; ++p; }
      v.w = value(p, c);
      *reinterpret_cast<float4*>(o + e) = v;
    }
  } else {
    for (unsigned e = e0 + threadIdx.x; e < e1; e += NT) {
      const unsigned p = __umulhi(e, g.magic), c = e - p * C;
      o[e] = value(p, c);
    }
  }
  // ---- and of the float32 image, for the rows whose backward reads it
  if (g.x_out && img >= g.x_first && img < g.x_first + g.x_count) {
    float* xo = g.x_out + size_t(img - g.x_first) * elems;
    const unsigned xper = ((unsigned(elems) + gridDim.y - 1) / gridDim.y + 255) & ~255u;
    const unsigned x0 = blockIdx.y * xper, x1 = min(unsigned(elems), x0 + xper);
    if ((elems & 3) == 0) {
      for (unsigned e = x0 + threadIdx.x * 4; e < x1; e += NT * 4)
        *reinterpret_cast<float4*>(xo + e) = *reinterpret_cast<const float4*>(px + e);
    } else {
      for (unsigned e = x0 + threadIdx.x; e < x1; e += NT) xo[e] = px[e];
    }
  }
}

// ------------------------------------------------------------------------------- finish
// One wave per (image, step): adds the image's bx block records in a fixed order and writes the final
// per-image values.  Steps of a chain (or the single step of any other entry point) are described by
// value in the kernel arguments.
enum FinishKind : int { kFinFilter = 0, kFinDispatch = 1, kFinApply = 2, kFinStats = 3, kFinPenalty = 4, kFinScaled = 5, kFinVignet = 6, kFinApplyDispatch = 7, kFinChainFused = 8 };
struct FinishStep {
  const float* params;    // [n][P] (filter / apply), [n][EXPO_MAX_PARAMS] (dispatch)
  float* out;             // dparams [n][P] | [n][24]; stats [n][3]; penalty [n]
  float* out2;            // apply: dmask_params [n][6]
  const float* records;   // [n][bx][kWsSlots]
  const int32_t* ids;     // dispatch: per-image filter ids
  int kind, filter_id, accumulate;  // kFinChainFused: filter_id = steps of the sequence (row stride multiplier)
  float scale;            // stats: 1 / (H W); penalty: 1 / (H W 3)
  int bx;                 // records per image (the streaming kernel's gridDim.x)
};
constexpr int kMaxFinishSteps = 12;
struct FinishArgs { FinishStep s[kMaxFinishSteps]; };

template <class F>
__device__ __forceinline__ void finish_filter(const FinishStep& st, const float* prm, const float* tot, float* dprm,
                                              int nout, int lane) {
  if (lane < nout) {
    const float g = lane < F::NP ? F::finish_one(prm, tot, lane) : 0.0f;
    dprm[lane] = st.accumulate ? dprm[lane] + g : g;
  }
}

constexpr int kFinishThreads = 256;
__global__ __launch_bounds__(kFinishThreads) void finish_kernel(const FinishArgs args) {
  const FinishStep st = args.s[blockIdx.y];
  const int bx = st.bx;
  const int n = blockIdx.x, lane = threadIdx.x & 63;
  __shared__ float part[kFinishThreads / kWsSlots][kWsSlots];
  __shared__ float tot[kWsSlots];
  int fid = st.filter_id;
  // expo_chain_fused_bwd: ids [n][steps], params / dparams [n][steps][EXPO_MAX_PARAMS], this launch row = one step
  const int seq = st.kind == kFinChainFused ? st.filter_id : 1;
  if (st.ids) fid = st.ids[size_t(n) * seq];  // per-image choice (dispatch entry points)
  const bool nothing = st.ids && (fid < 0 || fid >= EXPO_NUM_FILTERS);  // id -1: the image wrote no records
  // fixed summation order: thread (g, j) adds the records b = g, g + 8, g + 16, ... of slot j (independent loads,
  // all in flight together -- the first version walked them in one dependent chain and took 5.3 us), then the 8
  // partial sums of a slot are added in order g = 0..7
  const int j = threadIdx.x & (kWsSlots - 1), g = threadIdx.x / kWsSlots;
  constexpr int kGroups = kFinishThreads / kWsSlots;
  float s = 0.f;
  if (!nothing) {
    const float* r = st.records + (size_t(n) * bx) * kWsSlots + j;
    int b = g;
    for (; b + 3 * kGroups < bx; b += 4 * kGroups) {
      const float a0 = r[size_t(b) * kWsSlots], a1 = r[size_t(b + kGroups) * kWsSlots];
      const float a2 = r[size_t(b + 2 * kGroups) * kWsSlots], a3 = r[size_t(b + 3 * kGroups) * kWsSlots];
      s += a0;
      s += a1;
      s += a2;
      s += a3;
    }
    for (; b < bx; b += kGroups) s += r[size_t(b) * kWsSlots];
  }
  part[g][j] = s;
  __syncthreads();
  if (threadIdx.x >= 64) return;  // wave 0 finishes
  if (lane < kWsSlots) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < kGroups; ++k) v += part[k][lane];
    tot[lane] = v;
  }
  __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations execute in order
  switch (st.kind) {
    case kFinStats: {  // critics.py:51-62: mean, population variance (tf.nn.moments), mean saturation
      if (lane < 3) {
        const float m = tot[0] * st.scale, m2 = tot[1] * st.scale;
        st.out[n * 3 + lane] = lane == 0 ? m + 0.5f : (lane == 1 ? m2 - m * m : tot[2] * st.scale);
      }
    } break;
    case kFinPenalty:  // agent.py:249-251 (also the fused penalty of the dispatch forward; id -1 -> 0)
      if (lane == 0) st.out[n] = nothing ? 0.0f : tot[0] * st.scale;
      break;
    case kFinVignet:  // d mask parameters of VignetFilter.apply (scale = maximum_sharpness; masking off -> 0)
      if (lane < 5) st.out[n * 5 + lane] = st.accumulate ? VignetPrm::finish_one(st.params + n * 5, st.scale, tot, lane) : 0.0f;
      break;
    case kFinScaled:  // out[n][0..filter_id) = totals * scale (the statistics' Jacobian-vector product)
      if (lane < st.filter_id) st.out[n * st.filter_id + lane] = tot[lane] * st.scale;
      break;
    default: {
      const bool disp = st.kind == kFinDispatch || st.kind == kFinApplyDispatch || st.kind == kFinChainFused;
      const bool masked = st.kind == kFinApply || st.kind == kFinApplyDispatch;
      const int stride = disp ? EXPO_MAX_PARAMS : 0;
      if (nothing) {
        if (lane < EXPO_MAX_PARAMS) st.out[size_t(n) * seq * EXPO_MAX_PARAMS + lane] = 0.0f;
        if (masked && lane < 6) st.out2[n * 6 + lane] = 0.0f;
        break;
      }
#define EXPO_FIN(ID, F)                                                                                        \
  case ID: {                                                                                                   \
    const int row = disp ? stride : F::NP;                                                                     \
    const size_t at = size_t(n) * seq * row;                                                                   \
    finish_filter<F>(st, st.params + at, tot, st.out + at, row, lane);                                         \
    if (masked && lane < 6) st.out2[n * 6 + lane] = tot[F::NACC + lane];                                      \
  } break;
      switch (fid) {
        EXPO_FIN(0, ExposureF)
        EXPO_FIN(1, GammaF)
        EXPO_FIN(2, WhiteBalanceF)
        EXPO_FIN(3, SatPlusF)
        EXPO_FIN(4, ToneF)
        EXPO_FIN(5, ContrastF)
        EXPO_FIN(6, WnbF)
        EXPO_FIN(7, ColorF)
        EXPO_FIN(8, LevelF)
      }
#undef EXPO_FIN
    } break;
  }
}

#ifdef EXPO_PROBE
// register-pressure probe builds (tools/probe.sh): instantiate a few kernels, skip the host side
#define EXPO_PROBE_BWD(F)                                                                                         \
  template __global__ void filter_bwd_kernel<F, half_t, true, true, 0, IoStream>(const half_t*, const half_t*, half_t*, \
                                                                                 const float*, float*, int, int, int);
EXPO_PROBE_BWD(ToneF)
EXPO_PROBE_BWD(ColorF)
EXPO_PROBE_BWD(WnbF)
EXPO_PROBE_BWD(ExposureF)
#undef EXPO_PROBE_BWD
#define EXPO_PROBE_FWD(F)                                                                                     \
  template __global__ void filter_fwd_kernel<F, half_t, true, IoStream>(const half_t*, half_t*, const float*, int, int, int);
EXPO_PROBE_FWD(ExposureF)
EXPO_PROBE_FWD(GammaF)
EXPO_PROBE_FWD(SatPlusF)
EXPO_PROBE_FWD(ColorF)
#undef EXPO_PROBE_FWD
#define EXPO_PROBE_APPLY(F)                                                                                   \
  template __global__ void apply_bwd_kernel<F, half_t, true, true, 0>(const half_t*, const half_t*, half_t*, \
                                                                      const float*, const float*, float*,   \
                                                                      float, float, int, int, int);
EXPO_PROBE_APPLY(ExposureF)
EXPO_PROBE_APPLY(ContrastF)
EXPO_PROBE_APPLY(SatPlusF)
EXPO_PROBE_APPLY(ToneF)
EXPO_PROBE_APPLY(ColorF)
#undef EXPO_PROBE_APPLY
}  // namespace expo
#else
// ==================================================================== host side
thread_local std::string g_err;
// ---- workspace of the reducing kernels: float records[steps][n][bx][kWsSlots]; no initialisation needed
static size_t ws_step_bytes(int n, int blocks_x) { return size_t(n) * size_t(blocks_x) * kWsSlots * sizeof(float); }
static int ws_check(void* workspace, size_t workspace_bytes, int n, int blocks_x, int steps, float** records) {
  *records = nullptr;
  if (!workspace) return fail(EXPO_E_BADARG, "workspace is NULL (size it with expo_workspace_bytes)");
  if ((reinterpret_cast<uintptr_t>(workspace) & 3) != 0) return fail(EXPO_E_BADARG, "workspace must be 4-byte aligned");
  if (workspace_bytes < ws_step_bytes(n, blocks_x) * size_t(steps))
    return fail(EXPO_E_BADARG, "workspace too small (expo_workspace_bytes, times the steps of a chain)");
  *records = static_cast<float*>(workspace);
  return EXPO_OK;
}

static int geom_bx(int kind, int n, int h, int w, int dtype) {
  return dtype == EXPO_F16 ? make_geom<half_t>(n, h, w, {}, kind).blocks_x : make_geom<float>(n, h, w, {}, kind).blocks_x;
}
// the largest per-image record count any reducing entry point uses: the stride of a chain's per-step slices
static int geom_bx_max(int n, int h, int w, int dtype) {
  int bx = 1;
  for (int kind = kGeomReduce; kind < kNumGeomKinds; ++kind) {
    const int b = geom_bx(kind, n, h, w, dtype);
    if (b > bx) bx = b;
  }
  return bx;
}

static int launch_finish(const FinishArgs& args, int steps, int n, hipStream_t s) {
  hipLaunchKernelGGL(finish_kernel, dim3(n, steps), dim3(kFinishThreads), 0, s, args);
  HIP_TRY(hipGetLastError(), "finish launch");
  return EXPO_OK;
}

// (declared in host_common.h for chain_fused_bwd.hip)
int chain_records(void* workspace, size_t workspace_bytes, int n, int h, int w, int dtype, int steps, float** records,
                  size_t* step_floats) {
  const int bx_max = geom_bx_max(n, h, w, dtype);
  *step_floats = ws_step_bytes(n, bx_max) / sizeof(float);
  return ws_check(workspace, workspace_bytes, n, bx_max, steps, records);
}
int finish_chain_fused(const int32_t* ids, const float* params, float* dparams, const float* records,
                       size_t step_floats, int steps, int n, int blocks_x, hipStream_t s) {
  static_assert(EXPO_FUSED_BWD_MAX_STEPS <= kMaxFinishSteps, "one finish launch serves the whole sequence");
  FinishArgs fa{};
  for (int k = 0; k < steps; ++k)
    fa.s[k] = FinishStep{params + size_t(k) * EXPO_MAX_PARAMS, dparams + size_t(k) * EXPO_MAX_PARAMS, nullptr,
                         records + size_t(k) * step_floats, ids + k, kFinChainFused, steps, 0, 0.f, blocks_x};
  return launch_finish(fa, steps, n, s);
}

template <class F, typename T>
static int launch_fwd(const void* x, void* y, const float* params, int n, int h, int w, hipStream_t s, int rev,
                      int n_geom) {
  // n_geom: the batch size the launch geometry (blocks per image, cache policy) is chosen for -- the whole batch when
  // a chain runs its two halves on two streams, so both halves and the finish launch agree on the record count
  const Geom g = make_geom<T>(n_geom, h, w, {x, y}, kGeomMap);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  if (g.stream)
    hipLaunchKernelGGL((filter_fwd_kernel<F, T, true, IoStream>), grid, block, 0, s, (const T*)x, (T*)y, params, g.hw, g.groups, rev);
  else if (g.vec)
    hipLaunchKernelGGL((filter_fwd_kernel<F, T, true, IoCached>), grid, block, 0, s, (const T*)x, (T*)y, params, g.hw, g.groups, rev);
  else
    hipLaunchKernelGGL((filter_fwd_kernel<F, T, false, IoCached>), grid, block, 0, s, (const T*)x, (T*)y, params, g.hw, g.groups, rev);
  HIP_TRY(hipGetLastError(), "filter_fwd launch");
  return EXPO_OK;
}

template <class F, typename T>
static int launch_bwd(const void* x, const void* dy, void* dx, const float* params, float* records, int n,
                      int h, int w, int mode, hipStream_t s, int rev, int n_geom) {
  constexpr int kKind = F::kLutFloats > 0 ? (F::NP == kCurveSteps ? kGeomReduceTone : kGeomReduceColor) : kGeomReduce;
  const Geom g = make_geom<T>(n_geom, h, w, {x, dy, dx}, kKind);
  const dim3 grid(g.blocks_x, n), block(kThreads);
#define EXPO_LIO(VEC, HAS_DX, MODE, IO)                                                                  \
  hipLaunchKernelGGL((filter_bwd_kernel<F, T, VEC, HAS_DX, MODE, IO>), grid, block, 0, s, (const T*)x, \
                     (const T*)dy, (T*)dx, params, records, g.hw, g.groups, rev)
#define EXPO_L(VEC, HAS_DX, MODE)                                  \
  do {                                                             \
    if constexpr (VEC) {                                           \
      if (g.stream) EXPO_LIO(true, HAS_DX, MODE, IoStream);        \
      else EXPO_LIO(true, HAS_DX, MODE, IoCached);                 \
    } else {                                                       \
      EXPO_LIO(false, HAS_DX, MODE, IoCached);                     \
    }                                                              \
  } while (0)
  // only SaturationPlus has a mode-dependent backward
  const bool m1 = std::is_same<F, SatPlusF>::value && mode == 1;
  const int key = (g.vec ? 4 : 0) | (dx ? 2 : 0) | (m1 ? 1 : 0);
  switch (key) {
    case 7: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(true, true, 1); break;
    case 6: EXPO_L(true, true, 0); break;
    case 5: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(true, false, 1); break;
    case 4: EXPO_L(true, false, 0); break;
    case 3: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(false, true, 1); break;
    case 2: EXPO_L(false, true, 0); break;
    case 1: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(false, false, 1); break;
    default: EXPO_L(false, false, 0); break;
  }
#undef EXPO_L
#undef EXPO_LIO
  HIP_TRY(hipGetLastError(), "filter_bwd launch");
  return EXPO_OK;
}

template <typename T>
static int fwd_by_id(int id, const void* x, void* y, const float* p, int n, int h, int w, hipStream_t s,
                     int rev = 0, int n_geom = -1) {
  if (n_geom < 0) n_geom = n;
  switch (id) {
    case 0: return launch_fwd<ExposureF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 1: return launch_fwd<GammaF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 2: return launch_fwd<WhiteBalanceF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 3: return launch_fwd<SatPlusF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 4: return launch_fwd<ToneF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 5: return launch_fwd<ContrastF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 6: return launch_fwd<WnbF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 7: return launch_fwd<ColorF, T>(x, y, p, n, h, w, s, rev, n_geom);
    case 8: return launch_fwd<LevelF, T>(x, y, p, n, h, w, s, rev, n_geom);
  }
  return fail(EXPO_E_BADARG, "filter_id out of range");
}

template <typename T>
static int bwd_by_id(int id, const void* x, const void* dy, void* dx, const float* p, float* records, int n, int h,
                     int w, int mode, hipStream_t s, int rev = 0, int n_geom = -1) {
  if (n_geom < 0) n_geom = n;
  switch (id) {
    case 0: return launch_bwd<ExposureF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 1: return launch_bwd<GammaF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 2: return launch_bwd<WhiteBalanceF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 3: return launch_bwd<SatPlusF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 4: return launch_bwd<ToneF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 5: return launch_bwd<ContrastF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 6: return launch_bwd<WnbF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 7: return launch_bwd<ColorF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
    case 8: return launch_bwd<LevelF, T>(x, dy, dx, p, records, n, h, w, mode, s, rev, n_geom);
  }
  return fail(EXPO_E_BADARG, "filter_id out of range");
}

template <class F, typename T>
static int launch_apply_fwd(const void* x, void* y, const float* params, const float* mp, float sharp, float ms,
                            int n, int h, int w, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, y}, kGeomMap);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  // cache policy as for the plain kernels: tensors far beyond L2 stream (nt loads, write-through stores) -- 64x512x512
  // fp16, gpurun r03p40: E 36.6 -> 32.4 us, Ct 42.5 -> 38.9; the curve filters are bound by their arithmetic and lose
  // a microsecond (C 51.8 -> 52.9), so they keep the default policy
  if (g.stream && F::kLutFloats == 0)
    hipLaunchKernelGGL((apply_fwd_kernel<F, T, true, IoStream>), grid, block, 0, s, (const T*)x, (T*)y, params, mp, sharp, ms, h, w, g.groups);
  else if (g.vec)
    hipLaunchKernelGGL((apply_fwd_kernel<F, T, true>), grid, block, 0, s, (const T*)x, (T*)y, params, mp, sharp, ms, h, w, g.groups);
  else
    hipLaunchKernelGGL((apply_fwd_kernel<F, T, false>), grid, block, 0, s, (const T*)x, (T*)y, params, mp, sharp, ms, h, w, g.groups);
  HIP_TRY(hipGetLastError(), "apply_fwd launch");
  return EXPO_OK;
}

template <class F, typename T>
static int launch_apply_bwd(const void* x, const void* dy, void* dx, const float* params, const float* mp,
                            float* records, float sharp, float ms, int n, int h, int w, int mode, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, dy, dx}, kGeomApply);
  const dim3 grid(g.blocks_x, n), block(kThreads);
#define EXPO_LIO(VEC, HAS_DX, MODE, IO)                                                                    \
  hipLaunchKernelGGL((apply_bwd_kernel<F, T, VEC, HAS_DX, MODE, IO>), grid, block, 0, s, (const T*)x,      \
                     (const T*)dy, (T*)dx, params, mp, records, sharp, ms, h, w, g.groups)
#define EXPO_L(VEC, HAS_DX, MODE) EXPO_LIO(VEC, HAS_DX, MODE, IoCached)
  const bool m1 = std::is_same<F, SatPlusF>::value && mode == 1;
  // the streaming policy is instantiated for the full backward (dx wanted) only; it buys 1-2 % here (the masked backward
  // is bound by its arithmetic: E 60.1 -> 59.0 us, C 109 -> 107.3 at 64x512x512 fp16, gpurun r03p40)
  const int key = (g.stream && dx ? 8 : 0) | (g.vec ? 4 : 0) | (dx ? 2 : 0) | (m1 ? 1 : 0);
  switch (key) {
    case 15: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_LIO(true, true, 1, IoStream); break;
    case 14: EXPO_LIO(true, true, 0, IoStream); break;
    case 7: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(true, true, 1); break;
    case 6: EXPO_L(true, true, 0); break;
    case 5: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(true, false, 1); break;
    case 4: EXPO_L(true, false, 0); break;
    case 3: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(false, true, 1); break;
    case 2: EXPO_L(false, true, 0); break;
    case 1: if constexpr (std::is_same<F, SatPlusF>::value) EXPO_L(false, false, 1); break;
    default: EXPO_L(false, false, 0); break;
  }
#undef EXPO_L
#undef EXPO_LIO
  HIP_TRY(hipGetLastError(), "apply_bwd launch");
  return EXPO_OK;
}

template <typename T>
static int apply_fwd_by_id(int id, const void* x, void* y, const float* p, const float* mp, float sharp, float ms,
                           int n, int h, int w, hipStream_t s) {
  switch (id) {
    case 0: return launch_apply_fwd<ExposureF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 1: return launch_apply_fwd<GammaF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 2: return launch_apply_fwd<WhiteBalanceF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 3: return launch_apply_fwd<SatPlusF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 4: return launch_apply_fwd<ToneF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 5: return launch_apply_fwd<ContrastF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 6: return launch_apply_fwd<WnbF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 7: return launch_apply_fwd<ColorF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
    case 8: return launch_apply_fwd<LevelF, T>(x, y, p, mp, sharp, ms, n, h, w, s);
  }
  return fail(EXPO_E_BADARG, "filter_id out of range");
}

template <typename T>
static int apply_bwd_by_id(int id, const void* x, const void* dy, void* dx, const float* p, const float* mp,
                           float* records, float sharp, float ms, int n, int h, int w, int mode, hipStream_t s) {
  switch (id) {
    case 0: return launch_apply_bwd<ExposureF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 1: return launch_apply_bwd<GammaF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 2: return launch_apply_bwd<WhiteBalanceF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 3: return launch_apply_bwd<SatPlusF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 4: return launch_apply_bwd<ToneF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 5: return launch_apply_bwd<ContrastF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 6: return launch_apply_bwd<WnbF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 7: return launch_apply_bwd<ColorF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
    case 8: return launch_apply_bwd<LevelF, T>(x, dy, dx, p, mp, records, sharp, ms, n, h, w, mode, s);
  }
  return fail(EXPO_E_BADARG, "filter_id out of range");
}

template <typename T>
static int apply_dispatch_fwd_t(const int32_t* ids, const void* x, void* y, const float* params, const float* mp,
                                float sharp, float ms, int n, int h, int w, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, y}, kGeomMap);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  if (g.vec)
    hipLaunchKernelGGL((apply_dispatch_fwd_kernel<T, true>), grid, block, 0, s, ids, (const T*)x, (T*)y, params, mp, sharp, ms, h, w, g.groups);
  else
    hipLaunchKernelGGL((apply_dispatch_fwd_kernel<T, false>), grid, block, 0, s, ids, (const T*)x, (T*)y, params, mp, sharp, ms, h, w, g.groups);
  HIP_TRY(hipGetLastError(), "apply_dispatch_fwd launch");
  return EXPO_OK;
}

template <typename T>
static int apply_dispatch_bwd_t(const int32_t* ids, const void* x, const void* dy, void* dx, const float* params,
                                float* dparams, const float* mp, float* dmp, float sharp, float ms, int n, int h, int w,
                                int mode, void* workspace, size_t workspace_bytes, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, dy, dx}, kGeomApply);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, g.blocks_x, 1, &records)) return rc;
#define EXPO_L(VEC, HAS_DX, MODE)                                                                                   \
  hipLaunchKernelGGL((apply_dispatch_bwd_kernel<T, VEC, HAS_DX, MODE>), grid, block, 0, s, ids, (const T*)x,         \
                     (const T*)dy, (T*)dx, params, mp, records, sharp, ms, h, w, g.groups)
  const int key = (g.vec ? 4 : 0) | (dx ? 2 : 0) | (mode == 1 ? 1 : 0);
  switch (key) {
    case 7: EXPO_L(true, true, 1); break;
    case 6: EXPO_L(true, true, 0); break;
    case 5: EXPO_L(true, false, 1); break;
    case 4: EXPO_L(true, false, 0); break;
    case 3: EXPO_L(false, true, 1); break;
    case 2: EXPO_L(false, true, 0); break;
    case 1: EXPO_L(false, false, 1); break;
    default: EXPO_L(false, false, 0); break;
  }
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "apply_dispatch_bwd launch");
  FinishArgs fa{};
  fa.s[0] = FinishStep{params, dparams, dmp, records, ids, kFinApplyDispatch, 0, 0, 0.f, g.blocks_x};
  return launch_finish(fa, 1, n, s);
}

template <typename T>
static int dispatch_fwd_t(const int32_t* ids, const void* x, void* y, const float* params, float* penalty, int n,
                          int h, int w, void* workspace, size_t workspace_bytes, hipStream_t s) {
  // the geometry of the element-wise backward kernels: one group per thread below the Infinity Cache size (with the
  // fused penalty 38.5-39.9 -> 37.0 us at 64x512x512 against four groups), one record per block for the penalty
  const Geom g = make_geom<T>(n, h, w, {x, y}, kGeomReduce);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  float* records = nullptr;
  if (penalty) {
    if (int rc = ws_check(workspace, workspace_bytes, n, g.blocks_x, 1, &records)) return rc;
  }
  // forward: one launch handles every filter (the heaviest forward body needs ~106 VGPRs, fine for a
  // streaming kernel); only the backward is split into light / curve launches (124 vs 190 VGPRs)
#define EXPO_L(VEC, PEN, IO)                                                                              \
  hipLaunchKernelGGL((dispatch_fwd_kernel<T, VEC, PEN, kSetAll, IO>), grid, block, 0, s, ids, (const T*)x, \
                     (T*)y, params, records, g.hw, g.groups)
  if (g.stream) {
    if (penalty) EXPO_L(true, true, IoStream); else EXPO_L(true, false, IoStream);
  } else if (g.vec) {
    if (penalty) EXPO_L(true, true, IoCached); else EXPO_L(true, false, IoCached);
  } else {
    if (penalty) EXPO_L(false, true, IoCached); else EXPO_L(false, false, IoCached);
  }
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "dispatch_fwd launch");
  if (penalty) {
    FinishArgs fa{};
    fa.s[0] = FinishStep{nullptr, penalty, nullptr, records, ids, kFinPenalty, 0, 0, 1.0f / (float(g.hw) * 3.0f), g.blocks_x};
    return launch_finish(fa, 1, n, s);
  }
  return EXPO_OK;
}

template <typename T>
static int dispatch_bwd_t(const int32_t* ids, const void* x, const void* dy, void* dx, const float* params,
                          float* dparams, const float* dpenalty, int n, int h, int w, int mode, void* workspace,
                          size_t workspace_bytes, hipStream_t s) {
  // own geometry knob (EXPO_DISPATCH_GROUPS_PER_THREAD); 1 / 2 / 4 groups per thread measured 80 / 72 / 71 us for
  // the launch pair + finish at 64x512x512 with ids cycling over the 8 filters (gpurun r02p10)
  const Geom g = make_geom<T>(n, h, w, {x, dy, dx}, kGeomDispatch);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  const float inv_count = 1.0f / (float(g.hw) * 3.0f);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, g.blocks_x, 1, &records)) return rc;
  // two launches (light filters / curve filters): an image's records are written by exactly one of them
  // (none for id -1); ONE finish launch then writes every dparams row -- no zero-fill
#define EXPO_L2(VEC, HAS_DX, PEN, MODE, SET, IO)                                                          \
  hipLaunchKernelGGL((dispatch_bwd_kernel<T, VEC, HAS_DX, PEN, MODE, SET, IO>), grid, block, 0, s, ids,   \
                     (const T*)x, (const T*)dy, (T*)dx, params, records, dpenalty, g.hw, g.groups, inv_count)
#define EXPO_L(VEC, HAS_DX, PEN, IO)                                                                      \
  do {                                                                                                    \
    if (mode == 1) EXPO_L2(VEC, HAS_DX, PEN, 1, kSetLight, IO);                                           \
    else EXPO_L2(VEC, HAS_DX, PEN, 0, kSetLight, IO);                                                     \
    EXPO_L2(VEC, HAS_DX, PEN, 0, kSetCurves, IO);                                                         \
  } while (0)
  // (ONE launch for every filter measured the same as this pair -- 66.5-68.5 vs 68.7-68.9 us incl. the finish,
  // gpurun r02p16 -- and would double the instantiations.)
  // the streaming policy is instantiated for the full backward (dx wanted) only
  const int key = (g.stream && dx ? 8 : 0) | (g.vec ? 4 : 0) | (dx ? 2 : 0) | (dpenalty ? 1 : 0);
  switch (key) {
    case 15: EXPO_L(true, true, true, IoStream); break;
    case 14: EXPO_L(true, true, false, IoStream); break;
    case 7: EXPO_L(true, true, true, IoCached); break;
    case 6: EXPO_L(true, true, false, IoCached); break;
    case 5: EXPO_L(true, false, true, IoCached); break;
    case 4: EXPO_L(true, false, false, IoCached); break;
    case 3: EXPO_L(false, true, true, IoCached); break;
    case 2: EXPO_L(false, true, false, IoCached); break;
    case 1: EXPO_L(false, false, true, IoCached); break;
    default: EXPO_L(false, false, false, IoCached); break;
  }
#undef EXPO_L
#undef EXPO_L2
  HIP_TRY(hipGetLastError(), "dispatch_bwd launch");
  FinishArgs fa{};
  fa.s[0] = FinishStep{params, dparams, nullptr, records, ids, kFinDispatch, 0, 0, 0.f, g.blocks_x};
  return launch_finish(fa, 1, n, s);
}

template <typename T>
static int stats_t(const void* x, float* stats, int n, int h, int w, void* workspace, size_t workspace_bytes,
                   hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x}, kGeomReadReduce);  // read-only: the forward geometry streams best
  const dim3 grid(g.blocks_x, n), block(kThreads);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, g.blocks_x, 1, &records)) return rc;
  if (g.stream) hipLaunchKernelGGL((stats_kernel<T, true, IoStream>), grid, block, 0, s, (const T*)x, records, g.hw, g.groups);
  else if (g.vec) hipLaunchKernelGGL((stats_kernel<T, true, IoCached>), grid, block, 0, s, (const T*)x, records, g.hw, g.groups);
  else hipLaunchKernelGGL((stats_kernel<T, false, IoCached>), grid, block, 0, s, (const T*)x, records, g.hw, g.groups);
  HIP_TRY(hipGetLastError(), "stats launch");
  FinishArgs fa{};
  fa.s[0] = FinishStep{nullptr, stats, nullptr, records, nullptr, kFinStats, 0, 0, 1.0f / float(g.hw), g.blocks_x};
  return launch_finish(fa, 1, n, s);
}

template <typename T>
static int penalty_t(const void* y, float* pen, int n, int h, int w, void* workspace, size_t workspace_bytes,
                     hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {y}, kGeomReadReduce);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, g.blocks_x, 1, &records)) return rc;
  if (g.stream) hipLaunchKernelGGL((penalty_kernel<T, true, IoStream>), grid, block, 0, s, (const T*)y, records, g.hw, g.groups);
  else if (g.vec) hipLaunchKernelGGL((penalty_kernel<T, true, IoCached>), grid, block, 0, s, (const T*)y, records, g.hw, g.groups);
  else hipLaunchKernelGGL((penalty_kernel<T, false, IoCached>), grid, block, 0, s, (const T*)y, records, g.hw, g.groups);
  HIP_TRY(hipGetLastError(), "penalty launch");
  FinishArgs fa{};
  fa.s[0] = FinishStep{nullptr, pen, nullptr, records, nullptr, kFinPenalty, 0, 0, 1.0f / (float(g.hw) * 3.0f), g.blocks_x};
  return launch_finish(fa, 1, n, s);
}

template <typename T>
static int vignet_fwd_t(const void* x, void* y, const float* mp, float sharp, int masking, int n, int h, int w,
                        hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, y}, kGeomMap);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  if (g.stream) hipLaunchKernelGGL((vignet_fwd_kernel<T, true, IoStream>), grid, block, 0, s, (const T*)x, (T*)y, mp, sharp, masking, h, w, g.groups);
  else if (g.vec) hipLaunchKernelGGL((vignet_fwd_kernel<T, true>), grid, block, 0, s, (const T*)x, (T*)y, mp, sharp, masking, h, w, g.groups);
  else hipLaunchKernelGGL((vignet_fwd_kernel<T, false>), grid, block, 0, s, (const T*)x, (T*)y, mp, sharp, masking, h, w, g.groups);
  HIP_TRY(hipGetLastError(), "vignet_fwd launch");
  return EXPO_OK;
}

template <typename T>
static int vignet_bwd_t(const void* x, const void* dy, void* dx, const float* mp, float* dmp, float sharp, int masking,
                        int n, int h, int w, void* workspace, size_t workspace_bytes, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, dy, dx}, kGeomApply);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, g.blocks_x, 1, &records)) return rc;
#define EXPO_L(VEC, HAS_DX, IO) \
  hipLaunchKernelGGL((vignet_bwd_kernel<T, VEC, HAS_DX, IO>), grid, block, 0, s, (const T*)x, (const T*)dy, (T*)dx, mp, records, sharp, masking, h, w, g.groups)
  // (the streaming policy pays in the forward, 36.3 -> 33.3 us, not here: 53.2 -> 55.5 us, gpurun r03p40)
  if (g.vec) { if (dx) EXPO_L(true, true, IoCached); else EXPO_L(true, false, IoCached); }
  else { if (dx) EXPO_L(false, true, IoCached); else EXPO_L(false, false, IoCached); }
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "vignet_bwd launch");
  FinishArgs fa{};
  // `accumulate` carries the masking flag: with masking off the mask is the constant 1 and its parameters get 0
  fa.s[0] = FinishStep{mp, dmp, nullptr, records, nullptr, kFinVignet, 0, masking ? 1 : 0, sharp, g.blocks_x};
  return launch_finish(fa, 1, n, s);
}

template <typename T>
static int penalty_bwd_t(const void* y, const float* dpen, void* dy, int n, int h, int w, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {y, dy}, kGeomMap);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  const float inv_count = 1.0f / (float(g.hw) * 3.0f);
#define EXPO_L(VEC, IO) \
  hipLaunchKernelGGL((penalty_bwd_kernel<T, VEC, IO>), grid, block, 0, s, (const T*)y, dpen, (T*)dy, g.hw, g.groups, inv_count)
  if (g.stream) EXPO_L(true, IoStream); else if (g.vec) EXPO_L(true, IoCached); else EXPO_L(false, IoCached);
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "penalty_bwd launch");
  return EXPO_OK;
}

template <typename T>
static int stats_bwd_t(const void* x, const float* stats, const float* dstats, void* dx, int n, int h, int w,
                       hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, dx}, kGeomMap);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  const float inv_hw = 1.0f / float(g.hw);
#define EXPO_L(VEC, IO) \
  hipLaunchKernelGGL((stats_bwd_kernel<T, VEC, IO>), grid, block, 0, s, (const T*)x, stats, dstats, (T*)dx, g.hw, g.groups, inv_hw)
  if (g.stream) EXPO_L(true, IoStream); else if (g.vec) EXPO_L(true, IoCached); else EXPO_L(false, IoCached);
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "stats_bwd launch");
  return EXPO_OK;
}

template <typename T>
static int stats_jvp_t(const void* x, const float* stats, const void* v, float* jv, int n, int h, int w,
                       void* workspace, size_t workspace_bytes, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, v}, kGeomReadReduce);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, g.blocks_x, 1, &records)) return rc;
#define EXPO_L(VEC, IO) \
  hipLaunchKernelGGL((stats_jvp_kernel<T, VEC, IO>), grid, block, 0, s, (const T*)x, stats, (const T*)v, records, g.hw, g.groups)
  if (g.stream) EXPO_L(true, IoStream); else if (g.vec) EXPO_L(true, IoCached); else EXPO_L(false, IoCached);
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "stats_jvp launch");
  FinishArgs fa{};
  fa.s[0] = FinishStep{nullptr, jv, nullptr, records, nullptr, kFinScaled, 3, 0, 1.0f / float(g.hw), g.blocks_x};
  return launch_finish(fa, 1, n, s);
}

template <typename T>
static int stats_hvp_t(const void* x, const float* dstats, const float* jv, const void* v, void* out, int n, int h,
                       int w, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, v, out}, kGeomMap);
  const dim3 grid(g.blocks_x, n), block(kThreads);
  const float inv_hw = 1.0f / float(g.hw);
#define EXPO_L(VEC, IO) \
  hipLaunchKernelGGL((stats_hvp_kernel<T, VEC, IO>), grid, block, 0, s, (const T*)x, dstats, jv, (const T*)v, (T*)out, g.hw, g.groups, inv_hw)
  if (g.stream) EXPO_L(true, IoStream); else if (g.vec) EXPO_L(true, IoCached); else EXPO_L(false, IoCached);
#undef EXPO_L
  HIP_TRY(hipGetLastError(), "stats_hvp launch");
  return EXPO_OK;
}

}  // namespace expo

// ======================================================================== C-ABI
using namespace expo;

// ---- two half-batches on two streams inside a chain --------------------------------------------------------------
// The 17 launches of a chain step depend on each other, so every kernel's ramp-up and tail (and the 1.7-1.9 us
// boundary between two streaming kernels) is exposed.  Images are independent: the batch is split in two halves,
// each half's launches go to its own stream (the caller's and a library-owned helper stream, forked and joined with
// events inside the call -- the C-ABI contract "ordered only through `stream`" is unchanged, and the pattern is
// capturable into a hipGraph as two parallel branches), and one half's kernels fill the other's bubbles.
// Measured (gpurun r02p20, Python-level prototype): 64 images 0.613-0.622 -> 0.591-0.595 ms per chain step, 128
// images 1.19-1.22 -> 1.18; a LOSS at 16 / 32 images (half a batch no longer fills 256 CUs) and beyond the Infinity
// Cache (256 images: the two halves evict each other).  In the library itself (gpurun r03p15, ms per chain step off / on):
// 16 images 0.184 / 0.185, 32 images 0.322 / 0.305, 64 images 0.600 / 0.579, 128 images 1.180 / 1.163, 64x64x64 0.046 / 0.09-0.14.
// Hence the gate: on by default only when one tensor is in [40 MiB, 256 MiB); EXPO_CHAIN_STREAMS=1 / 2 forces it
// off / on; expo_chain_streams() answers what a shape gets.
struct ForkJoin {
  hipStream_t helper = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
// One helper stream + event pair PER (device, caller stream): two host threads that drive two user streams -- one of
// them possibly inside a hipGraph capture, which the helper then joins -- no longer meet on one helper and one event
// pair (advisor, round 3).  Entries are created on first use and live for the process (a handful of streams).
//
// A helper must not share a HARDWARE queue with its caller (round 4, r04p14): the runtime multiplexes all streams of a
// process over a few hardware queues (4 by default) in creation order, and two streams on one queue run their kernels
// strictly one after the other -- the two lanes then cost more than one (256x512x512 chain calls: 2.63 ms aliased,
// 2.47 one lane, 2.31 on separate queues).  Which queue a stream got cannot be asked, so the first EAGER two-lane call
// of a caller stream probes its helper: a one-wave kernel on the caller waits (at most 0.5 ms) for a flag that a
// one-wave kernel on the helper sets.  It sees the flag iff kernels of the two streams can run side by side.  A helper
// that fails goes to the device's spare list (kept: releasing it would hand the same queue to the next stream created)
// and the next candidate is tried.  Under stream capture nothing can be probed and nothing needs to be: the graph's
// branches get their queues from the graph executor.  EXPO_CHAIN_HELPER_PROBE=0 takes the first helper unprobed.
__device__ int g_lane_probe[2];  // [0] the flag (a generation number), [1] what the waiting kernel saw
__global__ void lane_probe_wait_kernel(int gen, long long limit_ticks) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();  // 100 MHz
  int seen = 0;
  do {
    seen = __hip_atomic_load(&g_lane_probe[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen == gen) break;
    __builtin_amdgcn_s_sleep(64);
  } while (wall_clock64() - t0 < limit_ticks);
  __hip_atomic_store(&g_lane_probe[1], seen == gen ? gen : -gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void lane_probe_set_kernel(int gen) {
  if (threadIdx.x == 0) __hip_atomic_store(&g_lane_probe[0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 1: kernels of `s` and `h` ran side by side; 0: they did not; -1: could not tell (treated as "take it")
static int lanes_run_side_by_side(hipStream_t s, hipStream_t h) {
  static int gen = 0;
  ++gen;
  hipLaunchKernelGGL(lane_probe_wait_kernel, dim3(1), dim3(64), 0, s, gen, 50000LL);
  hipLaunchKernelGGL(lane_probe_set_kernel, dim3(1), dim3(64), 0, h, gen);
  if (hipGetLastError() != hipSuccess) return -1;
  if (hipStreamSynchronize(h) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
  int got[2] = {0, 0};
  if (hipMemcpyFromSymbol(got, HIP_SYMBOL(g_lane_probe), sizeof(got)) != hipSuccess) return -1;
  return got[1] == gen ? 1 : got[1] == -gen ? 0 : -1;
}
static int g_helper_probes = 0, g_helper_rejected = 0;  // expo_chain_helper_stats()

// Table of (device, caller stream) -> helper.  A stream HANDLE can come back for a new stream after
// hipStreamDestroy, so an entry also remembers the runtime's stream id (hipStreamGetId): a recreated stream gets a
// fresh entry -- and is probed again -- while the stale entry's helper is recycled (advisor, round 4).
// Locking: `mu` guards the table and the spare list only; the probe (which launches a kernel that may wait up to
// 0.5 ms on the caller's stream and synchronises both streams) runs under the ENTRY's mutex, so callers on other
// streams are never held up by it.  At most kMaxProbeAttempts candidates are tried per pairing and at most
// kMaxSpares rejected streams are parked per process (beyond that a rejected stream is destroyed: its queue slot may
// then be handed out again, which only costs later probes an attempt).
constexpr int kMaxProbeAttempts = 4;
constexpr size_t kMaxSpares = 8;
struct HelperEntry {
  int dev;
  hipStream_t caller;
  unsigned long long caller_id;
  ForkJoin fj;
  bool probed;
  std::mutex mu;
};
static std::vector<HelperEntry*> g_helper_table;
static std::vector<hipStream_t> g_helper_spares;  // rejected helpers, never used again, kept so that their queue slot stays taken
static std::mutex g_helper_mu;
static bool helper_probe_enabled() {
  static const bool probe = !(getenv("EXPO_CHAIN_HELPER_PROBE") && atoi(getenv("EXPO_CHAIN_HELPER_PROBE")) == 0);
  return probe;
}
// hipStreamGetId entered the runtime with HIP 7.1; the process may run on an older libamdhip64 (PyTorch ships its own
// 7.0), so the symbol is looked up at run time -- without it a reused handle cannot be told from the old stream and
// keeps the old pairing (expo_chain_release is the way to drop it then).
static unsigned long long stream_id_of(hipStream_t s) {
  using GetId = hipError_t (*)(hipStream_t, unsigned long long*);
  static const GetId get_id = reinterpret_cast<GetId>(dlsym(RTLD_DEFAULT, "hipStreamGetId"));
  unsigned long long id = 0;
  if (!get_id) return 0;
  if (get_id(s, &id) != hipSuccess) {
    (void)hipGetLastError();
    id = 0;
  }
  return id;
}
static void park_or_destroy(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_helper_mu);
  if (g_helper_spares.size() < kMaxSpares)
    g_helper_spares.push_back(st);
  else
    (void)hipStreamDestroy(st);
}
// the probe of one pairing; the caller holds e->mu and `e` is eager (not capturing)
static void probe_helper(HelperEntry* e) {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;  // the probe synchronises: keep another thread's capture valid
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  hipStream_t first = e->fj.helper, cand = first;
  bool found = false;
  for (int attempt = 0; attempt < kMaxProbeAttempts; ++attempt) {
    {
      std::lock_guard<std::mutex> lock(g_helper_mu);
      ++g_helper_probes;
    }
    const int r = lanes_run_side_by_side(e->caller, cand);
    if (r != 0) {
      found = true;
      break;
    }
    {
      std::lock_guard<std::mutex> lock(g_helper_mu);
      ++g_helper_rejected;
    }
    if (cand != first) park_or_destroy(cand);
    cand = nullptr;
    if (attempt + 1 == kMaxProbeAttempts) break;
    if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      cand = nullptr;
      break;
    }
  }
  if (found && cand != first) {
    park_or_destroy(first);
    e->fj.helper = cand;
  }  // (!found: nothing ran side by side -- a one-queue configuration -- and the first helper stays)
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  e->probed = true;
}

// -> the (device, caller) entry, created on first use; `probe_now`: probe it if that has not happened and the caller
// is not capturing.  nullptr on a HIP error.
static HelperEntry* helper_entry(hipStream_t caller, bool probe_now) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
  const unsigned long long id = stream_id_of(caller);
  HelperEntry* e = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_helper_mu);
    for (HelperEntry* t : g_helper_table)
      if (t->dev == dev && t->caller == caller) e = t;
    if (e && e->caller_id != id) {
      // the handle names a NEW stream: the pairing has to be probed again (the helper itself is reusable -- it is idle,
      // every chain call joins it before returning)
      e->caller_id = id;
      e->probed = !helper_probe_enabled();
    }
    if (!e) {
      hipStream_t st;
      hipEvent_t e0, e1;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) return nullptr;
      e = new HelperEntry{dev, caller, id, ForkJoin{st, e0, e1}, !helper_probe_enabled(), {}};
      g_helper_table.push_back(e);
    }
  }
  if (!probe_now) return e;
  std::lock_guard<std::mutex> lock(e->mu);
  if (e->probed) return e;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(caller, &cap) != hipSuccess) {
    (void)hipGetLastError();
    return e;  // cannot tell: no probe now
  }
  if (cap != hipStreamCaptureStatusNone) return e;  // capturing: the helper only names a branch of the graph
  probe_helper(e);  // eager and unprobed; the helper is idle: it was created above or has only been part of captures
  return e;
}
static ForkJoin* fork_join_for_device(hipStream_t caller = nullptr) {
  HelperEntry* e = helper_entry(caller, true);
  return e ? &e->fj : nullptr;
}
// fork: helper waits for everything `s` holds; join: `s` waits for everything the helper holds.  The event is shared by
// all callers of a device, so record + wait must not interleave with another host thread's pair (its record would
// replace the state this thread's wait is meant to see).
static std::mutex g_fork_join_mu;
static int chain_fork(ForkJoin* fj, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_fork_join_mu);
  HIP_TRY(hipEventRecord(fj->fork, s), "chain fork");
  HIP_TRY(hipStreamWaitEvent(fj->helper, fj->fork, 0), "chain fork");
  return EXPO_OK;
}
static int chain_join(ForkJoin* fj, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_fork_join_mu);
  HIP_TRY(hipEventRecord(fj->join, fj->helper), "chain join");
  HIP_TRY(hipStreamWaitEvent(s, fj->join, 0), "chain join");
  return EXPO_OK;
}
static bool chain_split(int n, int h, int w, int dtype) {
  static const int forced = getenv("EXPO_CHAIN_STREAMS") ? atoi(getenv("EXPO_CHAIN_STREAMS")) : 0;
  if (n < 2) return false;
  if (forced == 1) return false;
  if (forced >= 2) return true;
  const long bytes = long(n) * h * w * 3L * (dtype == EXPO_F16 ? 2L : 4L);
  return bytes >= (40L << 20) && bytes < (256L << 20);
}

// Consecutive launches of a chain walk the images in alternating directions when ONE tensor is larger than
// the 256 MiB Infinity Cache: each launch then starts on the images its predecessor touched last, which are
// the ones still cached.  Measured on MI355X, 8-step chain fwd+bwd fp16 (gpurun r02p8, r02p11):
//   256x512x512 (384 MiB per tensor)  2.748 -> 2.534 ms  (5.86 -> 6.36 TB/s of algorithmic traffic)
//   64 / 128 x512x512 (96 / 192 MiB)  0.615 / 1.206 ms either way
//   64x64x64 (1.5 MiB, L2-resident)   0.0470 -> 0.0578 ms: block b runs on XCD b % 8, so with the SAME order
//                                     a consumer block finds its producer's lines in its own XCD's L2 --
//                                     reversing the order sends it to another XCD.  Hence the size gate.
// EXPO_CHAIN_SNAKE=0 / 1 forces it off / on.
static bool chain_snake(int n, int h, int w, int dtype) {
  static const int forced = getenv("EXPO_CHAIN_SNAKE") ? atoi(getenv("EXPO_CHAIN_SNAKE")) : -1;
  if (forced >= 0) return forced != 0;
  const long bytes = long(n) * h * w * 3L * (dtype == EXPO_F16 ? 2L : 4L);
  return bytes >= (256L << 20);
}

// Image tiles inside a chain when one tensor is larger than the Infinity Cache (round 4).  Images are independent, so
// a chain over n images may run TILE-MAJOR: all steps on the first tile, then all steps on the next.  A tile whose
// tensors fit the 256 MiB cache beside their neighbours (default 96 MiB per tensor = 64 images of 512x512 fp16) is
// written by step i and read by step i+1 while it is still cached -- the regime the 64-image metric shape runs in --
// instead of every launch streaming a 384 MiB tensor that its consumer finds evicted (the alternating image walk above
// saved the cached tail only).  Within the plan a tile is split over the two streams like a whole batch of its size.
// EXPO_CHAIN_TILE_MIB=<MiB per tensor and tile> (0: off, the alternating walk is used instead).
struct ChainChunk { int nb, np, lane; };  // images [nb, nb + np) on the caller's stream (lane 0) or the helper (lane 1)
struct ChainPlan {
  std::vector<ChainChunk> chunks;  // in launch order per lane
  bool two_lanes = false, snake = false;
};
static ChainPlan chain_plan(int n, int h, int w, int dtype) {
  static const int tile_mib = getenv("EXPO_CHAIN_TILE_MIB") ? atoi(getenv("EXPO_CHAIN_TILE_MIB")) : 96;
  ChainPlan plan;
  const long image_bytes = long(h) * w * 3L * (dtype == EXPO_F16 ? 2L : 4L);
  const long bytes = long(n) * image_bytes;
  int tiles = 1;
  static const long tile_min = long(getenv("EXPO_CHAIN_TILE_MIN_MIB") ? atoi(getenv("EXPO_CHAIN_TILE_MIN_MIB")) : 256) << 20;
  if (tile_mib > 0 && bytes >= tile_min && n > 1) {
    long tile_n = (long(tile_mib) << 20) / image_bytes;
    if (tile_n < 1) tile_n = 1;
    tiles = int((n + tile_n - 1) / tile_n);
  }
  if (tiles == 1) {
    plan.snake = chain_snake(n, h, w, dtype);
    plan.two_lanes = chain_split(n, h, w, dtype);
    const int n0 = plan.two_lanes ? n / 2 : n;
    plan.chunks.push_back({0, n0, 0});
    if (plan.two_lanes) plan.chunks.push_back({n0, n - n0, 1});
    return plan;
  }
  const int base = n / tiles, rem = n % tiles;  // balanced tiles
  plan.two_lanes = chain_split(base, h, w, dtype);
  int at = 0;
  for (int t = 0; t < tiles; ++t) {
    const int tn = base + (t < rem ? 1 : 0);
    const int n0 = (plan.two_lanes && tn >= 2) ? tn / 2 : tn;
    plan.chunks.push_back({at, n0, 0});
    if (n0 < tn) plan.chunks.push_back({at + n0, tn - n0, 1});
    at += tn;
  }
  return plan;
}

extern "C" {

int expo_version(void) { return EXPO_ABI_VERSION; }

const char* expo_last_error(void) { return g_err.c_str(); }

#ifndef EXPO_SOURCE_DIGEST
#define EXPO_SOURCE_DIGEST "unknown"
#endif
const char* expo_build_info(void) { return EXPO_SOURCE_DIGEST; }

int expo_num_filter_params(int filter_id) {
  if (filter_id < 0 || filter_id >= EXPO_NUM_FILTERS) return -1;
  return kNumParams[filter_id];
}

size_t expo_workspace_bytes(int n, int h, int w, int dtype) {
  if (check_common(n, h, w, dtype) != EXPO_OK || n == 0) return 0;
  const int bx = geom_bx_max(n, h, w, dtype);
  return ws_step_bytes(n, bx);
}

int expo_filter_fwd(int filter_id, const void* x, void* y, const float* params, int n, int h, int w, int dtype,
                    void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (filter_id < 0 || filter_id >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
  if (n == 0) return EXPO_OK;
  if (!x || !y || !params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? fwd_by_id<half_t>(filter_id, x, y, params, n, h, w, s)
                           : fwd_by_id<float>(filter_id, x, y, params, n, h, w, s);
}

static int filter_bwd_common(int filter_id, const void* x, const void* dy, void* dx, const float* params,
                             float* dparams, int n, int h, int w, int dtype, int hsv_grad_mode, void* workspace,
                             size_t workspace_bytes, void* stream, bool accum) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (filter_id < 0 || filter_id >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
  if (hsv_grad_mode != 0 && hsv_grad_mode != 1) return fail(EXPO_E_BADARG, "hsv_grad_mode must be 0 or 1");
  if (n == 0) return EXPO_OK;
  if (!x || !dy || !params || !dparams) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int bx = geom_bx(bwd_geom_kind(filter_id), n, h, w, dtype);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, bx, 1, &records)) return rc;
  const int rc = dtype == EXPO_F16
                     ? bwd_by_id<half_t>(filter_id, x, dy, dx, params, records, n, h, w, hsv_grad_mode, s)
                     : bwd_by_id<float>(filter_id, x, dy, dx, params, records, n, h, w, hsv_grad_mode, s);
  if (rc) return rc;
  FinishArgs fa{};
  fa.s[0] = FinishStep{params, dparams, nullptr, records, nullptr, kFinFilter, filter_id, accum ? 1 : 0, 0.f, bx};
  return launch_finish(fa, 1, n, s);
}

int expo_filter_bwd(int filter_id, const void* x, const void* dy, void* dx, const float* params, float* dparams,
                    int n, int h, int w, int dtype, int hsv_grad_mode, void* workspace, size_t workspace_bytes,
                    void* stream) {
  return filter_bwd_common(filter_id, x, dy, dx, params, dparams, n, h, w, dtype, hsv_grad_mode, workspace,
                           workspace_bytes, stream, false);
}

int expo_filter_bwd_accumulate(int filter_id, const void* x, const void* dy, void* dx, const float* params,
                               float* dparams, int n, int h, int w, int dtype, int hsv_grad_mode, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return filter_bwd_common(filter_id, x, dy, dx, params, dparams, n, h, w, dtype, hsv_grad_mode, workspace,
                           workspace_bytes, stream, true);
}

int expo_filter_apply_fwd(int filter_id, const void* x, void* y, const float* params, const float* mask_params,
                          float maximum_sharpness, float minimum_strength, int n, int h, int w, int dtype,
                          void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (filter_id < 0 || filter_id >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
  if (n == 0) return EXPO_OK;
  if (!x || !y || !params || !mask_params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? apply_fwd_by_id<half_t>(filter_id, x, y, params, mask_params, maximum_sharpness,
                                                     minimum_strength, n, h, w, s)
                           : apply_fwd_by_id<float>(filter_id, x, y, params, mask_params, maximum_sharpness,
                                                    minimum_strength, n, h, w, s);
}

int expo_filter_apply_bwd(int filter_id, const void* x, const void* dy, void* dx, const float* params,
                          float* dparams, const float* mask_params, float* dmask_params, float maximum_sharpness,
                          float minimum_strength, int n, int h, int w, int dtype, int hsv_grad_mode, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (filter_id < 0 || filter_id >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
  if (hsv_grad_mode != 0 && hsv_grad_mode != 1) return fail(EXPO_E_BADARG, "hsv_grad_mode must be 0 or 1");
  if (n == 0) return EXPO_OK;
  if (!x || !dy || !params || !dparams || !mask_params || !dmask_params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int bx = geom_bx(kGeomApply, n, h, w, dtype);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, bx, 1, &records)) return rc;
  const int rc = dtype == EXPO_F16
                     ? apply_bwd_by_id<half_t>(filter_id, x, dy, dx, params, mask_params, records, maximum_sharpness,
                                               minimum_strength, n, h, w, hsv_grad_mode, s)
                     : apply_bwd_by_id<float>(filter_id, x, dy, dx, params, mask_params, records, maximum_sharpness,
                                              minimum_strength, n, h, w, hsv_grad_mode, s);
  if (rc) return rc;
  FinishArgs fa{};
  fa.s[0] = FinishStep{params, dparams, dmask_params, records, nullptr, kFinApply, filter_id, 0, 0.f, bx};
  return launch_finish(fa, 1, n, s);
}

int expo_filter_apply_dispatch_fwd(const int32_t* filter_ids, const void* x, void* y, const float* params,
                                   const float* mask_params, float maximum_sharpness, float minimum_strength, int n,
                                   int h, int w, int dtype, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!filter_ids || !x || !y || !params || !mask_params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? apply_dispatch_fwd_t<half_t>(filter_ids, x, y, params, mask_params, maximum_sharpness,
                                                          minimum_strength, n, h, w, s)
                           : apply_dispatch_fwd_t<float>(filter_ids, x, y, params, mask_params, maximum_sharpness,
                                                         minimum_strength, n, h, w, s);
}

int expo_filter_apply_dispatch_bwd(const int32_t* filter_ids, const void* x, const void* dy, void* dx,
                                   const float* params, float* dparams, const float* mask_params, float* dmask_params,
                                   float maximum_sharpness, float minimum_strength, int n, int h, int w, int dtype,
                                   int hsv_grad_mode, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (hsv_grad_mode != 0 && hsv_grad_mode != 1) return fail(EXPO_E_BADARG, "hsv_grad_mode must be 0 or 1");
  if (n == 0) return EXPO_OK;
  if (!filter_ids || !x || !dy || !params || !dparams || !mask_params || !dmask_params)
    return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16
             ? apply_dispatch_bwd_t<half_t>(filter_ids, x, dy, dx, params, dparams, mask_params, dmask_params,
                                            maximum_sharpness, minimum_strength, n, h, w, hsv_grad_mode, workspace,
                                            workspace_bytes, s)
             : apply_dispatch_bwd_t<float>(filter_ids, x, dy, dx, params, dparams, mask_params, dmask_params,
                                           maximum_sharpness, minimum_strength, n, h, w, hsv_grad_mode, workspace,
                                           workspace_bytes, s);
}

int expo_filter_dispatch_fwd(const int32_t* filter_ids, const void* x, void* y, const float* params,
                             float* penalty, int n, int h, int w, int dtype, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!filter_ids || !x || !y || !params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16
             ? dispatch_fwd_t<half_t>(filter_ids, x, y, params, penalty, n, h, w, workspace, workspace_bytes, s)
             : dispatch_fwd_t<float>(filter_ids, x, y, params, penalty, n, h, w, workspace, workspace_bytes, s);
}

int expo_filter_dispatch_bwd(const int32_t* filter_ids, const void* x, const void* dy, void* dx,
                             const float* params, float* dparams, const float* dpenalty, int n, int h, int w,
                             int dtype, int hsv_grad_mode, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (hsv_grad_mode != 0 && hsv_grad_mode != 1) return fail(EXPO_E_BADARG, "hsv_grad_mode must be 0 or 1");
  if (n == 0) return EXPO_OK;
  if (!filter_ids || !x || !dy || !params || !dparams) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? dispatch_bwd_t<half_t>(filter_ids, x, dy, dx, params, dparams, dpenalty, n, h, w,
                                                    hsv_grad_mode, workspace, workspace_bytes, s)
                           : dispatch_bwd_t<float>(filter_ids, x, dy, dx, params, dparams, dpenalty, n, h, w,
                                                   hsv_grad_mode, workspace, workspace_bytes, s);
}

int expo_chain_streams(int n, int h, int w, int dtype) {
  if (check_common(n, h, w, dtype) != EXPO_OK) return 0;
  return chain_plan(n, h, w, dtype).two_lanes ? 2 : 1;
}

int expo_chain_prepare(void* stream) {
  // The probe of DESIGN.md 3.5 as an explicit initialisation step: an integrator who does not want the first eager
  // two-stream chain call of `stream` to stall that stream (one one-wave kernel waiting <= 0.5 ms per candidate, two
  // stream synchronisations) calls this once after creating the stream.  Idempotent; a no-op while `stream` is
  // capturing or with EXPO_CHAIN_HELPER_PROBE=0.
  return helper_entry(static_cast<hipStream_t>(stream), true) ? EXPO_OK : fail(EXPO_E_HIP, "chain helper stream");
}

int expo_chain_release(void* stream) {
  // Forget the pairing of a caller stream that is about to be destroyed: its helper stream and events are destroyed
  // (the helper is idle between chain calls: every call joins it before it returns).  Returns EXPO_OK whether or not
  // the stream had a pairing.  Must not race with a chain call on the same stream.
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
  HelperEntry* found = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_helper_mu);
    for (size_t i = 0; i < g_helper_table.size(); ++i)
      if (g_helper_table[i]->dev == dev && g_helper_table[i]->caller == static_cast<hipStream_t>(stream)) {
        found = g_helper_table[i];
        g_helper_table.erase(g_helper_table.begin() + i);
        break;
      }
  }
  if (!found) return EXPO_OK;
  {
    std::lock_guard<std::mutex> lock(found->mu);
    (void)hipStreamSynchronize(found->fj.helper);
    (void)hipEventDestroy(found->fj.fork);
    (void)hipEventDestroy(found->fj.join);
    (void)hipStreamDestroy(found->fj.helper);
  }
  delete found;
  return EXPO_OK;
}

int expo_chain_helper_stats(int* probed, int* rejected) {
  if (probed) *probed = g_helper_probes;
  if (rejected) *rejected = g_helper_rejected;
  return EXPO_OK;
}

int expo_chain_fwd(const int* filter_ids, int steps, void* const* acts, const float* const* params, int n, int h,
                   int w, int dtype, void* stream) {
  if (steps < 0 || !filter_ids || !acts || !params) return fail(EXPO_E_BADARG, "bad chain arguments");
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // validate EVERYTHING before the first launch: an error means nothing was enqueued
  for (int i = 0; i < steps; ++i) {
    if (filter_ids[i] < 0 || filter_ids[i] >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
    if (!acts[i] || !acts[i + 1] || !params[i]) return fail(EXPO_E_BADARG, "null pointer");
  }
  const size_t esz = dtype == EXPO_F16 ? 2 : 4;
  const ChainPlan plan = chain_plan(n, h, w, dtype);
  ForkJoin* fj = plan.two_lanes ? fork_join_for_device(s) : nullptr;
  if (plan.two_lanes && !fj) return fail(EXPO_E_HIP, "helper stream unavailable");
  if (fj) {
    if (int rc = chain_fork(fj, s)) return rc;
  }
  int rc = EXPO_OK;
  // chunk-major: every chunk (a tile, or half a tile per stream) goes through all the steps before the next one starts
  for (size_t c = 0; c < plan.chunks.size() && !rc; ++c) {
    const ChainChunk& ck = plan.chunks[c];
    hipStream_t sp = ck.lane ? fj->helper : s;
    const size_t ioff = size_t(ck.nb) * h * w * 3 * esz;
    for (int i = 0; i < steps && !rc; ++i) {
      const int rev = plan.snake ? (i & 1) : 0;
      const void* xin = static_cast<const char*>(acts[i]) + ioff;
      void* yout = static_cast<char*>(acts[i + 1]) + ioff;
      const float* prm = params[i] + size_t(ck.nb) * kNumParams[filter_ids[i]];
      rc = dtype == EXPO_F16 ? fwd_by_id<half_t>(filter_ids[i], xin, yout, prm, ck.np, h, w, sp, rev, n)
                             : fwd_by_id<float>(filter_ids[i], xin, yout, prm, ck.np, h, w, sp, rev, n);
    }
  }
  if (fj) {  // always joined, also on an error path: a forked helper must not stay outside the caller's stream order
    const int jrc = chain_join(fj, s);
    if (!rc) rc = jrc;
  }
  return rc;
}

int expo_filter_bwd_records(int filter_id, const void* x, const void* dy, void* dx, const float* params, int n,
                            int h, int w, int dtype, int hsv_grad_mode, void* records, size_t records_bytes,
                            void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (filter_id < 0 || filter_id >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
  if (hsv_grad_mode != 0 && hsv_grad_mode != 1) return fail(EXPO_E_BADARG, "hsv_grad_mode must be 0 or 1");
  if (n == 0) return EXPO_OK;
  if (!x || !dy || !params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int bx = geom_bx(bwd_geom_kind(filter_id), n, h, w, dtype);
  float* rec;
  if (int rc = ws_check(records, records_bytes, n, bx, 1, &rec)) return rc;
  return dtype == EXPO_F16 ? bwd_by_id<half_t>(filter_id, x, dy, dx, params, rec, n, h, w, hsv_grad_mode, s)
                           : bwd_by_id<float>(filter_id, x, dy, dx, params, rec, n, h, w, hsv_grad_mode, s);
}

int expo_finish_bwd(const int* filter_ids, int steps, const float* const* params, float* const* dparams, int n,
                    int h, int w, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (steps < 0 || !filter_ids || !params || !dparams) return fail(EXPO_E_BADARG, "bad finish arguments");
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0 || steps == 0) return EXPO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // a step's records sit at the start of its slice; slices are strided by the largest record count of any kernel,
  // each step's kernel wrote (and this launch reads) the count of ITS geometry
  const int bx_max = geom_bx_max(n, h, w, dtype);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, bx_max, steps, &records)) return rc;
  const size_t step_floats = ws_step_bytes(n, bx_max) / sizeof(float);
  for (int i = 0; i < steps; ++i) {
    if (filter_ids[i] < 0 || filter_ids[i] >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
    if (!params[i] || !dparams[i]) return fail(EXPO_E_BADARG, "null pointer");
  }
  for (int i0 = 0; i0 < steps; i0 += kMaxFinishSteps) {
    const int cnt = steps - i0 < kMaxFinishSteps ? steps - i0 : kMaxFinishSteps;
    FinishArgs fa{};
    for (int k = 0; k < cnt; ++k)
      fa.s[k] = FinishStep{params[i0 + k], dparams[i0 + k], nullptr, records + size_t(i0 + k) * step_floats, nullptr,
                           kFinFilter, filter_ids[i0 + k], 0, 0.f,
                           geom_bx(bwd_geom_kind(filter_ids[i0 + k]), n, h, w, dtype)};
    if (int rc = launch_finish(fa, cnt, n, s)) return rc;
  }
  return EXPO_OK;
}

int expo_chain_bwd(const int* filter_ids, int steps, void* const* acts, void* const* grads,
                   const float* const* params, float* const* dparams, int n, int h, int w, int dtype,
                   int hsv_grad_mode, void* workspace, size_t workspace_bytes, void* stream) {
  if (steps < 0 || !filter_ids || !acts || !grads || !params || !dparams)
    return fail(EXPO_E_BADARG, "bad chain arguments");
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (hsv_grad_mode != 0 && hsv_grad_mode != 1) return fail(EXPO_E_BADARG, "hsv_grad_mode must be 0 or 1");
  if (n == 0 || steps == 0) return EXPO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // validate EVERYTHING before the first launch: an error means nothing was enqueued and no gradient was touched
  for (int i = 0; i < steps; ++i) {
    if (filter_ids[i] < 0 || filter_ids[i] >= EXPO_NUM_FILTERS) return fail(EXPO_E_BADARG, "filter_id out of range");
    if (!dparams[i] || !acts[i] || !grads[i + 1] || !params[i]) return fail(EXPO_E_BADARG, "null pointer");
  }
  // every step's kernel writes its block records into its own slice of the workspace; ONE finish launch
  // (per kMaxFinishSteps steps) then produces all the dparams
  const int bx_max = geom_bx_max(n, h, w, dtype);
  float* records;
  if (int rc = ws_check(workspace, workspace_bytes, n, bx_max, steps, &records)) return rc;
  const size_t step_floats = ws_step_bytes(n, bx_max) / sizeof(float);
  const size_t esz = dtype == EXPO_F16 ? 2 : 4;
  const ChainPlan plan = chain_plan(n, h, w, dtype);
  ForkJoin* fj = plan.two_lanes ? fork_join_for_device(s) : nullptr;
  if (plan.two_lanes && !fj) return fail(EXPO_E_HIP, "helper stream unavailable");
  if (fj) {
    if (int rc = chain_fork(fj, s)) return rc;
  }
  int rc = EXPO_OK;
  for (size_t c = 0; c < plan.chunks.size() && !rc; ++c) {
    const ChainChunk& ck = plan.chunks[c];
    hipStream_t sp = ck.lane ? fj->helper : s;
    const size_t ioff = size_t(ck.nb) * h * w * 3 * esz;
    for (int i = steps - 1; i >= 0 && !rc; --i) {
      // the forward launch of step steps-1 walked in direction (steps-1) & 1 and ENDED on the other side: the backward
      // starts there, i.e. walks the opposite way, and alternates down to step 0, whose direction (1) is again the
      // opposite of the next forward call's first launch (0).  (Until r04p22 this read (steps - i) & 1, which is the
      // same thing for an odd number of steps only: with 8 steps the first backward launch started on the images its
      // predecessor had touched FIRST -- the read-latency counters of profiles/r04_p22_cold_counters.md show it.)
      const int rev = plan.snake ? ((i + 1) & 1) : 0;
      const int bx = geom_bx(bwd_geom_kind(filter_ids[i]), n, h, w, dtype);  // records per image of this step's kernel
      float* rec = records + size_t(i) * step_floats + size_t(ck.nb) * bx * kWsSlots;
      const void* xin = static_cast<const char*>(acts[i]) + ioff;
      const void* gin = static_cast<const char*>(grads[i + 1]) + ioff;
      void* gout = grads[i] ? static_cast<char*>(grads[i]) + ioff : nullptr;
      const float* prm = params[i] + size_t(ck.nb) * kNumParams[filter_ids[i]];
      rc = dtype == EXPO_F16
               ? bwd_by_id<half_t>(filter_ids[i], xin, gin, gout, prm, rec, ck.np, h, w, hsv_grad_mode, sp, rev, n)
               : bwd_by_id<float>(filter_ids[i], xin, gin, gout, prm, rec, ck.np, h, w, hsv_grad_mode, sp, rev, n);
    }
  }
  if (fj) {
    const int jrc = chain_join(fj, s);
    if (!rc) rc = jrc;
  }
  if (rc) return rc;
  return expo_finish_bwd(filter_ids, steps, params, dparams, n, h, w, dtype, workspace, workspace_bytes, stream);
}

int expo_critic_stats(const void* x, float* stats, int n, int h, int w, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !stats) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? stats_t<half_t>(x, stats, n, h, w, workspace, workspace_bytes, s)
                           : stats_t<float>(x, stats, n, h, w, workspace, workspace_bytes, s);
}

int expo_overexposure_penalty(const void* y, float* penalty, int n, int h, int w, int dtype, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!y || !penalty) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? penalty_t<half_t>(y, penalty, n, h, w, workspace, workspace_bytes, s)
                           : penalty_t<float>(y, penalty, n, h, w, workspace, workspace_bytes, s);
}

int expo_vignet_apply_fwd(const void* x, void* y, const float* mask_params, float maximum_sharpness, int masking, int n,
                          int h, int w, int dtype, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !y || !mask_params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? vignet_fwd_t<half_t>(x, y, mask_params, maximum_sharpness, masking, n, h, w, s)
                           : vignet_fwd_t<float>(x, y, mask_params, maximum_sharpness, masking, n, h, w, s);
}

int expo_vignet_apply_bwd(const void* x, const void* dy, void* dx, const float* mask_params, float* dmask_params,
                          float maximum_sharpness, int masking, int n, int h, int w, int dtype, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !dy || !mask_params || !dmask_params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? vignet_bwd_t<half_t>(x, dy, dx, mask_params, dmask_params, maximum_sharpness, masking, n, h,
                                                  w, workspace, workspace_bytes, s)
                           : vignet_bwd_t<float>(x, dy, dx, mask_params, dmask_params, maximum_sharpness, masking, n, h,
                                                 w, workspace, workspace_bytes, s);
}

int expo_overexposure_penalty_bwd(const void* y, const float* dpenalty, void* dy, int n, int h, int w, int dtype,
                                  void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!y || !dpenalty || !dy) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? penalty_bwd_t<half_t>(y, dpenalty, dy, n, h, w, s)
                           : penalty_bwd_t<float>(y, dpenalty, dy, n, h, w, s);
}

int expo_critic_stats_bwd(const void* x, const float* stats, const float* dstats, void* dx, int n, int h, int w,
                          int dtype, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !stats || !dstats || !dx) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? stats_bwd_t<half_t>(x, stats, dstats, dx, n, h, w, s)
                           : stats_bwd_t<float>(x, stats, dstats, dx, n, h, w, s);
}

int expo_critic_stats_jvp(const void* x, const float* stats, const void* v, float* jv, int n, int h, int w, int dtype,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !stats || !v || !jv) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? stats_jvp_t<half_t>(x, stats, v, jv, n, h, w, workspace, workspace_bytes, s)
                           : stats_jvp_t<float>(x, stats, v, jv, n, h, w, workspace, workspace_bytes, s);
}

int expo_critic_penalty_tangent(const float* u, const float* x, const float* stats, float scale, float* t0, float* norm,
                                float* term, int n, int h, int w, void* stream) {
  if (int rc = check_common(n, h, w, EXPO_F32)) return rc;
  if (n == 0) return EXPO_OK;
  if (!u || !x || !stats || !t0 || !norm || !term) return fail(EXPO_E_BADARG, "null pointer");
  if (long(h) * w > (1L << 24)) return fail(EXPO_E_BADARG, "critic_penalty_tangent: at most 2^24 pixels per image");
  if (h * w <= 4096 && (reinterpret_cast<uintptr_t>(u) & 7) == 0 && (reinterpret_cast<uintptr_t>(t0) & 7) == 0)
    hipLaunchKernelGGL(critic_penalty_tangent_reg_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), u, x, stats,
                       scale, t0, norm, term, h * w);
  else
    hipLaunchKernelGGL(critic_penalty_tangent_kernel, dim3(n), dim3(1024), 0, static_cast<hipStream_t>(stream), u, x, stats,
                       scale, t0, norm, term, h * w);
  HIP_TRY(hipGetLastError(), "critic_penalty_tangent launch");
  return EXPO_OK;
}

int expo_net_inputs(const void* a, const int64_t* a_rows, const void* b, const int64_t* b_rows, const float* alpha,
                    const float* vec_a, const float* vec_b, int v0, float* planes, float* stats, float* x_out, int x_first,
                    int x_count, int n, int h, int w, int dtype, float offset, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!a || !b || !planes || !stats) return fail(EXPO_E_BADARG, "null pointer");
  if (long(h) * w > kNetInputsMaxPixels) return fail(EXPO_E_BADARG, "net_inputs: at most 4096 pixels per image (the image is held in LDS)");
  if (v0 < 0 || v0 > 58) return fail(EXPO_E_BADARG, "net_inputs: 0 <= v0 <= 58");
  if (v0 > 0 && (!vec_a || !vec_b)) return fail(EXPO_E_BADARG, "net_inputs: v0 > 0 needs vec_a and vec_b");
  if (v0 > 0 && alpha) return fail(EXPO_E_BADARG, "net_inputs: the interpolated block takes no per-image values");
  const int m = (alpha ? 3 : 2) * n;
  if (x_out && (x_first < 0 || x_count < 0 || x_first + x_count > m)) return fail(EXPO_E_BADARG, "net_inputs: x rows outside the batch");
  NetInputsArgs g{a, b, reinterpret_cast<const long long*>(a_rows), reinterpret_cast<const long long*>(b_rows), alpha, vec_a, vec_b,
                  planes, stats, x_out, n, h * w, v0, x_first, x_out ? x_count : 0, offset,
                  unsigned((0x100000000ull + (6 + v0) - 1) / (6 + v0))};
  // blocks per image: ~one block per CU (a CU streams ~25 GB/s whatever runs on it, so the store phase wants every CU busy;
  // every extra block of an image repeats its loads and sums: 192 images x 1 block 11.2 us, x 2 12.3, x 4 15.6; 128 images with
  // 17 planes x 1 / 2 / 4: 16.6 / 12.9 / 13.8), at least 4 KB of planes each
  int parts = (256 + m / 2) / m;
  const int max_parts = (h * w * (6 + v0) + 1023) / 1024;
  if (parts > max_parts) parts = max_parts;
  if (parts > 16) parts = 16;
  if (parts < 1) parts = 1;
  const dim3 grid(m, parts);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == EXPO_F16) hipLaunchKernelGGL((net_inputs_kernel<half_t, 1024>), grid, dim3(1024), 0, s, g);
  else hipLaunchKernelGGL((net_inputs_kernel<float, 1024>), grid, dim3(1024), 0, s, g);
  HIP_TRY(hipGetLastError(), "net_inputs launch");
  return EXPO_OK;
}

int expo_critic_stats_hvp(const void* x, const float* dstats, const float* jv, const void* v, void* out, int n, int h,
                          int w, int dtype, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !dstats || !jv || !v || !out) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? stats_hvp_t<half_t>(x, dstats, jv, v, out, n, h, w, s)
                           : stats_hvp_t<float>(x, dstats, jv, v, out, n, h, w, s);
}

}  // extern "C"
#endif  // EXPO_PROBE
