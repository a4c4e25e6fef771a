// chain_fused_bwd.hip -- one-pass backward of a whole per-image filter sequence (expo_chain_fused_bwd).
//
// A benchmark construct, not a path of the reference: in the reference the parameters of step k+1 depend on the
// image after step k through the CNN (agent.py:30-125), so its backward is step by step and that is what
// expo_chain_bwd / expo_filter_bwd serve (the headline workload).  When the per-image sequence (filter id,
// parameters) x steps is fixed -- the situation of expo_chain_fused_fwd (net.py:796-821) -- the gradients with
// respect to the input image and to every step's parameters need ONE read of x and dy and ONE write of dx:
// 18 B/pixel instead of 18 B/pixel PER STEP, with the activations recomputed in registers.
//
//   forward recompute   a_0 = x;  a_{k+1} = f_k(a_k)  in fp32 (as expo_chain_fused_fwd computes them); the input of
//                       every step is kept as a CHECKPOINT in the storage type.  A lane's 48-byte group goes through
//                       the sequence in two halves (4 fp16 / 2 fp32 pixels = 6 VGPRs per checkpoint, 48 VGPRs for 8
//                       steps; whole groups need 96 and the kernel then spills at two waves per SIMD).  With
//                       fp16 storage the backward therefore linearises every step at the fp16-rounded activation,
//                       the value the per-step chain stores between its launches.
//   backward            d_K = dy;  d_k = J_k(a_k)^T d_{k+1}, fp32 between the steps (the per-step chain rounds d to
//                       the storage type after every launch); parameter-gradient partial sums per step.
//
// The step loop is FULLY UNROLLED (checkpoints and per-step sums are register arrays indexed at compile time; each
// step is a block-uniform switch over the filter id), so sequences are limited to EXPO_FUSED_BWD_MAX_STEPS steps.
// Everything a step needs that does not depend on the pixel is prepared once per block, in front of the chunk loop:
// the element-wise filters' derived parameters (SGPRs), the curve filters' forward segment tables and backward slope
// tables (LDS, one per step).  A block walks several 3 KiB chunks per wave, so that set-up and the epilogue are
// amortised; parameter-gradient partial sums of the element-wise filters (<= 3 per step) are carried per lane across
// the chunks (in an LDS column per thread: the registers are needed elsewhere) and cross the wave once at the end,
// those of the curve filters (8 / 24 per step) are reduce-scattered over the wave after every half group
// (kernel_common.h) and only the lane's slot is carried.  Block records + finish launch as everywhere else (no float
// atomics, bit-reproducible).  Compiled with -fno-slp-vectorize (build.sh): the packed-fp32 pairs the SLP vectoriser
// forms cost this kernel ~100 VGPRs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/exposure_hip.h"
#include "filter_math.h"
#include "pixel_io.h"
#include "kernel_common.h"
#include "host_common.h"

#ifndef EXPO_FUSED_BWD_MIN_WAVES
#define EXPO_FUSED_BWD_MIN_WAVES 2  // waves per SIMD the register budget is set for (probe builds: -DEXPO_FUSED_BWD_MIN_WAVES=3)
#endif

// fp16 storage: 1 = the curve steps use the per-step kernels' packed-fp16 accumulation and 256-entry bit-pattern slope
// table (CurveF::bwd_group<.., F16X>).  That path takes the upstream gradient as an fp16 value -- exact in the per-step
// chain, where it comes out of fp16 storage, but HERE the gradient is fp32 between the steps and would be rounded for
// the curve parameter sums -- and it costs 48 KB of LDS and ~100 VGPRs for nothing: 0.322 ms with it, 0.329 ms without
// at 64x512x512 (gpurun r03p25; the kernel is bound by the element-wise bodies).  Default: the fp32 path.
#ifndef EXPO_FUSED_BWD_F16X
#define EXPO_FUSED_BWD_F16X 0
#endif

namespace expo {

constexpr int kFusedBwdSteps = EXPO_FUSED_BWD_MAX_STEPS;
constexpr int kFusedBwdParts = 4;  // independent partial sums of an element-wise filter's accumulators (as kAccParts)

template <class Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn&& fn, std::integer_sequence<int, I...>) {
  (fn(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  static_for_impl(fn, std::make_integer_sequence<int, N>{});
}

// Half of a lane's group: rows (2 h, 2 h + 1) of the RawGroup = 4 fp16 / 2 fp32 whole pixels, 6 dwords.
struct RawHalf { u32x3_t q[2]; };
template <typename T> __device__ __forceinline__ void unpack_half(const RawHalf& r, float* out);
template <> __device__ __forceinline__ void unpack_half<half_t>(const RawHalf& r, float* out) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const uint32_t w = r.q[j][e];
      const half2_t h = __builtin_bit_cast(half2_t, w);
      out[j * 6 + e * 2] = float(h[0]);
      out[j * 6 + e * 2 + 1] = float(h[1]);
    }
  }
}
template <> __device__ __forceinline__ void unpack_half<float>(const RawHalf& r, float* out) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const uint32_t w = r.q[j][e];
      out[j * 3 + e] = __builtin_bit_cast(float, w);
    }
  }
}
template <typename T> __device__ __forceinline__ RawHalf pack_half(const float* in);
template <> __device__ __forceinline__ RawHalf pack_half<half_t>(const float* in) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  RawHalf r;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      half2_t h;  // saturating under MODE.FP16_OVFL (pixel_io.h: pack)
      h[0] = half_t(in[j * 6 + e * 2]);
      h[1] = half_t(in[j * 6 + e * 2 + 1]);
#if !EXPO_FP16_OVFL
      const half2_t hi = {half_t(65504.0f), half_t(65504.0f)}, lo = {half_t(-65504.0f), half_t(-65504.0f)};
      h = __builtin_elementwise_max(__builtin_elementwise_min(h, hi), lo);
#endif
      r.q[j][e] = __builtin_bit_cast(uint32_t, h);
    }
  }
  return r;
}
template <> __device__ __forceinline__ RawHalf pack_half<float>(const float* in) {
  RawHalf r;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) r.q[j][e] = __builtin_bit_cast(uint32_t, in[j * 3 + e]);
  }
  return r;
}

// accumulators per filter id (NACC of filter_math.h), 0 for id -1 / out of range
__device__ __forceinline__ int nacc_of(int id) {
  switch (id) {
    case 0: case 1: case 3: case 5: case 6: return 1;
    case 2: return 3;
    case 4: return 8;
    case 7: return 24;
    case 8: return 2;
    default: return 0;
  }
}

// derived parameters of an element-wise filter (<= 3 floats), wave-uniform -> SGPRs
struct StepPrm { float f[3]; };
template <class F>
__device__ __forceinline__ StepPrm step_prm(const float* __restrict__ p) {
  static_assert(sizeof(typename F::Prm) <= sizeof(StepPrm), "element-wise filters derive at most 3 values");
  const typename F::Prm q = F::load(p);
  StepPrm s{{0.f, 0.f, 0.f}};
  __builtin_memcpy(&s, &q, sizeof(q));
#pragma unroll
  for (int j = 0; j < 3; ++j)
    s.f[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s.f[j])));
  return s;
}
template <class F>
__device__ __forceinline__ typename F::Prm as_prm(const StepPrm& s) {
  typename F::Prm q;
  __builtin_memcpy(&q, &s, sizeof(q));
  return q;
}

template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads, EXPO_FUSED_BWD_MIN_WAVES) void chain_fused_bwd_kernel(
    const int32_t* __restrict__ ids, const float* __restrict__ params, int steps, const T* __restrict__ x,
    const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ records, size_t step_floats, int hw, int groups,
    int mode) {
  constexpr int PPL = PixTraits<T>::PPL / 2;  // pixels of a HALF group (see the header)
  constexpr int NV = PPL * 3;
  constexpr int KS = kFusedBwdSteps;
  constexpr bool kF16X = std::is_same<T, half_t>::value && EXPO_FUSED_BWD_F16X;
  // backward slope table of a curve step: fp16 storage -> 256-entry bit-pattern table per curve, else the segment table
  constexpr int kLut = kF16X ? ColorF::kLutFloats : 2 * ColorF::NP;
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int32_t* idn = ids + size_t(n) * steps;
  const float* prn = params + size_t(n) * steps * EXPO_MAX_PARAMS;

  __shared__ __attribute__((aligned(16))) float lut_all[KS][kLut];
  __shared__ float2_lut tab_all[KS][32];
  __shared__ float red[KS][kWaves][kWsSlots];
  __shared__ int id_sh[KS];
  __shared__ float tot_sh[KS][3][kThreads];

#if EXPO_FP16_OVFL
  // MODE.FP16_OVFL: fp16 conversions saturate at +-65504 (checkpoints and dx; pixel_io.h)
  if constexpr (sizeof(T) == 2) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
#endif
  // ---- per-block set-up: ids, derived parameters, curve tables
  int id[KS];
  StepPrm pq[KS];
  static_for<KS>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    id[K] = -1;
    pq[K] = StepPrm{{0.f, 0.f, 0.f}};
    if (K < steps) {
      id[K] = idn[K];
      const float* p = prn + K * EXPO_MAX_PARAMS;
      switch (id[K]) {
        case 0: pq[K] = step_prm<ExposureF>(p); break;
        case 1: pq[K] = step_prm<GammaF>(p); break;
        case 2: pq[K] = step_prm<WhiteBalanceF>(p); break;
        case 3: pq[K] = step_prm<SatPlusF>(p); break;
        case 5: pq[K] = step_prm<ContrastF>(p); break;
        case 6: pq[K] = step_prm<WnbF>(p); break;
        case 8: pq[K] = step_prm<LevelF>(p); break;
        case 4: {
          const float kl = p[lane % ToneF::NP];
          if constexpr (kF16X) ToneF::stage16_lanes(kl, lut_all[K]); else ToneF::stage(p, lut_all[K]);
          if (wv == (K & (kWaves - 1))) curve_lut_build<1>(kl, tab_all[K]);
        } break;
        case 7: {
          const float kl = p[lane % ColorF::NP];
          if constexpr (kF16X) ColorF::stage16_lanes(kl, lut_all[K]); else ColorF::stage(p, lut_all[K]);
          if (wv == (K & (kWaves - 1))) curve_lut_build<3>(kl, tab_all[K]);
        } break;
        default: break;
      }
    }
    if (threadIdx.x == 0) id_sh[K] = id[K];
  });
  __syncthreads();

  // per-step parameter-gradient sums carried across the chunks: element-wise filters per lane (<= 3), curve filters
  // the lane's slot of the wave total.  Kept in LDS, one column per thread (24 VGPRs otherwise, which the fp16 kernel
  // does not have at two waves per SIMD); a thread only ever touches its own column, so no synchronisation.
  float* const tot = &tot_sh[0][0][threadIdx.x];
  auto tot_at = [&](int k, int j) -> float& { return tot[(k * 3 + j) * kThreads]; };
  static_for<KS>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    if (K < steps) tot_at(K, 0) = tot_at(K, 1) = tot_at(K, 2) = 0.f;
  });

  auto fwd_step = [&](auto kc, const float* in, float* out) {
    constexpr int K = decltype(kc)::value;
#define EXPO_CASE(ID, F)                                                                  \
  case ID: {                                                                              \
    const typename F::Prm q = as_prm<F>(pq[K]);                                           \
    _Pragma("unroll") for (int k = 0; k < PPL; ++k) F::fwd(q, in + 3 * k, out + 3 * k); \
  } break;
    switch (id[K]) {
      EXPO_CASE(0, ExposureF)
      EXPO_CASE(1, GammaF)
      EXPO_CASE(2, WhiteBalanceF)
      EXPO_CASE(3, SatPlusF)
      case 4: curve_lut_map<1, PPL>(in, out, tab_all[K]); break;
      EXPO_CASE(5, ContrastF)
      EXPO_CASE(6, WnbF)
      case 7: curve_lut_map<3, PPL>(in, out, tab_all[K]); break;
      EXPO_CASE(8, LevelF)
      default:  // id -1: the image becomes 0 (written as arithmetic, see chain_fused.hip)
#pragma unroll
        for (int j = 0; j < NV; ++j) out[j] = fmaf(clamp01x(in[j], -65504.0f, 65504.0f), 0.0f, 0.0f);
        break;
    }
#undef EXPO_CASE
  };

  // element-wise filter: per-pixel backward, kFusedBwdParts independent partial sums, added to the lane's totals
  auto bwd_elementwise = [&](auto fc, auto kc, const float* xk, float* d) {
    using F = typename decltype(fc)::type;
    constexpr int K = decltype(kc)::value;
    constexpr int kParts = kFusedBwdParts < PPL ? kFusedBwdParts : PPL;
    const typename F::Prm q = as_prm<F>(pq[K]);
    float acc[kParts][F::NACC];
#pragma unroll
    for (int p = 0; p < kParts; ++p)
#pragma unroll
      for (int j = 0; j < F::NACC; ++j) acc[p][j] = 0.f;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      float dxp[3];
      F::bwd(q, nullptr, xk + 3 * k, d + 3 * k, dxp, acc[k % kParts], mode);
#pragma unroll
      for (int c = 0; c < 3; ++c) d[3 * k + c] = dxp[c];
    }
#pragma unroll
    for (int j = 0; j < F::NACC; ++j) {
      float s = acc[0][j];
#pragma unroll
      for (int p = 1; p < kParts; ++p) s += acc[p][j];
      tot_at(K, j) += s;
    }
  };
  auto bwd_curve = [&](auto fc, auto kc, const float* xk, float* d) {
    using F = typename decltype(fc)::type;
    constexpr int K = decltype(kc)::value;
    float acc[F::NACC];
#pragma unroll
    for (int j = 0; j < F::NACC; ++j) acc[j] = 0.f;
    typename F::Prm q;  // (bwd_group reads the tables only)
    F::template bwd_group<PPL, kF16X, false>(q, lut_all[K], xk, d, acc, nullptr);
    wave_reduce_scatter<F::NACC>(acc, lane);
    tot_at(K, 0) += acc[0];
  };
  auto bwd_step = [&](auto kc, const float* xk, float* d) {
    constexpr int K = decltype(kc)::value;
#define EXPO_CASE(ID, F) \
  case ID: bwd_elementwise(std::common_type<F>{}, kc, xk, d); break;
    switch (id[K]) {
      EXPO_CASE(0, ExposureF)
      EXPO_CASE(1, GammaF)
      EXPO_CASE(2, WhiteBalanceF)
      EXPO_CASE(3, SatPlusF)
      case 4: bwd_curve(std::common_type<ToneF>{}, kc, xk, d); break;
      EXPO_CASE(5, ContrastF)
      EXPO_CASE(6, WnbF)
      case 7: bwd_curve(std::common_type<ColorF>{}, kc, xk, d); break;
      EXPO_CASE(8, LevelF)
      default:  // id -1: the output does not depend on the input
#pragma unroll
        for (int j = 0; j < NV; ++j) d[j] = 0.f;
        break;
    }
#undef EXPO_CASE
  };

  // one pixel group through the whole sequence: xv = the loaded x, dv = dy on entry / dx on return
  auto run = [&](const float* xv, float* dv) {
    RawHalf ck[KS];
    float a[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) a[j] = xv[j];
    static_for<KS>([&](auto kc) {
      constexpr int K = decltype(kc)::value;
      if (K < steps) {
        ck[K] = pack_half<T>(a);
        if (K + 1 < steps) {  // (the last step's output is not needed: the caller holds y)
          float w[NV];
          fwd_step(kc, a, w);
#pragma unroll
          for (int j = 0; j < NV; ++j) a[j] = w[j];
        }
      }
    });
    float d[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) d[j] = dv[j];
    static_for<KS>([&](auto kr) {
      constexpr int K = KS - 1 - decltype(kr)::value;
      if (K < steps) {
        float xk[NV];
        unpack_half<T>(ck[K], xk);
        bwd_step(std::integral_constant<int, K>{}, xk, d);
      }
    });
#pragma unroll
    for (int j = 0; j < NV; ++j) dv[j] = d[j];
  };

  const int stride = gridDim.x * kThreads;
  if constexpr (VEC) {
    // (no software prefetch: the kernel is VALU-bound and the other waves of the SIMD cover the two loads)
    const __amdgpu_buffer_rsrc_t rx = make_image_rsrc(x + off, hw), rdy = make_image_rsrc(dy + off, hw);
    const __amdgpu_buffer_rsrc_t rdx = make_image_rsrc(dx + off, hw);
    for (int gw = blockIdx.x * kThreads + (threadIdx.x & ~63); gw * (2 * PPL) < hw; gw += stride) {  // wave-uniform
      const int bo = chunk_byte_offset<T>(gw, lane);
      // the two halves one after the other, ROLLED: each loads its own two rows (the unrolled form doubles the
      // 170 KB of code and keeps the other half's 12 registers alive across the whole sequence)
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        const int bh = bo + h * 1536;
        RawHalf gx, gd;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          gx.q[j] = __builtin_amdgcn_raw_buffer_load_b96(rx, bh + j * 768, 0, IO::kLoadX);
          gd.q[j] = __builtin_amdgcn_raw_buffer_load_b96(rdy, bh + j * 768, 0, IO::kLoadDy);
        }
        float v[NV], d[NV];
        unpack_half<T>(gx, v);
        unpack_half<T>(gd, d);
        run(v, d);
        const RawHalf o = pack_half<T>(d);
#pragma unroll
        for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_buffer_store_b96(o.q[j], rdx, bh + j * 768, 0, IO::kStore);
      }
    }
  } else {
    // wave-uniform trip count: the curve steps' reduce-scatter needs every lane of the wave (groups past the end load
    // zeros and store nothing); one whole group = two halves of PPL pixels
    for (int g0 = blockIdx.x * kThreads + (threadIdx.x & ~63); g0 < groups; g0 += stride) {
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        const int px0 = ((g0 + lane) * 2 + h) * PPL;  // first pixel of this lane's half group
        const T* xs = x + off;
        const T* ds = dy + off;
        T* os = dx + off;
        float v[NV], d[NV];
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          const bool in = px0 + k < hw;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            v[3 * k + c] = in ? float(xs[size_t(px0 + k) * 3 + c]) : 0.0f;
            d[3 * k + c] = in ? float(ds[size_t(px0 + k) * 3 + c]) : 0.0f;
          }
        }
        run(v, d);
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
          if (px0 + k < hw) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              float o = d[3 * k + c];
              if (sizeof(T) == 2) o = __builtin_amdgcn_fmed3f(o, -65504.0f, 65504.0f);  // saturate fp16 (store_slow)
              os[size_t(px0 + k) * 3 + c] = T(o);
            }
          }
        }
      }
    }
  }

  // ---- epilogue: wave totals -> block record of every step
  static_for<KS>([&](auto kc) {
    constexpr int K = decltype(kc)::value;
    if (K < steps) {
      const int na = nacc_of(id[K]);
      if (na > 0) {
        if (na <= 3) {  // element-wise: the per-lane sums still have to cross the wave
          float acc[3] = {tot_at(K, 0), tot_at(K, 1), tot_at(K, 2)};
          const int idx = wave_reduce_scatter<3>(acc, lane);
          if ((lane & 1) == 0 && idx >= 0) red[K][wv][idx] = acc[0];
        } else {
          const int idx = reduce_scatter_index(na, lane);
          if ((lane & 1) == 0 && idx >= 0) red[K][wv][idx] = tot_at(K, 0);
        }
      }
    }
  });
  __syncthreads();
  {
    const int k = threadIdx.x / kWsSlots, j = threadIdx.x % kWsSlots;
    static_assert(KS * kWsSlots <= kThreads, "one thread per (step, slot)");
    if (k < steps && j < nacc_of(id_sh[k])) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) v += red[k][w][j];
      records[size_t(k) * step_floats + (size_t(n) * gridDim.x + blockIdx.x) * kWsSlots + j] = v;
    }
  }
}

template <typename T>
static int chain_fused_bwd_t(const int32_t* ids, const float* params, int steps, const void* x, const void* dy,
                             void* dx, float* dparams, float* records, size_t step_floats, int n, int h, int w,
                             int mode, hipStream_t s) {
  const Geom g = make_geom<T>(n, h, w, {x, dy, dx}, kGeomFusedBwd);
  const dim3 grid(g.blocks_x, n), block(kThreads);
#define EXPO_LAUNCH(VEC, IO)                                                                                          \
  hipLaunchKernelGGL((chain_fused_bwd_kernel<T, VEC, IO>), grid, block, 0, s, ids, params, steps, (const T*)x,        \
                     (const T*)dy, (T*)dx, records, step_floats, g.hw, g.groups, mode)
  if (g.stream) EXPO_LAUNCH(true, IoStream);
  else if (g.vec) EXPO_LAUNCH(true, IoCached);
  else EXPO_LAUNCH(false, IoCached);
#undef EXPO_LAUNCH
  HIP_TRY(hipGetLastError(), "chain_fused_bwd launch");
  return finish_chain_fused(ids, params, dparams, records, step_floats, steps, n, g.blocks_x, s);
}

}  // namespace expo

using namespace expo;

extern "C" {

int expo_chain_fused_bwd(const int32_t* filter_ids, const float* params, int steps, const void* x, const void* dy,
                         void* dx, float* dparams, int n, int h, int w, int dtype, int hsv_grad_mode, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (steps < 0 || steps > EXPO_FUSED_BWD_MAX_STEPS) return fail(EXPO_E_BADARG, "steps must be in [0, EXPO_FUSED_BWD_MAX_STEPS]");
  if (hsv_grad_mode != 0 && hsv_grad_mode != 1) return fail(EXPO_E_BADARG, "hsv_grad_mode must be 0 or 1");
  if (n == 0) return EXPO_OK;
  if (!x || !dy || !dx || (steps > 0 && (!filter_ids || !params || !dparams))) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t esz = dtype == EXPO_F16 ? 2 : 4;
  if (steps == 0) {  // the empty sequence is the identity: dx = dy
    if (dx != dy) HIP_TRY(hipMemcpyAsync(dx, dy, size_t(n) * h * w * 3 * esz, hipMemcpyDeviceToDevice, s), "dx = dy");
    return EXPO_OK;
  }
  float* records = nullptr;
  size_t step_floats = 0;
  if (int rc = chain_records(workspace, workspace_bytes, n, h, w, dtype, steps, &records, &step_floats)) return rc;
  return dtype == EXPO_F16 ? chain_fused_bwd_t<half_t>(filter_ids, params, steps, x, dy, dx, dparams, records,
                                                       step_floats, n, h, w, hsv_grad_mode, s)
                           : chain_fused_bwd_t<float>(filter_ids, params, steps, x, dy, dx, dparams, records,
                                                      step_floats, n, h, w, hsv_grad_mode, s);
}

}  // extern "C"
