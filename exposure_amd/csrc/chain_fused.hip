// chain_fused.hip -- the fused multi-step forward of the high-resolution inference path
// (expo_chain_fused_fwd; /root/reference/net.py:796-821, BASELINE config 5), in its own translation unit
// because it is compiled with -fno-slp-vectorize: this kernel is compute-heavy (8 filter bodies back to
// back on values that stay in registers), and clang's SLP vectoriser turns pairs of independent fp32
// operations into v_pk_mul_f32 / v_pk_fma_f32 plus the v_mov's that assemble their operand pairs.  The
// packed forms issue at the scalar rate on gfx950 (tools/valubench), but the kernel is bound by its
// dependent chains, not by issue slots (SQ counters, profiles/r02_experiments.md r02p18/19), so the
// packing buys nothing and the moves cost: 1 203 VALU with 216 v_mov -> 24.6 us, against 1 262
// scalar VALU with 102 v_mov -> 22.7 us at 16x512x512x3 fp16 (gpurun r02p5); a hand-paired build with
// 963 VALU measured the same as this one.  The streaming kernels of exposure_hip.hip keep the default
// (-0.8 % for the chain with the flag).
// -fno-honor-nans drops the canonicalising v_max_f32 x, x, x in front of every min / max / med3 whose
// operand comes from memory or LDS (1 218 -> 1 171 VALU): a NaN pixel is not a defined input of the
// inference path (the result for such a pixel is unspecified; every other pixel is unaffected).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exposure_hip.h"
#include "filter_math.h"
#include "pixel_io.h"
#include "kernel_common.h"
#include "host_common.h"

#ifndef EXPO_FUSED_MIN_WAVES
#define EXPO_FUSED_MIN_WAVES  // e.g. -DEXPO_FUSED_MIN_WAVES=,8 : register budget for 8 waves per SIMD (probe builds)
#endif

namespace expo {

// ------------------------------------------------------- fused multi-step forward (inference)
// The high-resolution inference path (net.py:796-821; BASELINE config 5): the per-step parameters
// are regressed on the 64x64 proxy only, so by the time the full-resolution image is touched the
// whole per-image sequence (filter id, parameters) x steps is known.  This kernel applies all
// `steps` filters to a pixel group while it sits in registers (fp32 between steps -- no fp16
// rounding of intermediates): ONE read and ONE write of the image instead of one per step.
// Each wave owns exactly one 3 KiB chunk, so the per-step parameters are fetched once per wave
// through scalar loads; the step loop is rolled (block-uniform switch per step).
template <typename T, bool VEC, class IO>
__global__ __launch_bounds__(kThreads EXPO_FUSED_MIN_WAVES) void chain_fused_fwd_kernel(const int32_t* __restrict__ ids,
                                                                   const float* __restrict__ params, int steps,
                                                                   const T* __restrict__ x, T* __restrict__ y,
                                                                   int hw, int groups) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int n = blockIdx.y;
  const size_t off = size_t(n) * hw * 3;
  const T* xi = x + off;
  T* yi = y + off;
  const int32_t* idn = ids + size_t(n) * steps;
  const float* prn = params + size_t(n) * steps * EXPO_MAX_PARAMS;
  __shared__ float2_lut curve_tab[kWaves][32];
  float2_lut* const tab = curve_tab[threadIdx.x >> 6];
  const int plane = (threadIdx.x & 63) % EXPO_MAX_PARAMS;  // which parameter this lane mirrors
  // One step, OUT OF PLACE (in -> out).  The step loop below runs two steps per trip with the two pixel arrays (and
  // the two parameter sets) swapping roles, so no loop-carried value is ever copied: the rolled one-step loop paid 24
  // v_mov + 24 s_mov per step for its loop PHIs (the coupled filters cannot update a pixel in place), ~17 % of the
  // instructions a wave issued for an 8-step sequence.
  auto apply = [&](int id, const float* prm, float klane, const float* in, float* out) {
#define EXPO_CASE(ID, F)                              \
  case ID: {                                          \
    const typename F::Prm q = F::load(prm);           \
    _Pragma("unroll") for (int k = 0; k < PPL; ++k) F::fwd(q, in + 3 * k, out + 3 * k); \
  } break;
    switch (id) {
      EXPO_CASE(0, ExposureF)
      EXPO_CASE(1, GammaF)
      EXPO_CASE(2, WhiteBalanceF)
      EXPO_CASE(3, SatPlusF)
      case 4:
        curve_lut_build<1>(klane, tab);
        curve_lut_map<1, PPL>(in, out, tab);
        __builtin_amdgcn_wave_barrier();  // the next curve step of this wave rewrites the table
        break;
      EXPO_CASE(5, ContrastF)
      EXPO_CASE(6, WnbF)
      case 7:
        curve_lut_build<3>(klane, tab);
        curve_lut_map<3, PPL>(in, out, tab);
        __builtin_amdgcn_wave_barrier();
        break;
      EXPO_CASE(8, LevelF)
      default:  // id -1 (the all-zero one-hot selects nothing) -> the image becomes 0.  Written as arithmetic (clamp,
        // then x * 0 + 0: exactly +0 for every input incl. inf / NaN) rather than as 24 constant moves: those the
        // compiler executes speculatively in front of the neighbouring case (Exposure) on EVERY step.  (No
        // __builtin_unreachable() for ids outside [-1, 8] either: with it hipcc 7.2 drops the first two values of the
        // fp32 kernel's first pixel row -- found by the fp32 parity test, gpurun r03p17.)
#pragma unroll
        for (int j = 0; j < PPL * 3; ++j) out[j] = fmaf(clamp01x(in[j], -65504.0f, 65504.0f), 0.0f, 0.0f);
        break;
    }
#undef EXPO_CASE
  };
  auto run = [&](float* v) {
    // software-pipelined parameter fetch: a step's id and 24 parameters (wave-uniform -> scalar loads into SGPRs)
    // are requested one step ahead, hiding the scalar-load latency; `k*` is a per-lane copy (lane l <-> parameter l)
    // for the curve table.  Two parameter sets alternate like the pixel arrays.
    float pa[EXPO_MAX_PARAMS], pb[EXPO_MAX_PARAMS];
    float ka = 0.f, kb = 0.f;
    int ia = 0, ib = 0;
    // a step past the end (the second half of the last trip of an odd-length sequence) is the identity: Exposure with
    // 0 EV, x * 2^0 = x exactly -- no extra case in the switch
    auto fetch = [&](int st, float* p, float& kl, int& id) {
      const bool live = st < steps;
      const int sn = live ? st : steps - 1;  // (past the end: any valid row)
      id = live ? idn[sn] : 0;
      kl = prn[sn * EXPO_MAX_PARAMS + plane];
#pragma unroll
      for (int j = 0; j < EXPO_MAX_PARAMS; ++j) p[j] = prn[sn * EXPO_MAX_PARAMS + j];
      if (!live) p[0] = 0.0f;
    };
    if (steps <= 0) return;
    // (the pixel arrays are locals, copied in and out -- free in SSA form.  Running the loop directly on the caller's
    // array made hipcc 7.2 allocate the fp32 kernel's store ADDRESS register inside the 96-bit data tuple of the first
    // pixel row: R and G of every 64th pixel wrong.  Caught by the fp32 parity tests, gpurun r03p17;
    // tests/test_isa_sanity.py now scans every kernel's ISA for that overlap.)
    float a[PPL * 3], w[PPL * 3];
#pragma unroll
    for (int j = 0; j < PPL * 3; ++j) a[j] = v[j];
    fetch(0, pa, ka, ia);
#pragma unroll 1
    for (int st = 0; st < steps; st += 2) {
      fetch(st + 1, pb, kb, ib);
      apply(ia, pa, ka, a, w);
      fetch(st + 2, pa, ka, ia);
      apply(ib, pb, kb, w, a);
    }
#pragma unroll
    for (int j = 0; j < PPL * 3; ++j) v[j] = a[j];
  };
  const int stride = gridDim.x * kThreads;
  if constexpr (VEC) {
    const T* const ins[1] = {xi};
    stream_groups<T, 1, true, false, IO>(ins, yi, hw, blockIdx.x * kThreads + (threadIdx.x & ~63), stride,
                                     [&](float (&v)[1][PPL * 3], int) { run(v[0]); });
  } else {
    // wave-uniform trip count: curve_fwd_lut needs lanes 0..23 of every wave alive (groups past the
    // end load zeros and store nothing)
    for (int g0 = blockIdx.x * kThreads + (threadIdx.x & ~63); g0 < groups; g0 += stride) {
      const int g = g0 + (threadIdx.x & 63);
      float v[PPL * 3];
      load_slow<T>(xi, g, hw, v);
      run(v);
      store_slow<T>(yi, g, hw, v);
    }
  }
}

template <typename T>
static int chain_fused_fwd_t(const int32_t* ids, const float* params, int steps, const void* x, void* y, int n,
                             int h, int w, hipStream_t s) {
  Geom g = make_geom<T>(n, h, w, {x, y}, kGeomMap);
  g.blocks_x = (g.groups + kThreads - 1) / kThreads;  // one chunk per wave: parameters fetched once
  const dim3 grid(g.blocks_x, n), block(kThreads);
  if (g.stream)
    hipLaunchKernelGGL((chain_fused_fwd_kernel<T, true, IoStream>), grid, block, 0, s, ids, params, steps, (const T*)x, (T*)y, g.hw, g.groups);
  else if (g.vec)
    hipLaunchKernelGGL((chain_fused_fwd_kernel<T, true, IoCached>), grid, block, 0, s, ids, params, steps, (const T*)x, (T*)y, g.hw, g.groups);
  else
    hipLaunchKernelGGL((chain_fused_fwd_kernel<T, false, IoCached>), grid, block, 0, s, ids, params, steps, (const T*)x, (T*)y, g.hw, g.groups);
  HIP_TRY(hipGetLastError(), "chain_fused_fwd launch");
  return EXPO_OK;
}

}  // namespace expo

using namespace expo;

extern "C" {

int expo_chain_fused_fwd(const int32_t* filter_ids, const float* params, int steps, const void* x, void* y, int n,
                         int h, int w, int dtype, void* stream) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (steps < 0 || steps > 64) return fail(EXPO_E_BADARG, "steps must be in [0, 64]");
  if (n == 0) return EXPO_OK;
  if (!x || !y || (steps > 0 && (!filter_ids || !params))) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == EXPO_F16 ? chain_fused_fwd_t<half_t>(filter_ids, params, steps, x, y, n, h, w, s)
                           : chain_fused_fwd_t<float>(filter_ids, params, steps, x, y, n, h, w, s);
}

}  // extern "C"
