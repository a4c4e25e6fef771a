// curve_generic.hip -- ToneFilter / ColorFilter for ANY cfg.curve_steps (filters.py:264-273, 312-322;
// config_example.py:27 sets 8, and the tuned kernels of exposure_hip.hip are built for 8).  The reference loops
// `for i in range(cfg.curve_steps)`; a config with another step count runs here: one element-wise pass per direction,
// O(1) work per element whatever L is, no per-L instantiation.  Correctness first -- these kernels use plain
// coalesced element loads, not the dwordx3 streaming skeleton; L == 8 never comes here (exposure_amd/filters.py).
//
//   y = (L/S) sum_i clip(x - i/L, 0, 1/L) k_i,  S = sum_i k_i + 1e-30
// With x^ = clamp(x, 0, 1) and j = min(floor(L x^), L-1) (the segment x^ lies in) the clips are 1/L for i < j,
// x^ - j/L for i = j and 0 above:   sum_i clip_i k_i = prefix_j / L + (x^ - j/L) k_j,  prefix_j = sum_{i<j} k_i.
// Backward (TF conventions): dx = dy (L/S) sum_i k_i [0 <= x - i/L <= 1/L] (both bounds inclusive:
// exactly on a knot two segments contribute);  dk_i = sum dy ((L/S) clip_i - y/S).  Per element only segment j has a
// data-dependent clip, so a thread keeps, per curve, H_j = sum dy and R_j = sum dy (x^ - j/L) of the elements in
// segment j plus B = sum dy y:   sum dy clip_i = (1/L) sum_{j>i} H_j + R_i.   The accumulators live in a PRIVATE
// LDS column per thread (slot * 257 + thread: conflict-free, no atomics), are reduced per block in a fixed order into
// a workspace record, and a finish launch adds an image's records in a fixed order: bit-reproducible like the
// tuned path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exposure_hip.h"
#include "host_common.h"

namespace expo {

constexpr int kCgThreads = 256;
constexpr int kCgMaxSteps = EXPO_CURVE_MAX_STEPS;  // 16: 3 curves x (2 x 16 + 1) private slots x 257 floats = 99.7 KB of LDS
constexpr int kCgPad = kCgThreads + 1;

template <typename T> __device__ __forceinline__ float cg_load(const T* p, size_t i) { return float(p[i]); }
template <typename T> __device__ __forceinline__ void cg_store(T* p, size_t i, float v) { p[i] = T(v); }
template <> __device__ __forceinline__ void cg_store<_Float16>(_Float16* p, size_t i, float v) {
  p[i] = _Float16(fminf(fmaxf(v, -65504.0f), 65504.0f));  // fp16 stores saturate, like the tuned kernels'
}

struct CgCurve {  // per-curve constants in LDS
  float k[kCgMaxSteps], prefix[kCgMaxSteps], scale, inv_s;
};

__device__ __forceinline__ void cg_stage(const float* __restrict__ prm, int curves, int L, CgCurve* cv) {
  if (threadIdx.x < unsigned(curves)) {
    const float* k = prm + threadIdx.x * L;
    CgCurve& c = cv[threadIdx.x];
    float s = 0.f;
    for (int i = 0; i < L; ++i) {
      c.k[i] = k[i];
      c.prefix[i] = s;
      s += k[i];
    }
    s += 1e-30f;
    c.scale = float(L) / s;
    c.inv_s = 1.0f / s;
  }
  __syncthreads();
}

// T(x) * L / S through the segment form; also returns the segment and the offset inside it
__device__ __forceinline__ float cg_eval(const CgCurve& c, int L, float x, int* seg, float* off) {
  const float xc = fminf(fmaxf(x, 0.0f), 1.0f);
  int j = int(xc * float(L));
  j = j > L - 1 ? L - 1 : j;
  const float o = xc - float(j) / float(L);
  *seg = j;
  *off = o;
  return (c.prefix[j] / float(L) + o * c.k[j]) * c.scale;
}

template <typename T>
__global__ __launch_bounds__(kCgThreads) void curve_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                const float* __restrict__ params, int hw, int curves, int L) {
  __shared__ CgCurve cv[3];
  const int n = blockIdx.y;
  cg_stage(params + size_t(n) * curves * L, curves, L, cv);
  const size_t base = size_t(n) * hw * 3;
  for (int p = blockIdx.x * kCgThreads + threadIdx.x; p < hw; p += gridDim.x * kCgThreads) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int j;
      float o;
      const float v = cg_eval(cv[curves == 1 ? 0 : c], L, cg_load(x, base + size_t(p) * 3 + c), &j, &o);
      cg_store(y, base + size_t(p) * 3 + c, v);
    }
  }
}

// record layout per block: [curve][H_0..H_{L-1} | R_0..R_{L-1} | B]  (curves * (2 L + 1) floats)
template <typename T, bool HAS_DX>
__global__ __launch_bounds__(kCgThreads) void curve_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                T* __restrict__ dx, const float* __restrict__ params,
                                                                float* __restrict__ records, int hw, int curves, int L) {
  __shared__ CgCurve cv[3];
  extern __shared__ float acc[];  // slots * kCgPad
  const int n = blockIdx.y;
  cg_stage(params + size_t(n) * curves * L, curves, L, cv);
  const int per_curve = 2 * L + 1, slots = curves * per_curve;
  for (int s = 0; s < slots; ++s) acc[s * kCgPad + threadIdx.x] = 0.f;
  const size_t base = size_t(n) * hw * 3;
  for (int p = blockIdx.x * kCgThreads + threadIdx.x; p < hw; p += gridDim.x * kCgThreads) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int cc = curves == 1 ? 0 : c;
      const CgCurve& cu = cv[cc];
      const float xv = cg_load(x, base + size_t(p) * 3 + c);
      const float g = cg_load(dy, base + size_t(p) * 3 + c);
      int j;
      float o;
      const float yv = cg_eval(cu, L, xv, &j, &o);
      float* a = acc + cc * per_curve * kCgPad + threadIdx.x;
      a[j * kCgPad] += g;
      a[(L + j) * kCgPad] = fmaf(g, o, a[(L + j) * kCgPad]);
      a[2 * L * kCgPad] = fmaf(g, yv, a[2 * L * kCgPad]);
      if constexpr (HAS_DX) {
        // tf.clip_by_value passes the gradient on 0 <= x - i/L <= 1/L: segment floor(u), plus segment u - 1 when
        // u = L x is an integer >= 1; nothing outside [0, 1]; -0.0 passes like +0.0; NaN gives 0
        const float u = xv * float(L);
        float sl = 0.f;
        if (u >= 0.0f && u <= float(L)) {
          const float jf = floorf(u);
          const int ju = int(jf);
          sl = ju < L ? cu.k[ju] : 0.0f;
          if (jf == u && ju >= 1) sl += cu.k[ju - 1];
        }
        cg_store(dx, base + size_t(p) * 3 + c, g * sl * cu.scale);
      }
    }
  }
  __syncthreads();
  // one thread per slot adds the 256 private columns in a fixed order
  float* rec = records + (size_t(n) * gridDim.x + blockIdx.x) * size_t(slots);
  for (int s = threadIdx.x; s < slots; s += kCgThreads) {
    float t = 0.f;
    for (int i = 0; i < kCgThreads; ++i) t += acc[s * kCgPad + i];
    rec[s] = t;
  }
}

// one block per image, one thread per parameter
__global__ __launch_bounds__(128) void curve_finish_kernel(const float* __restrict__ records, const float* __restrict__ params,
                                                           float* __restrict__ dparams, int blocks, int curves, int L) {
  const int n = blockIdx.x, per_curve = 2 * L + 1, slots = curves * per_curve;
  __shared__ float tot[3 * (2 * kCgMaxSteps + 1)];
  for (int s = threadIdx.x; s < slots; s += blockDim.x) {
    float t = 0.f;
    for (int b = 0; b < blocks; ++b) t += records[(size_t(n) * blocks + b) * size_t(slots) + s];
    tot[s] = t;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < curves * L; j += blockDim.x) {
    const int c = j / L, i = j % L;
    const float* k = params + size_t(n) * curves * L + c * L;
    float s = 0.f;
    for (int m = 0; m < L; ++m) s += k[m];
    s += 1e-30f;
    const float* t = tot + c * per_curve;
    float above = 0.f;  // sum_{j > i} H_j
    for (int m = L - 1; m > i; --m) above += t[m];
    const float clip_sum = above / float(L) + t[L + i];
    dparams[size_t(n) * curves * L + j] = (float(L) / s) * clip_sum - t[2 * L] / s;
  }
}

static int cg_blocks(int hw) {
  int b = (hw + kCgThreads * 8 - 1) / (kCgThreads * 8);  // ~8 pixels per thread
  return b < 1 ? 1 : (b > 256 ? 256 : b);
}

static int cg_check(int n, int h, int w, int dtype, int curves, int steps) {
  if (int rc = check_common(n, h, w, dtype)) return rc;
  if (curves != 1 && curves != 3) return fail(EXPO_E_BADARG, "curves must be 1 (ToneFilter) or 3 (ColorFilter)");
  if (steps < 1 || steps > kCgMaxSteps) return fail(EXPO_E_BADARG, "curve steps must be in [1, EXPO_CURVE_MAX_STEPS]");
  return EXPO_OK;
}

}  // namespace expo

using namespace expo;

extern "C" {

size_t expo_curve_workspace_bytes(int n, int h, int w, int curves, int steps) {
  if (n < 0 || h < 1 || w < 1 || (curves != 1 && curves != 3) || steps < 1 || steps > kCgMaxSteps) return 0;
  return size_t(n) * cg_blocks(h * w) * size_t(curves * (2 * steps + 1)) * sizeof(float);
}

int expo_curve_fwd(const void* x, void* y, const float* params, int n, int h, int w, int dtype, int curves, int steps,
                   void* stream) {
  if (int rc = cg_check(n, h, w, dtype, curves, steps)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !y || !params) return fail(EXPO_E_BADARG, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(cg_blocks(h * w), n), block(kCgThreads);
  if (dtype == EXPO_F16)
    hipLaunchKernelGGL(curve_fwd_kernel<_Float16>, grid, block, 0, s, (const _Float16*)x, (_Float16*)y, params, h * w, curves, steps);
  else
    hipLaunchKernelGGL(curve_fwd_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, params, h * w, curves, steps);
  HIP_TRY(hipGetLastError(), "curve_fwd launch");
  return EXPO_OK;
}

int expo_curve_bwd(const void* x, const void* dy, void* dx, const float* params, float* dparams, int n, int h, int w,
                   int dtype, int curves, int steps, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = cg_check(n, h, w, dtype, curves, steps)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !dy || !params || !dparams) return fail(EXPO_E_BADARG, "null pointer");
  if (!workspace || workspace_bytes < expo_curve_workspace_bytes(n, h, w, curves, steps))
    return fail(EXPO_E_BADARG, "workspace missing or too small (expo_curve_workspace_bytes)");
  if ((reinterpret_cast<uintptr_t>(workspace) & 3) != 0) return fail(EXPO_E_BADARG, "workspace must be 4-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = cg_blocks(h * w);
  const dim3 grid(blocks, n), block(kCgThreads);
  const size_t lds = size_t(curves * (2 * steps + 1)) * kCgPad * sizeof(float);
  float* rec = static_cast<float*>(workspace);
#define EXPO_CG(T, HAS_DX)                                                                                        \
  do {                                                                                                            \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&curve_bwd_kernel<T, HAS_DX>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)), "curve_bwd LDS size");      \
    hipLaunchKernelGGL((curve_bwd_kernel<T, HAS_DX>), grid, block, lds, s, (const T*)x, (const T*)dy, (T*)dx,     \
                       params, rec, h * w, curves, steps);                                                        \
  } while (0)
  if (dtype == EXPO_F16) { if (dx) EXPO_CG(_Float16, true); else EXPO_CG(_Float16, false); }
  else { if (dx) EXPO_CG(float, true); else EXPO_CG(float, false); }
#undef EXPO_CG
  HIP_TRY(hipGetLastError(), "curve_bwd launch");
  hipLaunchKernelGGL(curve_finish_kernel, dim3(n), dim3(128), 0, s, rec, params, dparams, blocks, curves, steps);
  HIP_TRY(hipGetLastError(), "curve_finish launch");
  return EXPO_OK;
}

}  // extern "C"
