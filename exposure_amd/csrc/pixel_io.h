// pixel_io.h -- NHWC pixel-group loads/stores for gfx950.
//
// A lane owns whole pixels (the coupled filters S+, Ct, BW need R,G,B of a pixel in one
// lane).  One "group" is 48 contiguous bytes = three 16-byte accesses per lane:
//   fp16: 8 pixels (24 halves)      fp32: 4 pixels (12 floats)
// so every global access is a dwordx4 and a wave covers 3 KiB contiguous per group-row.
// Images whose pixel count is not a multiple of the group size (or whose base is not
// 16-byte aligned) take the element-wise path (VEC = false), which is slow but exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace expo {

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <typename T> struct PixTraits;
template <> struct PixTraits<half_t> { static constexpr int PPL = 8; };
template <> struct PixTraits<float> { static constexpr int PPL = 4; };

template <bool NT>
__device__ __forceinline__ u32x4_t ld16(const u32x4_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st16(u32x4_t* p, u32x4_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// raw 48-byte group held in registers between the load issue and its first use
struct RawGroup { u32x4_t q[3]; };

template <bool NT>
__device__ __forceinline__ RawGroup load_raw(const void* img, int g) {
  const u32x4_t* p = reinterpret_cast<const u32x4_t*>(img) + size_t(g) * 3;
  RawGroup r;
  r.q[0] = ld16<NT>(p);
  r.q[1] = ld16<NT>(p + 1);
  r.q[2] = ld16<NT>(p + 2);
  return r;
}

template <typename T> __device__ __forceinline__ void unpack(const RawGroup& r, float* out);
template <> __device__ __forceinline__ void unpack<half_t>(const RawGroup& r, float* out) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const half8_t h = __builtin_bit_cast(half8_t, r.q[j]);
#pragma unroll
    for (int e = 0; e < 8; ++e) out[j * 8 + e] = float(h[e]);
  }
}
template <> __device__ __forceinline__ void unpack<float>(const RawGroup& r, float* out) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float4_t f = __builtin_bit_cast(float4_t, r.q[j]);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[j * 4 + e] = f[e];
  }
}

template <typename T> __device__ __forceinline__ RawGroup pack(const float* in);
template <> __device__ __forceinline__ RawGroup pack<half_t>(const float* in) {
  RawGroup r;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    half8_t h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = half_t(in[j * 8 + e]);  // round-to-nearest-even
    r.q[j] = __builtin_bit_cast(u32x4_t, h);
  }
  return r;
}
template <> __device__ __forceinline__ RawGroup pack<float>(const float* in) {
  RawGroup r;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float4_t f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = in[j * 4 + e];
    r.q[j] = __builtin_bit_cast(u32x4_t, f);
  }
  return r;
}

template <bool NT>
__device__ __forceinline__ void store_raw(void* img, int g, const RawGroup& r) {
  u32x4_t* p = reinterpret_cast<u32x4_t*>(img) + size_t(g) * 3;
  st16<NT>(p, r.q[0]);
  st16<NT>(p + 1, r.q[1]);
  st16<NT>(p + 2, r.q[2]);
}

// element-wise (ragged / unaligned) path: group g covers pixels [g*PPL, g*PPL+PPL) ∩ [0,hw)
template <typename T>
__device__ __forceinline__ void load_slow(const T* img, int g, int hw, float* out) {
  constexpr int PPL = PixTraits<T>::PPL;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int px = g * PPL + k;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[k * 3 + c] = (px < hw) ? float(img[size_t(px) * 3 + c]) : 0.0f;
  }
}
template <typename T>
__device__ __forceinline__ void store_slow(T* img, int g, int hw, const float* in) {
  constexpr int PPL = PixTraits<T>::PPL;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int px = g * PPL + k;
    if (px < hw) {
#pragma unroll
      for (int c = 0; c < 3; ++c) img[size_t(px) * 3 + c] = T(in[k * 3 + c]);
    }
  }
}

}  // namespace expo
