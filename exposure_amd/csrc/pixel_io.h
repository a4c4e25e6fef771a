// pixel_io.h -- NHWC pixel-group loads/stores for gfx950.
//
// A lane owns whole pixels (the coupled filters S+, Ct, BW need R,G,B of a pixel in one
// lane).  One "group" is the work of one thread iteration: 48 bytes =
//   fp16: 8 pixels (24 halves)      fp32: 4 pixels (12 floats)
// and a wave iteration covers 3 KiB of contiguous image.  Images the vector path cannot take
// (odd fp16 pixel counts, unaligned bases) use the element-wise path (VEC = false): slow, exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace expo {

typedef _Float16 half_t;

template <typename T> struct PixTraits;
template <> struct PixTraits<half_t> { static constexpr int PPL = 8; };
template <> struct PixTraits<float> { static constexpr int PPL = 4; };

// ---------------------------------------------------------------------------------------
// Vector path: dwordx3 buffer accesses.  12 bytes = 2 fp16 pixels = 1 fp32 pixel, so lane i of a
// wave touches bytes [12 i, 12 i + 12) of a 768-byte row: every access is fully coalesced
// (consecutive lanes -> consecutive addresses) AND pixel-aligned (a lane owns whole pixels, which
// the coupled filters S+ / Ct / BW need) with no LDS transpose.  A thread iteration is four such
// rows (8 fp16 / 4 fp32 pixels per lane, 3 KiB per wave); the four rows share ONE 32-bit byte
// offset VGPR and differ only in the instruction's immediate offset (0/768/1536/2304).
//
// Each image is addressed through a raw buffer resource (SRD in SGPRs, num_records = image bytes):
// the hardware bounds check returns 0 for loads and drops stores past the end of the image, so the
// image's partial last chunk runs the same branch-free code as every other chunk.
//
// (The first design read 48 contiguous bytes per lane as three global_load_dwordx4: its 48-byte
// lane stride touches every cache line from three instructions and measured 15-20 % below the
// coalesced pattern -- tools/membench.hip copy48 vs copy12/copy16, profiles/r01_*_membench.txt.)
// Requirements: fp16: H*W even and a 4-byte aligned base; fp32: always (4-byte aligned).
// ---------------------------------------------------------------------------------------
// fp16 stores saturate at +-65504 through MODE.FP16_OVFL (set once per wave by stream_groups)
// instead of 24 packed min/max per pixel group; 0 = explicit clamps.
#ifndef EXPO_FP16_OVFL
#define EXPO_FP16_OVFL 1
#endif

typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));

template <typename T> struct VecTraits;
template <> struct VecTraits<half_t> { static constexpr int PPV = 2; };  // pixels per 12-byte vector
template <> struct VecTraits<float> { static constexpr int PPV = 1; };

struct RawGroup { u32x3_t q[4]; };

// gfx9-family raw buffer descriptor word 3 (DATA_FORMAT = 32-bit); stride 0 = raw, bounds-checked
constexpr int kBufferRsrcFlags = 0x00020000;

template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_image_rsrc(const T* img, int hw) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(img), 0, hw * 3 * int(sizeof(T)), kBufferRsrcFlags);
}

// byte offset of this lane's first vector in the wave chunk that starts at group gw
template <typename T>
__device__ __forceinline__ int chunk_byte_offset(int gw, int lane) {
  constexpr int PPL = PixTraits<T>::PPL, PPV = VecTraits<T>::PPV;
  return (gw * (PPL / PPV) + lane) * 12;
}

// Cache policy of the image streams (buffer-instruction aux bits on gfx940+: 1 = sc0, 2 = nt, 16 = sc1).
// IoCached: default policy everywhere -- right while a tensor fits the 32 MiB of L2 (training on
//   64x64 proxies: the next kernel, ours or a MIOpen conv, finds the lines in L2).
// IoStream: nt loads (stream through L2, evict first) + sc1 stores (write through, line dropped from
//   L2) for tensors far beyond L2.  Measured on MI355X (tools/membench.hip, 96 MiB buffers, dwordx3
//   SRD skeleton, launch i+1 reads what launch i wrote):
//     copy  plain/plain 6.26 TB/s | nt loads 7.44 | sc1 stores 6.59 | nt loads + sc1 stores 7.73
//     read-read-write  plain 6.28 TB/s | nt x 7.21 | nt x, nt dy, sc1 store 7.61
//     (nt STORES lose: 6.13; sc1|nt stores 6.17; sc0|sc1 stores = sc1.)
//   8-step chain 64x512x512x3 fp16: 0.668-0.690 ms/step cached -> 0.616 ms/step streaming, and the
//   backward "hangover" of the cached policy (the forward's dirty lines draining from the 256 MiB
//   Infinity Cache under the first two backward kernels, +10..20 us) disappears.
struct IoCached { static constexpr int kLoadX = 0, kLoadDy = 0, kStore = 0; };
struct IoStream { static constexpr int kLoadX = 2, kLoadDy = 2, kStore = 16; };

template <int AUX>
__device__ __forceinline__ RawGroup load_raw(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  RawGroup r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.q[j] = __builtin_amdgcn_raw_buffer_load_b96(rsrc, byte_off + j * 768, 0, AUX);
  return r;
}

template <int AUX>
__device__ __forceinline__ void store_raw(__amdgpu_buffer_rsrc_t rsrc, int byte_off, const RawGroup& r) {
#pragma unroll
  for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b96(r.q[j], rsrc, byte_off + j * 768, 0, AUX);
}

template <typename T> __device__ __forceinline__ void unpack(const RawGroup& r, float* out);
template <> __device__ __forceinline__ void unpack<half_t>(const RawGroup& r, float* out) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      // NB: bit_cast of a vector-element lvalue reads element 0 (clang quirk) -> copy to a scalar
      const uint32_t w = r.q[j][e];
      const half2_t h = __builtin_bit_cast(half2_t, w);
      out[j * 6 + e * 2] = float(h[0]);
      out[j * 6 + e * 2 + 1] = float(h[1]);
    }
  }
}
template <> __device__ __forceinline__ void unpack<float>(const RawGroup& r, float* out) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const uint32_t w = r.q[j][e];
      out[j * 3 + e] = __builtin_bit_cast(float, w);
    }
  }
}
template <typename T> __device__ __forceinline__ RawGroup pack(const float* in);
template <> __device__ __forceinline__ RawGroup pack<half_t>(const float* in) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  RawGroup r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      // round-to-nearest-even, SATURATING at +-65504: five +3.5 EV exposure steps of an untrained
      // policy exceed the fp16 range, and an inf pixel would poison every later loss (the fp32
      // reference just carries the large value; fp16 storage carries the largest finite one).
      // With EXPO_FP16_OVFL the conversion itself saturates (MODE.FP16_OVFL; a true +-inf input
      // stays inf), otherwise an explicit packed clamp does (+-inf -> +-65504 too).
      half2_t h;
      h[0] = half_t(in[j * 6 + e * 2]);
      h[1] = half_t(in[j * 6 + e * 2 + 1]);
#if !EXPO_FP16_OVFL
      const half2_t hi = {half_t(65504.0f), half_t(65504.0f)}, lo = {half_t(-65504.0f), half_t(-65504.0f)};
      h = __builtin_elementwise_max(__builtin_elementwise_min(h, hi), lo);  // +-inf -> +-65504 (packed)
#endif
      r.q[j][e] = __builtin_bit_cast(uint32_t, h);
    }
  }
  return r;
}
template <> __device__ __forceinline__ RawGroup pack<float>(const float* in) {
  RawGroup r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int e = 0; e < 3; ++e) r.q[j][e] = __builtin_bit_cast(uint32_t, in[j * 3 + e]);
  }
  return r;
}

// Streaming driver of the vector path.  Each wave walks 3 KiB chunks of one image with a block
// stride; NIN input streams (x, or x and dy) are unpacked to fp32, fn(v, g) transforms them in
// place, and the LAST stream is written back when HAS_OUT.  With PF the next chunk's loads are in
// flight while the current one computes (software prefetch; costs 12 VGPRs per stream).
// Stream 0 is the image x (policy IO::kLoadX), stream 1 the upstream gradient dy (IO::kLoadDy).
// `pre` runs once per wave AFTER the first chunk's loads are issued and before anything waits on
// them: block-wide set-up that must not delay the first loads (the curve backward stages its LDS
// slope table there, including the __syncthreads -- every wave calls it, also one without work).
struct NoPrologue { __device__ void operator()() const {} };
template <typename T, int NIN, bool HAS_OUT, bool PF, class IO, class Fn, class Pre = NoPrologue>
__device__ __forceinline__ void stream_groups(const T* const (&in)[NIN], T* out, int hw, int first_gw,
                                              int stride, Fn&& fn, Pre&& pre = Pre()) {
  constexpr int PPL = PixTraits<T>::PPL;
  const int lane = threadIdx.x & 63;
#if EXPO_FP16_OVFL
  // MODE.FP16_OVFL (hwreg 1, bit 23): fp16 results that overflow clamp to +-65504 instead of +-inf
  if constexpr (HAS_OUT && sizeof(T) == 2) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
#endif
  __amdgpu_buffer_rsrc_t rin[NIN];
#pragma unroll
  for (int s = 0; s < NIN; ++s) rin[s] = make_image_rsrc(in[s], hw);
  __amdgpu_buffer_rsrc_t rout = rin[0];
  if constexpr (HAS_OUT) rout = make_image_rsrc(out, hw);
  int gw = first_gw;  // wave-uniform
  const bool work = gw * PPL < hw;
  static_assert(NIN <= 2, "at most two input streams");
  auto load_all = [&](RawGroup (&dst)[NIN], int g) {
    dst[0] = load_raw<IO::kLoadX>(rin[0], chunk_byte_offset<T>(g, lane));
    if constexpr (NIN > 1) dst[1] = load_raw<IO::kLoadDy>(rin[1], chunk_byte_offset<T>(g, lane));
  };
  RawGroup cur[NIN];
  if (work) load_all(cur, gw);
  pre();
  if (!work) return;
  // (A two-chunks-per-trip variant whose buffers swap roles instead of the "cur = nxt" copy measured 1 % slower
  // on the chain and cost ~30 VGPRs on the light backward kernels; gpurun r02p5.)
  while (true) {
    const int gn = gw + stride;
    const bool more = gn * PPL < hw;  // wave-uniform
    RawGroup nxt[NIN];
#pragma unroll
    for (int s = 0; s < NIN; ++s) nxt[s] = cur[s];
    if (PF && more) load_all(nxt, gn);
    float v[NIN][PPL * 3];
#pragma unroll
    for (int s = 0; s < NIN; ++s) unpack<T>(cur[s], v[s]);
    fn(v, gw + lane);
    if constexpr (HAS_OUT) store_raw<IO::kStore>(rout, chunk_byte_offset<T>(gw, lane), pack<T>(v[NIN - 1]));
    if (!PF && more) load_all(nxt, gn);
    if (!more) break;
    gw = gn;
#pragma unroll
    for (int s = 0; s < NIN; ++s) cur[s] = nxt[s];
  }
}

// Is pixel slot k of the group a real pixel?  (group index g = wave base + lane)
template <typename T, bool VEC>
__device__ __forceinline__ bool live_pixel(int g, int k, int lane, int hw) {
  constexpr int PPL = PixTraits<T>::PPL;
  if constexpr (VEC) {
    constexpr int PPV = VecTraits<T>::PPV;
    const int gw = g - lane;
    return gw * PPL + (k / PPV) * 64 * PPV + lane * PPV + (k % PPV) < hw;
  }
  return g * PPL + k < hw;
}

// Linear pixel index (within the image) of pixel slot k of group g.
template <typename T, bool VEC>
__device__ __forceinline__ int pixel_index(int g, int k, int lane) {
  constexpr int PPL = PixTraits<T>::PPL;
  if constexpr (VEC) {
    constexpr int PPV = VecTraits<T>::PPV;
    return (g - lane) * PPL + (k / PPV) * 64 * PPV + lane * PPV + (k % PPV);
  }
  return g * PPL + k;
}

// Inside a group the linear pixel index of slot k follows slot k - 1 by one of two steps (PixelWalk, filter_math.h):
// step a = 1 (the second pixel of a 12-byte vector; every slot of the element-wise path), step b = the jump to the next
// 768-byte row of the chunk.
template <typename T, bool VEC> constexpr int pixel_step_a() { return 1; }
template <typename T, bool VEC> constexpr int pixel_step_b() {
  return VEC ? 64 * VecTraits<T>::PPV - (VecTraits<T>::PPV - 1) : 1;
}
template <typename T, bool VEC> constexpr bool pixel_step_is_b(int k) { return VEC && (k % VecTraits<T>::PPV) == 0; }

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
      for (int c = 0; c < 3; ++c) {
        float v = in[k * 3 + c];
        if (sizeof(T) == 2) v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);  // saturate fp16 (see pack)
        img[size_t(px) * 3 + c] = T(v);
      }
    }
  }
}

}  // namespace expo
