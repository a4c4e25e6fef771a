// host_common.h -- host-side helpers shared by the translation units of libexposure_hip.so:
// error reporting, launch geometry, argument checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <initializer_list>
#include <string>

#include "../../include/exposure_hip.h"
#include "kernel_common.h"

namespace expo {

extern thread_local std::string g_err;  // defined in exposure_hip.hip

inline int fail(int code, const char* what) {
  g_err = what;
  return code;
}
inline int fail_hip(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return EXPO_E_HIP;
}
#define HIP_TRY(expr, where)                         \
  do {                                               \
    hipError_t e_ = (expr);                          \
    if (e_ != hipSuccess) return fail_hip(e_, where); \
  } while (0)

constexpr int kNumParams[EXPO_NUM_FILTERS] = {1, 1, 3, 1, 8, 1, 1, 24, 2};

struct Geom {
  int hw, groups, blocks_x;
  bool vec;
  bool stream;  // IoStream policy (pixel_io.h): vector path and the tensor is far beyond L2
};

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  const int x = atoi(v);
  return x > 0 ? x : dflt;
}

// a reducing kernel writes one workspace record per block: bound the records of one image
constexpr int kMaxReduceBlocksX = 1024;
enum GeomKind {
  kGeomMap = 0,         // forward kernels
  kGeomReduce = 1,      // backward of the element-wise filters
  kGeomReadReduce = 2,  // read-only reductions (statistics, penalty)
  kGeomDispatch = 3,    // per-image dispatch backward
  kGeomReduceTone = 4,  // backward of Tone / Color: every block first stages its image's slope table, so fewer,
  kGeomReduceColor = 5, //   fatter blocks pay (Tone: 2 KB table, Color: 6 KB)
  kGeomApply = 6,       // masked apply backward
  kGeomFusedBwd = 7,    // one-pass backward of a whole sequence: per-block set-up of every step's tables, VALU-bound
  kNumGeomKinds = 8
};
// geometry of the plain backward of one filter (expo_filter_bwd / _records / expo_chain_bwd / expo_finish_bwd)
inline int bwd_geom_kind(int filter_id) { return filter_id == 4 ? kGeomReduceTone : filter_id == 7 ? kGeomReduceColor : kGeomReduce; }

// groups of 48 bytes per image; blocks per image chosen so the whole grid is >= ~1024
// blocks when the problem allows and each thread walks a few groups (amortises the
// reduction epilogue); capped at 2048-ish total blocks (grid-stride the rest).
template <typename T>
inline Geom make_geom(int n, int h, int w, std::initializer_list<const void*> ptrs, int kind) {
  const bool reduces = kind != kGeomMap;
  constexpr int PPL = PixTraits<T>::PPL;
  Geom g;
  g.hw = h * w;
  g.groups = (g.hw + PPL - 1) / PPL;
  g.vec = (g.hw % VecTraits<T>::PPV) == 0;  // dwordx3 path: whole 12-byte vectors, 4-byte aligned
  for (const void* p : ptrs) g.vec = g.vec && (p == nullptr || (reinterpret_cast<uintptr_t>(p) & 3) == 0);
  const int max_bx = (g.groups + kThreads - 1) / kThreads;
  // groups each thread walks (measured per kernel at 64x512x512, gpurun r02p35: the light backward kernels stream
  // best with ONE group per thread like the forward kernels -- 45.6-46.2 vs 47.4-48.3 us at four --, while the
  // curve kernels pay their per-block table staging: Tone 54.7 / 47.7 / 48.2 and Color 70.3 / 53.9 / 53.1 us
  // at 1 / 2 / 4 groups per thread)
  static const int gpt_kind[kNumGeomKinds] = {
      env_int("EXPO_FWD_GROUPS_PER_THREAD", 1),       env_int("EXPO_BWD_GROUPS_PER_THREAD", 1),
      env_int("EXPO_RED_GROUPS_PER_THREAD", 4),       env_int("EXPO_DISPATCH_GROUPS_PER_THREAD", 4),
      env_int("EXPO_TONE_GROUPS_PER_THREAD", 2),      env_int("EXPO_COLOR_GROUPS_PER_THREAD", 2),
      env_int("EXPO_APPLY_GROUPS_PER_THREAD", 4),     env_int("EXPO_FUSED_BWD_GROUPS_PER_THREAD", 8)};
  int gpt = gpt_kind[kind >= 0 && kind < kNumGeomKinds ? kind : kGeomReduce];
  // beyond the Infinity Cache (one tensor >= 256 MiB: HBM-cold streams, images walked in alternating order) the
  // light backward kernels are back to four groups per thread: 2.556 vs 2.595 ms per chain step at 256x512x512
  static const bool bwd_gpt_forced = getenv("EXPO_BWD_GROUPS_PER_THREAD") != nullptr;
  if (kind == kGeomReduce && !bwd_gpt_forced && long(n) * g.hw * 3L * long(sizeof(T)) >= (256L << 20) && gpt < 4) gpt = 4;
  int bx = (g.groups + kThreads * gpt - 1) / (kThreads * gpt);
  const long want = 1024;
  if (long(bx) * n < want) bx = int((want + n - 1) / n);
  if (bx > max_bx) bx = max_bx;
  if (reduces && bx > kMaxReduceBlocksX) bx = kMaxReduceBlocksX;
  if (bx < 1) bx = 1;
  g.blocks_x = bx;
  // cache policy: tensors of at least EXPO_STREAM_MIN_BYTES (default 8 MiB; L2 is 8 x 4 MiB) stream
  static const long stream_min = env_int("EXPO_STREAM_MIN_BYTES", 8 << 20);
  g.stream = g.vec && long(n) * g.hw * 3L * long(sizeof(T)) >= stream_min;
  return g;
}

// defined in exposure_hip.hip (the finish kernel and the workspace layout live there), used by chain_fused_bwd.hip:
// the per-step record slices of a chain's workspace, and the finish launch of expo_chain_fused_bwd
int chain_records(void* workspace, size_t workspace_bytes, int n, int h, int w, int dtype, int steps, float** records,
                  size_t* step_floats);
int finish_chain_fused(const int32_t* ids, const float* params, float* dparams, const float* records,
                       size_t step_floats, int steps, int n, int blocks_x, hipStream_t s);

inline int check_common(int n, int h, int w, int dtype) {
  if (n < 0 || h < 1 || w < 1) return fail(EXPO_E_BADARG, "n >= 0, h >= 1, w >= 1 required");
  if (n > 65535) return fail(EXPO_E_BADARG, "n > 65535 not supported (grid.y)");
  if (dtype != EXPO_F16 && dtype != EXPO_F32) return fail(EXPO_E_BADDTYPE, "dtype must be EXPO_F16 or EXPO_F32");
  // one image is addressed through a raw buffer resource with 32-bit byte offsets
  const long image_bytes = long(h) * long(w) * 3L * (dtype == EXPO_F16 ? 2L : 4L);
  if (image_bytes > (1L << 31) - 8192) return fail(EXPO_E_BADARG, "one image must be smaller than 2 GiB");
  return EXPO_OK;
}

}  // namespace expo
