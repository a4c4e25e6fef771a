// conv_ops.hip -- the one convolution of the convnets around the filter path (feature_extractor agent.py:21-32, cnn
// critics.py:13-35: `ly.conv2d(kernel_size=4, stride=2)` SAME, NHWC float32) as hand-written implicit-GEMM kernels on
// the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 157 TFLOP/s chip peak), with the layer's bias + lrelu
// (util.py:225-229) in the epilogue.
//
// Why in-house: the layers are SMALL -- batch 64 / 128 of 64x64 proxies give GEMMs of ~1 GFLOP (M x Cout x K =
// 65536 x 32 x 224 ... 1024 x 256 x 2048) -- and the library kernels MIOpen picks for them (asm igemm / CK grouped
// xdlops, tuned for large problems) take 17-46 us each, 0.2-0.5 of the f32 matrix peak, plus a zero-fill launch in
// front of every split-K kernel and a bias / activation launch behind every forward (profiles/r04_final_kernel_
// stats_train.csv: 228 convolution launches = 5.1 ms of a 9.5 ms training iteration, 191 fills = 1.0 ms).  Decompositions
// chosen for THESE sizes run the forward (+ bias + lrelu, one launch) 1.2-1.7x and the deeper layers' data gradient
// 1.2-1.6x faster (45-80 TFLOP/s; DESIGN.md 3.11 says what bounds them).  In this file, in order: the LDS-tiled forward
// (conv_fwd_kernel, four shapes), the flat forward (conv_fwd_flat_kernel: the default except where 64 x 64 tiles still
// give every CU a block or two), the data gradient as four parity-class GEMMs (conv_bwd_flat_kernel), the weight
// gradient (conv_wrw_kernel + its reduce launch: parity-green and deterministic, on par with MIOpen at best -> opt-in).
//
// Forward as a GEMM:  Y[m][co] = sum_k A[m][k] W[co][k],  m = (n, oh, ow),  k = (kh, kw, ci),  K = 16 Cin
//   A[m][k] = X[n][2 oh - 1 + kh][2 ow - 1 + kw][ci]  (0 outside the image)
//   * for a fixed kh the (kw, ci) range is 4 Cin CONTIGUOUS floats of X (NHWC: w and c adjacent), so a K-chunk of 4
//     floats is one 16-byte load whatever Cin is (14, 6, 17 for the first layers);
//   * W is the nn.Conv2d weight (Cout, Cin, 4, 4) in channels_last memory order = [co][kh][kw][ci] = [co][k].
// Block tile BM x BN, K step 32 per wave group; WM x WN x WK waves: WM x WN tile the block tile, WK wave groups split
// every K step between them (intra-block split-K, reduced through LDS at the end -- no atomics, no zero fill, a fixed
// summation order).  Operand tiles go global -> registers -> LDS (K-minor rows padded by 4 floats: conflict-free
// ds_read_b128 / ds_write_b128), one K step ahead of the MFMAs; each lane reads its A / B fragment as ONE b128 per 8
// k: lane l holds row l & 31 and k = 8c + 4 (l >> 5) + j for the j-th MFMA of the chunk (any pairing of k between the
// two lane halves is a valid order of the K sum as long as A and B use the same one).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/exposure_hip.h"
#include "host_common.h"

namespace expo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };  // a 4-float chunk that is only dword-aligned
struct __attribute__((packed, aligned(4))) F3U { float v[3]; };  // a pixel's three image planes

struct ConvDims {
  int n, h, w, cin, cout;  // h, w: the LARGER spatial size (forward input / data-gradient output), even
  int ho, wo;              // h / 2, w / 2
  int kdim;                // 16 cin
  int m;                   // n ho wo
  int sh_w, sh_hw;         // log2 wo, log2 (ho wo) when both are powers of two, else -1 (pixel index -> (n, oh, ow) by shifts)
};
// A PAIR of launches as one grid (gridDim.y = 2): two problems of the same geometry and plan -- the agent's two feature
// extractors, the critic's and the value net's pair passes of a G / V step -- whose layers are independent and at batch 64 /
// 128 too small to fill the chip alone.  Blocks with blockIdx.y = 1 take the second problem's pointers.
struct ConvSecond {
  const float* x;      // forward: input; data gradient: dY
  const float* w;
  const float* bias;   // forward only
  const float* zmask;
  float* y;            // forward: output; data gradient: dX
};

// pixel index m = (n ho + a) wo + b  ->  (n, a, b)
__device__ __forceinline__ void split_pixel(const ConvDims& d, int m, int& n, int& a, int& b) {
  if (d.sh_w >= 0) {  // (wave-uniform)
    n = m >> d.sh_hw;
    const int rem = m & ((1 << d.sh_hw) - 1);
    a = rem >> d.sh_w;
    b = rem & ((1 << d.sh_w) - 1);
  } else {
    n = m / (d.ho * d.wo);
    const int rem = m - n * (d.ho * d.wo);
    a = rem / d.wo;
    b = rem - a * d.wo;
  }
}

__device__ __forceinline__ float lrelu_v(float v, float leak) { return v > 0.f ? v : v * leak; }
// d lrelu / d (its argument), read from the OUTPUT z (nn_ops.hip: lrelu keeps sign and zero; TF's sub-gradient at 0)
__device__ __forceinline__ float lrelu_slope_v(float z, float leak) {
  return z > 0.f ? 1.0f : (z < 0.f ? leak : 0.5f * (1.0f + leak));
}

constexpr int kConvPad = 4;  // floats of row padding in LDS

// ---- forward ---------------------------------------------------------------------------------------------------
// One 4-float K-chunk of the implicit im2col matrix: row (n, oh, ow) at k..k+3.
struct FwdRow {
  int img;       // n H W Cin: element offset of the image
  int ih0, c0;   // 2 oh - 1;  (2 ow - 1) Cin  (may be negative)
  bool ok;
};
__device__ __forceinline__ FwdRow fwd_row(const ConvDims& d, int m) {
  FwdRow r;
  r.ok = m < d.m;
  const int mm = r.ok ? m : 0;
  const int n = mm / (d.ho * d.wo), rem = mm - n * (d.ho * d.wo);
  const int oh = rem / d.wo, ow = rem - oh * d.wo;
  r.img = n * d.h * d.w * d.cin;
  r.ih0 = 2 * oh - 1;
  r.c0 = (2 * ow - 1) * d.cin;
  return r;
}
// Branch-free: a load sits in straight-line code behind a clamped address and its result is masked afterwards.  (The
// first version put the image-edge cases in branches; the compiler closes such a branch with s_waitcnt vmcnt(0), every
// 64-pixel tile has edge pixels, so every wave waited out its loads right after issuing them: 3.7x the MFMA time.)
// CIN4: Cin % 4 == 0 -- a chunk never straddles a pixel, so it is inside the image row or outside it as a whole (one
// 16-byte load); otherwise the four elements are loaded and masked one by one (the first layers: K is small there).
// Loads go through raw buffer resources (SRD over the whole tensor): an element outside the image gets a byte offset
// beyond the buffer, for which the hardware returns 0 without touching memory -- no select BEHIND the load either (a
// v_cndmask on the loaded value is a use, and the compiler waits for the load right there; a pointer select between the
// tensor and a block of zeros turns the loads into flat_load, which must be waited out before any LDS access).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kConvRsrcFlags = 0x00020000;  // gfx9-family raw buffer descriptor word 3 (DATA_FORMAT = 32-bit); stride 0
constexpr int kConvOob = -16;               // 0xFFFFFFF0 as a byte offset: beyond any buffer this code accepts (< 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const float* p, size_t floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, int(floats * 4), kConvRsrcFlags);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, int elem_off, bool ok) {
  const u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? elem_off * 4 : kConvOob, 0, 0);
  // (bit_cast of a vector-element lvalue reads element 0 -- clang quirk, pixel_io.h: copy to scalars first)
  const uint32_t u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
  return make_float4(__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1), __builtin_bit_cast(float, u2),
                     __builtin_bit_cast(float, u3));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int elem_off, bool ok) {
  const uint32_t u = __builtin_amdgcn_raw_buffer_load_b32(r, ok ? elem_off * 4 : kConvOob, 0, 0);
  return __builtin_bit_cast(float, u);
}
// (stores likewise: an offset beyond the buffer drops the element -- epilogues without a branch per element)
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, int elem_off, bool ok, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, ok ? elem_off * 4 : kConvOob, 0, 0);
}
// A 4-float chunk of an image row: row_off = element offset of the row's start, c = offset inside the row (may be < 0).
// Cin % 4 == 0 (CIN4): a chunk never straddles a pixel, so it lies inside the row or outside it as a whole -- one
// 16-byte load.  Otherwise (the first layers: 14, 6, 17 planes) the image's left / right edge can cut THROUGH a chunk:
// the 16-byte load serves the chunks that lie inside as a whole, four 4-byte loads serve the elements of the cut ones
// (out of range -- free -- in every other lane), and the two are added where the chunk is CONSUMED (one of them is zero),
// not behind the loads.  (Element-wise loads for every chunk cost 4x the vector-L1 line look-ups: 27 us for the first
// generator layer against 17 for the deeper ones with the same FLOPs.)
template <bool CIN4> struct Chunk;
template <> struct Chunk<true> {
  float4 v;
  __device__ __forceinline__ float4 get() const { return v; }
};
template <> struct Chunk<false> {
  float4 v, p;
  __device__ __forceinline__ float4 get() const { return make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w); }
};
template <bool CIN4>
__device__ __forceinline__ Chunk<CIN4> row_chunk(__amdgpu_buffer_rsrc_t rx, int row_off, int c, int lim, bool ok) {
  Chunk<CIN4> r;
  const bool whole = ok && c >= 0 && c + 3 < lim;
  r.v = buf_load4(rx, row_off + c, whole);
  if constexpr (!CIN4) {
    const bool cut = ok && !whole;
    r.p = make_float4(buf_load1(rx, row_off + c + 0, cut && c + 0 >= 0 && c + 0 < lim),
                      buf_load1(rx, row_off + c + 1, cut && c + 1 >= 0 && c + 1 < lim),
                      buf_load1(rx, row_off + c + 2, cut && c + 2 >= 0 && c + 2 < lim),
                      buf_load1(rx, row_off + c + 3, cut && c + 3 >= 0 && c + 3 < lim));
  }
  return r;
}
template <bool CIN4>
__device__ __forceinline__ Chunk<CIN4> fwd_load_a(__amdgpu_buffer_rsrc_t rx, const FwdRow& r, const ConvDims& d, int k) {
  const int rr = 4 * d.cin;
  const int kh = (k >= rr) + (k >= 2 * rr) + (k >= 3 * rr);
  const int ih = r.ih0 + kh;
  const bool ok = r.ok && k < d.kdim && unsigned(ih) < unsigned(d.h);
  const int lim = d.w * d.cin;  // valid offsets inside an image row: [0, lim)
  return row_chunk<CIN4>(rx, r.img + ih * lim, r.c0 + (k - kh * rr), lim, ok);
}

template <int BM, int BN, int WM, int WN, int WK, int BKS, bool CIN4>
__global__ __launch_bounds__(64 * WM * WN * WK) void conv_fwd_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ w,
                                                                      const float* __restrict__ bias,
                                                                      const float* zmask, float* y, ConvDims d,
                                                                      int act, float leak, ConvSecond sec) {
  if (blockIdx.y) { x = sec.x; w = sec.w; bias = sec.bias; zmask = sec.zmask; y = sec.y; }  // (the second problem of a pair)
  // zmask (nullable, may alias y): the result is multiplied by the lrelu slope read from zmask at the same position --
  // the tangent pass of the gradient penalty's double backward, t_l = F(t_{l-1}, W_l) * slope(z_l), written over z_l
  constexpr int T = 64 * WM * WN * WK;
  static_assert(BKS % 8 == 0, "a wave group's K slice is consumed in chunks of 8");
  constexpr int BKT = BKS * WK;         // k per block step (BKS per wave group)
  constexpr int LDK = BKT + kConvPad;   // LDS row length
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tiles are multiples of the 32x32 MFMA tile");
  constexpr int QK = BKT / 4;            // float4 chunks per tile row
  static_assert((BM * QK) % T == 0 && (BN * QK) % T == 0, "every thread moves the same number of chunks");
  constexpr int NA = BM * QK / T, NB = BN * QK / T;  // chunks per thread
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDK];
  float* const As = lds;
  float* const Bs = lds + 2 * BM * LDK;

  // XCD-aware tile order: consecutive block ids go round-robin over the 8 XCDs; blocks that share an M tile (the
  // same image rows, all N tiles) get the same XCD so that its L2 serves the re-reads
  const int tiles_n = (d.cout + BN - 1) / BN, tiles_m = (d.m + BM - 1) / BM;
  const int nblocks = tiles_m * tiles_n;
  int bid = blockIdx.x;
  if (nblocks % 8 == 0) bid = (bid % 8) * (nblocks / 8) + bid / 8;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WM * WN), wmn = wave - wk * (WM * WN), wm = wmn / WN, wn = wmn - wm * WN;

  // per-thread chunk slots
  FwdRow arow[NA];
  int a_r[NA], a_q[NA], b_r[NB], b_q[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int f = tid + j * T;
    a_r[j] = f / QK;
    a_q[j] = f - a_r[j] * QK;
    arow[j] = fwd_row(d, m0 + a_r[j]);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int f = tid + j * T;
    b_r[j] = f / QK;
    b_q[j] = f - b_r[j] * QK;
  }
  const __amdgpu_buffer_rsrc_t rx = conv_rsrc(x, size_t(d.n) * d.h * d.w * d.cin);
  const __amdgpu_buffer_rsrc_t rw = conv_rsrc(w, size_t(d.cout) * d.kdim);
  Chunk<CIN4> ra[NA];
  float4 rb[NB];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = fwd_load_a<CIN4>(rx, arow[j], d, k0 + 4 * a_q[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int co = n0 + b_r[j], k = k0 + 4 * b_q[j];
      const bool ok = co < d.cout && k < d.kdim;
      rb[j] = buf_load4(rw, co * d.kdim + k, ok);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA; ++j)
      *reinterpret_cast<float4*>(As + (buf * BM + a_r[j]) * LDK + 4 * a_q[j]) = ra[j].get();
#pragma unroll
    for (int j = 0; j < NB; ++j)
      *reinterpret_cast<float4*>(Bs + (buf * BN + b_r[j]) * LDK + 4 * b_q[j]) = rb[j];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int steps = (d.kdim + BKT - 1) / BKT;
  load_tiles(0);
  store_tiles(0);
  load_tiles(BKT);
  __syncthreads();
  const int lrow = lane & 31, lk = (lane >> 5) * 4;
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    // (unconditional -- one basic block per step: chunks past K come back as zeros without touching memory, and the
    // last step's store lands in the buffer nobody reads again)
    store_tiles(buf ^ 1);            // data of step s + 1 (loaded during step s - 1's MFMAs)
    load_tiles((s + 2) * BKT);       // lands while this step computes
    __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler sinks these loads behind the step's MFMAs, where
                                        // nothing hides their latency from the next step's ds_write)
    const float* a_base = As + (buf * BM + wm * TM + lrow) * LDK + wk * BKS + lk;
    const float* b_base = Bs + (buf * BN + wn * TN + lrow) * LDK + wk * BKS + lk;
#pragma unroll
    for (int c = 0; c < BKS / 8; ++c) {
      float4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * LDK + c * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const float4*>(b_base + j * 32 * LDK + c * 8);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // ---- epilogue: sum the WK partial tiles (LDS, fixed order), bias + lrelu, store NHWC ----
  // C / D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  if constexpr (WK > 1) {
    // partial tiles of groups 1 .. WK-1 -> LDS [wk - 1][wmn][MI][NI][16][64]; group 0 adds them in order
    constexpr int per_wave = MI * NI * 16 * 64;
    static_assert((WK - 1) * WM * WN * per_wave <= 2 * (BM + BN) * LDK, "partials fit the operand buffers");
    if (wk > 0) {
      float* dst = lds + ((wk - 1) * WM * WN + wmn) * per_wave;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((i * NI + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int g = 1; g < WK; ++g) {
      const float* src = lds + ((g - 1) * WM * WN + wmn) * per_wave;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += src[((i * NI + j) * 16 + e) * 64 + lane];
    }
  }
  // (straight-line: slope-mask values through buffer loads issued sixteen at a time, results through buffer stores --
  // lanes outside the tensor touch nothing; see conv_fwd_flat_kernel)
  const __amdgpu_buffer_rsrc_t ry = conv_rsrc(y, size_t(d.m) * d.cout);
  const __amdgpu_buffer_rsrc_t rz = conv_rsrc(zmask ? zmask : y, size_t(d.m) * d.cout);
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + wn * TN + j * 32 + (lane & 31);
    const float b = (bias && co < d.cout) ? bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float z[16];
      int off[16];
      bool ok[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        ok[e] = m < d.m && co < d.cout;
        off[e] = m * d.cout + co;
        z[e] = buf_load1(rz, off[e], zmask != nullptr && ok[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[i][j][e] + b;
        if (act) v = lrelu_v(v, leak);
        if (zmask) v *= lrelu_slope_v(z[e], leak);
        buf_store1(ry, off[e], ok[e], v);
      }
    }
  }
}

// ---- forward, flat decomposition (v2) ------------------------------------------------------------------------------
// One WAVE = one 32 x 32 output tile over one K slice; no LDS staging, no barrier in the main loop: every lane loads its
// own A / B fragments straight from global memory (16 bytes per lane and operand per 8 k: lane l -> row / column l & 31,
// k = 8c + 4 (l >> 5) .. + 3; four consecutive chunks walk one 128-byte line per row, so the vector L1 sees every line
// once), and 4-8 waves per SIMD hide each other's load latency.  A block = the S K-slices (x NT column tiles sharing the
// A rows through the L1) of one tile; the S partial tiles meet in LDS once, at the end, and are summed in slice order.
// K slices are cut along the (kh, r) structure of K = 4 rows x R floats (R = 4 Cin): G = 4 S2 segments (kh, j) of
// RS = roundup8(R / S2) floats, G / S consecutive segments per slice.
struct FlatPlan {
  int s;     // K slices (waves per tile), divides g
  int s2;    // segments per kh row
  int rs;    // floats per segment (multiple of 8)
  int tiles_m, tiles_n;
};

template <int NI, bool CIN4>
__global__ __launch_bounds__(1024) void conv_fwd_flat_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* zmask,
                                                             float* y, ConvDims d, FlatPlan pl, int act, float leak,
                                                             ConvSecond sec) {
  if (blockIdx.y) { x = sec.x; w = sec.w; bias = sec.bias; zmask = sec.zmask; y = sec.y; }  // (the second problem of a pair)
  // NI = column tiles per WAVE (32 x 32 NI outputs): the A fragment is loaded once and feeds NI MFMAs -- 3 operand
  // loads per 8 MFMAs instead of 2 per 4 for NI = 2
  extern __shared__ __attribute__((aligned(16))) float part[];  // [s][NI][16][64]
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;  // wave = K slice
  const int nblocks = pl.tiles_m * pl.tiles_n;
  int bid = blockIdx.x;
  if (nblocks % 8 == 0) bid = (bid % 8) * (nblocks / 8) + bid / 8;  // neighbouring tiles on one XCD (shared L2)
  const int tm = bid / pl.tiles_n, tn = bid - tm * pl.tiles_n;
  const int row = lane & 31, half = lane >> 5;
  const int m = tm * 32 + row;  // this lane's A row (output pixel); its B rows (output channels): co[ni]
  const int rr = 4 * d.cin, lim = d.w * d.cin;
  const bool m_ok = m < d.m;
  int n, oh, ow;
  split_pixel(d, m_ok ? m : 0, n, oh, ow);
  const int img = n * d.h * lim;
  const int ih0 = 2 * oh - 1, c0 = (2 * ow - 1) * d.cin + 4 * half;
  int wrow[NI];
  bool co_ok[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int co = (tn * NI + i) * 32 + row;
    co_ok[i] = co < d.cout;
    wrow[i] = (co_ok[i] ? co : 0) * d.kdim + 4 * half;
  }
  const __amdgpu_buffer_rsrc_t rx = conv_rsrc(x, size_t(d.n) * d.h * d.w * d.cin);
  const __amdgpu_buffer_rsrc_t rw = conv_rsrc(w, size_t(d.cout) * d.kdim);

  f32x16 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  auto mma = [&](const float4& a, const float4 (&b)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[i].x, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[i].y, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[i].z, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[i].w, acc[i], 0, 0, 0);
    }
  };

  // ONE software pipeline over all the wave's segments: groups of U chunks of 8 k, the next group's loads in flight while
  // this group's MFMAs issue -- across segment boundaries too.  (Until round 6 every segment started its own pipeline: a
  // first layer's wave walks 4 segments of 3 chunks and waited out a full memory latency four times over -- most of its
  // 35 us at batch 192.)  Loads past a segment's end, and the group past the last one, are masked: they come back as zeros
  // without touching memory.
  constexpr int U = (CIN4 && NI == 1) ? 4 : 2;  // (cut-chunk fix-ups / a second column tile double the registers)
  const int g = 4 * pl.s2, per = g / pl.s;
  const int gps = (pl.rs + 8 * U - 1) / (8 * U);  // groups per segment
  const int total = per * gps;
  const int usl = __builtin_amdgcn_readfirstlane(sl);
  // (wave-uniform walk, advanced without divisions) the group to load next: segment (kh, j), group lg inside it
  int l_kh = (usl * per) / pl.s2, l_j = usl * per - l_kh * pl.s2, l_g = 0, l_t = 0;
  auto load_group = [&](Chunk<CIN4> (&a)[U], float4 (&b)[U][NI]) {
    const int r0 = l_j * pl.rs;
    const int seg_end = l_t < total ? min(r0 + pl.rs, rr) : 0;  // 0: nothing of this group exists
    const int ih = ih0 + l_kh;
    const bool row_ok = m_ok && unsigned(ih) < unsigned(d.h);
    const int prow = img + ih * lim;
    const int base = r0 + l_g * (8 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int off = base + 8 * u;  // this lane's chunk offset inside the kh row (without the lane half's 4: c0 carries it)
      a[u] = row_chunk<CIN4>(rx, prow, c0 + off, lim, row_ok && off + 4 * half < seg_end);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        b[u][i] = buf_load4(rw, wrow[i] + l_kh * rr + off, co_ok[i] && off + 4 * half < seg_end);
    }
    ++l_t;
    if (++l_g == gps) {
      l_g = 0;
      if (++l_j == pl.s2) {
        l_j = 0;
        ++l_kh;
      }
    }
  };
  Chunk<CIN4> ac[U], an[U];
  float4 bc[U][NI], bn[U][NI];
  load_group(ac, bc);
#pragma unroll 2
  for (int t = 0; t < total; ++t) {
    load_group(an, bn);
    __builtin_amdgcn_sched_barrier(0);  // the loads stay in front of the MFMAs that hide them
#pragma unroll
    for (int u = 0; u < U; ++u) mma(ac[u].get(), bc[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ac[u] = an[u];
#pragma unroll
      for (int i = 0; i < NI; ++i) bc[u][i] = bn[u][i];
    }
  }

  // ---- epilogue.  Straight-line code: bias and slope-mask values come through buffer loads whose out-of-range lanes read
  // nothing, results leave through buffer stores whose out-of-range lanes write nothing -- the element-wise branches of the
  // first version closed every `if` with a wait, so a wave walked 16 (32) DEPENDENT load -> store round trips.
  const __amdgpu_buffer_rsrc_t ry = conv_rsrc(y, size_t(d.m) * d.cout);
  const __amdgpu_buffer_rsrc_t rz = conv_rsrc(zmask ? zmask : y, size_t(d.m) * d.cout);
  const __amdgpu_buffer_rsrc_t rb = conv_rsrc(bias ? bias : w, bias ? size_t(d.cout) : 0);
  float bv[NI];
  int colv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    colv[i] = (tn * NI + i) * 32 + (lane & 31);
    bv[i] = buf_load1(rb, colv[i], bias != nullptr && colv[i] < d.cout);
  }
  auto slot_off = [&](int v, bool& ok) -> int {  // accumulator slot v = ni 16 + e of this lane -> element offset in y
    const int i = v >> 4, e = v & 15;
    const int col = (tn * NI + (NI == 1 ? 0 : i)) * 32 + (lane & 31);
    const int mo = tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
    ok = mo < d.m && col < d.cout;
    return mo * d.cout + col;
  };
  auto finish = [&](float val, int i, float z) -> float {
    val += bv[NI == 1 ? 0 : i];
    if (act) val = lrelu_v(val, leak);
    if (zmask) val *= lrelu_slope_v(z, leak);
    return val;
  };
  if (pl.s > 1) {
    // the S partial tiles meet in LDS; wave sl finishes accumulator slots v = sl, sl + S, ... in batches of four
    float* mine = part + sl * (NI * 1024);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) mine[(i * 16 + e) * 64 + lane] = acc[i][e];
    __syncthreads();
    for (int v0 = sl; v0 < NI * 16; v0 += 4 * pl.s) {
      float val[4], z[4];
      int off[4];
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = v0 + j * pl.s;
        const bool in = v < NI * 16;
        const int vv = in ? v : 0;
        off[j] = slot_off(vv, ok[j]);
        ok[j] = ok[j] && in;
        z[j] = buf_load1(rz, off[j], zmask != nullptr && ok[j]);
        float t = part[vv * 64 + lane];
        for (int q = 1; q < pl.s; ++q) t += part[q * (NI * 1024) + vv * 64 + lane];
        val[j] = t;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vv = v0 + j * pl.s < NI * 16 ? v0 + j * pl.s : 0;
        buf_store1(ry, off[j], ok[j], finish(val[j], vv >> 4, z[j]));
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float z[16];
      int off[16];
      bool ok[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        off[e] = slot_off(i * 16 + e, ok[e]);
        z[e] = buf_load1(rz, off[e], zmask != nullptr && ok[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) buf_store1(ry, off[e], ok[e], finish(acc[i][e], i, z[e]));
    }
  }
}

// ---- forward of the FIRST layers (6 / 14 / 17 planes -> 32 channels on 64-wide proxies): input rows staged in LDS ------
// K = 16 Cin is short (96 .. 272): a flat wave spends its life waiting for the first loads of 48 .. 136 MFMAs, and the
// im2col rows it loads overlap fourfold with its neighbours'.  Here a BLOCK owns R output rows of one image (wave r = output
// row oh0 + r = one 32-pixel M tile; all 32 output channels): the 2 R + 2 input rows they touch are staged ONCE with
// coalesced 16-byte loads -- consecutive rows of an image are one contiguous range -- into LDS rows with zero pixels left
// and right (no edge cases in the main loop), the weights beside them with every kh row padded to whole 8-k chunks (the
// pad multiplies zeros).  After one barrier the waves read their A / B fragments from LDS only: one global round trip per
// block instead of one per chunk group.  27.5 -> 14 us for the critic's first layer at batch 192 (0.4 GFLOP per 64 images).
__global__ __launch_bounds__(512) void conv_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* zmask, float* y,
                                                            ConvDims d, int rows, int act, float leak, ConvSecond sec) {
  if (blockIdx.y) { x = sec.x; w = sec.w; bias = sec.bias; zmask = sec.zmask; y = sec.y; }  // (the second problem of a pair)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cin = d.cin;
  const int lpad = (cin + 3) / 4 * 4;             // zero floats in front of a staged row (>= one pixel, whole float4s)
  const int rpad = (cin + 8 + 3) / 4 * 4;         // behind it: one pixel + the chunk padding of the last kh-row reads
  const int rowf = d.w * cin;                     // floats of an image row (a multiple of 4: w is even, and 16 | w here)
  const int pitch = lpad + rowf + rpad;
  const int rowk = (4 * cin + 7) / 8 * 8;         // a kh row of K, padded to whole chunks
  const int kp = 4 * rowk + 4;                    // LDS pitch of a weight row (conflict-free b128 reads: kp mod 32 in {4, 20})
  float* const xs = smem;                         // [2 rows + 2][pitch]
  float* const ws = smem + (2 * rows + 2) * pitch;  // [32][kp]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int blocks_per_image = d.ho / rows;
  const int n = blockIdx.x / blocks_per_image, oh0 = (blockIdx.x - n * blocks_per_image) * rows;
  // ---- stage: wave wv takes input rows wv, wv + rows, ... and output channels wv, wv + rows, ...
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int q4 = rowf / 4;
  for (int r = wv; r < 2 * rows + 2; r += rows) {
    const int ih = 2 * oh0 - 1 + r;
    float* const dst = xs + r * pitch;
    const bool inside = unsigned(ih) < unsigned(d.h);
    const float4* src = reinterpret_cast<const float4*>(x + (size_t(n) * d.h + (inside ? ih : 0)) * rowf);
    for (int q = lane; q < q4; q += 64) *reinterpret_cast<float4*>(dst + lpad + 4 * q) = inside ? src[q] : zero4;
    if (lane < lpad / 4) *reinterpret_cast<float4*>(dst + 4 * lane) = zero4;
    if (lane < rpad / 4) *reinterpret_cast<float4*>(dst + lpad + rowf + 4 * lane) = zero4;
  }
  for (int co = wv; co < 32; co += rows) {
    float* const dst = ws + co * kp;
    const bool co_in = co < d.cout;
    const float4* src = reinterpret_cast<const float4*>(w + size_t(co_in ? co : 0) * d.kdim);
    for (int j = lane; j < 4 * (rowk / 4); j += 64) {  // float4 slot j of the padded row: kh = j / (rowk / 4)
      const int kh = j / (rowk / 4), r4 = j - kh * (rowk / 4);
      *reinterpret_cast<float4*>(dst + kh * rowk + 4 * r4) = (co_in && r4 < cin) ? src[kh * cin + r4] : zero4;
    }
  }
  __syncthreads();
  // ---- compute: this wave's output row
  const int ow = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* const abase = xs + (2 * wv) * pitch + lpad + (2 * ow - 1) * cin + 4 * half;
  const float* const bbase = ws + ow * kp + 4 * half;  // (B rows are output channels: column `ow` of the lane)
  const int chunks = rowk / 8;
#pragma unroll
  for (int kh = 0; kh < 4; ++kh) {
    const float* ap = abase + kh * pitch;
    const float* bp = bbase + kh * rowk;
    for (int q = 0; q < chunks; ++q) {
      const float a0 = ap[8 * q], a1 = ap[8 * q + 1], a2 = ap[8 * q + 2], a3 = ap[8 * q + 3];  // (dword-aligned only)
      const float4 b = *reinterpret_cast<const float4*>(bp + 8 * q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b.w, acc, 0, 0, 0);
    }
  }
  // ---- epilogue (straight-line, as conv_fwd_flat_kernel): register e = pixel (e & 3) + 8 (e >> 2) + 4 half, lane = channel
  const __amdgpu_buffer_rsrc_t ry = conv_rsrc(y, size_t(d.m) * d.cout);
  const __amdgpu_buffer_rsrc_t rz = conv_rsrc(zmask ? zmask : y, size_t(d.m) * d.cout);
  const int co = lane & 31;
  const bool co_ok = co < d.cout;
  const float bv = (bias && co_ok) ? bias[co] : 0.f;
  const int m0 = (n * d.ho + oh0 + wv) * d.wo;
  float z[16];
  int off[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    off[e] = (m0 + (e & 3) + 8 * (e >> 2) + 4 * half) * d.cout + co;
    z[e] = buf_load1(rz, off[e], zmask != nullptr && co_ok);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float v = acc[e] + bv;
    if (act) v = lrelu_v(v, leak);
    if (zmask) v *= lrelu_slope_v(z[e], leak);
    buf_store1(ry, off[e], co_ok, v);
  }
}

// ---- the same layers when their input comes from expo_planes_concat / expo_net_inputs / expo_critic_penalty_tangent: channels
// 3 .. C - 1 are PER-IMAGE CONSTANTS (critics.py:64-76, agent.py:17-19: states and statistics broadcast as planes) -----------
// The convolution is linear: y = conv(image planes, W[..., :3]) + conv(constant planes, W[..., 3:]), and the second term of an
// output pixel depends only on the image, the channel and on WHICH TAPS fall inside the image -- 3 x 3 border classes
// (top / inner / bottom row x left / inner / right column; 4 x 4 taps, stride 2, pad 1: the first tap row is outside for
// oh = 0, the last for oh = ho - 1).  So K shrinks from 16 C (96 / 224 / 272) to 16 x 3 = 48 for EVERY first layer: a block
// stages three floats per pixel (14.7 KB of LDS instead of 42 - 119 KB: five blocks per CU instead of one for 17 planes),
// forms U[co][tap] = sum_c W[co][tap][3 + c] v_c once (v read from the image's first pixel), each wave adds the taps valid
// for its row into T[column class][co] (+ bias), and the epilogue adds T.  A different order of the same sum (float32
// rounding); the callers ask for it explicitly -- the kernel cannot know that planes are constant.
__global__ __launch_bounds__(512) void conv_fwd_planes_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* zmask, float* y,
                                                              ConvDims d, int rows, int act, float leak, ConvSecond sec) {
  if (blockIdx.y) { x = sec.x; w = sec.w; bias = sec.bias; zmask = sec.zmask; y = sec.y; }  // (the second problem of a pair)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int cin = d.cin, cv = cin - 3;            // constant planes
  constexpr int lpad = 4, rpad = 8, rowk = 16, kp = 4 * rowk + 4;  // a kh row of K: 4 pixels x 3 planes, padded to 16
  const int pitch = lpad + 3 * d.w + rpad;
  float* const xs = smem;                           // [2 rows + 2][pitch]: the image planes, zero pixels left and right
  float* const ws = xs + (2 * rows + 2) * pitch;    // [32][kp]: W[co][kh][kw][0..2], every kh row padded with zeros
  float* const us = ws + 32 * kp;                   // [32][16]: U[co][tap]
  float* const ts = us + 32 * 16;                   // [rows][3][32]: T[column class][co] of every wave's row
  float* const vs = ts + rows * 96;                 // [cv]: the constant planes' values
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int blocks_per_image = d.ho / rows;
  const int n = blockIdx.x / blocks_per_image, oh0 = (blockIdx.x - n * blocks_per_image) * rows;
  const size_t img = size_t(n) * d.h * d.w * cin;
  // ---- stage: image planes of the input rows (wave wv: rows wv, wv + rows, ...), weights of the image planes, constants
  for (int r = wv; r < 2 * rows + 2; r += rows) {
    const int ih = 2 * oh0 - 1 + r;
    float* const dst = xs + r * pitch;
    const bool inside = unsigned(ih) < unsigned(d.h);
    const float* src = x + img + size_t(inside ? ih : 0) * d.w * cin;
    for (int p = lane; p < d.w; p += 64) {  // a pixel per lane: its three image planes as one 12-byte load
      F3U v = {{0.f, 0.f, 0.f}};
      if (inside) v = *reinterpret_cast<const F3U*>(src + p * cin);
      dst[lpad + 3 * p] = v.v[0];
      dst[lpad + 3 * p + 1] = v.v[1];
      dst[lpad + 3 * p + 2] = v.v[2];
    }
    if (lane < lpad) dst[lane] = 0.f;
    if (lane < rpad) dst[lpad + 3 * d.w + lane] = 0.f;
  }
  for (int i = tid; i < 32 * 4 * rowk; i += 64 * rows) {
    const int co = i / (4 * rowk), rem = i - co * (4 * rowk), kh = rem / rowk, j = rem - kh * rowk;  // j = kw * 3 + c (< 12)
    const int kw = j / 3, c = j - 3 * kw;
    ws[co * kp + kh * rowk + j] = (co < d.cout && j < 12) ? w[(size_t(co) * 16 + kh * 4 + kw) * cin + c] : 0.f;
  }
  if (tid < cv) vs[tid] = x[img + 3 + tid];
  __syncthreads();
  // ---- U[co][tap] = sum_c W[co][tap][3 + c] v_c (one value per thread; 512 threads = 32 x 16)
  for (int i = tid; i < 512; i += 64 * rows) {
    const int co = i >> 4, tap = i & 15;
    float u = 0.f;
    if (co < d.cout) {
      const float* wr = w + (size_t(co) * 16 + tap) * cin + 3;
      for (int c = 0; c < cv; ++c) u = fmaf(wr[c], vs[c], u);
    }
    us[i] = u;
  }
  __syncthreads();
  // ---- T[cc][co] of this wave's output row: the taps inside the image for (row class, column class cc), + bias
  {
    const int oh = oh0 + wv;
    const int kh_lo = oh == 0 ? 1 : 0, kh_hi = oh == d.ho - 1 ? 2 : 3;
    for (int i = lane; i < 96; i += 64) {
      const int cc = i >> 5, co = i & 31;
      const int kw_lo = cc == 0 ? 1 : 0, kw_hi = cc == 2 ? 2 : 3;
      float t = (bias && co < d.cout) ? bias[co] : 0.f;
      for (int kh = kh_lo; kh <= kh_hi; ++kh)
        for (int kw = kw_lo; kw <= kw_hi; ++kw) t += us[co * 16 + kh * 4 + kw];
      ts[wv * 96 + i] = t;
    }
  }
  __syncthreads();
  // ---- compute: this wave's output row over the image planes (K = 4 x 16)
  const int ow = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* const abase = xs + (2 * wv) * pitch + lpad + (2 * ow - 1) * 3 + 4 * half;
  const float* const bbase = ws + ow * kp + 4 * half;
#pragma unroll
  for (int kh = 0; kh < 4; ++kh) {
    const float* ap = abase + kh * pitch;
    const float* bp = bbase + kh * rowk;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float a0 = ap[8 * q], a1 = ap[8 * q + 1], a2 = ap[8 * q + 2], a3 = ap[8 * q + 3];
      const float4 b = *reinterpret_cast<const float4*>(bp + 8 * q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b.w, acc, 0, 0, 0);
    }
  }
  // ---- epilogue: register e = pixel (e & 3) + 8 (e >> 2) + 4 half of the row, lane = channel
  const __amdgpu_buffer_rsrc_t ry = conv_rsrc(y, size_t(d.m) * d.cout);
  const __amdgpu_buffer_rsrc_t rz = conv_rsrc(zmask ? zmask : y, size_t(d.m) * d.cout);
  const int co = lane & 31;
  const bool co_ok = co < d.cout;
  const float t0 = ts[wv * 96 + co], t1 = ts[wv * 96 + 32 + co], t2 = ts[wv * 96 + 64 + co];
  const int m0 = (n * d.ho + oh0 + wv) * d.wo;
  float z[16];
  int off[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    off[e] = (m0 + (e & 3) + 8 * (e >> 2) + 4 * half) * d.cout + co;
    z[e] = buf_load1(rz, off[e], zmask != nullptr && co_ok);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int px = (e & 3) + 8 * (e >> 2) + 4 * half;
    float v = acc[e] + (px == 0 ? t0 : (px == d.wo - 1 ? t2 : t1));
    if (act) v = lrelu_v(v, leak);
    if (zmask) v *= lrelu_slope_v(z[e], leak);
    buf_store1(ry, off[e], co_ok, v);
  }
}

// K slicing for a target of ~16 waves per CU: S in {1, 2, 4, 8, 16} (S <= 16: one accumulator register per slice at
// least in the final sum), S2 = segments per kh row (S <= 4: whole rows)
static FlatPlan flat_plan(const ConvDims& d, int ni, int forced_s) {
  FlatPlan p;
  p.tiles_m = (d.m + 31) / 32;
  p.tiles_n = (d.cout + 32 * ni - 1) / (32 * ni);
  const long tiles = long(p.tiles_m) * p.tiles_n;
  const long target = 4096 / ni;  // wave count: ~4096 MFMA streams' worth of work chip-wide
  int s = 1;
  while (s < 16 && tiles * s * 2 <= target + target / 2) s *= 2;  // the power of two that brings tiles * s closest
  const int rr = 4 * d.cin;
  while (s > 4 && (rr + (s / 4) - 1) / (s / 4) < 8) s /= 2;  // a segment holds at least one chunk
  if (forced_s > 0) s = forced_s > 16 ? 16 : forced_s;
  p.s = s;
  p.s2 = s <= 4 ? 1 : s / 4;
  p.rs = ((rr + p.s2 - 1) / p.s2 + 7) / 8 * 8;
  return p;
}

// ---- data gradient (transposed convolution), flat decomposition -----------------------------------------------------
//   dX[n][ih][iw][ci] = sum_{kh, kw, co} dY[n][oh][ow][co] W[co][kh][kw][ci]   over the taps with ih = 2 oh - 1 + kh
// An input pixel of parity (ph, pw) = (ih & 1, iw & 1) is reached by 2 x 2 of the 4 x 4 taps: kh = 1 - ph + 2 th,
// oh = a + ph - th for ih = 2 a + ph (th = 0, 1), the same in w.  So the data gradient is FOUR dense GEMMs, one per parity
// class:  M' = N Ho Wo pixels (a, b),  N' = Cin,  K' = 4 taps x Cout:
//   A[m'][(t, co)] = dY[n][a + ph - th][b + pw - tw][co]      co contiguous: one 16-byte load per 4 k (Cout % 4 == 0)
//   B[(t, co)][ci] = W[co][kh_t][kw_t][ci]                     ci contiguous: lane l reads column l & 31 of FOUR rows co
//                                                              (4-byte loads; 32 lanes cover one 128-byte line)
// One wave = one 32 x 32 tile (pixels of one class x input channels) over one K' slice, like the forward; K' slices are
// cut along (tap, co range): G = 4 S2 segments of RS = roundup8(Cout / S2) channels.
template <int NI>
__global__ __launch_bounds__(1024) void conv_bwd_flat_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             const float* __restrict__ zmask, float* __restrict__ dx,
                                                             ConvDims d, FlatPlan pl, float leak, ConvSecond sec) {
  if (blockIdx.y) { dy = sec.x; w = sec.w; zmask = sec.zmask; dx = sec.y; }  // (the second problem of a pair)
  // zmask (nullable, the layer BELOW's activation z, shaped like dx): dx *= slope(z) -- the activation gradient of the
  // layer below in this kernel's epilogue instead of a launch of its own
  // NI = input-channel tiles per wave (32 pixels x 32 NI channels): the dY fragment is loaded once for NI MFMAs
  extern __shared__ __attribute__((aligned(16))) float part[];  // [s][NI][16][64]
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int per_class = pl.tiles_m * pl.tiles_n, nblocks = 4 * per_class;
  int bid = blockIdx.x;
  if (nblocks % 8 == 0) bid = (bid % 8) * (nblocks / 8) + bid / 8;
  const int cls = bid / per_class, rem_b = bid - cls * per_class;
  const int tm = rem_b / pl.tiles_n, tn = rem_b - tm * pl.tiles_n;
  const int ph = cls >> 1, pw = cls & 1;
  const int row = lane & 31, half = lane >> 5;
  const int m = tm * 32 + row;  // this lane's A row (input pixel of the class); its B columns: ci[i]
  const bool m_ok = m < d.m;
  int n, a, b;
  split_pixel(d, m_ok ? m : 0, n, a, b);
  const __amdgpu_buffer_rsrc_t rg = conv_rsrc(dy, size_t(d.n) * d.ho * d.wo * d.cout);
  const __amdgpu_buffer_rsrc_t rw = conv_rsrc(w, size_t(d.cout) * d.kdim);
  const int wstride = d.kdim;  // floats between two output channels of W
  int ci[NI];
  bool ci_ok[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    ci[i] = (tn * NI + i) * 32 + row;
    ci_ok[i] = ci[i] < d.cin;
  }

  f32x16 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  auto mma = [&](const float4& fa, const float4 (&fb)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[i].x, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[i].y, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[i].z, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[i].w, acc[i], 0, 0, 0);
    }
  };

  // one software pipeline over all the wave's segments (see conv_fwd_flat_kernel): the second layer's data gradient walks
  // 4 taps x 64 channels = 4 segments of 8 chunks per wave
  // (two chunks per group for either tile width: 64 registers, 8 waves per SIMD -- the second layer's 6 144 single-tile
  // waves at batch 192 are 6 per SIMD and run as one round; four chunks per group measured 46.7 against 45.4 us)
  constexpr int U = 2;
  const int g = 4 * pl.s2, per = g / pl.s;
  const int gps = (pl.rs + 8 * U - 1) / (8 * U);
  const int total = per * gps;
  const int usl = __builtin_amdgcn_readfirstlane(sl);
  int l_tap = (usl * per) / pl.s2, l_j = usl * per - l_tap * pl.s2, l_g = 0, l_t = 0;
  auto load_group = [&](float4 (&fa)[U], float4 (&fb)[U][NI]) {
    const int th = l_tap >> 1, tw = l_tap & 1;
    const int oh = a + ph - th, ow = b + pw - tw;
    const int kh = 1 - ph + 2 * th, kw = 1 - pw + 2 * tw;
    const bool pix_ok = m_ok && unsigned(oh) < unsigned(d.ho) && unsigned(ow) < unsigned(d.wo);
    const int abase = ((n * d.ho + oh) * d.wo + ow) * d.cout + 4 * half;        // + co
    const int bbase = (4 * half) * wstride + (kh * 4 + kw) * d.cin;             // + co * wstride + ci
    const int r0 = l_j * pl.rs;
    const int seg_end = l_t < total ? min(r0 + pl.rs, d.cout) : 0;
    const int base = r0 + l_g * (8 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int co = base + 8 * u;
      const bool in = co + 4 * half < seg_end;  // (Cout % 4 == 0: the four rows exist together)
      fa[u] = buf_load4(rg, abase + co, pix_ok && in);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const bool ok = ci_ok[i] && in;
        const int o = bbase + co * wstride + ci[i];
        fb[u][i] = make_float4(buf_load1(rw, o, ok), buf_load1(rw, o + wstride, ok), buf_load1(rw, o + 2 * wstride, ok),
                               buf_load1(rw, o + 3 * wstride, ok));
      }
    }
    ++l_t;
    if (++l_g == gps) {
      l_g = 0;
      if (++l_j == pl.s2) {
        l_j = 0;
        ++l_tap;
      }
    }
  };
  float4 ac[U], an[U], bc[U][NI], bn[U][NI];
  load_group(ac, bc);
#pragma unroll 2
  for (int t = 0; t < total; ++t) {
    load_group(an, bn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) mma(ac[u], bc[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ac[u] = an[u];
#pragma unroll
      for (int i = 0; i < NI; ++i) bc[u][i] = bn[u][i];
    }
  }

  // ---- epilogue, straight-line like the forward's: the slope mask through buffer loads issued together, the results
  // through buffer stores (out-of-range lanes touch nothing); pixel -> (n, a, b) by shifts for power-of-two geometry
  const size_t dx_floats = size_t(d.n) * d.h * d.w * d.cin;
  const __amdgpu_buffer_rsrc_t rdx = conv_rsrc(dx, dx_floats);
  const __amdgpu_buffer_rsrc_t rz = conv_rsrc(zmask ? zmask : dx, dx_floats);
  auto slot_off = [&](int v, bool& ok) -> int {  // accumulator slot v = ni 16 + e of this lane -> element offset in dx
    const int i = v >> 4, e = v & 15;
    const int mo = tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;  // the pixel of accumulator register e
    const int c = (tn * NI + (NI == 1 ? 0 : i)) * 32 + (lane & 31);
    ok = mo < d.m && c < d.cin;
    int no, ao, bo;
    split_pixel(d, ok ? mo : 0, no, ao, bo);
    return ((no * d.h + 2 * ao + ph) * d.w + 2 * bo + pw) * d.cin + c;
  };
  if (pl.s > 1) {
    float* mine = part + sl * (NI * 1024);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) mine[(i * 16 + e) * 64 + lane] = acc[i][e];
    __syncthreads();
    for (int v0 = sl; v0 < NI * 16; v0 += 4 * pl.s) {
      float val[4], z[4];
      int off[4];
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = v0 + j * pl.s;
        const bool in = v < NI * 16;
        const int vv = in ? v : 0;
        off[j] = slot_off(vv, ok[j]);
        ok[j] = ok[j] && in;
        z[j] = buf_load1(rz, off[j], zmask != nullptr && ok[j]);
        float t = part[vv * 64 + lane];
        for (int q = 1; q < pl.s; ++q) t += part[q * (NI * 1024) + vv * 64 + lane];
        val[j] = t;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        buf_store1(rdx, off[j], ok[j], zmask ? val[j] * lrelu_slope_v(z[j], leak) : val[j]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float z[16];
      int off[16];
      bool ok[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        off[e] = slot_off(i * 16 + e, ok[e]);
        z[e] = buf_load1(rz, off[e], zmask != nullptr && ok[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e)
        buf_store1(rdx, off[e], ok[e], zmask ? acc[i][e] * lrelu_slope_v(z[e], leak) : acc[i][e]);
    }
  }
}

// K' slicing of the data gradient: tiles = 4 classes x pixel tiles x channel tiles; segments along (tap, co range)
static FlatPlan bwd_plan(const ConvDims& d, int ni, int forced_s) {
  FlatPlan p;
  p.tiles_m = (d.m + 31) / 32;
  p.tiles_n = (d.cin + 32 * ni - 1) / (32 * ni);
  const long tiles = 4L * p.tiles_m * p.tiles_n;
  // ~4096 waves whatever the tile width: with two channel tiles per wave the third layer at batch 192 ran 54 us on 3072
  // waves and 43 us on 6144 (tools/r06/conv_sweep.py); the smaller batches are flat between 2048 and 4096
  const long target = 4096;
  int s = 1;
  while (s < 16 && tiles * s * 2 <= target + target / 2) s *= 2;
  while (s > 4 && (d.cout + (s / 4) - 1) / (s / 4) < 8) s /= 2;
  if (forced_s > 0) s = forced_s;
  if (s > 16) s = 16;
  p.s = s;
  p.s2 = s <= 4 ? 1 : s / 4;
  p.rs = ((d.cout + p.s2 - 1) / p.s2 + 7) / 8 * 8;
  return p;
}

// ---- data gradient of the critic's FIRST layer (6 input planes) on the vector ALUs -------------------------------------
// (a template over the plane count: the value net's 17 planes ran here too until the matrix-core kernel overtook it)
// With Cin = 6 a 32-wide tile of input channels is 81 % padding: the matrix-core kernel above takes 47 us for 0.4 GFLOP.
// Here a BLOCK owns four dY rows' worth of input pixels of one image (input rows 2 a0 .. 2 a0 + 7): it stages the six dY
// rows a0 - 1 .. a0 + 4 they touch in LDS with coalesced 16-byte loads (pixel stride padded to Cout + 4 floats: the
// per-pixel 16-byte reads below are conflict-free), and the weights of all four parity classes as [class][tap][co][ci].
// WAVE = parity class (ph, pw) -- its taps and weights are wave-uniform, read from LDS as broadcasts -- and a lane = a
// column b and two of the four rows: 2 pixels x CIN accumulators, 4 taps x Cout x CIN FMAs each, every weight read feeding
// both pixels.  No padding work, no scattered global loads: 0.4 GFLOP for Cin = 6 at batch 64.  (Earlier versions of this
// round: weights in scalar registers, one s_load per FMA group -- 23 / 84 us, scalar-load latency; four adjacent pixels per
// thread straight from global memory -- 29 / 58 us, 64 cache lines per wave load.  profiles/r06_experiments.md)
template <int CIN>
__global__ __launch_bounds__(256) void conv_bwd_small_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             const float* __restrict__ zmask, float* __restrict__ dx,
                                                             ConvDims d, float leak) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CINP = (CIN + 3) / 4 * 4;  // weight rows padded to whole float4s
  const int ps = d.cout + 4;               // floats per staged dY pixel
  const int rowf = (d.wo + 2) * ps;        // a staged row: pixels ow = -1 .. wo (the two outer ones zero)
  float* const wl = smem;                           // [4 classes][4 taps][cout][CINP]  (first: a statically 16-byte aligned base)
  float* const gs = smem + 16 * d.cout * CINP;      // [6 rows][wo + 2][ps]
  const int blocks_per_image = d.ho / 4;
  const int n = blockIdx.x / blocks_per_image, a0 = (blockIdx.x - n * blocks_per_image) * 4;
  const int tid = threadIdx.x;
  // ---- stage: weights (every class), then the six dY rows (rows outside the image and the outer pixels as zeros) ----
  for (int i = tid; i < 16 * d.cout * CINP; i += 256) {
    const int c = i % CINP, rest = i / CINP, co = rest % d.cout, ct = rest / d.cout, t = ct & 3, cls = ct >> 2;
    const int kh = 1 - (cls >> 1) + 2 * (t >> 1), kw = 1 - (cls & 1) + 2 * (t & 1);
    wl[i] = c < CIN ? w[(size_t(co) * 16 + kh * 4 + kw) * CIN + c] : 0.f;
  }
  const int q4 = d.cout / 4;  // float4 chunks per pixel
  for (int i = tid; i < 6 * (d.wo + 2) * q4; i += 256) {
    const int q = i % q4, rest = i / q4, px = rest % (d.wo + 2), r = rest / (d.wo + 2);
    const int oh = a0 - 1 + r, ow = px - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (unsigned(oh) < unsigned(d.ho) && unsigned(ow) < unsigned(d.wo))
      v = *reinterpret_cast<const float4*>(dy + (size_t(n * d.ho + oh) * d.wo + ow) * d.cout + 4 * q);
    *reinterpret_cast<float4*>(gs + r * rowf + px * ps + 4 * q) = v;
  }
  __syncthreads();
  // ---- compute: wave = class, lane = (row pair, column) ------------------------------------------------------------------
  const int cls = tid >> 6, ph = cls >> 1, pw = cls & 1, lane = tid & 63;
  const float4* const wc = reinterpret_cast<const float4*>(wl) + size_t(cls) * 4 * d.cout * (CINP / 4);
  for (int b = lane & 31; b < d.wo; b += 32) {  // (wo = 32 for the 64 x 64 proxies: one pass)
    const int al = lane >> 5;                   // pixels a0 + al and a0 + al + 2
    float acc[2][CIN];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc[j][c] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int th = t >> 1, tw = t & 1;
      // staged row of pixel j: oh - (a0 - 1) = al + 2 j + ph - th + 1; staged column: ow + 1 = b + pw - tw + 1
      // (float4 units: rows and pixels are whole float4s, so the reads are single 16-byte LDS operations)
      const float4* g0 = reinterpret_cast<const float4*>(gs) + ((al + ph - th + 1) * rowf + (b + pw - tw + 1) * ps) / 4;
      const float4* g1 = g0 + 2 * rowf / 4;
      const float4* wt = wc + size_t(t) * d.cout * (CINP / 4);
      for (int co = 0; co < d.cout; co += 4) {
        const float4 ga = g0[co / 4], gb = g1[co / 4];
        const float va[4] = {ga.x, ga.y, ga.z, ga.w}, vb[4] = {gb.x, gb.y, gb.z, gb.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float wv[CINP];
#pragma unroll
          for (int c4 = 0; c4 < CINP; c4 += 4) {
            const float4 v = wt[(co + q) * (CINP / 4) + c4 / 4];  // wave-uniform address: a broadcast read
            wv[c4] = v.x; wv[c4 + 1] = v.y; wv[c4 + 2] = v.z; wv[c4 + 3] = v.w;
          }
#pragma unroll
          for (int c = 0; c < CIN; ++c) {
            acc[0][c] = fmaf(va[q], wv[c], acc[0][c]);
            acc[1][c] = fmaf(vb[q], wv[c], acc[1][c]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int a = a0 + al + 2 * j;
      const size_t o = (size_t(n * d.h + 2 * a + ph) * d.w + 2 * b + pw) * CIN;
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        float v = acc[j][c];
        if (zmask) v *= lrelu_slope_v(zmask[o + c], leak);
        dx[o + c] = v;
      }
    }
  }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
//   dW[co][k2] = sum_m dY[m][co] A[m][k2],   m = (n, oh, ow),  k2 = (kh, kw, ci),  A = the forward's im2col matrix
// A GEMM with the PIXELS as its K dimension (65536 ... 1024 of them at batch 64) and a small output (Cout x 16 Cin).
// One wave = a 32 (co) x 128 (k2) tile of dW over a range of pixel pairs: per pair (MFMA k = 2: lane half h <-> pixel
// 2 q + h) a lane loads ONE float of dY (row co0 + (l & 31): 32 lanes cover one 128-byte line) and ONE 16-byte chunk of
// the im2col row (k2 = k2_0 + 4 (l & 31) .. + 3: 32 lanes cover 512 contiguous bytes of the image row) -- and that
// chunk's four elements are the B operands of FOUR MFMAs on four interleaved 32 x 32 tiles (tile r holds the columns
// k2 = 4 j + r): 2 loads per 4 MFMAs, and accumulator register e of the four tiles is one float4 of dW[co][4 j .. 4 j + 3],
// so the stores are whole 512-byte rows.  (The first version, r05p3, loaded 4 bytes per operand and MFMA: 2-8x slower
// than MIOpen.)  The pixel range of a tile is spread over P blocks x S waves: the S partial tiles of a block meet in
// LDS, the P block partials are written as P full-size copies of dW into a workspace and summed in block order by a
// second, element-wise launch (P = 1: straight to dW).  No atomics, no zero fill, no fence; a fixed summation order.
struct WrwPlan {
  int tiles_co, tiles_k;  // 32 x 128 tiles of dW
  int s, p;               // waves per block, blocks per tile
  int pairs_per_wave;     // pixel pairs (2 consecutive ow of one row) per wave
};

// Bias gradient in the same pass (db_out non-null): db[co] = the column sums of dY over the pixel pairs below `bias_pairs`
// (the critic step batches passes whose bias gradients differ: only the first images' rows count) -- the waves of the
// tiles with tk = 0 add up the dY values they feed to the matrix cores anyway; one float per block copy and channel.
// (the body of the kernel: `block` = the block's index among THIS layer's blocks -- blockIdx.x for a launch of its own,
// blockIdx.x - first_block inside a grouped launch; waves beyond pl.s (a grouped launch has 256 threads per block whatever
// the layer's plan) own no pixels and take part in the barriers only)
template <bool CIN4>
__device__ __forceinline__ void conv_wrw_body(const float* __restrict__ x, const float* __restrict__ dy,
                                              float* __restrict__ out, size_t out_stride, float* __restrict__ db_out,
                                              size_t db_stride, int bias_pairs, const ConvDims& d, const WrwPlan& pl,
                                              int block, float* part, float (*bpart)[32]) {
  const int lane = threadIdx.x & 63;
  const int sl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool idle = sl >= pl.s;
  const int tiles = pl.tiles_co * pl.tiles_k;
  const int tile = block % tiles, pb = block / tiles;
  const int tc = tile / pl.tiles_k, tk = tile - tc * pl.tiles_k;
  const int col = lane & 31, half = lane >> 5;
  const int rr = 4 * d.cin, lim = d.w * d.cin;
  const __amdgpu_buffer_rsrc_t rx = conv_rsrc(x, size_t(d.n) * d.h * d.w * d.cin);
  const __amdgpu_buffer_rsrc_t rg = conv_rsrc(dy, size_t(d.n) * d.ho * d.wo * d.cout);
  // A operand: dY column co (this lane's output row); B operand: the im2col chunk k2 .. k2 + 3 (its output columns)
  const int co = tc * 32 + col;
  const bool co_ok = co < d.cout;
  const int k2 = tk * 128 + 4 * col;
  const bool k2_ok = k2 < d.kdim;  // (kdim = 16 cin is a multiple of 4: a chunk exists as a whole or not at all)
  const int kk = k2_ok ? k2 : 0;
  const int kh = (kk >= rr) + (kk >= 2 * rr) + (kk >= 3 * rr);
  const int b_c = (2 * half - 1) * d.cin + (kk - kh * rr);  // + 4 owp cin: offset inside the image row
  const int a_off = half * d.cout + co;                      // + (the pair's first pixel) * cout

  f32x16 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

  const int wpairs = d.wo / 2, total_pairs = d.m / 2;  // (wo is even: see the host side)
  const int q0 = (pb * pl.s + sl) * pl.pairs_per_wave;
  const int q1 = idle ? q0 : min(q0 + pl.pairs_per_wave, total_pairs);
  // wave-uniform walk over the pixel pairs (scalar registers), advanced without divisions
  int it_row = q0 / wpairs;  // n ho + oh
  int it_owp = q0 - it_row * wpairs;
  int it_n = it_row / d.ho;
  int it_oh = it_row - it_n * d.ho;
  int it_q = q0;
  auto load_pair = [&](float& fa, Chunk<CIN4>& fb) {
    const bool q_ok = it_q < q1;
    fa = buf_load1(rg, (it_row * d.wo + 2 * it_owp) * d.cout + a_off, q_ok && co_ok);
    const int ih = 2 * it_oh - 1 + kh;
    fb = row_chunk<CIN4>(rx, (it_n * d.h + ih) * lim, b_c + 4 * it_owp * d.cin, lim,
                         q_ok && k2_ok && unsigned(ih) < unsigned(d.h));
    ++it_q;
    if (++it_owp == wpairs) {
      it_owp = 0;
      ++it_row;
      if (++it_oh == d.ho) {
        it_oh = 0;
        ++it_n;
      }
    }
  };
  constexpr int U = 4;  // (8 pairs in flight measured no gain: 44.2 vs 43.4 us for the second layer at batch 192)
  float ac[U], an[U];
  Chunk<CIN4> bc[U], bn[U];
  const bool want_bias = db_out != nullptr && tk == 0;
  float bsum = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) load_pair(ac[u], bc[u]);
#pragma unroll 2
  for (int q = q0; q < q1; q += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) load_pair(an[u], bn[u]);
    __builtin_amdgcn_sched_barrier(0);  // the loads stay in front of the MFMAs that hide them
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (want_bias && q + u < bias_pairs) bsum += ac[u];  // (pairs past q1 were loaded as zeros)
      const float4 b = bc[u].get();
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], b.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], b.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], b.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], b.w, acc[3], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ac[u] = an[u];
      bc[u] = bn[u];
    }
  }

  // accumulator register e of tile r, lane l: row (co) = (e & 3) + 8 (e >> 2) + 4 (l >> 5), column (k2) = 4 (l & 31) + r
  // -> the four tiles' register e is dW[row][k2_0 + 4 (l & 31) .. + 3]: one float4 per lane, 512 contiguous bytes per row
  if (want_bias) {  // (block-uniform) the two pixels of a pair sit in the two lane halves; waves are added in order
    bsum += __shfl_xor(bsum, 32);
    if (half == 0 && !idle) bpart[sl][col] = bsum;
    __syncthreads();
    if (sl == 0 && half == 0 && co_ok) {
      float v = bpart[0][col];
      for (int q = 1; q < pl.s; ++q) v += bpart[q][col];
      db_out[size_t(pb) * db_stride + co] = v;
    }
  }
  float* const dst = out + size_t(pb) * out_stride;  // this block's copy of dW (P = 1: dW itself)
  auto store_row = [&](int e, const float4& v) {
    const int r_co = tc * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
    if (r_co < d.cout && k2_ok) *reinterpret_cast<float4*>(dst + size_t(r_co) * d.kdim + k2) = v;
  };
  if (pl.s > 1) {
    float* mine = part + sl * 4096;
    if (!idle) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        *reinterpret_cast<float4*>(mine + (e * 64 + lane) * 4) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
    }
    __syncthreads();
    for (int e = idle ? 16 : sl; e < 16; e += pl.s) {
      float4 v = *reinterpret_cast<const float4*>(part + (e * 64 + lane) * 4);
      for (int q = 1; q < pl.s; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(part + q * 4096 + (e * 64 + lane) * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      store_row(e, v);
    }
  } else if (!idle) {
#pragma unroll
    for (int e = 0; e < 16; ++e) store_row(e, make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]));
  }
}

template <bool CIN4>
__global__ __launch_bounds__(256) void conv_wrw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ out, size_t out_stride,
                                                       float* __restrict__ db_out, size_t db_stride, int bias_pairs,
                                                       ConvDims d, WrwPlan pl) {
  extern __shared__ __attribute__((aligned(16))) float part[];  // [s][64][64]: slot v = e * 4 + r
  __shared__ float bpart[4][32];
  conv_wrw_body<CIN4>(x, dy, out, out_stride, db_out, db_stride, bias_pairs, d, pl, blockIdx.x, part, bpart);
}

// The weight gradients of a whole STACK of layers as one launch: the four layers' kernels are independent, and launched
// one after the other each pays its own ramp and tail (17-21 us per launch at batch 64 for 7 us of matrix-core work;
// ~12 us of a 40 us launch at batch 192).  A block finds its layer in the table of first blocks and runs that layer's body.
constexpr int kWrwGroupMax = 8;
struct WrwGroupItem {
  const float* x;
  const float* dy;
  float* out;
  float* db_out;
  size_t stride;  // floats between two block copies (of dW and of the bias sums alike)
  ConvDims d;
  WrwPlan pl;
  int bias_pairs, cin4;
};
struct WrwGroupArgs {
  WrwGroupItem it[kWrwGroupMax];
  unsigned first_block[kWrwGroupMax + 1];
  int layers;
};
__global__ __launch_bounds__(256) void conv_wrw_group_kernel(const WrwGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) float part[];
  __shared__ float bpart[4][32];
  int l = 0;
  while (l + 1 < g.layers && blockIdx.x >= g.first_block[l + 1]) ++l;
  const WrwGroupItem& it = g.it[l];
  const int block = int(blockIdx.x - g.first_block[l]);
  if (it.cin4)
    conv_wrw_body<true>(it.x, it.dy, it.out, it.stride, it.db_out, it.stride, it.bias_pairs, it.d, it.pl, block, part, bpart);
  else
    conv_wrw_body<false>(it.x, it.dy, it.out, it.stride, it.db_out, it.stride, it.bias_pairs, it.d, it.pl, block, part, bpart);
}

// dW = the sum of the P block copies.  A block owns 16 float4 elements; its 16 thread groups each add every 16th copy
// (g, g + 16, ...), then the 16 group sums are added in group order through LDS: a fixed order, and P / 16 loads deep
// instead of P (the first layers' dW is 7 168 floats under 128 copies: one thread per element walked 128 dependent-ish
// loads in 7 blocks -- most of that layer's 59 us).
__global__ __launch_bounds__(256) void conv_wrw_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                              float* __restrict__ dbias, size_t dw4, size_t count4,
                                                              size_t stride4, int p) {
  // a copy = stride4 float4 elements of which count4 are summed: the first dw4 dW, the rest (if any) the bias gradient
  __shared__ float4 part[16][16];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t i = size_t(blockIdx.x) * 16 + el;
  const float4* src = reinterpret_cast<const float4*>(ws);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < count4) {
    for (int q = grp; q < p; q += 16) {
      const float4 t = src[size_t(q) * stride4 + i];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
  }
  part[grp][el] = v;
  __syncthreads();
  if (grp == 0 && i < count4) {
    float4 r = part[0][el];
#pragma unroll
    for (int g = 1; g < 16; ++g) {
      const float4 t = part[g][el];
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    if (i < dw4) reinterpret_cast<float4*>(dw)[i] = r;
    else reinterpret_cast<float4*>(dbias)[i - dw4] = r;
  }
}

// The same sum for SEVERAL layers in one launch (the weight gradients of a whole stack of layers: their main kernels run
// back to back, one reduce launch finishes them all).  A block finds its layer in the table of first blocks.
struct WrwReduceGroup {
  const float* ws[kWrwGroupMax];
  float* dw[kWrwGroupMax];
  float* dbias[kWrwGroupMax];
  unsigned dw4[kWrwGroupMax], count4[kWrwGroupMax], stride4[kWrwGroupMax], p[kWrwGroupMax];
  unsigned g_log2[kWrwGroupMax];  // log2 of the thread groups that share the copies of one element (<= 4)
  unsigned first_block[kWrwGroupMax + 1];
  int layers;
};
__global__ __launch_bounds__(256) void conv_wrw_reduce_group_kernel(const WrwReduceGroup g) {
  // G = min(16, P) thread groups (a power of two >= P when P < 16), 256 / G elements per block: with 4 copies (the last
  // layer) the fixed 16 x 16 split left 12 of 16 groups idle and cut 2 MB of dW into 8 192 blocks of 1 KB.  The order of
  // the sum is the one of conv_wrw_reduce_kernel: group q adds copies q, q + 16, ..., the groups are added in order
  // (empty groups contribute exact zeros there), so the results are bit-identical.
  __shared__ float4 part[256];
  int l = 0;
  while (l + 1 < g.layers && blockIdx.x >= g.first_block[l + 1]) ++l;
  const int gl = int(g.g_log2[l]), groups = 1 << gl, per = 256 >> gl;  // groups x per = 256 threads
  const int el = threadIdx.x & (per - 1), grp = threadIdx.x >> (8 - gl);
  const size_t i = size_t(blockIdx.x - g.first_block[l]) * per + el;
  const float4* src = reinterpret_cast<const float4*>(g.ws[l]);
  const size_t count4 = g.count4[l], stride4 = g.stride4[l];
  const int p = int(g.p[l]);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < count4) {
    for (int q = grp; q < p; q += groups) {
      const float4 t = src[size_t(q) * stride4 + i];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
  }
  part[grp * per + el] = v;
  __syncthreads();
  if (grp == 0 && i < count4) {
    float4 r = part[el];
    for (int k = 1; k < groups; ++k) {
      const float4 t = part[k * per + el];
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    if (i < g.dw4[l]) reinterpret_cast<float4*>(g.dw[l])[i] = r;
    else reinterpret_cast<float4*>(g.dbias[l])[i - g.dw4[l]] = r;
  }
}

static WrwPlan wrw_plan(const ConvDims& d, int forced_s, int forced_p) {
  WrwPlan p;
  p.tiles_co = (d.cout + 31) / 32;
  p.tiles_k = (d.kdim + 127) / 128;
  const long tiles = long(p.tiles_co) * p.tiles_k;
  const int total_pairs = d.m / 2;
  const int batch = 4;
  long waves = (2048 + tiles - 1) / tiles;  // per tile: two waves of 4 MFMA streams per SIMD chip-wide
  if (waves > (total_pairs + 2 * batch - 1) / (2 * batch)) waves = (total_pairs + 2 * batch - 1) / (2 * batch);
  if (waves < 1) waves = 1;
  int s = 1;
  while (s < 4 && s * 2 <= waves) s *= 2;
  if (forced_s > 0) s = forced_s > 4 ? 4 : forced_s;
  int pb = int((waves + s - 1) / s);
  if (pb > 256) pb = 256;  // beyond 256 copies of dW the reduce launch costs more than the shorter pixel ranges save
  // one round of blocks: a 4-wave block holds 64 KB of LDS for its reduction, so a CU takes two and the chip 512 --
  // 3 tiles x 171 parts = 513 blocks ran the 17-plane layer in two rounds (49.6 us; 39.3 with 128 parts)
  if (long(pb) * tiles > 512) pb = int(512 / tiles) > 0 ? int(512 / tiles) : 1;
  if (forced_p > 0) pb = forced_p;
  p.s = s;
  p.pairs_per_wave = (total_pairs + pb * s - 1) / (pb * s);
  p.pairs_per_wave = (p.pairs_per_wave + batch - 1) / batch * batch;
  p.p = (total_pairs + p.pairs_per_wave * s - 1) / (p.pairs_per_wave * s);  // no empty blocks
  if (p.p < 1) p.p = 1;
  return p;
}

template <int BM, int BN, int WM, int WN, int WK, int BKS>
static void launch_fwd(const float* x, const float* w, const float* bias, const float* zmask, float* y,
                       const ConvDims& d, int act, float leak, hipStream_t s, const ConvSecond* sec) {
  const int blocks = ((d.m + BM - 1) / BM) * ((d.cout + BN - 1) / BN);
  const dim3 grid(blocks, sec ? 2 : 1);
  const ConvSecond s2 = sec ? *sec : ConvSecond{};
  if (d.cin % 4 == 0)
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, WK, BKS, true>), grid, dim3(64 * WM * WN * WK), 0, s, x, w,
                       bias, zmask, y, d, act, leak, s2);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, WK, BKS, false>), grid, dim3(64 * WM * WN * WK), 0, s, x, w,
                       bias, zmask, y, d, act, leak, s2);
}

static int conv_dims(ConvDims* d, int n, int h, int w, int cin, int cout) {
  if (n < 0 || h < 2 || w < 2 || (h & 1) || (w & 1) || cin < 1 || cout < 1)
    return fail(EXPO_E_BADARG, "conv4x4s2: n >= 0, even h, w >= 2, cin, cout >= 1 required");
  // tensors are addressed through 32-bit byte offsets (raw buffer resources)
  if (double(n) * h * w * cin * 4 > 2.0e9 || double(n) * (h / 2) * (w / 2) * cout * 4 > 2.0e9 || double(cout) * 16 * cin * 4 > 2.0e9)
    return fail(EXPO_E_BADARG, "conv4x4s2: a tensor of 2 GB or more is not supported");
  d->n = n; d->h = h; d->w = w; d->cin = cin; d->cout = cout;
  d->ho = h / 2; d->wo = w / 2; d->kdim = 16 * cin; d->m = n * d->ho * d->wo;
  d->sh_w = d->sh_hw = -1;
  if ((d->wo & (d->wo - 1)) == 0 && (d->ho & (d->ho - 1)) == 0) {
    d->sh_w = __builtin_ctz(unsigned(d->wo));
    d->sh_hw = d->sh_w + __builtin_ctz(unsigned(d->ho));
  }
  return EXPO_OK;
}

// Decomposition overrides (probes and tests; 0 = the library's own choice): read from the environment ONCE --
// EXPO_CONV_TILE (1-4: an LDS-tiled forward shape, 5: the flat kernel, 6: the first layers' row-staged kernel), EXPO_CONV_NT (column tiles per wave, 1 | 2),
// EXPO_CONV_SLICES (K slices of the forward / data-gradient kernels: 1, 2, 4, 8 or 16), EXPO_CONV_WRW_SLICES (waves per
// block of the weight-gradient kernel: 1 .. 4), EXPO_CONV_PARTS (its block copies) -- and settable through
// expo_conv_tuning() / expo_conv_wrw_tuning() afterwards (no getenv on the launch path: the backward kernels are launched
// from autograd's worker thread while the host thread may be in putenv).
static bool pow2_upto(int v, int hi) { return v >= 1 && v <= hi && (v & (v - 1)) == 0; }
struct ConvTuning {
  std::atomic<int> tile, nt, slices, wrw_slices, parts;
  ConvTuning() : tile(env_int("EXPO_CONV_TILE", 0)), nt(env_int("EXPO_CONV_NT", 0)), slices(env_int("EXPO_CONV_SLICES", 0)),
                 wrw_slices(env_int("EXPO_CONV_WRW_SLICES", 0)), parts(env_int("EXPO_CONV_PARTS", 0)) {
    // the kernels cut K into 4 S2 segments, S | 4 S2: anything but a power of two would silently drop segments
    if (slices.load() != 0 && !pow2_upto(slices.load(), 16)) slices.store(0);
    if (wrw_slices.load() > 4) wrw_slices.store(4);  // (the weight-gradient kernel takes any 1 .. 4 waves per block)
  }
};
static ConvTuning& conv_tuning() {
  static ConvTuning t;
  return t;
}

static int conv_fwd_impl(const float* x, const float* w, const float* bias, const float* zmask, float* y, int n, int h,
                         int wd, int cin, int cout, int act, float leak, void* stream, const ConvSecond* sec = nullptr) {
  ConvDims d;
  if (int rc = conv_dims(&d, n, h, wd, cin, cout)) return rc;
  if (n == 0) return EXPO_OK;
  if (!x || !w || !y) return fail(EXPO_E_BADARG, "null pointer");
  if ((reinterpret_cast<uintptr_t>(w) & 15) != 0) return fail(EXPO_E_BADARG, "conv4x4s2: weight must be 16-byte aligned");
  if (sec && (!sec->x || !sec->w || !sec->y || (reinterpret_cast<uintptr_t>(sec->w) & 15) != 0 ||
              ((reinterpret_cast<uintptr_t>(sec->x) ^ reinterpret_cast<uintptr_t>(x)) & 15) != 0 || !sec->bias != !bias ||
              !sec->zmask != !zmask))
    return fail(EXPO_E_BADARG, "conv4x4s2 pair: the second problem needs the same operands, equally aligned");
  const dim3 gy2(1, sec ? 2 : 1);
  const ConvSecond s2 = sec ? *sec : ConvSecond{};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int forced = conv_tuning().tile.load(), forced_nt = conv_tuning().nt.load(), forced_s = conv_tuning().slices.load();
  const int pair_mul = sec ? 2 : 1;  // a pair is twice the blocks: tile shape and K slices for the grid that actually runs
  // (the G / V step 1.27 -> 1.25 ms against planning each half as if it ran alone)
  auto blocks = [&](int bm, int bn) { return ((d.m + bm - 1) / bm) * ((d.cout + bn - 1) / bn); };
  int shape = forced;
  // Measured on MI355X (tools/r05/conv_bench.py, profiles/r05_final_conv_bench.txt): the LDS-tiled shapes win where 64 x 64
  // tiles still give every CU a block or two (the second layer; the third at batch 128), the flat decomposition
  // everywhere else (first layers: K is short; deep layers: few rows, long K)
  if (shape == 0 && d.cout > 32 && forced_nt == 0 && forced_s == 0) {
    const int b64 = blocks(64, 64) * pair_mul;
    if (b64 >= 512) shape = 2;
    else if (b64 >= 256) shape = 3;
  }
  // the first layers on 64-wide proxies (agent.py:21, critics.py:13: 64 x 64 inputs, base_channels = 32): input rows staged
  // in LDS (conv_fwd_rows_kernel); tile code 6 forces it where it applies, any other forced plan keeps it out
  if ((shape == 0 || shape == 6) && forced_nt == 0 && forced_s == 0 && d.wo == 32 && d.cout <= 32 && d.cin <= 20 &&
      d.ho % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // eight output rows per block (the weights are staged once per block; 42 / 97 / 119 KB of LDS for 6 / 14 / 17 planes:
    // three / one / one block of eight waves per CU), four where the image is not a multiple of eight rows
    const int rows = d.ho % 8 == 0 ? 8 : 4;
    const int cin_ = d.cin, lpad = (cin_ + 3) / 4 * 4, rpad = (cin_ + 8 + 3) / 4 * 4, rowk = (4 * cin_ + 7) / 8 * 8;
    const size_t lds = (size_t(2 * rows + 2) * (lpad + d.w * cin_ + rpad) + size_t(32) * (4 * rowk + 4)) * 4;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_rows_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr;
    if (lds <= 160 * 1024) {
      hipLaunchKernelGGL(conv_fwd_rows_kernel, dim3(unsigned(d.n * (d.ho / rows)), gy2.y), dim3(64 * rows), lds, s, x, w, bias,
                         zmask, y, d, rows, act, leak, s2);
      HIP_TRY(hipGetLastError(), "conv4x4s2_fwd (rows) launch");
      return EXPO_OK;
    }
  }
  if (shape == 6) shape = 0;
  if (shape < 1 || shape > 4) {
    // the flat decomposition (default)
    int ni = forced_nt;  // column tiles per wave (0: the library's choice)
    // two column tiles per wave (A fragment shared in registers) where that still leaves >= 256 tiles: the third layer,
    // the fourth at batch 128 (17.3 vs 20.0 us, 26.3 vs 31.5 us)
    if (ni == 0) {
      const long t2 = pair_mul * long((d.m + 31) / 32) * ((d.cout + 63) / 64), t1 = pair_mul * long((d.m + 31) / 32) * ((d.cout + 31) / 32);
      ni = t2 >= 256 ? 2 : 1;
      // blocks that do not divide among the 256 CUs leave half of them a round short: the fourth layer at batch 192 is 384
      // blocks with two column tiles per wave (48.8 us) and 768 with one (44.4 us)
      if (ni == 2 && t2 % 256 != 0 && t2 < 1024 && t1 % 256 == 0) ni = 1;
    }
    if (ni != 1 && ni != 2) ni = 1;
    if (d.cout <= 32) ni = 1;
    FlatPlan pl = flat_plan(d, ni, forced_s);
    if (pair_mul == 2 && forced_s == 0) {  // K slices as for twice the pixels
      ConvDims d2 = d;
      d2.m *= 2;
      const FlatPlan p2 = flat_plan(d2, ni, 0);
      pl.s = p2.s; pl.s2 = p2.s2; pl.rs = p2.rs;
    }
    const int nblocks = pl.tiles_m * pl.tiles_n;
    const size_t lds = pl.s > 1 ? size_t(pl.s) * ni * 4096 : 0;
#define EXPO_FLAT(NI, C4) \
  hipLaunchKernelGGL((conv_fwd_flat_kernel<NI, C4>), dim3(nblocks, gy2.y), dim3(64 * pl.s), lds, s, x, w, bias, zmask, y, d, pl, act, \
                     leak, s2)
    if (ni == 2) { if (d.cin % 4 == 0) EXPO_FLAT(2, true); else EXPO_FLAT(2, false); }
    else { if (d.cin % 4 == 0) EXPO_FLAT(1, true); else EXPO_FLAT(1, false); }
#undef EXPO_FLAT
    HIP_TRY(hipGetLastError(), "conv4x4s2_fwd launch");
    return EXPO_OK;
  }
  switch (shape) {
    case 1: launch_fwd<64, 32, 2, 1, 2, 32>(x, w, bias, zmask, y, d, act, leak, s, sec); break;
    case 2: launch_fwd<64, 64, 2, 2, 1, 32>(x, w, bias, zmask, y, d, act, leak, s, sec); break;
    case 3: launch_fwd<32, 64, 1, 2, 2, 32>(x, w, bias, zmask, y, d, act, leak, s, sec); break;
    default: launch_fwd<32, 32, 1, 1, 4, 16>(x, w, bias, zmask, y, d, act, leak, s, sec); break;
  }
  HIP_TRY(hipGetLastError(), "conv4x4s2_fwd launch");
  return EXPO_OK;
}

// The first layers on inputs whose channels 3 .. are per-image constants (conv_fwd_planes_kernel); EXPO_E_BADARG for any
// other geometry -- the callers then take expo_conv4x4s2_fwd.
static int conv_fwd_planes_impl(const float* x, const float* w, const float* bias, const float* zmask, float* y, int n, int h,
                                int wd, int cin, int cout, int act, float leak, void* stream, const ConvSecond* sec = nullptr) {
  ConvDims d;
  if (int rc = conv_dims(&d, n, h, wd, cin, cout)) return rc;
  if (d.wo != 32 || cout > 32 || cin < 4 || cin > 20 || d.ho % 4 != 0)
    return fail(EXPO_E_BADARG, "conv4x4s2_fwd_planes: 64-wide inputs of 4 .. 20 planes, at most 32 output channels");
  if (n == 0) return EXPO_OK;
  if (!x || !w || !y || (sec && (!sec->x || !sec->w || !sec->y || !sec->bias != !bias || !sec->zmask != !zmask)))
    return fail(EXPO_E_BADARG, "null pointer");
  const int rows = d.ho % 8 == 0 ? 8 : 4;  // (eight: 10.8 us for 14 planes at 64 images; four 11.8, two 15.3)
  const size_t lds = (size_t(2 * rows + 2) * (4 + 3 * d.w + 8) + size_t(32) * 68 + 512 + size_t(rows) * 96 + 32) * 4;
  const ConvSecond s2 = sec ? *sec : ConvSecond{};
  hipLaunchKernelGGL(conv_fwd_planes_kernel, dim3(unsigned(d.n * (d.ho / rows)), sec ? 2 : 1), dim3(64 * rows), lds,
                     static_cast<hipStream_t>(stream), x, w, bias, zmask, y, d, rows, act, leak, s2);
  HIP_TRY(hipGetLastError(), "conv4x4s2_fwd_planes launch");
  return EXPO_OK;
}

static int conv_bwd_data_impl(const float* dy, const float* w, const float* zmask, float* dx, int n, int h, int wd,
                              int cin, int cout, float leak, void* stream, const ConvSecond* sec = nullptr) {
  ConvDims d;
  if (int rc = conv_dims(&d, n, h, wd, cin, cout)) return rc;
  if (n == 0) return EXPO_OK;
  if (!dy || !w || !dx) return fail(EXPO_E_BADARG, "null pointer");
  if (cout % 4 != 0) return fail(EXPO_E_BADARG, "conv4x4s2_bwd_data: cout must be a multiple of 4");
  if (sec && (!sec->x || !sec->w || !sec->y || !sec->zmask != !zmask ||
              ((reinterpret_cast<uintptr_t>(sec->x) ^ reinterpret_cast<uintptr_t>(dy)) & 15) != 0))
    return fail(EXPO_E_BADARG, "conv4x4s2_bwd_data pair: the second problem needs the same operands, equally aligned");
  const unsigned gy2 = sec ? 2 : 1;
  const ConvSecond s2 = sec ? *sec : ConvSecond{};
  hipStream_t s = static_cast<hipStream_t>(stream);
  int ni = conv_tuning().nt.load();  // input-channel tiles per wave (0: the library's choice)
  const int forced_s = conv_tuning().slices.load();
  // the critic's first layer (6 input planes: a 32-wide matrix-core tile would be 81 % padding) on the vector ALUs
  // (conv_bwd_small_kernel), unless a probe forces a plan.  The value net's 17 planes took that kernel too until the flat
  // kernel's epilogue stopped serialising its loads (round 6): 37 against 56 us at batch 64, 102 against 149 at batch 192.
  const size_t small_lds = (size_t(6) * (d.wo + 2) * (cout + 4) + size_t(16) * cout * ((cin + 3) / 4 * 4)) * 4;
  if (ni == 0 && forced_s == 0 && cin == 6 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && d.ho % 4 == 0 &&
      small_lds <= 160 * 1024) {
    const dim3 grid(unsigned(d.n * (d.ho / 4)));
    static const hipError_t attr6 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bwd_small_kernel<6>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr6;
    hipLaunchKernelGGL(conv_bwd_small_kernel<6>, grid, dim3(256), small_lds, s, dy, w, zmask, dx, d, leak);
    if (sec)  // (the vector-ALU kernel takes no pair: two launches)
      hipLaunchKernelGGL(conv_bwd_small_kernel<6>, grid, dim3(256), small_lds, s, sec->x, sec->w, sec->zmask, sec->y, d, leak);
    HIP_TRY(hipGetLastError(), "conv4x4s2_bwd_data (small) launch");
    return EXPO_OK;
  }
  if (ni == 0) ni = d.cin >= 64 ? 2 : 1;  // (two tiles share the dY fragment: 2-10 % on the deeper layers)
  if (ni != 1 && ni != 2) ni = 1;
  if (d.cin <= 32) ni = 1;
  FlatPlan pl = bwd_plan(d, ni, forced_s);
  if (sec && forced_s == 0) {  // a pair: K' slices as for twice the pixels
    ConvDims d2 = d;
    d2.m *= 2;
    const FlatPlan p2 = bwd_plan(d2, ni, 0);
    pl.s = p2.s; pl.s2 = p2.s2; pl.rs = p2.rs;
  }
  const int nblocks = 4 * pl.tiles_m * pl.tiles_n;
  const size_t lds = pl.s > 1 ? size_t(pl.s) * ni * 4096 : 0;
  if (ni == 2) hipLaunchKernelGGL(conv_bwd_flat_kernel<2>, dim3(nblocks, gy2), dim3(64 * pl.s), lds, s, dy, w, zmask, dx, d, pl, leak, s2);
  else hipLaunchKernelGGL(conv_bwd_flat_kernel<1>, dim3(nblocks, gy2), dim3(64 * pl.s), lds, s, dy, w, zmask, dx, d, pl, leak, s2);
  HIP_TRY(hipGetLastError(), "conv4x4s2_bwd_data launch");
  return EXPO_OK;
}

// a block copy of the weight-gradient workspace: dW (cout x 16 cin floats) followed by the bias gradient (cout floats,
// padded to whole float4s)
static size_t wrw_copy_floats(const ConvDims& d) { return size_t(d.cout) * d.kdim + size_t((d.cout + 3) / 4 * 4); }

// `group` non-null: the reduce launch is DEFERRED -- the layer is appended to the group and expo_conv4x4s2_wrw_group issues
// one reduce launch for all of them
static int conv_wrw_impl(const float* x, const float* dy, float* dw, float* dbias, int bias_images, int n, int h, int wd,
                         int cin, int cout, void* workspace, size_t workspace_bytes, void* stream,
                         WrwReduceGroup* group = nullptr, WrwGroupArgs* main_group = nullptr, size_t* main_lds = nullptr) {
  ConvDims d;
  if (int rc = conv_dims(&d, n, h, wd, cin, cout)) return rc;
  if (!dw) return fail(EXPO_E_BADARG, "null pointer");
  if ((reinterpret_cast<uintptr_t>(dw) & 15) != 0) return fail(EXPO_E_BADARG, "conv4x4s2_wrw: dw must be 16-byte aligned");
  if (dbias && ((reinterpret_cast<uintptr_t>(dbias) & 15) != 0 || cout % 4 != 0))
    return fail(EXPO_E_BADARG, "conv4x4s2_wrw: dbias must be 16-byte aligned and cout a multiple of 4");
  if (bias_images < 0 || bias_images > n) return fail(EXPO_E_BADARG, "conv4x4s2_wrw: 0 <= bias_images <= n");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(dw, 0, size_t(cout) * d.kdim * 4, s), "conv4x4s2_wrw (empty batch)");
    if (dbias) HIP_TRY(hipMemsetAsync(dbias, 0, size_t(cout) * 4, s), "conv4x4s2_wrw (empty batch)");
    return EXPO_OK;
  }
  if (!x || !dy) return fail(EXPO_E_BADARG, "null pointer");
  if (d.wo & 1) return fail(EXPO_E_BADARG, "conv4x4s2_wrw: w / 2 must be even (pixels are consumed in pairs)");
  const WrwPlan pl = wrw_plan(d, conv_tuning().wrw_slices.load(), conv_tuning().parts.load());
  const size_t count = size_t(cout) * d.kdim, copy = wrw_copy_floats(d);
  if (pl.p > 1 && (!workspace || workspace_bytes < size_t(pl.p) * copy * 4 || (reinterpret_cast<uintptr_t>(workspace) & 15)))
    return fail(EXPO_E_BADARG, "conv4x4s2_wrw: workspace too small or not 16-byte aligned (expo_conv4x4s2_wrw_workspace_bytes)");
  float* const ws = static_cast<float*>(workspace);
  float* out = pl.p > 1 ? ws : dw;
  float* db_out = !dbias ? nullptr : (pl.p > 1 ? ws + count : dbias);
  const size_t stride = pl.p > 1 ? copy : 0;
  const int bias_pairs = bias_images * d.ho * (d.wo / 2);
  const dim3 grid(unsigned(pl.tiles_co * pl.tiles_k) * pl.p), block(64 * pl.s);
  const size_t lds = pl.s > 1 ? size_t(pl.s) * 16384 : 0;
  if (main_group) {  // the caller launches the stack's main kernels as one grid
    const int l = main_group->layers++;
    WrwGroupItem& it = main_group->it[l];
    it.x = x; it.dy = dy; it.out = out; it.db_out = db_out; it.stride = stride; it.d = d; it.pl = pl;
    it.bias_pairs = bias_pairs; it.cin4 = d.cin % 4 == 0;
    main_group->first_block[l + 1] = main_group->first_block[l] + grid.x;
    if (lds > *main_lds) *main_lds = lds;
  } else {
    if (d.cin % 4 == 0)
      hipLaunchKernelGGL(conv_wrw_kernel<true>, grid, block, lds, s, x, dy, out, stride, db_out, stride, bias_pairs, d, pl);
    else
      hipLaunchKernelGGL(conv_wrw_kernel<false>, grid, block, lds, s, x, dy, out, stride, db_out, stride, bias_pairs, d, pl);
    HIP_TRY(hipGetLastError(), "conv4x4s2_wrw launch");
  }
  if (pl.p > 1) {
    const size_t dw4 = count / 4, count4 = dbias ? copy / 4 : dw4;
    // (the copies' stride is `copy` floats either way: without a bias gradient the tail of a copy is not read)
    if (group) {
      const int l = group->layers++;
      group->ws[l] = ws; group->dw[l] = dw; group->dbias[l] = dbias;
      group->dw4[l] = unsigned(dw4); group->count4[l] = unsigned(count4); group->stride4[l] = unsigned(copy / 4);
      group->p[l] = unsigned(pl.p);
      return EXPO_OK;
    }
    hipLaunchKernelGGL(conv_wrw_reduce_kernel, dim3(unsigned((count4 + 15) / 16)), dim3(256), 0, s,
                       static_cast<const float*>(workspace), dw, dbias, dw4, count4, copy / 4, pl.p);
    HIP_TRY(hipGetLastError(), "conv4x4s2_wrw reduce launch");
  }
  return EXPO_OK;
}

// ---- the first FC layer behind the convolutions (critics.py:27-31, agent.py:33-35: `ly.fully_connected(flat, 128)`) -----
// y = x W^T with x [M][K] (K = 4096 features of a 4 x 4 x 256 map), W [N][K] (N = 128): 0.2 GFLOP at M = 192 that the
// library runs as one workgroup per 16 x 16 tile walking all of K -- 10 us whatever M is (profiles/r06_experiments.md),
// 14 calls per training iteration.  Here K is SPLIT: a wave = one 32 x 32 tile over K / (4 S) features (16-byte operand
// loads like the flat convolution: lane l -> row l & 31, k = 8 c + 4 (l >> 5) ..), the four waves of a block meet in LDS,
// and the S block partials leave as S slabs [S][M][N] that the consumer -- expo_critic_head_fwd / _bwd, which read the
// pre-activation exactly once -- adds in slab order together with the bias.  No atomics, a fixed summation order.
template <int CH>  // CH > 0: the wave's kw = 8 CH features are loaded in ONE burst (a single memory round trip); 0: any kw, pipelined
__global__ __launch_bounds__(256) void fc_fwd_slabs_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ slabs, int m, int n, int k, int tiles_n,
                                                           int kw) {
  __shared__ __attribute__((aligned(16))) float part[4][16][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int row = lane & 31, half = lane >> 5;
  const int am = tm * 32 + row, bn = tn * 32 + row;
  const bool a_ok = am < m, b_ok = bn < n;
  const int k0 = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(sl)) * kw + 4 * half;
  const int aoff = (a_ok ? am : 0) * k + k0, boff = (b_ok ? bn : 0) * k + k0;
  const __amdgpu_buffer_rsrc_t rx = conv_rsrc(x, size_t(m) * k);
  const __amdgpu_buffer_rsrc_t rw = conv_rsrc(w, size_t(n) * k);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  auto mma4 = [&](const float4& a, const float4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  };
  if constexpr (CH > 0) {
    float4 a[CH], b[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      a[c] = buf_load4(rx, aoff + c * 8, a_ok);
      b[c] = buf_load4(rw, boff + c * 8, b_ok);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) mma4(a[c], b[c]);  // (the compiler waits chunk by chunk: vmcnt counts down in issue order)
  } else {
    constexpr int U = 4;  // chunks of 8 k per group; the next group's loads are in flight under this group's MFMAs
    const int groups = kw / (8 * U);
    float4 ac[U], bc[U], an[U], bnx[U];
    auto load_group = [&](int g, float4 (&a)[U], float4 (&b)[U]) {
      const bool in = g < groups;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = buf_load4(rx, aoff + (g * U + u) * 8, a_ok && in);
        b[u] = buf_load4(rw, boff + (g * U + u) * 8, b_ok && in);
      }
    };
    load_group(0, ac, bc);
    for (int g = 0; g < groups; ++g) {
      load_group(g + 1, an, bnx);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) mma4(ac[u], bc[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) ac[u] = an[u], bc[u] = bnx[u];
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) part[sl][e][lane] = acc[e];
  __syncthreads();
  // wave sl adds accumulator slots 4 sl .. 4 sl + 3 of the four waves (in wave order) and stores them: slot e of lane l is
  // row (e & 3) + 8 (e >> 2) + 4 (l >> 5), column l & 31 of the tile -- 32 lanes write 128 contiguous bytes
  const __amdgpu_buffer_rsrc_t ry = conv_rsrc(slabs + size_t(blockIdx.y) * m * n, size_t(m) * n);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = 4 * sl + j;
    const float v = ((part[0][e][lane] + part[1][e][lane]) + part[2][e][lane]) + part[3][e][lane];
    const int mo = tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half, col = tn * 32 + row;
    buf_store1(ry, mo * n + col, mo < m && col < n, v);
  }
}

// Its data gradient with the activation gradient of the layer below in the epilogue (the convolutions' pattern):
//   gy[m][c] = (sum_j dh[m][j] W[j][c]) slope(z[m][c]),   dh [M][J] (J = 128), W [J][C] (C = 4096), z the top feature map.
// A wave = one 32 x 32 tile over all of J: A fragments are 16-byte loads of dh rows, B fragments dword loads of W rows (32
// lanes cover 128 contiguous bytes of row j = 8 c + 4 (l >> 5) + i).  The library's GEMM + the separate lrelu_bwd launch
// took 10 + 4.4 us at M = 192.
__global__ __launch_bounds__(256) void fc_bwd_data_mask_kernel(const float* __restrict__ dh, const float* __restrict__ w,
                                                               const float* __restrict__ z, float* __restrict__ gy, int m,
                                                               int jdim, int c, int tiles_c, float leak) {
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(sl);
  const int tm = tile / tiles_c, tc = tile - tm * tiles_c;  // (the four waves of a block: neighbouring column tiles, same dh rows)
  const int row = lane & 31, half = lane >> 5;
  const int am = tm * 32 + row, col = tc * 32 + row;
  const bool a_ok = am < m, c_ok = col < c;
  const __amdgpu_buffer_rsrc_t ra = conv_rsrc(dh, size_t(m) * jdim);
  const __amdgpu_buffer_rsrc_t rw = conv_rsrc(w, size_t(jdim) * c);
  const int aoff = (a_ok ? am : 0) * jdim + 4 * half, boff = 4 * half * c + (c_ok ? col : 0);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  constexpr int U = 2;
  const int groups = jdim / (8 * U);
  float4 ac[U], an[U];
  float bc[U][4], bnx[U][4];
  auto load_group = [&](int g, float4 (&a)[U], float (&b)[U][4]) {
    const bool in = g < groups;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ch = g * U + u;
      a[u] = buf_load4(ra, aoff + ch * 8, a_ok && in);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[u][i] = buf_load1(rw, boff + (ch * 8 + i) * c, c_ok && in);
    }
  };
  load_group(0, ac, bc);
  for (int g = 0; g < groups; ++g) {
    load_group(g + 1, an, bnx);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].x, bc[u][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].y, bc[u][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].z, bc[u][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u].w, bc[u][3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ac[u] = an[u];
#pragma unroll
      for (int i = 0; i < 4; ++i) bc[u][i] = bnx[u][i];
    }
  }
  const __amdgpu_buffer_rsrc_t rz = conv_rsrc(z, size_t(m) * c);
  const __amdgpu_buffer_rsrc_t ry = conv_rsrc(gy, size_t(m) * c);
  float zv[16];
  int off[16];
  bool ok[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int mo = tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
    ok[e] = mo < m && c_ok;
    off[e] = mo * c + col;
    zv[e] = buf_load1(rz, off[e], ok[e]);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) buf_store1(ry, off[e], ok[e], acc[e] * lrelu_slope_v(zv[e], leak));
}

// Its weight gradient: dW[j][c] = sum_m dh[m][j] x[m][c] (J = 128 rows of C = 4096): a GEMM whose K dimension is the BATCH
// (64 .. 192 rows) with a large output -- the library picks 128 x 128 tiles for it at m = 64 (32 workgroups: 13.5 us) and
// 32 x 128 at m = 192 (9 us).  Here a wave = one 32 x 32 tile of dW over a quarter of the rows; both operands are dword
// loads of 128 contiguous bytes per row (lane l -> row 2 p + (l >> 5), column l & 31), issued in groups of 16 before their
// MFMAs; the four waves of a block meet in LDS and are added in wave order.
__global__ __launch_bounds__(256) void fc_wrw_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                     float* __restrict__ dw, int m, int jdim, int c, int tiles_c, int kw) {
  __shared__ __attribute__((aligned(16))) float part[4][16][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int tj = blockIdx.x / tiles_c, tc = blockIdx.x - tj * tiles_c;
  const int col = lane & 31, half = lane >> 5;
  const int aj = tj * 32 + col, bc = tc * 32 + col;
  const bool a_ok = aj < jdim, b_ok = bc < c;
  const int r0 = __builtin_amdgcn_readfirstlane(sl) * kw + half;  // this lane's first row; it walks every second one
  const __amdgpu_buffer_rsrc_t ra = conv_rsrc(dh, size_t(m) * jdim);
  const __amdgpu_buffer_rsrc_t rb = conv_rsrc(x, size_t(m) * c);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  constexpr int U = 8;  // row pairs per group
  const int groups = (kw / 2 + U - 1) / U;
  float ac[U], bcur[U], an[U], bnx[U];
  auto load_group = [&](int g, float (&a)[U], float (&b)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = g * U + u, row = r0 + 2 * p;
      const bool in = g < groups && 2 * p < kw && row < m;
      a[u] = buf_load1(ra, row * jdim + aj, a_ok && in);
      b[u] = buf_load1(rb, row * c + bc, b_ok && in);
    }
  };
  load_group(0, ac, bcur);
  for (int g = 0; g < groups; ++g) {
    load_group(g + 1, an, bnx);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bcur[u], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) ac[u] = an[u], bcur[u] = bnx[u];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) part[sl][e][lane] = acc[e];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t ry = conv_rsrc(dw, size_t(jdim) * c);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = 4 * sl + q;
    const float v = ((part[0][e][lane] + part[1][e][lane]) + part[2][e][lane]) + part[3][e][lane];
    const int jo = tj * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
    buf_store1(ry, jo * c + bc, jo < jdim && b_ok, v);
  }
}

}  // namespace expo

using namespace expo;

extern "C" {

int expo_fc_wrw(const float* dh, const float* x, float* dw, int m, int j, int c, void* stream) {
  if (m < 0 || j <= 0 || c <= 0) return fail(EXPO_E_BADARG, "fc_wrw: m >= 0, j > 0, c > 0 required");
  if (!dw || (m > 0 && (!dh || !x))) return fail(EXPO_E_BADARG, "null pointer");
  if (size_t(m) * c >= (1ull << 29) || size_t(j) * c >= (1ull << 29) || size_t(m) * j >= (1ull << 29))
    return fail(EXPO_E_BADARG, "fc_wrw: operands of at most 2 GiB");
  const int tiles_j = (j + 31) / 32, tiles_c = (c + 31) / 32;
  const int kw = ((m + 3) / 4 + 1) & ~1;  // rows per wave: a quarter of the batch, even (m = 0: the tiles are zero-filled)
  hipLaunchKernelGGL(fc_wrw_kernel, dim3(tiles_j * tiles_c), dim3(256), 0, static_cast<hipStream_t>(stream), dh, x, dw, m, j, c,
                     tiles_c, kw);
  HIP_TRY(hipGetLastError(), "fc_wrw launch");
  return EXPO_OK;
}

int expo_fc_fwd_slabs_count(int m, int k) {
  if (m <= 0 || k <= 0) return 0;
  int s = (m + 31) / 32 >= 6 ? 8 : 16;  // ~200 blocks of four waves for 128 columns
  while (s >= 1 && k % (128 * s) != 0) s >>= 1;  // a wave walks whole groups of 32 features
  return s;
}

int expo_fc_fwd_slabs(const float* x, const float* w, float* slabs, int m, int n, int k, void* stream) {
  if (m < 0 || n <= 0 || k <= 0) return fail(EXPO_E_BADARG, "fc_fwd_slabs: m >= 0, n > 0, k > 0 required");
  if (m == 0) return EXPO_OK;
  if (!x || !w || !slabs) return fail(EXPO_E_BADARG, "null pointer");
  const int s = expo_fc_fwd_slabs_count(m, k);
  if (s == 0) return fail(EXPO_E_BADARG, "fc_fwd_slabs: k must be a multiple of 128 (whole groups of 32 features per wave)");
  if (size_t(m) * k >= (1ull << 29) || size_t(n) * k >= (1ull << 29) || size_t(m) * n >= (1ull << 29))
    return fail(EXPO_E_BADARG, "fc_fwd_slabs: operands of at most 2 GiB");
  const int tiles_m = (m + 31) / 32, tiles_n = (n + 31) / 32;
  const int kw = k / (s * 4);
  const dim3 grid(tiles_m * tiles_n, s);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kw == 128) hipLaunchKernelGGL(fc_fwd_slabs_kernel<16>, grid, dim3(256), 0, st, x, w, slabs, m, n, k, tiles_n, kw);
  else if (kw == 64) hipLaunchKernelGGL(fc_fwd_slabs_kernel<8>, grid, dim3(256), 0, st, x, w, slabs, m, n, k, tiles_n, kw);
  else if (kw == 32) hipLaunchKernelGGL(fc_fwd_slabs_kernel<4>, grid, dim3(256), 0, st, x, w, slabs, m, n, k, tiles_n, kw);
  else hipLaunchKernelGGL(fc_fwd_slabs_kernel<0>, grid, dim3(256), 0, st, x, w, slabs, m, n, k, tiles_n, kw);
  HIP_TRY(hipGetLastError(), "fc_fwd_slabs launch");
  return EXPO_OK;
}

int expo_fc_bwd_data_mask(const float* dh, const float* w, const float* z, float* gy, int m, int j, int c, float leak,
                          void* stream) {
  if (m < 0 || j <= 0 || c <= 0) return fail(EXPO_E_BADARG, "fc_bwd_data_mask: m >= 0, j > 0, c > 0 required");
  if (m == 0) return EXPO_OK;
  if (!dh || !w || !z || !gy) return fail(EXPO_E_BADARG, "null pointer");
  if (j % 16 != 0) return fail(EXPO_E_BADARG, "fc_bwd_data_mask: the hidden width must be a multiple of 16");
  if (size_t(m) * c >= (1ull << 29) || size_t(j) * c >= (1ull << 29)) return fail(EXPO_E_BADARG, "fc_bwd_data_mask: operands of at most 2 GiB");
  const int tiles_m = (m + 31) / 32, tiles_c = (c + 31) / 32;
  if (tiles_c % 4 != 0) return fail(EXPO_E_BADARG, "fc_bwd_data_mask: the feature count must be a multiple of 128");
  hipLaunchKernelGGL(fc_bwd_data_mask_kernel, dim3(tiles_m * tiles_c / 4), dim3(256), 0, static_cast<hipStream_t>(stream), dh, w, z,
                     gy, m, j, c, tiles_c, leak);
  HIP_TRY(hipGetLastError(), "fc_bwd_data_mask launch");
  return EXPO_OK;
}

int expo_conv4x4s2_fwd(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cin,
                       int cout, int act, float leak, void* stream) {
  return conv_fwd_impl(x, w, bias, nullptr, y, n, h, wd, cin, cout, act, leak, stream);
}

int expo_conv4x4s2_fwd_mask(const float* x, const float* w, const float* zmask, float* y, int n, int h, int wd, int cin,
                            int cout, float leak, void* stream) {
  if (!zmask) return fail(EXPO_E_BADARG, "null pointer");
  return conv_fwd_impl(x, w, nullptr, zmask, y, n, h, wd, cin, cout, 0, leak, stream);
}

int expo_conv4x4s2_fwd_pair(const float* x_a, const float* w_a, const float* bias_a, float* y_a, const float* x_b,
                            const float* w_b, const float* bias_b, float* y_b, int n, int h, int wd, int cin, int cout, int act,
                            float leak, void* stream) {
  const ConvSecond sec{x_b, w_b, bias_b, nullptr, y_b};
  return conv_fwd_impl(x_a, w_a, bias_a, nullptr, y_a, n, h, wd, cin, cout, act, leak, stream, &sec);
}

int expo_conv4x4s2_fwd_planes(const float* x, const float* w, const float* bias, const float* zmask, float* y, int n, int h,
                              int wd, int cin, int cout, int act, float leak, void* stream) {
  return conv_fwd_planes_impl(x, w, bias, zmask, y, n, h, wd, cin, cout, act, leak, stream);
}

int expo_conv4x4s2_fwd_planes_pair(const float* x_a, const float* w_a, const float* bias_a, float* y_a, const float* x_b,
                                   const float* w_b, const float* bias_b, float* y_b, int n, int h, int wd, int cin, int cout,
                                   int act, float leak, void* stream) {
  const ConvSecond sec{x_b, w_b, bias_b, nullptr, y_b};
  return conv_fwd_planes_impl(x_a, w_a, bias_a, nullptr, y_a, n, h, wd, cin, cout, act, leak, stream, &sec);
}

int expo_conv4x4s2_bwd_data_mask_pair(const float* dy_a, const float* w_a, const float* zmask_a, float* dx_a, const float* dy_b,
                                      const float* w_b, const float* zmask_b, float* dx_b, int n, int h, int wd, int cin,
                                      int cout, float leak, void* stream) {
  if (!zmask_a || !zmask_b) return fail(EXPO_E_BADARG, "null pointer");
  const ConvSecond sec{dy_b, w_b, nullptr, zmask_b, dx_b};
  return conv_bwd_data_impl(dy_a, w_a, zmask_a, dx_a, n, h, wd, cin, cout, leak, stream, &sec);
}

int expo_conv_tuning(int tile, int nt, int slices) {
  // probes / tests: override the decomposition of the convolution kernels (negative: leave as is; 0: the library's choice)
  if (tile > 6 || nt > 2 || (slices > 0 && !pow2_upto(slices, 16)))
    return fail(EXPO_E_BADARG, "conv tuning: tile <= 5, nt <= 2, slices in {1, 2, 4, 8, 16}");
  if (tile >= 0) conv_tuning().tile.store(tile);
  if (nt >= 0) conv_tuning().nt.store(nt);
  if (slices >= 0) conv_tuning().slices.store(slices);
  return EXPO_OK;
}

int expo_conv4x4s2_bwd_data(const float* dy, const float* w, float* dx, int n, int h, int wd, int cin, int cout,
                            void* stream) {
  return conv_bwd_data_impl(dy, w, nullptr, dx, n, h, wd, cin, cout, 0.f, stream);
}

int expo_conv4x4s2_bwd_data_mask(const float* dy, const float* w, const float* zmask, float* dx, int n, int h, int wd,
                                 int cin, int cout, float leak, void* stream) {
  if (!zmask) return fail(EXPO_E_BADARG, "null pointer");
  return conv_bwd_data_impl(dy, w, zmask, dx, n, h, wd, cin, cout, leak, stream);
}

int expo_conv_wrw_tuning(int slices, int parts) {
  // probes / tests: waves per block (1 .. 4) and blocks per tile of the weight-gradient kernel (negative: leave; 0: auto)
  if (slices > 4) return fail(EXPO_E_BADARG, "conv wrw tuning: slices <= 4");
  if (slices >= 0) conv_tuning().wrw_slices.store(slices);
  if (parts >= 0) conv_tuning().parts.store(parts);
  return EXPO_OK;
}

size_t expo_conv4x4s2_wrw_workspace_bytes(int n, int h, int wd, int cin, int cout) {
  ConvDims d;
  if (conv_dims(&d, n, h, wd, cin, cout) != EXPO_OK || n == 0) return 0;
  const WrwPlan pl = wrw_plan(d, conv_tuning().wrw_slices.load(), conv_tuning().parts.load());
  return pl.p > 1 ? size_t(pl.p) * wrw_copy_floats(d) * 4 : 0;
}

int expo_conv4x4s2_wrw(const float* x, const float* dy, float* dw, int n, int h, int wd, int cin, int cout,
                       void* workspace, size_t workspace_bytes, void* stream) {
  return conv_wrw_impl(x, dy, dw, nullptr, 0, n, h, wd, cin, cout, workspace, workspace_bytes, stream);
}

int expo_conv4x4s2_wrw_bias(const float* x, const float* dy, float* dw, float* dbias, int bias_images, int n, int h,
                            int wd, int cin, int cout, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dbias) return fail(EXPO_E_BADARG, "null pointer");
  return conv_wrw_impl(x, dy, dw, dbias, bias_images, n, h, wd, cin, cout, workspace, workspace_bytes, stream);
}

int expo_conv4x4s2_wrw_group(int count, const float* const* x, const float* const* dy, float* const* dw,
                             float* const* dbias, const int* bias_images, const int* n, const int* h, const int* wd,
                             const int* cin, const int* cout, void* const* workspace, const size_t* workspace_bytes,
                             void* stream) {
  if (count < 0 || count > kWrwGroupMax) return fail(EXPO_E_BADARG, "conv4x4s2_wrw_group: 0 <= count <= 8 layers");
  if (count == 0) return EXPO_OK;
  if (!x || !dy || !dw || !dbias || !bias_images || !n || !h || !wd || !cin || !cout || !workspace || !workspace_bytes)
    return fail(EXPO_E_BADARG, "null pointer");
  WrwReduceGroup g;
  g.layers = 0;
  WrwGroupArgs mg;
  mg.layers = 0;
  mg.first_block[0] = 0;
  size_t lds = 0;
  for (int l = 0; l < count; ++l)
    if (int rc = conv_wrw_impl(x[l], dy[l], dw[l], dbias[l], bias_images[l], n[l], h[l], wd[l], cin[l], cout[l],
                               workspace[l], workspace_bytes[l], stream, &g, &mg, &lds))
      return rc;
  if (mg.layers > 0) {  // (empty batches were zero-filled by their own calls)
    hipLaunchKernelGGL(conv_wrw_group_kernel, dim3(mg.first_block[mg.layers]), dim3(256), lds,
                       static_cast<hipStream_t>(stream), mg);
    HIP_TRY(hipGetLastError(), "conv4x4s2_wrw_group launch");
  }
  if (g.layers == 0) return EXPO_OK;  // every layer fitted one block copy: written in place
  unsigned blocks = 0;
  for (int l = 0; l < g.layers; ++l) {
    unsigned gl = 0;
    while (gl < 4 && (1u << gl) < g.p[l]) ++gl;  // groups = the power of two that holds P copies, at most 16
    g.g_log2[l] = gl;
    const unsigned per = 256u >> gl;
    g.first_block[l] = blocks;
    blocks += (g.count4[l] + per - 1) / per;
  }
  g.first_block[g.layers] = blocks;
  hipLaunchKernelGGL(conv_wrw_reduce_group_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g);
  HIP_TRY(hipGetLastError(), "conv4x4s2_wrw_group reduce launch");
  return EXPO_OK;
}

}  // extern "C"
