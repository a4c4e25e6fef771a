// filter_math.h -- per-pixel maths of the eight Exposure filters, forward and backward,
// written for gfx950 wave64 VALU (fp32 arithmetic, hardware transcendentals).
//
// Each functor restates one `process()` of /root/reference/filters.py and the gradient
// TF-1 derives for it (tie conventions: SURVEY.md section 8a; oracle: oracle/filters_np.py).
//
//   NP    packed parameters per image (C-ABI layout, include/exposure_hip.h)
//   NACC  per-thread fp32 accumulators the backward keeps; they are *linear* in the
//         per-pixel contributions, so block sums can be finished (finish_one) and added
//         atomically across blocks.
//   Prm   per-image derived constants; loaded through a block-uniform pointer, so the
//         compiler keeps them in SGPRs (scalar loads) -- no LDS staging is needed for
//         wave-uniform data on CDNA.
#pragma once
#include <hip/hip_runtime.h>

namespace expo {

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_sin_f32 / v_cos_f32 take their argument in revolutions (x * 2*pi radians)
__device__ __forceinline__ float sin_rev(float x) { return __builtin_amdgcn_sinf(x); }
__device__ __forceinline__ float cos_rev(float x) { return __builtin_amdgcn_cosf(x); }
__device__ __forceinline__ float clamp01x(float x, float lo, float hi) {
  return __builtin_amdgcn_fmed3f(x, lo, hi);
}

constexpr float kLn2 = 0.6931471805599453f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kLumR = 0.27f, kLumG = 0.67f, kLumB = 0.06f;  // util.py:271-274
constexpr int kCurveSteps = 8;                                 // config_example.py:27
struct alignas(8) float2_lut { float x, y; };

// util.py:271-274 -- same association order as the reference: (.27 r + .67 g) + .06 b
__device__ __forceinline__ float lum3(const float x[3]) {
  return (kLumR * x[0] + kLumG * x[1]) + kLumB * x[2];
}

// ---------------------------------------------------------------------------------
// 0  ExposureFilter   filters.py:181-182   y = x * exp(p ln2)
// ---------------------------------------------------------------------------------
struct ExposureF {
  static constexpr int NP = 1, NACC = 1, kLutFloats = 0;
  static constexpr bool kHasGroupBwd = false;
  struct Prm { float s; };
  __device__ static Prm load(const float* __restrict__ p) { return {exp2f(p[0])}; }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = x[c] * q.s;
  }
  __device__ static void bwd(const Prm& q, const float*, const float x[3], const float dy[3], float dx[3],
                             float acc[NACC], int) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dx[c] = dy[c] * q.s;
      acc[0] = fmaf(dy[c], x[c], acc[0]);
    }
  }
  // dp = ln2 * sum dy*y = ln2 * s * sum dy*x
  __device__ static float finish_one(const float* __restrict__ p, const float* a, int) { return kLn2 * exp2f(p[0]) * a[0]; }
};

// ---------------------------------------------------------------------------------
// 1  GammaFilter   filters.py:205-206   y = pow(max(x, 0.001), g)
// ---------------------------------------------------------------------------------
struct GammaF {
  static constexpr int NP = 1, NACC = 1, kLutFloats = 0;
  static constexpr bool kHasGroupBwd = false;
  struct Prm { float g; };
  __device__ static Prm load(const float* __restrict__ p) { return {p[0]}; }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = fast_exp2(q.g * fast_log2(fmaxf(x[c], 0.001f)));
  }
  __device__ static void bwd(const Prm& q, const float*, const float x[3], const float dy[3], float dx[3],
                             float acc[NACC], int) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xm = fmaxf(x[c], 0.001f);
      const float lg = fast_log2(xm);
      const float y = fast_exp2(q.g * lg);
      const float t = dy[c] * y;
      // tf.maximum passes the gradient to x on equality (x >= 0.001)
      dx[c] = (x[c] >= 0.001f) ? t * q.g * fast_rcp(xm) : 0.0f;
      acc[0] = fmaf(t, lg, acc[0]);
    }
  }
  __device__ static float finish_one(const float* __restrict__, const float* a, int) { return kLn2 * a[0]; }
};

// ---------------------------------------------------------------------------------
// 2  ImprovedWhiteBalanceFilter   filters.py:237-238   y_c = x_c * s_c
// ---------------------------------------------------------------------------------
struct WhiteBalanceF {
  static constexpr int NP = 3, NACC = 3, kLutFloats = 0;
  static constexpr bool kHasGroupBwd = false;
  struct Prm { float s[3]; };
  __device__ static Prm load(const float* __restrict__ p) { return {{p[0], p[1], p[2]}}; }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = x[c] * q.s[c];
  }
  __device__ static void bwd(const Prm& q, const float*, const float x[3], const float dy[3], float dx[3],
                             float acc[NACC], int) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dx[c] = dy[c] * q.s[c];
      acc[c] = fmaf(dy[c], x[c], acc[c]);
    }
  }
  __device__ static float finish_one(const float* __restrict__, const float* a, int j) { return a[j]; }
};

// ---------------------------------------------------------------------------------
// 3  SaturationPlusFilter   filters.py:484-498
//    xc = min(x,1); (h,s,v) = rgb_to_hsv(xc); s' = s + (1-s)(.5-|.5-v|).8;
//    full = hsv_to_rgb(h,s',v); y = xc(1-p) + full p
//    The hue is never materialised: for rng = v - min > 0 the HSV->RGB ramp values are
//    d_c = (xc_c - min)/rng (hue only encodes the position of the middle channel), and
//    for rng == 0 TF's hue is 0, i.e. d = (1,0,0).  full_c = ((1-s') + s' d_c) v.
//    Round 4: full_c - xc_c is formed WITHOUT subtracting two nearly equal colours.  With xc_c = mn + rng d_c,
//      full_c - xc_c = (rng - s' v) (1 - d_c) =: K (1 - d_c),   1 - d_c = (v - xc_c) / rng   (rng = 0: (0, 1, 1)),
//    and for v > 0, s' v = rng + 0.8 mn tri (tri = .5 - |.5 - v|; (1 - s) v = mn), so K = -0.8 mn tri: a product of
//    values formed exactly from the inputs.  (v <= 0: s = 0, tri = v, K = rng - 0.8 v^2.)  The forward is
//    y_c = xc_c + p K (1 - d_c) and the parameter gradient sum dy_c K (1 - d_c): the fp32 terms are good to a few
//    ulps of THEMSELVES -- `full - xc` in fp32 was good to an ulp of the colour, 10-100x the term on dark pixels
//    (parameter-gradient error 8e-6 of the sum of absolute terms, r04p1).
// ---------------------------------------------------------------------------------
struct SatPlusF {
  static constexpr int NP = 1, NACC = 1, kLutFloats = 0;
  static constexpr bool kHasGroupBwd = false;
  struct Prm { float p; };
  __device__ static Prm load(const float* __restrict__ p) { return {p[0]}; }

  struct Hsv { float xc[3], v, mn, rng, s, sp, d[3], K, omd[3]; };
  __device__ static Hsv analyse(const float x[3]) {
    Hsv a;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.xc[c] = fminf(x[c], 1.0f);
    a.v = fmaxf(fmaxf(a.xc[0], a.xc[1]), a.xc[2]);
    a.mn = fminf(fminf(a.xc[0], a.xc[1]), a.xc[2]);
    a.rng = a.v - a.mn;
    a.s = (a.v > 0.0f) ? a.rng * fast_rcp(a.v) : 0.0f;
    const float tri = 0.5f - fabsf(0.5f - a.v);
    a.sp = a.s + (1.0f - a.s) * tri * 0.8f;
    const bool col = a.rng > 0.0f;
    const float ir = col ? fast_rcp(a.rng) : 0.0f;
    a.d[0] = col ? (a.xc[0] - a.mn) * ir : 1.0f;
    a.d[1] = (a.xc[1] - a.mn) * ir;
    a.d[2] = (a.xc[2] - a.mn) * ir;
    // full_c - xc_c = K (1 - d_c), cancellation-free (header comment)
    a.K = (a.v > 0.0f) ? -0.8f * a.mn * tri : a.rng - 0.8f * a.v * a.v;
    a.omd[0] = col ? (a.v - a.xc[0]) * ir : 0.0f;
    a.omd[1] = col ? (a.v - a.xc[1]) * ir : 1.0f;
    a.omd[2] = col ? (a.v - a.xc[2]) * ir : 1.0f;
    return a;
  }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
    const Hsv a = analyse(x);
    const float pk = q.p * a.K;
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = fmaf(pk, a.omd[c], a.xc[c]);
  }
  __device__ static void bwd(const Prm& q, const float*, const float x[3], const float dy[3], float dx[3],
                             float acc[NACC], int mode) {
    const Hsv a = analyse(x);
    const float oms = 1.0f - a.sp;
    float gfull[3] = {0.f, 0.f, 0.f};
    // dp = sum_c dy_c (full_c - xc_c) = K sum_c dy_c (1 - d_c)
    acc[0] = fmaf(a.K, fmaf(dy[0], a.omd[0], fmaf(dy[1], a.omd[1], dy[2] * a.omd[2])), acc[0]);
#pragma unroll
    for (int c = 0; c < 3; ++c) gfull[c] = dy[c] * q.p;
    float dxc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dxc[c] = dy[c] * (1.0f - q.p);
    if (mode == 1 && a.rng > 0.0f && a.v > 0.0f) {
      // analytic d full / d xc (oracle/filters_np.py::_satplus_full_grad)
      int imax = 0, imin = 0;
      if (a.xc[1] > a.xc[imax]) imax = 1;
      if (a.xc[2] > a.xc[imax]) imax = 2;
      if (a.xc[1] < a.xc[imin]) imin = 1;
      if (a.xc[2] < a.xc[imin]) imin = 2;
      const float iv = 1.0f / a.v, ir = 1.0f / a.rng;
      const float tri = 0.5f - fabsf(0.5f - a.v);
      const float dtri = (0.5f - a.v) > 0.f ? 1.f : ((0.5f - a.v) < 0.f ? -1.f : 0.f);
      const float gsum = gfull[0] + gfull[1] + gfull[2];
      const float gd = gfull[0] * a.d[0] + gfull[1] * a.d[1] + gfull[2] * a.d[2];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float amax = (j == imax) ? 1.f : 0.f, amin = (j == imin) ? 1.f : 0.f;
        const float ds = (amax - amin) * iv - a.rng * iv * iv * amax;
        const float dsp = ds * (1.0f - tri * 0.8f) + (1.0f - a.s) * 0.8f * dtri * amax;
        float o = amax * oms * gsum - a.v * dsp * gsum + dsp * a.v * gd + a.sp * amax * gd;
        o += a.sp * a.v * ir * (gfull[j] - amin * gsum - (amax - amin) * gd);
        dxc[j] += o;
      }
    }
    // tf.minimum(img, 1.0) passes the gradient on x <= 1
#pragma unroll
    for (int c = 0; c < 3; ++c) dx[c] = (x[c] <= 1.0f) ? dxc[c] : 0.0f;
  }
  __device__ static float finish_one(const float* __restrict__, const float* a, int) { return a[0]; }
};

// ---------------------------------------------------------------------------------
// 4 / 7  ToneFilter (filters.py:312-322) and ColorFilter (filters.py:264-273):
//    y = (L/S) sum_i clip(x - i/L, 0, 1/L) k_i,  S = sum_i k_i + 1e-30, L = 8.
//    Tone shares one curve over the channels (NC = 1), Color has one per channel.
//
//    Forward: with E_i = clamp(x, 0, i/L) (E_0 = 0) the clips telescope,
//      clip_i = E_{i+1} - E_i  =>  T = sum_i k_i clip_i = sum_{i=1..L} E_i (k_{i-1} - k_i),  k_L := 0
//    i.e. 8 v_min + 8 v_fma per element with the knot differences in SGPRs.
//
//    Backward: per-curve LUT staged in LDS, entry j = (L/S) {k_j, k_{j+1}} (the two slopes dx needs):
//      j = clamp(ceil(L x) - 1, 0, L-1)      (x in (j/L, (j+1)/L] -> segment j)
//      dT/dx = [0 <= x <= 1] k_j + [L x integer, 1 <= L x <= L-1] k_{j+1}
//    (tf.clip_by_value passes the gradient on BOTH inclusive bounds, so exactly on a
//    knot the two neighbouring segments both contribute -- SURVEY.md section 8a-6).
//    Accumulators per curve: Q_i = sum dy E_i (i = 1..L) only.  sum dy clip_i = Q_{i+1} - Q_i, and
//    B = sum dy y needs no per-element work either: y = (L/S) sum_i E_i (k_{i-1} - k_i), so
//    B = (L/S) sum_i (k_{i-1} - k_i) Q_i.   dk_i = (L/S)(Q_{i+1} - Q_i) - B/S.
// ---------------------------------------------------------------------------------
template <int NC>
struct CurveF {
  static constexpr int L = kCurveSteps;
  static constexpr int NP = NC * L, NACC = NC * L;
  static constexpr int kLutFloats = NC * 256 * 2;  // fp16 path: 256-entry bit-pattern table per curve
  struct Prm { float delta[NC][L]; float scale[NC]; };  // delta[c][i-1] = k_{i-1} - k_i
  __device__ static Prm load(const float* __restrict__ p) {
    Prm q;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float S = 0.f;
#pragma unroll
      for (int i = 0; i < L; ++i) {
        const float k = p[c * L + i];
        q.delta[c][i] = k - ((i + 1 < L) ? p[c * L + i + 1] : 0.0f);
        S += k;
      }
      S += 1e-30f;
      q.scale[c] = float(L) / S;
    }
    return q;
  }
  // LDS LUT for the backward; call with all threads of the block, then __syncthreads().
  __device__ static void stage(const float* __restrict__ p, float* lut) {
    const int t = threadIdx.x;
    if (t < NC * L) {
      const int c = t / L, j = t % L;
      float S = 0.f;
      for (int i = 0; i < L; ++i) S += p[c * L + i];
      S += 1e-30f;
      const float scale = float(L) / S;  // entries are pre-scaled: dx = dy * (L/S) * slope
      lut[t * 2 + 0] = scale * p[c * L + j];
      lut[t * 2 + 1] = (j + 1 < L) ? scale * p[c * L + j + 1] : 0.0f;
    }
  }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int cc = (NC == 1) ? 0 : c;
      const float xc = clamp01x(x[c], 0.0f, 1.0f);
      float t = xc * q.delta[cc][L - 1];  // E_L = clamp(x,0,1)
#pragma unroll
      for (int i = 1; i < L; ++i) t = fmaf(fminf(xc, float(i) / L), q.delta[cc][i - 1], t);
      y[c] = t * q.scale[cc];
    }
  }
  __device__ static void bwd(const Prm& q, const float* lut, const float x[3], const float dy[3],
                             float dx[3], float acc[NACC], int) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int cc = (NC == 1) ? 0 : c;
      const float xv = x[c], g = dy[c];
      const float xc = clamp01x(xv, 0.0f, 1.0f);
      dx[c] = g * lut_slope(lut, cc, xv, xc);
      float* a = acc + cc * L;
#pragma unroll
      for (int i = 1; i < L; ++i) a[i - 1] = fmaf(g, fminf(xc, float(i) / L), a[i - 1]);
      a[L - 1] = fmaf(g, xc, a[L - 1]);
    }
  }
  // (L/S) dT/dx from the LDS LUT, with TF's inclusive clip gradient (both neighbours on a knot)
  __device__ static float lut_slope(const float* lut, int cc, float xv, float xc) {
    const float u = xc * float(L);  // exact; the clamped value indexes the same segment as x
    const float cu = ceilf(u);
    const float jf = clamp01x(cu - 1.0f, 0.0f, float(L - 1));
    const float2_lut e = *reinterpret_cast<const float2_lut*>(lut + (cc * L + int(jf)) * 2);
    // inside: 0 <= x <= 1  <=>  clamp(x) == x.   knot: L x^ is an integer in [1, L-1]; the test
    // (cu - 1 == jf) accepts [1, L] and the LUT's k_L := 0 makes L x^ == L contribute 0.
    const bool inside = (xc == xv);
    const bool knot = (cu == u) && (cu - 1.0f == jf);
    return inside ? (e.x + (knot ? e.y : 0.0f)) : 0.0f;
  }
  // ---- fp16 storage: slope table indexed by the HIGH BYTE of the fp16 bit pattern -------------
  // Every knot i/8 has a zero low byte (0x3000, 0x3400, 0x3600, 0x3800, 0x3900, 0x3A00, 0x3B00,
  // 0x3C00), so all fp16 values that share a high byte h and have a NON-zero low byte lie strictly
  // inside one segment; only the value h<<8 itself can sit on a knot (or be +-0, 1.0).  Entry h =
  // {slope for low byte != 0, slope for low byte == 0}, both evaluated with TF's inclusive rule on a
  // representative value -- this also encodes "outside [0,1] -> 0", "-0.0 passes", "1.0 passes".
  // Per element: byte extract + ds_read_b64 + one select, instead of ceil/clamp/convert/compare chains.
  // (L/S) dT/dx at x with TF's rule -- segment i contributes k_i iff 0 <= x - i/L <= 1/L, both bounds
  // inclusive (tf.clip_by_value gradient) -- in closed form: with u = L x exact, that is segment
  // floor(u) when it exists, plus segment u-1 when u is an integer >= 1.  -0.0 passes like +0.0,
  // NaN and everything outside [0, 1] give 0.  (The table is rebuilt by every block, so this is
  // written to be cheap: ~15 instructions instead of a loop over the L segments.)
  __device__ static float slope_closed(const float* __restrict__ k, float scale, float x) {
    const float u = x * float(L);
    if (!(u >= 0.0f && u <= float(L))) return 0.0f;
    const float jf = floorf(u);
    const int j = int(jf);  // 0..L
    float sl = (j < L) ? k[j] : 0.0f;
    if (jf == u && j >= 1) sl = k[j - 1] + sl;
    return scale * sl;
  }
  __device__ static void stage16(const float* __restrict__ p, float* lut) {
    typedef _Float16 half_s;
    for (int t = threadIdx.x; t < NC * 256; t += blockDim.x) {
      const int c = t >> 8, h = t & 255;
      const float* k = p + c * L;
      float S = 0.f;
      for (int i = 0; i < L; ++i) S += k[i];
      S += 1e-30f;
      const float scale = float(L) / S;
      const unsigned short b_in = (unsigned short)((h << 8) | 1), b_ex = (unsigned short)(h << 8);
      const float x_in = float(__builtin_bit_cast(half_s, b_in)), x_ex = float(__builtin_bit_cast(half_s, b_ex));
      lut[t * 2 + 0] = slope_closed(k, scale, x_in);
      lut[t * 2 + 1] = slope_closed(k, scale, x_ex);
    }
  }
  // The same table from a PER-LANE copy of the parameters (lane l < NC*L of every wave holds k[l], fetched by ONE
  // vector load issued before the image loads), indexed through wave shuffles: staging then has no dependent
  // global loads queued behind the first image chunk (vector loads return in order), which was most of its
  // latency.  All 64 lanes of every wave must be active; bit-identical to stage16 (same summation order).
  __device__ static float slope_closed_lanes(float klane, int c, float scale, float x) {
    const float u = x * float(L);
    const bool in = (u >= 0.0f && u <= float(L));
    const float jf = floorf(in ? u : 0.0f);
    const int j = int(jf);  // 0..L
    const float kj = __shfl(klane, c * L + (j < L ? j : L - 1));
    const float kp = __shfl(klane, c * L + (j >= 1 ? j - 1 : 0));
    float sl = (j < L) ? kj : 0.0f;
    if (jf == u && j >= 1) sl = kp + sl;
    return in ? scale * sl : 0.0f;
  }
  __device__ static void stage16_lanes(float klane, float* lut) {
    typedef _Float16 half_s;
#pragma unroll
    for (int c = 0; c < NC; ++c) {  // entries t = c * 256 + threadIdx.x (blockDim.x == 256)
      float S = 0.f;
#pragma unroll
      for (int i = 0; i < L; ++i) S += __shfl(klane, c * L + i);
      S += 1e-30f;
      const float scale = float(L) / S;
      const int h = threadIdx.x & 255, t = c * 256 + h;
      const unsigned short b_in = (unsigned short)((h << 8) | 1), b_ex = (unsigned short)(h << 8);
      const float x_in = float(__builtin_bit_cast(half_s, b_in)), x_ex = float(__builtin_bit_cast(half_s, b_ex));
      lut[t * 2 + 0] = slope_closed_lanes(klane, c, scale, x_in);
      lut[t * 2 + 1] = slope_closed_lanes(klane, c, scale, x_ex);
    }
  }
  __device__ static float slope16(const float* lut, int cc, unsigned short bits) {
    const float2_lut e = *reinterpret_cast<const float2_lut*>(lut + (cc * 256 + (bits >> 8)) * 2);
    return (bits & 0xFF) ? e.x : e.y;
  }
  template <bool F16X>
  __device__ static void stage_for(const float* __restrict__ p, float* lut) {
    if constexpr (F16X) stage16(p, lut); else stage(p, lut);
  }

  // Group backward (PPL pixels at once).  F16X: the inputs are exactly representable in fp16
  // (fp16 storage), so the eight accumulator updates Q_i += dy * min(x^, i/8) of TWO pixels run
  // as one v_pk_min_f16 + one v_dot2c_f32_f16 (fp16 x fp16 products are exact in fp32): 17 VALU
  // per element pair instead of 32.  Everything else (LUT, dx, B) stays fp32 per element.
  static constexpr bool kHasGroupBwd = true;
  // HAS_PEN: `pen` holds a per-element fp32 addend to the upstream gradient (the fused over-exposure penalty of
  // the dispatch backward, 2 max(y-1,0) dpen / (H W 3)).  dy itself stays an exact fp16 value, so the packed
  // accumulation keeps running on it; the addend is zero except on over-exposed pixels, where its share of the
  // eight Q_i is added in fp32 under a (rarely taken) per-lane branch.
  template <int PPL, bool F16X, bool HAS_PEN = false>
  __device__ static void bwd_group(const Prm& q, const float* lut, const float* x, float* d, float acc[NACC],
                                   const float* pen = nullptr) {
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    static_assert(PPL % 2 == 0, "pixels are processed in pairs");
    // Element pairs that share a curve: Tone (one curve for all channels) pairs the two halves of
    // each raw dword -- elements (2m, 2m+1) -- so the packed operands ARE the loaded registers;
    // Color pairs channel c of pixel k with channel c of pixel k+1 (a v_perm per operand).
    constexpr int NPAIR = PPL * 3 / 2;
#pragma unroll
    for (int m = 0; m < NPAIR; ++m) {
      const int iA = (NC == 1) ? 2 * m : 3 * (2 * (m / 3)) + (m % 3);
      const int iB = (NC == 1) ? 2 * m + 1 : iA + 3;
      const int cc = (NC == 1) ? 0 : (m % 3);
      float* a = acc + cc * L;
      const float xA = x[iA], xB = x[iB];
      float gA = d[iA], gB = d[iB];
      float pA = 0.f, pB = 0.f;
      if constexpr (HAS_PEN) {
        pA = pen[iA];
        pB = pen[iB];
        if constexpr (!F16X) {  // fp32 storage: one fp32 gradient, generic path below
          gA += pA;
          gB += pB;
        }
      }
      if constexpr (F16X) {
        const half2_t x2 = {_Float16(xA), _Float16(xB)};  // exact: values came from fp16 storage
        const half2_t g2 = {_Float16(gA), _Float16(gB)};
        // x^ = clamp(x, 0, 1) of both halves: ONE v_pk_max_f16 x, x clamp (it also canonicalises, so the seven
        // v_pk_min below need no quieting first); E_8 = min(x^, 1) = x^ needs no min at all
        half2_t xp;
        asm("v_pk_max_f16 %0, %1, 0 clamp" : "=v"(xp) : "v"(x2));  // max(-0, +0) = +0: the bit-pattern order below needs it
        // x^ and the thresholds are non-negative fp16 values, whose order is the order of their bit patterns:
        // the seven minima run as v_pk_min_u16 (an integer op is never preceded by a quieting v_pk_max)
        typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
        const ushort2_t xb = __builtin_bit_cast(ushort2_t, xp);
#pragma unroll
        for (int i = 1; i < L; ++i) {
          const unsigned short tb = __builtin_bit_cast(unsigned short, _Float16(float(i) / L));
          const ushort2_t mb = __builtin_elementwise_min(xb, ushort2_t{tb, tb});
          a[i - 1] = __builtin_amdgcn_fdot2(g2, __builtin_bit_cast(half2_t, mb), a[i - 1], false);
        }
        a[L - 1] = __builtin_amdgcn_fdot2(g2, xp, a[L - 1], false);
      } else {
        const float xcA = clamp01x(xA, 0.0f, 1.0f), xcB = clamp01x(xB, 0.0f, 1.0f);
#pragma unroll
        for (int i = 1; i < L; ++i) {
          a[i - 1] = fmaf(gA, fminf(xcA, float(i) / L), a[i - 1]);
          a[i - 1] = fmaf(gB, fminf(xcB, float(i) / L), a[i - 1]);
        }
        a[L - 1] = fmaf(gA, xcA, a[L - 1]);
        a[L - 1] = fmaf(gB, xcB, a[L - 1]);
      }
      if constexpr (F16X) {
        if constexpr (HAS_PEN) {
          if (pA != 0.f) {
            const float xc = clamp01x(xA, 0.0f, 1.0f);
#pragma unroll
            for (int i = 1; i <= L; ++i) a[i - 1] = fmaf(pA, fminf(xc, float(i) / L), a[i - 1]);
          }
          if (pB != 0.f) {
            const float xc = clamp01x(xB, 0.0f, 1.0f);
#pragma unroll
            for (int i = 1; i <= L; ++i) a[i - 1] = fmaf(pB, fminf(xc, float(i) / L), a[i - 1]);
          }
        }
        if constexpr (HAS_PEN) {  // (x + 0.0f is not a no-op for the compiler: -0 + 0 = +0)
          gA += pA;
          gB += pB;
        }
        d[iA] = gA * slope16(lut, cc, __builtin_bit_cast(unsigned short, _Float16(xA)));
        d[iB] = gB * slope16(lut, cc, __builtin_bit_cast(unsigned short, _Float16(xB)));
      } else {
        d[iA] = gA * lut_slope(lut, cc, xA, clamp01x(xA, 0.0f, 1.0f));
        d[iB] = gB * lut_slope(lut, cc, xB, clamp01x(xB, 0.0f, 1.0f));
      }
    }
  }
  // Fused over-exposure penalty as a FIX-UP after the plain group backward (everything is linear in the upstream
  // gradient): y through the per-wave segment table `tab` (kernel_common.h: curve_lut_build), addend
  // pe = max(y - 1, 0) * pen_scale; where it is non-zero (rare), dx += pe * slope and Q_i += pe * min(x^, i/L) in
  // fp32.  Keeping the addend out of the main pass keeps its registers out of the kernel's budget.
  template <int PPL, bool F16X>
  __device__ static void bwd_pen_fixup(const float* lut, const float2_lut* tab, const float* x, float* d,
                                       float acc[NACC], float pen_scale) {
#pragma unroll
    for (int e = 0; e < PPL * 3; ++e) {
      const int cc = (NC == 1) ? 0 : (e % 3);
      const float xc = clamp01x(x[e], 0.0f, 1.0f);
      const float2_lut seg = tab[cc * (L + 1) + int(xc * float(L))];
      const float pe = fmaxf(fmaf(xc, seg.x, seg.y) - 1.0f, 0.0f) * pen_scale;
      if (pe != 0.f) {
        float sl;
        if constexpr (F16X) sl = slope16(lut, cc, __builtin_bit_cast(unsigned short, _Float16(x[e])));
        else sl = lut_slope(lut, cc, x[e], xc);
        d[e] = fmaf(pe, sl, d[e]);
        float* a = acc + cc * L;
#pragma unroll
        for (int i = 1; i <= L; ++i) a[i - 1] = fmaf(pe, fminf(xc, float(i) / L), a[i - 1]);
      }
    }
  }
  // a[] per curve: Q_1..Q_L.  B = (L/S) sum_m (k_{m-1} - k_m) Q_m;  dk_i = (L/S)(Q_{i+1} - Q_i) - B/S
  __device__ static float finish_one(const float* __restrict__ p, const float* a, int j) {
    const int c = j / L, i = j % L;
    const float* k = p + c * L;
    const float* ac = a + c * L;
    float S = 0.f, B = 0.f;
    for (int m = 0; m < L; ++m) {
      S += k[m];
      B += (k[m] - ((m + 1 < L) ? k[m + 1] : 0.0f)) * ac[m];  // Q_{m+1} is stored at ac[m]
    }
    S += 1e-30f;
    const float scale = float(L) / S;
    const float qi = (i == 0) ? 0.0f : ac[i - 1];
    return scale * (ac[i] - qi) - scale * B / S;
  }
};
using ToneF = CurveF<1>;
using ColorF = CurveF<3>;

// ---------------------------------------------------------------------------------
// 5  ContrastFilter   filters.py:415-419
//    l = clip(lum(x),0,1); cl = -cos(pi l)/2 + 1/2 = sin^2(pi l / 2) (no cancellation);
//    ci = x/(l+1e-6)*cl; y = (1-p) x + p ci
// ---------------------------------------------------------------------------------
struct ContrastF {
  static constexpr int NP = 1, NACC = 1, kLutFloats = 0;
  static constexpr bool kHasGroupBwd = false;
  struct Prm { float p; };
  __device__ static Prm load(const float* __restrict__ p) { return {p[0]}; }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
    const float l = clamp01x(lum3(x), 0.0f, 1.0f);
    const float sh = sin_rev(l * 0.25f);
    const float ratio = sh * sh * fast_rcp(l + 1e-6f);
    const float f = (1.0f - q.p) + q.p * ratio;
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = x[c] * f;
  }
  __device__ static void bwd(const Prm& q, const float*, const float x[3], const float dy[3], float dx[3],
                             float acc[NACC], int) {
    const float lraw = lum3(x);
    const float l = clamp01x(lraw, 0.0f, 1.0f);
    const float sh = sin_rev(l * 0.25f), ch = cos_rev(l * 0.25f);
    const float cl = sh * sh;
    const float dcl = kPi * sh * ch;  // 0.5 pi sin(pi l)
    const float inv = fast_rcp(l + 1e-6f);
    const float ratio = cl * inv;
    const float G = (dcl - ratio) * inv;  // d/dl [cl/(l+eps)]
    const float dot = dy[0] * x[0] + dy[1] * x[1] + dy[2] * x[2];
    // maximum(.,0) / minimum(.,1) pass the gradient on 0 <= lum <= 1 (inclusive)
    const float common = (lraw >= 0.0f && lraw <= 1.0f) ? q.p * G * dot : 0.0f;
    const float f = (1.0f - q.p) + q.p * ratio;
    dx[0] = fmaf(dy[0], f, kLumR * common);
    dx[1] = fmaf(dy[1], f, kLumG * common);
    dx[2] = fmaf(dy[2], f, kLumB * common);
    acc[0] = fmaf(dot, ratio - 1.0f, acc[0]);  // sum_c dy_c (ci_c - x_c)
  }
  __device__ static float finish_one(const float* __restrict__, const float* a, int) { return a[0]; }
};

// ---------------------------------------------------------------------------------
// 6  WNBFilter   filters.py:438-440   y_c = (1-p) x_c + p lum(x)
// ---------------------------------------------------------------------------------
struct WnbF {
  static constexpr int NP = 1, NACC = 1, kLutFloats = 0;
  static constexpr bool kHasGroupBwd = false;
  struct Prm { float p; };
  __device__ static Prm load(const float* __restrict__ p) { return {p[0]}; }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
    const float pl = q.p * lum3(x);
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = fmaf(1.0f - q.p, x[c], pl);
  }
  __device__ static void bwd(const Prm& q, const float*, const float x[3], const float dy[3], float dx[3],
                             float acc[NACC], int) {
    const float l = lum3(x);
    const float sdy = dy[0] + dy[1] + dy[2];
    const float dot = dy[0] * x[0] + dy[1] * x[1] + dy[2] * x[2];
    const float ps = q.p * sdy;
    dx[0] = fmaf(1.0f - q.p, dy[0], kLumR * ps);
    dx[1] = fmaf(1.0f - q.p, dy[1], kLumG * ps);
    dx[2] = fmaf(1.0f - q.p, dy[2], kLumB * ps);
    acc[0] += l * sdy - dot;  // sum_c dy_c (lum - x_c)
  }
  __device__ static float finish_one(const float* __restrict__, const float* a, int) { return a[0]; }
};

// ---------------------------------------------------------------------------------
// 8  LevelFilter   filters.py:449-466 (defined by the reference but not in cfg.filters)
//    lower = p0, upper = p1 + 1;  y = clip((x - lower) / (upper - lower + 1e-6), 0, 1)
// ---------------------------------------------------------------------------------
struct LevelF {
  static constexpr int NP = 2, NACC = 2, kLutFloats = 0;
  static constexpr bool kHasGroupBwd = false;
  struct Prm { float lower, r; };
  __device__ static Prm load(const float* __restrict__ p) {
    return {p[0], 1.0f / ((p[1] + 1.0f) - p[0] + 1e-6f)};
  }
  __device__ static void fwd(const Prm& q, const float x[3], float y[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = clamp01x((x[c] - q.lower) * q.r, 0.0f, 1.0f);
  }
  __device__ static void bwd(const Prm& q, const float*, const float x[3], const float dy[3], float dx[3],
                             float acc[NACC], int) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = (x[c] - q.lower) * q.r;
      const float g = (t >= 0.0f && t <= 1.0f) ? dy[c] : 0.0f;  // clip_by_value: inclusive both sides
      dx[c] = g * q.r;
      acc[0] = fmaf(g, t - 1.0f, acc[0]);
      acc[1] = fmaf(g, t, acc[1]);
    }
  }
  // d/dlower = r sum g (t-1);  d/dp1 = d/dupper = -r sum g t
  __device__ static float finish_one(const float* __restrict__ p, const float* a, int j) {
    const float r = 1.0f / ((p[1] + 1.0f) - p[0] + 1e-6f);
    return j == 0 ? r * a[0] : -r * a[1];
  }
};

// ---------------------------------------------------------------------------------
// Spatial mask of Filter.apply with cfg.masking = True   filters.py:110-148, 86-88
//   mp = tanh_range(-5, 5)(mask_parameters)            (done by the caller, differentiably)
//   inp = gx mp0 + gy mp1 + mp2 (lum(x) - .5) + 2 mp3;  inp *= sharp mp4 / 5
//   mask = sigmoid(inp) (mp5/5 * .5 + .5)(1 - min_strength) + min_strength
//   out = (1 - mask) x + mask process(x)
// gx, gy: the constant grid (row + (se-H)/2)/se - .5, (col + (se-W)/2)/se - .5, se = min(H, W).
// Backward accumulators (6): d/d mp0..mp5.
// ---------------------------------------------------------------------------------
// Row and column of the pixels a lane visits inside one group, without a division (or two quarter-rate integer
// multiplies) per pixel: the first pixel's (row, column) comes from one multiply by 1/w with a +-1 correction, the
// others follow by constant index steps -- inside a 48-byte group the linear pixel index advances by one of two
// compile-time steps (pixel_io.h: pixel_step_a / pixel_step_b) whose row / column parts (step / w, step % w) are
// block-uniform.  Exact integer arithmetic: the same (row, column) as px / w, px % w.
struct PixelWalk {
  int w, qa, ra, qb, rb;
  float inv_w;
  bool small;
  __device__ static PixelWalk make(int h, int w, int step_a, int step_b) {
    PixelWalk p;
    p.w = w;
    p.inv_w = 1.0f / float(w);
    p.small = long(h) * long(w) < (1L << 22);  // float(px) is exact and one correction step suffices
    p.qa = step_a / w; p.ra = step_a % w;
    p.qb = step_b / w; p.rb = step_b % w;
    return p;
  }
  __device__ void start(int px, int& row, int& col) const {
    if (small) {
      row = int(float(px) * inv_w);
      const int r = px - row * w;
      row += (r >= w) ? 1 : 0;
      row -= (r < 0) ? 1 : 0;
    } else {
      row = px / w;
    }
    col = px - row * w;
  }
  // Keeps the walk in step with the pixel loop: the chain of updates is cheap, and left alone the scheduler runs it
  // ahead of the pixel arithmetic and holds all the (row, column) pairs of a group in registers (+10 ... +56 VGPRs,
  // an occupancy step for most of the mask kernels: gpurun r03p42).  `anchor`: a value the previous pixel produced.
  __device__ static void after(int& row, int& col, float anchor) { asm volatile("" : "+v"(row), "+v"(col) : "v"(anchor)); }
  __device__ void step(bool b, int& row, int& col) const {
    col += b ? rb : ra;
    row += b ? qb : qa;
    const bool wrap = col >= w;
    col -= wrap ? w : 0;
    row += wrap ? 1 : 0;
  }
};

struct MaskPrm {
  float a, b, c, d2, k, S, ms, k_over_mp4, dS;  // k = sharp mp4/5; S = strength factor; dS = dS/dmp5
  float inv_se, oi, oj;
  PixelWalk pw;
  __device__ static MaskPrm load(const float* __restrict__ mp, float sharp, float min_strength, int h, int w,
                                 int step_a = 1, int step_b = 1) {
    MaskPrm m;
    m.a = mp[0]; m.b = mp[1]; m.c = mp[2]; m.d2 = 2.0f * mp[3];
    m.k_over_mp4 = sharp / 5.0f;
    m.k = m.k_over_mp4 * mp[4];
    m.ms = min_strength;
    m.dS = (0.5f / 5.0f) * (1.0f - min_strength);
    m.S = (mp[5] / 5.0f * 0.5f + 0.5f) * (1.0f - min_strength);
    const int se = h < w ? h : w;
    m.inv_se = 1.0f / float(se);
    m.oi = float(se - h) * 0.5f;
    m.oj = float(se - w) * 0.5f;
    m.pw = PixelWalk::make(h, w, step_a, step_b);
    return m;
  }
  struct Eval { float m, sg, inp_raw, gx, gy, lumc; };
  __device__ Eval eval(int px, const float x[3]) const {
    int row, col;
    pw.start(px, row, col);
    return eval_rc(row, col, x);
  }
  __device__ Eval eval_rc(int row, int col, const float x[3]) const {
    Eval e;
    e.gx = (float(row) + oi) * inv_se - 0.5f;
    e.gy = (float(col) + oj) * inv_se - 0.5f;
    e.lumc = lum3(x) - 0.5f;
    e.inp_raw = e.gx * a + e.gy * b + c * e.lumc + d2;
    e.sg = fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * (e.inp_raw * k)));
    e.m = e.sg * S + ms;
    return e;
  }
};

// ---------------------------------------------------------------------------------
// VignetFilter   filters.py:341-396.  process() is img * 0 (the additive term is commented out, filters.py:351-352),
// so Filter.apply = lerp(img, 0, mask) = img (1 - mask) with the elliptical mask
//   mp = tanh_range(-5, 5)(mask_parameters)            (done by the caller, differentiably; 5 parameters)
//   u = (gx mp0)^2 + (gy mp1)^2 + mp2 - 5;  inp = u sharp mp3 / 5;  mask = sigmoid(inp) (mp4/5 * .5 + .5)
// on the same constant grid as MaskPrm; with cfg.masking off the reference forces mask = 1 (out = 0).
// The mask does not depend on the image.  Backward accumulators (5), with t = -sum_c x_c dy_c per pixel:
//   t s' gx^2, t s' gy^2, t s', t s' u, t s      (s = sigmoid(inp), s' = s (1 - s)); finish_vignet scales them.
// ---------------------------------------------------------------------------------
struct VignetPrm {
  float a, b, c5, k, S;  // c5 = mp2 - 5; k = sharp mp3 / 5; S = mp4/5 * .5 + .5
  float inv_se, oi, oj;
  PixelWalk pw;
  bool masking;
  __device__ static VignetPrm load(const float* __restrict__ mp, float sharp, int masking, int h, int w,
                                   int step_a = 1, int step_b = 1) {
    VignetPrm m;
    m.a = mp[0]; m.b = mp[1]; m.c5 = mp[2] - 5.0f;
    m.k = sharp * mp[3] / 5.0f;
    m.S = mp[4] / 5.0f * 0.5f + 0.5f;
    const int se = h < w ? h : w;
    m.inv_se = 1.0f / float(se);
    m.oi = float(se - h) * 0.5f;
    m.oj = float(se - w) * 0.5f;
    m.pw = PixelWalk::make(h, w, step_a, step_b);
    m.masking = masking != 0;
    return m;
  }
  struct Eval { float m, sg, u, gx2, gy2; };
  __device__ Eval eval(int px) const {
    int row, col;
    pw.start(px, row, col);
    return eval_rc(row, col);
  }
  __device__ Eval eval_rc(int row, int col) const {
    Eval e;
    const float gx = (float(row) + oi) * inv_se - 0.5f, gy = (float(col) + oj) * inv_se - 0.5f;
    e.gx2 = gx * gx;
    e.gy2 = gy * gy;
    e.u = (gx * a) * (gx * a) + (gy * b) * (gy * b) + c5;
    e.sg = fast_rcp(1.0f + fast_exp2(-1.4426950408889634f * (e.u * k)));
    e.m = masking ? e.sg * S : 1.0f;
    return e;
  }
  // d mask / d mp_j from the image totals of the five accumulators
  __device__ static float finish_one(const float* __restrict__ mp, float sharp, const float* a, int j) {
    const float k = sharp * mp[3] / 5.0f, S = mp[4] / 5.0f * 0.5f + 0.5f;
    switch (j) {
      case 0: return S * k * 2.0f * mp[0] * a[0];
      case 1: return S * k * 2.0f * mp[1] * a[1];
      case 2: return S * k * a[2];
      case 3: return S * (sharp / 5.0f) * a[3];
      default: return 0.1f * a[4];
    }
  }
};

}  // namespace expo
