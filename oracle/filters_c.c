/* filters_c.c -- plain-C restatement of the reference's eight per-pixel filters, forward AND backward.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/__init__.py): imported, linked or executed only by tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py -- never by the product.  PARITY UNPINNED by the
 * reference (it ships no tests and TensorFlow 1.x cannot run here); this file is the THIRD independent
 * restatement beside oracle/filters_np.py (NumPy, hand-derived backward) and oracle/filters_torch.py (op-by-op,
 * autograd backward); tests/test_oracle_c.py requires all three to agree on the golden vectors.
 *
 * It follows the reference's formulas literally (paths relative to /root/reference):
 *   Exposure   filters.py:181-182    y = x * exp(p * ln 2)
 *   Gamma      filters.py:205-206    y = pow(max(x, 0.001), g)
 *   WhiteBal.  filters.py:237-238    y_c = x_c * s_c
 *   Sat.Plus   filters.py:484-498    xc = min(x, 1); hsv; s' = s + (1-s)(.5-|.5-v|).8; full = rgb(h, s', v);
 *                                    y = xc (1-p) + full p   (TF RGBToHSV / HSVToRGB)
 *   Tone       filters.py:312-322    y = sum_i clip(x - i/L, 0, 1/L) k_i * L / (sum_i k_i + 1e-30), one curve
 *   Contrast   filters.py:415-419    l = min(max(lum, 0), 1); cl = -cos(pi l)/2 + 1/2; y = lerp(x, x/(l+1e-6) cl, p)
 *   WNB        filters.py:438-440    y = lerp(x, lum, p)
 *   Color      filters.py:264-273    Tone's formula with one curve per channel
 *   lum        util.py:271-274       .27 r + .67 g + .06 b
 * Gradient conventions (TF): maximum / minimum pass the gradient to x on equality; clip_by_value passes it on
 * lo <= x <= hi (both inclusive); abs'(0) = 0; RGBToHSV / HSVToRGB are not differentiable in TF 1.x
 * (no gradient through `full`; the build's hsv_grad_mode = 1 extension is not restated here).
 *
 * Packed parameters (N, P), the C-ABI layout of include/exposure_hip.h: P = 1,1,3,1,8,1,1,24 for ids 0..7 =
 * E,G,W,S+,T,Ct,BW,C; Color packs channel*8 + knot.
 *
 * Compiled twice by oracle/build_c.sh: REAL=double (the checker) and REAL=float with OpenMP (the CPU baseline:
 * the reference's own dtype, one fused pass per step and direction, all host cores).  Parameter gradients are
 * accumulated in double in both.  The checker build (-DORACLE_ABS_TERMS) also accumulates, per parameter, the sum
 * of the ABSOLUTE per-element terms |dy_e * dy_e/dp_k| (slots 24..47 of every accumulator): the scale A that
 * tests/_tol.py judges an fp32 accumulation against (tolerance 1e-4 |ref| + 2e-6 A).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define L 8
#define NACC 48 /* 24 sums + 24 sums of absolute terms */
#ifdef ORACLE_ABS_TERMS
#define ABS_TERM(dp, k, v) ((dp)[24 + (k)] += fabs((double)(v)))
#else
#define ABS_TERM(dp, k, v) ((void)0)
#endif
int oracle_c_has_abs_terms(void) {
#ifdef ORACLE_ABS_TERMS
  return 1;
#else
  return 0;
#endif
}
#define LUM_R ((real)0.27)
#define LUM_G ((real)0.67)
#define LUM_B ((real)0.06)
static const int kNumParams[8] = {1, 1, 3, 1, 8, 1, 1, 24};

int oracle_c_num_params(int fid) { return (fid >= 0 && fid < 8) ? kNumParams[fid] : -1; }
int oracle_c_real_bytes(void) { return (int)sizeof(real); }
/* OpenMP threads: sets when n > 0, returns the count in effect (results do not depend on it: fixed-order sums) */
int oracle_c_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* math in `real`: float functions in the float build, double in the double build (constant-folded) */
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_log(real x) { return sizeof(real) == 4 ? (real)logf((float)x) : (real)log((double)x); }
static inline real r_pow(real x, real y) { return sizeof(real) == 4 ? (real)powf((float)x, (float)y) : (real)pow((double)x, (double)y); }
static inline real r_cos(real x) { return sizeof(real) == 4 ? (real)cosf((float)x) : (real)cos((double)x); }
static inline real r_sin(real x) { return sizeof(real) == 4 ? (real)sinf((float)x) : (real)sin((double)x); }
static inline real r_abs(real x) { return x < 0 ? -x : x; }
#define R_PI ((real)3.14159265358979323846)
#define R_LN2 ((real)0.6931471805599453)
static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real clip(real x, real lo, real hi) { return rmin(rmax(x, lo), hi); }
static inline real lum(const real* x) { return (LUM_R * x[0] + LUM_G * x[1]) + LUM_B * x[2]; }

/* TF RGBToHSV */
static void rgb_to_hsv(const real* c, real* h, real* s, real* v) {
  const real r = c[0], g = c[1], b = c[2];
  const real mx = rmax(rmax(r, g), b), mn = rmin(rmin(r, g), b);
  const real rng = mx - mn;
  *v = mx;
  *s = mx > 0 ? rng / mx : 0;
  real hh = 0;
  if (rng > 0) {
    const real norm = (real)1 / ((real)6 * rng);
    if (r == mx) hh = norm * (g - b);
    else if (g == mx) hh = norm * (b - r) + (real)(2.0 / 6.0);
    else hh = norm * (r - g) + (real)(4.0 / 6.0);
    if (hh < 0) hh += 1;
  }
  *h = hh;
}
/* TF HSVToRGB */
static void hsv_to_rgb(real h, real s, real v, real* out) {
  const real dh = h * 6;
  const real dr = clip(r_abs(dh - 3) - 1, 0, 1);
  const real dg = clip(2 - r_abs(dh - 2), 0, 1);
  const real db = clip(2 - r_abs(dh - 4), 0, 1);
  const real oms = 1 - s;
  out[0] = (oms + s * dr) * v;
  out[1] = (oms + s * dg) * v;
  out[2] = (oms + s * db) * v;
}

/* ------------------------------------------------------------------ per-pixel forward */
static void satplus_full(const real* xc, real* full) {
  real h, s, v;
  rgb_to_hsv(xc, &h, &s, &v);
  const real sp = s + (1 - s) * ((real)0.5 - r_abs((real)0.5 - v)) * (real)0.8;
  hsv_to_rgb(h, sp, v, full);
}
static real curve_eval(const real* k, real x) {
  real S = 0, t = 0;
  for (int i = 0; i < L; ++i) {
    S += k[i];
    t += clip(x - (real)i / L, 0, (real)1 / L) * k[i];
  }
  return t * L / (S + (real)1e-30);
}
static void pixel_fwd(int fid, const real* p, const real* x, real* y) {
  switch (fid) {
    case 0: {
      const real s = r_exp(p[0] * R_LN2);
      for (int c = 0; c < 3; ++c) y[c] = x[c] * s;
    } break;
    case 1:
      for (int c = 0; c < 3; ++c) y[c] = r_pow(rmax(x[c], (real)0.001), p[0]);
      break;
    case 2:
      for (int c = 0; c < 3; ++c) y[c] = x[c] * p[c];
      break;
    case 3: {
      real xc[3], full[3];
      for (int c = 0; c < 3; ++c) xc[c] = rmin(x[c], 1);
      satplus_full(xc, full);
      for (int c = 0; c < 3; ++c) y[c] = xc[c] * (1 - p[0]) + full[c] * p[0];
    } break;
    case 4:
      for (int c = 0; c < 3; ++c) y[c] = curve_eval(p, x[c]);
      break;
    case 5: {
      const real l = rmin(rmax(lum(x), 0), 1);
      const real cl = -r_cos(R_PI * l) * (real)0.5 + (real)0.5;
      for (int c = 0; c < 3; ++c) {
        const real ci = x[c] / (l + (real)1e-6) * cl;
        y[c] = (1 - p[0]) * x[c] + p[0] * ci;
      }
    } break;
    case 6: {
      const real l = lum(x);
      for (int c = 0; c < 3; ++c) y[c] = (1 - p[0]) * x[c] + p[0] * l;
    } break;
    case 7:
      for (int c = 0; c < 3; ++c) y[c] = curve_eval(p + c * L, x[c]);
      break;
    default:
      y[0] = y[1] = y[2] = 0;
  }
}

/* ------------------------------------------------------------------ per-pixel backward
 * dx for the pixel; parameter gradients are ADDED to dp[0..P) (double). */
static void curve_bwd(const real* k, real x, real dy, real* dx, double* dp) {
  real S = 0, t = 0, slope = 0;
  real cl[L];
  for (int i = 0; i < L; ++i) {
    S += k[i];
    const real u = x - (real)i / L;
    cl[i] = clip(u, 0, (real)1 / L);
    t += cl[i] * k[i];
    if (u >= 0 && u <= (real)1 / L) slope += k[i]; /* clip_by_value: inclusive on both sides */
  }
  const real Se = S + (real)1e-30;
  *dx = dy * slope * L / Se;
  /* y = t L / Se:  dy/dk_i = L cl_i / Se - t L / Se^2 */
  for (int i = 0; i < L; ++i) {
    const real term = dy * (L * cl[i] / Se - t * L / (Se * Se));
    dp[i] += (double)term;
    ABS_TERM(dp, i, term);
  }
}
static void pixel_bwd(int fid, const real* p, const real* x, const real* dy, real* dx, double* dp) {
  switch (fid) {
    case 0: {
      const real s = r_exp(p[0] * R_LN2);
      for (int c = 0; c < 3; ++c) {
        dx[c] = dy[c] * s;
        dp[0] += (double)(R_LN2 * dy[c] * (x[c] * s));
        ABS_TERM(dp, 0, R_LN2 * dy[c] * (x[c] * s));
      }
    } break;
    case 1:
      for (int c = 0; c < 3; ++c) {
        const real xm = rmax(x[c], (real)0.001);
        const real y = r_pow(xm, p[0]);
        dx[c] = (x[c] >= (real)0.001) ? dy[c] * p[0] * y / xm : 0; /* tf.maximum passes on x >= 0.001 */
        dp[0] += (double)(dy[c] * y * r_log(xm));
        ABS_TERM(dp, 0, dy[c] * y * r_log(xm));
      }
      break;
    case 2:
      for (int c = 0; c < 3; ++c) {
        dx[c] = dy[c] * p[c];
        dp[c] += (double)dy[c] * (double)x[c];
        ABS_TERM(dp, c, (double)dy[c] * (double)x[c]);
      }
      break;
    case 3: {
      real xc[3], full[3], g[3];
      for (int c = 0; c < 3; ++c) xc[c] = rmin(x[c], 1);
      satplus_full(xc, full);
      for (int c = 0; c < 3; ++c) {
        dp[0] += (double)dy[c] * (double)(full[c] - xc[c]);
        ABS_TERM(dp, 0, (double)dy[c] * (double)(full[c] - xc[c]));
        g[c] = dy[c] * (1 - p[0]);
      }
      /* TF 1.x: no gradient through rgb_to_hsv / hsv_to_rgb (the build's optional analytic mode is an extension
         of the reference and is checked by the NumPy / torch oracles only) */
      for (int c = 0; c < 3; ++c) dx[c] = (x[c] <= 1) ? g[c] : 0; /* tf.minimum passes on x <= 1 */
    } break;
    case 4:
      for (int c = 0; c < 3; ++c) curve_bwd(p, x[c], dy[c], &dx[c], dp);
      break;
    case 5: {
      const real lraw = lum(x);
      const real l = rmin(rmax(lraw, 0), 1);
      const real cl = -r_cos(R_PI * l) * (real)0.5 + (real)0.5;
      const real dcl = (real)0.5 * R_PI * r_sin(R_PI * l);
      const real den = l + (real)1e-6;
      const real ratio = cl / den;
      real dot = 0;
      for (int c = 0; c < 3; ++c) dot += dy[c] * x[c];
      /* ci_c = x_c ratio(l);  y_c = (1-p) x_c + p ci_c */
      const real dratio = (dcl * den - cl) / (den * den);
      const int pass = (lraw >= 0 && lraw <= 1); /* maximum(.,0) then minimum(.,1): inclusive */
      const real common = pass ? p[0] * dratio * dot : 0;
      const real f = (1 - p[0]) + p[0] * ratio;
      dx[0] = dy[0] * f + LUM_R * common;
      dx[1] = dy[1] * f + LUM_G * common;
      dx[2] = dy[2] * f + LUM_B * common;
      dp[0] += (double)(dot * (ratio - 1));
      for (int c = 0; c < 3; ++c) ABS_TERM(dp, 0, dy[c] * x[c] * (ratio - 1)); /* per output ELEMENT: dy_c (ci_c - x_c) */
    } break;
    case 6: {
      const real l = lum(x);
      real sdy = 0, dot = 0;
      for (int c = 0; c < 3; ++c) {
        sdy += dy[c];
        dot += dy[c] * x[c];
      }
      dx[0] = (1 - p[0]) * dy[0] + p[0] * sdy * LUM_R;
      dx[1] = (1 - p[0]) * dy[1] + p[0] * sdy * LUM_G;
      dx[2] = (1 - p[0]) * dy[2] + p[0] * sdy * LUM_B;
      dp[0] += (double)(l * sdy - dot);
      for (int c = 0; c < 3; ++c) ABS_TERM(dp, 0, dy[c] * (l - x[c]));
    } break;
    case 7:
      for (int c = 0; c < 3; ++c) curve_bwd(p + c * L, x[c], dy[c], &dx[c], dp + c * L);
      break;
    default:
      dx[0] = dx[1] = dx[2] = 0;
  }
}

/* ------------------------------------------------------------------ batched entry points (ctypes)
 * x, y, dy, dx: (n, hw, 3) contiguous; p: (n, P); dp: (n, P), overwritten. */
void oracle_c_process(int fid, const real* x, const real* p, real* y, long n, long hw) {
  const int P = oracle_c_num_params(fid);
#pragma omp parallel for collapse(2) schedule(static)
  for (long i = 0; i < n; ++i)
    for (long j = 0; j < hw; ++j) pixel_fwd(fid, p + i * P, x + (i * hw + j) * 3, y + (i * hw + j) * 3);
}

#define ROW_BLOCK 4096
/* adp (n, P), may be NULL: the sums of absolute terms (checker build only; zeros otherwise) */
void oracle_c_backward_abs(int fid, const real* x, const real* p, const real* dy, real* dx, real* dp, double* adp,
                           long n, long hw) {
  const int P = oracle_c_num_params(fid);
  const long nb = (hw + ROW_BLOCK - 1) / ROW_BLOCK;
  double* part = (double*)calloc((size_t)(n * nb) * NACC, sizeof(double));
#pragma omp parallel for collapse(2) schedule(static)
  for (long i = 0; i < n; ++i)
    for (long b = 0; b < nb; ++b) {
      double acc[NACC];
      memset(acc, 0, sizeof(acc));
      const long j1 = (b + 1) * ROW_BLOCK < hw ? (b + 1) * ROW_BLOCK : hw;
      for (long j = b * ROW_BLOCK; j < j1; ++j) {
        const long o = (i * hw + j) * 3;
        pixel_bwd(fid, p + i * P, x + o, dy + o, dx + o, acc);
      }
      memcpy(part + (i * nb + b) * NACC, acc, sizeof(acc));
    }
  for (long i = 0; i < n; ++i) /* fixed order: reproducible whatever the thread count */
    for (int k = 0; k < P; ++k) {
      double s = 0, a = 0;
      for (long b = 0; b < nb; ++b) {
        s += part[(i * nb + b) * NACC + k];
        a += part[(i * nb + b) * NACC + 24 + k];
      }
      dp[i * P + k] = (real)s;
      if (adp) adp[i * P + k] = a;
    }
  free(part);
}
void oracle_c_backward(int fid, const real* x, const real* p, const real* dy, real* dx, real* dp, long n, long hw) {
  oracle_c_backward_abs(fid, x, p, dy, dx, dp, NULL, n, hw);
}

/* The benchmark chain of BASELINE.json: `steps` filters applied in sequence (acts[s+1] = f_s(acts[s])), then the
 * backward from grads[steps] down to grads[0] with every step's parameter gradients -- one fused pass per step
 * and direction, like the HIP chain.  acts: steps+1 buffers, grads: 2 ping-pong buffers + the upstream one. */
void oracle_c_chain(const int* fids, int steps, real** acts, const real* const* params, real* dy_top, real* g0,
                    real* g1, real* const* dparams, long n, long hw) {
  for (int s = 0; s < steps; ++s) oracle_c_process(fids[s], acts[s], params[s], acts[s + 1], n, hw);
  const real* up = dy_top;
  for (int s = steps - 1; s >= 0; --s) {
    real* down = (s % 2 == 0) ? g0 : g1;
    oracle_c_backward(fids[s], acts[s], params[s], up, down, dparams[s], n, hw);
    up = down;
  }
}
