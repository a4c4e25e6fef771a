// valubench.hip -- issue rate of the VALU instructions the filter kernels are built from, on gfx950.
// Each kernel runs ITER x 16 independent instructions per wave (inline asm, register operands only);
// the grid fills every SIMD with WAVES waves.  Prints cycles per wave-instruction per SIMD at the
// measured shader clock (time / (ITER*16*waves_per_simd)), i.e. 4.0 = one wave64 instruction per 4 clocks.
// build: hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

// 16 independent scalar-fp32 accumulators a[i], 16 float2 accumulators p[i], 16 dword h[i] (packed fp16)
#define KERNEL(NAME, BODY)                                                                       \
  __global__ __launch_bounds__(256) void NAME(float* out, float s, float t) {                    \
    float a[16];                                                                                 \
    typedef float f2 __attribute__((ext_vector_type(2)));                                        \
    f2 p[16];                                                                                    \
    unsigned h[16];                                                                              \
    f2 s2 = {s, t}; float tv = t + threadIdx.x * 0.f; unsigned long long mask = __ballot(threadIdx.x & 1); unsigned three = 3 + (threadIdx.x >> 10), zero = threadIdx.x >> 10; __shared__ float lds[4096]; lds[threadIdx.x] = s; __syncthreads(); unsigned ldsaddr = (threadIdx.x & 255) * 8; unsigned ldsaddr4 = (threadIdx.x & 255) * 4;                                                                           \
    for (int i = 0; i < 16; ++i) {                                                               \
      a[i] = threadIdx.x * 1e-3f + i;                                                            \
      p[i] = f2{a[i], a[i] + 0.5f};                                                              \
      h[i] = 0x3c003800u + i;                                                                    \
    }                                                                                            \
    for (int it = 0; it < ITER; ++it) {                                                          \
      BODY                                                                                       \
    }                                                                                            \
    float r = 0.f;                                                                               \
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y + __uint_as_float(h[i]);           \
    if (r == 12345.678f) out[0] = r;                                                             \
  }

#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(tv));
#define OP_MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
#define OP_MAX(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
#define OP_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(tv));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
#define OP_PKMULS(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "s"(s2));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
#define OP_CVT(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(h[i]));
#define OP_CVTS(i) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(h[i]));
#define OP_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(a[(i + 1) & 15]));
#define OP_MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(h[i]) : "v"(h[(i + 1) & 15]), "s"(s));
#define OP_MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h[i]) : "v"(h[(i + 1) & 15]), "s"(s));
#define OP_MIX32(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(h[i]), "s"(s));
#define OP_DOT2(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(h[i]), "v"(h[(i + 1) & 15]));
#define OP_PKMINH(i) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 15]));
#define OP_PKMULH(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 15]));
#define OP_PKFMAH(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 15]));
#define OP_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define OP_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define OP_SIN(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]) : "vcc");
#define OP_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define OP_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(h[i]) : "v"(h[(i + 1) & 15]));
#define OP_CNDMASK64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "s"(mask));
#define OP_CNDMASKC(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(tv) : "vcc");
#define OP_CNDMASKI(i) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]) : : "vcc");
#define OP_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 15]) : "vcc");
#define OP_CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 15]) : "vcc");
#define OP_CMP64CND(i) asm volatile("v_cmp_lt_f32_e64 %2, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "s"(mask));
#define OP_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(h[i]) : "s"(s), "v"(h[(i + 1) & 15]));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %2, %1" : "+v"(h[i]) : "s"(s), "v"(h[(i + 1) & 15]));
#define OP_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(h[i]) : "s"(s));
#define OP_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(tv));
#define OP_ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
#define OP_LSHLSDWA(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(h[i]) : "v"(three), "v"(h[(i + 1) & 15]));
#define OP_CMPSDWA(i) asm volatile("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:BYTE_0 src1_sel:DWORD" : : "v"(h[i]), "v"(zero) : "vcc");
#define OP_DSREAD(i) asm volatile("ds_read_b64 %0, %1" : "=v"(p[i]) : "v"(ldsaddr));
#define OP_DSREADW(i) asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(p[i]) : "v"(ldsaddr));
#define OP_DSADD(i) asm volatile("ds_add_f32 %0, %1 offset:" #i "*1024" : : "v"(ldsaddr4), "v"(a[i]) : "memory");
#define OP_DSWRITE(i) asm volatile("ds_write_b32 %0, %1 offset:" #i "*1024" : : "v"(ldsaddr4), "v"(a[i]) : "memory");
#define OP_PKMOV(i) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(p[i]) : "v"(p[(i + 1) & 15]));
// a mixed pair: one scalar fma + one transcendental (do they overlap?)
#define OP_FMAEXP(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_exp_f32 %1, %1" : "+v"(a[i]), "+v"(p[i].x) : "s"(s), "v"(tv));

KERNEL(k_fma, REP16(OP_FMA))
KERNEL(k_mul, REP16(OP_MUL))
KERNEL(k_max, REP16(OP_MAX))
KERNEL(k_med3, REP16(OP_MED3))
KERNEL(k_pk_fma, REP16(OP_PKFMA))
KERNEL(k_pk_mul, REP16(OP_PKMUL))
KERNEL(k_pk_add, REP16(OP_PKADD))
KERNEL(k_pk_mul_sgpr, REP16(OP_PKMULS))
KERNEL(k_mov, REP16(OP_MOV))
KERNEL(k_cvt_f32_f16, REP16(OP_CVT))
KERNEL(k_cvt_f32_f16_sdwa, REP16(OP_CVTS))
KERNEL(k_cvt_pk_f16_f32, REP16(OP_CVTPK))
KERNEL(k_fma_mixlo_f16, REP16(OP_MIXLO))
KERNEL(k_fma_mixhi_f16, REP16(OP_MIXHI))
KERNEL(k_fma_mix_f32, REP16(OP_MIX32))
KERNEL(k_dot2c_f32_f16, REP16(OP_DOT2))
KERNEL(k_pk_min_f16, REP16(OP_PKMINH))
KERNEL(k_pk_mul_f16, REP16(OP_PKMULH))
KERNEL(k_pk_fma_f16, REP16(OP_PKFMAH))
KERNEL(k_exp, REP16(OP_EXP))
KERNEL(k_log, REP16(OP_LOG))
KERNEL(k_rcp, REP16(OP_RCP))
KERNEL(k_sin, REP16(OP_SIN))
KERNEL(k_cndmask, REP16(OP_CNDMASK))
KERNEL(k_cvt_i32_f32, REP16(OP_CVTI))
KERNEL(k_lshl_add, REP16(OP_LSHLADD))
KERNEL(k_fma_plus_exp, REP16(OP_FMAEXP))
KERNEL(k_cndmask64, REP16(OP_CNDMASK64))
KERNEL(k_cndmaskc, REP16(OP_CNDMASKC))
KERNEL(k_cndmaski, REP16(OP_CNDMASKI))
KERNEL(k_cmp, REP16(OP_CMP))
KERNEL(k_cmpcnd, REP16(OP_CMPCND))
KERNEL(k_cmp64cnd, REP16(OP_CMP64CND))
KERNEL(k_bfi, REP16(OP_BFI))
KERNEL(k_perm, REP16(OP_PERM))
KERNEL(k_and, REP16(OP_AND))
KERNEL(k_max3, REP16(OP_MAX3))
KERNEL(k_add, REP16(OP_ADD))
KERNEL(k_lshlsdwa, REP16(OP_LSHLSDWA))
KERNEL(k_cmpsdwa, REP16(OP_CMPSDWA))
KERNEL(k_dsread, REP16(OP_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");)
KERNEL(k_dsreadw, REP16(OP_DSREADW))
KERNEL(k_pkmov, REP16(OP_PKMOV))
KERNEL(k_dsadd, REP16(OP_DSADD) asm volatile("s_waitcnt lgkmcnt(0)");)
KERNEL(k_dswrite, REP16(OP_DSWRITE) asm volatile("s_waitcnt lgkmcnt(0)");)

typedef void (*kern_t)(float*, float, float);
struct Entry { const char* name; kern_t k; int per_rep; };

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;  // kHz -> GHz
  float* out;
  CK(hipMalloc(&out, 4));
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  Entry es[] = {
      {"v_fma_f32", k_fma, 16}, {"v_mul_f32", k_mul, 16}, {"v_max_f32", k_max, 16}, {"v_med3_f32", k_med3, 16},
      {"v_pk_fma_f32", k_pk_fma, 16}, {"v_pk_mul_f32", k_pk_mul, 16}, {"v_pk_add_f32", k_pk_add, 16},
      {"v_pk_mul_f32 (sgpr pair)", k_pk_mul_sgpr, 16}, {"v_mov_b32", k_mov, 16},
      {"v_cvt_f32_f16", k_cvt_f32_f16, 16}, {"v_cvt_f32_f16_sdwa", k_cvt_f32_f16_sdwa, 16},
      {"v_cvt_pk_f16_f32", k_cvt_pk_f16_f32, 16}, {"v_fma_mixlo_f16", k_fma_mixlo_f16, 16},
      {"v_fma_mixhi_f16", k_fma_mixhi_f16, 16}, {"v_fma_mix_f32", k_fma_mix_f32, 16},
      {"v_dot2c_f32_f16", k_dot2c_f32_f16, 16}, {"v_pk_min_f16", k_pk_min_f16, 16},
      {"v_pk_mul_f16", k_pk_mul_f16, 16}, {"v_pk_fma_f16", k_pk_fma_f16, 16},
      {"v_exp_f32", k_exp, 16}, {"v_log_f32", k_log, 16}, {"v_rcp_f32", k_rcp, 16}, {"v_sin_f32", k_sin, 16},
      {"v_cndmask_b32", k_cndmask, 16}, {"v_cvt_i32_f32", k_cvt_i32_f32, 16}, {"v_lshl_add_u32", k_lshl_add, 16},
      {"v_fma_f32 + v_exp_f32 (pair)", k_fma_plus_exp, 16},
      {"v_cndmask_b32_e64 (sgpr mask)", k_cndmask64, 16}, {"v_cndmask_b32 v, s, vcc", k_cndmaskc, 16},
      {"v_cndmask_b32 0, v, vcc", k_cndmaski, 16}, {"v_cmp_lt_f32 vcc", k_cmp, 16},
      {"v_cmp vcc + v_cndmask vcc (pair)", k_cmpcnd, 16}, {"v_cmp_e64 + v_cndmask_e64 (pair)", k_cmp64cnd, 16},
      {"v_bfi_b32", k_bfi, 16}, {"v_perm_b32", k_perm, 16}, {"v_and_b32", k_and, 16}, {"v_max3_f32", k_max3, 16},
      {"v_add_f32", k_add, 16}, {"v_lshlrev_b32_sdwa", k_lshlsdwa, 16}, {"v_cmp_eq_u32_sdwa", k_cmpsdwa, 16},
      {"ds_read_b64 (16 in flight)", k_dsread, 16}, {"ds_read_b64 + wait", k_dsreadw, 16},
      {"v_pk_mov_b32", k_pkmov, 16},
      {"ds_add_f32 (no return, lane-private)", k_dsadd, 16}, {"ds_write_b32", k_dswrite, 16},
  };
  printf("device %s, %d CUs, %.3f GHz (hipDeviceProp clockRate), %d wave(s) per SIMD, %d x 16 instructions per wave\n",
         prop.name, cus, ghz, waves_per_simd, ITER);
  printf("%-32s %10s %22s\n", "instruction", "us", "clk / wave-instr / SIMD");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (auto& e : es) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double instr = double(ITER) * e.per_rep * waves_per_simd;
    printf("%-32s %10.1f %22.2f\n", e.name, best * 1e3, best * 1e-3 * ghz * 1e9 / instr);
  }
  return 0;
}
