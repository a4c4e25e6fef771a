// r03p48: where does buffer_load_dwordx3 ... lds put its data?  (gfx950; LDS address = M0 base + instruction offset + lane * 12 ?)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
template <int MODE>
__global__ void k(const uint32_t* __restrict__ x, uint32_t* __restrict__ y, int n) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[2048];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = 0xdead0000u + i;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(x), 0, n * 4, 0x00020000);
  const int vo = lane * 12;
  if (MODE == 0) {  // same LDS pointer, instruction offsets 0 / 768
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(buf), 12, vo, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(buf), 12, vo, 0, 768, 0);
  } else {  // LDS pointer advanced by 768 B for the second row, instruction offset 768 too
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(buf), 12, vo, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(buf + 192), 12, vo, 0, 768, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) y[i] = buf[i];
}
int main() {
  const int n = 4096;
  std::vector<uint32_t> h(n), out(2048);
  for (int i = 0; i < n; ++i) h[i] = i;
  uint32_t *x, *y;
  hipMalloc(&x, n * 4); hipMalloc(&y, 2048 * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, x, y, n);
    else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, x, y, n);
    hipMemcpy(out.data(), y, 2048 * 4, hipMemcpyDeviceToHost);
    printf("mode %d:", mode);
    int run_start = -1;
    for (int i = 0; i <= 2048; ++i) {
      bool data = i < 2048 && (out[i] & 0xffff0000u) != 0xdead0000u;
      if (data && run_start < 0) run_start = i;
      if (!data && run_start >= 0) { printf(" lds[%d..%d) = x[%u..%u]", run_start, i, out[run_start], out[i - 1]); run_start = -1; }
    }
    printf("\n");
  }
  return 0;
}
