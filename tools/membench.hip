// membench.hip -- access-pattern ceilings on MI355X for the filter kernels' I/O skeleton.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
// Variants (all stream 16-byte accesses, 256-thread blocks, grid-stride):
//   copy16      lane i -> byte 16 i of each 4 KiB block-row (classic fully coalesced copy)
//   copy48      lane i -> bytes [48 i, 48 i + 48) as three dwordx4 (the pixel-group pattern)
//   copy48nt    same with nontemporal loads + stores
//   copy48u2    same, two groups in flight per thread
//   rrw48       read x, read dy, write dx with the pixel-group pattern (backward skeleton)
//   rrw48nt     same, nontemporal
// Buffers rotate over NBUF distinct allocations so the 256 MiB Infinity Cache cannot serve re-reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

__global__ __launch_bounds__(256) void copy16(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n16) {
  const size_t stride = size_t(gridDim.x) * 256;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) {
    u32x4 v = a[i];
    v.x ^= 1u;
    b[i] = v;
  }
}

template <bool NT, int U>
__global__ __launch_bounds__(256) void copy48(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t ngroups) {
  const size_t stride = size_t(gridDim.x) * 256;
  for (size_t g = size_t(blockIdx.x) * 256 + threadIdx.x; g < ngroups; g += stride * U) {
    u32x4 v[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t gg = g + u * stride;
      if (gg < ngroups) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[u][j] = ld<NT>(a + gg * 3 + j);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t gg = g + u * stride;
      if (gg < ngroups) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { v[u][j].x ^= 1u; st<NT>(b + gg * 3 + j, v[u][j]); }
      }
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void rrw48(const u32x4* __restrict__ a, const u32x4* __restrict__ c,
                                             u32x4* __restrict__ b, size_t ngroups) {
  const size_t stride = size_t(gridDim.x) * 256;
  for (size_t g = size_t(blockIdx.x) * 256 + threadIdx.x; g < ngroups; g += stride) {
    u32x4 v[3], w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = ld<NT>(a + g * 3 + j);
#pragma unroll
    for (int j = 0; j < 3; ++j) w[j] = ld<NT>(c + g * 3 + j);
#pragma unroll
    for (int j = 0; j < 3; ++j) { v[j] ^= w[j]; st<NT>(b + g * 3 + j, v[j]); }
  }
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 96) * (1ul << 20);  // per buffer
  const int nbuf = argc > 2 ? atoi(argv[2]) : 9;
  const int reps = 40;
  std::vector<u32x4*> buf(nbuf);
  for (auto& p : buf) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
  const size_t n16 = bytes / 16, ngroups = bytes / 48;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {1024, 2048, 4096, 8192, 16384};
  printf("buffer %zu MiB x %d, %d reps\n", bytes >> 20, nbuf, reps);
  for (int grid : grids) {
    auto run = [&](const char* name, int streams, auto launch) {
      for (int i = 0; i < 3; ++i) launch(i);
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch(i);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("grid %5d  %-9s %8.1f GB/s  (%.1f us/launch)\n", grid, name, double(bytes) * streams * reps / (ms * 1e-3) / 1e9,
             ms / reps * 1e3);
    };
    run("copy16", 2, [&](int i) { copy16<<<grid, 256>>>(buf[i % nbuf], buf[(i + 1) % nbuf], n16); });
    run("copy48", 2, [&](int i) { copy48<false, 1><<<grid, 256>>>(buf[i % nbuf], buf[(i + 1) % nbuf], ngroups); });
    run("copy48nt", 2, [&](int i) { copy48<true, 1><<<grid, 256>>>(buf[i % nbuf], buf[(i + 1) % nbuf], ngroups); });
    run("copy48u2", 2, [&](int i) { copy48<false, 2><<<grid, 256>>>(buf[i % nbuf], buf[(i + 1) % nbuf], ngroups); });
    run("rrw48", 3, [&](int i) { rrw48<false><<<grid, 256>>>(buf[i % nbuf], buf[(i + 4) % nbuf], buf[(i + 1) % nbuf], ngroups); });
    run("rrw48nt", 3, [&](int i) { rrw48<true><<<grid, 256>>>(buf[i % nbuf], buf[(i + 4) % nbuf], buf[(i + 1) % nbuf], ngroups); });
  }
  return 0;
}
