// Micro-benchmark: bytes per second entering the CUs from L2, by path.
//   mode 0: global_load_dwordx4 -> VGPR (accumulated so the loads stay)      mode 1: global_load_lds_dwordx4 -> LDS
// Every workgroup re-reads its own `span` bytes (L2-resident for small spans) `iters` times; `depth` loads in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void reader(const f32x4* base, size_t span_vec, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const f32x4* p = base + (size_t)blockIdx.x * span_vec;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  f32x4 acc = {0, 0, 0, 0};
  const int per_iter = (int)(span_vec / THREADS) / 8 * 8;  // whole groups of 8 loads only: never past the span
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < per_iter; j += 8) {
      if constexpr (MODE == 0) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)(j + u) * THREADS + tid);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          __builtin_amdgcn_global_load_lds((const void*)(p + (size_t)(j + u) * THREADS + w * 64 + lane),
                                           LDS_PTR(smem + ((u * (THREADS / 64) + w) * 1024)), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  if (MODE == 0 && acc[0] == 123.456f) sink[0] = acc[1];
}

template <int MODE, int THREADS>
double run(const f32x4* buf, size_t span_bytes, int blocks, int iters, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = MODE ? 8 * (THREADS / 64) * 1024 : 0;
  hipFuncSetAttribute((const void*)reader<MODE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((reader<MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, buf, span_bytes / 16, 2, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((reader<MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, buf, span_bytes / 16, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)span_bytes * blocks * iters / (ms * 1e-3) / 1e12;
}

int main() {
  const int blocks = 256;
  float* sink; hipMalloc(&sink, 64);
  for (size_t span : {size_t(256) << 10, size_t(1) << 20, size_t(4) << 20}) {  // multiples of 1024 threads x 8 loads x 16 B
    f32x4* buf; hipMalloc(&buf, span * blocks); hipMemset(buf, 0, span * blocks);
    const int iters = (int)((size_t(256) << 20) / span);
    printf("span %6zu KiB/WG x 256 WGs (total %5zu MiB): VGPR 256thr %.2f TB/s  512thr %.2f  1024thr %.2f | LDS-DMA 256thr %.2f  512thr %.2f  1024thr %.2f\n",
           span >> 10, span * blocks >> 20, run<0, 256>(buf, span, blocks, iters, sink), run<0, 512>(buf, span, blocks, iters, sink),
           run<0, 1024>(buf, span, blocks, iters, sink), run<1, 256>(buf, span, blocks, iters, sink),
           run<1, 512>(buf, span, blocks, iters, sink), run<1, 1024>(buf, span, blocks, iters, sink));
    fflush(stdout);
    hipFree(buf);
  }
  return 0;
}
