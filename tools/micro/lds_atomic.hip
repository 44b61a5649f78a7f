// Microbenchmark: LDS atomics on 8160 tile counters held in LDS (returning u32 add, non-returning u32 add, f32 add),
// random counters.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>  // 0 returning u32, 1 non-returning u32, 2 non-returning f32
__global__ void __launch_bounds__(1024) k(uint32_t per, uint32_t* sink) {
  __shared__ uint32_t ctr[8192];
  for (int i = threadIdx.x; i < 8192; i += 1024) ctr[i] = 0;
  __syncthreads();
  uint32_t h = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u, acc = 0;
  for (uint32_t j = 0; j < per; j++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t t = (h >> 8) % 8160u;
    if (MODE == 0) acc += atomicAdd(&ctr[t], 1u);
    else if (MODE == 1) atomicAdd(&ctr[t], 1u);
    else atomicAdd(reinterpret_cast<float*>(&ctr[t]), 1.0f);
  }
  __syncthreads();
  if (acc == 0xffffffffu || ctr[threadIdx.x] == 0xfffffff0u) sink[0] = acc;
}

int main() {
  uint32_t* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t per = 256; const int blocks = 1024;
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipDeviceSynchronize(); hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, per, sink);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, per, sink);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, per, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double n = (double)blocks * 1024 * per;
    printf("mode %d: %.1f us, %.1f G LDS atomics/s chip-wide (%.2f per CU per ns)\n", mode, best * 1e3, n / (best * 1e-3) / 1e9,
           n / (best * 1e-3) / 1e9 / 256);
  }
  return 0;
}
