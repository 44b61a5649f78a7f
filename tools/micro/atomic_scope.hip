// Microbenchmark: returning 32-bit atomic adds on tile counters (one counter per 64 B), device (agent) scope vs
// workgroup scope on per-XCD private copies.  hipcc --offload-arch=gfx950 -O3 atomic_scope.hip -o atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}

template <int MODE>  // 0 agent, 1 workgroup scope + per-XCD copy, 2 workgroup scope shared copy (expected wrong)
__global__ void __launch_bounds__(256) k(uint32_t* ctr, uint32_t tiles, uint32_t n, uint32_t per, uint32_t* sink) {
  const uint32_t xcd = xcc_id();
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    uint32_t h = i * 2654435761u;
#pragma unroll 4
    for (uint32_t j = 0; j < per; j++) {
      h = h * 1664525u + 1013904223u;
      const uint32_t t = (h >> 8) % tiles;
      if (MODE == 0)
        acc += __hip_atomic_fetch_add(ctr + (size_t)t * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 1)
        acc += __hip_atomic_fetch_add(ctr + ((size_t)xcd * tiles + t) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        acc += __hip_atomic_fetch_add(ctr + (size_t)t * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}

int main() {
  const uint32_t tiles = 8160, n = 1000000, per = 4;
  uint32_t *ctr, *sink;
  const size_t bytes = (size_t)8 * tiles * 16 * 4;
  hipMalloc(&ctr, bytes); hipMalloc(&sink, 4);
  std::vector<uint32_t> h(bytes / 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; mode++) {
    for (int blocks : {512, 2048}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        hipMemset(ctr, 0, bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, ctr, tiles, n, per, sink);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, ctr, tiles, n, per, sink);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, ctr, tiles, n, per, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      hipMemcpy(h.data(), ctr, bytes, hipMemcpyDeviceToHost);
      unsigned long long sum = 0;
      for (size_t i = 0; i < h.size(); i += 16) sum += h[i];
      printf("mode %d blocks %4d: %.1f us  %.2f G atomics/s  sum %llu (expected %llu) %s\n", mode, blocks, best * 1e3,
             (double)n * per / (best * 1e-3) / 1e9, sum, (unsigned long long)n * per,
             sum == (unsigned long long)n * per ? "OK" : "MISMATCH");
    }
  }
  return 0;
}
