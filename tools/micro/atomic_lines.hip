// Microbenchmark: what does a non-returning fp32 atomic add cost at the memory side -- per LANE or per 64-byte LINE?
// Every wave issues `iters` global_atomic_add_f32 instructions with `lanes` active lanes, spread over `lines` distinct
// random 64-byte lines of a table (lanes / lines consecutive dwords in each line).  The reverse walk commits nine sums per
// (wave, Gaussian): SoA accumulators put them in 4 lines (colour, opacity, mean, conic); an AoS row would put them in 1.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_lines.hip -o atomic_lines && ./atomic_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) k(float* tab, uint32_t nlines, int iters, int lanes, int lines) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int per = lanes / lines;                 // consecutive dwords per line
  const bool on = lane < lanes;
  const int myline = on ? lane / per : 0, mydw = on ? lane % per : 0;
  uint32_t h = wave * 2654435761u + 12345u;
  for (int i = 0; i < iters; i++) {
    h = h * 1664525u + 1013904223u;
    // `lines` distinct pseudo-random lines per instruction (wave-uniform base, per-lane line offset hashed)
    uint32_t hl = (h ^ (uint32_t)myline * 0x9e3779b9u) * 2246822519u;
    const uint32_t line = (hl >> 7) % nlines;
    if (on) atomicAdd(tab + (size_t)line * 16 + mydw, 1.0f);
  }
}

int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cfgs[][2] = {{1, 1}, {4, 4}, {4, 1}, {9, 9}, {9, 4}, {9, 2}, {9, 1}, {16, 16}, {16, 4}, {16, 1}, {64, 64}, {64, 16}, {64, 4}};
  for (uint32_t mb : {64u, 2048u}) {
    const uint32_t nlines = mb * 1024u * 1024u / 64u;
    float* tab; hipMalloc(&tab, (size_t)nlines * 64);
    hipMemset(tab, 0, (size_t)nlines * 64);
    for (auto& c : cfgs) {
      const int lanes = c[0], lines = c[1], iters = 256, blocks = 4096;
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, nlines, iters, lanes, lines);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double instr = (double)blocks * 4 * iters;
      printf("table %4u MB  lanes %2d in %2d lines: %8.1f us  %6.2f G instr/s  %7.2f G lanes/s  %7.2f G lines/s\n", mb, lanes,
             lines, best * 1e3, instr / (best * 1e-3) / 1e9, instr * lanes / (best * 1e-3) / 1e9, instr * lines / (best * 1e-3) / 1e9);
    }
    hipFree(tab);
  }
  return 0;
}
