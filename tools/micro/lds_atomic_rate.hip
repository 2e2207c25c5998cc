// Microbenchmark: cost of LDS atomics by type on gfx950 -- ds_add_f64 vs ds_add_u64 vs ds_add_f32/u32, plain b64 read/write.
// One workgroup of 256 threads per CU slot, every lane its own address (stride 1: no bank conflicts beyond the width).
// build: hipcc -O3 --offload-arch=gfx950 lds_atomic_rate.hip -o lds_atomic_rate ; run: ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, int spread) {
  __shared__ double s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.0;
  __syncthreads();
  const int base = (threadIdx.x * spread) & 2047;
  double acc = 0.0;
  double v = 1.0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = (base + u * 256) & 4095;
      if (MODE == 0) atomicAdd(&s[a], v);
      if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(&s[a]), (unsigned long long)(threadIdx.x + it));
      if (MODE == 2) atomicAdd(reinterpret_cast<float*>(&s[a]), (float)v);
      if (MODE == 3) atomicAdd(reinterpret_cast<unsigned int*>(&s[a]), (unsigned)(threadIdx.x + it));
      if (MODE == 4) acc += s[a];
      if (MODE == 5) s[a] = v + it;
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x] + acc;
}

template <int MODE>
void run(const char* name, int spread) {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 2048);
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, spread);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, spread);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per CU: blocks/256 WGs per CU * 4 waves * iters * 8
  const double winst_per_cu = (blocks / 256.0) * 4 * iters * 8.0;
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-10s spread %d: %.3f ms  -> %.1f cycles per wave instruction per CU (at 2.4 GHz)\n", name, spread, ms, cycles / winst_per_cu);
  hipFree(out);
}

int main() {
  for (int spread : {1, 2}) {
    run<0>("add_f64", spread);
    run<1>("add_u64", spread);
    run<2>("add_f32", spread);
    run<3>("add_u32", spread);
    run<4>("read_b64", spread);
    run<5>("write_b64", spread);
  }
  return 0;
}
