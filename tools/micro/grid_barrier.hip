// Latency of an in-kernel grid barrier on MI355X for 257 / 512 / 1280 co-resident workgroups: flat (one counter) against
// two-level (8 group counters + a root), with and without an agent-scope release/acquire of ordinary data around it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Bar {
  unsigned int grp[8][32];   // one 128-byte line per group
  unsigned int root[32];
  unsigned int gen[32];
};
__device__ __forceinline__ void flat_barrier(Bar* b, unsigned int nwg, unsigned int& phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++phase;
    const unsigned int old = __hip_atomic_fetch_add(&b->root[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg * phase - 1) __hip_atomic_store(&b->gen[0], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
__device__ __forceinline__ void tree_barrier(Bar* b, unsigned int nwg, unsigned int& phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++phase;
    const unsigned int g = blockIdx.x & 7u;
    const unsigned int members = nwg / 8 + (g < nwg % 8 ? 1 : 0);
    const unsigned int old = __hip_atomic_fetch_add(&b->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == members * phase - 1) {
      const unsigned int o2 = __hip_atomic_fetch_add(&b->root[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 == 8 * phase - 1) __hip_atomic_store(&b->gen[0], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    while (__hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
template <int MODE>  // 0 flat, 1 tree, 2 tree + release/acquire fences, 3 tree + every workgroup writes a line (sc1) and reads its neighbour's
__global__ __launch_bounds__(256) void k(Bar* b, int iters, double* data, double* sink) {
  unsigned int phase = 0;
  double acc = 0.0;
  const unsigned int nwg = gridDim.x;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 3) {
      __hip_atomic_store(&data[16 * blockIdx.x + (threadIdx.x & 15)], (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (MODE == 0) flat_barrier(b, nwg, phase); else tree_barrier(b, nwg, phase);
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (MODE == 3) {
      const unsigned int nb = (blockIdx.x + 1) % nwg;
      acc += __hip_atomic_load(&data[16 * nb + (threadIdx.x & 15)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}
template <int MODE>
void run(const char* name, int nwg, Bar* d_bar, double* d_data, double* d_sink) {
  const int iters = 2000;
  hipMemset(d_bar, 0, sizeof(Bar));
  hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, d_bar, 10, d_data, d_sink);
  hipDeviceSynchronize();
  hipMemset(d_bar, 0, sizeof(Bar));
  hipEvent_t a, e;
  hipEventCreate(&a);
  hipEventCreate(&e);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, d_bar, iters, d_data, d_sink);
  hipEventRecord(e, 0);
  hipEventSynchronize(e);
  float ms = 0;
  hipEventElapsedTime(&ms, a, e);
  printf("%-34s %5d workgroups: %.2f us per barrier (%s)\n", name, nwg, 1e3 * ms / iters, hipGetErrorString(hipGetLastError()));
}
int main() {
  Bar* d_bar;
  double *d_data, *d_sink;
  hipMalloc(&d_bar, sizeof(Bar));
  hipMalloc(&d_data, 8 * 16 * 2048);
  hipMalloc(&d_sink, 8 * 2048);
  hipMemset(d_data, 0, 8 * 16 * 2048);
  for (int nwg : {64, 257, 512, 1024}) {
    run<0>("flat counter", nwg, d_bar, d_data, d_sink);
    run<1>("8 groups + root", nwg, d_bar, d_data, d_sink);
    run<2>("8 groups + root, release/acquire", nwg, d_bar, d_data, d_sink);
    run<3>("8 groups + root, sc1 line exchange", nwg, d_bar, d_data, d_sink);
  }
  return 0;
}
