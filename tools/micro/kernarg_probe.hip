// How large may a kernel's argument block be, and what does reading it cost?  (tools/micro: measurements behind
// DESIGN's choice of where the one-workgroup interpreter's command packs travel.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
template <int N>
struct Pack {
  unsigned int w[N / 4];
};
template <int N>
__global__ void k(Pack<N> p, unsigned int* out) {
  typedef const __attribute__((address_space(4))) unsigned int* cptr;
  cptr q = (cptr)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned int s = 0;
  for (int i = 0; i < N / 4; i += 16) s += q[i];
  if (threadIdx.x == 0) *out = s;
}
template <int N>
void probe(unsigned int* d_out) {
  Pack<N> p;
  for (int i = 0; i < N / 4; ++i) p.w[i] = i % 16 == 0 ? 1u : 0u;
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, p, d_out);
  hipError_t e = hipGetLastError();
  hipError_t e2 = hipDeviceSynchronize();
  unsigned int r = 0;
  hipMemcpy(&r, d_out, 4, hipMemcpyDeviceToHost);
  auto t0 = std::chrono::steady_clock::now();
  const int reps = 200;
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, p, d_out);
    hipDeviceSynchronize();
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("kernarg %6d B: launch %s, sync %s, sum %u (want %d), launch+sync %.1f us\n", N, hipGetErrorString(e),
         hipGetErrorString(e2), r, N / 64, us);
}
int main() {
  unsigned int* d_out;
  hipMalloc(&d_out, 4);
  probe<256>(d_out);
  probe<1024>(d_out);
  probe<3584>(d_out);
  probe<4032>(d_out);
  probe<8192>(d_out);
  probe<16384>(d_out);
  probe<65536>(d_out);
  return 0;
}
