// Probe: can hipStreamWaitValue64 / hipStreamWriteValue64 be used ACROSS PROCESSES on one GPU (flag word in memory that is
// exported with hipIpcGetMemHandle)?  Parent = waiter, child = writer; tried with plain hipMalloc memory and with
// hipMallocSignalMemory.  Every wait is bounded by a host watchdog that writes the value itself after 2 s.
// build: hipcc -O2 --offload-arch=gfx950 stream_wait_ipc.hip -o stream_wait_ipc ; run: ./stream_wait_ipc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <sys/wait.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%s] %s -> %s\n", who, #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void mark(unsigned long long* out, unsigned long long v) { *out = v; }

static int run(int mode) {  // 0 plain hipMalloc, 1 signal memory
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 1;
  pid_t pid = fork();  // (before any HIP call in this process)
  const char* who = pid ? "waiter" : "writer";
  if (pid == 0) {
    hipIpcMemHandle_t h;
    if (read(p2c[0], &h, sizeof(h)) != (ssize_t)sizeof(h)) return 1;
    void* flag = nullptr;
    CK(hipIpcOpenMemHandle(&flag, h, hipIpcMemLazyEnablePeerAccess));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::this_thread::sleep_for(std::chrono::milliseconds(200));  // the waiter is waiting by now
    hipError_t e = hipStreamWriteValue64(s, flag, 42ull, 0);
    printf("[writer] hipStreamWriteValue64 on the IPC mapping -> %s\n", hipGetErrorString(e));
    CK(hipStreamSynchronize(s));
    char ok = 1;
    (void)!write(c2p[1], &ok, 1);
    CK(hipIpcCloseMemHandle(flag));
    _exit(0);
  }
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("[waiter] mode %d (%s): CanUseStreamWaitValue = %d\n", mode, mode ? "signal memory" : "plain hipMalloc", can);
  unsigned long long* flag = nullptr;
  if (mode == 0) CK(hipMalloc(reinterpret_cast<void**>(&flag), 8));
  else CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&flag), 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  hipError_t eh = hipIpcGetMemHandle(&h, flag);
  printf("[waiter] hipIpcGetMemHandle -> %s\n", hipGetErrorString(eh));
  if (eh != hipSuccess) { kill(pid, SIGKILL); waitpid(pid, nullptr, 0); return 0; }
  (void)!write(p2c[1], &h, sizeof(h));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned long long* out;
  CK(hipHostMalloc(reinterpret_cast<void**>(&out), 8));
  *out = 0;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t ew = hipStreamWaitValue64(s, flag, 42ull, hipStreamWaitValueEq, ~0ull);
  printf("[waiter] hipStreamWaitValue64 enqueue -> %s\n", hipGetErrorString(ew));
  if (ew == hipSuccess) {
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, s, out, 7ull);
    bool released = false;
    while (true) {  // watchdog: never wait on the stream for more than 2 s
      if (hipStreamQuery(s) == hipSuccess) break;
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > 2.0 && !released) {
        printf("[waiter] watchdog: releasing the wait from the host\n");
        unsigned long long v = 42;
        hipMemcpy(flag, &v, 8, hipMemcpyHostToDevice);
        released = true;
      }
      if (el > 6.0) { printf("[waiter] still blocked after 6 s -- giving up\n"); break; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("[waiter] kernel behind the wait ran: out = %llu after %.1f ms (the writer sleeps 200 ms)%s\n", *out, ms,
           released ? "  [released by the watchdog]" : "");
  }
  char ok = 0;
  (void)!read(c2p[0], &ok, 1);
  waitpid(pid, nullptr, 0);
  hipFree(flag);
  return 0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  // each mode in its own pair of processes (fork must come before HIP initialises)
  for (int mode = 0; mode < 2; ++mode) {
    pid_t p = fork();
    if (p == 0) _exit(run(mode));
    int st = 0;
    waitpid(p, &st, 0);
    printf("mode %d exit %d\n", mode, WEXITSTATUS(st));
  }
  return 0;
}
