// Time from launching a kernel to the host knowing that it has finished, two ways: hipStreamSynchronize, and
// the kernel's last lane writing a counter into pinned host memory that the host polls.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/sync_latency tools/sync_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ void k_work(volatile uint32_t* flag, uint32_t value, int spin) {
  // a chain of dependent operations of about `spin` steps, so that the launch is not empty
  uint32_t x = threadIdx.x;
  for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (x == 0xdeadbeefu) value++;
    if (flag) {
      __threadfence_system();
      *flag = value;
    }
  }
}

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  uint32_t* flag = nullptr;
  hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
  *flag = 0;
  const int reps = 2000;
  for (int spin : {0, 20000}) {
    // warm-up
    for (int i = 0; i < 50; i++) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, nullptr, 0u, spin); hipStreamSynchronize(st); }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) {
      hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, nullptr, 0u, spin);
      hipStreamSynchronize(st);
    }
    auto t1 = std::chrono::steady_clock::now();
    uint32_t v = *flag;
    for (int i = 0; i < reps; i++) {
      v++;
      hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, (volatile uint32_t*)flag, v, spin);
      while (*(volatile uint32_t*)flag != v) {}
    }
    auto t2 = std::chrono::steady_clock::now();
    hipStreamSynchronize(st);
    printf("spin %d: launch + hipStreamSynchronize %.2f us, launch + polled flag %.2f us\n", spin,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
           std::chrono::duration<double, std::micro>(t2 - t1).count() / reps);
  }
  return 0;
}
