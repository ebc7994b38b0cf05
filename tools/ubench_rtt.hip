// ubench_rtt.hip -- what a host round trip per lock period costs (the lock-period walk reads every search's outcome back: DESIGN 8, config 5 at the prescribed noise):
// (a) a small kernel + a 256-byte device-to-host copy + hipStreamSynchronize, (b) the same with hipEventSynchronize on a spinning event, (c) the kernel writes the result into
// page-locked host memory itself and the host polls a sequence number.  Microseconds per round trip, median of 2,000.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void work(int *out, int seq) { if (threadIdx.x < 64) out[threadIdx.x] = seq; }
__global__ void work_host(volatile int *host, int seq) { if (threadIdx.x < 63) host[threadIdx.x + 1] = seq; __threadfence_system(); if (threadIdx.x == 0) host[0] = seq; }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main()
{
  hipStream_t s; hipStreamCreate(&s);
  int *d, *h; hipMalloc((void **)&d, 256); hipHostMalloc((void **)&h, 256); for (int i = 0; i < 64; i++) h[i] = 0;
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int N = 2000; std::vector<double> a, b, c;
  for (int i = 1; i <= N; i++) {
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d, i); hipMemcpyAsync(h, d, 256, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
    a.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  for (int i = 1; i <= N; i++) {
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, d, i); hipMemcpyAsync(h, d, 256, hipMemcpyDeviceToHost, s); hipEventRecord(ev, s); hipEventSynchronize(ev);
    b.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  volatile int *hv = h; hv[0] = 0;
  for (int i = 1; i <= N; i++) {
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(work_host, dim3(1), dim3(64), 0, s, hv, i + N);
    while (hv[0] != i + N) { }
    c.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  hipStreamSynchronize(s);
  printf("{\"kernel_copy_stream_sync_us\": %.1f, \"kernel_copy_event_sync_us\": %.1f, \"kernel_writes_pinned_host_polls_us\": %.1f}\n", med(a), med(b), med(c));
  return 0;
}
