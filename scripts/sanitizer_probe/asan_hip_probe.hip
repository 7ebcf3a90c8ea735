// Does a process that carries ROCm's AddressSanitizer runtime (host-only instrumentation, -Xarch_host -fsanitize=address)
// get results back from a kernel at all?  Twenty lines, no library of ours involved: fills a device array with 7s and
// reads it back.  (profiles/r3_sanitizers.txt quotes its output next to the sanitizer runs of the host glue.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void fill(int* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 7;
}
int main() {
  const int n = 4096;
  int* d = nullptr;
  if (hipMalloc(&d, n * sizeof(int)) != hipSuccess) return 2;
  if (hipMemset(d, 0, n * sizeof(int)) != hipSuccess) return 3;
  hipLaunchKernelGGL(fill, dim3(n / 256), dim3(256), 0, 0, d, n);
  const hipError_t launch = hipGetLastError(), sync = hipDeviceSynchronize();
  std::vector<int> h(n, -1);
  const hipError_t copy = hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
  long sum = 0;
  for (int v : h) sum += v;
  std::printf("launch=%d sync=%d copy=%d sum=%ld (expected %d)\n", int(launch), int(sync), int(copy), sum, 7 * n);
  std::fflush(stdout);
  return sum == 7L * n ? 0 : 1;
}
