#include <hip/hip_runtime.h>
template <int W> __global__ void kk(double* p, int n) { if (threadIdx.x < n) p[threadIdx.x] = W; }
template <int W> int launch(double* p, int n, hipStream_t s) {
  auto kern = kk<W>;
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, s, p, n);
  return (int)hipGetLastError();
}
int go(double* p, int n, hipStream_t s) { return launch<8>(p, n, s); }
