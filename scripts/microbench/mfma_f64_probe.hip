// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand / result lane layout, accumulation order and
// rounding (compared against candidate host emulations), and issue rate.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void one_mfma(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  v4d acc;
  for (int r = 0; r < 4; ++r) acc[r] = c[l * 4 + r];
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int CHAINS>
__global__ void rate(double* out, int iters) {
  const int l = threadIdx.x;
  v4d acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 4; ++r) acc[c][r] = l + r + c;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 4; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = double(t1 - t0);
}

int main() {
  std::vector<double> a(64), b(64), c(256), d(256);
  srand(7);
  auto rnd = [] { return (rand() / double(RAND_MAX) - 0.5) * 4.0; };
  for (auto& v : a) v = rnd();
  for (auto& v : b) v = rnd();
  for (auto& v : c) v = rnd() * 1e-3;
  double *da, *db, *dc, *dd;
  hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dc, 256 * 8); hipMalloc(&dd, 256 * 8);
  hipMemcpy(da, a.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(dc, c.data(), 256 * 8, hipMemcpyHostToDevice);
  one_mfma<<<1, 64>>>(da, db, dc, dd);
  hipMemcpy(d.data(), dd, 256 * 8, hipMemcpyDeviceToHost);
  // hypothesis: A[i][k] in lane i + 16k, B[k][j] in lane j + 16k, D[i][j] in lane j + 16*(i%4), reg i/4  (or row = (lane>>4) + 4 reg)
  int ok_layout1 = 0, ok_layout2 = 0, n_fma_fwd = 0, n_fma_rev = 0, n_plain = 0, n_fma_fwd2 = 0, n_sumfirst = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int j = l & 15;
      const int i1 = (l >> 4) + 4 * r;   // row = (lane>>4) + 4*reg
      const int i2 = 4 * (l >> 4) + r;   // row = 4*(lane>>4) + reg
      for (int variant = 0; variant < 2; ++variant) {
        const int i = variant ? i2 : i1;
        const double c0 = c[l * 4 + r];
        double fwd = c0, rev = c0, plain = c0;
        for (int k = 0; k < 4; ++k) fwd = std::fma(a[i + 16 * k], b[j + 16 * k], fwd);
        for (int k = 3; k >= 0; --k) rev = std::fma(a[i + 16 * k], b[j + 16 * k], rev);
        for (int k = 0; k < 4; ++k) plain = plain + a[i + 16 * k] * b[j + 16 * k];
        double prods = 0;  // sum of products first (fma chain from 0), then + c
        for (int k = 0; k < 4; ++k) prods = std::fma(a[i + 16 * k], b[j + 16 * k], prods);
        const double sumfirst = prods + c0;
        const double got = d[l * 4 + r];
        if (std::fabs(got - fwd) < 1e-9 * (1 + std::fabs(fwd))) (variant ? ok_layout2 : ok_layout1)++;
        if (variant == 0) {
          n_fma_fwd += (got == fwd);
          n_fma_rev += (got == rev);
          n_plain += (got == plain);
          n_sumfirst += (got == sumfirst);
        } else {
          n_fma_fwd2 += (got == fwd);
        }
      }
    }
  printf("layout row=(lane>>4)+4*reg: %d/256 close; layout row=4*(lane>>4)+reg: %d/256 close\n", ok_layout1, ok_layout2);
  printf("bitwise matches (layout 1): fma chain k=0..3 %d, k=3..0 %d, mul+add %d, products-first %d; (layout 2) fma fwd %d\n",
         n_fma_fwd, n_fma_rev, n_plain, n_sumfirst, n_fma_fwd2);
  double* dout;
  hipMalloc(&dout, (64 * 1024 + 1) * 8);
  const int iters = 20000;
  for (int chains = 1; chains <= 4; chains *= 2) {
    for (int blocks : {1, 1024}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (chains == 1) rate<1><<<blocks, 64>>>(dout, iters);
      if (chains == 2) rate<2><<<blocks, 64>>>(dout, iters);
      if (chains == 4) rate<4><<<blocks, 64>>>(dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double cyc; hipMemcpy(&cyc, dout + blocks * 64, 8, hipMemcpyDeviceToHost);
      printf("chains %d blocks %4d: %.1f clock64 ticks per MFMA (one wave), kernel %.3f ms -> %.2f TFLOP/s\n", chains, blocks,
             cyc / (double(iters) * chains), ms, 2.0 * 16 * 16 * 4 * double(iters) * chains * blocks / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
