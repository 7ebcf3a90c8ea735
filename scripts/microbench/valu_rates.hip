// Micro-benchmark: issue cost (shader cycles per wave-instruction) of the VALU instructions the
// L-BFGS kernel is made of, on gfx950.  One wave per SIMD (4 waves/CU) or 4 waves per SIMD, so
// both the dependent-chain latency and the throughput are visible.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256

template <int KIND, bool DEP>
__global__ void bench(double* out, long long* cyc, int iters) {
  double a[8];
  const double seed = out[threadIdx.x & 63];
  for (int i = 0; i < 8; ++i) a[i] = seed + i;
  const double c1 = 1.0000001, c2 = 0.25;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double& v = DEP ? a[0] : a[i];
        if (KIND == 0) v = v * c1;                              // v_mul_f64
        if (KIND == 1) v = v + c2;                              // v_add_f64
        if (KIND == 2) v = __builtin_fma(v, c1, c2);            // v_fma_f64
        if (KIND == 3) {                                         // 2 x v_mov_b32_dpp (quad_perm)
          int lo = __double2loint(v), hi = __double2hiint(v);
          lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
          hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
          v = __hiloint2double(hi, lo);
        }
        if (KIND == 4) {                                         // 2 x v_mov_b32_dpp (row_mirror)
          int lo = __double2loint(v), hi = __double2hiint(v);
          lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true);
          hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true);
          v = __hiloint2double(hi, lo);
        }
        if (KIND == 5) {                                         // 2 x v_permlane32_swap + 2 v_mov
          unsigned lo = __double2loint(v), hi = __double2hiint(v);
          auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
          v = __hiloint2double(r1[0] ^ r1[1], r0[0] ^ r0[1]);
        }
        if (KIND == 6) {                                         // full xor-butterfly level: dpp + add
          int lo = __double2loint(v), hi = __double2hiint(v);
          lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
          hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
          v = v + __hiloint2double(hi, lo);
        }
        if (KIND == 7) {                                         // v_cndmask x2 (f64 select)
          v = (threadIdx.x & 1) ? v : a[(i + 1) & 7];
          asm volatile("" : "+v"(v));
        }
        if (KIND == 8) {                                         // v_add_u32 (32-bit int)
          int lo = __double2loint(v);
          lo += 3;
          v = __hiloint2double(__double2hiint(v), lo);
        }
        if (KIND == 9) v = 1.0 / v;                              // fp64 division expansion
        if (KIND == 10) v = __builtin_sqrt(v);                   // fp64 sqrt expansion
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, bool DEP>
void run(const char* name, int waves_per_simd, int ops_per_rep) {
  const int blocks = 256 * 4 * waves_per_simd;
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * 64);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipMemset(out, 0, sizeof(double) * blocks * 64);
  const int iters = 200;
  hipLaunchKernelGGL((bench<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((bench<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  const double n_inst = double(iters) * REP;
  printf("%-28s %s waves/SIMD=%d  cycles(clock64)/op = %7.2f   wall: %.3f ms -> %.2f ns/op/wave => SIMD-cycles/op@2.4GHz = %.2f (x%d vinst/op)\n",
         name, DEP ? "dep  " : "indep", waves_per_simd, mean / n_inst, ms, ms * 1e6 / n_inst,
         ms * 1e6 / n_inst * 2.4 / waves_per_simd, ops_per_rep);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 4}) {
    run<0, false>("v_mul_f64", w, 1);
    run<1, false>("v_add_f64", w, 1);
    run<2, false>("v_fma_f64", w, 1);
    run<3, false>("2x v_mov_dpp quad_perm", w, 2);
    run<4, false>("2x v_mov_dpp row_mirror", w, 2);
    run<5, false>("permlane32_swap pair(+xor)", w, 6);
    run<6, false>("dpp level (2 dpp + add)", w, 3);
    run<7, false>("f64 select (2 cndmask)", w, 2);
    run<8, false>("v_add_u32", w, 1);
    run<9, false>("fp64 1/x", w, 1);
    run<10, false>("fp64 sqrt", w, 1);
  }
  run<1, true>("v_add_f64", 1, 1);
  run<2, true>("v_fma_f64", 1, 1);
  run<6, true>("dpp level (2 dpp + add)", 1, 3);
  run<3, true>("2x v_mov_dpp quad_perm", 1, 2);
  return 0;
}
