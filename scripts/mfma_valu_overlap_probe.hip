// Does fp64 VALU work overlap with v_mfma_f64_16x16x4_f64 on gfx950, or do both run on the same fp64
// datapath?  Each wave runs ITER x { 16 independent MFMAs, NV FMAs interleaved 1:NV/16 } with the FMAs
// in fp64, fp32 or absent; 2 waves per SIMD (as the fused posterior kernel).  Prints time per variant
// and the time the pure-VALU part takes alone, so "sum" vs "max" behaviour can be read off directly.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV, bool MF>  // MODE 0: no VALU, 1: fp64 FMA, 2: fp32 FMA; MF: issue the MFMAs
__global__ __launch_bounds__(256, 2) void probe(double* out, int iters) {
  d4 acc[16];
  for (int j = 0; j < 16; j++) acc[j] = (d4){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double x[8];
  float xf[8];
  for (int u = 0; u < 8; u++) { x[u] = 0.5 + u + threadIdx.x; xf[u] = 0.5f + u + threadIdx.x; }
  const double m = 0.999999, c = 1e-7;
  const float mf = 0.999999f, cf = 1e-7f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (MF) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV / 16; v++) {
        const int u = (j * (NV / 16) + v) % 8;
        if (MODE == 1) x[u] = fma(x[u], m, c);
        if (MODE == 2) xf[u] = fmaf(xf[u], mf, cf);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = 0;
  for (int j = 0; j < 16; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int u = 0; u < 8; u++) s += x[u] + xf[u];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV, bool MF>
static void run(const char* name, double* d, int grid, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NV, MF>), dim3(grid), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, NV, MF>), dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_flops = MF ? (double)grid * 4 * iters * 16 * 2.0 * 16 * 16 * 4 : 0.0;
  printf("%-34s %8.3f ms   MFMA %6.1f TFLOP/s\n", name, ms, mfma_flops / ms * 1e-9);
}

int main() {
  const int grid = 512, iters = 20000;  // 2 blocks per CU -> 2 waves per SIMD
  double* d;
  hipMalloc(&d, sizeof(double) * grid * 256);
  run<0, 0, true>("MFMA only", d, grid, iters);
  run<1, 32, false>("fp64 FMA x32 only", d, grid, iters);
  run<1, 32, true>("MFMA + fp64 FMA x32", d, grid, iters);
  run<1, 64, false>("fp64 FMA x64 only", d, grid, iters);
  run<1, 64, true>("MFMA + fp64 FMA x64", d, grid, iters);
  run<1, 128, false>("fp64 FMA x128 only", d, grid, iters);
  run<1, 128, true>("MFMA + fp64 FMA x128", d, grid, iters);
  run<2, 64, false>("fp32 FMA x64 only", d, grid, iters);
  run<2, 64, true>("MFMA + fp32 FMA x64", d, grid, iters);
  run<2, 128, false>("fp32 FMA x128 only", d, grid, iters);
  run<2, 128, true>("MFMA + fp32 FMA x128", d, grid, iters);
  return 0;
}
