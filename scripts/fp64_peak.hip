// SUPERSEDED by scripts/mfma_valu_overlap_probe.hip / mfma_stream_probe.hip: with at most 8 accumulators per
// wave this probe is latency-bound and under-measures the pipe (47-49 TFLOP/s); 16 independent accumulators
// reach 77.8.  Kept because profiles/r01_fp64_peak_microbench.log and DESIGN.md refer to it.
//
// Micro-benchmark: sustained fp64 MFMA (v_mfma_f64_16x16x4_f64) and fp64 VALU FMA rates on
// this GPU — grounds the `peak` of bench.py's roofline (the guide lists no fp64 figure).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/fp64_peak.hip -o scripts/fp64_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ long long g_clk[4];

template <int NACC>
__global__ __launch_bounds__(256) void mfma_k(double* out, int iters, double scale = 1.0) {
  long long c0 = 0, w0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    c0 = clock64();        // shader clock (s_memtime)
    w0 = wall_clock64();   // constant-rate counter
  }
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = scale * threadIdx.x * 1e-3, b = scale * (1.0 + threadIdx.x * 1e-4);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk[0] = clock64() - c0;
    g_clk[1] = wall_clock64() - w0;
  }
}

__global__ __launch_bounds__(256) void valu_k(double* out, int iters) {
  double x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3 + i;
  const double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int blocks = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_k<4>, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 /*waves*/ * iters * 4 * 2048.0;
      if (rep) printf("mfma_f64_16x16x4 NACC=4 blocks=%d: %.3f ms  %.2f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_k<8>, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 * iters * 8 * 2048.0;
      if (rep) printf("mfma_f64_16x16x4 NACC=8 blocks=%d: %.3f ms  %.2f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
  }
  // zero operands: same instruction stream, far less switching power -> shows whether the rate above
  // is bounded by the matrix pipe (64 cycles / instruction) or by the power-limited clock (DVFS)
  for (int rep = 0; rep < 2; rep++) {
    const int blocks = 512;
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_k<8>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * 2048.0;
    if (rep) printf("mfma_f64_16x16x4 NACC=8 blocks=%d ZERO operands: %.3f ms  %.2f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    if (rep) {
      long long hc[4];
      hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc));
      int wrate = 0;
      hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
      const double wall_s = (double)hc[1] / ((double)wrate * 1e3);
      printf("  block 0: %lld shader cycles over %.3f ms wall (wall clock %d kHz) -> %.0f MHz shader clock; %.1f cycles per MFMA per wave\n",
             hc[0], wall_s * 1e3, wrate, hc[0] / wall_s / 1e6, (double)hc[0] / (iters * 8.0));
    }
  }
  for (int wpc = 1; wpc <= 4; wpc *= 2) {
    const int blocks = 256 * wpc;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(valu_k, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 256 * iters * 8 * 2.0;
      if (rep) printf("v_fma_f64 blocks=%d: %.3f ms  %.2f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
