// Probe: what does the "one 512-B operand fragment from L2 per MFMA" pattern of the fused posterior
// kernel sustain on its own?  Each wave walks a 1 MB fragment stream (L2-resident, shared by all waves)
// through an 8-deep register ring into 16 accumulators — exactly the inner loop of the variance GEMM,
// without the distance/kernel-function stages.  Variants: 16x16x4 (1 MFMA per fragment), 4x4x4 (4 MFMAs
// per fragment), and a 2-tiles-per-wave form (2 MFMAs per fragment: half the fragment traffic per flop).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: 16x16x4, 1: 4x4x4 x4, 2: 16x16x4 two tiles per wave
__global__ __launch_bounds__(256, 2) void stream_k(const double* __restrict__ frag, int nfrag, int reps, double* out) {
  const int l = threadIdx.x & 63;
  d4 acc[16], acc2[MODE == 2 ? 16 : 1];
  for (int i = 0; i < 16; i++) acc[i] = (d4){0, 0, 0, 0};
  for (int i = 0; i < (MODE == 2 ? 16 : 1); i++) acc2[i] = (d4){0, 0, 0, 0};
  double a = 1e-3 * l, a2 = 2e-3 * l;
  for (int rep = 0; rep < reps; rep++) {
    const double* rf = frag + l;
    double ring[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ring[i] = rf[i * 64];
    for (int base = 0; base + 64 <= nfrag - 8; base += 64) {
#pragma unroll
      for (int i = 0; i < 64; i++) {
        const double b = ring[i % 8];
        if (MODE == 0) {
          acc[i % 16] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % 16], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
          for (int s = 0; s < 4; s++) acc[i % 16][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i % 16][s], 0, 0, 0);
        } else {
          acc[i % 16] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % 16], 0, 0, 0);
          acc2[i % 16] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc2[i % 16], 0, 0, 0);
        }
        ring[i % 8] = rf[(i + 8) * 64];
        if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
      }
      rf += 64 * 64;
    }
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (MODE == 2 ? 16 : 1); i++) s += acc2[i][0] + acc2[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int nfrag = 2112 + 8;  // ~1.08 MB, the n = 512 stream
  double *frag, *out;
  hipMalloc(&frag, sizeof(double) * 64 * (nfrag + 64));
  hipMemset(frag, 0, sizeof(double) * 64 * (nfrag + 64));
  hipMalloc(&out, sizeof(double) * 256 * 8192);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 8;
  const int usable = ((nfrag - 8) / 64) * 64;
  for (int mode = 0; mode < 3; mode++) {
    for (int blocks : {512, 2048, 8192}) {
      float ms = 0;
      for (int r = 0; r < 2; r++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(stream_k<0>, dim3(blocks), dim3(256), 0, 0, frag, nfrag, reps, out);
        if (mode == 1) hipLaunchKernelGGL(stream_k<1>, dim3(blocks), dim3(256), 0, 0, frag, nfrag, reps, out);
        if (mode == 2) hipLaunchKernelGGL(stream_k<2>, dim3(blocks), dim3(256), 0, 0, frag, nfrag, reps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double flops = (double)blocks * 4 * reps * usable * 2048.0 * (mode == 2 ? 2 : 1);
      const double bytes = (double)blocks * 4 * reps * usable * 512.0;
      printf("mode %d (%s) blocks=%d: %.3f ms  %.1f TFLOP/s  fragment stream %.1f TB/s\n", mode,
             mode == 0 ? "16x16x4" : mode == 1 ? "4x4x4 x4" : "16x16x4, 2 tiles/wave", blocks, ms, flops / ms / 1e9, bytes / ms / 1e9);
    }
  }
  return 0;
}
