// Probe: throughput of v_mfma_f64_4x4x4_4b_f64 vs v_mfma_f64_16x16x4_f64 (flops/s), and the lane
// mapping of the 4x4x4 (4 blocks) form, determined empirically with one-hot operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k44(double* out, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void probe(const double* A, const double* B, double* D, int cbsz, int abid, int blgp) {
  const int l = threadIdx.x;
  double d = 0.0;
  // builtin needs immediate modifiers: enumerate the few combinations we care about
  if (cbsz == 0 && blgp == 0) d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
  else if (cbsz == 2 && abid == 0 && blgp == 0) d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 2, 0, 0);
  else if (cbsz == 0 && blgp == 1) d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 1);
  else if (cbsz == 0 && blgp == 4) d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 4);
  D[l] = d;
}

int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int blocks = 256; blocks <= 2048; blocks *= 2) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k44<8>, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 4 * iters * 8 * 512.0;
      if (rep) printf("mfma_f64_4x4x4_4b NACC=8 blocks=%d: %.3f ms  %.2f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
  }
  // lane mapping: for each (source lane of A, source lane of B) with a one-hot pair, which D lanes light up?
  double hA[64], hB[64], hD[64], *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  const int modes[4][3] = {{0, 0, 0}, {2, 0, 0}, {0, 0, 1}, {0, 0, 4}};
  for (int m = 0; m < 4; m++) {
    printf("mode cbsz=%d abid=%d blgp=%d: D lanes hit by (A lane, B lane) one-hot pairs [subset]\n", modes[m][0], modes[m][1], modes[m][2]);
    for (int la = 0; la < 64; la += 1) {
      if (!(la < 8 || la == 16 || la == 17 || la == 32 || la == 48)) continue;
      for (int lb = 0; lb < 64; lb++) {
        if (!(lb < 8 || lb == 16 || lb == 20 || lb == 32 || lb == 48)) continue;
        for (int i = 0; i < 64; i++) { hA[i] = (i == la); hB[i] = (i == lb); }
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, modes[m][0], modes[m][1], modes[m][2]);
        hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
        int cnt = 0; char buf[256]; int pos = 0;
        for (int i = 0; i < 64; i++) if (hD[i] != 0.0) { cnt++; if (pos < 200) pos += snprintf(buf + pos, 256 - pos, "%d ", i); }
        if (cnt) printf("  A%d B%d -> D{%s}\n", la, lb, buf);
      }
    }
  }
  return 0;
}
