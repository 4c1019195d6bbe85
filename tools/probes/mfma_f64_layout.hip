// Probe (GPU box): operand / result layout of v_mfma_f64_16x16x4f64 as seen through the builtin.
// A (16x4) x B (4x16): record k = 0 only: a_l = (l % 16) + 1 for l / 16 == 0, b_l likewise  =>  D[i][j] = (i+1)(j+1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int krec) {
  const int l = threadIdx.x;
  const double a = (l / 16 == krec) ? (double)(l % 16 + 1) : 0.0;
  const double b = (l / 16 == krec) ? (double)(100 * (l % 16 + 1)) : 0.0;
  v4f64 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = acc[r];
}
int main() {
  double* d;
  hipMalloc(&d, 256 * 8);
  double h[256];
  for (int krec = 0; krec < 4; krec += 3) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, krec);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok1 = 1, ok2 = 1;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 4; r++) {
        const double v = h[l * 4 + r];
        // D[i][j] = a_i * b_j = (i+1) * 100 (j+1)
        const int i1 = 4 * (l / 16) + r, j1 = l % 16;  // candidate 1
        const int i2 = (l / 16) + 4 * r, j2 = l % 16;  // candidate 2
        if (v != (double)((i1 + 1) * 100 * (j1 + 1))) ok1 = 0;
        if (v != (double)((i2 + 1) * 100 * (j2 + 1))) ok2 = 0;
      }
    printf("krec %d: row = 4*(l/16)+r, col = l%%16: %s;  row = l/16 + 4r, col = l%%16: %s\n", krec, ok1 ? "yes" : "no", ok2 ? "yes" : "no");
    printf("  lane 0: %g %g %g %g | lane 1: %g %g | lane 16: %g %g %g %g | lane 17: %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[64], h[65], h[66], h[67], h[68]);
  }
  return 0;
}
