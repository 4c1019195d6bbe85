// synth_kernels.hip — device side of the synthetic scan-pair generator (synth.h). Bench/test
// workload only; bit-identical to loamx_synth_scan_host.
#include "loamx_internal.h"
#include "synth.h"

namespace loamx {
namespace {
__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, uint64_t first_pair, size_t n_pairs, uint32_t H,
                                                    uint32_t W, double sigma, double* __restrict__ xyz) {
  const size_t N = (size_t)H * W;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_pairs * 2 * N) return;
  const size_t scan = gid / N, pt = gid - scan * N;
  const uint64_t pair = first_pair + scan / 2;
  const uint32_t which = (uint32_t)(scan & 1);
  const loamx_synth::Pose7 pose = loamx_synth::pair_pose(seed, pair);
  double out[3];
  loamx_synth::scan_point(seed, pair, which, pose, (uint32_t)(pt / W), (uint32_t)(pt % W), H, W, sigma, out);
  xyz[3 * gid] = out[0], xyz[3 * gid + 1] = out[1], xyz[3 * gid + 2] = out[2];
}
}  // namespace

void launch_synth_pairs(uint64_t seed, uint64_t first_pair, size_t n_pairs, uint32_t H, uint32_t W, double sigma,
                        double* d_xyz, hipStream_t s) {
  const size_t total = n_pairs * 2 * (size_t)H * W;
  if (total == 0) return;
  hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, seed, first_pair, n_pairs,
                     H, W, sigma, d_xyz);
}
}  // namespace loamx
