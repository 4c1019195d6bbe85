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
// CHECK_FINITE (loamx.h: "Non-finite input"): does any of the first n[set * pitch] points of a set (stride points apart;
// n == nullptr: every one of the `stride` points) hold a coordinate that is not finite? One flag word, set, never cleared.
// `scalars` != 0: one flat array of that many numbers instead (an array that is not made of points: the initial poses).
template <typename T>
__global__ __launch_bounds__(256) void finite_kernel(const T* __restrict__ pts, const uint32_t* __restrict__ n, size_t stride, uint32_t pitch,
                                                     uint32_t* __restrict__ flag, size_t scalars) {
  const size_t set = blockIdx.y;
  const size_t cnt = n ? (n[set * pitch] < stride ? n[set * pitch] : stride) : stride;
  const size_t end = scalars ? scalars : cnt * 3;
  const T* __restrict__ p = pts + set * pitch * stride * 3;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (size_t)gridDim.x * blockDim.x) {
    const double v = (double)p[i];
    bad = bad || !(fabs(v) <= 1.7976931348623157e308);
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
}  // namespace

void launch_check_finite(const void* d_pts, bool f32, const uint32_t* d_n, size_t n_sets, size_t stride, uint32_t pitch, uint32_t* d_flag,
                         hipStream_t s) {
  if (n_sets == 0 || stride == 0) return;
  const unsigned bx = (unsigned)((stride * 3 + 255) / 256 < 64 ? (stride * 3 + 255) / 256 : 64);
  for (size_t s0 = 0; s0 < n_sets; s0 += 32768) {  // (grid.y limit)
    const size_t ns = n_sets - s0 < 32768 ? n_sets - s0 : 32768;
    if (f32)
      hipLaunchKernelGGL(finite_kernel<float>, dim3(bx, (unsigned)ns), dim3(256), 0, s, static_cast<const float*>(d_pts) + s0 * pitch * stride * 3,
                         d_n ? d_n + s0 * pitch : nullptr, stride, pitch, d_flag, (size_t)0);
    else
      hipLaunchKernelGGL(finite_kernel<double>, dim3(bx, (unsigned)ns), dim3(256), 0, s, static_cast<const double*>(d_pts) + s0 * pitch * stride * 3,
                         d_n ? d_n + s0 * pitch : nullptr, stride, pitch, d_flag, (size_t)0);
  }
}
// exactly n_scalars doubles (nothing is rounded up to whole points: the caller's buffer ends there)
void launch_check_finite_scalars(const double* d_v, size_t n_scalars, uint32_t* d_flag, hipStream_t s) {
  if (n_scalars == 0) return;
  const unsigned bx = (unsigned)((n_scalars + 255) / 256 < 64 ? (n_scalars + 255) / 256 : 64);
  hipLaunchKernelGGL(finite_kernel<double>, dim3(bx, 1u), dim3(256), 0, s, d_v, static_cast<const uint32_t*>(nullptr), (size_t)0, 0u, d_flag, n_scalars);
}

void launch_synth_pairs(uint64_t seed, uint64_t first_pair, size_t n_pairs, uint32_t H, uint32_t W, double sigma,
                        double* d_xyz, hipStream_t s) {
  const size_t total = n_pairs * 2 * (size_t)H * W;
  if (total == 0) return;
  hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, seed, first_pair, n_pairs,
                     H, W, sigma, d_xyz);
}
}  // namespace loamx
