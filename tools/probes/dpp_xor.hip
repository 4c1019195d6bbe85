// Probe: lane exchanges v[l ^ J] for J = 1..32 without the LDS pipe (DPP row operations, gfx950 permlane swaps)
// against __shfl_xor. Build: hipcc --offload-arch=gfx950 -O2 tools/probes/dpp_xor.hip -o tools/probes/dpp_xor
#include <hip/hip_runtime.h>
#include <cstdio>

template <int J>
__device__ __forceinline__ int xor_lane_b32(int v) {
  if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  else if constexpr (J == 4) {
    int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);  // row_shl:4 -> quads 0, 2 take lane + 4
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);   // row_shr:4 -> quads 1, 3 take lane - 4
  } else if constexpr (J == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);  // row_ror:8
  else if constexpr (J == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = rows [v0 v0 v2 v2], r[1] = [v1 v1 v3 v3]
    return (threadIdx.x & 16) ? r[0] : r[1];
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // r[0] = [lo lo], r[1] = [hi hi]
    return (threadIdx.x & 32) ? r[0] : r[1];
  }
}

// inclusive prefix sum over the 64 lanes: row_shr 1, 2, 4, 8 inside a row, then row_bcast15 / row_bcast31
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1 (lanes without a source add 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast31 -> rows 2, 3
  return v;
}

__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  const int v = lane * 7 + 3;
  out[0 * 64 + lane] = xor_lane_b32<1>(v) - __shfl_xor(v, 1);
  out[1 * 64 + lane] = xor_lane_b32<2>(v) - __shfl_xor(v, 2);
  out[2 * 64 + lane] = xor_lane_b32<4>(v) - __shfl_xor(v, 4);
  out[3 * 64 + lane] = xor_lane_b32<8>(v) - __shfl_xor(v, 8);
  out[4 * 64 + lane] = xor_lane_b32<16>(v) - __shfl_xor(v, 16);
  out[5 * 64 + lane] = xor_lane_b32<32>(v) - __shfl_xor(v, 32);
  int w = (lane * 2654435761u >> 27) & 7, ref = w;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(ref, off);
    if (lane >= off) ref += t;
  }
  out[6 * 64 + lane] = wave_incl_scan(w) - ref;
}

int main() {
  int* d;
  hipMalloc(&d, 7 * 64 * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[7 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const int js[7] = {1, 2, 4, 8, 16, 32, 0};  // 0: the prefix sum
  int bad = 0;
  for (int k = 0; k < 7; k++) {
    int wrong = 0;
    for (int l = 0; l < 64; l++) wrong += h[k * 64 + l] != 0;
    printf("xor %2d: %s (%d lanes differ)\n", js[k], wrong ? "MISMATCH" : "ok", wrong);
    bad += wrong;
  }
  return bad ? 1 : 0;
}
