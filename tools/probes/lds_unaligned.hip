// Probe (round 3): what does a 16-byte LDS read cost on gfx950 when its address is only 4-byte aligned, and how fast does
// one SIMD issue plain 32-bit vector instructions with 1..8 wavefronts resident?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_unaligned.hip -o tools/probes/lds_unaligned && tools/probes/lds_unaligned
// Part 1: every lane reads 16 bytes at LDS byte address  A(lane) + mis  with ONE ds_read_b128 (inline asm, so the
// compiler cannot split it), for mis = 0, 4, 8, 12 and three address patterns; the values are checked (a part that
// silently rounded the address down would return the wrong floats) and the loop is timed with s_memtime.
// Part 2: v_med3_u32 / v_and_or_b32 / v_pk_fma_f32 streams (independent chains), cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kLdsFloats = 8192;  // 32 KB image
struct F4 { float v[4]; };

template <int MODE>  // 0: ds_read_b128, 1: ds_read_b96 + ds_read_b32, 2: 2 x ds_read2_b32, 3: 4 x ds_read_b32
__device__ __forceinline__ F4 lds_read16(uint32_t addr) {
  F4 r;
  if constexpr (MODE == 0) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 t;
    asm volatile("ds_read_b128 %0, %1\n" : "=v"(t) : "v"(addr));
    r.v[0] = t.x, r.v[1] = t.y, r.v[2] = t.z, r.v[3] = t.w;
  } else if constexpr (MODE == 1) {
    typedef float f3 __attribute__((ext_vector_type(3)));
    f3 t;
    float u;
    asm volatile("ds_read_b96 %0, %2\n ds_read_b32 %1, %2 offset:12\n" : "=v"(t), "=v"(u) : "v"(addr));
    r.v[0] = t.x, r.v[1] = t.y, r.v[2] = t.z, r.v[3] = u;
  } else if constexpr (MODE == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a, b;
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n ds_read2_b32 %1, %2 offset0:2 offset1:3\n" : "=v"(a), "=v"(b) : "v"(addr));
    r.v[0] = a.x, r.v[1] = a.y, r.v[2] = b.x, r.v[3] = b.y;
  } else {
    asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n"
                 : "=v"(r.v[0]), "=v"(r.v[1]), "=v"(r.v[2]), "=v"(r.v[3]) : "v"(addr));
  }
  return r;
}

// pattern 0: lane l reads slot l (consecutive 16-byte slots: conflict free when aligned)
// pattern 1: lanes in runs of 9 share a slot, runs 5 slots apart (the k-NN walk: lanes of one cell read the same batch)
// pattern 2: pseudo-random slot within a 2 KB window per lane
__device__ __forceinline__ uint32_t pattern_addr(int pattern, int lane, int it) {
  uint32_t slot;
  if (pattern == 0) slot = lane;
  else if (pattern == 1) slot = (lane / 9) * 5;
  else slot = ((uint32_t)(lane * 2654435761u + it * 40503u) >> 7) & 127u;
  return ((slot + (uint32_t)it * 3u) & 1023u) * 16u;  // stays inside the first 16 KB + 16 B
}

template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(int pattern, int mis, int iters, float* out, unsigned long long* cyc, int* bad) {
  __shared__ float s[kLdsFloats];
  for (int i = threadIdx.x; i < kLdsFloats; i += blockDim.x) s[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  int wrong = 0;
  const uint32_t base = (uint32_t)(size_t)s;  // LDS byte address of the image (low 32 bits of the local pointer)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    const uint32_t a = pattern_addr(pattern, lane, it) + (uint32_t)mis;
    const F4 r = lds_read16<MODE>(base + a);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float want = (float)(a / 4u);
    wrong += (r.v[0] != want) + (r.v[1] != want + 1.f) + (r.v[2] != want + 2.f) + (r.v[3] != want + 3.f);
    acc += r.v[0] + r.v[3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (wrong) atomicAdd(bad, wrong);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// the same without the dependent wait / check in the loop: 8 reads in flight (throughput)
template <int MODE>
__global__ __launch_bounds__(256) void lds_tput_kernel(int pattern, int mis, int iters, float* out, unsigned long long* cyc) {
  __shared__ float s[kLdsFloats];
  for (int i = threadIdx.x; i < kLdsFloats; i += blockDim.x) s[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  const uint32_t base = (uint32_t)(size_t)s;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 8) {
    F4 r[8];
#pragma unroll
    for (int u = 0; u < 8; u++) r[u] = lds_read16<MODE>(base + pattern_addr(pattern, lane, it + u) + (uint32_t)mis);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; u++) acc += r[u].v[0] + r[u].v[3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// shader clock: s_memtime ticks per s_memrealtime tick (100 MHz) over a busy loop
__global__ void clock_kernel(unsigned long long* o) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < 200000ull) r1 = __builtin_amdgcn_s_memrealtime();  // 2 ms
  o[0] = __builtin_amdgcn_s_memtime() - c0, o[1] = r1 - r0;
}

// Part 2: KIND 0 = v_med3_u32 (6 independent values per step, as the collector), 1 = v_and_or_b32, 2 = v_pk_fma_f32,
// 3 = v_fma_f64
template <int KIND>
__global__ __launch_bounds__(64) void valu_kernel(int iters, uint32_t* out, unsigned long long* cyc) {
  uint32_t k[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  double d[8];
#pragma unroll
  for (int j = 0; j < 8; j++) k[j] = threadIdx.x * 77u + j * 1000u, p[j] = f2{(float)j, 1.0f + threadIdx.x}, d[j] = j + 0.5 * threadIdx.x;
  uint32_t x = threadIdx.x * 2654435761u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {  // 32 instructions per trip
      if constexpr (KIND == 0) {
#pragma unroll
        for (int j = 7; j >= 1; j--) asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[j - 1]), "v"(k[j]), "v"(x));
        asm volatile("v_min_u32 %0, %1, %2" : "=v"(k[0]) : "v"(k[0]), "v"(x));
      } else if constexpr (KIND == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[j]), "v"(x), "v"(k[(j + 1) & 7]));
      } else if constexpr (KIND == 2) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(p[(j + 1) & 7]), "v"(p[(j + 2) & 7]));
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(d[(j + 1) & 7]), "v"(d[(j + 2) & 7]));
      }
      x += 0x9E3779B9u;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  uint32_t r = x;
#pragma unroll
  for (int j = 0; j < 8; j++) r ^= k[j] ^ (uint32_t)p[j].x ^ (uint32_t)d[j];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

int main() {
  float* out;
  unsigned long long* cyc;
  int* bad;
  CHECK(hipMalloc(&out, 4096 * 256 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 8));
  CHECK(hipMalloc(&bad, 4));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  const int iters = 4096;
  unsigned long long* clk;
  CHECK(hipMalloc(&clk, 16));
  hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(1), 0, 0, clk);
  unsigned long long hclk[2];
  CHECK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
  printf("shader clock (idle chip, one lane): %.3f GHz\n", ghz);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // wall time of a launch in shader cycles at that clock (the stamped wavefront of block 0 is the OLDEST one on its SIMD and
  // wins the issue arbitration: its own duration says nothing about throughput)
  auto wall_cycles = [&](auto&& launch) {
    launch();  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms * 1e-3 * ghz * 1e9;
  };
  const char* mode_name[4] = {"ds_read_b128", "ds_read_b96+b32", "2 x ds_read2_b32", "4 x ds_read_b32"};
  const char* pat_name[3] = {"consecutive slots", "runs of 9 lanes share a slot", "random slot in 2 KB"};
  printf("\n== part 1: 16-byte LDS reads, cycles (s_memtime) per read instruction group; one workgroup of 4 waves per CU ==\n");
  for (int mode = 0; mode < 4; mode++)
    for (int pattern = 0; pattern < 3; pattern++) {
      printf("%-18s %-30s", mode_name[mode], pat_name[pattern]);
      for (int mis = 0; mis < 16; mis += 4) {
        unsigned long long c_lat = 0, c_tp = 0;
        int nb = 0;
        CHECK(hipMemset(bad, 0, 4));
        auto run = [&](auto kern_lat, auto kern_tp) {
          hipLaunchKernelGGL(kern_lat, dim3(cus), dim3(256), 0, 0, pattern, mis, iters, out, cyc, bad);
          CHECK(hipDeviceSynchronize());
          CHECK(hipMemcpy(&c_lat, cyc, 8, hipMemcpyDeviceToHost));
          c_tp = (unsigned long long)wall_cycles([&] { hipLaunchKernelGGL(kern_tp, dim3(cus * 4), dim3(256), 0, 0, pattern, mis, iters * 4, out, cyc); }) / 4;  // 16 waves per CU
        };
        if (mode == 0) run(lds_kernel<0>, lds_tput_kernel<0>);
        else if (mode == 1) run(lds_kernel<1>, lds_tput_kernel<1>);
        else if (mode == 2) run(lds_kernel<2>, lds_tput_kernel<2>);
        else run(lds_kernel<3>, lds_tput_kernel<3>);
        CHECK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
        // throughput: 16 waves per CU share one LDS: cycles per wave-read seen by the LDS = wave cycles / iters / 16
        printf(" | +%2d B: lat %6.1f, tput %5.2f cyc/CU%s", mis, (double)c_lat / iters, (double)c_tp / iters / 16.0, nb ? " WRONG DATA" : "");
      }
      printf("\n");
    }
  printf("\n== part 2: vector issue rate, cycles per wave-instruction per SIMD (1 = one per cycle) ==\n");
  const char* kind_name[4] = {"v_med3_u32 / v_min_u32 chain of 8", "v_and_or_b32 x 8", "v_pk_fma_f32 x 8", "v_fma_f64 x 8"};
  for (int kind = 0; kind < 4; kind++) {
    printf("%-36s", kind_name[kind]);
    for (int wps : {1, 2, 3, 4, 5, 8}) {  // wavefronts per SIMD: 4 * wps one-wave workgroups per CU
      const int vit = 16384;
      auto launch = [&](auto kern) { return wall_cycles([&] { hipLaunchKernelGGL(kern, dim3(cus * 4 * wps), dim3(64), 0, 0, vit, (uint32_t*)out, cyc); }); };
      double c;
      if (kind == 0) c = launch(valu_kernel<0>);
      else if (kind == 1) c = launch(valu_kernel<1>);
      else if (kind == 2) c = launch(valu_kernel<2>);
      else c = launch(valu_kernel<3>);
      // every SIMD ran wps waves of vit * 36 vector instructions each
      printf(" | %d w/SIMD: %5.2f", wps, c / (vit * 36.0) / wps);
    }
    printf("\n");
  }
  return 0;
}
