// Probe (round 5): vector issue rate of plain 2-source 32-bit operations beside the 3-source / packed /
// FP64 ones the round-3 probe timed, at 1..8 wavefronts per SIMD, and a check of the row-local DPP forms the row-per-ring
// selection kernel relies on (row_newbcast, row_mirror, row_half_mirror, bank-masked v_min/v_max with a DPP source).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/valu_issue.hip -o tools/probes/valu_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

__global__ void clock_kernel(unsigned long long* o) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < 200000ull) r1 = __builtin_amdgcn_s_memrealtime();  // 2 ms
  o[0] = __builtin_amdgcn_s_memtime() - c0, o[1] = r1 - r0;
}

enum Kind {
  K_ADD_U32, K_AND_B32, K_LSHL_B32, K_FMA_F32, K_MIN_U32, K_CNDMASK, K_OR3, K_AND_OR, K_MED3, K_LSHL_OR, K_BFE, K_PERM,
  K_MUL_LO, K_FFBL, K_BCNT, K_MOV_DPP, K_MIN_DPP, K_ADD_DPP, K_CMP_U32, K_CMP_F64, K_ADDC, K_LSHL_B64, K_ADD_F64, K_MUL_F64,
  K_FMA_F64, K_MIN_F64, K_PK_FMA_F32, K_PK_MUL_F32, K_PK_ADD_F32, K_MUL_F32, K_ADD_F32, K_AND_B32_SGPR, K_XOR_B32, K_SUB_U32,
  K_MAX3_U32,
  K_OR_B32, K_MAX_U32, K_MIN_I32, K_MOV_B32, K_NOT_B32, K_LSHR_B32, K_LSHLV_B32, K_MUL_U24, K_MAD_U24, K_ADD3, K_LSHL_ADD, K_BFI, K_ALIGNBIT, K_AND_LIT, K_AND_INL, K_ADD_INL, K_ADD_SGPR, K_ADD_CO, K_MAX_F32, K_MIN_F32, K_MED3_F32, K_MAX3_F32, K_MIN3_F32, K_FMAC_F32, K_SUB_F32, K_CMP_F32, K_CMP_SG, K_CND_E64, K_CND_PAIR, K_CVT_F32_U32, K_MUL_HI, K_MOV_DPP_QP, K_AND_DPP, K_SDWA, K_XNOR, K_SUBREV, K_ADD_LSHL, K_MIX_ADD_MIN, K_MIX_ADD_F64, K_MIX_FMA_MED3, K_COUNT
};
static const char* kind_name[K_COUNT] = {
  "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_fma_f32", "v_min_u32", "v_cndmask_b32 (vcc)", "v_or3_b32", "v_and_or_b32", "v_med3_u32",
  "v_lshl_or_b32", "v_bfe_u32", "v_perm_b32", "v_mul_lo_u32", "v_ffbl_b32", "v_bcnt_u32_b32", "v_mov_b32 dpp row_shr:1",
  "v_min_u32 dpp row_shr:1", "v_add_u32 dpp row_shr:1", "v_cmp_gt_u32 (vcc)", "v_cmp_gt_f64 (vcc)", "v_addc_co_u32", "v_lshlrev_b64",
  "v_add_f64", "v_mul_f64", "v_fma_f64", "v_min_f64", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "v_add_f32",
  "v_and_b32 (sgpr src)", "v_xor_b32", "v_sub_u32", "v_max3_u32",
  "v_or_b32", "v_max_u32", "v_min_i32", "v_mov_b32", "v_not_b32", "v_lshrrev_b32", "v_lshlrev_b32 (vgpr shift)", "v_mul_u32_u24", "v_mad_u32_u24", "v_add3_u32", "v_lshl_add_u32", "v_bfi_b32", "v_alignbit_b32", "v_and_b32 (32-bit literal)", "v_and_b32 (inline const)", "v_add_u32 (inline const)", "v_add_u32 (sgpr src)", "v_add_co_u32 (writes vcc)", "v_max_f32", "v_min_f32", "v_med3_f32", "v_max3_f32", "v_min3_f32", "v_fmac_f32", "v_sub_f32", "v_cmp_gt_f32 (vcc)", "v_cmp_gt_u32 (sgpr pair dst)", "v_cndmask_b32 (sgpr pair)", "v_cmp_gt_u32 + v_cndmask (vcc) pairs", "v_cvt_f32_u32", "v_mul_hi_u32", "v_mov_b32 dpp quad_perm", "v_and_b32 dpp row_shr:1", "v_and_b32 sdwa (byte select)", "v_xnor_b32", "v_subrev_u32", "v_add_lshl_u32", "v_add_u32 + v_min_u32 alternating", "v_add_u32 + v_fma_f64 alternating", "v_fma_f32 + v_med3_u32 alternating"};

template <int KIND>
__global__ __launch_bounds__(64) void valu_kernel(int iters, uint32_t* out, unsigned long long* cyc) {
  uint32_t k[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  double d[8];
  float f[8];
  unsigned long long q[8];
#pragma unroll
  for (int j = 0; j < 8; j++)
    k[j] = threadIdx.x * 77u + j * 1000u, p[j] = f2{(float)j, 1.0f + threadIdx.x}, d[j] = j + 0.5 * threadIdx.x, f[j] = 0.5f * j + threadIdx.x,
    q[j] = (unsigned long long)threadIdx.x * 0x9E3779B97F4A7C15ull + j;
  uint32_t x = threadIdx.x * 2654435761u;
  const uint32_t sx = (uint32_t)iters * 3u;  // uniform: lives in an SGPR
  unsigned long long sm[2] = {0, 0};
  const unsigned long long smask = 0x5555555555555555ull * (unsigned long long)(iters & 3);  // uniform
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {  // 32 instructions per trip
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int a = (j + 1) & 7, b = (j + 2) & 7;
        if constexpr (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_SUB_U32) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_AND_B32) asm volatile("v_and_b32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_XOR_B32) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_AND_B32_SGPR) asm volatile("v_and_b32 %0, %1, %2" : "=v"(k[j]) : "s"(sx), "v"(k[b]));
        else if constexpr (KIND == K_LSHL_B32) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(k[j]) : "v"(k[a]));
        else if constexpr (KIND == K_FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[j]) : "v"(f[a]), "v"(f[b]));
        else if constexpr (KIND == K_MUL_F32) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f[j]) : "v"(f[a]), "v"(f[b]));
        else if constexpr (KIND == K_ADD_F32) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[j]) : "v"(f[a]), "v"(f[b]));
        else if constexpr (KIND == K_MIN_U32) asm volatile("v_min_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_OR3) asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[j]), "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_MAX3_U32) asm volatile("v_max3_u32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[j]), "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_AND_OR) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[j]), "v"(x), "v"(k[a]));
        else if constexpr (KIND == K_MED3) asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[a]), "v"(k[j]), "v"(x));
        else if constexpr (KIND == K_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %1, 3, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_BFE) asm volatile("v_bfe_u32 %0, %1, 3, 11" : "=v"(k[j]) : "v"(k[a]));
        else if constexpr (KIND == K_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]), "v"(x));
        else if constexpr (KIND == K_MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_FFBL) asm volatile("v_ffbl_b32 %0, %1" : "=v"(k[j]) : "v"(k[a]));
        else if constexpr (KIND == K_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(k[j]) : "v"(k[a]));
        else if constexpr (KIND == K_MIN_DPP) asm volatile("v_min_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_ADD_DPP) asm volatile("v_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]));
        else if constexpr (KIND == K_CMP_U32) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(k[a]), "v"(k[b]) : "vcc");
        else if constexpr (KIND == K_CMP_F64) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[a]), "v"(d[b]) : "vcc");
        else if constexpr (KIND == K_ADDC) asm volatile("v_addc_co_u32 %0, vcc, %1, %1, vcc" : "=v"(k[j]) : "v"(k[a]) : "vcc");
        else if constexpr (KIND == K_LSHL_B64) asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(q[j]) : "v"(q[a]));
        else if constexpr (KIND == K_ADD_F64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[j]) : "v"(d[a]), "v"(d[b]));
        else if constexpr (KIND == K_MUL_F64) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[j]) : "v"(d[a]), "v"(d[b]));
        else if constexpr (KIND == K_FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(d[a]), "v"(d[b]));
        else if constexpr (KIND == K_MIN_F64) asm volatile("v_min_f64 %0, %1, %2" : "=v"(d[j]) : "v"(d[a]), "v"(d[b]));
        else if constexpr (KIND == K_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(p[a]), "v"(p[b]));
        else if constexpr (KIND == K_PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[j]) : "v"(p[a]), "v"(p[b]));
        else if constexpr (KIND == K_PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[j]) : "v"(p[a]), "v"(p[b]));
        else if constexpr (KIND == K_OR_B32) { asm volatile("v_or_b32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MAX_U32) { asm volatile("v_max_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MIN_I32) { asm volatile("v_min_i32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MOV_B32) { asm volatile("v_mov_b32 %0, %1" : "=v"(k[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_NOT_B32) { asm volatile("v_not_b32 %0, %1" : "=v"(k[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_LSHR_B32) { asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(k[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_LSHLV_B32) { asm volatile("v_lshlrev_b32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MUL_U24) { asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MAD_U24) { asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]), "v"(k[j])); }
        else if constexpr (KIND == K_ADD3) { asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]), "v"(k[j])); }
        else if constexpr (KIND == K_LSHL_ADD) { asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_BFI) { asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]), "v"(k[j])); }
        else if constexpr (KIND == K_ALIGNBIT) { asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_AND_LIT) { asm volatile("v_and_b32 %0, 0x12345678, %1" : "=v"(k[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_AND_INL) { asm volatile("v_and_b32 %0, 15, %1" : "=v"(k[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_ADD_INL) { asm volatile("v_add_u32 %0, 15, %1" : "=v"(k[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_ADD_SGPR) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(k[j]) : "s"(sx), "v"(k[b])); }
        else if constexpr (KIND == K_ADD_CO) { asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]) : "vcc"); }
        else if constexpr (KIND == K_MAX_F32) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[j]) : "v"(f[a]), "v"(f[b])); }
        else if constexpr (KIND == K_MIN_F32) { asm volatile("v_min_f32 %0, %1, %2" : "=v"(f[j]) : "v"(f[a]), "v"(f[b])); }
        else if constexpr (KIND == K_MED3_F32) { asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(f[j]) : "v"(f[a]), "v"(f[j]), "v"(f[b])); }
        else if constexpr (KIND == K_MAX3_F32) { asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[j]) : "v"(f[a]), "v"(f[j]), "v"(f[b])); }
        else if constexpr (KIND == K_MIN3_F32) { asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(f[j]) : "v"(f[a]), "v"(f[j]), "v"(f[b])); }
        else if constexpr (KIND == K_FMAC_F32) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[j]) : "v"(f[a]), "v"(f[b])); }
        else if constexpr (KIND == K_SUB_F32) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[j]) : "v"(f[a]), "v"(f[b])); }
        else if constexpr (KIND == K_CMP_F32) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(f[a]), "v"(f[b]) : "vcc"); }
        else if constexpr (KIND == K_CMP_SG) { asm volatile("v_cmp_gt_u32 %0, %1, %2" : "=s"(sm[j & 1]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_CND_E64) { asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]), "s"(smask)); }
        else if constexpr (KIND == K_CND_PAIR) { asm volatile("v_cmp_gt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(k[j]) : "v"(k[a]), "v"(k[b]) : "vcc"); }
        else if constexpr (KIND == K_CVT_F32_U32) { asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_MUL_HI) { asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MOV_DPP_QP) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(k[j]) : "v"(k[a])); }
        else if constexpr (KIND == K_AND_DPP) { asm volatile("v_and_b32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_SDWA) { asm volatile("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_XNOR) { asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_SUBREV) { asm volatile("v_subrev_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_ADD_LSHL) { asm volatile("v_add_lshl_u32 %0, %1, %2, 3" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MIX_ADD_MIN) { if (j & 1) asm volatile("v_add_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); else asm volatile("v_min_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); }
        else if constexpr (KIND == K_MIX_ADD_F64) { if (j & 1) asm volatile("v_add_u32 %0, %1, %2" : "=v"(k[j]) : "v"(k[a]), "v"(k[b])); else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j]) : "v"(d[a]), "v"(d[b])); }
        else if constexpr (KIND == K_MIX_FMA_MED3) { if (j & 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[j]) : "v"(f[a]), "v"(f[b])); else asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(k[j]) : "v"(k[a]), "v"(k[j]), "v"(x)); }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  uint32_t r = x;
#pragma unroll
  for (int j = 0; j < 8; j++) r ^= k[j] ^ (uint32_t)p[j].x ^ (uint32_t)d[j] ^ (uint32_t)f[j] ^ (uint32_t)q[j];
  r ^= (uint32_t)sm[0] ^ (uint32_t)sm[1];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

// ---- row-local DPP forms (16-lane rows) against plain index arithmetic ------------------------------------
template <int CTRL, int ROWMASK = 0xF, int BANKMASK = 0xF, bool BC = true>
__device__ __forceinline__ int dpp(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWMASK, BANKMASK, BC);
}
__global__ void dpp_probe(int* out) {
  const int lane = threadIdx.x, l = lane & 15, rb = lane & 48;
  const int v = lane * 7 + 3;
  auto val = [](int ln) { return ln * 7 + 3; };
  int n = 0;
  out[n++ * 64 + lane] = dpp<0x150 + 5>(0, v) - val(rb + 5);     // row_newbcast:5
  out[n++ * 64 + lane] = dpp<0x150 + 15>(0, v) - val(rb + 15);   // row_newbcast:15
  out[n++ * 64 + lane] = dpp<0x140>(0, v) - val(rb + 15 - l);    // row_mirror
  out[n++ * 64 + lane] = dpp<0x141>(0, v) - val(rb + (l ^ 7));   // row_half_mirror
  out[n++ * 64 + lane] = dpp<0x1B>(0, v) - val(lane ^ 3);        // quad_perm [3,2,1,0]
  out[n++ * 64 + lane] = dpp<0x128>(0, v) - val(rb + (l ^ 8));   // row_ror:8
  out[n++ * 64 + lane] = dpp<0x111>(0, v) - (l >= 1 ? val(lane - 1) : 0);  // row_shr:1, zero fill
  out[n++ * 64 + lane] = dpp<0x101>(0, v) - (l <= 14 ? val(lane + 1) : 0);  // row_shl:1, zero fill
  {  // compare-exchange with lane ^ 4 in two bank-masked instructions (banks = groups of four lanes of a row)
    int d = -1;
    asm volatile("v_min_u32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n"
                 "v_max_u32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n" : "+v"(d) : "v"(v));
    const int o = val(lane ^ 4);
    out[n++ * 64 + lane] = d - ((l & 4) ? (v > o ? v : o) : (v < o ? v : o));
  }
  {  // ... with lane ^ 8 (row_ror:8), lanes 0-7 of a row keep the minimum
    int d = -1;
    asm volatile("v_min_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                 "v_max_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n" : "+v"(d) : "v"(v));
    const int o = val(lane ^ 8);
    out[n++ * 64 + lane] = d - ((l & 8) ? (v > o ? v : o) : (v < o ? v : o));
  }
  {  // ... with lane ^ 15 (row_mirror)
    int d = -1;
    asm volatile("v_min_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n"
                 "v_max_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc\n" : "+v"(d) : "v"(v));
    const int o = val(rb + 15 - l);
    out[n++ * 64 + lane] = d - ((l & 8) ? (v > o ? v : o) : (v < o ? v : o));
  }
  {  // ... with lane ^ 7 (row_half_mirror)
    int d = -1;
    asm volatile("v_min_u32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n"
                 "v_max_u32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n" : "+v"(d) : "v"(v));
    const int o = val(rb + (l ^ 7));
    out[n++ * 64 + lane] = d - ((l & 4) ? (v > o ? v : o) : (v < o ? v : o));
  }
  {  // row-local inclusive prefix sum: row_shr 1, 2, 4, 8
    int w = (lane * 2654435761u >> 27) & 7, ref = 0;
    for (int j = 0; j <= l; j++) ref += ((rb + j) * 2654435761u >> 27) & 7;
    int s = w;
    s += dpp<0x111>(0, s), s += dpp<0x112>(0, s), s += dpp<0x114>(0, s), s += dpp<0x118>(0, s);
    out[n++ * 64 + lane] = s - ref;
  }
}
constexpr int kDppChecks = 13;

int main() {
  uint32_t* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 4096 * 256 * sizeof(uint32_t)));
  CHECK(hipMalloc(&cyc, 8));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  unsigned long long* clk;
  CHECK(hipMalloc(&clk, 16));
  hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(1), 0, 0, clk);
  unsigned long long hclk[2];
  CHECK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)hclk[0] / (double)hclk[1] * 0.1;
  printf("shader clock (idle chip, one lane): %.3f GHz\n", ghz);
  {
    int* d;
    CHECK(hipMalloc(&d, kDppChecks * 64 * sizeof(int)));
    hipLaunchKernelGGL(dpp_probe, dim3(1), dim3(64), 0, 0, d);
    int h[kDppChecks * 64];
    CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const char* names[kDppChecks] = {"row_newbcast:5", "row_newbcast:15", "row_mirror", "row_half_mirror", "quad_perm [3,2,1,0]", "row_ror:8",
                                     "row_shr:1 zero fill", "row_shl:1 zero fill", "cmp-exchange ^4 (bank masks)", "cmp-exchange ^8", "cmp-exchange ^15",
                                     "cmp-exchange ^7", "row prefix sum"};
    printf("\n== row-local DPP forms ==\n");
    for (int k = 0; k < kDppChecks; k++) {
      int wrong = 0;
      for (int l = 0; l < 64; l++) wrong += h[k * 64 + l] != 0;
      printf("%-32s %s (%d lanes differ)\n", names[k], wrong ? "MISMATCH" : "ok", wrong);
    }
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto wall_ms = [&](auto&& launch) {
    launch();  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms;
  };
  printf("\n== vector issue rate: cycles per wave-instruction per SIMD at the idle clock (wall time of a launch that fills every SIMD);\n"
         "   in brackets: s_memtime cycles per instruction of wavefront 0 itself (it shares its SIMD with w - 1 others) ==\n");
  const int vit = 8192;
  for (int kind = 0; kind < K_COUNT; kind++) {
    printf("%-38s", kind_name[kind]);
    for (int wps : {1, 2, 4, 8}) {
      double ms = 0;
      auto go = [&](auto kern) { ms = wall_ms([&] { hipLaunchKernelGGL(kern, dim3(cus * 4 * wps), dim3(64), 0, 0, vit, out, cyc); }); };
      switch (kind) {
#define CASE(K) case K: go(valu_kernel<K>); break;
        CASE(K_ADD_U32) CASE(K_AND_B32) CASE(K_LSHL_B32) CASE(K_FMA_F32) CASE(K_MIN_U32) CASE(K_CNDMASK) CASE(K_OR3) CASE(K_AND_OR) CASE(K_MED3)
        CASE(K_LSHL_OR) CASE(K_BFE) CASE(K_PERM) CASE(K_MUL_LO) CASE(K_FFBL) CASE(K_BCNT) CASE(K_MOV_DPP) CASE(K_MIN_DPP) CASE(K_ADD_DPP)
        CASE(K_CMP_U32) CASE(K_CMP_F64) CASE(K_ADDC) CASE(K_LSHL_B64) CASE(K_ADD_F64) CASE(K_MUL_F64) CASE(K_FMA_F64) CASE(K_MIN_F64)
        CASE(K_PK_FMA_F32) CASE(K_PK_MUL_F32) CASE(K_PK_ADD_F32) CASE(K_MUL_F32) CASE(K_ADD_F32) CASE(K_AND_B32_SGPR) CASE(K_XOR_B32)
        CASE(K_SUB_U32) CASE(K_MAX3_U32)
        CASE(K_OR_B32) CASE(K_MAX_U32) CASE(K_MIN_I32) CASE(K_MOV_B32) CASE(K_NOT_B32) CASE(K_LSHR_B32) CASE(K_LSHLV_B32) CASE(K_MUL_U24) CASE(K_MAD_U24) CASE(K_ADD3) CASE(K_LSHL_ADD) CASE(K_BFI) CASE(K_ALIGNBIT) CASE(K_AND_LIT) CASE(K_AND_INL) CASE(K_ADD_INL) CASE(K_ADD_SGPR) CASE(K_ADD_CO) CASE(K_MAX_F32) CASE(K_MIN_F32) CASE(K_MED3_F32) CASE(K_MAX3_F32) CASE(K_MIN3_F32) CASE(K_FMAC_F32) CASE(K_SUB_F32) CASE(K_CMP_F32) CASE(K_CMP_SG) CASE(K_CND_E64) CASE(K_CND_PAIR) CASE(K_CVT_F32_U32) CASE(K_MUL_HI) CASE(K_MOV_DPP_QP) CASE(K_AND_DPP) CASE(K_SDWA) CASE(K_XNOR) CASE(K_SUBREV) CASE(K_ADD_LSHL) CASE(K_MIX_ADD_MIN) CASE(K_MIX_ADD_F64) CASE(K_MIX_FMA_MED3)
#undef CASE
      }
      unsigned long long c = 0;
      CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      // every SIMD ran wps waves of vit * 32 vector instructions each (+ the loop's scalar bookkeeping)
      printf(" | %d w: %5.2f [%5.2f]", wps, ms * 1e-3 * ghz * 1e9 / (vit * 32.0) / wps, (double)c / (vit * 32.0));
    }
    printf("\n");
  }
  return 0;
}
