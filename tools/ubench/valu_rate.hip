// valu_rate.hip — issue rate of the VALU instructions the mix kernel leans on (gfx950).
// Each kernel runs 8 independent chains of one instruction; every SIMD holds `waves` wavefronts.
// Prints cycles per wave-instruction per SIMD assuming the measured shader clock.
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAIN8(INSTR)                                   \
  asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : "vcc", "s4", "s5", "s6");

#define KERNEL32(NAME, INSTR)                                                           \
  __global__ void NAME(uint32_t* out, int iters) {                                      \
    uint32_t a[8]; uint32_t b = threadIdx.x | 0x3f800000u, c = 0x3f000001u;             \
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 17u + i + 0x3f800000u;             \
    for (int it = 0; it < iters; it++) { CHAIN8(INSTR) CHAIN8(INSTR) CHAIN8(INSTR) CHAIN8(INSTR) } \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= a[i];                              \
    if (s == 0x12345u) out[0] = s;                                                      \
  }
#define KERNEL64(NAME, INSTR)                                                           \
  __global__ void NAME(uint32_t* out, int iters) {                                      \
    double a[8]; double b = 1.0 + threadIdx.x * 1e-9, c = 0.999;                        \
    for (int i = 0; i < 8; i++) a[i] = 1.0 + threadIdx.x * 1e-3 + i;                    \
    for (int it = 0; it < iters; it++) { CHAIN8(INSTR) CHAIN8(INSTR) CHAIN8(INSTR) CHAIN8(INSTR) } \
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];                                \
    if (s == 0.12345) out[0] = 1;                                                       \
  }
// 64-bit source, 32-bit destination (conversions)
#define KERNEL6432(NAME, INSTR)                                                         \
  __global__ void NAME(uint32_t* out, int iters) {                                      \
    uint32_t a[8]; double b = 1.0 + threadIdx.x * 1e-3, c = 0.999;                      \
    for (int i = 0; i < 8; i++) a[i] = i;                                               \
    for (int it = 0; it < iters; it++) { CHAIN8(INSTR) CHAIN8(INSTR) CHAIN8(INSTR) CHAIN8(INSTR) } \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= a[i];                              \
    if (s == 0x12345u) out[0] = s;                                                      \
  }

#define I_ADD_F32(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_PK_MUL_F32(k) "v_pk_mul_f32 %" #k ", %" #k ", %9\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_MAX_DPP(k) "v_max_u32_dpp %" #k ", %" #k ", %" #k " row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", |%" #k "|, |%8|, %8\n"
#define I_CVT_F32_I32(k) "v_cvt_f32_i32 %" #k ", %" #k "\n"
#define I_ADD_F64(k) "v_add_f64 %" #k ", %" #k ", %8\n"
#define I_MUL_F64(k) "v_mul_f64 %" #k ", %" #k ", %9\n"
#define I_FMA_F64(k) "v_fma_f64 %" #k ", %" #k ", %9, %8\n"
#define I_FRACT_F64(k) "v_fract_f64 %" #k ", %" #k "\n"
#define I_FLOOR_F64(k) "v_floor_f64 %" #k ", %" #k "\n"
#define I_CVT_I32_F64(k) "v_cvt_i32_f64 %" #k ", %8\n"
#define I_CVT_F32_F64(k) "v_cvt_f32_f64 %" #k ", %8\n"
#define I_LSHL_ADD_U64(k) "v_lshl_add_u64 %" #k ", %" #k ", 2, %8\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_SUB_U32(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define I_CNDMASK_S(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[4:5]\n"
#define I_CNDMASK_NIP(k) "v_cndmask_b32 %" #k ", %8, %9, vcc\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %8, %" #k ", %9\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_ASHR(k) "v_ashrrev_i32 %" #k ", 31, %" #k "\n"
#define I_CMP(k) "v_cmp_lt_i32 vcc, %" #k ", %8\n"
#define I_CMP_S(k) "v_cmp_lt_i32_e64 s[4:5], %" #k ", %8\n"
#define I_AND_OR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_READLANE(k) "v_readlane_b32 s6, %" #k ", 63\n"
#define I_WRITELANE(k) "v_writelane_b32 %" #k ", s6, 5\n"
#define I_SWAP32(k) "v_permlane32_swap_b32 %" #k ", %8\n"
#define I_MAD_U24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define I_FMA_F32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_PK_FMA_F32(k) "v_pk_fma_f32 %" #k ", %" #k ", %9, %8\n"
#define I_CVT_SDWA(k) "v_cvt_f32_i32_sdwa %" #k ", sext(%" #k ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"

KERNEL32(k_add_f32, I_ADD_F32)
KERNEL64(k_pk_mul_f32, I_PK_MUL_F32)
KERNEL32(k_cndmask, I_CNDMASK)
KERNEL32(k_max_dpp, I_MAX_DPP)
KERNEL32(k_max3, I_MAX3)
KERNEL32(k_cvt_f32_i32, I_CVT_F32_I32)
KERNEL32(k_mov, I_MOV)
KERNEL32(k_sub_u32, I_SUB_U32)
KERNEL32(k_cndmask_s, I_CNDMASK_S)
KERNEL32(k_cndmask_nip, I_CNDMASK_NIP)
KERNEL32(k_bfi, I_BFI)
KERNEL32(k_perm, I_PERM)
KERNEL32(k_ashr, I_ASHR)
KERNEL32(k_cmp, I_CMP)
KERNEL32(k_cmp_s, I_CMP_S)
KERNEL32(k_and_or, I_AND_OR)
KERNEL32(k_med3, I_MED3)
KERNEL32(k_readlane, I_READLANE)
KERNEL32(k_writelane, I_WRITELANE)
KERNEL32(k_swap32, I_SWAP32)
KERNEL32(k_mad_u24, I_MAD_U24)
KERNEL32(k_fma_f32, I_FMA_F32)
KERNEL64(k_pk_fma_f32, I_PK_FMA_F32)
KERNEL32(k_cvt_sdwa, I_CVT_SDWA)
KERNEL64(k_add_f64, I_ADD_F64)
KERNEL64(k_mul_f64, I_MUL_F64)
KERNEL64(k_fma_f64, I_FMA_F64)
KERNEL64(k_fract_f64, I_FRACT_F64)
KERNEL64(k_floor_f64, I_FLOOR_F64)
KERNEL64(k_lshl_add_u64, I_LSHL_ADD_U64)
KERNEL6432(k_cvt_i32_f64, I_CVT_I32_F64)
KERNEL6432(k_cvt_f32_f64, I_CVT_F32_F64)

typedef void (*kern_t)(uint32_t*, int);

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double clk = p.clockRate * 1e3;   // Hz (peak shader clock)
  uint32_t* out; hipMalloc(&out, 64);
  struct { const char* name; kern_t k; } ks[] = {
    {"v_add_f32", k_add_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_cndmask_b32", k_cndmask}, {"v_max_u32_dpp", k_max_dpp},
    {"v_max3_f32", k_max3}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_mov_b32", k_mov}, {"v_sub_u32", k_sub_u32},
    {"v_cndmask e64 sgpr", k_cndmask_s}, {"v_cndmask notinplace", k_cndmask_nip}, {"v_bfi_b32", k_bfi}, {"v_perm_b32", k_perm},
    {"v_ashrrev_i32", k_ashr}, {"v_cmp_lt_i32 vcc", k_cmp}, {"v_cmp_lt_i32 sgpr", k_cmp_s}, {"v_and_or_b32", k_and_or},
    {"v_med3_f32", k_med3}, {"v_readlane_b32", k_readlane}, {"v_writelane_b32", k_writelane}, {"v_permlane32_swap", k_swap32},
    {"v_mad_u32_u24", k_mad_u24}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_cvt_f32_i32_sdwa", k_cvt_sdwa},
    {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_fract_f64", k_fract_f64},
    {"v_floor_f64", k_floor_f64}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f32_f64", k_cvt_f32_f64}};
  printf("device %s, %d CUs, clock %.0f MHz\n", p.name, cus, clk / 1e6);
  printf("%-22s %12s %12s\n", "instr", "cyc/instr@1w", "cyc/instr@4w");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& k : ks) {
    double res[2];
    int wi = 0;
    for (int waves : {1, 4}) {
      const int iters = 20000;
      // block = 256 threads = 4 waves = one per SIMD; `waves` blocks per CU
      dim3 grid(cus * waves), block(256);
      hipLaunchKernelGGL(k.k, grid, block, 0, 0, out, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k.k, grid, block, 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = (double)iters * 32 * waves;
      res[wi++] = ms * 1e-3 * clk / instr_per_simd;
    }
    printf("%-22s %12.2f %12.2f\n", k.name, res[0], res[1]);
  }
  return 0;
}
