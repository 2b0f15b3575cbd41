// Issue cost of the VALU instructions the kernels are built from: whole GPU (8192 workgroups x 256 threads, 8 independent
// chains per lane), hipEvent timing -> cycles per wave-instruction per SIMD at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 1024
#define OPS2(I) I " %0, %0, %8\n" I " %1, %1, %8\n" I " %2, %2, %8\n" I " %3, %3, %8\n" I " %4, %4, %8\n" I " %5, %5, %8\n" I " %6, %6, %8\n" I " %7, %7, %8\n"
#define OPS3(I) I " %0, %0, %8, %8\n" I " %1, %1, %8, %8\n" I " %2, %2, %8, %8\n" I " %3, %3, %8, %8\n" I " %4, %4, %8, %8\n" I " %5, %5, %8, %8\n" I " %6, %6, %8, %8\n" I " %7, %7, %8, %8\n"
#define OPS1(I) I " %0, %0\n" I " %1, %1\n" I " %2, %2\n" I " %3, %3\n" I " %4, %4\n" I " %5, %5\n" I " %6, %6\n" I " %7, %7\n"
#define OPSQ(I) I " %0, %0, %8, %0\n" I " %1, %1, %8, %1\n" I " %2, %2, %8, %2\n" I " %3, %3, %8, %3\n" I " %4, %4, %8, %4\n" I " %5, %5, %8, %5\n" I " %6, %6, %8, %6\n" I " %7, %7, %8, %7\n"

#define OPSC(I) I " %0, %0, %8, vcc\n" I " %1, %1, %8, vcc\n" I " %2, %2, %8, vcc\n" I " %3, %3, %8, vcc\n" I " %4, %4, %8, vcc\n" I " %5, %5, %8, vcc\n" I " %6, %6, %8, vcc\n" I " %7, %7, %8, vcc\n"
#define OPSS(I) I " %0, %0, %8, s[20:21]\n" I " %1, %1, %8, s[20:21]\n" I " %2, %2, %8, s[20:21]\n" I " %3, %3, %8, s[20:21]\n" I " %4, %4, %8, s[20:21]\n" I " %5, %5, %8, s[20:21]\n" I " %6, %6, %8, s[20:21]\n" I " %7, %7, %8, s[20:21]\n"
#define OPSV(I) I " vcc, %0, %8\n" I " vcc, %1, %8\n" I " vcc, %2, %8\n" I " vcc, %3, %8\n" I " vcc, %4, %8\n" I " vcc, %5, %8\n" I " vcc, %6, %8\n" I " vcc, %7, %8\n"
#define OPSW(I) I " s[20:21], %0, %8\n" I " s[22:23], %1, %8\n" I " s[24:25], %2, %8\n" I " s[26:27], %3, %8\n" I " s[20:21], %4, %8\n" I " s[22:23], %5, %8\n" I " s[24:25], %6, %8\n" I " s[26:27], %7, %8\n"
#define OPSP(I) "v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
#define OPSD(I) I " %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" I " %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" I " %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n" I " %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" I " %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n" I " %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" I " %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n" I " %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OPSR(I) I " s20, %0, 3\n" I " s21, %1, 3\n" I " s22, %2, 3\n" I " s23, %3, 3\n" I " s24, %4, 3\n" I " s25, %5, 3\n" I " s26, %6, 3\n" I " s27, %7, 3\n"
#define CLOB : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"
#define KERNEL(NAME, T, OPS)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t *sink) {                                                    \
        T a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; uint32_t c = 11585; \
        for (int i = 0; i < REP; i++)                                                                                \
            asm volatile(OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) CLOB);  \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) sink[threadIdx.x] = 1;                               \
    }
#define K2(N, I) KERNEL(N, uint32_t, OPS2(I))
#define K3(N, I) KERNEL(N, uint32_t, OPS3(I))
K2(k0, "v_add_u32") K2(k1, "v_mul_lo_u32") K2(k2, "v_mul_i32_i24") K2(k3, "v_pk_mul_lo_u16") K2(k4, "v_pk_add_u16")
K2(k5, "v_pk_ashrrev_i16") K2(k6, "v_lshlrev_b32") K2(k7, "v_min_u32") K2(k8, "v_xor_b32") K2(k9, "v_cndmask_b32")
K3(k10, "v_perm_b32") K3(k11, "v_alignbyte_b32") K3(k12, "v_sad_u8") K3(k13, "v_mad_i32_i24") K3(k14, "v_pk_mad_i16")
K3(k15, "v_lerp_u8") K3(k16, "v_dot4_i32_i8") K3(k17, "v_add3_u32") K3(k18, "v_lshl_add_u32") K3(k19, "v_med3_i32")
K3(k20, "v_bfe_u32") K3(k21, "v_mad_u32_u24") K3(k22, "v_lshl_or_b32") K3(k23, "v_msad_u8") K3(k24, "v_and_or_b32")
KERNEL(k25, uint32_t, OPS1("v_sat_pk_u8_i16")) KERNEL(k26, uint32_t, OPS1("v_mov_b32"))
KERNEL(k27, uint64_t, OPSQ("v_qsad_pk_u16_u8"))
KERNEL(k40, uint32_t, OPSC("v_cndmask_b32")) KERNEL(k41, uint32_t, OPSS("v_cndmask_b32")) KERNEL(k42, uint32_t, OPSV("v_cmp_lt_u32"))
KERNEL(k43, uint32_t, OPSW("v_cmp_lt_u32")) KERNEL(k44, uint32_t, OPSP("")) KERNEL(k45, uint32_t, OPSD("v_mov_b32_dpp")) KERNEL(k46, uint32_t, OPSR("v_readlane_b32"))
K2(k47, "v_and_b32") K2(k48, "v_or_b32") K2(k49, "v_lshrrev_b32") K2(k50, "v_max_u32") K2(k51, "v_min_i32") K3(k52, "v_bfi_b32") K2(k53, "v_subrev_u32") K2(k54, "v_mul_u32_u24") K3(k55, "v_xad_u32") K3(k56, "v_or3_b32")
K3(k29, "v_dot4_u32_u8") K3(k30, "v_max3_i32") K2(k31, "v_max_i32") K2(k32, "v_ashrrev_i32") K2(k33, "v_sub_u32")
int main() {
    uint32_t *s; (void)hipMalloc(&s, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void (*ks[])(uint32_t *) = {k0, k1, k2, k3, k4, k5, k6, k7, k8, k9, k10, k11, k12, k13, k14, k15, k16, k17, k18, k19, k20, k21, k22, k23, k24, k25, k26, k27, k29, k30, k31, k32, k33, k40, k41, k42, k43, k44, k45, k46, k47, k48, k49, k50, k51, k52, k53, k54, k55, k56};
    const char *n[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_i32_i24", "v_pk_mul_lo_u16", "v_pk_add_u16", "v_pk_ashrrev_i16", "v_lshlrev_b32", "v_min_u32", "v_xor_b32", "v_cndmask_b32",
                       "v_perm_b32", "v_alignbyte_b32", "v_sad_u8", "v_mad_i32_i24", "v_pk_mad_i16", "v_lerp_u8", "v_dot4_i32_i8", "v_add3_u32", "v_lshl_add_u32", "v_med3_i32",
                       "v_bfe_u32", "v_mad_u32_u24", "v_lshl_or_b32", "v_msad_u8", "v_and_or_b32", "v_sat_pk_u8_i16", "v_mov_b32", "v_qsad_pk_u16_u8", "v_dot4_u32_u8", "v_max3_i32", "v_max_i32", "v_ashrrev_i32", "v_sub_u32", "v_cndmask(vcc)", "v_cndmask(sgpr)", "v_cmp->vcc", "v_cmp->sgpr", "cmp+cndmask pair/2", "v_mov_dpp", "v_readlane", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_max_u32", "v_min_i32", "v_bfi_b32", "v_subrev_u32", "v_mul_u32_u24", "v_xad_u32", "v_or3_b32"};
    const int WG = 8192, NK = sizeof ks / sizeof ks[0];
    for (int i = 0; i < NK; i++) {
        ks[i]<<<WG, 256>>>(s); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); ks[i]<<<WG, 256>>>(s); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double insts = (double)WG * 4 * REP * 8; // wave instructions
        printf("%-18s %.2f\n", n[i], ms * 1e-3 * 2.4e9 * 1024 / insts);
    }
    return 0;
}
