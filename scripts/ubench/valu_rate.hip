// Micro-benchmark: issue rate of the integer VALU / LDS-crossbar ops the scan kernels are built from (gfx950).
// Prints cycles per wave-instruction per SIMD, assuming every SIMD is saturated (8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define OPS8(STR)                                                                                         \
    asm volatile(STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR "\n" STR                        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(sc));

#define DEF_KERNEL(NAME, I0, I1, I2, I3)                                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned b_in, unsigned sc, int iters) {   \
        unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, b = b_in + threadIdx.x;   \
        for (int i = 0; i < iters; i++) {                                                                  \
            asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I0 "\n" I1 "\n" I2 "\n" I3 "\n"                    \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I0 "\n" I1 "\n" I2 "\n" I3 "\n"                    \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I0 "\n" I1 "\n" I2 "\n" I3 "\n"                    \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I0 "\n" I1 "\n" I2 "\n" I3                         \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(sc));                       \
        }                                                                                                  \
        if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678u) out[0] = a0;                                               \
    }

DEF_KERNEL(k_add, "v_add_u32 %0, %0, %4", "v_add_u32 %1, %1, %4", "v_add_u32 %2, %2, %4", "v_add_u32 %3, %3, %4")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %0, %4", "v_xor_b32 %1, %1, %4", "v_xor_b32 %2, %2, %4", "v_xor_b32 %3, %3, %4")
DEF_KERNEL(k_min, "v_min_u32 %0, %0, %4", "v_min_u32 %1, %1, %4", "v_min_u32 %2, %2, %4", "v_min_u32 %3, %3, %4")
DEF_KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0", "v_lshlrev_b32 %1, 1, %1", "v_lshlrev_b32 %2, 1, %2", "v_lshlrev_b32 %3, 1, %3")
DEF_KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %4", "v_mul_u32_u24 %1, %1, %4", "v_mul_u32_u24 %2, %2, %4", "v_mul_u32_u24 %3, %3, %4")
DEF_KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %4, %0", "v_mad_u32_u24 %1, %1, %4, %1", "v_mad_u32_u24 %2, %2, %4, %2", "v_mad_u32_u24 %3, %3, %4, %3")
DEF_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %4, 31", "v_alignbit_b32 %1, %1, %4, 31", "v_alignbit_b32 %2, %2, %4, 31", "v_alignbit_b32 %3, %3, %4, 31")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 9", "v_bfe_u32 %1, %1, 3, 9", "v_bfe_u32 %2, %2, 3, 9", "v_bfe_u32 %3, %3, 3, 9")
DEF_KERNEL(k_min3, "v_min3_i32 %0, %0, %4, %1", "v_min3_i32 %1, %1, %4, %2", "v_min3_i32 %2, %2, %4, %3", "v_min3_i32 %3, %3, %4, %0")
DEF_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %4", "v_lshl_add_u32 %1, %1, 2, %4", "v_lshl_add_u32 %2, %2, 2, %4", "v_lshl_add_u32 %3, %3, 2, %4")
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %4, %1", "v_add3_u32 %1, %1, %4, %2", "v_add3_u32 %2, %2, %4, %3", "v_add3_u32 %3, %3, %4, %0")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %4, %1", "v_perm_b32 %1, %1, %4, %2", "v_perm_b32 %2, %2, %4, %3", "v_perm_b32 %3, %3, %4, %0")
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %4, %1 bitop3:0x6c", "v_bitop3_b32 %1, %1, %4, %2 bitop3:0x6c", "v_bitop3_b32 %2, %2, %4, %3 bitop3:0x6c", "v_bitop3_b32 %3, %3, %4, %0 bitop3:0x6c")
DEF_KERNEL(k_pk_add, "v_pk_add_u16 %0, %0, %4", "v_pk_add_u16 %1, %1, %4", "v_pk_add_u16 %2, %2, %4", "v_pk_add_u16 %3, %3, %4")
DEF_KERNEL(k_pk_min, "v_pk_min_u16 %0, %0, %4", "v_pk_min_u16 %1, %1, %4", "v_pk_min_u16 %2, %2, %4", "v_pk_min_u16 %3, %3, %4")
DEF_KERNEL(k_pk_mul, "v_pk_mul_lo_u16 %0, %0, %4", "v_pk_mul_lo_u16 %1, %1, %4", "v_pk_mul_lo_u16 %2, %2, %4", "v_pk_mul_lo_u16 %3, %3, %4")
DEF_KERNEL(k_pk_lshl, "v_pk_lshlrev_b16 %0, 2, %0", "v_pk_lshlrev_b16 %1, 2, %1", "v_pk_lshlrev_b16 %2, 2, %2", "v_pk_lshlrev_b16 %3, 2, %3")
DEF_KERNEL(k_pk_mad, "v_pk_mad_u16 %0, %0, %4, %0", "v_pk_mad_u16 %1, %1, %4, %1", "v_pk_mad_u16 %2, %2, %4, %2", "v_pk_mad_u16 %3, %3, %4, %3")
DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %4, %0", "v_fma_f32 %1, %1, %4, %1", "v_fma_f32 %2, %2, %4, %2", "v_fma_f32 %3, %3, %4, %3")
DEF_KERNEL(k_sub_sdwa, "v_sub_u32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_sub_u32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD", "v_sub_u32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD", "v_sub_u32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
DEF_KERNEL(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_1", "v_xor_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2", "v_xor_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_3", "v_xor_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_0")
DEF_KERNEL(k_lshl_sdwa, "v_lshlrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_lshlrev_b32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD", "v_lshlrev_b32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD", "v_lshlrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
DEF_KERNEL(k_mul24_sdwa, "v_mul_u32_u24_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_mul_u32_u24_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD", "v_mul_u32_u24_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD", "v_mul_u32_u24_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")

struct Entry { const char* name; void (*fn)(unsigned*, unsigned, unsigned, int); };

int main() {
    unsigned* out;
    hipMalloc(&out, 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    std::vector<Entry> es = {
        {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_min_u32", k_min}, {"v_lshlrev_b32", k_lshl},
        {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_alignbit_b32", k_alignbit}, {"v_bfe_u32", k_bfe},
        {"v_min3_i32", k_min3}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_perm_b32", k_perm},
        {"v_bitop3_b32", k_bitop3}, {"v_pk_add_u16", k_pk_add}, {"v_pk_min_u16", k_pk_min}, {"v_pk_mul_lo_u16", k_pk_mul},
        {"v_pk_lshlrev_b16", k_pk_lshl}, {"v_pk_mad_u16", k_pk_mad}, {"v_fma_f32", k_fma},
        {"v_sub_u32_sdwa", k_sub_sdwa}, {"v_xor_b32_sdwa", k_xor_sdwa}, {"v_lshlrev_b32_sdwa", k_lshl_sdwa},
        {"v_mul_u32_u24_sdwa", k_mul24_sdwa},
    };
    const int iters = 2000, blocks = cus * 8;  // 8 blocks x 4 waves = 8 waves per SIMD
    printf("device: %s, %d CUs, clockRate %.2f GHz\n", prop.name, cus, ghz);
    for (auto& e : es) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        e.fn<<<blocks, 256>>>(out, 3, 5, 10);
        hipDeviceSynchronize();
        hipEventRecord(a);
        e.fn<<<blocks, 256>>>(out, 3, 5, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double insts_per_simd = double(iters) * 32 * 8;  // 32 instrs per iteration per wave, 8 waves per SIMD
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("%-22s %8.3f ms  %.2f cycles/wave-instr/SIMD (at %.2f GHz nominal)\n", e.name, ms, cyc / insts_per_simd, ghz);
    }
    return 0;
}
