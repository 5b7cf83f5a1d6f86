// VALU issue-rate microbenchmark for gfx950: how many wave64 instructions per cycle per
// SIMD do the candidate bin-match instructions sustain?  (Calibration for DESIGN.md.)
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(X) X X X X X X X X
#define BODY(INSTR)                                                             \
    for (int it = 0; it < iters; ++it) {                                        \
        REP8(asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
                          : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), \
                            "+v"(m[6]), "+v"(m[7])                              \
                          : "s"(sa), "v"(vb), "v"(vc));)                        \
    }

#define I_XOR_S(i) "v_xor_b32 %" #i ", %8, %" #i "\n"
#define I_XOR_V(i) "v_xor_b32 %" #i ", %9, %" #i "\n"
#define I_OR3(i) "v_or3_b32 %" #i ", %" #i ", %9, %10\n"
#define I_BITOP_S(i) "v_bitop3_b32 %" #i ", %8, %" #i ", %9 bitop3:0xde\n"
#define I_BITOP_V(i) "v_bitop3_b32 %" #i ", %10, %" #i ", %9 bitop3:0xde\n"
#define I_BCNT(i) "v_bcnt_u32_b32 %" #i ", %9, %" #i "\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %9, %10, %" #i "\n"
#define I_XAD(i) "v_xad_u32 %" #i ", %9, %10, %" #i "\n"
#define I_ADD(i) "v_add_u32 %" #i ", %9, %" #i "\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %9, %10\n"

#define KERNEL(NAME, INSTR)                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t sa_in, int iters) \
    {                                                                           \
        uint32_t m[8];                                                          \
        for (int i = 0; i < 8; ++i) m[i] = threadIdx.x * 2654435761u + i;       \
        uint32_t sa = __builtin_amdgcn_readfirstlane(sa_in);                    \
        uint32_t vb = threadIdx.x ^ 0x5bd1e995u, vc = threadIdx.x * 7u;         \
        BODY(INSTR)                                                             \
        uint32_t r = 0;                                                         \
        for (int i = 0; i < 8; ++i) r ^= m[i];                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                         \
    }

KERNEL(k_xor_s, I_XOR_S)
KERNEL(k_xor_v, I_XOR_V)
KERNEL(k_or3, I_OR3)
KERNEL(k_bitop_s, I_BITOP_S)
KERNEL(k_bitop_v, I_BITOP_V)
KERNEL(k_bcnt, I_BCNT)
KERNEL(k_andor, I_ANDOR)
KERNEL(k_xad, I_XAD)
KERNEL(k_add, I_ADD)
KERNEL(k_add3, I_ADD3)

typedef void (*kern_t)(uint32_t *, uint32_t, int);

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk_ghz = prop.clockRate / 1e6;
    printf("device %s CUs %d clock %.2f GHz\n", prop.gcnArchName, cus, clk_ghz);
    uint32_t *out;
    const int waves_per_simd[] = {1, 2, 4};
    hipMalloc(&out, (size_t)cus * 16 * 256 * 4 * sizeof(uint32_t));
    struct { const char *name; kern_t k; } ks[] = {
        {"v_xor_b32 v,s,v", k_xor_s}, {"v_xor_b32 v,v,v", k_xor_v}, {"v_or3_b32", k_or3},
        {"v_bitop3_b32 v,s,v,v", k_bitop_s}, {"v_bitop3_b32 v,v,v,v", k_bitop_v},
        {"v_bcnt_u32_b32", k_bcnt}, {"v_and_or_b32", k_andor}, {"v_xad_u32", k_xad},
        {"v_add_u32", k_add}, {"v_add3_u32", k_add3}};
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &kk : ks) {
        for (int w : waves_per_simd) {
            // blocks of 256 threads = 4 waves = 1 wave per SIMD; w blocks per CU
            const int blocks = cus * w;
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, out, 123u, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, out, 123u, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_wave = (double)iters * 64.0;
            const double wave_instr = instr_per_wave * blocks * 4;
            const double per_simd_per_s = wave_instr / (cus * 4) / (ms * 1e-3);
            printf("%-24s waves/SIMD %d: %.3f ms  %.3f wave-instr/ns/SIMD  (%.2f clk/instr/SIMD at %.2f GHz)\n",
                   kk.name, w, ms, per_simd_per_s * 1e-9, clk_ghz * 1e9 / per_simd_per_s, clk_ghz);
        }
    }
    return 0;
}
