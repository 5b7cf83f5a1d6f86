// Does the VGPR bank of the source operands change the issue rate of the bin-match
// instructions on gfx950?  Fixed physical registers via inline asm.  Also reports the
// shader clock under load (s_memtime vs s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

// accumulators v0,v4,...,v28 (bank 0 if bank = index % 4); sources per case
#define ACC8(OP, SA, SB)                                                        \
    OP " v0, " SA ", v0, " SB "\n" OP " v4, " SA ", v4, " SB "\n"              \
    OP " v8, " SA ", v8, " SB "\n" OP " v12, " SA ", v12, " SB "\n"            \
    OP " v16, " SA ", v16, " SB "\n" OP " v20, " SA ", v20, " SB "\n"          \
    OP " v24, " SA ", v24, " SB "\n" OP " v28, " SA ", v28, " SB "\n"
#define ACC8_B(OP, SA, SB) ACC8(OP, SA, SB " bitop3:0xde")

#define CLOB "v1","v2","v5","v6","v0","v4","v8","v12","v16","v20","v24","v28","v32","v33","v34","v35","v36","v37","v38","v39","v40"

#define KERNEL(NAME, BODY)                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint64_t *clk)  \
    {                                                                            \
        asm volatile("v_mov_b32 v0, 1\n v_mov_b32 v4, 2\n v_mov_b32 v8, 3\n v_mov_b32 v12, 4\n" \
                     "v_mov_b32 v16, 5\n v_mov_b32 v20, 6\n v_mov_b32 v24, 7\n v_mov_b32 v28, 8\n" \
                     "v_mov_b32 v32, 9\n v_mov_b32 v33, 10\n v_mov_b32 v34, 11\n v_mov_b32 v35, 12\n" \
                     "v_mov_b32 v36, 13\n v_mov_b32 v37, 14\n v_mov_b32 v38, 15\n v_mov_b32 v40, 16\n v_mov_b32 v39, 3\n" \
                     "v_mov_b32 v1, 1\n v_mov_b32 v2, 2\n v_mov_b32 v5, 5\n v_mov_b32 v6, 6\n" ::: CLOB); \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
        for (int it = 0; it < iters; ++it) {                                     \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY ::: CLOB);      \
        }                                                                        \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
        uint32_t r;                                                              \
        asm volatile("v_xor_b32 %0, v0, v4\n v_xor_b32 %0, %0, v8" : "=v"(r) :: CLOB); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
    }

KERNEL(bitop_diff, ACC8_B("v_bitop3_b32", "v33", "v34"))     // banks acc0, a1, b2
KERNEL(bitop_a_same, ACC8_B("v_bitop3_b32", "v32", "v34"))   // a in acc's bank
KERNEL(bitop_ab_same, ACC8_B("v_bitop3_b32", "v33", "v37"))  // a and b share a bank (1), acc differs
KERNEL(bitop_all_same, ACC8_B("v_bitop3_b32", "v32", "v36")) // all three in bank 0
KERNEL(bitop_sgpr, ACC8_B("v_bitop3_b32", "s4", "v34"))      // SGPR source
KERNEL(bitop_b_same, ACC8_B("v_bitop3_b32", "v33", "v36"))   // b (src2) in acc's (src1) bank
// accumulator as src2 / src0 instead of src1
#define ACC8_SRC2(SA, SB)                                                       \
    "v_bitop3_b32 v0, " SA ", " SB ", v0 bitop3:0xf6\n v_bitop3_b32 v4, " SA ", " SB ", v4 bitop3:0xf6\n"   \
    "v_bitop3_b32 v8, " SA ", " SB ", v8 bitop3:0xf6\n v_bitop3_b32 v12, " SA ", " SB ", v12 bitop3:0xf6\n" \
    "v_bitop3_b32 v16, " SA ", " SB ", v16 bitop3:0xf6\n v_bitop3_b32 v20, " SA ", " SB ", v20 bitop3:0xf6\n" \
    "v_bitop3_b32 v24, " SA ", " SB ", v24 bitop3:0xf6\n v_bitop3_b32 v28, " SA ", " SB ", v28 bitop3:0xf6\n"
KERNEL(bitop_acc2_ab_same, ACC8_SRC2("v33", "v37"))   // src0,src1 same bank(1), acc src2 bank 0
KERNEL(bitop_acc2_ab_diff, ACC8_SRC2("v33", "v34"))   // all different, acc in src2
KERNEL(bitop_acc2_a_acc, ACC8_SRC2("v32", "v34"))     // src0 in acc's bank, acc in src2
KERNEL(bitop_acc2_b_acc, ACC8_SRC2("v33", "v36"))     // src1 in acc's bank, acc in src2
// accumulator as src0 (and dst)
#define ACC8_SRC0(SA, SB)                                                       \
    "v_bitop3_b32 v0, v0, " SA ", " SB " bitop3:0xf6\n v_bitop3_b32 v4, v4, " SA ", " SB " bitop3:0xf6\n"   \
    "v_bitop3_b32 v8, v8, " SA ", " SB " bitop3:0xf6\n v_bitop3_b32 v12, v12, " SA ", " SB " bitop3:0xf6\n" \
    "v_bitop3_b32 v16, v16, " SA ", " SB " bitop3:0xf6\n v_bitop3_b32 v20, v20, " SA ", " SB " bitop3:0xf6\n" \
    "v_bitop3_b32 v24, v24, " SA ", " SB " bitop3:0xf6\n v_bitop3_b32 v28, v28, " SA ", " SB " bitop3:0xf6\n"
KERNEL(bitop_acc0_diff, ACC8_SRC0("v33", "v34"))
KERNEL(bitop_acc0_s1same, ACC8_SRC0("v32", "v34"))
KERNEL(bitop_acc0_s2same, ACC8_SRC0("v33", "v36"))
KERNEL(bitop_acc0_allsame, ACC8_SRC0("v32", "v36"))
KERNEL(bitop_src0_par, ACC8_B("v_bitop3_b32", "v34", "v33"))   // src0 bank 2 vs acc bank 0: parity or mod 4?
// two dependent chains only (like the real inner loop), 16 instructions per body
#define CH2 "v_bitop3_b32 v1, v34, v1, v37 bitop3:0xde\n v_bitop3_b32 v2, v33, v2, v36 bitop3:0xde\n"
KERNEL(bitop_2chains, CH2 CH2 CH2 CH2)
#define CH4 "v_bitop3_b32 v1, v34, v1, v37 bitop3:0xde\n v_bitop3_b32 v2, v33, v2, v36 bitop3:0xde\n v_bitop3_b32 v5, v38, v5, v37 bitop3:0xde\n v_bitop3_b32 v6, v39, v6, v36 bitop3:0xde\n"
KERNEL(bitop_4chains, CH4 CH4)

#define XOR8(SA)                                                                 \
    "v_xor_b32 v0, " SA ", v0\n v_xor_b32 v4, " SA ", v4\n v_xor_b32 v8, " SA ", v8\n" \
    "v_xor_b32 v12, " SA ", v12\n v_xor_b32 v16, " SA ", v16\n v_xor_b32 v20, " SA ", v20\n" \
    "v_xor_b32 v24, " SA ", v24\n v_xor_b32 v28, " SA ", v28\n"
KERNEL(xor_diff, XOR8("v33"))
KERNEL(xor_same, XOR8("v32"))
#define BCNT8(SA)                                                                \
    "v_bcnt_u32_b32 v0, " SA ", v0\n v_bcnt_u32_b32 v4, " SA ", v4\n v_bcnt_u32_b32 v8, " SA ", v8\n" \
    "v_bcnt_u32_b32 v12, " SA ", v12\n v_bcnt_u32_b32 v16, " SA ", v16\n v_bcnt_u32_b32 v20, " SA ", v20\n" \
    "v_bcnt_u32_b32 v24, " SA ", v24\n v_bcnt_u32_b32 v28, " SA ", v28\n"
KERNEL(bcnt_diff, BCNT8("v33"))

typedef void (*kern_t)(uint32_t *, int, uint64_t *);

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    uint64_t *clk;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(uint32_t)));
    CHECK(hipMalloc(&clk, 16));
    struct { const char *name; kern_t k; } ks[] = {
        {"bitop3 acc/a/b in 3 banks", bitop_diff}, {"bitop3 a in acc's bank", bitop_a_same},
        {"bitop3 a,b same bank", bitop_ab_same}, {"bitop3 all one bank", bitop_all_same},
        {"bitop3 SGPR a", bitop_sgpr}, {"bitop3 b in acc's bank", bitop_b_same},
        {"acc=src2: src0,src1 same bank", bitop_acc2_ab_same}, {"acc=src2: all different", bitop_acc2_ab_diff},
        {"acc=src2: src0 in acc bank", bitop_acc2_a_acc}, {"acc=src2: src1 in acc bank", bitop_acc2_b_acc},
        {"acc=src0=dst: others differ", bitop_acc0_diff}, {"acc=src0=dst: src1 same bank", bitop_acc0_s1same},
        {"acc=src0=dst: src2 same bank", bitop_acc0_s2same}, {"acc=src0=dst: all same bank", bitop_acc0_allsame},
        {"src0 bank2 vs acc bank0", bitop_src0_par},
        {"2 dependent chains", bitop_2chains}, {"4 dependent chains", bitop_4chains}, {"xor 2 banks", xor_diff}, {"xor same bank", xor_same},
        {"bcnt", bcnt_diff}};
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto &kk : ks) {
        for (int w : {1, 2, 3, 4}) {
            const int blocks = cus * w;
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, out, 10, clk);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t h[2];
            CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // memrealtime ticks at 100 MHz
            const double instr = (double)iters * 64.0;
            const double cyc_per_instr_per_simd = (double)h[0] / (instr * w);
            (void)cyc_per_instr_per_simd;
            printf("%-32s waves/SIMD %d: %.3f ms, clock %.2f GHz, %.2f clk per wave-instr per SIMD (wall)\n",
                   kk.name, w, ms, ghz, ms * 1e-3 * ghz * 1e9 / (instr * w));
        }
    }
    return 0;
}
