// valu_clock.hip -- what the vector ALU and the LDS read pipe of an MI355X sustain, IN CYCLES,
// and the clock the chip holds while they do (MI355X_MICROARCH.md "DVFS give-back" item 6):
//
//   * every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop,
//     after ~1 s of back-to-back launches; clock = d(memtime) / d(memrealtime) x 100 MHz;
//   * part 1: single-instruction streams (8 independent chains per wave) at 1..4 waves per SIMD
//     -> cycles per wave-instruction per SIMD;
//   * part 2: the pair kernel's inner loop without global memory: R rows broadcast from LDS
//     (7 x ds_read_b128 per row and chunk, rolling one-row prefetch), JL columns per lane held in
//     registers, 28 v_bitop3/v_xor + 2 v_bcnt per (row, column, chunk) -> how close each
//     (R, JL) shape gets to the VALU's 2 cycles per instruction, and what the LDS pipe allows.
//
// Build: hipcc --offload-arch=gfx950 -O3 valu_clock.hip -o valu_clock
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

struct Stamp {
    uint64_t cycles, real;
    uint32_t hw, xcc;
};

__device__ __forceinline__ void stamp_begin(uint64_t &t, uint64_t &r)
{
    __builtin_amdgcn_s_barrier();
    t = __builtin_amdgcn_s_memtime();
    r = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ void stamp_end(Stamp *out, uint64_t t0, uint64_t r0)
{
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63u) == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        Stamp s;
        s.cycles = t1 - t0;
        s.real = r1 - r0;
        s.hw = hw;
        s.xcc = xcc;
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}

#define REP8(X) X X X X X X X X
#define I_ADD(i) "v_add_u32 %" #i ", %9, %" #i "\n"
#define I_XOR_V(i) "v_xor_b32 %" #i ", %9, %" #i "\n"
#define I_XOR_S(i) "v_xor_b32 %" #i ", %8, %" #i "\n"
#define I_BITOP_V(i) "v_bitop3_b32 %" #i ", %10, %9, %" #i " bitop3:0xbe\n"
#define I_BCNT(i) "v_bcnt_u32_b32 %" #i ", %9, %" #i "\n"
#define I_FMA(i) "v_fma_f32 %" #i ", %9, %10, %" #i "\n"
#define I_OR3(i) "v_or3_b32 %" #i ", %" #i ", %9, %10\n"

#define KERNEL(NAME, INSTR)                                                                       \
    __global__ __launch_bounds__(256) void NAME(Stamp *stamps, uint32_t *sink, uint32_t sa_in, int iters) \
    {                                                                                             \
        uint32_t m[8];                                                                            \
        for (int i = 0; i < 8; ++i) m[i] = threadIdx.x * 2654435761u + i;                         \
        uint32_t sa = __builtin_amdgcn_readfirstlane(sa_in);                                      \
        uint32_t vb = threadIdx.x ^ 0x5bd1e995u, vc = threadIdx.x * 7u + 3u;                      \
        uint64_t t0, r0;                                                                          \
        stamp_begin(t0, r0);                                                                      \
        for (int it = 0; it < iters; ++it) {                                                      \
            REP8(asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
                              : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), \
                                "+v"(m[6]), "+v"(m[7])                                            \
                              : "s"(sa), "v"(vb), "v"(vc));)                                      \
        }                                                                                         \
        stamp_end(stamps, t0, r0);                                                                \
        uint32_t r = 0;                                                                           \
        for (int i = 0; i < 8; ++i) r ^= m[i];                                                    \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = r;                                          \
    }

KERNEL(k_add, I_ADD)
KERNEL(k_xor_v, I_XOR_V)
KERNEL(k_xor_s, I_XOR_S)
KERNEL(k_bitop_v, I_BITOP_V)
KERNEL(k_bcnt, I_BCNT)
KERNEL(k_fma, I_FMA)
KERNEL(k_or3, I_OR3)

// the same v_bitop3 stream with D accumulators instead of 8: instruction i depends on i - D
#define I_BITOP_D1(i) "v_bitop3_b32 %0, %10, %9, %0 bitop3:0xbe\n"
#define I_BITOP_D2(i) "v_bitop3_b32 %" #i ", %10, %9, %" #i " bitop3:0xbe\n"
#define KERNEL_D(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                            \
    __global__ __launch_bounds__(256) void NAME(Stamp *stamps, uint32_t *sink, uint32_t sa_in, int iters) \
    {                                                                                             \
        uint32_t m[8];                                                                            \
        for (int i = 0; i < 8; ++i) m[i] = threadIdx.x * 2654435761u + i;                         \
        uint32_t sa = __builtin_amdgcn_readfirstlane(sa_in);                                      \
        uint32_t vb = threadIdx.x ^ 0x5bd1e995u, vc = threadIdx.x * 7u + 3u;                      \
        uint64_t t0, r0;                                                                          \
        stamp_begin(t0, r0);                                                                      \
        for (int it = 0; it < iters; ++it) {                                                      \
            REP8(asm volatile(I_BITOP_D2(I0) I_BITOP_D2(I1) I_BITOP_D2(I2) I_BITOP_D2(I3) I_BITOP_D2(I4) I_BITOP_D2(I5) I_BITOP_D2(I6) I_BITOP_D2(I7) \
                              : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), \
                                "+v"(m[6]), "+v"(m[7])                                            \
                              : "s"(sa), "v"(vb), "v"(vc));)                                      \
        }                                                                                         \
        stamp_end(stamps, t0, r0);                                                                \
        uint32_t r = 0;                                                                           \
        for (int i = 0; i < 8; ++i) r ^= m[i];                                                    \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = r;                                          \
    }
KERNEL_D(k_bitop_d1, 0, 0, 0, 0, 0, 0, 0, 0)
KERNEL_D(k_bitop_d2, 0, 1, 0, 1, 0, 1, 0, 1)
KERNEL_D(k_bitop_d3, 0, 1, 2, 0, 1, 2, 0, 1)
KERNEL_D(k_bitop_d4, 0, 1, 2, 3, 0, 1, 2, 3)
KERNEL_D(k_bitop_pairs, 0, 0, 1, 1, 2, 2, 3, 3)   // the compiler's order in the pair kernel: dependent pairs back to back

// does a non-VALU instruction between VALU instructions cost the SIMD an issue slot?  The same 64
// v_bitop3 per iteration with one extra instruction after every 4th (16 per iteration)
#define I_BITOP_X(i, EXTRA) "v_bitop3_b32 %" #i ", %10, %9, %" #i " bitop3:0xbe\n" EXTRA
#define KERNEL_X(NAME, EXTRA)                                                                     \
    __global__ __launch_bounds__(256) void NAME(Stamp *stamps, uint32_t *sink, uint32_t sa_in, int iters) \
    {                                                                                             \
        uint32_t m[8];                                                                            \
        for (int i = 0; i < 8; ++i) m[i] = threadIdx.x * 2654435761u + i;                         \
        uint32_t sa = __builtin_amdgcn_readfirstlane(sa_in);                                      \
        uint32_t vb = threadIdx.x ^ 0x5bd1e995u, vc = threadIdx.x * 7u + 3u;                      \
        uint64_t t0, r0;                                                                          \
        stamp_begin(t0, r0);                                                                      \
        for (int it = 0; it < iters; ++it) {                                                      \
            REP8(asm volatile(I_BITOP_X(0, "") I_BITOP_X(1, "") I_BITOP_X(2, "") I_BITOP_X(3, EXTRA)      \
                              I_BITOP_X(4, "") I_BITOP_X(5, "") I_BITOP_X(6, "") I_BITOP_X(7, EXTRA)      \
                              : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), \
                                "+v"(m[6]), "+v"(m[7])                                            \
                              : "s"(sa), "v"(vb), "v"(vc));)                                      \
        }                                                                                         \
        stamp_end(stamps, t0, r0);                                                                \
        uint32_t r = 0;                                                                           \
        for (int i = 0; i < 8; ++i) r ^= m[i];                                                    \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = r;                                          \
    }
KERNEL_X(k_x_none, "")
KERNEL_X(k_x_waitcnt, "s_waitcnt lgkmcnt(0)\n")
KERNEL_X(k_x_nop, "s_nop 0\n")
KERNEL_X(k_x_salu, "s_add_u32 s40, s40, 1\n")
KERNEL_X(k_x_vmov, "v_mov_b32 v60, v61\n")

// ---- part 2: the pair kernel's inner loop, operands already on chip ----
__device__ __forceinline__ uint32_t bitop_vvv(uint32_t m, uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_bitop3_b32(a, b, m, 0xBE);   // m | (a ^ b)
}

template <int R, int JL, bool LDS_ROWS, int ORD = 0>
__global__ __launch_bounds__(256) void k_inner(Stamp *stamps, uint32_t *sink, uint32_t seed, int iters)
{
    __shared__ uint4 lds_rows[4][R * 7];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t p = lane; p < R * 7; p += 64) {
        lds_rows[wave][p] = make_uint4(p * seed, p ^ seed, p + seed, p * 31u + seed);
    }
    uint4 b[JL][7];
#pragma unroll
    for (int j = 0; j < JL; ++j) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            b[j][q] = make_uint4(tid * 2654435761u + q, tid ^ (seed + j), tid * 7u + q * j, tid + q);
        }
    }
    uint32_t cnt[R * JL];
#pragma unroll
    for (int x = 0; x < R * JL; ++x) cnt[x] = 0;
    __syncthreads();
    uint64_t t0, r0;
    stamp_begin(t0, r0);
    const uint4 *rows = &lds_rows[wave][0];
    for (int it = 0; it < iters; ++it) {
        uint4 a[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) a[q] = rows[q];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t mlo[JL], mhi[JL];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                if constexpr (ORD == 0) {
#pragma unroll
                    for (int j = 0; j < JL; ++j) {
                        if (q == 0) {
                            mlo[j] = a[0].x ^ b[j][0].y;
                            mhi[j] = a[0].y ^ b[j][0].x;
                        } else {
                            mlo[j] = bitop_vvv(mlo[j], a[q].x, b[j][q].y);
                            mhi[j] = bitop_vvv(mhi[j], a[q].y, b[j][q].x);
                        }
                        mlo[j] = bitop_vvv(mlo[j], a[q].z, b[j][q].w);
                        mhi[j] = bitop_vvv(mhi[j], a[q].w, b[j][q].z);
                    }
                } else {
                    // every chain advances once, then every chain again: dependency distance 2 JL
#pragma unroll
                    for (int j = 0; j < JL; ++j) {
                        if (q == 0) {
                            mlo[j] = a[0].x ^ b[j][0].y;
                            mhi[j] = a[0].y ^ b[j][0].x;
                        } else {
                            mlo[j] = bitop_vvv(mlo[j], a[q].x, b[j][q].y);
                            mhi[j] = bitop_vvv(mhi[j], a[q].y, b[j][q].x);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < JL; ++j) {
                        mlo[j] = bitop_vvv(mlo[j], a[q].z, b[j][q].w);
                        mhi[j] = bitop_vvv(mhi[j], a[q].w, b[j][q].z);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (LDS_ROWS) {
                    if (r + 1 < R) a[q] = rows[(r + 1) * 7 + q];
                } else {
                    asm volatile("" : "+v"(a[q].x), "+v"(a[q].y), "+v"(a[q].z), "+v"(a[q].w));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < JL; ++j) {
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r * JL + j]) : "v"(mlo[j]));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r * JL + j]) : "v"(mhi[j]));
            }
        }
        // keep the column registers opaque so nothing is hoisted out of the loop
#pragma unroll
        for (int j = 0; j < JL; ++j) {
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                asm volatile("" : "+v"(b[j][q].x), "+v"(b[j][q].y), "+v"(b[j][q].z), "+v"(b[j][q].w));
            }
        }
    }
    stamp_end(stamps, t0, r0);
    uint32_t s = 0;
#pragma unroll
    for (int x = 0; x < R * JL; ++x) s += cnt[x];
    sink[blockIdx.x * blockDim.x + tid] = s;
}

typedef void (*kern_t)(Stamp *, uint32_t *, uint32_t, int);

struct Result {
    double cycles_med, clock_ghz, ms;
    int waves_min, waves_max;
};

static Result run(kern_t k, int blocks, int iters, Stamp *d_stamps, uint32_t *d_sink, double warm_s)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // back-to-back launches until the clock has settled
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, 123u, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, 123u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms1 = 0.f;
    hipEventElapsedTime(&ms1, e0, e1);
    const int reps = std::max(2, (int)(warm_s * 1e3 / std::max(ms1, 0.01f)));
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, 123u, iters);
    }
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, 123u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const int n_waves = blocks * 4;
    std::vector<Stamp> st(n_waves);
    hipMemcpy(st.data(), d_stamps, n_waves * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc(n_waves), clk(n_waves);
    std::map<uint64_t, int> per_simd;
    for (int i = 0; i < n_waves; ++i) {
        cyc[i] = (double)st[i].cycles;
        clk[i] = (double)st[i].cycles / (double)st[i].real * 0.1;   // GHz
        per_simd[((uint64_t)st[i].xcc << 32) | (st[i].hw & 0xFFFFFFF0u)]++;
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    Result r;
    r.cycles_med = cyc[n_waves / 2];
    r.clock_ghz = clk[n_waves / 2];
    r.ms = ms;
    r.waves_min = 1 << 30;
    r.waves_max = 0;
    for (auto &kv : per_simd) {
        r.waves_min = std::min(r.waves_min, kv.second);
        r.waves_max = std::max(r.waves_max, kv.second);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return r;
}

int main(int argc, char **argv)
{
    const double warm_s = argc > 1 ? atof(argv[1]) : 1.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d nominal clock %.2f GHz; %.1f s of back-to-back launches before each stamp\n",
           prop.gcnArchName, cus, prop.clockRate / 1e6, warm_s);
    Stamp *d_stamps;
    uint32_t *d_sink;
    hipMalloc(&d_stamps, (size_t)cus * 8 * 4 * sizeof(Stamp));
    hipMalloc(&d_sink, (size_t)cus * 8 * 256 * sizeof(uint32_t));

    printf("\n== part 1: single-instruction streams (64 instructions x iters per wave, 8 chains) ==\n");
    struct { const char *name; kern_t k; } ks[] = {
        {"v_add_u32 v,v,v", k_add},       {"v_xor_b32 v,v,v", k_xor_v},   {"v_xor_b32 v,s,v", k_xor_s},
        {"v_bitop3_b32 v,v,v,v", k_bitop_v}, {"v_bcnt_u32_b32 v,v,v", k_bcnt}, {"v_fma_f32 v,v,v,v", k_fma},
        {"v_or3_b32 v,v,v,v", k_or3},
        {"v_bitop3 dep. distance 1", k_bitop_d1}, {"v_bitop3 dep. distance 2", k_bitop_d2},
        {"v_bitop3 dep. distance 3", k_bitop_d3}, {"v_bitop3 dep. distance 4", k_bitop_d4},
        {"v_bitop3 dependent pairs", k_bitop_pairs},
        {"64 bitop3 (reference)", k_x_none}, {"64 bitop3 + 16 s_waitcnt", k_x_waitcnt}, {"64 bitop3 + 16 s_nop", k_x_nop},
        {"64 bitop3 + 16 s_add_u32", k_x_salu}, {"64 bitop3 + 16 v_mov_b32", k_x_vmov}};
    const int iters = 20000;
    for (auto &kk : ks) {
        for (int w : {1, 2, 3, 4}) {
            const Result r = run(kk.k, cus * w, iters, d_stamps, d_sink, warm_s);
            const double instr = (double)iters * 64.0;
            printf("%-24s waves/SIMD %d (seen %d..%d): %7.3f ms  in-kernel clock %.3f GHz  %.2f cycles/instr/wave  -> %.2f cycles/instr/SIMD\n",
                   kk.name, w, r.waves_min, r.waves_max, r.ms, r.clock_ghz, r.cycles_med / instr,
                   r.cycles_med / instr / w);
        }
    }

    printf("\n== part 2: pair-kernel inner loop, operands on chip (per (row, column, chunk): 28 bitop/xor + 2 bcnt = 32 issue slots of 2 cycles at datasheet rate) ==\n");
    struct { const char *name; kern_t k; int R, JL; int max_w; } ps[] = {
        {"R=16 JL=2 rows from LDS", k_inner<16, 2, true>, 16, 2, 4},
        {"R=16 JL=2 rows in regs ", k_inner<16, 2, false>, 16, 2, 3},
        {"R=16 JL=2 LDS, chains interleaved", k_inner<16, 2, true, 1>, 16, 2, 4},
        {"R=16 JL=3 LDS, chains interleaved", k_inner<16, 3, true, 1>, 16, 3, 2},
        {"R=16 JL=3 rows from LDS", k_inner<16, 3, true>, 16, 3, 2},
        {"R=16 JL=4 rows from LDS", k_inner<16, 4, true>, 16, 4, 2},
        {"R=16 JL=4 rows in regs ", k_inner<16, 4, false>, 16, 4, 2},
        {"R=8  JL=3 rows from LDS", k_inner<8, 3, true>, 8, 3, 3},
        {"R=8  JL=4 rows from LDS", k_inner<8, 4, true>, 8, 4, 2},
    };
    for (auto &p : ps) {
        for (int w = 1; w <= p.max_w; ++w) {
            const int it2 = 4000 / p.JL;
            const Result r = run(p.k, cus * w, it2, d_stamps, d_sink, warm_s);
            const double slots = (double)it2 * p.R * p.JL * 32.0;          // 28 full-rate + 2 half-rate
            const double instr = (double)it2 * p.R * p.JL * 30.0;
            const double cyc_per_slot_simd = r.cycles_med / slots / w;
            printf("%-26s waves/SIMD %d (seen %d..%d): %7.3f ms  clock %.3f GHz  %.2f cycles/issue-slot/SIMD (%.1f %% of the 2-cycle rate), %.2f cycles/instr/SIMD\n",
                   p.name, w, r.waves_min, r.waves_max, r.ms, r.clock_ghz, cyc_per_slot_simd,
                   100.0 * 2.0 / cyc_per_slot_simd, r.cycles_med / instr / w);
        }
    }
    return 0;
}
