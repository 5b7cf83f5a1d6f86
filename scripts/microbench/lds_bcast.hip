// LDS read-rate microbenchmark for gfx950: how many wave64 ds_read_b128 per ns does one CU
// sustain when every lane reads the same address (row broadcast, the way the pair kernels
// fetch reference rows), compared with lane-distinct addresses, and how much of it overlaps
// with full-rate VALU work.  (Calibration for DESIGN.md.)
// Build: hipcc --offload-arch=gfx950 -O3 lds_bcast.hip -o lds_bcast
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: uniform address (broadcast); 1: lane-distinct 16-byte slots (conflict-free);
// VALU = number of independent v_bitop3 per ds_read_b128 issued alongside.
template <int MODE, int VALU>
__global__ __launch_bounds__(1024) void k_lds(uint32_t *out, int iters)
{
    __shared__ u32x4 buf[4096];   // 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = u32x4{(uint32_t)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    uint32_t m[8];
    for (int i = 0; i < 8; ++i) m[i] = threadIdx.x * 2654435761u + i;
    uint32_t vb = threadIdx.x ^ 0x5bd1e995u, vc = threadIdx.x * 7u;
    int base = (threadIdx.x >> 6) * 64;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = MODE == 0 ? ((base + u * 8 + it) & 4095) : ((base + u * 64 + lane + it) & 4095);
            const u32x4 v = buf[idx];
            acc ^= v;
#pragma unroll
            for (int j = 0; j < VALU; ++j) {
                asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xde" : "+v"(m[(u * VALU + j) & 7]) : "v"(vb), "v"(vc));
            }
        }
    }
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    for (int i = 0; i < 8; ++i) r ^= m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef void (*kern_t)(uint32_t *, int);

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d\n", prop.gcnArchName, cus);
    uint32_t *out;
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(uint32_t));
    struct { const char *name; kern_t k; int valu; } ks[] = {
        {"broadcast ds_read_b128, no VALU", k_lds<0, 0>, 0},
        {"lane-distinct ds_read_b128, no VALU", k_lds<1, 0>, 0},
        {"broadcast + 4 bitop3/read", k_lds<0, 4>, 4},
        {"broadcast + 8 bitop3/read", k_lds<0, 8>, 8},
        {"broadcast + 16 bitop3/read", k_lds<0, 16>, 16},
        {"lane-distinct + 8 bitop3/read", k_lds<1, 8>, 8},
    };
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &kk : ks) {
        for (int waves : {4, 8, 16}) {   // waves per CU (one workgroup per CU)
            hipLaunchKernelGGL(kk.k, dim3(cus), dim3(waves * 64), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kk.k, dim3(cus), dim3(waves * 64), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double reads = (double)waves * iters * 8;       // wave-level ds_read_b128 per CU
            const double ns = ms * 1e6;
            printf("%-38s waves/CU %2d: %.3f ms  %.4f wave-reads/ns/CU (%.1f B/ns/CU returned)", kk.name, waves, ms,
                   reads / ns, reads * 1024 / ns);
            if (kk.valu) printf("  %.3f wave-bitop3/ns/SIMD", reads * kk.valu / ns / 4);
            printf("\n");
        }
    }
    return 0;
}
