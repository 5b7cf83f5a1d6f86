// Does data written by one kernel stay in the writing XCD's L2 for the next kernel?  Kernel A: workgroup b
// writes region b of a buffer (4 KB per workgroup).  Kernel B: workgroup b reads region (b + shift) % grid and
// adds it up.  shift = 0: the reader is on the writer's XCD (blockIdx % 8 labels the XCD); shift = 1: on the
// next XCD; shift = 8: same XCD, another CU.  Times of kernel B by HIP events, median of many launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void writer(uint32_t *buf, uint32_t words_per_wg, uint32_t v)
{
    uint32_t *p = buf + (size_t)blockIdx.x * words_per_wg;
    for (uint32_t i = threadIdx.x; i < words_per_wg; i += blockDim.x) p[i] = v + i;
}

__global__ void reader(const uint32_t *buf, uint32_t words_per_wg, uint32_t shift, uint32_t *out)
{
    const uint32_t src = (blockIdx.x + shift) % gridDim.x;
    const uint32_t *p = buf + (size_t)src * words_per_wg;
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < words_per_wg; i += blockDim.x) s += p[i];
    if (s == 0x12345678u) out[0] = s;
}

int main(int argc, char **argv)
{
    const uint32_t mb = argc > 1 ? atoi(argv[1]) : 10;
    const uint32_t words_per_wg = 1024;   // 4 KB
    const uint32_t grid = mb * 256;       // 256 workgroups per MB
    uint32_t *buf, *out;
    CK(hipMalloc(&buf, (size_t)grid * words_per_wg * 4));
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t shift : {0u, 1u, 8u, 4u, 0u, 1u}) {
        std::vector<float> t;
        for (int it = 0; it < 60; ++it) {
            hipLaunchKernelGGL(writer, dim3(grid), dim3(256), 0, 0, buf, words_per_wg, (uint32_t)it);
            hipEventRecord(e0);
            hipLaunchKernelGGL(reader, dim3(grid), dim3(256), 0, 0, buf, words_per_wg, shift, out);
            hipEventRecord(e1);
            CK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("%u MB, reader shift %u: median %.2f us, min %.2f us\n", mb, shift, t[t.size() / 2], t[0]);
    }
    return 0;
}
