// Micro-test: does gfx950 serve byte-misaligned ds_read_b32 / ds_read_b64 / ds_read2_b32 correctly?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = uint8_t(i * 7 + 3);
    __syncthreads();
    const uint32_t base = uint32_t(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)&lds[0]));
    for (int m = 0; m < 8; ++m) {
        const uint32_t addr = base + threadIdx.x * 24 + m;
        u32x2_t a; uint32_t b;
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
        out[(m * 64 + threadIdx.x) * 3 + 0] = a.x;
        out[(m * 64 + threadIdx.x) * 3 + 1] = a.y;
        out[(m * 64 + threadIdx.x) * 3 + 2] = b;
    }
}
int main()
{
    uint32_t *d, h[8 * 64 * 3];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int m = 0; m < 8; ++m) {
        int bad64 = 0, bad32 = 0;
        for (int t = 0; t < 64; ++t) {
            uint8_t e[8];
            for (int i = 0; i < 8; ++i) e[i] = uint8_t((t * 24 + m + i) * 7 + 3);
            uint32_t e0, e1; memcpy(&e0, e, 4); memcpy(&e1, e + 4, 4);
            bad64 += h[(m * 64 + t) * 3] != e0 || h[(m * 64 + t) * 3 + 1] != e1;
            bad32 += h[(m * 64 + t) * 3 + 2] != e0;
        }
        printf("misalign %d: ds_read_b64 bad lanes %d, ds_read_b32 bad lanes %d\n", m, bad64, bad32);
    }
    return 0;
}
