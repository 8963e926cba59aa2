// v_ashr_pk_u8_i32 on gfx950: (a) the low-half form, (b) op_sel:[0,0,0,1] writing the HIGH half of the destination
// while the low half stays -- four vertical sums to four clipped bytes in two instructions.  Checked against the
// host for edge values.  Build: hipcc --offload-arch=gfx950 -O2 -o ashr_pk ashr_pk.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k(const int *v, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v0 = v[4 * i], v1 = v[4 * i + 1], v2 = v[4 * i + 2], v3 = v[4 * i + 3];
    uint32_t d = 0xdeadbeefu;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 19" : "+v"(d) : "v"(v0), "v"(v1));
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 19 op_sel:[0,0,0,1]" : "+v"(d) : "v"(v2), "v"(v3));
    out[i] = d;
}

static uint32_t clip8(int x) { return uint32_t(x < 0 ? 0 : x > 255 ? 255 : x); }

int main()
{
    std::vector<int> h;
    const int edge[] = {0, 1, -1, (1 << 19) - 1, 1 << 19, -(1 << 19), 255 << 19, (255 << 19) + 524287, 256 << 19, 2147483647, -2147483647 - 1, 64 << 12, 123456789, -123456789};
    for (int a : edge) for (int b : edge) { h.push_back(a); h.push_back(b); h.push_back(b ^ 0x55555); h.push_back(a + 77777); }
    uint32_t s = 12345;
    for (int i = 0; i < 4 * 100000; ++i) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; h.push_back(int(s) >> (s & 7)); }
    const int n = int(h.size() / 4);
    int *dv; uint32_t *dout;
    hipMalloc(&dv, h.size() * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dv, dout, n);
    std::vector<uint32_t> o(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t want = clip8(h[4 * i] >> 19) | clip8(h[4 * i + 1] >> 19) << 8 | clip8(h[4 * i + 2] >> 19) << 16 | clip8(h[4 * i + 3] >> 19) << 24;
        if (o[i] != want && bad++ < 10) std::printf("i=%d v=%d %d %d %d got %08x want %08x\n", i, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3], o[i], want);
    }
    std::printf("%d of %d differ\n", bad, n);
    return bad != 0;
}
