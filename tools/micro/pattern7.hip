// Micro-benchmark 7: paced strip writes (1024-thread workgroups of neighbouring strips, barrier per row),
// 1080-byte rows, with window reads as in pattern4.  Variant "shift": at output row y every wave
// stores the 64 strips that START at the next 64-byte sector boundary (strip index + (y mod 8)), i.e.
// whole sectors only -- what a neighbour-fragment exchange through LDS would make possible.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint32_t xcd_remap(uint32_t b, uint32_t n) { const uint32_t per = n / 8; return b < per * 8 ? (b % 8) * per + b / 8 : b; }

template <int SHIFT, int BAR, int W>
__global__ __launch_bounds__(1024) void k(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int frames, int seg_rows)
{
    const uint32_t strips = W / 8, nseg = 1920 / seg_rows, total = uint32_t(frames) * strips;
    const uint32_t nsb = (total + 1023) / 1024;
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t sb = blk % nsb, sg = blk / nsb;
    if (sg >= nseg) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g0 = sb * 1024 + wave * 64;
    const uint32_t src_rows = seg_rows * 36 / 64;
    const uint32_t gs = min(g0, total - 1), fs = gs / strips, ss = gs % strips;
    const size_t sbase = size_t(fs) * 1920 * 1080 + size_t(sg * src_rows) * 1920 + ((ss * 8 * 608 / 1080) & ~15u) + (lane % 20) * 16;
    u32x4 acc = {0, 0, 0, 0};
    int srow = -1;
    for (int r = 0; r < seg_rows; ++r) {
        const int need = r * 36 / 64;
        if (need != srow) {
            srow = need;
            if (lane < 20) { const u32x4 v = *reinterpret_cast<const u32x4 *>(src + sbase + size_t(srow) * 1920); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        }
        const uint32_t y = sg * seg_rows + r;
        const uint32_t ph = 7 * y + (g0 % strips); const uint32_t g = g0 + lane + (SHIFT == 1 ? ((8 - (ph & 7)) & 7) : SHIFT == 2 ? ((16 - (ph & 15)) & 15) : 0);
        if (BAR) __builtin_amdgcn_s_barrier();
        if (g < total) {
            const size_t addr = size_t(g / strips) * 1920 * W + size_t(y) * W + (g % strips) * 8;
            *reinterpret_cast<u32x2 *>(dst + addr) = u32x2{acc.x + uint32_t(r), acc.y ^ acc.z ^ acc.w};
        }
    }
}

template <int SHIFT, int BAR, int W = 1080>
double run(const uint8_t *src, uint8_t *dst, int frames, int seg_rows)
{
    const uint32_t total = uint32_t(frames) * (W / 8), nsb = (total + 1023) / 1024, nblk = nsb * (1920 / seg_rows);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHIFT, BAR, W>), dim3((nblk + 7) / 8 * 8), dim3(1024), 0, 0, src, dst, frames, seg_rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int frames = 300;
    uint8_t *src, *dst;
    hipMalloc(&src, size_t(frames) * 1920 * 1080 + (1 << 20)); hipMalloc(&dst, size_t(frames) * 1920 * 1080 + (1 << 20));
    hipMemset(src, 1, size_t(frames) * 1920 * 1080);
    const double bytes = double(frames) * (1920.0 * 1080 + 1080.0 * 608);
    for (int seg : {64, 96}) {
        const double a = run<0, 0>(src, dst, frames, seg), b = run<0, 1>(src, dst, frames, seg), c = run<1, 0>(src, dst, frames, seg), d = run<1, 1>(src, dst, frames, seg);
        const double z0 = run<0, 0, 1024>(src, dst, frames, seg), z1 = run<0, 1, 1024>(src, dst, frames, seg);
        printf("seg %3d: 1024-byte rows free %.3f ms paced %.3f ms (x1.0547 for equal bytes: %.3f / %.3f)\n", seg, z0, z1, z0 * 1.0547, z1 * 1.0547);
        const double e = run<2, 0>(src, dst, frames, seg), f = run<2, 1>(src, dst, frames, seg);
        printf("seg %3d: free %.3f ms (%.0f GB/s)  paced %.3f (%.0f)  64B-shifted free %.3f (%.0f) paced %.3f (%.0f)  128B-shifted free %.3f (%.0f) paced %.3f (%.0f)\n", seg,
               a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6, e, bytes / e / 1e6, f, bytes / f / 1e6);
    }
    return 0;
}
