// Micro-benchmark 3: can split 128-B lines be merged in L2 if their two writers are kept close
// (same XCD, block barrier per row)?  All variants: 300 frames of 1080x1920 bytes, 8 B per lane per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ inline uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t per = n / 8, full = per * 8;
    if (b >= full) return b;
    return (b % 8) * per + b / 8;
}

// mode bit0: barrier per row; bit1: xcd remap; bit2: constant 56-byte misalignment on a 1024 grid
template <int BAR, int XCD>
__global__ __launch_bounds__(256) void k(uint8_t *__restrict__ dst, int frames, int wrow, int pitch, int rows, int seg, int shift)
{
    uint32_t blk = XCD ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t wave = (blk * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t strips = wrow / 8, nseg = rows / seg;
    const uint32_t nwr = (uint32_t(frames) * strips + 63) / 64;
    // consecutive waves of a block take consecutive strips of the SAME segment
    const uint32_t nwr4 = (nwr + 3) / 4 * 4;
    const uint32_t sg = wave / nwr4, wr = wave % nwr4;
    const bool live = wr < nwr && sg < nseg;
    const uint32_t g = min(wr * 64 + lane, uint32_t(frames) * strips - 1);
    size_t base = size_t(g / strips) * rows * pitch + size_t(sg * seg) * pitch + (g % strips) * 8 + shift;
    for (int r = 0; r < seg; ++r) {
        if (live) *reinterpret_cast<u32x2 *>(dst + base + size_t(r) * pitch) = u32x2{uint32_t(r), lane};
        if (BAR) __syncthreads();
    }
}

// block owns whole rows of K frames: K*135 strips in ceil(K*135/64) waves
template <int BAR, int XCD, int K, int T>
__global__ __launch_bounds__(T) void kb(uint8_t *__restrict__ dst, int frames, int rows, int seg)
{
    uint32_t blk = XCD ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t nfb = frames / K, nseg = rows / seg;
    const uint32_t sg = blk / nfb, fb = blk % nfb;
    const uint32_t t = threadIdx.x;
    const bool live = t < K * 135 && sg < nseg;
    const uint32_t tt = min(t, uint32_t(K * 135 - 1));
    size_t base = size_t(fb * K + tt / 135) * rows * 1080 + size_t(sg * seg) * 1080 + (tt % 135) * 8;
    for (int r = 0; r < seg; ++r) {
        if (live) *reinterpret_cast<u32x2 *>(dst + base + size_t(r) * 1080) = u32x2{uint32_t(r), t};
        if (BAR) __syncthreads();
    }
}

template <typename F>
double timeit(F launch, double bytes)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return bytes / (best * 1e-3) / 1e9;
}

template <int BAR, int XCD>
double run(uint8_t *dst, int frames, int wrow, int pitch, int rows, int seg, int shift)
{
    const uint32_t strips = wrow / 8;
    const uint32_t nwr4 = ((uint32_t(frames) * strips + 63) / 64 + 3) / 4 * 4;
    const uint32_t nblk = nwr4 / 4 * (rows / seg);
    return timeit([&] { hipLaunchKernelGGL((k<BAR, XCD>), dim3(nblk), dim3(256), 0, 0, dst, frames, wrow, pitch, rows, seg, shift); },
                  double(frames) * rows * wrow);
}
template <int BAR, int XCD, int K, int T>
double runb(uint8_t *dst, int frames, int rows, int seg)
{
    const uint32_t nblk = frames / K * (rows / seg);
    return timeit([&] { hipLaunchKernelGGL((kb<BAR, XCD, K, T>), dim3(nblk), dim3(T), 0, 0, dst, frames, rows, seg); },
                  double(frames) * rows * 1080);
}

int main()
{
    const int frames = 300;
    uint8_t *dst; hipMalloc(&dst, size_t(frames) * 1920 * 1152 + (1 << 20));
    printf("w1080 plain                          %.0f GB/s\n", run<0, 0>(dst, frames, 1080, 1080, 1920, 64, 0));
    printf("w1080 barrier                        %.0f GB/s\n", run<1, 0>(dst, frames, 1080, 1080, 1920, 64, 0));
    printf("w1080 xcd                            %.0f GB/s\n", run<0, 1>(dst, frames, 1080, 1080, 1920, 64, 0));
    printf("w1080 barrier+xcd                    %.0f GB/s\n", run<1, 1>(dst, frames, 1080, 1080, 1920, 64, 0));
    printf("w1024 aligned                        %.0f GB/s\n", run<0, 0>(dst, frames, 1024, 1024, 1920, 64, 0));
    printf("w1024 shifted 56                     %.0f GB/s\n", run<0, 0>(dst, frames, 1024, 1024, 1920, 64, 56));
    printf("w1024 shifted 56 barrier+xcd         %.0f GB/s\n", run<1, 1>(dst, frames, 1024, 1024, 1920, 64, 56));
    printf("w1024 shifted 64                     %.0f GB/s\n", run<0, 0>(dst, frames, 1024, 1024, 1920, 64, 64));
    printf("w1024 shifted 32                     %.0f GB/s\n", run<0, 0>(dst, frames, 1024, 1024, 1920, 64, 32));
    printf("rows-per-block K=4 T=576             %.0f GB/s\n", runb<0, 0, 4, 576>(dst, frames, 1920, 64));
    printf("rows-per-block K=4 T=576 barrier     %.0f GB/s\n", runb<1, 0, 4, 576>(dst, frames, 1920, 64));
    printf("rows-per-block K=4 T=576 barrier+xcd %.0f GB/s\n", runb<1, 1, 4, 576>(dst, frames, 1920, 64));
    printf("rows-per-block K=1 T=192             %.0f GB/s\n", runb<0, 0, 1, 192>(dst, frames, 1920, 64));
    printf("rows-per-block K=1 T=192 barrier     %.0f GB/s\n", runb<1, 0, 1, 192>(dst, frames, 1920, 64));
    printf("rows-per-block K=2 T=320 barrier     %.0f GB/s\n", runb<1, 0, 2, 320>(dst, frames, 1920, 64));
    return 0;
}
