// Micro-benchmark 2: which property of the strip write pattern costs the bandwidth?
// Every variant writes frames * H rows of W bytes; only the lane->address map changes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// B = bytes per lane per row (8 or 16); pitch = row pitch in bytes; wrow = bytes written per row
template <int B>
__global__ __launch_bounds__(256) void k(uint8_t *__restrict__ dst, int frames, int wrow, int pitch, int rows, int seg, int wave_private)
{
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t strips = wrow / B, nseg = rows / seg;
    const uint32_t nwr = (uint32_t(frames) * strips + 63) / 64;
    const uint32_t sg = wave % nseg, wr = wave / nseg;
    if (wr >= nwr) return;
    size_t base;
    if (wave_private) {
        // each wave owns whole rows: lanes cover B*64 contiguous bytes, rows of the same frame follow each other
        base = size_t(wr) * rows * pitch + size_t(sg * seg) * pitch + lane * B;   // pitch == 64*B here
    } else {
        const uint32_t g = min(wr * 64 + lane, uint32_t(frames) * strips - 1);
        base = size_t(g / strips) * rows * pitch + size_t(sg * seg) * pitch + (g % strips) * B;
    }
    for (int r = 0; r < seg; ++r) {
        if (B == 8) *reinterpret_cast<u32x2 *>(dst + base + size_t(r) * pitch) = u32x2{uint32_t(r), lane};
        else *reinterpret_cast<u32x4 *>(dst + base + size_t(r) * pitch) = u32x4{uint32_t(r), lane, 0u, 1u};
    }
}

template <int B>
double run(uint8_t *dst, int frames, int wrow, int pitch, int rows, int seg, int wp)
{
    const uint32_t strips = wrow / B, nwaves = ((uint32_t(frames) * strips + 63) / 64) * (rows / seg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<B>, dim3((nwaves + 3) / 4), dim3(256), 0, 0, dst, frames, wrow, pitch, rows, seg, wp);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return double(frames) * rows * wrow / (best * 1e-3) / 1e9;
}

int main()
{
    const int frames = 300;
    uint8_t *dst; hipMalloc(&dst, size_t(frames) * 1920 * 1152 + (1 << 20)); 
    printf("8B/lane  w1080 pitch1080 seg64 (the kernel)      %.0f GB/s\n", run<8>(dst, frames, 1080, 1080, 1920, 64, 0));
    printf("8B/lane  w1024 pitch1024 seg64 (aligned rows)    %.0f GB/s\n", run<8>(dst, frames, 1024, 1024, 1920, 64, 0));
    printf("8B/lane  w1080 pitch1152 seg64 (aligned pitch)   %.0f GB/s\n", run<8>(dst, frames, 1080, 1152, 1920, 64, 0));
    printf("16B/lane w1072 pitch1080 seg64                   %.0f GB/s\n", run<16>(dst, frames, 1072, 1080, 1920, 64, 0));
    printf("16B/lane w1024 pitch1024 seg64 (aligned)         %.0f GB/s\n", run<16>(dst, frames, 1024, 1024, 1920, 64, 0));
    printf("8B/lane  w512  pitch512  seg64 wave-private rows  %.0f GB/s\n", run<8>(dst, frames * 2, 512, 512, 1920, 64, 1));
    printf("16B/lane w1024 pitch1024 seg64 wave-private rows  %.0f GB/s\n", run<16>(dst, frames, 1024, 1024, 1920, 64, 1));
    printf("8B/lane  w1080 pitch1080 seg16                    %.0f GB/s\n", run<8>(dst, frames, 1080, 1080, 1920, 16, 0));
    printf("8B/lane  w1080 pitch1080 seg240                   %.0f GB/s\n", run<8>(dst, frames, 1080, 1080, 1920, 240, 0));
    printf("8B/lane  w1080 pitch1080 seg1920                  %.0f GB/s\n", run<8>(dst, frames, 1080, 1080, 1920, 1920, 0));
    return 0;
}
