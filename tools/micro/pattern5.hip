// Micro-benchmark 5: strip writes (64 lanes x 8 B per row, 64 rows per wave), flat strip->lane map.
// Which matters: the row width (strips per row not a multiple of 64) or the pitch (phase of the
// 128-byte lines drifting from row to row)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t xcd_remap(uint32_t b, uint32_t n) { const uint32_t per = n / 8; return b < per * 8 ? (b % 8) * per + b / 8 : b; }

__global__ __launch_bounds__(256) void k(uint8_t *__restrict__ dst, int frames, int wrow, int pitch, int segfast)
{
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t wave = (blk * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t strips = wrow / 8, nseg = 30;
    const uint32_t nwr = (uint32_t(frames) * strips + 63) / 64, nwr4 = (nwr + 3) / 4 * 4;
    uint32_t sg, wr;
    if (segfast) { sg = (wave >> 2) % nseg; wr = ((wave >> 2) / nseg) * 4 + (wave & 3); }   // block = 4 consecutive wave-rows, segments fastest
    else { sg = wave / nwr4; wr = wave % nwr4; }
    if (wr >= nwr || sg >= nseg) return;
    const uint32_t g = min(wr * 64 + lane, uint32_t(frames) * strips - 1);
    const size_t base = size_t(g / strips) * 1920 * pitch + size_t(sg * 64) * pitch + (g % strips) * 8;
    for (int r = 0; r < 64; ++r) *reinterpret_cast<u32x2 *>(dst + base + size_t(r) * pitch) = u32x2{uint32_t(r), lane};
}

double run(uint8_t *dst, int frames, int wrow, int pitch, int segfast)
{
    const uint32_t strips = wrow / 8, nwr4 = ((uint32_t(frames) * strips + 63) / 64 + 3) / 4 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(nwr4 / 4 * 30), dim3(256), 0, 0, dst, frames, wrow, pitch, segfast);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return double(frames) * 1920 * wrow / (best * 1e-3) / 1e9;
}

int main()
{
    const int frames = 300;
    uint8_t *dst; hipMalloc(&dst, size_t(frames) * 1920 * 1280 + (1 << 20));
    const int cfg[][2] = {{1024, 1024}, {1024, 1152}, {1024, 1088}, {1024, 1080}, {1024, 1040}, {1080, 1080}, {1080, 1152}, {1080, 1280},
                          {1088, 1088}, {1088, 1152}, {1056, 1056}, {1152, 1152}, {960, 960}, {960, 1024}};
    for (auto &c : cfg)
        printf("row %4d B pitch %4d B   strips-fastest %5.0f GB/s   segments-fastest %5.0f GB/s\n", c[0], c[1],
               run(dst, frames, c[0], c[1], 0), run(dst, frames, c[0], c[1], 1));
    return 0;
}
