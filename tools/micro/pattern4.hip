// Micro-benchmark 4: the read/write MIX of the portrait job without arithmetic.
// Per wave: 512 output bytes per row (64 lanes x 8 B), 64 output rows per segment, and one source
// row of ~304 bytes (19 lanes x 16 B) per 64/36 output rows.  Variants change only the layouts:
//   reads  windowed: 608-byte window of 1920-byte rows (the real source)  | packed: 608-byte rows
//   writes 1080-byte rows (real, chunks unaligned)                        | 1024-byte rows (aligned chunks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline uint32_t xcd_remap(uint32_t b, uint32_t n) { const uint32_t per = n / 8; return b < per * 8 ? (b % 8) * per + b / 8 : b; }

__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int frames,
                                         int wrow, int dpitch, int spitch, int sframe, int do_read, int do_write)
{
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t wave = (blk * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t strips = wrow / 8, nseg = 1920 / 64;
    const uint32_t nwr = (uint32_t(frames) * strips + 63) / 64;
    const uint32_t sg = wave % nseg, wr = wave / nseg;      // segment fastest, like the library
    if (wr >= nwr) return;
    const uint32_t g = min(wr * 64 + lane, uint32_t(frames) * strips - 1);
    const uint32_t f = g / strips, st = g % strips;
    const size_t dbase = size_t(f) * 1920 * dpitch + size_t(sg * 64) * dpitch + st * 8;
    // source: wave's window starts at (first strip of the wave) * 8 * 608/1080, 16-byte aligned; lanes < 20 fetch
    const uint32_t g0 = wr * 64, f0 = g0 / strips, st0 = g0 % strips;
    const size_t sbase = size_t(f0) * sframe + size_t(sg * 36) * spitch + ((st0 * 8 * 608 / 1080) & ~15u) + (lane % 20) * 16;
    u32x4 acc = {0, 0, 0, 0};
    int srow = -1;
    for (int r = 0; r < 64; ++r) {
        const int need = r * 36 / 64;
        if (do_read && need != srow) {
            srow = need;
            if (lane < 20) { const u32x4 v = *reinterpret_cast<const u32x4 *>(src + sbase + size_t(srow) * spitch); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        }
        if (do_write) *reinterpret_cast<u32x2 *>(dst + dbase + size_t(r) * dpitch) = u32x2{acc.x + uint32_t(r), acc.y ^ acc.z ^ acc.w};
    }
    if (!do_write && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) dst[0] = 1;
}

double run(const uint8_t *src, uint8_t *dst, int frames, int wrow, int dpitch, int spitch, int sframe, int rd, int wr, double *ms_out)
{
    const uint32_t strips = wrow / 8, nwaves = ((uint32_t(frames) * strips + 63) / 64) * 30;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3((nwaves + 3) / 4), dim3(256), 0, 0, src, dst, frames, wrow, dpitch, spitch, sframe, rd, wr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    *ms_out = best;
    const double bytes = (wr ? double(frames) * 1920 * wrow : 0) + (rd ? double(frames) * 1080 * 608 * (double(wrow) / 1080) : 0);
    return bytes / (best * 1e-3) / 1e9;
}

int main()
{
    const int frames = 300;
    uint8_t *src, *dst;
    hipMalloc(&src, size_t(frames) * 1920 * 1080 + (1 << 20)); hipMalloc(&dst, size_t(frames) * 1920 * 1152 + (1 << 20));
    hipMemset(src, 1, size_t(frames) * 1920 * 1080);
    double ms;
    const char *names[4] = {"windowed reads + 1080 writes (real)", "windowed reads + 1024 writes", "packed reads   + 1080 writes", "packed reads   + 1024 writes"};
    for (int v = 0; v < 4; ++v) {
        const int wrow = (v & 1) ? 1024 : 1080, sp = (v & 2) ? 608 : 1920, sf = (v & 2) ? 608 * 1080 : 1920 * 1080;
        const double both = run(src, dst, frames, wrow, wrow, sp, sf, 1, 1, &ms); const double msb = ms;
        const double rd = run(src, dst, frames, wrow, wrow, sp, sf, 1, 0, &ms); const double msr = ms;
        const double wr = run(src, dst, frames, wrow, wrow, sp, sf, 0, 1, &ms);
        printf("%-38s both %.0f GB/s (%.3f ms)  reads %.0f GB/s (%.3f ms)  writes %.0f GB/s (%.3f ms)\n", names[v], both, msb, rd, msr, wr, ms);
    }
    return 0;
}
