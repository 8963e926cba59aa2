// Micro-benchmark: what HBM delivers for the strip kernel's access pattern, without any arithmetic.
//   mode 0: strip pattern writes  (each wave writes 512 B per output row, rows 1080 B apart, 64 rows)
//   mode 1: linear writes         (same bytes, each wave writes 32 KiB contiguous)
//   mode 2: strip pattern reads   (each wave reads 1 KiB-per-lane-16B run per source row, rows 1920 B apart, 39 rows)
//   mode 3: strip reads + strip writes together (the kernel's real mix)
//   mode 4: linear reads + linear writes (a copy of the same byte counts)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int OW = 1080, OH = 1920, IW = 1920, IH = 1080, SEG = 64, NSEG = OH / SEG, STRIPS = OW / 8;   // luma only

__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int frames, int mode, uint32_t *sink)
{
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t nwaves_row = (uint32_t(frames) * STRIPS + 63) / 64;
    const uint32_t seg = wave % NSEG, wr = wave / NSEG;
    if (wr >= nwaves_row) return;
    const uint32_t g = min(wr * 64 + lane, uint32_t(frames) * STRIPS - 1);
    const uint32_t frame = g / STRIPS, strip = g % STRIPS;
    uint32_t acc = 0;
    if (mode == 0 || mode == 3 || mode == 2) {
        const size_t dbase = size_t(frame) * OW * OH + size_t(seg * SEG) * OW + strip * 8;
        const size_t sbase = size_t(frame) * IW * IH + size_t(seg * 36) * IW + ((654 + strip * 9 / 2) & ~15u) + 0;
        for (int r = 0; r < SEG; ++r) {
            if ((mode == 2 || mode == 3) && r < 39) {
                // 16 B per lane, contiguous over the wave (like the LDS-DMA run; only ~300 B are needed)
                const u32x4 v = *reinterpret_cast<const u32x4 *>(src + sbase + size_t(r) * IW + (lane < 20 ? lane : 19) * 16);
                acc += v.x + v.y + v.z + v.w;
            }
            if (mode == 0 || mode == 3) {
                u32x2 o = {acc + r, acc ^ lane};
                *reinterpret_cast<u32x2 *>(dst + dbase + size_t(r) * OW) = o;
            }
        }
    } else {
        // linear: wave w owns 64 rows * 512 B = 32 KiB contiguous of dst, and 39 * 320 B ~ 12.5 KiB of src
        const size_t dbase = size_t(wave) * SEG * 512 + lane * 8;
        const size_t sbase = size_t(wave) * 39 * 320;
        for (int r = 0; r < SEG; ++r) {
            if (mode == 4 && r < 39) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(src + sbase + size_t(r) * 320 + (lane < 20 ? lane : 19) * 16);
                acc += v.x + v.y + v.z + v.w;
            }
            u32x2 o = {acc + r, acc ^ lane};
            *reinterpret_cast<u32x2 *>(dst + dbase + size_t(r) * 512) = o;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 300;
    uint8_t *src, *dst; uint32_t *sink;
    const size_t sb = size_t(frames) * IW * IH * 3 / 2, db = size_t(frames) * OW * OH * 3 / 2;
    hipMalloc(&src, sb); hipMalloc(&dst, db); hipMalloc(&sink, 4);
    hipMemset(src, 1, sb); hipMemset(dst, 0, db);
    const uint32_t nwaves = ((uint32_t(frames) * STRIPS + 63) / 64) * NSEG;
    const dim3 grid((nwaves + 3) / 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"strip writes", "linear writes", "strip reads", "strip reads+writes", "linear reads+writes"};
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, src, dst, frames, mode, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it && ms < best) best = ms;
        }
        const double wbytes = (mode == 2) ? 0 : double(nwaves) * SEG * 512;
        const double rbytes = (mode >= 2) ? double(nwaves) * 39 * 320 : 0;
        printf("%-22s %.4f ms   write %.0f MB read %.0f MB   %.0f GB/s\n", names[mode], best, wbytes / 1e6, rbytes / 1e6,
               (wbytes + rbytes) / (best * 1e-3) / 1e9);
    }
    return 0;
}
