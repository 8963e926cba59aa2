// Micro-benchmark: the horizontal pass of the 1080p thumbnail chain (1920 -> 320 columns, 24-byte windows,
// 14-bit taps) two ways, on a source that stays in L2, to price the matrix-core form DESIGN.md 7 note 3 names:
//   "dot4": what column_lds_kernel's v_dot4 form does per lane and source row (7 window dwords, one v_xor and
//           two v_dot4 each), a lane per output column;
//   "mfma": 16 source rows x 16 output columns per v_mfma_i32_16x16x64_i8 -- A = the biased source bytes of 16
//           rows (lane: row l & 15, 16 consecutive bytes at 16 (l >> 4)), B = the taps' high / low digits laid out
//           as a banded 128 x 16 matrix per tile (two K steps), D: column l & 15, rows 4 (l >> 4) + reg.
// Both produce the 15-bit intermediate of hScale8To15 (sum >> 7, clipped to 32767) and are checked against the
// host, value for value, before they are timed.  (The k order inside a fragment does not matter as long as A and
// B use the same one; the row / column lane maps are the ones the guide gives for every 16x16 shape.)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_hpass tools/micro/mfma_hpass.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int W = 1920, H = 1088, OW = 320, NT = 24, NDW = 7;   // H: 68 blocks of 16 rows
constexpr int SEG = 17;                                          // row blocks a wave walks (tables stay in registers): 4 segments a plane
typedef int v4i __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// lane = output column; item = (repeat, 16-row block, 64-column block)
__global__ __launch_bounds__(256) void hpass_dot4(const uint8_t *__restrict__ src, const uint32_t *__restrict__ tab,
                                                  const int *__restrict__ pos, int16_t *__restrict__ out, int items)
{
    const int item = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (item >= items) return;
    const int cb = item % 5, rb0 = (item / 5) % (H / 16 / SEG) * SEG, rep = item / (5 * (H / 16 / SEG));
    const int c = cb * 64 + int(threadIdx.x & 63);
    const int p = pos[c], ph = p & 3;
    uint32_t th[NDW], tl[NDW];
    const uint32_t *e = tab + (size_t(c) * 4 + ph) * (2 * NDW + 1);
    const int bias = int(e[0]);
    for (int i = 0; i < NDW; ++i) { th[i] = e[1 + i]; tl[i] = e[1 + NDW + i]; }
    int16_t *o = out + size_t(rep) * H * OW;      // (every repeat its own output: repeats storing to the same lines serialise in L2)
    uint32_t cur[NDW], nxt[NDW];
    const uint8_t *base = src + (p & ~3);
    for (int i = 0; i < NDW; ++i) cur[i] = reinterpret_cast<const uint32_t *>(base + size_t(rb0 * 16) * W)[i];
    for (int r = rb0 * 16; r < (rb0 + SEG) * 16; ++r) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(base + size_t(min(r + 1, (rb0 + SEG) * 16 - 1)) * W);
#pragma unroll
        for (int i = 0; i < NDW; ++i) nxt[i] = w[i];                  // (the next row's window, requested before this row's arithmetic)
        uint32_t ws[NDW];
#pragma unroll
        for (int i = 0; i < NDW; ++i) ws[i] = cur[i] ^ 0x80808080u;
        int hi = __builtin_amdgcn_sdot4(int(ws[0]), int(th[0]), 0, true), lo = __builtin_amdgcn_sdot4(int(ws[0]), int(tl[0]), bias, true);
#pragma unroll
        for (int i = 1; i < NDW; ++i) { hi = __builtin_amdgcn_sdot4(int(ws[i]), int(th[i]), hi, false); lo = __builtin_amdgcn_sdot4(int(ws[i]), int(tl[i]), lo, false); }
        int v = ((hi << 7) + lo) >> 7;
        o[size_t(r) * OW + c] = int16_t(v < 32767 ? v : 32767);
#pragma unroll
        for (int i = 0; i < NDW; ++i) cur[i] = nxt[i];
    }
}

// wave = 16 rows x 64 columns = 4 tiles of 16 columns, 2 K steps of 64 source bytes each
__global__ __launch_bounds__(256) void hpass_mfma(const uint8_t *__restrict__ src, const v4i *__restrict__ btab,
                                                  const int *__restrict__ tile_start, const int *__restrict__ bias_tab,
                                                  int16_t *__restrict__ out, int items)
{
    const int item = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (item >= items) return;
    const int cb = item % 5, rb0 = (item / 5) % (H / 16 / SEG) * SEG, rep = item / (5 * (H / 16 / SEG));
    const int lane = int(threadIdx.x & 63), n = lane & 15, g = lane >> 4;
    int16_t *o = out + size_t(rep) * H * OW;      // (every repeat its own output: repeats storing to the same lines serialise in L2)
    v4i bh[4][2], bl[4][2];
    int s0[4], bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tile = cb * 4 + j;
        s0[j] = tile_start[tile];                                      // first source byte of the tile's span (16-byte aligned)
        bias[j] = bias_tab[tile * 16 + n];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bh[j][ks] = btab[((size_t(tile) * 2 + ks) * 2 + 0) * 64 + lane];
            bl[j][ks] = btab[((size_t(tile) * 2 + ks) * 2 + 1) * 64 + lane];
        }
    }
    // A: lane's source row, its 16 bytes of each K step; the next row block's eight loads are issued before this
    // one's arithmetic (issued where they are used, every load waited an L2 round trip of its own)
    v4i a[4][2], nx[4][2];
    auto load_block = [&](v4i (&d)[4][2], int rb) {
        const uint8_t *row = src + size_t(rb * 16 + n) * W + 16 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) d[j][ks] = *reinterpret_cast<const v4i *>(row + s0[j] + 64 * ks);
    };
    load_block(a, rb0);
    for (int rb = rb0; rb < rb0 + SEG; ++rb) {
        load_block(nx, min(rb + 1, rb0 + SEG - 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v4i hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                v4i x = a[j][ks];
                x.x ^= int(0x80808080u); x.y ^= int(0x80808080u); x.z ^= int(0x80808080u); x.w ^= int(0x80808080u);
                hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, bh[j][ks], hi, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, bl[j][ks], lo, 0, 0, 0);
            }
            const int c = (cb * 4 + j) * 16 + n;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int v = ((hi[i] << 7) + lo[i] + bias[j]) >> 7;
                o[size_t(rb * 16 + 4 * g + i) * OW + c] = int16_t(v < 32767 ? v : 32767);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j][0] = nx[j][0]; a[j][1] = nx[j][1]; }
    }
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 300;
    std::vector<uint8_t> src(size_t(W) * H + 64);
    uint32_t seed = 12345;
    auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (auto &b : src) b = uint8_t(rnd());
    // taps: a bicubic-looking kernel per column (a positive lobe, negative side lobes), made to sum to 16384
    std::vector<int> pos(OW), taps(size_t(OW) * NT);
    for (int c = 0; c < OW; ++c) {
        pos[c] = std::min(std::max(6 * c - 9, 0), W - NT);
        int sum = 0;
        for (int k = 0; k < NT; ++k) {
            const int d = std::abs(2 * k - NT + 1);
            int t = d < 12 ? 1500 - 90 * d + int(rnd() % 64) : -(int(rnd() % 200)) - 20;
            taps[size_t(c) * NT + k] = t; sum += t;
        }
        taps[size_t(c) * NT + NT / 2] += 16384 - sum;
    }
    // host result
    std::vector<int16_t> want(size_t(H) * OW);
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < OW; ++c) {
            int acc = 0;
            for (int k = 0; k < NT; ++k) acc += int(src[size_t(r) * W + pos[c] + k]) * taps[size_t(c) * NT + k];
            acc >>= 7;
            want[size_t(r) * OW + c] = int16_t(acc < 32767 ? acc : 32767);
        }
    auto digits = [](int t, int &h, int &l) { l = t & 127; h = (t - l) >> 7; };
    // dot4 tables: per column and byte phase [bias][th x 7][tl x 7]
    std::vector<uint32_t> tab(size_t(OW) * 4 * (2 * NDW + 1), 0);
    for (int c = 0; c < OW; ++c)
        for (int ph = 0; ph < 4; ++ph) {
            uint8_t hb[4 * NDW] = {0}, lb[4 * NDW] = {0};
            int sum = 0;
            for (int k = 0; k < NT; ++k) { int h, l; digits(taps[size_t(c) * NT + k], h, l); if (h < -128 || h > 127) { std::printf("tap digit out of range\n"); return 1; } hb[ph + k] = uint8_t(int8_t(h)); lb[ph + k] = uint8_t(l); sum += taps[size_t(c) * NT + k]; }
            uint32_t *e = tab.data() + (size_t(c) * 4 + ph) * (2 * NDW + 1);
            e[0] = uint32_t(128 * sum);
            for (int i = 0; i < NDW; ++i)
                for (int b = 0; b < 4; ++b) { e[1 + i] |= uint32_t(hb[4 * i + b]) << (8 * b); e[1 + NDW + i] |= uint32_t(lb[4 * i + b]) << (8 * b); }
        }
    // mfma tables: per tile of 16 columns the span start, and B[k][n] digits as the lanes hold them
    const int tiles = OW / 16;
    std::vector<int> tile_start(tiles), bias_tab(size_t(tiles) * 16);
    std::vector<uint32_t> btab(size_t(tiles) * 2 * 2 * 64 * 4, 0);
    for (int t = 0; t < tiles; ++t) {
        tile_start[t] = pos[16 * t] & ~15;
        if (pos[16 * t + 15] + NT - tile_start[t] > 128) { std::printf("tile %d: span too wide\n", t); return 1; }
        for (int n = 0; n < 16; ++n) {
            int sum = 0;
            for (int k = 0; k < NT; ++k) sum += taps[size_t(16 * t + n) * NT + k];
            bias_tab[size_t(t) * 16 + n] = 128 * sum;
        }
        for (int ks = 0; ks < 2; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 15, g = lane >> 4;
                for (int i = 0; i < 16; ++i) {
                    const int k = tile_start[t] + 64 * ks + 16 * g + i - pos[16 * t + n];    // tap index this source byte meets
                    int h = 0, l = 0;
                    if (k >= 0 && k < NT) digits(taps[size_t(16 * t + n) * NT + k], h, l);
                    btab[((((size_t(t) * 2 + ks) * 2 + 0) * 64 + lane) * 4) + i / 4] |= uint32_t(uint8_t(int8_t(h))) << (8 * (i % 4));
                    btab[((((size_t(t) * 2 + ks) * 2 + 1) * 64 + lane) * 4) + i / 4] |= uint32_t(uint8_t(l)) << (8 * (i % 4));
                }
            }
    }
    uint8_t *d_src; uint32_t *d_tab, *d_btab; int *d_pos, *d_start, *d_bias; int16_t *d_out;
    CHECK(hipMalloc(&d_src, src.size())); CHECK(hipMalloc(&d_tab, tab.size() * 4)); CHECK(hipMalloc(&d_btab, btab.size() * 4));
    CHECK(hipMalloc(&d_pos, OW * 4)); CHECK(hipMalloc(&d_start, tiles * 4)); CHECK(hipMalloc(&d_bias, bias_tab.size() * 4));
    CHECK(hipMalloc(&d_out, size_t(reps) * H * OW * 2));
    CHECK(hipMemcpy(d_src, src.data(), src.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_btab, btab.data(), btab.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_pos, pos.data(), OW * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_start, tile_start.data(), tiles * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_bias, bias_tab.data(), bias_tab.size() * 4, hipMemcpyHostToDevice));
    std::vector<int16_t> got(size_t(H) * OW);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        const char *name = which ? "mfma" : "dot4";
        auto launch = [&](int r) {
            const int items = r * (H / 16 / SEG) * 5;
            if (which) hpass_mfma<<<(items + 3) / 4, 256>>>(d_src, reinterpret_cast<const v4i *>(d_btab), d_start, d_bias, d_out, items);
            else hpass_dot4<<<(items + 3) / 4, 256>>>(d_src, d_tab, d_pos, d_out, items);
        };
        CHECK(hipMemset(d_out, 0, size_t(reps) * H * OW * 2));
        launch(1);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), d_out, got.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
        std::printf("%s: %zu of %zu values differ from the host\n", name, bad, got.size());
        for (int i = 0; i < 5; ++i) launch(reps);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch(reps);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%s: %.1f us per %d planes of %dx%d -> %d columns (%.2f us a plane; the whole 1080p luma plane is %d rows)\n",
                    name, ms * 100, reps, W, H, OW, ms * 100 / reps, H);
    }
    return 0;
}
