// ttscale_rgbwalk.hip -- packed-RGB output of a scale with 5 to 8 taps an axis (shrinks between 1x and 2x: the web
// `compress` box as pixels, web/src/tests/video/portrait.test.js:70-97 + `format=rgb24`): ONE walk that filters luma
// and chroma side by side and writes every pixel once.
//
// Before this kernel such chains took three launches (luma strips with a 16-bit store, chroma strips, an in-place
// conversion that moved the frame twice more: DESIGN.md 7 finding 6).  Here a lane owns 8 adjacent output pixels of
// one frame -- 8 luma columns and the 4 chroma columns under them, U and V -- and walks down a segment of output rows:
//   * source rows arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`), one coalesced run per plane and row for the
//     whole wave, D luma rows in flight; the lanes pick their windows out of the wave's own LDS slots as aligned
//     dwords -- 12-byte windows shared by two columns (two columns x ratio + 8 taps <= 12 bytes up to 2x);
//   * hScale8To15 per column: 4 v_perm + 4 v_dot2 (the second half of the taps sits one dword on: same selectors);
//   * luma results enter a register ring of 8 rows by SLOT (source row - first row of the segment) mod 8 -- the walk
//     is unrolled by 8 source rows, so the slot is known at compile time and the vertical taps come from the host
//     arranged by slot; chroma, which swscale's RGB writers interpolate to EVERY output row from half as many source
//     rows (an enlargement: <= 4 taps), enters 4-row shift registers when an output row asks for new rows;
//   * an output row leaves when its last luma source row has entered: yuv2rgb_X_c_template's unclipped sums, the
//     closed-form table look-up of ttscale_rgb.hip, 24 (32) bytes a lane through the whole-sector store exchange.
// Packed batches only (the staged runs may round into bytes around a plane); per-frame pointers keep the in-place
// scheme.  The arithmetic is the general kernel's (rgb_generic_kernel) in another order of evaluation: same bytes.
#include "ttscale_rgb.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <tuple>
#include <vector>

#include "filter.h"

namespace {

using tts::RgbCoef;

struct RgbWalkArgs {
    uint32_t hoffl, sell, coefl;       // luma tables (dword offsets into tab): [strips][4] window starts, [strips][8] selectors, [strips][8][4] tap pairs
    uint32_t hoffc, selc, coefc;       // chroma: [strips][2], [strips][4], [strips][4][4]
    uint32_t vrow;                     // [ph][8]: last luma source row, new chroma rows, luma taps by ring slot (4 dwords), chroma taps (2 dwords)
    uint32_t seginfo;                  // [nseg][2]: first luma source row of the segment, first chroma source row of its ring fill (may be < 0: clamped, zero taps)
    int32_t strips, n_items, nseg, seg_rows, ph, pw;
    int32_t src_h_l, src_h_c;          // rows of the (pre-cropped) source planes
    uint32_t end_rel[2];               // luma / chroma: bytes from a plane's (pre-cropped) origin to one past its last byte
    uint32_t row_rel[2];               // ... to the end of its row
    int32_t src_stride_y, src_stride_c, dst_stride;
    int32_t swap;
    uint64_t src_span, dst_span;
    RgbCoef coef;
    FrameAddr addr;
};

typedef __attribute__((address_space(3))) void walk_lds_t;
constexpr int kWalkMaxPieces = 8;
[[maybe_unused]] constexpr int kWalkCSlot = 1024 + 16;

// wait until at most `outstanding` vector-memory operations are in flight (a smaller count is always safe)
[[maybe_unused]] __device__ __forceinline__ void walk_wait_vm(int outstanding)
{
    if (outstanding >= 8) {
        if (outstanding >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (outstanding >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        return;
    }
    switch (outstanding) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    }
}

// four 12-byte windows at any byte phase: the four aligned dwords that hold each, and the wait, in ONE asm statement
// (no output register exists, as far as the compiler can tell, before its data has landed)
[[maybe_unused]] __device__ __forceinline__ void walk_lds_read4(uint32_t (&w)[4][4], const uint32_t (&p)[4])
{
    u32x2_t t[4][2];
    asm volatile("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %8 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %2, %9 offset1:1\n\tds_read2_b32 %3, %9 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %4, %10 offset1:1\n\tds_read2_b32 %5, %10 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %6, %11 offset1:1\n\tds_read2_b32 %7, %11 offset0:2 offset1:3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[1][0]), "=&v"(t[1][1]), "=&v"(t[2][0]), "=&v"(t[2][1]), "=&v"(t[3][0]), "=&v"(t[3][1])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[j][0] = t[j][0].x; w[j][1] = t[j][0].y; w[j][2] = t[j][1].x; w[j][3] = t[j][1].y; }
}

// The source runs of one plane for this wave (see rgb_plan_run in ttscale_rgb.hip and plan_runs in ttscale_kernels.h):
// one piece per frame row its 64 lanes touch, each rounded out to 16-byte chunks, packed one after another into the
// wave's slot of 64 * NDMA chunks; chunk c is fetched by lane c % 64 of DMA c / 64.  A run that would pass the plane's
// end on its last row is slid down to end exactly on the piece's last byte.  Packed batches only.
template <int NDMA>
__device__ __forceinline__ void walk_plan_run(uint32_t first, uint32_t last, uint32_t soff, uint32_t end_rel, uint32_t last_rowoff,
                                              bool new_row, int lane, uint32_t (&csrc)[NDMA], uint32_t &lbase)
{
    uint64_t pm = __builtin_amdgcn_ballot_w64(new_row) | 1ull;
    uint32_t prefix = 0;
    lbase = 0;
#pragma unroll
    for (int d = 0; d < NDMA; ++d) csrc[d] = ~0u;
    for (int p = 0; p < kWalkMaxPieces && pm; ++p) {
        const int l0 = __builtin_ctzll(pm);
        pm &= pm - 1;
        const int l1 = pm ? __builtin_ctzll(pm) : 64;
        const uint32_t hi_abs = uint32_t(__builtin_amdgcn_readlane(int(soff), l0)) + end_rel;
        const uint32_t end = uint32_t(__builtin_amdgcn_readlane(int(last), l1 - 1));
        uint32_t start = uint32_t(__builtin_amdgcn_readlane(int(first), l0)) & ~15u;
        const uint32_t n = (end - start + 15u) >> 4;
        if (start + 16u * n + last_rowoff > hi_abs && end >= 16u * n) start = end - 16u * n;
        if (lane >= l0 && lane < l1) lbase = 16u * prefix - start;
        if (p == 0) {                               // spare lanes re-fetch the first piece's first chunk
#pragma unroll
            for (int d = 0; d < NDMA; ++d) csrc[d] = start;
        }
#pragma unroll
        for (int d = 0; d < NDMA; ++d) {
            const uint32_t ch = uint32_t(lane) + 64u * d;
            if (ch >= prefix && ch < prefix + n) csrc[d] = start + 16u * (ch - prefix);
        }
        prefix += n;
    }
    // the host picks this kernel only for geometries whose runs fit; anything else here would be silent corruption
    if (pm || prefix > 64u * NDMA) __builtin_trap();
}

[[maybe_unused]] __device__ __forceinline__ int walk_clip255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }   // (v_med3_i32)

// whole-sector stores: see rgb_lds_exchange3 / 4 in ttscale_rgb.hip
[[maybe_unused]] __device__ __forceinline__ void walk_exchange3(uint32_t waddr, const uint32_t (&o)[6], uint32_t r1, uint32_t r2, u32x4_t &v1, u32x4_t &v2)
{
    asm volatile("ds_write_b64 %2, %3\n\tds_write_b64 %2, %4 offset:8\n\tds_write_b64 %2, %5 offset:16\n\t"
                 "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v1), "=&v"(v2)
                 : "v"(waddr), "v"(u32x2_t{o[0], o[1]}), "v"(u32x2_t{o[2], o[3]}), "v"(u32x2_t{o[4], o[5]}), "v"(r1), "v"(r2) : "memory");
}
[[maybe_unused]] __device__ __forceinline__ void walk_exchange4(uint32_t waddr, const uint32_t (&o)[8], uint32_t r1, uint32_t r2, u32x4_t &v1, u32x4_t &v2)
{
    asm volatile("ds_write_b128 %2, %3\n\tds_write_b128 %2, %4 offset:16\n\t"
                 "ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v1), "=&v"(v2)
                 : "v"(waddr), "v"(u32x4_t{o[0], o[1], o[2], o[3]}), "v"(u32x4_t{o[4], o[5], o[6], o[7]}), "v"(r1), "v"(r2) : "memory");
}

// BPP: 3 / 4 bytes a pixel; NL: 1 KiB LDS-DMAs per luma source row (a wave's luma runs hold <= 64 NL chunks);
// TP: tap pairs per column and ring row pairs (3: every filter but the vertical chroma one has <= 6 taps -- shrinks up
// to 1.5x: the web box -- 4: <= 8 taps); XP: whole-sector stores through the LDS exchange.  D = 2 luma source rows are in
// flight per wave, one LDS slot each (one: 7 % slower; four: 2 % faster at 8 more registers).
template <int BPP, int NL, int TP, bool XP>
__global__ __launch_bounds__(kBlock, TP == 3 ? 3 : 2) void rgb_walk_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                             const uint32_t *__restrict__ tab, const RgbWalkArgs a)
{
    static_assert(BPP == 3 || BPP == 4, "rgb24 / bgr24 or rgba / bgra");
    static_assert((NL == 1 || NL == 2) && (TP == 3 || TP == 4), "slots rotate with the ring of 2 TP rows");
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin exists in the device pass only
    constexpr int RL = 2 * TP, D = 2;
    constexpr int kLSlot = NL * 1024 + 16;
    __shared__ __attribute__((aligned(16))) uint8_t ldsl[kBlock / 64][D][kLSlot];
    __shared__ __attribute__((aligned(16))) uint8_t ldsc[kBlock / 64][2][kWalkCSlot];
    __shared__ __attribute__((aligned(16))) uint8_t xlds[XP ? kBlock / 64 : 1][XP ? 64 * 8 * BPP : 16];   // the waves' output rows in exchange
    const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = int(lb % uint32_t(a.nseg));
    const int sb  = int(lb / uint32_t(a.nseg));
    int g = sb * kBlock + int(threadIdx.x);
    const bool live = g < a.n_items;
    if (!live) g = a.n_items - 1;                  // dead lanes (a suffix of the last workgroup) shadow the last live one
    const int strip = g % a.strips;
    const int frame = g / a.strips;
    const int lane = int(threadIdx.x) & 63;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);

    // this wave's bases and the lane's 32-bit offsets from them (lanes are ordered by frame: the wave addresses from
    // its first frame; the host checks that the frames 64 lanes can touch lie within kMaxSpan)
    const uint32_t f0 = uni(uint32_t(frame));
    const uint64_t sbase = uint64_t(f0) * a.addr.src_pitch, dbase = uint64_t(f0) * a.addr.dst_pitch;
    const uint8_t *wsrc = src + sbase;
    uint8_t *wdst = dst + dbase;
    const uint32_t src_rec = a.src_span > sbase ? uint32_t(min(a.src_span - sbase, uint64_t(kMaxSpan))) : 0u;
    const uint32_t dst_rec = a.dst_span > dbase ? uint32_t(min(a.dst_span - dbase, uint64_t(kMaxSpan))) : 0u;
    uint32_t so[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) so[k] = (uint32_t(frame) - f0) * a.addr.src_pitch + a.addr.src_plane[k];
    uint32_t doff = (uint32_t(frame) - f0) * a.addr.dst_pitch + uint32_t(min(strip * 8, a.pw - 8)) * BPP;   // the last strip is slid left, not cut short
    if (!live) doff = kDiscard;                               // the descriptor's range check drops these stores
    // XP: where this lane's two chunks go (row offsets come as the stores' scalar offset), the held-back chunk
    [[maybe_unused]] uint32_t dest1 = kDiscard, dest2 = kDiscard, dest_a = kDiscard, dest_b = kDiscard, xw = 0, xr1 = 0, xr2 = 0;
    [[maybe_unused]] bool split_second = false, any_split = false;
    if constexpr (XP) {
        constexpr int kChunks = 64 * 8 * BPP / 16;             // 96 (rgb24) or 128 (rgba)
        const uint32_t xbase = uint32_t(reinterpret_cast<uintptr_t>((walk_lds_t *)&xlds[wave][0]));
        xw = xbase + uint32_t(lane) * (8u * BPP);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = lane + 64 * q;
            const int cc = c < kChunks ? c : 0;                 // (lanes without a second chunk read chunk 0 and store nowhere)
            const int L = 16 * cc / (8 * BPP), inner = 16 * cc - 8 * BPP * L;
            uint32_t d = uint32_t(__builtin_amdgcn_ds_bpermute(L * 4, int(doff))) + uint32_t(inner);
            if constexpr (BPP == 3) {
                const uint32_t d0 = d - uint32_t(inner), dn = uint32_t(__builtin_amdgcn_ds_bpermute(min(L + 1, 63) * 4, int(doff)));
                if (c < kChunks && inner == 16 && dn != d0 + 24u) { dest_a = d; dest_b = dn; split_second = q == 1; d = kDiscard; }
            }
            if (c >= kChunks) d = kDiscard;
            (q ? dest2 : dest1) = d;
            (q ? xr2 : xr1) = xbase + 16u * uint32_t(cc);
        }
        any_split = __builtin_amdgcn_ballot_w64(dest_a != kDiscard || dest_b != kDiscard) != 0;
    }

    // per-lane horizontal tables stay in registers for the whole walk
    uint32_t offl[4], offu[2], offv[2], sell[8], coefl[8][TP], selc[4], coefc[4][TP];
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) offl[j] = tab[a.hoffl + strip * 4 + j] + so[0];
#pragma unroll
        for (int j = 0; j < 2; ++j) { const uint32_t oc = tab[a.hoffc + strip * 2 + j]; offu[j] = oc + so[1]; offv[j] = oc + so[2]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sell[i] = tab[a.sell + strip * 8 + i];
#pragma unroll
            for (int k = 0; k < TP; ++k) coefl[i][k] = tab[a.coefl + (strip * 8 + i) * 4 + k];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            selc[i] = tab[a.selc + strip * 4 + i];
#pragma unroll
            for (int k = 0; k < TP; ++k) coefc[i][k] = tab[a.coefc + (strip * 4 + i) * 4 + k];
        }
    }

    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(wsrc), 0, src_rec, 0x00020000);
    const uint64_t dbase64 = reinterpret_cast<uint64_t>(wdst);
    const u32x4_t drs = {uni(uint32_t(dbase64)), uni(uint32_t(dbase64 >> 32) & 0xffffu), uni(dst_rec), 0x00020000u};

    const int lmax = a.src_h_l - 1, cmax = a.src_h_c - 1;
    const int sy = int(uni(uint32_t(a.src_stride_y))), sc = int(uni(uint32_t(a.src_stride_c))), ds = int(uni(uint32_t(a.dst_stride)));
    const int rstart = int(uni(tab[a.seginfo + 2 * seg]));
    int crow = int(uni(tab[a.seginfo + 2 * seg + 1]));

    // what this lane fetches in each plane's DMA, and where its windows lie in the slots
    uint32_t csl[NL], csu[1], csv[1], lposl[4], lposc[4];
    {
        const bool new_row = lane > 0 && strip == 0 && live;
        uint32_t lbl, lbu, lbv;
        // (a 12-byte window may hang over its row's end by bytes its taps do not use: the run stops at the row's end)
        walk_plan_run<NL>(offl[0], min(offl[3] + 12u, so[0] + a.row_rel[0]), so[0], a.end_rel[0], uint32_t(lmax) * uint32_t(sy), new_row, lane, csl, lbl);
        walk_plan_run<1>(offu[0], min(offu[1] + 12u, so[1] + a.row_rel[1]), so[1], a.end_rel[1], uint32_t(cmax) * uint32_t(sc), new_row, lane, csu, lbu);
        walk_plan_run<1>(offv[0], min(offv[1] + 12u, so[2] + a.row_rel[1]), so[2], a.end_rel[1], uint32_t(cmax) * uint32_t(sc), new_row, lane, csv, lbv);
        const uint32_t sl = uint32_t(reinterpret_cast<uintptr_t>((walk_lds_t *)&ldsl[wave][0][0]));
        const uint32_t su = uint32_t(reinterpret_cast<uintptr_t>((walk_lds_t *)&ldsc[wave][0][0])), sv = uint32_t(reinterpret_cast<uintptr_t>((walk_lds_t *)&ldsc[wave][1][0]));
#pragma unroll
        for (int j = 0; j < 4; ++j) lposl[j] = offl[j] + lbl + sl;
        lposc[0] = offu[0] + lbu + su; lposc[1] = offu[1] + lbu + su; lposc[2] = offv[0] + lbv + sv; lposc[3] = offv[1] + lbv + sv;
    }

    // vector-memory operations issued after the DMA of each luma slot / after the pending chroma DMAs (the counter
    // retires in issue order: before a slot is read, wait until only those are outstanding)
    int after[D], since_c;
    auto issued = [&](int n) {
#pragma unroll
        for (int d = 0; d < D; ++d) after[d] += n;
        since_c += n;
    };
    auto chroma_dma = [&](int row) {
        const int co = min(max(row, 0), cmax) * sc;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (walk_lds_t *)&ldsc[wave][0][0], 16, csu[0], co, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (walk_lds_t *)&ldsc[wave][1][0], 16, csv[0], co, 0, 0);
        issued(2);
        since_c = 0;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) after[d] = 0;
    since_c = 0;
    chroma_dma(crow);
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int q = 0; q < NL; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (walk_lds_t *)&ldsl[wave][d][q * 1024], 16, csl[q], min(rstart + d, lmax) * sy, 0, 0);
        issued(NL);
        after[d] = 0;
    }

    uint32_t ringl[8][RL / 2], ru[4][2], rv[4][2];       // luma by slot; chroma shift registers (oldest, .) (., newest)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < RL / 2; ++k) ringl[i][k] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) ru[i][0] = ru[i][1] = rv[i][0] = rv[i][1] = 0;

    // the closed form's constants (wave-uniform)
    const int cy = a.coef.cy, crv = a.coef.crv, cbu = a.coef.cbu, cgu = a.coef.cgu, cgv = a.coef.cgv;
    const int kr = a.coef.off_r * cy + a.coef.k, kg = a.coef.off_g * cy + a.coef.k, kb = a.coef.off_b * cy + a.coef.k;
    const bool swap = a.swap != 0;

    const int y0 = seg * a.seg_rows, y1 = min(y0 + a.seg_rows, a.ph);
    int y = y0;
    // The per-row table entries are read through the constant address space: wave-uniform reads of it are scalar loads
    // whatever else the kernel does (as plain global loads they became vector loads behind a full `s_waitcnt vmcnt(0)`
    // -- every row in flight drained -- and a v_readfirstlane: the asm stores and LDS-DMAs clobber "memory", so the
    // compiler cannot tell that the table never changes).  The entry of row y + 1 is requested while row y is worked on.
    typedef const __attribute__((address_space(4))) uint32_t *ctab_t;
    ctab_t vr = (ctab_t)(tab + a.vrow) + 8 * size_t(y);
    const int last_y = a.ph - 1;
    uint32_t cur[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[k] = vr[k];
    int next_last = int(cur[0]);

    for (int r0 = rstart; y < y1 && r0 <= lmax; r0 += RL) {
#pragma unroll
        for (int s = 0; s < RL; ++s) {
            const int r = r0 + s;
            if (y >= y1) continue;                               // wave-uniform (no early exit: the loop must unroll fully)
            // ---- luma source row r: out of slot s % D, which is then refilled with row r + D
            walk_wait_vm(after[s % D]);
            uint32_t wl[4][4];
            {
                uint32_t p[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = (lposl[j] & ~3u) + uint32_t(s % D) * uint32_t(kLSlot);
                walk_lds_read4(wl, p);
            }
#pragma unroll
            for (int q = 0; q < NL; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (walk_lds_t *)&ldsl[wave][s % D][q * 1024], 16, csl[q], min(r + D, lmax) * sy, 0, 0);
            issued(NL);
            after[s % D] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) wl[j][q] = __builtin_amdgcn_alignbyte(wl[j][q + 1], wl[j][q], lposl[j]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t *w = wl[i / 2];
                const uint32_t s0 = sell[i], s1 = sell[i] + 0x00020002u;
                int acc = dot2(__builtin_amdgcn_perm(w[1], w[0], s0), coefl[i][0], 0);
                acc = dot2(__builtin_amdgcn_perm(w[1], w[0], s1), coefl[i][1], acc);
                acc = dot2(__builtin_amdgcn_perm(w[2], w[1], s0), coefl[i][2], acc);      // taps 4..7 sit four bytes on: the same selectors, one dword further
                if constexpr (TP == 4) acc = dot2(__builtin_amdgcn_perm(w[2], w[1], s1), coefl[i][3], acc);
                const uint32_t t = hfinish(acc);
                ringl[i][s / 2] = (s & 1) ? __builtin_amdgcn_perm(t, ringl[i][s / 2], 0x05040100)
                                          : __builtin_amdgcn_perm(ringl[i][s / 2], t, 0x07060100);
            }
            // ---- output rows whose last luma source row this was
            while (next_last == r) {                             // wave-uniform
                const int nc = y == y0 ? 4 : int(cur[1]);
                const uint32_t vl0 = cur[2], vl1 = cur[3], vl2 = cur[4], vc0 = cur[6], vc1 = cur[7];
                [[maybe_unused]] const uint32_t vl3 = cur[5];
                vr = (ctab_t)(tab + a.vrow) + 8 * size_t(min(y + 1, last_y));
#pragma unroll
                for (int k = 0; k < 8; ++k) cur[k] = vr[k];
                for (int n = 0; n < nc; ++n) {                   // wave-uniform: new chroma source rows first
                    walk_wait_vm(since_c);
                    uint32_t wc[4][4];
                    {
                        uint32_t p[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) p[j] = lposc[j] & ~3u;
                        walk_lds_read4(wc, p);
                    }
                    chroma_dma(crow + 1);                        // the slots are free again (their bytes sit in registers)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int q = 0; q < 3; ++q) wc[j][q] = __builtin_amdgcn_alignbyte(wc[j][q + 1], wc[j][q], lposc[j]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t *wu = wc[i / 2], *wv = wc[2 + i / 2];
                        const uint32_t s0 = selc[i], s1 = selc[i] + 0x00020002u;
                        int au = dot2(__builtin_amdgcn_perm(wu[1], wu[0], s0), coefc[i][0], 0);
                        au = dot2(__builtin_amdgcn_perm(wu[1], wu[0], s1), coefc[i][1], au);
                        au = dot2(__builtin_amdgcn_perm(wu[2], wu[1], s0), coefc[i][2], au);
                        if constexpr (TP == 4) au = dot2(__builtin_amdgcn_perm(wu[2], wu[1], s1), coefc[i][3], au);
                        int av = dot2(__builtin_amdgcn_perm(wv[1], wv[0], s0), coefc[i][0], 0);
                        av = dot2(__builtin_amdgcn_perm(wv[1], wv[0], s1), coefc[i][1], av);
                        av = dot2(__builtin_amdgcn_perm(wv[2], wv[1], s0), coefc[i][2], av);
                        if constexpr (TP == 4) av = dot2(__builtin_amdgcn_perm(wv[2], wv[1], s1), coefc[i][3], av);
                        const uint32_t tu = hfinish(au), tv = hfinish(av);
                        ru[i][0] = __builtin_amdgcn_alignbit(ru[i][1], ru[i][0], 16);
                        ru[i][1] = __builtin_amdgcn_alignbit(tu, ru[i][1], 16);
                        rv[i][0] = __builtin_amdgcn_alignbit(rv[i][1], rv[i][0], 16);
                        rv[i][1] = __builtin_amdgcn_alignbit(tv, rv[i][1], 16);
                    }
                    ++crow;
                }
                // vertical pass: yuv2rgb_X_c_template's 1 << 18 rounding, >> 19, no clip
                int P[8], CR[4], CG[4], CB[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int t = dot2(ringl[i][2], vl2, dot2(ringl[i][1], vl1, dot2(ringl[i][0], vl0, 1 << 18)));
                    if constexpr (TP == 4) t = dot2(ringl[i][3], vl3, t);
                    P[i] = t >> 19;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int U = walk_clip255(dot2(ru[j][1], vc1, dot2(ru[j][0], vc0, 1 << 18)) >> 19);
                    const int V = walk_clip255(dot2(rv[j][1], vc1, dot2(rv[j][0], vc0, 1 << 18)) >> 19);
                    const int cr = __mul24(__mul24(V, crv) >> 16, cy) + kr;
                    const int cb = __mul24(__mul24(U, cbu) >> 16, cy) + kb;
                    CG[j] = __mul24((__mul24(U, cgu) >> 16) + (__mul24(V, cgv) >> 16), cy) + kg;
                    CR[j] = swap ? cb : cr;                      // (bgr: the first byte of a pixel is B)
                    CB[j] = swap ? cr : cb;
                }
                const int yoff = y * ds;
                int stores = 2;
                if constexpr (BPP == 3) {
                    int v[24];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { v[3 * i] = __mul24(P[i], cy) + CR[i / 2]; v[3 * i + 1] = __mul24(P[i], cy) + CG[i / 2]; v[3 * i + 2] = __mul24(P[i], cy) + CB[i / 2]; }   // (v_mad_i32_i24: the product rides in the add)
                    uint32_t o[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) o[q] = pack4<16>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);      // (>> 16, clip, pack: two instructions)
                    if constexpr (XP) {
                        u32x4_t v1, v2;
                        walk_exchange3(xw, o, xr1, xr2, v1, v2);
                        store_b128(v1, drs, dest1, yoff);
                        store_b128(v2, drs, dest2, yoff);
                        if (any_split) {                         // wave-uniform: the one chunk whose halves lie apart
                            const u32x4_t h = split_second ? v2 : v1;
                            store_b64(u32x2_t{h.x, h.y}, drs, dest_a, yoff);
                            store_b64(u32x2_t{h.z, h.w}, drs, dest_b, yoff);
                            stores = 4;
                        }
                    } else {
                        store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, drs, doff, yoff);
                        store_b64(u32x2_t{o[4], o[5]}, drs, doff + 16u, yoff);
                    }
                } else {
                    uint32_t o[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        o[i] = pack4<16>(__mul24(P[i], cy) + CR[i / 2], __mul24(P[i], cy) + CG[i / 2], __mul24(P[i], cy) + CB[i / 2], 0x7fff0000);   // alpha 255
                    if constexpr (XP) {
                        u32x4_t v1, v2;
                        walk_exchange4(xw, o, xr1, xr2, v1, v2);
                        store_b128(v1, drs, dest1, yoff);
                        store_b128(v2, drs, dest2, yoff);
                    } else {
                        store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, drs, doff, yoff);
                        store_b128(u32x4_t{o[4], o[5], o[6], o[7]}, drs, doff + 16u, yoff);
                    }
                }
                issued(stores);
                ++y;
                next_last = y < y1 ? int(cur[0]) : -1;
            }
        }
    }
#endif
}

template <class... P, class... A>
int launch_walk(const char *what, void (*fn)(P...), dim3 grid, dim3 block, hipStream_t st, A &&...args)
{
    std::tuple<P...> held(std::forward<A>(args)...);
    void *ptrs[sizeof...(P)];
    size_t i = 0;
    std::apply([&](auto &...v) { ((ptrs[i++] = const_cast<void *>(static_cast<const void *>(&v))), ...); }, held);
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(fn), grid, block, ptrs, 0, st);
    if (e == hipSuccess) return TTS_OK;
    tts::set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    return TTS_ERR_PROCESSING_FAILED;
}

typedef void (*rgb_walk_fn)(const uint8_t *, uint8_t *, const uint32_t *, const RgbWalkArgs);

template <int BPP, int NL, int TP>
rgb_walk_fn walk_pick_xp(bool xp) { return xp ? rgb_walk_kernel<BPP, NL, TP, true> : rgb_walk_kernel<BPP, NL, TP, false>; }
template <int BPP, int NL>
rgb_walk_fn walk_pick_tp(int tp, bool xp) { return tp == 3 ? walk_pick_xp<BPP, NL, 3>(xp) : walk_pick_xp<BPP, NL, 4>(xp); }
rgb_walk_fn walk_pick(int bpp, int nl, int tp, bool xp)
{
    if (bpp == 3) return nl == 2 ? walk_pick_tp<3, 2>(tp, xp) : walk_pick_tp<3, 1>(tp, xp);
    return nl == 2 ? walk_pick_tp<4, 2>(tp, xp) : walk_pick_tp<4, 1>(tp, xp);
}

}  // namespace

namespace tts {

struct RgbWalkPlan {
    tts_geom g{};
    int n_cus = 256, bpp = 3;
    int nl = 1, tp = 4;                // luma DMAs a row; tap pairs per column = ring row pairs (3: every filter <= 6 taps)
    bool no_xp = false;
    uint32_t *d_tab = nullptr;
    RgbWalkArgs args{};                // table offsets and constants; per-launch fields are filled in by rgb_walk_launch
    struct Seg { int seg_rows, nseg; uint32_t seginfo, vrow; };
    std::vector<Seg> segs;
    int occ = 2;                       // resident workgroups per CU (asked at plan creation, on the plan's device)
    char name[64] = "";
};

RgbWalkPlan *rgb_walk_create(const tts_geom &g, const AxisFilter &lh, const AxisFilter &lv, const AxisFilter &ch, const AxisFilter &cv,
                             int bpp, int swap, const RgbCoef &coef, int n_cus)
{
    auto env = [](const char *k) { const char *v = std::getenv(k); return v ? std::atoi(v) : 0; };
    if (env("TTS_FORCE_GENERIC") || env("TTS_NO_RGB_WALK") || env("TTS_NO_LDS")) return nullptr;
    if (g.rgb_stage != TTS_RGB_AT_SCALE || g.do_pad || g.out_range != TTS_RANGE_KEEP) return nullptr;
    if (lh.size > 8 || ch.size > 8 || lv.size > 8 || cv.size > 4) return nullptr;
    if (lv.size <= 2 && cv.size <= 2) return nullptr;                                // rows of swscale's 1- / 2-line writers: the general kernel
    if ((g.crop_x & 1) || (g.crop_w & 1) || g.crop_w < 16) return nullptr;           // pixel pairs must not straddle a strip
    if (lh.src < 16 || ch.src < 16) return nullptr;
    auto monotone = [](const std::vector<int32_t> &v) { for (size_t i = 1; i < v.size(); ++i) if (v[i] < v[i - 1]) return false; return true; };
    if (!monotone(lh.pos) || !monotone(ch.pos) || !monotone(lv.pos) || !monotone(cv.pos)) return nullptr;
    const int pw = g.crop_w, ph = g.crop_h, strips = (pw + 7) / 8;
    if (g.crop_x + pw > lh.dst || (g.crop_x >> 1) + pw / 2 > ch.dst || g.crop_y + ph > lv.dst || g.crop_y + ph > cv.dst) return nullptr;

    RgbWalkPlan *p = new (std::nothrow) RgbWalkPlan();
    if (!p) return nullptr;
    p->g = g; p->n_cus = n_cus; p->bpp = bpp;
    std::vector<uint32_t> tab;
    RgbWalkArgs &a = p->args;
    auto pack16 = [](int16_t lo, int16_t hi) { return uint32_t(uint16_t(lo)) | (uint32_t(uint16_t(hi)) << 16); };
    // one 12-byte window per 2 columns; a column's taps start d <= 4 bytes into it, and the selector picks bytes d, d + 1
    // (taps 2, 3: the selector + 2 in both halves; taps 4..7: the same two selectors one dword on)
    auto group = [&](const AxisFilter &f, int first_col, uint32_t &hoff_out, uint32_t *sel, uint32_t *coef) {
        const int glo = f.pos[first_col];
        for (int j = 0; j < 2; ++j) {
            const int col = std::min(first_col + j, f.dst - 1);
            const int d = f.pos[col] - glo;
            if (d < 0 || d > 4) return false;
            int16_t t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < f.size; ++k) t8[k] = f.taps[size_t(col) * f.size + k];
            sel[j] = 0x0c000c00u | uint32_t(d) | (uint32_t(d + 1) << 16);
            for (int k = 0; k < 4; ++k) coef[4 * j + k] = pack16(t8[2 * k], t8[2 * k + 1]);
        }
        hoff_out = uint32_t(glo);
        return true;
    };
    a.hoffl = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips) * 4);
    a.sell = uint32_t(tab.size());  tab.resize(tab.size() + size_t(strips) * 8);
    a.coefl = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips) * 32);
    a.hoffc = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips) * 2);
    a.selc = uint32_t(tab.size());  tab.resize(tab.size() + size_t(strips) * 4);
    a.coefc = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips) * 16);
    bool ok = true;
    for (int s0 = 0; s0 < strips && ok; ++s0) {
        const int vx = g.crop_x + std::min(s0 * 8, pw - 8);                      // even: crop_x and pw are
        for (int q = 0; q < 4 && ok; ++q)
            ok = group(lh, vx + 2 * q, tab[a.hoffl + s0 * 4 + q], &tab[a.sell + size_t(s0) * 8 + 2 * q], &tab[a.coefl + (size_t(s0) * 8 + 2 * q) * 4]);
        for (int q = 0; q < 2 && ok; ++q)
            ok = group(ch, (vx >> 1) + 2 * q, tab[a.hoffc + s0 * 2 + q], &tab[a.selc + size_t(s0) * 4 + 2 * q], &tab[a.coefc + (size_t(s0) * 4 + 2 * q) * 4]);
    }
    // LDS staging: any 64 consecutive strips (they wrap over frames: one piece per row touched) must need at most
    // kWalkMaxPieces runs and 64 (luma: 64 nl) sixteen-byte chunks per plane, whatever the runs' alignment
    auto chunks_needed = [&](uint32_t hoff, int windows, int row_len) {
        uint32_t worst = 0;
        for (int s0 = 0; s0 < strips; ++s0) {
            uint32_t chunks = 0;
            int pieces = 0, s1 = s0, left = 64;
            while (left > 0) {
                const int take = std::min(left, strips - s1);
                const uint32_t b = tab[hoff + size_t(s1) * windows];
                const uint32_t e = std::min(tab[hoff + size_t(s1 + take - 1) * windows + windows - 1] + 12u, uint32_t(row_len));
                chunks += (e - b + 30u) >> 4;
                ++pieces; left -= take; s1 = 0;
            }
            if (pieces > kWalkMaxPieces) return ~0u;
            worst = std::max(worst, chunks);
        }
        return worst;
    };
    if (ok) {
        const uint32_t cl = chunks_needed(a.hoffl, 4, lh.src), cc = chunks_needed(a.hoffc, 2, ch.src);
        ok = strips >= 2 && cl <= 128u && cc <= 64u;
        p->nl = cl <= 64u ? 1 : 2;
    }
    if (!ok) { delete p; return nullptr; }
    p->tp = lh.size <= 6 && ch.size <= 6 && lv.size <= 6 && !env("TTS_RGB_WALK_TP4") ? 3 : 4;
    const int rl = 2 * p->tp;
    // per output row and segment length: the luma source row that completes it, the chroma rows that enter first, the
    // luma taps by ring slot (source row - the segment's first row) mod 2 tp, the chroma taps right-aligned in 4 rows
    for (int sr : {16, 24, 32, 48, 64, 96, 128, 192, 256}) {
        RgbWalkPlan::Seg sg;
        sg.seg_rows = sr; sg.nseg = (ph + sr - 1) / sr;
        sg.seginfo = uint32_t(tab.size());
        for (int k = 0; k < sg.nseg; ++k) {
            const int vy = g.crop_y + k * sr;
            tab.push_back(uint32_t(lv.pos[vy]));
            tab.push_back(uint32_t(cv.pos[vy] + cv.size - 4));
        }
        sg.vrow = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(ph) * 8);
        for (int y = 0; y < ph; ++y) {
            const int vy = g.crop_y + y, rstart = lv.pos[g.crop_y + (y / sr) * sr];
            uint32_t *r = &tab[sg.vrow + size_t(y) * 8];
            r[0] = uint32_t(lv.pos[vy] + lv.size - 1);
            r[1] = y ? uint32_t(cv.pos[vy] - cv.pos[vy - 1]) : 0u;
            int16_t tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc[4] = {0, 0, 0, 0};
            for (int k = 0; k < lv.size; ++k) tl[(lv.pos[vy] + k - rstart) % rl] = lv.taps[size_t(vy) * lv.size + k];
            for (int k = 0; k < cv.size; ++k) tc[4 - cv.size + k] = cv.taps[size_t(vy) * cv.size + k];
            for (int k = 0; k < 4; ++k) r[2 + k] = pack16(tl[2 * k], tl[2 * k + 1]);
            r[6] = pack16(tc[0], tc[1]); r[7] = pack16(tc[2], tc[3]);
            if (r[1] > 4u) ok = false;                                           // (a 4-row chroma ring: an output row never skips more)
        }
        p->segs.push_back(sg);
        if (sg.nseg == 1) break;
    }
    a.strips = strips; a.ph = ph; a.pw = pw;
    a.src_h_l = lv.src; a.src_h_c = cv.src;
    a.row_rel[0] = uint32_t(lh.src); a.row_rel[1] = uint32_t(ch.src);            // (the context's source rows: what chunks_needed counted)
    a.swap = swap; a.coef = coef;
    p->no_xp = env("TTS_NO_XP") != 0;
    if (ok) {
        const size_t bytes = std::max<size_t>(tab.size(), 1) * sizeof(uint32_t);
        ok = hipMalloc(reinterpret_cast<void **>(&p->d_tab), bytes) == hipSuccess && hipMemcpy(p->d_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { rgb_walk_destroy(p); return nullptr; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p->occ, walk_pick(bpp, p->nl, p->tp, true), kBlock, 0) != hipSuccess || p->occ < 1) p->occ = 2;
    std::snprintf(p->name, sizeof p->name, "rgb_walk<%d,%d,%d,%d,%d,dma%d,tp%d>", bpp, lh.size, lv.size, ch.size, cv.size, p->nl, p->tp);
    return p;
}

void rgb_walk_destroy(RgbWalkPlan *p)
{
    if (!p) return;
    (void)hipFree(p->d_tab);
    delete p;
}

const char *rgb_walk_name(const RgbWalkPlan *p) { return p ? p->name : ""; }

// Does the walk take this packed batch?  (per-frame pointers, sources under 4 KiB, frames so far apart that the few one
// wave touches do not fit 32-bit offsets: the in-place scheme or the general kernel)
bool rgb_walk_takes(const RgbWalkPlan *p, int n, uint64_t src_span, const FrameAddr &addr)
{
    if (!p || addr.use_table || src_span < 4096u) return false;
    const size_t wave_frames = size_t(63 / p->args.strips + 2);
    return wave_frames * addr.src_pitch <= kMaxSpan && wave_frames * addr.dst_pitch <= kMaxSpan && int64_t(n) * p->args.strips < (int64_t(1) << 30);
}

int rgb_walk_launch(const RgbWalkPlan *p, const uint8_t *src, uint8_t *dst, int n, const int src_stride[2], int dst_stride,
                    uint64_t src_span, uint64_t dst_span, const FrameAddr &addr, hipStream_t st)
{
    const tts_geom &g = p->g;
    const int64_t items = int64_t(n) * p->args.strips;
    // whole-sector stores: rgba always (a chunk never leaves its lane); rgb24 where a wave's strips wrap over frames at
    // most once and no last strip is slid over its neighbour (one held-back chunk at most)
    const bool xp = !p->no_xp && (p->bpp == 4 || (p->args.pw % 8 == 0 && p->args.strips >= 64));
    // the longest segment <= 96 rows that still leaves 3 workgroups per resident slot, else the shortest
    const double slots = double(p->n_cus) * p->occ;
    const int64_t sblocks = (items + kBlock - 1) / kBlock;
    size_t best = 0;
    for (size_t i = 0; i < p->segs.size() && p->segs[i].seg_rows <= 96; ++i)
        if (i == 0 || double(sblocks) * p->segs[i].nseg >= 3.0 * slots) best = i;
    RgbWalkArgs a = p->args;
    a.seginfo = p->segs[best].seginfo; a.vrow = p->segs[best].vrow; a.nseg = p->segs[best].nseg; a.seg_rows = p->segs[best].seg_rows;
    a.n_items = int(items);
    a.src_stride_y = src_stride[0]; a.src_stride_c = src_stride[1]; a.dst_stride = dst_stride;
    a.src_span = src_span; a.dst_span = dst_span; a.addr = addr;
    for (int k = 0; k < 2; ++k) {    // the source planes, seen from the pre-cropped origin the frame offsets point at
        const int pw_ = (g.in_w + k) >> k, ph_ = (g.in_h + k) >> k, px_ = g.pre_x >> k, py_ = g.pre_y >> k;
        a.end_rel[k] = uint32_t(ph_ - py_ - 1) * uint32_t(src_stride[k]) + uint32_t(pw_ - px_);
    }
    return launch_walk("rgb_walk_kernel", walk_pick(p->bpp, p->nl, p->tp, xp), dim3(uint32_t(sblocks) * uint32_t(a.nseg)), dim3(kBlock), st,
                       src, dst, static_cast<const uint32_t *>(p->d_tab), a);
}

}  // namespace tts
