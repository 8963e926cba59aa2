// libttscale: yuv420p 8-bit scale / crop / pad for MI355X (gfx950), HIP kernels + C ABI.
//
// Replaces the per-frame `scale`/`crop`/`pad` loop that the reference hands to an
// external ffmpeg (scripts/processVideos.js:71-108, src/services/video.js:329-334).
// Arithmetic contract: horizontal pass to a 15-bit intermediate
// with a truncating >>7 and a clip to 32767, vertical pass with +64<<12 rounding,
// >>19 and a clip to 8 bits; integer only, bit-exact against the CPU oracle.
//
// Kernel families
//   strip_kernel<LW,R,N>  the hot path.  One lane owns N adjacent output columns of one
//       plane of one frame and walks down a segment of output rows.  Per source row it
//       gathers one LW-byte window per column straight through the vector L1 (the
//       windows of a wave are a contiguous run of the row; the next row's windows are
//       already in flight while this row is computed), runs the horizontal taps with
//       v_dot2 on the window, and drops the 15-bit result into a register ring of R
//       source rows kept as packed row pairs.  Every output row whose last source row
//       has arrived is produced from the ring with v_dot2 against wave-uniform vertical
//       taps (scalar loads) and stored as N contiguous bytes.  The intermediate never
//       leaves registers: HBM sees the source window once and the output once.  No LDS,
//       no barriers, no inter-workgroup traffic.  Luma and chroma planes of a batch go
//       out in ONE launch (block ranges), so the chroma blocks fill the luma tail.
//   generic_kernel        any tap counts (e.g. the 29-tap 720->101 thumbnail), one
//       lane per output sample; correctness path for shapes the strip kernel does not
//       cover.
//   fill_kernel           pad colour outside the picture.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <string>
#include <vector>

#include "filter.h"

namespace {

// ------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------

constexpr int kBlock = 256;
constexpr int kMaxLdsBlock = 1024;    // largest workgroup of the LDS-staged kernel
constexpr size_t kMaxSpan = 0xFF000000u;   // bytes one launch may address from its base pointers
[[maybe_unused]] constexpr uint32_t kDiscard = 0xFFFFFF00u; // a lane offset the buffer range check always rejects (> kMaxSpan)
constexpr int kMaxTableFrames = 32;   // frames addressed through the in-kernarg offset table

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Where the planes of frame f live, relative to the base pointers of the launch.
struct FrameAddr {
    uint32_t src_pitch, dst_pitch;            // packed mode: bytes between frames
    uint32_t src_plane[3], dst_plane[3];      // packed mode: plane origin inside a frame
    int32_t  use_table;                        // table mode: explicit per-frame offsets
    int32_t  wave_base;                        // packed mode: spans pass kMaxSpan, waves address from their own first frame
    uint32_t src_tab[3][kMaxTableFrames];
    uint32_t dst_tab[3][kMaxTableFrames];
};

// One plane class (0 = luma, 1 = both chroma planes) of a strip launch.
struct StripClass {
    uint32_t hoff, hsel, hcoef;    // dword offsets into the table buffer
    uint32_t seginfo, emit, vcoef;
    int32_t  maxrows, nseg, seg_rows;
    int32_t  strips, planes, plane0, n_items;
    int32_t  ow, oh, src_stride, dst_stride;
    uint32_t nblocks;              // workgroups of this class in the launch
    int32_t  paced;                // LDS-staged kernel: strip-major workgroup order + barrier before every row's store
    uint32_t src_begin_rel;        // bytes from a plane's first byte to its (pre-cropped) origin
    uint32_t src_end_rel;          // bytes from a plane's (pre-cropped) origin to one past its last byte
    uint32_t src_row_rel;          // bytes from that origin to the end of its row
};

struct StripArgs {
    StripClass c[2];
    uint64_t src_span, dst_span;   // bytes reachable from the base pointers (buffer range check)
    FrameAddr addr;
};

// clamp = true selects the VOP3P form (v_dot2_i32_i16 d, a, b, c) whose accumulator input is a
// free operand; the sums here never reach the int32 range, so the clamp never acts.
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), c, true);
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }

// Where this lane's plane lies.  Lane offsets are 32-bit, a launch is not: in packed mode every wave
// addresses from ITS first frame (lanes are ordered by frame, so that is lane 0's; the host checks that
// the few frames a wave's 64 items can touch fit kMaxSpan), with the buffer range cut to what is left of
// the launch's span from there.  Table mode: offsets from the launch's bases, as given.
// Call with all lanes active.
__device__ __forceinline__ void frame_offsets(const StripArgs &a, int plane, int frame, const uint8_t *&src, uint8_t *&dst,
                                              uint32_t &src_rec, uint32_t &dst_rec, uint32_t &soff, uint32_t &doff)
{
    if (a.addr.use_table) {
        soff = a.addr.src_tab[plane][frame];
        doff = a.addr.dst_tab[plane][frame];
        src_rec = uint32_t(a.src_span); dst_rec = uint32_t(a.dst_span);
    } else if (!a.addr.wave_base) {                  // the whole launch lies within kMaxSpan of its bases
        soff = uint32_t(frame) * a.addr.src_pitch + a.addr.src_plane[plane];
        doff = uint32_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane];
        src_rec = uint32_t(a.src_span); dst_rec = uint32_t(a.dst_span);
    } else {
        const uint32_t f0 = uni(uint32_t(frame));
        const uint64_t so = uint64_t(f0) * a.addr.src_pitch, dof = uint64_t(f0) * a.addr.dst_pitch;
        src += so; dst += dof;
        src_rec = a.src_span > so ? uint32_t(min(a.src_span - so, uint64_t(kMaxSpan))) : 0u;
        dst_rec = a.dst_span > dof ? uint32_t(min(a.dst_span - dof, uint64_t(kMaxSpan))) : 0u;
        soff = (uint32_t(frame) - f0) * a.addr.src_pitch + a.addr.src_plane[plane];
        doff = (uint32_t(frame) - f0) * a.addr.dst_pitch + a.addr.dst_plane[plane];
    }
}

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of
// logical blocks so neighbouring strips and row segments share one L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// 15-bit intermediate of swscale's horizontal pass: truncating >>7, clip to 32767
__device__ __forceinline__ uint32_t hfinish(int acc)
{
    acc >>= 7;
    return uint32_t(acc < 32767 ? acc : 32767);
}

// four vertical sums -> four clipped bytes (val >> 19, clip to 0..255)
__device__ __forceinline__ uint32_t pack4(int v0, int v1, int v2, int v3)
{
    // high halves of each pair, then >>3 on the packed int16s == >>19 on the int32s
    const uint32_t h01 = __builtin_amdgcn_perm(uint32_t(v1), uint32_t(v0), 0x07060302);
    const uint32_t h23 = __builtin_amdgcn_perm(uint32_t(v3), uint32_t(v2), 0x07060302);
    const short2_t s01 = __builtin_bit_cast(short2_t, h01) >> short2_t{3, 3};
    const short2_t s23 = __builtin_bit_cast(short2_t, h23) >> short2_t{3, 3};
    uint32_t b01, b23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(s01));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(s23));
    return b01 | (b23 << 16);
}

// Output stores are issued from inline asm.  hipcc's s_waitcnt pass counts the vector-memory
// operations it can see; with stores visible inside the variable-trip emission loop it can no
// longer count and falls back to vmcnt(0) before every use of a prefetched row, which drains the
// rows still in flight AND every store acknowledgement once per source row.  Stores it cannot see
// leave it counting the row loads alone, vmcnt((D-1)*loads), which (the counter retires in issue
// order) is never too weak: the hidden stores only make the wait retire a few more operations.
__device__ __forceinline__ void store_b128(u32x4_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
__device__ __forceinline__ void store_b64(u32x2_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
__device__ __forceinline__ void store_b8(uint32_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_byte %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
[[maybe_unused]] __device__ __forceinline__ void store_b16(uint32_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_short %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
[[maybe_unused]] __device__ __forceinline__ void store_b32(uint32_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_dword %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}

// How a lane fetches the source bytes of its N columns from one source row:
//   kGroup4 (taps <= 4, 4 columns span <= 8 bytes): one 8-byte window per 4 columns; each column
//           picks its taps out of the window with a per-lane byte selector (v_perm).
//   else    one LW-byte window per column (LW = 4 or 8), taps at fixed byte positions.
// The texture addresser handles a gather at 4 lanes per clock whatever its width, so fewer,
// wider loads are what keeps it off the critical path.
template <int LW, int N, bool kGroup4> struct RowWin {
    static constexpr int NL = kGroup4 ? N / 4 : N;        // loads per row per lane
    static constexpr int WD = kGroup4 ? 2 : LW / 4;       // dwords per load
};

template <int NL, int WD>
__device__ __forceinline__ void load_row(uint32_t (&w)[NL][WD], __amdgpu_buffer_rsrc_t srs,
                                         const uint32_t (&off)[NL], int rowoff)
{
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        if constexpr (WD == 1) {
            w[i][0] = __builtin_amdgcn_raw_buffer_load_b32(srs, off[i], rowoff, 0);
        } else {
            const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(srs, off[i], rowoff, 0);
            w[i][0] = t.x; w[i][1] = t.y;
        }
    }
}

// tab layout per class (dword offsets in StripClass):
//   hoff   [cols/4] or [cols]   first byte of each window
//   hsel   [cols][2]            (kGroup4 only) v_perm selectors: taps 0,1 and taps 2,3 as 16-bit lanes
//   hcoef  [cols][LW/2]         horizontal taps as int16 pairs
//   seginfo[nseg][2]            first source row, number of source rows of a row segment
//   emit   [nseg][maxrows]      output rows completed by the k-th source row of the segment
//   vcoef  [oh][R/2]            vertical taps by ring slot ((row - first row of segment) % R), int16 pairs
// `tab` is a __restrict__ kernel argument so the wave-uniform reads become scalar loads.
typedef __attribute__((address_space(3))) void lds_void_t;

template <int LW, int R, int N, int D, bool G4>
__global__ __launch_bounds__(kBlock) void strip_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                       const uint32_t *__restrict__ tab, const StripArgs a)
{
    static_assert(LW == 4 || LW == 8, "window is one or two dwords");
    static_assert(R % 2 == 0 && N % 4 == 0, "ring holds row pairs; stores are dwords");
    static_assert(R % D == 0, "in-flight rows rotate with the ring");
    static_assert(!G4 || LW == 4, "grouped windows carry at most 4 taps");
    using RW = RowWin<LW, N, G4>;

    uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const int ci = lb >= a.c[0].nblocks ? 1 : 0;
    if (ci) lb -= a.c[0].nblocks;
    const StripClass &c = a.c[ci];

    const int seg = int(lb % uint32_t(c.nseg));
    const int sb  = int(lb / uint32_t(c.nseg));
    int       g   = sb * kBlock + int(threadIdx.x);
    const bool live = g < c.n_items;
    if (!live) g = c.n_items - 1;

    const int strip = g % c.strips;
    const int fp    = g / c.strips;
    const int plane = c.plane0 + fp % c.planes;
    const int frame = fp / c.planes;
    uint32_t soff, doff, src_rec, dst_rec;
    const uint8_t *wsrc = src;                       // this wave's bases
    uint8_t *wdst = dst;
    frame_offsets(a, plane, frame, wsrc, wdst, src_rec, dst_rec, soff, doff);
    doff += uint32_t(min(strip * N, max(c.ow - N, 0)));      // the last strip is slid left, not cut short

    // per-lane horizontal tables stay in registers for the whole walk
    uint32_t off[RW::NL];
    uint32_t hc[N][LW / 2];
    uint32_t sel[G4 ? N : 1][2];
    {
        const uint32_t *ho = tab + c.hoff + strip * RW::NL;
        const uint32_t *hk = tab + c.hcoef + strip * N * (LW / 2);
#pragma unroll
        for (int i = 0; i < RW::NL; ++i) off[i] = ho[i] + soff;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < LW / 2; ++k) hc[i][k] = hk[i * (LW / 2) + k];
        if constexpr (G4) {
            const uint32_t *hs = tab + c.hsel + strip * N * 2;
#pragma unroll
            for (int i = 0; i < N; ++i) { sel[i][0] = hs[2 * i]; sel[i][1] = hs[2 * i + 1]; }
        }
    }

    uint32_t ring[N][R / 2];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < R / 2; ++k) ring[i][k] = 0;

    const int rstart = int(uni(tab[c.seginfo + 2 * seg]));
    const int nrows  = int(uni(tab[c.seginfo + 2 * seg + 1]));
    const uint32_t *emit = tab + c.emit + seg * c.maxrows;
    const uint32_t *vco  = tab + c.vcoef;
    int y = seg * c.seg_rows;
    const int cols = live ? min(N, c.ow) : 0;                  // < N only for planes narrower than a strip
    const bool all_full = __builtin_amdgcn_ballot_w64(cols != N) == 0;   // wave-uniform

    // Buffer addressing: wave-uniform descriptor + per-lane 32-bit byte offset + scalar row
    // offset, so a row step costs no vector instructions; reads past the span return 0.
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(wsrc), 0, src_rec, 0x00020000);
    const uint64_t dbase = reinterpret_cast<uint64_t>(wdst);
    const u32x4_t drs = {uni(uint32_t(dbase)), uni(uint32_t(dbase >> 32) & 0xffffu), uni(dst_rec), 0x00020000u};

    // D source rows are kept in flight so a row's bytes are requested D iterations before use.
    uint32_t w[D][RW::NL][RW::WD];
#pragma unroll
    for (int d = 0; d < D; ++d) load_row<RW::NL, RW::WD>(w[d], srs, off, (rstart + min(d, nrows - 1)) * c.src_stride);
    int cnt_next = int(uni(emit[0]));

    for (int k0 = 0; k0 < nrows; k0 += R) {
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int k = k0 + s;
            if (k >= nrows) break;                              // wave-uniform
            const int cnt = cnt_next;
            cnt_next = int(uni(emit[min(k + 1, nrows - 1)]));
            uint32_t (&wk)[RW::NL][RW::WD] = w[s % D];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                int acc = 0;
                if constexpr (G4) {
                    acc = dot2(__builtin_amdgcn_perm(wk[i / 4][1], wk[i / 4][0], sel[i][0]), hc[i][0], acc);
                    acc = dot2(__builtin_amdgcn_perm(wk[i / 4][1], wk[i / 4][0], sel[i][1]), hc[i][1], acc);
                } else {
#pragma unroll
                    for (int q = 0; q < LW / 4; ++q) {
                        acc = dot2(__builtin_amdgcn_perm(0, wk[i][q], 0x0c010c00), hc[i][2 * q], acc);
                        acc = dot2(__builtin_amdgcn_perm(0, wk[i][q], 0x0c030c02), hc[i][2 * q + 1], acc);
                    }
                }
                const uint32_t t = hfinish(acc);
                ring[i][s / 2] = (s & 1) ? __builtin_amdgcn_perm(t, ring[i][s / 2], 0x05040100)
                                         : __builtin_amdgcn_perm(ring[i][s / 2], t, 0x07060100);
            }
            // refill the slot just consumed with row k + D (rows past the segment re-read its last row)
            load_row<RW::NL, RW::WD>(wk, srs, off, (rstart + min(k + D, nrows - 1)) * c.src_stride);
            for (int e = 0; e < cnt; ++e) {                     // wave-uniform
                uint32_t vk[R / 2];
#pragma unroll
                for (int q = 0; q < R / 2; ++q) vk[q] = uni(vco[y * (R / 2) + q]);
                uint32_t o[N / 4];
#pragma unroll
                for (int q = 0; q < N / 4; ++q) {
                    int v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int acc = 64 << 12;
#pragma unroll
                        for (int p = 0; p < R / 2; ++p) acc = dot2(ring[4 * q + j][p], vk[p], acc);
                        v[j] = acc;
                    }
                    o[q] = pack4(v[0], v[1], v[2], v[3]);
                }
                const int outoff = y * c.dst_stride;
                if (all_full || cols == N) {
                    if constexpr (N == 16) {
                        store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, drs, doff, outoff);
                    } else {
                        store_b64(u32x2_t{o[0], o[1]}, drs, doff, outoff);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i < cols) store_b8(o[i / 4] >> (8 * (i % 4)), drs, doff + i, outoff);
                }
                ++y;
            }
        }
    }
}

// ---- LDS-staged variant of the strip kernel ------------------------------------------------------
// Same walk as strip_kernel, but per source row a wave first copies the runs of source bytes that
// hold all of its lanes' windows HBM/L2 -> LDS with one or two lane-contiguous 16-byte-per-lane
// LDS-DMAs (`buffer_load_dwordx4 ... lds`): every 128-byte line is requested exactly once, by one
// coalesced instruction, instead of being met by dozens of overlapping per-lane window reads (each
// new line then stalls the vector L1's pipe, stores included, on hit-under-miss; 8 gathers a row
// also keep the texture addresser busy for ~16 cycles each).  The lanes pick their windows out of
// the wave's private LDS slot.  A wave's 64 strips wrap over plane rows (items are flattened over
// frames), so there is one run ("piece") per plane row touched.
// Nothing in flight lives in VGPRs, and an LDS read hipcc can see is ordered behind EVERY LDS-DMA
// still in flight (vmcnt(0)), so the window reads are inline asm and the waits are ours: before row
// k is read we wait, exactly, until only the operations issued after row k's DMA are outstanding
// -- the DMAs of the D-1 later rows plus the stores issued since (counted at run time; gfx950
// retires vector-memory operations in issue order).
// What its speed depends on most is the output row pitch (tools/micro/pattern*.hip):
// 1080-byte rows run at 42 % of the HBM roofline (46 % paced), 1024-byte rows at 50-52 %.
template <int V>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(V) : "memory"); }

// wait until at most B + min(extra, 3) vector-memory operations are outstanding (a smaller count
// than the true one is always safe, it only waits for a little more)
template <int B>
__device__ __forceinline__ void wait_vm_plus(int extra)
{
    if (extra >= 2) {
        if (extra >= 3) wait_vm<B + 3>(); else wait_vm<B + 2>();
    } else {
        if (extra == 1) wait_vm<B + 1>(); else wait_vm<B>();
    }
}

// Window reads of one source row out of the wave's LDS slot.  A window starts at any byte, and a
// misaligned ds_read costs ~64 LDS cycles (one lane at a time; SQ_LDS_UNALIGNED_STALL showed the
// LDS pipe 70-95 % busy with nothing else), so every window is fetched as the three ALIGNED dwords
// that contain it (ds_read2_b32 + ds_read_b32 at addr & ~3) and the byte phase is folded into the
// v_perm selectors / a v_alignbyte afterwards.  Reads and their lgkmcnt wait sit in the SAME asm
// statement: no output register exists, as far as the compiler can tell, before its data landed.
struct Win3 { u32x2_t lo; uint32_t hi; };
// (the two-window reader takes the slot's byte offset as an immediate: with D > 1 rows in flight the
// slot changes from row to row, and ds_read2's 8-bit dword offsets cannot reach the second slot)
#define TTS_LDS_READ_WIN3_X2(w, p, off)                                                                          \
    asm volatile("ds_read_b32 %0, %6 offset:%8\n\tds_read_b32 %1, %6 offset:%8+4\n\tds_read_b32 %2, %6 offset:%8+8\n\t" \
                 "ds_read_b32 %3, %7 offset:%8\n\tds_read_b32 %4, %7 offset:%8+4\n\tds_read_b32 %5, %7 offset:%8+8\n\t" \
                 "s_waitcnt lgkmcnt(0)"                                                                           \
                 : "=&v"(w[0][0]), "=&v"(w[0][1]), "=&v"(w[0][2]), "=&v"(w[1][0]), "=&v"(w[1][1]), "=&v"(w[1][2])   \
                 : "v"(p[0]), "v"(p[1]), "n"(off) : "memory")
// two windows of four aligned dwords (grouped 12-byte windows at any byte phase)
#define TTS_LDS_READ_WIN4_X2(w, p)                                                                                \
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:2 offset1:3\n\t"                     \
                 "ds_read2_b32 %2, %5 offset1:1\n\tds_read2_b32 %3, %5 offset0:2 offset1:3\n\t"                     \
                 "s_waitcnt lgkmcnt(0)"                                                                           \
                 : "=&v"(w[0][0]), "=&v"(w[0][1]), "=&v"(w[1][0]), "=&v"(w[1][1]) : "v"(p[0]), "v"(p[1]) : "memory")
#define TTS_LDS_READ_WIN4_X4(w, p)                                                                                \
    asm volatile("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %8 offset0:2 offset1:3\n\t"                     \
                 "ds_read2_b32 %2, %9 offset1:1\n\tds_read2_b32 %3, %9 offset0:2 offset1:3\n\t"                     \
                 "ds_read2_b32 %4, %10 offset1:1\n\tds_read2_b32 %5, %10 offset0:2 offset1:3\n\t"                   \
                 "ds_read2_b32 %6, %11 offset1:1\n\tds_read2_b32 %7, %11 offset0:2 offset1:3\n\t"                   \
                 "s_waitcnt lgkmcnt(0)"                                                                           \
                 : "=&v"(w[0][0]), "=&v"(w[0][1]), "=&v"(w[1][0]), "=&v"(w[1][1]),                                 \
                   "=&v"(w[2][0]), "=&v"(w[2][1]), "=&v"(w[3][0]), "=&v"(w[3][1])                                  \
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory")
[[maybe_unused]] __device__ __forceinline__ void lds_read_win3(Win3 (&w)[8], const uint32_t (&p)[8])
{
    asm volatile("ds_read2_b32 %0, %16 offset1:1\n\tds_read_b32 %1, %16 offset:8\n\t"
                 "ds_read2_b32 %2, %17 offset1:1\n\tds_read_b32 %3, %17 offset:8\n\t"
                 "ds_read2_b32 %4, %18 offset1:1\n\tds_read_b32 %5, %18 offset:8\n\t"
                 "ds_read2_b32 %6, %19 offset1:1\n\tds_read_b32 %7, %19 offset:8\n\t"
                 "ds_read2_b32 %8, %20 offset1:1\n\tds_read_b32 %9, %20 offset:8\n\t"
                 "ds_read2_b32 %10, %21 offset1:1\n\tds_read_b32 %11, %21 offset:8\n\t"
                 "ds_read2_b32 %12, %22 offset1:1\n\tds_read_b32 %13, %22 offset:8\n\t"
                 "ds_read2_b32 %14, %23 offset1:1\n\tds_read_b32 %15, %23 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0].lo), "=&v"(w[0].hi), "=&v"(w[1].lo), "=&v"(w[1].hi), "=&v"(w[2].lo), "=&v"(w[2].hi),
                   "=&v"(w[3].lo), "=&v"(w[3].hi), "=&v"(w[4].lo), "=&v"(w[4].hi), "=&v"(w[5].lo), "=&v"(w[5].hi),
                   "=&v"(w[6].lo), "=&v"(w[6].hi), "=&v"(w[7].lo), "=&v"(w[7].hi)
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}
// 4-byte windows lie inside two aligned dwords
[[maybe_unused]] __device__ __forceinline__ void lds_read_win2(u32x2_t (&w)[8], const uint32_t (&p)[8])
{
    asm volatile("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %9 offset1:1\n\tds_read2_b32 %2, %10 offset1:1\n\t"
                 "ds_read2_b32 %3, %11 offset1:1\n\tds_read2_b32 %4, %12 offset1:1\n\tds_read2_b32 %5, %13 offset1:1\n\t"
                 "ds_read2_b32 %6, %14 offset1:1\n\tds_read2_b32 %7, %15 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
}

constexpr int kMaxPieces = 8;      // source runs a wave's 64 strips may need (they wrap over plane rows)

// The source runs of one wave of an LDS-staged kernel.  The wave's source bytes are one run per
// plane row its 64 lanes touch ("pieces": a new one starts at every lane whose windows restart in
// another plane, `new_row`).  Piece p covers lanes [l0, l1), bytes [start, end) rounded out to
// 16-byte chunks, and takes the next n chunks of the wave's LDS slot; chunk c of the slot is fetched
// by lane c % 64 of DMA c / 64.  In: this lane's bytes [first, last) in the segment's FIRST source
// row (the DMAs add (row - first row) * stride as their scalar offset, so nothing here can be
// "negative"), its plane's origin `soff`.  Out: csrc[d] = what this lane fetches in DMA d;
// lbase = slot position minus source address for this lane's piece.
template <int NDMA>
__device__ __forceinline__ void plan_runs(const StripArgs &a, const StripClass &c, uint32_t first, uint32_t last, uint32_t soff,
                                          bool new_row, int lane, uint32_t last_rowoff, uint32_t (&csrc)[NDMA], uint32_t &lbase)
{
    uint64_t pm = __builtin_amdgcn_ballot_w64(new_row) | 1ull;
    uint32_t prefix = 0;
    lbase = 0;
    // Spare lanes (chunks past the last piece) re-fetch the first piece's first chunk, so that every
    // lane of an LDS-DMA reads bytes of a plane it was sent to.
#pragma unroll
    for (int d = 0; d < NDMA; ++d) csrc[d] = ~0u;
    for (int p = 0; p < kMaxPieces && pm; ++p) {
        const int l0 = __builtin_ctzll(pm);
        pm &= pm - 1;
        const int l1 = pm ? __builtin_ctzll(pm) : 64;
        // Where may this piece's run lie?  Its chunk grid is the same for every source row of the
        // segment, rows differ by the scalar row offset only.  The run is rounded out to whole
        // chunks; rounding into a neighbouring row of the same plane is harmless, rounding out of
        // the PLANE is not: with per-frame plane pointers those bytes may not be mapped, and at the
        // end of a launch's span the range check zero-fills whole dwords, one that straddles the end
        // and still holds needed bytes included.  So: start on a 16-byte boundary unless that falls
        // before the plane (pointer mode; in packed batches the bytes before a plane are the previous
        // plane), and if the last chunk would pass the plane's end on the segment's last row, slide
        // the run down so that it ends exactly on the piece's last byte.  (A segment holding BOTH the
        // first and the last row of a narrow plane can satisfy only one end; the host never sends
        // such a segment here in pointer mode, and in a packed batch the lower end gives way.)
        const uint32_t porg = uint32_t(__builtin_amdgcn_readlane(int(soff), l0));
        const uint32_t lo_abs = a.addr.use_table ? porg - c.src_begin_rel : 0u, hi_abs = porg + c.src_end_rel;
        const uint32_t end = uint32_t(__builtin_amdgcn_readlane(int(last), l1 - 1));
        uint32_t start = uint32_t(__builtin_amdgcn_readlane(int(first), l0)) & ~15u;
        if (start < lo_abs) start = lo_abs;
        const uint32_t n = (end - start + 15u) >> 4;
        if (start + 16u * n + last_rowoff > hi_abs && end >= lo_abs + 16u * n) start = end - 16u * n;
        if (lane >= l0 && lane < l1) lbase = 16u * prefix - start;
        if (p == 0) {
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
    // the host picks these kernels only for geometries whose runs fit; anything else here would be
    // silent corruption, so trap
    if (pm || prefix > 64u * NDMA) __builtin_trap();
}

// LW, R as in strip_kernel; G4: grouped windows (two 8-byte windows per lane, taps <= 4), else one
// LW-byte window per column; NDMA: 1 KiB LDS-DMAs per source row (the wave's runs hold <= 64 * NDMA
// sixteen-byte chunks).  Eight columns per lane.
template <int LW, int R, int GC, int NDMA, int D>
__global__ __launch_bounds__(kMaxLdsBlock, (GC == 4 && R == 4 && LW == 4) ? 8 : 4) void strip_lds_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                               const uint32_t *__restrict__ tab, const StripArgs a)
{
    // GC: columns that share one window (0: every column its own LW-byte window; 4 with LW = 4: 8-byte
    // windows; 4 with LW = 6, 2 with LW = 6 or 8: 12-byte windows)
    constexpr bool G4 = GC != 0;
    static_assert(R % 2 == 0 && (LW == 4 || LW == 6 || LW == 8) && (NDMA == 1 || NDMA == 2), "see strip_kernel");
    static_assert(GC == 0 || (GC == 4 && (LW == 4 || LW == 6)) || (GC == 2 && (LW == 6 || LW == 8)), "window shapes on offer");
    static_assert(!(G4 && LW >= 6) || D == 1, "the 12-byte group readers address slot 0 only");
    static_assert(D >= 1 && R % D == 0, "D source rows in flight, one LDS slot each, rotating with the ring");
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin exists in the device pass only
    constexpr int N = 8;
    constexpr int NW = G4 ? N / GC : N;              // windows per row per lane
    // HB ("high byte", the 4-column groups): the horizontal sum is formed as 256 x itself so that its
    // finish is two instructions instead of three.  Pixels enter as (p - 128) << 8 -- one v_xor 0x80808080
    // per window dword, the selectors place the byte high -- and the accumulator starts at
    // 128 * 256 * (sum of taps) = 1 << 29 (every column's taps sum to 1 << 14 exactly), which gives
    // 256 * sum(p * tap) with no overflow on the way.  Then val >> 7 clipped to 32767 is the high half of
    // the SATURATING sum acc + acc (v_add_i32 clamp), and the ring insert picks that half.
    constexpr bool HB = GC == 4;
    constexpr int kSlotBytes = NDMA * 1024 + 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];      // (blockDim.x / 64) x D slots
    uint8_t (*lds)[D][kSlotBytes] = reinterpret_cast<uint8_t (*)[D][kSlotBytes]>(lds_raw);

    uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const int ci = lb >= a.c[0].nblocks ? 1 : 0;
    if (ci) lb -= a.c[0].nblocks;
    const StripClass &c = a.c[ci];

    int seg, sb;
    if (c.paced) { const uint32_t nsb = c.nblocks / uint32_t(c.nseg); sb = int(lb % nsb); seg = int(lb / nsb); }
    else { seg = int(lb % uint32_t(c.nseg)); sb = int(lb / uint32_t(c.nseg)); }
    int       g   = sb * int(blockDim.x) + int(threadIdx.x);
    const bool live = g < c.n_items;
    if (!live) g = c.n_items - 1;                  // dead lanes (a suffix of the last workgroup) shadow the last live one
    const int strip = g % c.strips;
    const int fp    = g / c.strips;
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    const int lane = int(threadIdx.x) & 63;

    const int plane = c.plane0 + fp % c.planes;
    const int frame = fp / c.planes;
    uint32_t soff, doff, src_rec, dst_rec;
    const uint8_t *wsrc = src;                       // this wave's bases
    uint8_t *wdst = dst;
    frame_offsets(a, plane, frame, wsrc, wdst, src_rec, dst_rec, soff, doff);
    doff += uint32_t(min(strip * N, c.ow - N));              // the last strip is slid left, not cut short
    // Dead lanes run the same instruction stream (so the wave's store count stays uniform) with an
    // offset past the end of the buffer: the descriptor's range check discards their stores.
    if (!live) doff = kDiscard;

    uint32_t off[NW];                                // first source byte of each window
    uint32_t hc[N][LW / 2];
    uint32_t sel[G4 ? N : 1][2];
    {
        const uint32_t *ho = tab + c.hoff + strip * NW;
        const uint32_t *hk = tab + c.hcoef + strip * N * (LW / 2);
#pragma unroll
        for (int i = 0; i < NW; ++i) off[i] = ho[i] + soff;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < LW / 2; ++k) hc[i][k] = hk[i * (LW / 2) + k];
        if constexpr (G4) {
            const uint32_t *hs = tab + c.hsel + strip * N * 2;
#pragma unroll
            for (int i = 0; i < N; ++i) { sel[i][0] = hs[2 * i]; sel[i][1] = hs[2 * i + 1]; }
            // HB: the selectors put each pixel into the HIGH byte of its 16-bit lane (low byte zero)
            if constexpr (HB) {
#pragma unroll
                for (int i = 0; i < N; ++i) { sel[i][0] = (sel[i][0] << 8) | 0x0cu; sel[i][1] = (sel[i][1] << 8) | 0x0cu; }
            }
        }
    }

    constexpr uint32_t kWin = G4 ? (LW == 4 ? 8u : 12u) : uint32_t(LW);   // grouped: one 8- / 12-byte window per 4 columns
    const int rstart = int(uni(tab[c.seginfo + 2 * seg]));
    const int nrows  = int(uni(tab[c.seginfo + 2 * seg + 1]));
    const int src_stride = int(uni(uint32_t(c.src_stride)));
    const uint32_t row0_off = uint32_t(rstart) * uint32_t(src_stride);
    uint32_t csrc[NDMA];
    uint32_t lbase;
    // (a 12-byte group window may hang over the row's end by the bytes its taps do not use: the run stops
    // at the row's end, and what the window read picks up behind it meets zero taps)
    plan_runs<NDMA>(a, c, off[0] + row0_off, min(off[NW - 1] + kWin, soff + c.src_row_rel) + row0_off, soff,
                    lane > 0 && strip == 0 && live, lane, uint32_t(nrows - 1) * uint32_t(src_stride), csrc, lbase);

    const uint32_t lds_wave = uint32_t(reinterpret_cast<uintptr_t>((lds_void_t *)&lds[wave][0][0]));   // LDS byte address
    uint32_t lpos[NW];                               // aligned LDS address of each window's first dword
    uint32_t phase[G4 ? NW : 1];                     // G4: byte phase of the window inside it (v_alignbyte uses bits 1:0)
    uint32_t psel[G4 ? 1 : N][2];                    // per-column windows: the phase folded into v_perm selectors
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const uint32_t pos = off[j] + row0_off + lbase + lds_wave;
        lpos[j] = pos & ~3u;
        if constexpr (G4) phase[j] = pos;
        else {
            const uint32_t o = pos & 3u;             // taps 2q, 2q+1 (as 16-bit lanes) sit at bytes o, o+1 / o+2, o+3 of a dword pair
            psel[j][0] = 0x0c000c00u | o | ((o + 1u) << 16);
            psel[j][1] = 0x0c000c00u | (o + 2u) | ((o + 3u) << 16);
        }
    }

    uint32_t ring[N][R / 2];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < R / 2; ++k) ring[i][k] = 0;

    const uint32_t *emit = tab + c.emit + seg * c.maxrows;
    const uint32_t *vco  = tab + c.vcoef;
    int y = seg * c.seg_rows;                                  // (the host picks this kernel only if ow >= N: no ragged strips)
    // class fields used inside the loop, read once (the asm statements below clobber "memory", which
    // would otherwise make the compiler reload them from the kernel arguments every row)
    const int dst_stride = int(uni(uint32_t(c.dst_stride)));
    const int last_y = c.oh - 1;
    const bool bar = c.paced != 0;

    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(wsrc), 0, src_rec, 0x00020000);
    const uint64_t dbase = reinterpret_cast<uint64_t>(wdst);
    const u32x4_t drs = {uni(uint32_t(dbase)), uni(uint32_t(dbase >> 32) & 0xffffu), uni(dst_rec), 0x00020000u};

#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_void_t *)&lds[wave][r][d * 1024], 16, csrc[d],
                                                     min(r, nrows - 1) * src_stride, 0, 0);
    int cnt_next = int(uni(emit[0]));
    uint32_t vk_next[R / 2];                       // vertical taps of the next output row, requested a row ahead
#pragma unroll
    for (int q = 0; q < R / 2; ++q) vk_next[q] = uni(vco[y * (R / 2) + q]);
    int hist[D];                                   // stores issued in each of the last D iterations
#pragma unroll
    for (int r = 0; r < D; ++r) hist[r] = 0;

    for (int k0 = 0; k0 < nrows; k0 += R) {
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int k = k0 + s;
            if (k >= nrows) break;                              // wave-uniform
            const int cnt = cnt_next;
            cnt_next = int(uni(emit[min(k + 1, nrows - 1)]));
            // row k's DMA is followed by the DMAs of D-1 younger rows and the stores of the last D iterations
            int younger = 0;
#pragma unroll
            for (int r = 0; r < D; ++r) younger += hist[r];
            wait_vm_plus<(D - 1) * NDMA>(younger);
            constexpr int GD = G4 && LW >= 6 ? 4 : 3;
            uint32_t wk[NW][GD];                                // the aligned dwords holding each window
            if constexpr (G4 && LW >= 6) {
                u32x2_t wv[NW][2];
                if constexpr (NW == 2) TTS_LDS_READ_WIN4_X2(wv, lpos); else TTS_LDS_READ_WIN4_X4(wv, lpos);
#pragma unroll
                for (int j = 0; j < NW; ++j) { wk[j][0] = wv[j][0].x; wk[j][1] = wv[j][0].y; wk[j][2] = wv[j][1].x; wk[j][3] = wv[j][1].y; }
            } else if constexpr (G4) {
                TTS_LDS_READ_WIN3_X2(wk, lpos, (s % D) * kSlotBytes);
            } else if constexpr (LW > 4) {
                static_assert(G4 || D == 1, "per-column readers address slot 0 only");
                Win3 wv[NW];
                lds_read_win3(wv, lpos);
#pragma unroll
                for (int j = 0; j < NW; ++j) { wk[j][0] = wv[j].lo.x; wk[j][1] = wv[j].lo.y; wk[j][2] = wv[j].hi; }
            } else {
                u32x2_t wv[NW];
                lds_read_win2(wv, lpos);
#pragma unroll
                for (int j = 0; j < NW; ++j) { wk[j][0] = wv[j].x; wk[j][1] = wv[j].y; wk[j][2] = 0; }
            }
            if constexpr (G4) {                                 // rebuild the two 8- / 12-byte windows
#pragma unroll
                for (int j = 0; j < NW; ++j)
#pragma unroll
                    for (int q = 0; q + 1 < GD; ++q) {
                        wk[j][q] = __builtin_amdgcn_alignbyte(wk[j][q + 1], wk[j][q], phase[j]);
                        if constexpr (HB) wk[j][q] ^= 0x80808080u;
                    }
            }
            // the slot is free again (its bytes sit in registers): request row k + D into it
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_void_t *)&lds[wave][s % D][d * 1024], 16, csrc[d],
                                                         min(k + D, nrows - 1) * src_stride, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                int acc = HB ? (1 << 29) : 0;
                if constexpr (G4) {
                    constexpr int gi_div = GC;
                    acc = dot2(__builtin_amdgcn_perm(wk[i / gi_div][1], wk[i / gi_div][0], sel[i][0]), hc[i][0], acc);
                    acc = dot2(__builtin_amdgcn_perm(wk[i / gi_div][1], wk[i / gi_div][0], sel[i][1]), hc[i][1], acc);
                    // taps 4..7 sit four bytes on: the same selectors, one dword further
                    if constexpr (LW >= 6) acc = dot2(__builtin_amdgcn_perm(wk[i / gi_div][2], wk[i / gi_div][1], sel[i][0]), hc[i][2], acc);
                    if constexpr (LW == 8) acc = dot2(__builtin_amdgcn_perm(wk[i / gi_div][2], wk[i / gi_div][1], sel[i][1]), hc[i][3], acc);
                } else {
#pragma unroll
                    for (int tp = 0; tp < LW / 2; ++tp)         // taps 2tp, 2tp+1 sit at bytes o + 2 (tp % 2) of dwords tp / 2, tp / 2 + 1
                        acc = dot2(__builtin_amdgcn_perm(wk[i][tp / 2 + 1], wk[i][tp / 2], psel[i][tp % 2]), hc[i][tp], acc);
                }
                if constexpr (HB) {
                    uint32_t t2;
                    asm("v_add_i32 %0, %1, %1 clamp" : "=v"(t2) : "v"(acc));
                    ring[i][s / 2] = (s & 1) ? __builtin_amdgcn_perm(t2, ring[i][s / 2], 0x07060100)
                                             : __builtin_amdgcn_perm(ring[i][s / 2], t2, 0x07060302);
                } else {
                    const uint32_t t = hfinish(acc);
                    ring[i][s / 2] = (s & 1) ? __builtin_amdgcn_perm(t, ring[i][s / 2], 0x05040100)
                                             : __builtin_amdgcn_perm(ring[i][s / 2], t, 0x07060100);
                }
            }
            for (int e = 0; e < cnt; ++e) {                     // wave-uniform
                uint32_t vk[R / 2];
#pragma unroll
                for (int q = 0; q < R / 2; ++q) vk[q] = vk_next[q];
                const int yn = min(y + 1, last_y);
#pragma unroll
                for (int q = 0; q < R / 2; ++q) vk_next[q] = uni(vco[yn * (R / 2) + q]);
                uint32_t o[N / 4];
#pragma unroll
                for (int q = 0; q < N / 4; ++q) {
                    int v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int acc = 64 << 12;
#pragma unroll
                        for (int p = 0; p < R / 2; ++p) acc = dot2(ring[4 * q + j][p], vk[p], acc);
                        v[j] = acc;
                    }
                    o[q] = pack4(v[0], v[1], v[2], v[3]);
                }
                if (bar) asm volatile("s_barrier" ::: "memory");
                store_b64(u32x2_t{o[0], o[1]}, drs, doff, y * dst_stride);
                ++y;
            }
            hist[s % D] = cnt;
        }
    }
#endif
}

// ---- column walk: many-tap downscales (9..64 taps per axis) ------------------------------------
// The reference's own thumbnail chain (src/services/video.js:329-332) shrinks 1080p by 6x and 4K by
// 12x with bicubic: 25 to 49 taps per axis.  Same walk as strip_lds_kernel with ONE output column
// per lane: a wave's 64 columns read one run of (64 x ratio + taps) source bytes per source row
// through the LDS slot, every lane dots its 2*NTP-byte window with its own taps (NTP v_perm +
// NTP v_dot2), and the 15-bit results of the last R source rows sit in a register ring of R/2
// int16 pairs; an output row is R/2 v_dot2 against the wave-uniform vertical taps and one byte
// store per lane (a wave stores 64 contiguous bytes).  Every source byte is read once per row
// segment; the generic kernel re-reads each one taps^2 / ratio^2 times.
template <int NDW>
__device__ __forceinline__ void lds_read_dwords(uint32_t (&w)[NDW], uint32_t p);
template <>
[[maybe_unused]] __device__ __forceinline__ void lds_read_dwords<5>(uint32_t (&w)[5], uint32_t p)
{
    u32x2_t a, b; uint32_t e;
    asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\t"
                 "ds_read_b32 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(e) : "v"(p) : "memory");
    w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; w[4] = e;
}
template <>
[[maybe_unused]] __device__ __forceinline__ void lds_read_dwords<9>(uint32_t (&w)[9], uint32_t p)
{
    u32x2_t a, b, c, d; uint32_t e;
    asm volatile("ds_read2_b32 %0, %5 offset1:1\n\tds_read2_b32 %1, %5 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %2, %5 offset0:4 offset1:5\n\tds_read2_b32 %3, %5 offset0:6 offset1:7\n\t"
                 "ds_read_b32 %4, %5 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p) : "memory");
    w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; w[4] = c.x; w[5] = c.y; w[6] = d.x; w[7] = d.y; w[8] = e;
}
template <>
[[maybe_unused]] __device__ __forceinline__ void lds_read_dwords<17>(uint32_t (&w)[17], uint32_t p)
{
    u32x2_t a, b, c, d, e, f, g, h; uint32_t i;
    asm volatile("ds_read2_b32 %0, %9 offset1:1\n\tds_read2_b32 %1, %9 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %2, %9 offset0:4 offset1:5\n\tds_read2_b32 %3, %9 offset0:6 offset1:7\n\t"
                 "ds_read2_b32 %4, %9 offset0:8 offset1:9\n\tds_read2_b32 %5, %9 offset0:10 offset1:11\n\t"
                 "ds_read2_b32 %6, %9 offset0:12 offset1:13\n\tds_read2_b32 %7, %9 offset0:14 offset1:15\n\t"
                 "ds_read_b32 %8, %9 offset:64\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h), "=&v"(i) : "v"(p) : "memory");
    w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; w[4] = c.x; w[5] = c.y; w[6] = d.x; w[7] = d.y;
    w[8] = e.x; w[9] = e.y; w[10] = f.x; w[11] = f.y; w[12] = g.x; w[13] = g.y; w[14] = h.x; w[15] = h.y; w[16] = i;
}

// NTP: tap pairs per column (window = 2 * NTP bytes); R: ring depth >= vertical taps; NDMA as above;
// NC: adjacent output columns per lane (1, 2 or 4: the per-row bookkeeping -- DMA, waits, loop control,
// the store -- is the same for 1 byte or 4, so small ratios, where it weighs most, take the widest
// strip their registers allow; the last strip of a row is slid left like strip_lds_kernel's).
template <int NTP, int R, int NDMA, int NC>
__global__ __launch_bounds__(kBlock) void column_lds_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                            const uint32_t *__restrict__ tab, const StripArgs a)
{
    static_assert(NTP % 2 == 0 && R % 2 == 0 && (NDMA == 1 || NDMA == 2), "window is whole dwords; ring holds row pairs");
    static_assert(NC == 1 || NC == 2 || NC == 4, "one byte, short or dword store per lane and row");
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin exists in the device pass only
    constexpr int NDW = NTP / 2 + 1;                 // aligned dwords that hold a 2 * NTP-byte window at any byte phase
    constexpr int kSlotBytes = NDMA * 1024 + 16;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kBlock / 64][kSlotBytes];

    uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const int ci = lb >= a.c[0].nblocks ? 1 : 0;
    if (ci) lb -= a.c[0].nblocks;
    const StripClass &c = a.c[ci];

    const int seg = int(lb % uint32_t(c.nseg));
    const int sb  = int(lb / uint32_t(c.nseg));
    int       g   = sb * kBlock + int(threadIdx.x);
    const bool live = g < c.n_items;
    if (!live) g = c.n_items - 1;                  // dead lanes (a suffix of the last workgroup) shadow the last live one
    const int strip = g % c.strips;                // NC output columns per lane
    const int fp    = g / c.strips;
    const int col   = min(strip * NC, c.ow - NC);  // the last strip is slid left, not cut short
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    const int lane = int(threadIdx.x) & 63;

    const int plane = c.plane0 + fp % c.planes;
    const int frame = fp / c.planes;
    uint32_t soff, doff, src_rec, dst_rec;
    const uint8_t *wsrc = src;                       // this wave's bases
    uint8_t *wdst = dst;
    frame_offsets(a, plane, frame, wsrc, wdst, src_rec, dst_rec, soff, doff);
    doff += uint32_t(col);
    if (!live) doff = kDiscard;                    // see strip_lds_kernel

    uint32_t hc[NC][NTP];
    uint32_t off[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
#pragma unroll
        for (int k = 0; k < NTP; ++k) hc[j][k] = tab[c.hcoef + (col + j) * NTP + k];
        off[j] = tab[c.hoff + col + j] + soff;
    }

    const int rstart = int(uni(tab[c.seginfo + 2 * seg]));
    const int nrows  = int(uni(tab[c.seginfo + 2 * seg + 1]));
    const int src_stride = int(uni(uint32_t(c.src_stride)));
    const uint32_t row0_off = uint32_t(rstart) * uint32_t(src_stride);
    uint32_t csrc[NDMA];
    uint32_t lbase;
    plan_runs<NDMA>(a, c, off[0] + row0_off, off[NC - 1] + 2u * NTP + row0_off, soff, lane > 0 && strip == 0 && live, lane,
                    uint32_t(nrows - 1) * uint32_t(src_stride), csrc, lbase);

    uint32_t lpos[NC], psel[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const uint32_t pos = off[j] + row0_off + lbase + uint32_t(reinterpret_cast<uintptr_t>((lds_void_t *)&lds[wave][0]));
        const uint32_t o = pos & 3u;
        lpos[j] = pos & ~3u;
        // taps 2tp, 2tp+1 (as 16-bit lanes) sit at bytes o + 2 (tp % 2), +1 of dwords tp / 2, tp / 2 + 1
        psel[j][0] = 0x0c000c00u | o | ((o + 1u) << 16);
        psel[j][1] = 0x0c000c00u | (o + 2u) | ((o + 3u) << 16);
    }

    uint32_t ring[NC][R / 2];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int k = 0; k < R / 2; ++k) ring[j][k] = 0;

    const uint32_t *emit = tab + c.emit + seg * c.maxrows;
    const uint32_t *vco  = tab + c.vcoef;
    int y = seg * c.seg_rows;
    const int dst_stride = int(uni(uint32_t(c.dst_stride)));

    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(wsrc), 0, src_rec, 0x00020000);
    const uint64_t dbase = reinterpret_cast<uint64_t>(wdst);
    const u32x4_t drs = {uni(uint32_t(dbase)), uni(uint32_t(dbase >> 32) & 0xffffu), uni(dst_rec), 0x00020000u};

#pragma unroll
    for (int d = 0; d < NDMA; ++d)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_void_t *)&lds[wave][d * 1024], 16, csrc[d], 0, 0, 0);
    int cnt_next = int(uni(emit[0]));
    int prev_cnt = 0;                              // stores issued since the DMA of the row about to be read

    for (int k0 = 0; k0 < nrows; k0 += R) {
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int k = k0 + s;
            if (k >= nrows) continue;                           // wave-uniform (no early exit: the loop must unroll fully)
            const int cnt = cnt_next;
            cnt_next = int(uni(emit[min(k + 1, nrows - 1)]));
            wait_vm_plus<0>(prev_cnt);                          // row k's DMA has landed; younger stores may still fly
            uint32_t w[NC][NDW];
#pragma unroll
            for (int j = 0; j < NC; ++j) lds_read_dwords<NDW>(w[j], lpos[j]);
            // the slot is free again (its bytes sit in registers): request row k + 1 into it
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_void_t *)&lds[wave][d * 1024], 16, csrc[d],
                                                         min(k + 1, nrows - 1) * src_stride, 0, 0);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                int acc = 0;
#pragma unroll
                for (int tp = 0; tp < NTP; ++tp)
                    acc = dot2(__builtin_amdgcn_perm(w[j][tp / 2 + 1], w[j][tp / 2], psel[j][tp % 2]), hc[j][tp], acc);
                const uint32_t t = hfinish(acc);
                ring[j][s / 2] = (s & 1) ? __builtin_amdgcn_perm(t, ring[j][s / 2], 0x05040100)
                                         : __builtin_amdgcn_perm(ring[j][s / 2], t, 0x07060100);
            }
            for (int e = 0; e < cnt; ++e) {                     // wave-uniform
                int v[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) v[j] = 64 << 12;
#pragma unroll
                for (int q = 0; q < R / 2; ++q) {
                    const uint32_t vk = uni(vco[y * (R / 2) + q]);
#pragma unroll
                    for (int j = 0; j < NC; ++j) v[j] = dot2(ring[j][q], vk, v[j]);
                }
                const int yoff = int(uni(uint32_t(y * dst_stride)));
                if constexpr (NC == 4) {
                    store_b32(pack4(v[0], v[1], v[2], v[3]), drs, doff, yoff);
                } else {
#pragma unroll
                    for (int j = 0; j < NC; ++j) { v[j] >>= 19; v[j] = v[j] < 0 ? 0 : v[j] > 255 ? 255 : v[j]; }
                    if constexpr (NC == 2) store_b16(uint32_t(v[0]) | (uint32_t(v[1]) << 8), drs, doff, yoff);
                    else store_b8(uint32_t(v[0]), drs, doff, yoff);
                }
                ++y;
            }
            prev_cnt = cnt;
        }
    }
#endif
}

struct GenericArgs {
    const uint8_t *src;
    uint8_t       *dst;
    const int32_t *hpos;  const int16_t *htaps;   // indexed by virtual column
    const int32_t *vpos;  const int16_t *vtaps;   // indexed by virtual row
    int32_t ht, vt;
    int32_t x0, y0, ow, oh;
    int32_t src_stride, dst_stride;
    int32_t planes, plane0;
    int32_t frame0;                    // first frame of this launch (grid.z holds at most 65535 planes)
    FrameAddr addr;
};

__global__ __launch_bounds__(kBlock) void generic_kernel(const GenericArgs a)
{
    const int x  = int(blockIdx.x) * 64 + int(threadIdx.x & 63);
    const int y  = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    const int fp = int(blockIdx.z);
    if (x >= a.ow || y >= a.oh) return;
    const int plane = a.plane0 + fp % a.planes, frame = a.frame0 + fp / a.planes;
    size_t soff, doff;
    if (a.addr.use_table) {
        soff = a.addr.src_tab[plane][frame];
        doff = a.addr.dst_tab[plane][frame];
    } else {
        soff = size_t(frame) * a.addr.src_pitch + a.addr.src_plane[plane];
        doff = size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane];
    }
    const int vx = a.x0 + x, vy = a.y0 + y;
    const int hp = a.hpos[vx], vp = a.vpos[vy];
    const int16_t *hc = a.htaps + size_t(vx) * a.ht;
    const int16_t *vc = a.vtaps + size_t(vy) * a.vt;
    int val = 64 << 12;
    for (int j = 0; j < a.vt; ++j) {
        const uint8_t *row = a.src + soff + size_t(vp + j) * size_t(a.src_stride) + hp;
        int acc = 0;
        for (int k = 0; k < a.ht; ++k) acc += int(row[k]) * int(hc[k]);
        acc >>= 7;
        acc = acc < 32767 ? acc : 32767;
        val += acc * int(vc[j]);
    }
    val >>= 19;
    val = val < 0 ? 0 : val > 255 ? 255 : val;
    a.dst[doff + size_t(y) * size_t(a.dst_stride) + x] = uint8_t(val);
}

struct FillArgs {
    uint8_t *dst;
    int32_t w, h, stride;              // canvas of this plane class
    int32_t px, py, pw, ph;            // picture rectangle that is left alone
    int32_t planes, plane0;
    int32_t fill[3];
    int32_t frame0;
    uint32_t origin;                   // bytes from the canvas origin to the picture origin
    FrameAddr addr;                    // dst side only (plane origins point at the picture)
};

__global__ __launch_bounds__(kBlock) void fill_kernel(const FillArgs a)
{
    const int x  = (int(blockIdx.x) * 64 + int(threadIdx.x & 63)) * 4;
    const int y  = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    const int fp = int(blockIdx.z);
    if (x >= a.w || y >= a.h) return;
    const int plane = a.plane0 + fp % a.planes, frame = a.frame0 + fp / a.planes;
    const size_t doff = (a.addr.use_table ? size_t(a.addr.dst_tab[plane][frame])
                                          : size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane]) - a.origin;
    uint8_t *row = a.dst + doff + size_t(y) * size_t(a.stride);
    const uint8_t f = uint8_t(a.fill[plane]);
    const bool in_rows = y >= a.py && y < a.py + a.ph;
    for (int i = x; i < min(x + 4, a.w); ++i)
        if (!(in_rows && i >= a.px && i < a.px + a.pw)) row[i] = f;
}

// Range step (tts_geom.out_range): every byte of the finished canvas through a 256-entry table,
// in place.  Thumbnail-sized work; four bytes per lane.
struct RangeArgs {
    uint8_t *dst;
    int32_t w, h, stride;              // canvas of this plane class
    int32_t planes, plane0;
    int32_t frame0;
    uint32_t origin;                   // bytes from the canvas origin to the picture origin
    FrameAddr addr;
    uint8_t lut[256];
};

__global__ __launch_bounds__(kBlock) void range_kernel(const RangeArgs a)
{
    const int x  = (int(blockIdx.x) * 64 + int(threadIdx.x & 63)) * 4;
    const int y  = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    const int fp = int(blockIdx.z);
    if (x >= a.w || y >= a.h) return;
    const int plane = a.plane0 + fp % a.planes, frame = a.frame0 + fp / a.planes;
    const size_t doff = (a.addr.use_table ? size_t(a.addr.dst_tab[plane][frame])
                                          : size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane]) - a.origin;
    uint8_t *row = a.dst + doff + size_t(y) * size_t(a.stride);
    for (int i = x; i < min(x + 4, a.w); ++i) row[i] = a.lut[row[i]];
}

}  // namespace

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------

namespace {

constexpr int kSegOptions[] = {16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160, 192, 256};
constexpr int kNumSegOptions = int(sizeof(kSegOptions) / sizeof(kSegOptions[0]));

struct SegTables {
    int seg_rows = 0, nseg = 0, maxrows = 0;
    uint32_t seginfo = 0, emit = 0, vcoef = 0;     // dword offsets into the plan's table buffer
    int total_src_rows = 0;                        // sum over segments (halo cost model)
    bool spans_plane = false;                      // some segment needs both the first and the last source row
};

// Horizontal tables of one plane class for one strip width (dword offsets into the table buffer).
struct HTables {
    int strips = 0;
    uint32_t hoff = 0, hcoef = 0;                  // one window per column
    bool g4 = false;                               // grouped 8-byte windows usable (taps <= 4, 4 columns span <= 8 bytes)
    uint32_t hoff4 = 0, hsel4 = 0, hcoef4 = 0;
    int lds_g4 = 0, lds_col = 0;                   // LDS-staged kernel: 1 KiB DMAs per row it needs with grouped /
                                                   // per-column windows (0 = not usable)
    uint32_t hoff6 = 0, hcoef6 = 0;                // 5-6 taps: 6-byte windows for the LDS-staged kernel (3 tap pairs a column)
    // grouped 12-byte windows (LDS-staged kernel only): 5-6 taps with 4 or 2 columns a window, 7-8 taps with 2
    struct Grouped12 { int lw = 0, gc = 0; uint32_t hoff = 0, hsel = 0, hcoef = 0; int lds = 0; };
    Grouped12 g12;                                 // the variant this class takes (lds == 0: none)
};

struct ClassPlan {
    tts::AxisFilter h, v;
    int sub = 0;
    int src_w = 0, src_h = 0;          // (pre-cropped) source plane
    int x0 = 0, y0 = 0, ow = 0, oh = 0; // region of the virtual scaled plane that is produced
    int canvas_w = 0, canvas_h = 0;    // output plane
    int pad_x = 0, pad_y = 0;
    int kind = 0;                      // 0 generic, 1 strip, 2 column
    int LW = 0, R = 0;
    int LWlds = 0;                     // window width of the LDS-staged kernel's per-column mode (6 when 5-6 taps)
    // kind 2: column walk (9..64 taps): one output column per lane
    int NTP = 0, Rc = 0, col_dmas = 0; // tap pairs per column, ring depth, LDS-DMAs per source row
    int col_nc = 1;                    // output columns per lane (1, 2, 4)
    uint32_t hoff1 = 0, hcoef1 = 0;    // window start per column, taps per column [ow][NTP]
    HTables ht[2];                     // horizontal tables for N = 8 and N = 16 columns per lane
    std::vector<SegTables> segs;
    // generic tables
    int32_t *d_hpos = nullptr, *d_vpos = nullptr;
    int16_t *d_htaps = nullptr, *d_vtaps = nullptr;
};

thread_local std::string g_last_error;

template <class T>
bool upload(T *&dptr, const std::vector<T> &host)
{
    if (hipMalloc(reinterpret_cast<void **>(&dptr), std::max<size_t>(host.size(), 1) * sizeof(T)) != hipSuccess) return false;
    return hipMemcpy(dptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

inline int ceil_shift(int v, int s) { return (v + (1 << s) - 1) >> s; }

int env_int(const char *name, int dflt)
{
    const char *e = std::getenv(name);
    return e && *e ? std::atoi(e) : dflt;
}

}  // namespace

// Tuning knobs (environment), read ONCE when a plan is created so that tts_run* never calls getenv.
// Defaults are the product behaviour; none of them changes a pixel (tests/test_gpu_parity.py checks
// that); tools/sweep.py sets them per plan.  The TTS_EXP_* timing experiments of DESIGN.md 4.5 DO
// produce wrong output (they drop or alias traffic to see what a launch costs without it) and exist
// only in builds made with -DTTS_EXPERIMENTS.
struct Knobs {
    int strip_n = 8, strip_d = 0, paced = -1, lds_block = 0, no_merge = 0;
    int exp_nosrc = 0, exp_nodst = 0, exp_srcpitch0 = 0, exp_dstpitch0 = 0, exp_stride0 = 0;
    int planewise = 0;                // tts_run: always take the plane-by-plane path (test knob)
    int span_chunks = 0;              // tts_run_batch: always cut batches into launches whose whole span fits 32 bits (test knob)
    void read()
    {
        strip_n = env_int("TTS_STRIP_N", 8); strip_d = env_int("TTS_STRIP_D", 0);
        paced = env_int("TTS_PACED", -1); lds_block = env_int("TTS_LDS_BLOCK", 0); no_merge = env_int("TTS_NO_MERGE", 0);
#ifdef TTS_EXPERIMENTS
        exp_nosrc = env_int("TTS_EXP_NOSRC", 0); exp_nodst = env_int("TTS_EXP_NODST", 0);
        exp_srcpitch0 = env_int("TTS_EXP_SRCPITCH0", 0); exp_dstpitch0 = env_int("TTS_EXP_DSTPITCH0", 0);
        exp_stride0 = env_int("TTS_EXP_STRIDE0", 0);
#endif
        planewise = env_int("TTS_PLANEWISE", 0);
        span_chunks = env_int("TTS_SPAN_CHUNKS", 0);
    }
};

struct tts_plan {
    tts_geom g;
    Knobs knobs;
    ClassPlan cls[2];
    uint32_t *d_tab = nullptr;        // strip tables of both classes
    int device = 0;
    int n_cus = 256;
};

namespace {

bool monotone(const std::vector<int32_t> &p)
{
    for (size_t i = 1; i < p.size(); ++i) if (p[i] < p[i - 1]) return false;
    return true;
}

inline uint32_t pack16(int16_t lo, int16_t hi) { return uint32_t(uint16_t(lo)) | (uint32_t(uint16_t(hi)) << 16); }

void build_seg_tables(const ClassPlan &c, int R, int seg_rows, std::vector<uint32_t> &tab, SegTables &st)
{
    const int VT = c.v.size;
    st.seg_rows = seg_rows;
    st.nseg = (c.oh + seg_rows - 1) / seg_rows;
    st.maxrows = 0; st.total_src_rows = 0;
    std::vector<int> first(st.nseg), rows(st.nseg);
    for (int s = 0; s < st.nseg; ++s) {
        const int y0 = s * seg_rows, y1 = std::min(c.oh, y0 + seg_rows);
        first[s] = c.v.pos[c.y0 + y0];
        rows[s] = c.v.pos[c.y0 + y1 - 1] + VT - first[s];
        st.maxrows = std::max(st.maxrows, rows[s]);
        st.total_src_rows += rows[s];
        st.spans_plane |= first[s] == 0 && first[s] + rows[s] >= c.src_h;
    }
    st.seginfo = uint32_t(tab.size());
    for (int s = 0; s < st.nseg; ++s) { tab.push_back(uint32_t(first[s])); tab.push_back(uint32_t(rows[s])); }
    st.emit = uint32_t(tab.size());
    tab.resize(tab.size() + size_t(st.nseg) * st.maxrows, 0);
    st.vcoef = uint32_t(tab.size());
    tab.resize(tab.size() + size_t(c.oh) * (R / 2), 0);
    for (int y = 0; y < c.oh; ++y) {
        const int s = y / seg_rows, vy = c.y0 + y;
        const int p = c.v.pos[vy];
        tab[st.emit + size_t(s) * st.maxrows + (p + VT - 1 - first[s])] += 1;
        int16_t slot[64] = {0};
        for (int j = 0; j < VT; ++j) slot[(p + j - first[s]) % R] = c.v.taps[size_t(vy) * VT + j];
        for (int k = 0; k < R / 2; ++k) tab[st.vcoef + size_t(y) * (R / 2) + k] = pack16(slot[2 * k], slot[2 * k + 1]);
    }
}

// LDS-staged kernel: how many 1 KiB LDS-DMAs per source row does a wave need?  Any 64 consecutive
// strips of the flattened (plane, strip) order -- they wrap over plane rows -- need one run of source
// bytes per row touched, each rounded out to 16-byte chunks (worst case over the run's alignment);
// the runs must fit 64 chunks per DMA and kMaxPieces runs.  `lo` holds the first byte of each of the
// `nw` windows of a strip, `win` their width.  0 = not usable.
int lds_dmas(const ClassPlan &c, const HTables &t, int N, const int *lo, int nw, int win)
{
    if ((N != 8 && N != 4 && N != 2 && N != 1) || c.ow < N || t.strips < 2 || env_int("TTS_NO_LDS", 0)) return 0;
    auto chunks = [&](int s0, int s1) {   // strips [s0, s1] of one row
        const uint32_t b = uint32_t(lo[size_t(s0) * nw]), e = uint32_t(lo[size_t(s1) * nw + nw - 1] + win);
        return (e - b + 30u) >> 4;
    };
    uint32_t worst = 0;
    for (int s0 = 0; s0 < t.strips; ++s0) {
        uint32_t n = 0;
        int pieces = 0, s = s0, left = 64;
        while (left > 0) {
            const int take = std::min(left, t.strips - s);
            n += chunks(s, s + take - 1);
            ++pieces; left -= take; s = 0;
        }
        if (pieces > kMaxPieces) return 0;
        worst = std::max(worst, n);
    }
    return worst <= 64u ? 1 : worst <= 128u ? 2 : 0;
}

int build_class(const tts_geom &g, int sub, ClassPlan &c, std::vector<uint32_t> &tab)
{
    c.sub = sub;
    c.src_w = ceil_shift(g.pre_w, sub); c.src_h = ceil_shift(g.pre_h, sub);
    const int vw = ceil_shift(g.sc_w, sub), vh = ceil_shift(g.sc_h, sub);
    if (!tts::build_axis_filter(c.src_w, vw, 1 << 14, g.alg, c.h)) return TTS_ERR_INVALID_INPUT;
    if (!tts::build_axis_filter(c.src_h, vh, 1 << 12, g.alg, c.v)) return TTS_ERR_INVALID_INPUT;
    const int content_w = g.do_pad ? (g.crop_w & ~1) : g.crop_w;   // pad's right border eats an odd last column
    c.x0 = g.crop_x >> sub; c.y0 = g.crop_y >> sub;
    c.ow = ceil_shift(content_w, sub); c.oh = ceil_shift(g.crop_h, sub);
    c.canvas_w = ceil_shift(g.out_w, sub); c.canvas_h = ceil_shift(g.out_h, sub);
    c.pad_x = g.do_pad ? g.pad_x >> sub : 0; c.pad_y = g.do_pad ? g.pad_y >> sub : 0;
    if (c.ow < 1 || c.oh < 1) return TTS_ERR_INVALID_INPUT;

    // generic tables (also what tts_plan_filter reports)
    if (!upload(c.d_hpos, c.h.pos) || !upload(c.d_htaps, c.h.taps) ||
        !upload(c.d_vpos, c.v.pos) || !upload(c.d_vtaps, c.v.taps)) return TTS_ERR_INITIALIZATION_FAILED;

    const int HT = c.h.size, VT = c.v.size;
    const int LW = HT <= 4 ? 4 : 8;
    const int R = VT <= 4 ? 4 : VT <= 6 ? 6 : 8;
    const bool strip_ok = HT <= 8 && VT <= 8 && c.src_w >= LW && monotone(c.v.pos) &&
                          !env_int("TTS_FORCE_GENERIC", 0);
    if (!strip_ok) {
        // Many taps: the column walk, if the taps fit its windows and ring and the runs fit its slot.
        c.kind = 0;
        // three instantiations, by the larger tap count of the two axes: up to 16, 32, 64 taps
        const int mt = std::max(HT, VT);
        const int NTP = mt <= 16 ? 8 : mt <= 32 ? 16 : 32, Rc = 2 * NTP;
        if (HT > 64 || VT > 64 || c.src_w < 2 * NTP || !monotone(c.v.pos) || env_int("TTS_FORCE_GENERIC", 0)) return TTS_OK;
        HTables t;                                     // (only `strips` is used by lds_dmas)
        t.strips = c.ow;
        std::vector<int> lo(static_cast<size_t>(c.ow));
        const uint32_t hoff1 = uint32_t(tab.size());
        tab.resize(tab.size() + c.ow);
        const uint32_t hcoef1 = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(c.ow) * NTP, 0);
        for (int i = 0; i < c.ow; ++i) {
            const int vx = c.x0 + i, pos = c.h.pos[vx];
            lo[i] = std::min(pos, c.src_w - 2 * NTP);           // keep the window inside the row
            int16_t slot[64] = {0};
            for (int j = 0; j < HT; ++j) slot[pos - lo[i] + j] = c.h.taps[size_t(vx) * HT + j];
            tab[hoff1 + i] = uint32_t(lo[i]);
            for (int k = 0; k < NTP; ++k) tab[hcoef1 + size_t(i) * NTP + k] = pack16(slot[2 * k], slot[2 * k + 1]);
        }
        // columns per lane: two at 8 tap pairs (2160x3840 -> 720x1280: 25 -> 41 % of peak; four measured the
        // same there and 5 % slower on 4K -> 720p, at 110 VGPRs), one above (two at 16 tap pairs made no
        // difference to the 1080p thumbnail), if the 64-strip runs fit the LDS slot
        int dmas = 0, nc = NTP == 8 ? 2 : 1;
        const int forced_nc = env_int("TTS_COL_NC", 0);
        if (forced_nc == 1 || forced_nc == 2) nc = std::min(nc, forced_nc);
        for (; nc >= 1 && !dmas; nc = dmas ? nc : nc / 2) {
            t.strips = (c.ow + nc - 1) / nc;
            std::vector<int> lo_nc(size_t(t.strips) * nc);
            for (int s = 0; s < t.strips; ++s)
                for (int j = 0; j < nc; ++j) lo_nc[size_t(s) * nc + j] = lo[std::min(std::min(s * nc, std::max(c.ow - nc, 0)) + j, c.ow - 1)];
            dmas = lds_dmas(c, t, nc, lo_nc.data(), nc, 2 * NTP);
            if (dmas) break;
        }
        if (!dmas) return TTS_OK;
        c.kind = 2; c.NTP = NTP; c.Rc = Rc; c.col_dmas = dmas; c.col_nc = nc; c.hoff1 = hoff1; c.hcoef1 = hcoef1;
        // Segment lengths on offer: the luma plane cut into 1..32 EQUAL parts (outputs here are small --
        // a ragged last segment would be most of the tail), the same list for both classes, ascending.
        std::vector<int> opts;
        for (int parts : {32, 24, 16, 12, 10, 8, 6, 5, 4, 3, 2, 1}) {
            const int sr = std::max(8, (g.crop_h + parts - 1) / parts);
            if (opts.empty() || sr > opts.back()) opts.push_back(sr);
        }
        const int forced = env_int("TTS_SEG_ROWS", 0);
        if (forced > 0) opts.assign(1, forced);
        for (int sr : opts) {
            SegTables st;
            build_seg_tables(c, Rc, sr, tab, st);
            c.segs.push_back(st);
        }
        return TTS_OK;
    }
    c.kind = 1; c.LW = LW; c.R = R;

    // Horizontal tables, one set per strip width N.  Strip s covers columns [xs, xs + N) with
    // xs = min(s * N, ow - N): the last strip is slid left over its neighbour instead of being cut
    // short (the shared columns are simply produced twice, identically), so every lane stores N
    // full bytes and the kernels need no ragged-edge path unless the plane is narrower than N.
    for (int v = 0; v < 2; ++v) {
        HTables &t = c.ht[v];
        const int N = v ? 16 : 8, NL = N / 4;
        t.strips = (c.ow + N - 1) / N;
        const int ncols = t.strips * N;
        auto column = [&](int i) { return c.x0 + std::min(std::min((i / N) * N, std::max(c.ow - N, 0)) + i % N, c.ow - 1); };
        t.hoff = uint32_t(tab.size());
        tab.resize(tab.size() + ncols);
        t.hcoef = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(ncols) * (LW / 2), 0);
        for (int i = 0; i < ncols; ++i) {
            const int vx = column(i);
            const int pos = c.h.pos[vx];
            const int lo = std::min(pos, c.src_w - LW);           // keep the window inside the row
            int16_t slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < HT; ++j) slot[pos - lo + j] = c.h.taps[size_t(vx) * HT + j];
            tab[t.hoff + i] = uint32_t(lo);
            for (int k = 0; k < LW / 2; ++k) tab[t.hcoef + size_t(i) * (LW / 2) + k] = pack16(slot[2 * k], slot[2 * k + 1]);
        }
        // grouped windows: one 8-byte window per 4 columns, taps picked by per-lane byte selectors
        t.g4 = LW == 4 && c.src_w >= 8 && !env_int("TTS_NO_G4", 0);
        std::vector<int> glo(ncols / 4);
        for (int q = 0; q < ncols / 4 && t.g4; ++q) {
            const int vx0 = column(4 * q), vx3 = column(4 * q + 3);
            glo[q] = std::min(c.h.pos[vx0], c.src_w - 8);
            if (c.h.pos[vx3] + HT - glo[q] > 8 || c.h.pos[vx0] < glo[q]) t.g4 = false;
        }
        if (!t.g4) continue;
        t.hoff4 = uint32_t(tab.size());
        tab.resize(tab.size() + ncols / 4);
        t.hsel4 = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(ncols) * 2);
        t.hcoef4 = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(ncols) * 2, 0);
        for (int i = 0; i < ncols; ++i) {
            const int vx = column(i);
            const uint32_t d = uint32_t(c.h.pos[vx] - glo[i / 4]);          // 0..4: first tap's byte in the window
            int16_t t4[4] = {0, 0, 0, 0};
            for (int j = 0; j < HT; ++j) t4[j] = c.h.taps[size_t(vx) * HT + j];
            if (i % 4 == 0) tab[t.hoff4 + i / 4] = uint32_t(glo[i / 4]);
            tab[t.hsel4 + 2 * size_t(i)]     = 0x0c000c00u | d | ((d + 1) << 16);
            tab[t.hsel4 + 2 * size_t(i) + 1] = 0x0c000c00u | (d + 2) | ((d + 3) << 16);
            tab[t.hcoef4 + 2 * size_t(i)]     = pack16(t4[0], t4[1]);
            tab[t.hcoef4 + 2 * size_t(i) + 1] = pack16(t4[2], t4[3]);
        }
        t.lds_g4 = lds_dmas(c, t, N, glo.data(), NL, 8);
    }
    c.LWlds = LW == 8 && HT <= 6 && c.src_w >= 6 ? 6 : LW;
    for (int v = 0; v < 2; ++v) {
        HTables &t = c.ht[v];
        const int N = v ? 16 : 8;
        const int ncols = t.strips * N;
        std::vector<int> lo(static_cast<size_t>(ncols));
        if (c.LWlds == 6) {
            auto column = [&](int i) { return c.x0 + std::min(std::min((i / N) * N, std::max(c.ow - N, 0)) + i % N, c.ow - 1); };
            t.hoff6 = uint32_t(tab.size());
            tab.resize(tab.size() + ncols);
            t.hcoef6 = uint32_t(tab.size());
            tab.resize(tab.size() + size_t(ncols) * 3, 0);
            for (int i = 0; i < ncols; ++i) {
                const int vx = column(i), pos = c.h.pos[vx];
                lo[i] = std::min(pos, c.src_w - 6);
                int16_t slot[6] = {0, 0, 0, 0, 0, 0};
                for (int j = 0; j < HT; ++j) slot[pos - lo[i] + j] = c.h.taps[size_t(vx) * HT + j];
                tab[t.hoff6 + i] = uint32_t(lo[i]);
                for (int k = 0; k < 3; ++k) tab[t.hcoef6 + size_t(i) * 3 + k] = pack16(slot[2 * k], slot[2 * k + 1]);
            }
        } else {
            for (int i = 0; i < ncols; ++i) lo[i] = int(tab[t.hoff + i]);
        }
        t.lds_col = lds_dmas(c, t, N, lo.data(), N, c.LWlds);
        // Grouped 12-byte windows: GC columns share one window, read as four aligned dwords -- 8 (GC = 4) or
        // 16 (GC = 2) LDS dwords per lane and row instead of 24.  Tap j of a column whose first tap is byte d
        // of the window is picked out of the dword pair (j/4, j/4 + 1): d + j - 4 (j/4) must stay <= 7, and
        // d + taps <= 12.  5-6 taps: four columns up to ratios of about 1.5 (4K -> portrait is 1.125), two
        // beyond (the 3.4x area thumbnails); 7-8 taps: two columns (1.5x .. 2x: 4K -> 1080p).
        if (c.src_w >= 12 && HT >= 5 && !env_int("TTS_NO_G4", 0) && !env_int("TTS_NO_G6", 0)) {
            auto column = [&](int i) { return c.x0 + std::min(std::min((i / N) * N, std::max(c.ow - N, 0)) + i % N, c.ow - 1); };
            const int lw = HT <= 6 ? 6 : 8;
            const int forced_gc = env_int("TTS_G12_GC", 0);
            for (int gc : {4, 2}) {
                if ((lw == 8 && gc == 4) || t.g12.lds || (forced_gc && gc != forced_gc)) continue;
                std::vector<int> glo(ncols / gc);
                bool ok = true;
                for (int q = 0; q < ncols / gc && ok; ++q) {
                    glo[q] = c.h.pos[column(gc * q)];   // not pulled back from the row's end: see the kernel's run bounds
                    for (int j = 0; j < gc && ok; ++j) {
                        const int d = c.h.pos[column(gc * q + j)] - glo[q];
                        if (d < 0 || d + HT > 12) ok = false;
                        for (int k = 0; k < HT && ok; ++k)
                            if (d + k - 4 * (k / 4) > 7) ok = false;
                    }
                }
                if (!ok) continue;
                const int lds = lds_dmas(c, t, N, glo.data(), N / gc, 12);
                if (!lds || (gc == 4 && lds != 1)) continue;            // (the 4-column kernels exist for one DMA only)
                HTables::Grouped12 &gr = t.g12;
                gr.lw = lw; gr.gc = gc; gr.lds = lds;
                gr.hoff = uint32_t(tab.size());
                tab.resize(tab.size() + ncols / gc);
                gr.hsel = uint32_t(tab.size());
                tab.resize(tab.size() + size_t(ncols) * 2);
                gr.hcoef = uint32_t(tab.size());
                tab.resize(tab.size() + size_t(ncols) * (lw / 2), 0);
                for (int i = 0; i < ncols; ++i) {
                    const int vx = column(i);
                    const uint32_t d = uint32_t(c.h.pos[vx] - glo[i / gc]);
                    int16_t t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int k = 0; k < HT; ++k) t8[k] = c.h.taps[size_t(vx) * HT + k];
                    if (i % gc == 0) tab[gr.hoff + i / gc] = uint32_t(glo[i / gc]);
                    tab[gr.hsel + 2 * size_t(i)]     = 0x0c000c00u | d | ((d + 1) << 16);
                    tab[gr.hsel + 2 * size_t(i) + 1] = 0x0c000c00u | (d + 2) | ((d + 3) << 16);
                    for (int k = 0; k < lw / 2; ++k) tab[gr.hcoef + size_t(lw / 2) * i + k] = pack16(t8[2 * k], t8[2 * k + 1]);
                }
            }
        }
    }
    std::vector<int> opts(kSegOptions, kSegOptions + kNumSegOptions);
    const int forced = env_int("TTS_SEG_ROWS", 0);
    if (forced > 0) opts.assign(1, forced);
    for (int sr : opts) {
        SegTables st;
        build_seg_tables(c, c.R, sr, tab, st);
        c.segs.push_back(st);
    }
    return TTS_OK;
}

void free_class(ClassPlan &c)
{
    (void)hipFree(c.d_hpos); (void)hipFree(c.d_vpos); (void)hipFree(c.d_htaps); (void)hipFree(c.d_vtaps);
}

typedef void (*strip_fn)(const uint8_t *, uint8_t *, const uint32_t *, const StripArgs);

// lds: LDS-DMAs per row of the LDS-staged kernel, 0 = register kernel; depth: its source rows in flight
struct StripChoice { int LW, R, N; bool g4; int lds; int depth; int g12 = 0; };   // g12: columns per 12-byte window, 0 = none

StripChoice strip_choice(const ClassPlan &c, const Knobs &kn)
{
    StripChoice ch{c.LW, c.R, 8, false, 0, 1};
    if (c.LW == 4 && c.R == 4) ch.N = kn.strip_n == 16 ? 16 : 8;
    const HTables &t = c.ht[ch.N == 16];
    ch.g4 = t.g4;
    ch.lds = t.g4 ? t.lds_g4 : t.lds_col;
    if (t.g4 && ch.lds != 1) { ch.lds = t.lds_col; if (ch.lds) ch.g4 = false; }   // grouped windows ride in one DMA only
    if (ch.lds && !ch.g4) ch.LW = c.LWlds;
    if (!ch.g4 && t.g12.lds) { ch.g4 = true; ch.g12 = t.g12.gc; ch.lds = t.g12.lds; ch.LW = t.g12.lw; }   // (LDS-staged kernel only)
    if (ch.lds && ch.g4 && ch.R == 4) ch.depth = kn.strip_d == 2 ? 2 : 1;
    return ch;
}

strip_fn pick_strip(const StripChoice &ch, const Knobs &kn)
{
    const int D = kn.strip_d;                          // rows in flight of the register kernels (tuning knob)
    if (ch.lds && ch.g4 && !ch.g12) {
        if (ch.R == 4) return ch.depth == 2 ? strip_lds_kernel<4, 4, 4, 1, 2> : strip_lds_kernel<4, 4, 4, 1, 1>;
        return ch.R == 6 ? strip_lds_kernel<4, 6, 4, 1, 1> : strip_lds_kernel<4, 8, 4, 1, 1>;
    }
    if (ch.lds && ch.g12 == 4) return ch.R == 4 ? strip_lds_kernel<6, 4, 4, 1, 1> : ch.R == 6 ? strip_lds_kernel<6, 6, 4, 1, 1> : strip_lds_kernel<6, 8, 4, 1, 1>;
    if (ch.lds && ch.g12 == 2) {
#define TTS_G2_PICK(LW_, R_) (ch.lds == 1 ? strip_lds_kernel<LW_, R_, 2, 1, 1> : strip_lds_kernel<LW_, R_, 2, 2, 1>)
        if (ch.LW == 6) return ch.R == 4 ? TTS_G2_PICK(6, 4) : ch.R == 6 ? TTS_G2_PICK(6, 6) : TTS_G2_PICK(6, 8);
        return ch.R == 4 ? TTS_G2_PICK(8, 4) : ch.R == 6 ? TTS_G2_PICK(8, 6) : TTS_G2_PICK(8, 8);
#undef TTS_G2_PICK
    }
    if (ch.lds) {
#define TTS_LDS_PICK(LW_, R_) (ch.lds == 1 ? strip_lds_kernel<LW_, R_, 0, 1, 1> : strip_lds_kernel<LW_, R_, 0, 2, 1>)
        if (ch.LW == 4) return ch.R == 4 ? TTS_LDS_PICK(4, 4) : ch.R == 6 ? TTS_LDS_PICK(4, 6) : TTS_LDS_PICK(4, 8);
        if (ch.LW == 6) return ch.R == 4 ? TTS_LDS_PICK(6, 4) : ch.R == 6 ? TTS_LDS_PICK(6, 6) : TTS_LDS_PICK(6, 8);
        return ch.R == 4 ? TTS_LDS_PICK(8, 4) : ch.R == 6 ? TTS_LDS_PICK(8, 6) : TTS_LDS_PICK(8, 8);
#undef TTS_LDS_PICK
    }
    if (ch.LW == 4 && ch.R == 4 && ch.g4) {
        if (ch.N == 16) return strip_kernel<4, 4, 16, 1, true>;
        return D == 4 ? strip_kernel<4, 4, 8, 4, true> : D == 2 ? strip_kernel<4, 4, 8, 2, true> : strip_kernel<4, 4, 8, 1, true>;
    }
    if (ch.LW == 4 && ch.R == 4) return ch.N == 16 ? strip_kernel<4, 4, 16, 1, false> : strip_kernel<4, 4, 8, 2, false>;
    if (ch.LW == 4 && ch.R == 6) return ch.g4 ? strip_kernel<4, 6, 8, 2, true> : strip_kernel<4, 6, 8, 2, false>;
    if (ch.LW == 4 && ch.R == 8) return ch.g4 ? strip_kernel<4, 8, 8, 2, true> : strip_kernel<4, 8, 8, 2, false>;
    if (ch.LW == 8 && ch.R == 4) return strip_kernel<8, 4, 8, 2, false>;
    if (ch.LW == 8 && ch.R == 6) return strip_kernel<8, 6, 8, 2, false>;
    if (ch.LW == 8 && ch.R == 8) return strip_kernel<8, 8, 8, 2, false>;
    return nullptr;
}

void fill_strip_class(const tts_plan *p, int ci, int n, int N, bool g4, int lw, int src_stride, int dst_stride,
                      const SegTables &st, int block, int paced, int planes, int plane0, StripClass &sc)
{
    const ClassPlan &c = p->cls[ci];
    const HTables &t = c.ht[N == 16];
    sc.hoff = g4 ? t.hoff4 : t.hoff; sc.hcoef = g4 ? t.hcoef4 : t.hcoef; sc.hsel = t.hsel4;
    if (!g4 && lw == 6) { sc.hoff = t.hoff6; sc.hcoef = t.hcoef6; }
    if (g4 && lw >= 6) { sc.hoff = t.g12.hoff; sc.hcoef = t.g12.hcoef; sc.hsel = t.g12.hsel; }
    sc.seginfo = st.seginfo; sc.emit = st.emit; sc.vcoef = st.vcoef;
    sc.maxrows = st.maxrows; sc.nseg = st.nseg; sc.seg_rows = st.seg_rows;
    sc.strips = t.strips;
    sc.planes = planes; sc.plane0 = plane0;
    sc.n_items = n * sc.planes * sc.strips;
    sc.ow = c.ow; sc.oh = c.oh; sc.src_stride = p->knobs.exp_stride0 ? 0 : src_stride; sc.dst_stride = dst_stride;
    sc.nblocks = uint32_t((sc.n_items + block - 1) / block) * uint32_t(st.nseg);
    sc.paced = paced;
    {   // the source plane of this class, seen from the pre-cropped origin the frame offsets point at
        const int sub = ci ? 1 : 0;
        const int plane_w = ceil_shift(p->g.in_w, sub), plane_h = ceil_shift(p->g.in_h, sub);
        sc.src_begin_rel = uint32_t(p->g.pre_y >> sub) * uint32_t(src_stride) + uint32_t(p->g.pre_x >> sub);
        sc.src_end_rel = uint32_t(plane_h - (p->g.pre_y >> sub) - 1) * uint32_t(src_stride) + uint32_t(plane_w - (p->g.pre_x >> sub));
    sc.src_row_rel = uint32_t(plane_w - (p->g.pre_x >> sub));
        sc.src_row_rel = uint32_t(plane_w - (p->g.pre_x >> sub));
    }
}

// Resident workgroups per CU of a strip kernel (VGPR- and LDS-limited); asked once per kernel.
int blocks_per_cu(strip_fn fn, int block, size_t lds_bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<std::pair<strip_fn, int>, int>> cache;
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<strip_fn, int> key(fn, block);
    for (auto &e : cache) if (e.first == key) return e.second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, lds_bytes) != hipSuccess || nb < 1) nb = 1;
    cache.emplace_back(key, nb);
    return nb;
}

// Workgroup shape of a strip launch.  The register kernels run 256 threads.  The LDS-staged kernel
// takes any multiple of 64 up to 1024 (one LDS slot per wave) and has a PACED mode for jobs that
// are bound by their writes, large enough to fill the chip with such workgroups (~100 1080p frames),
// while the output rows are not sector-aligned (1080-byte rows: a wave's
// 512-byte chunk then ends inside a 64-byte sector it shares with the next wave on 7 rows of 8):
// 1024-thread workgroups of neighbouring strips, dealt strip-major so that neighbours also share
// an XCD, whose waves meet at a barrier before every output row's store.  The halves of a shared
// sector then reach L2 within one row's time instead of whenever.  Measured (300 frames):
// 1080p->portrait 0.370 -> 0.330 ms, 720p->portrait 0.380 -> 0.327 ms; jobs that read more than they
// write (4K->portrait) or write aligned rows lose 3-10 % to the barrier and stay unpaced.
struct BlockShape { int threads; size_t lds_bytes; int paced; };
BlockShape block_shape(const tts_plan *p, int ci, const StripChoice &ch, int dst_stride, int max_items, int64_t total_items)
{
    if (!ch.lds) return {kBlock, 0, 0};
    const ClassPlan &c = p->cls[ci];
    const bool misaligned = dst_stride % 64 != 0 || c.ow % 64 != 0 || c.pad_x % 64 != 0;
    const int src_rows = c.v.pos[c.y0 + c.oh - 1] + c.v.size - c.v.pos[c.y0];
    const int src_cols = c.h.pos[c.x0 + c.ow - 1] + c.h.size - c.h.pos[c.x0];
    const bool write_bound = 2.0 * c.ow * c.oh >= 3.0 * src_rows * src_cols;     // writes >= 60 % of the traffic
    // (below ~100 strips per CU there are too few 1024-thread workgroups to keep the chip busy: 16
    // 1080p frames take 44 us paced, 29 us not; the gain starts at ~100 frames)
    int paced = misaligned && write_bound && total_items >= int64_t(96) * p->n_cus;
    if (p->knobs.paced >= 0) paced = p->knobs.paced;
    // (a paced workgroup is no larger than the biggest class needs: idle waves still walk every row)
    int t = p->knobs.lds_block > 0 ? p->knobs.lds_block : paced ? std::max(kBlock, std::min(kMaxLdsBlock, (max_items + 63) / 64 * 64)) : kBlock;
    t = std::min(std::max(t / 64 * 64, 64), kMaxLdsBlock);
    return {t, size_t(t / 64) * ch.depth * (size_t(ch.lds) * 1024 + 16), paced};
}

// Pick the row-segment length (one for both plane classes).  Long segments amortise the (taps-1)-row
// halo repeated at every segment start; short ones give enough workgroups to fill the chip and to
// keep the tail short when the last ones drain.  Measured on MI355X the time is flat from 64 to 96
// rows once there are >= ~3 workgroups per resident slot, so: the longest segment <= 96 rows that
// still leaves that many workgroups, else the shortest on offer.
void choose_segs(const tts_plan *p, const int *cis, int ncls, int n, int N, strip_fn fn, const BlockShape &bs, int min_segs, size_t out_idx[2])
{
    out_idx[0] = out_idx[1] = 0;
    const size_t nopt = p->cls[cis[0]].segs.size();
    if (nopt == 1) return;
    const double slots = double(p->n_cus) * blocks_per_cu(fn, bs.threads, bs.lds_bytes);
    size_t best = 0;
    for (size_t i = 0; i < nopt; ++i) {
        if (p->cls[cis[0]].segs[i].seg_rows > 96) break;
        double blocks = 0;
        for (int k = 0; k < ncls; ++k) {
            const ClassPlan &c = p->cls[cis[k]];
            const double sb = double((int64_t(n) * (cis[k] ? 2 : 1) * c.ht[N == 16].strips + bs.threads - 1) / bs.threads);
            blocks += sb * c.segs[i].nseg;
        }
        if (i == 0 || blocks >= (bs.paced ? 2.0 : 3.0) * slots) best = i;   // (paced workgroups are 4x as large)
    }
    for (bool ok = false; !ok && best > 0; ) {         // pointer mode + LDS-staged kernel: at least min_segs segments per plane
        ok = true;
        for (int k = 0; k < ncls; ++k) ok &= p->cls[cis[k]].segs[best].nseg >= min_segs && !(min_segs > 1 && p->cls[cis[k]].segs[best].spans_plane);
        if (!ok) --best;
    }
    out_idx[0] = out_idx[1] = best;
}

strip_fn pick_column(const ClassPlan &c)
{
#define TTS_COL_PICK(NTP_, R_, NC_) (c.col_dmas == 1 ? column_lds_kernel<NTP_, R_, 1, NC_> : column_lds_kernel<NTP_, R_, 2, NC_>)
    if (c.NTP == 8 && c.Rc == 16) return c.col_nc == 2 ? TTS_COL_PICK(8, 16, 2) : TTS_COL_PICK(8, 16, 1);
    if (c.NTP == 16 && c.Rc == 32 && c.col_nc == 1) return TTS_COL_PICK(16, 32, 1);
    if (c.NTP == 32 && c.Rc == 64 && c.col_nc == 1) return TTS_COL_PICK(32, 64, 1);
#undef TTS_COL_PICK
    return nullptr;
}

void fill_column_class(const tts_plan *p, int ci, int n, int src_stride, int dst_stride, const SegTables &st,
                       int planes, int plane0, StripClass &sc)
{
    const ClassPlan &c = p->cls[ci];
    sc = StripClass{};
    sc.hoff = c.hoff1; sc.hcoef = c.hcoef1;
    sc.seginfo = st.seginfo; sc.emit = st.emit; sc.vcoef = st.vcoef;
    sc.maxrows = st.maxrows; sc.nseg = st.nseg; sc.seg_rows = st.seg_rows;
    sc.strips = (c.ow + c.col_nc - 1) / c.col_nc;      // col_nc columns per lane
    sc.planes = planes; sc.plane0 = plane0;
    sc.n_items = n * planes * sc.strips;
    sc.ow = c.ow; sc.oh = c.oh; sc.src_stride = src_stride; sc.dst_stride = dst_stride;
    sc.nblocks = uint32_t((sc.n_items + kBlock - 1) / kBlock) * uint32_t(st.nseg);
    const int sub = ci ? 1 : 0;
    const int plane_w = ceil_shift(p->g.in_w, sub), plane_h = ceil_shift(p->g.in_h, sub);
    sc.src_begin_rel = uint32_t(p->g.pre_y >> sub) * uint32_t(src_stride) + uint32_t(p->g.pre_x >> sub);
    sc.src_end_rel = uint32_t(plane_h - (p->g.pre_y >> sub) - 1) * uint32_t(src_stride) + uint32_t(plane_w - (p->g.pre_x >> sub));
}

// Row-segment length of a column-walk launch: the longest segment on offer that leaves >= 1.5
// workgroups per resident slot (every segment start re-reads taps - 1 source rows: 24 of 120 at
// 16-row segments of the 1080p thumbnail), else the shortest; at least `min_segs` segments.
size_t choose_column_segs(const tts_plan *p, const int *cis, int ncls, int n, int planes_each[2], strip_fn fn, int min_segs)
{
    const size_t nopt = p->cls[cis[0]].segs.size();
    const double slots = double(p->n_cus) * blocks_per_cu(fn, kBlock, 0);
    size_t best = 0;
    for (size_t i = 0; i < nopt; ++i) {
        double blocks = 0;
        for (int k = 0; k < ncls; ++k) {
            const ClassPlan &c = p->cls[cis[k]];
            blocks += double((int64_t(n) * planes_each[k] * ((c.ow + c.col_nc - 1) / c.col_nc) + kBlock - 1) / kBlock) * c.segs[i].nseg;
        }
        if (i == 0 || blocks >= 1.5 * slots) best = i;
    }
    for (bool ok = false; !ok && best > 0; ) {
        ok = true;
        for (int k = 0; k < ncls; ++k) ok &= p->cls[cis[k]].segs[best].nseg >= min_segs && !(min_segs > 1 && p->cls[cis[k]].segs[best].spans_plane);
        if (!ok) --best;
    }
    return best;
}

// One launch set: frames [0, n) addressed by `addr` relative to (src, dst).
// `only_plane` >= 0 restricts the launch set to that one plane (per-frame pointers whose planes lie
// too far apart for one base + 32-bit offsets are processed plane by plane).
int launch_all(const tts_plan *p, const uint8_t *src, uint8_t *dst, int n, const int src_stride[2],
               const int dst_stride[2], uint64_t src_span, uint64_t dst_span, const FrameAddr &addr, hipStream_t st,
               int only_plane = -1)
{
    // the plain kernels index frames by grid.z (at most 65535 planes a launch)
    auto z_launches = [&](int planes, auto &&launch) {
        const int per = 65535 / planes;
        for (int f0 = 0; f0 < n; f0 += per) launch(f0, std::min(per, n - f0));
    };
    auto active  = [&](int ci) { return only_plane < 0 || ci == (only_plane ? 1 : 0); };
    auto nplanes = [&](int ci) { return only_plane < 0 ? (ci ? 2 : 1) : 1; };
    auto plane0  = [&](int ci) { return only_plane < 0 ? (ci ? 1 : 0) : only_plane; };
    if (p->g.do_pad) {
        for (int ci = 0; ci < 2; ++ci) {
            if (!active(ci)) continue;
            const ClassPlan &c = p->cls[ci];
            FillArgs f{};
            f.dst = dst; f.w = c.canvas_w; f.h = c.canvas_h; f.stride = dst_stride[ci];
            f.px = c.pad_x; f.py = c.pad_y; f.pw = c.ow; f.ph = c.oh;
            f.planes = nplanes(ci); f.plane0 = plane0(ci);
            f.fill[0] = p->g.fill_y; f.fill[1] = p->g.fill_u; f.fill[2] = p->g.fill_v;
            f.origin = uint32_t(c.pad_y) * uint32_t(dst_stride[ci]) + uint32_t(c.pad_x);
            f.addr = addr;
            z_launches(f.planes, [&](int f0, int nf) {
                f.frame0 = f0;
                dim3 grid((c.canvas_w + 255) / 256, (c.canvas_h + 3) / 4, nf * f.planes);
                hipLaunchKernelGGL(fill_kernel, grid, dim3(kBlock), 0, st, f);
            });
        }
    }

    // strip classes: one launch when both classes run the same instantiation, else one each
    int strip_cls[2], ns = 0;
    for (int ci = 0; ci < 2; ++ci) if (p->cls[ci].kind == 1 && active(ci)) strip_cls[ns++] = ci;
    bool merged = false;
    if (ns == 2 && !p->knobs.no_merge) {
        const StripChoice a0 = strip_choice(p->cls[0], p->knobs), a1 = strip_choice(p->cls[1], p->knobs);
        merged = a0.LW == a1.LW && a0.R == a1.R && a0.N == a1.N && a0.g4 == a1.g4 && a0.g12 == a1.g12 && a0.lds == a1.lds && a0.depth == a1.depth;
    }
    for (int k = 0; k < ns; ++k) {
        StripArgs a{};
        a.src_span = src_span; a.dst_span = dst_span; a.addr = addr;
        // timing experiments (outputs are wrong): a zero-record descriptor makes the range check drop that
        // buffer's traffic while the instruction stream stays
        if (p->knobs.exp_nosrc) a.src_span = 0;
        if (p->knobs.exp_srcpitch0) a.addr.src_pitch = 0;    // every frame reads frame 0: source stays in L2
        if (p->knobs.exp_dstpitch0) a.addr.dst_pitch = 0;    // every frame writes frame 0: output stays in L2
        if (p->knobs.exp_nodst) a.dst_span = 0;
        const int ci = strip_cls[k];
        StripChoice ch = strip_choice(p->cls[ci], p->knobs);
        // Per-frame plane pointers: a segment of the LDS-staged kernel must not hold both the first and
        // the last source row of a plane (see the kernel): planes whose shortest segments still do -- a
        // single segment, or so many vertical taps that the first segment reaches the last row -- take
        // the register kernel; others are cut into segments that do not (choose_segs).
        // (source planes under 48 bytes a row or 32 rows likewise: a run slid down from the plane's end
        // must still begin inside the plane)
        bool short_plane = false;
        for (int j = 0; j < ns; ++j) {
            const ClassPlan &cj = p->cls[strip_cls[j]];
            short_plane |= cj.segs[0].nseg < 2 || cj.segs[0].spans_plane || cj.src_w < 48 || cj.src_h < 32;
        }
        if (src_span < 4096u || (addr.use_table && short_plane)) { ch.lds = 0; ch.depth = 1; ch.LW = p->cls[ci].LW; if (ch.g12) { ch.g4 = false; ch.g12 = 0; } }
        strip_fn fn = pick_strip(ch, p->knobs);
        if (!fn) return TTS_ERR_UNSUPPORTED;
        size_t si[2];
        int max_items = 0;
        int64_t total_items = 0;
        for (int j = 0; j < ns; ++j) {
            const int cj = strip_cls[j];
            if (!(merged || cj == ci)) continue;
            const int items = n * nplanes(cj) * p->cls[cj].ht[ch.N == 16].strips;
            max_items = std::max(max_items, items); total_items += items;
        }
        const BlockShape bs = block_shape(p, ci, ch, dst_stride[ci], max_items, total_items);
        const int min_segs = addr.use_table && ch.lds ? 2 : 1;
        if (merged) {
            choose_segs(p, strip_cls, 2, n, ch.N, fn, bs, min_segs, si);
            for (int j = 0; j < 2; ++j)
                fill_strip_class(p, j, n, ch.N, ch.g4, ch.LW, src_stride[j], dst_stride[j], p->cls[j].segs[si[j]], bs.threads, bs.paced, nplanes(j), plane0(j), a.c[j]);
        } else {
            // a single class rides in slot 0; slot 1 is empty
            choose_segs(p, &ci, 1, n, ch.N, fn, bs, min_segs, si);
            fill_strip_class(p, ci, n, ch.N, ch.g4, ch.LW, src_stride[ci], dst_stride[ci], p->cls[ci].segs[si[0]], bs.threads, bs.paced, nplanes(ci), plane0(ci), a.c[0]);
            a.c[1] = a.c[0]; a.c[1].nblocks = 0;
        }
        const uint32_t blocks = a.c[0].nblocks + (merged ? a.c[1].nblocks : 0);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(bs.threads), bs.lds_bytes, st, src, dst, p->d_tab, a);
        if (merged) break;
    }

    // column classes (9..64 taps): one launch when both classes run the same instantiation
    bool to_generic[2] = {false, false};               // column classes this launch hands to the generic kernel instead
    {
        int col_cls[2], nc = 0;
        for (int ci = 0; ci < 2; ++ci) {
            const ClassPlan &c = p->cls[ci];
            if (c.kind != 2 || !active(ci)) continue;
            // (per-frame plane pointers: see the strip classes above; tiny sources likewise)
            if (src_span < 4096u || (addr.use_table && (c.segs[0].nseg < 2 || c.segs[0].spans_plane || c.src_w < 48 || c.src_h < 32))) to_generic[ci] = true;
            else col_cls[nc++] = ci;
        }
        const bool cmerged = nc == 2 && !p->knobs.no_merge && pick_column(p->cls[0]) == pick_column(p->cls[1]);
        for (int k = 0; k < nc; ++k) {
            const int ci = col_cls[k];
            strip_fn fn = pick_column(p->cls[ci]);
            if (!fn) return TTS_ERR_UNSUPPORTED;
            StripArgs a{};
            a.src_span = src_span; a.dst_span = dst_span; a.addr = addr;
            const int min_segs = addr.use_table ? 2 : 1;
            if (cmerged) {
                int planes_each[2] = {nplanes(0), nplanes(1)};
                const size_t si = choose_column_segs(p, col_cls, 2, n, planes_each, fn, min_segs);
                for (int j = 0; j < 2; ++j)
                    fill_column_class(p, j, n, src_stride[j], dst_stride[j], p->cls[j].segs[si], nplanes(j), plane0(j), a.c[j]);
            } else {
                int planes_each[2] = {nplanes(ci), 0};
                const size_t si = choose_column_segs(p, &ci, 1, n, planes_each, fn, min_segs);
                fill_column_class(p, ci, n, src_stride[ci], dst_stride[ci], p->cls[ci].segs[si], nplanes(ci), plane0(ci), a.c[0]);
                a.c[1] = a.c[0]; a.c[1].nblocks = 0;
            }
            const uint32_t blocks = a.c[0].nblocks + (cmerged ? a.c[1].nblocks : 0);
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(kBlock), 0, st, src, dst, p->d_tab, a);
            if (cmerged) break;
        }
    }

    for (int ci = 0; ci < 2; ++ci) {
        const ClassPlan &c = p->cls[ci];
        if (!(c.kind == 0 || to_generic[ci]) || !active(ci)) continue;
        GenericArgs a{};
        a.src = src; a.dst = dst;
        a.hpos = c.d_hpos; a.htaps = c.d_htaps; a.vpos = c.d_vpos; a.vtaps = c.d_vtaps;
        a.ht = c.h.size; a.vt = c.v.size;
        a.x0 = c.x0; a.y0 = c.y0; a.ow = c.ow; a.oh = c.oh;
        a.src_stride = src_stride[ci]; a.dst_stride = dst_stride[ci];
        a.planes = nplanes(ci); a.plane0 = plane0(ci);
        a.addr = addr;
        z_launches(a.planes, [&](int f0, int nf) {
            a.frame0 = f0;
            dim3 grid((c.ow + 63) / 64, (c.oh + 3) / 4, nf * a.planes);
            hipLaunchKernelGGL(generic_kernel, grid, dim3(kBlock), 0, st, a);
        });
    }
    if (p->g.out_range != TTS_RANGE_KEEP) {
        for (int ci = 0; ci < 2; ++ci) {
            if (!active(ci)) continue;
            const ClassPlan &c = p->cls[ci];
            RangeArgs r{};
            r.dst = dst; r.w = c.canvas_w; r.h = c.canvas_h; r.stride = dst_stride[ci];
            r.planes = nplanes(ci); r.plane0 = plane0(ci);
            r.origin = uint32_t(c.pad_y) * uint32_t(dst_stride[ci]) + uint32_t(c.pad_x);
            r.addr = addr;
            tts::build_range_lut(p->g.out_range, ci != 0, r.lut);
            z_launches(r.planes, [&](int f0, int nf) {
                r.frame0 = f0;
                dim3 grid((c.canvas_w + 255) / 256, (c.canvas_h + 3) / 4, nf * r.planes);
                hipLaunchKernelGGL(range_kernel, grid, dim3(kBlock), 0, st, r);
            });
        }
    }
    return hipGetLastError() == hipSuccess ? TTS_OK : TTS_ERR_PROCESSING_FAILED;
}

}  // namespace

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------

extern "C" {

int tts_parse_vf(const char *vf, int iw, int ih, tts_geom *out)
{
    if (!out) return TTS_ERR_INVALID_INPUT;
    std::string err;
    tts_geom g;
    const int rc = tts::parse_vf(vf, iw, ih, g, err);
    g_last_error = err;
    if (rc == TTS_OK) *out = g;
    return rc;
}

int tts_plan_create(const tts_geom *g, tts_plan **out)
{
    if (!g || !out) return TTS_ERR_INVALID_INPUT;
    *out = nullptr;
    int rc = tts::check_geom(*g);
    if (rc) return rc;
    int dev = 0;
    const hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) { g_last_error = std::string("hipGetDevice: ") + hipGetErrorString(he); return TTS_ERR_INITIALIZATION_FAILED; }
    tts_plan *p = new (std::nothrow) tts_plan();
    if (!p) return TTS_ERR_INITIALIZATION_FAILED;
    p->g = *g; p->device = dev;
    p->knobs.read();
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) p->n_cus = cus;
    std::vector<uint32_t> tab;
    for (int ci = 0; ci < 2 && rc == TTS_OK; ++ci) rc = build_class(*g, ci, p->cls[ci], tab);
    if (rc == TTS_OK && !upload(p->d_tab, tab)) rc = TTS_ERR_INITIALIZATION_FAILED;
    if (rc) {
        if (rc == TTS_ERR_INITIALIZATION_FAILED) g_last_error = std::string("table upload: ") + hipGetErrorString(hipGetLastError());
        tts_plan_destroy(p);
        return rc;
    }
    *out = p;
    return TTS_OK;
}

void tts_plan_destroy(tts_plan *p)
{
    if (!p) return;
    free_class(p->cls[0]); free_class(p->cls[1]);
    (void)hipFree(p->d_tab);
    delete p;
}

int tts_plan_geom(const tts_plan *p, tts_geom *out)
{
    if (!p || !out) return TTS_ERR_INVALID_INPUT;
    *out = p->g;
    return TTS_OK;
}

int tts_plan_filter(const tts_plan *p, int plane_class, int axis, int *taps, int *n_out, int16_t *coef, int32_t *pos)
{
    if (!p || plane_class < 0 || plane_class > 1 || axis < 0 || axis > 1) return TTS_ERR_INVALID_INPUT;
    const tts::AxisFilter &f = axis == 0 ? p->cls[plane_class].h : p->cls[plane_class].v;
    if (taps) *taps = f.size;
    if (n_out) *n_out = f.dst;
    if (coef) std::memcpy(coef, f.taps.data(), f.taps.size() * sizeof(int16_t));
    if (pos) std::memcpy(pos, f.pos.data(), f.pos.size() * sizeof(int32_t));
    return TTS_OK;
}

int tts_build_filter(int src, int dst, int one, int alg, int *taps, int16_t *coef, int32_t *pos)
{
    if (src < 1 || dst < 1 || one < 1 || alg < TTS_ALG_BICUBIC || alg > TTS_ALG_BILINEAR) return TTS_ERR_INVALID_INPUT;
    tts::AxisFilter f;
    if (!tts::build_axis_filter(src, dst, one, alg, f)) return TTS_ERR_INVALID_INPUT;
    if (taps) *taps = f.size;
    if (coef) std::memcpy(coef, f.taps.data(), f.taps.size() * sizeof(int16_t));
    if (pos) std::memcpy(pos, f.pos.data(), f.pos.size() * sizeof(int32_t));
    return TTS_OK;
}

int tts_plan_kernel_kind(const tts_plan *p, int plane_class)
{
    if (!p || plane_class < 0 || plane_class > 1) return TTS_ERR_INVALID_INPUT;
    const ClassPlan &c = p->cls[plane_class];
    if (c.kind == 0) return 0;
    if (c.kind == 2) return 3 + c.col_dmas;
    return 1 + strip_choice(c, p->knobs).lds;
}

const char *tts_plan_kernel_name(const tts_plan *p, int plane_class)
{
    static thread_local char buf[96];
    if (!p || plane_class < 0 || plane_class > 1) return "";
    const ClassPlan &c = p->cls[plane_class];
    if (c.kind == 0) std::snprintf(buf, sizeof buf, "generic");
    else if (c.kind == 2) std::snprintf(buf, sizeof buf, "column_lds<%d,%d,%d,%d>", c.NTP, c.Rc, c.col_dmas, c.col_nc);
    else {
        const StripChoice ch = strip_choice(c, p->knobs);
        if (ch.lds) std::snprintf(buf, sizeof buf, "strip_lds<%d,%d,%s,%d,%d>", ch.LW, ch.R, ch.g12 == 2 ? "grouped2" : ch.g4 ? "grouped" : "percolumn", ch.lds, ch.depth);
        else std::snprintf(buf, sizeof buf, "strip<%d,%d,%d,%s>", ch.LW, ch.R, ch.N, ch.g4 ? "grouped" : "percolumn");
    }
    return buf;
}

size_t tts_frame_bytes(int w, int h)
{
    if (w < 1 || h < 1) return 0;
    return size_t(w) * h + 2 * size_t((w + 1) / 2) * ((h + 1) / 2);
}

int tts_run_batch(const tts_plan *p, const void *src, size_t src_pitch, void *dst, size_t dst_pitch,
                  int n_frames, void *hip_stream)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    const tts_geom &g = p->g;
    const size_t sbytes = tts_frame_bytes(g.in_w, g.in_h), dbytes = tts_frame_bytes(g.out_w, g.out_h);
    if (src_pitch < sbytes || dst_pitch < dbytes) return TTS_ERR_INVALID_INPUT;
    if (src_pitch >= (size_t(1) << 31) || dst_pitch >= (size_t(1) << 31)) return TTS_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);

    FrameAddr addr{};
    addr.src_pitch = uint32_t(src_pitch); addr.dst_pitch = uint32_t(dst_pitch);
    int sstride[2], dstride[2];
    uint32_t splane = 0, dplane = 0;
    for (int k = 0; k < 3; ++k) {
        const int ci = k ? 1 : 0;
        const ClassPlan &c = p->cls[ci];
        sstride[ci] = ceil_shift(g.in_w, ci); dstride[ci] = ceil_shift(g.out_w, ci);
        addr.src_plane[k] = splane + uint32_t(g.pre_y >> ci) * sstride[ci] + uint32_t(g.pre_x >> ci);
        addr.dst_plane[k] = dplane + uint32_t(c.pad_y) * dstride[ci] + uint32_t(c.pad_x);
        splane += uint32_t(sstride[ci]) * ceil_shift(g.in_h, ci);
        dplane += uint32_t(dstride[ci]) * ceil_shift(g.out_h, ci);
    }
    // Lane offsets are 32-bit and count from the first frame of the lane's WAVE (frame_offsets), so a
    // launch may cover any number of frames as long as the frames one wave can touch stay within
    // kMaxSpan: a wave holds 64 items, a frame at least `ipf` of them.  (Pitches so large that they do
    // not -- hundreds of megabytes between frames -- fall back to launches whose whole span fits.)
    if (sbytes > kMaxSpan || dbytes > kMaxSpan) return TTS_ERR_UNSUPPORTED;
    const int ipf = std::max(1, (std::min(p->cls[0].ow, p->cls[1].ow) + 15) / 16);
    const size_t wave_frames = size_t(63 / ipf + 2);
    const bool any_size = wave_frames * src_pitch + sbytes <= kMaxSpan && wave_frames * dst_pitch + dbytes <= kMaxSpan && !p->knobs.span_chunks;
    // items and workgroups are counted in 32 bits: at most 2^30 output columns a launch
    const size_t cols_per_frame = size_t(p->cls[0].ow) + 2 * size_t(p->cls[1].ow);
    const size_t max_frames = std::max<size_t>(1, (size_t(1) << 30) / std::max<size_t>(cols_per_frame, 1));
    const size_t by_span = 1 + std::min((kMaxSpan - sbytes) / src_pitch, (kMaxSpan - dbytes) / dst_pitch);
    const int chunk = int(std::min(any_size ? max_frames : std::min(by_span, max_frames), size_t(1) << 30));
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int n = std::min(chunk, n_frames - f0);
        const uint8_t *s = static_cast<const uint8_t *>(src) + size_t(f0) * src_pitch;
        uint8_t *d = static_cast<uint8_t *>(dst) + size_t(f0) * dst_pitch;
        const uint64_t sspan = uint64_t(n - 1) * src_pitch + sbytes, dspan = uint64_t(n - 1) * dst_pitch + dbytes;
        addr.wave_base = sspan > kMaxSpan || dspan > kMaxSpan;
        const int rc = launch_all(p, s, d, n, sstride, dstride, sspan, dspan, addr, st);
        if (rc) return rc;
    }
    return TTS_OK;
}

int tts_run(const tts_plan *p, const tts_frame *src, const tts_frame *dst, int n_frames, void *hip_stream)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    const tts_geom &g = p->g;
    for (int i = 0; i < n_frames; ++i) {
        if (src[i].mem != TTS_MEM_DEVICE || dst[i].mem != TTS_MEM_DEVICE) return TTS_ERR_INVALID_INPUT;
        if (src[i].width != g.in_w || src[i].height != g.in_h || dst[i].width != g.out_w || dst[i].height != g.out_h) return TTS_ERR_INVALID_INPUT;
        for (int k = 0; k < 3; ++k) {
            if (!src[i].plane[k] || !dst[i].plane[k]) return TTS_ERR_INVALID_INPUT;
            if (src[i].stride[k] < ceil_shift(g.in_w, k ? 1 : 0) || dst[i].stride[k] < ceil_shift(g.out_w, k ? 1 : 0)) return TTS_ERR_INVALID_INPUT;
        }
        if (src[i].stride[1] != src[i].stride[2] || dst[i].stride[1] != dst[i].stride[2]) return TTS_ERR_UNSUPPORTED;
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // Frames are grouped so that every plane of a group lies within kMaxSpan of a common base (32-bit
    // lane offsets).  `only` < 0: the three planes of a frame share the base (one launch set for the
    // group); `only` = k: plane k alone -- the fallback for frames whose planes lie further apart
    // (each plane its own far-away allocation).
    auto run_groups = [&](int first, int count, int only) -> int {
        const int k0 = only < 0 ? 0 : only, k1 = only < 0 ? 3 : only + 1;
        int f0 = first;
        while (f0 < first + count) {
            uintptr_t smin = UINTPTR_MAX, dmin = UINTPTR_MAX, smax = 0, dmax = 0;   // [min start, max end) over the group's planes
            int n = 0;
            while (f0 + n < first + count && n < kMaxTableFrames) {
                const tts_frame &s = src[f0 + n], &d = dst[f0 + n];
                if (n && (s.stride[0] != src[f0].stride[0] || s.stride[1] != src[f0].stride[1] ||
                          d.stride[0] != dst[f0].stride[0] || d.stride[1] != dst[f0].stride[1])) break;
                uintptr_t a = smin, b = smax, c2 = dmin, e = dmax;
                for (int k = k0; k < k1; ++k) {
                    const int ci = k ? 1 : 0;
                    const size_t sext = size_t(s.stride[k]) * (ceil_shift(g.in_h, ci) - 1) + ceil_shift(g.in_w, ci);
                    const size_t dext = size_t(d.stride[k]) * (ceil_shift(g.out_h, ci) - 1) + ceil_shift(g.out_w, ci);
                    a = std::min(a, uintptr_t(s.plane[k])); b = std::max(b, uintptr_t(s.plane[k]) + sext);
                    c2 = std::min(c2, uintptr_t(d.plane[k])); e = std::max(e, uintptr_t(d.plane[k]) + dext);
                }
                if (b - a > kMaxSpan || e - c2 > kMaxSpan) break;
                smin = a; smax = b; dmin = c2; dmax = e;
                ++n;
            }
            if (n == 0) return TTS_ERR_UNSUPPORTED;     // this frame does not fit one base (caller falls back or gives up)
            FrameAddr addr{};
            addr.use_table = 1;
            const int sstride[2] = {src[f0].stride[0], src[f0].stride[1]};
            const int dstride[2] = {dst[f0].stride[0], dst[f0].stride[1]};
            for (int i = 0; i < n; ++i)
                for (int k = k0; k < k1; ++k) {
                    const int ci = k ? 1 : 0;
                    const ClassPlan &c = p->cls[ci];
                    const size_t so = size_t(uintptr_t(src[f0 + i].plane[k]) - smin), dof = size_t(uintptr_t(dst[f0 + i].plane[k]) - dmin);
                    addr.src_tab[k][i] = uint32_t(so) + uint32_t(g.pre_y >> ci) * sstride[ci] + uint32_t(g.pre_x >> ci);
                    addr.dst_tab[k][i] = uint32_t(dof) + uint32_t(c.pad_y) * dstride[ci] + uint32_t(c.pad_x);
                }
            const int rc = launch_all(p, reinterpret_cast<const uint8_t *>(smin), reinterpret_cast<uint8_t *>(dmin), n,
                                      sstride, dstride, uint32_t(smax - smin), uint32_t(dmax - dmin), addr, st, only);
            if (rc) return rc;
            f0 += n;
        }
        return TTS_OK;
    };
    int rc = p->knobs.planewise ? TTS_ERR_UNSUPPORTED : run_groups(0, n_frames, -1);
    if (rc == TTS_ERR_UNSUPPORTED) {
        // some frame's planes are spread over more than kMaxSpan: go plane by plane (a repeat of the
        // frames already done above is harmless -- same inputs, same outputs)
        for (int k = 0; k < 3 && (rc == TTS_ERR_UNSUPPORTED || rc == TTS_OK); ++k) rc = run_groups(0, n_frames, k);
    }
    return rc;
}

int tts_run_host(const tts_plan *p, const void *src, void *dst, int n_frames)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    const size_t sb = tts_frame_bytes(p->g.in_w, p->g.in_h), db = tts_frame_bytes(p->g.out_w, p->g.out_h);
    // The clip goes through in chunks over two slots (a source and a destination buffer in HBM each) on
    // two streams: chunk i + 1 is on its way up while chunk i is scaled and sent back, both directions of
    // the link busy, device memory bounded whatever the clip's length.  Buffers the caller has pinned
    // (hipHostMalloc / hipHostRegister) move by DMA at link speed -- 10,100 frames/s of 1080p -> portrait,
    // 31.6 GB/s each way at once, against 8,900 without the overlap; pageable ones are staged by the
    // runtime and run at host-memory speed (2,400 - 3,600 frames/s; registering them here for the call
    // cost more than it saved, so that is left to the caller).
    const size_t chunk_bytes = size_t(std::max(1, env_int("TTS_HOST_CHUNK_MB", 128))) << 20;
    const int per = int(std::max<size_t>(1, std::min<size_t>(size_t(n_frames), chunk_bytes / (sb + db))));
    const int nchunks = (n_frames + per - 1) / per;
    const int slots = std::min(2, nchunks);
    void *ds[2] = {nullptr, nullptr}, *dd[2] = {nullptr, nullptr};
    hipStream_t st[2] = {nullptr, nullptr};
    int rc = TTS_OK;
    for (int s = 0; s < slots && !rc; ++s)
        if (hipMalloc(&ds[s], sb * per) != hipSuccess || hipMalloc(&dd[s], db * per) != hipSuccess ||
            hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking) != hipSuccess) rc = TTS_ERR_INITIALIZATION_FAILED;
    for (int i = 0; i < nchunks && !rc; ++i) {
        const int s = i % slots, f0 = i * per, cnt = std::min(per, n_frames - f0);
        const uint8_t *hs = static_cast<const uint8_t *>(src) + size_t(f0) * sb;
        uint8_t *hd = static_cast<uint8_t *>(dst) + size_t(f0) * db;
        if (hipMemcpyAsync(ds[s], hs, sb * cnt, hipMemcpyHostToDevice, st[s]) != hipSuccess) { rc = TTS_ERR_PROCESSING_FAILED; break; }
        rc = tts_run_batch(p, ds[s], sb, dd[s], db, cnt, st[s]);
        if (!rc && hipMemcpyAsync(hd, dd[s], db * cnt, hipMemcpyDeviceToHost, st[s]) != hipSuccess) rc = TTS_ERR_PROCESSING_FAILED;
    }
    for (int s = 0; s < slots; ++s)
        if (st[s] && hipStreamSynchronize(st[s]) != hipSuccess && !rc) rc = TTS_ERR_PROCESSING_FAILED;
    for (int s = 0; s < slots; ++s) {
        if (st[s]) (void)hipStreamDestroy(st[s]);
        (void)hipFree(ds[s]); (void)hipFree(dd[s]);
    }
    return rc;
}

const char *tts_strerror(int status)
{
    switch (status) {
    case TTS_OK: return "ok";
    case TTS_ERR_INVALID_INPUT: return "INVALID_INPUT";
    case TTS_ERR_INVALID_FORMAT: return "INVALID_FORMAT";
    case TTS_ERR_INITIALIZATION_FAILED: return "INITIALIZATION_FAILED";
    case TTS_ERR_PROCESSING_FAILED: return "PROCESSING_FAILED";
    case TTS_ERR_UNSUPPORTED: return "UNSUPPORTED";
    default: return "unknown status";
    }
}

const char *tts_last_error(void) { return g_last_error.c_str(); }

const char *tts_version(void) { return "ttscale 0.1 (gfx950)"; }

}  // extern "C"
