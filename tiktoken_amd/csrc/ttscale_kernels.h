// ttscale_kernels.h -- device side of libttscale: the gfx950 kernels and the argument structs they take.
// Included by ttscale.hip only (one translation unit; the host planner in ttscale_plan.h picks kernel
// instantiations by address).  DESIGN.md section 4 describes every kernel; in short:
//   strip_lds_kernel   <= 8 taps per axis: a lane owns 8 adjacent output columns and walks down a row
//                      segment; source rows arrive by LDS-DMA, the 15-bit intermediate lives in a register ring
//   strip_kernel       the same walk with register gathers (sources too small or too ragged for the LDS slot)
//   column_lds_kernel  9..64 taps per axis (the reference's own thumbnail chain): 1-2 columns per lane
//   generic_kernel     any tap count, one lane per output sample
//   fill_kernel, range_kernel   pad colour; limited <-> full range table
#pragma once
#include "ttscale_device.h"
namespace {

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

// The range step of `format=yuvj420p` (or back) as a closed form per plane class, applied to the finished
// byte in the scaler's store epilogue: clip_u8((p * mul + add) >> shift) -- what swscale's second, unscaled
// pass computes (p << 7, lum/chrRangeToJpeg_c, (v + 64) >> 7), folded (the host checks it against
// build_range_lut for all 256 values before a plan uses it).
struct RangeStep { int32_t on; int32_t mul[2], add[2], shift[2]; };
// The range step of a context that converts the range ITSELF (`scale=...,format=yuvj420p`): swscale.c lumRangeToJpeg_c & co.
// on the 15-bit rows between the two passes, (min(v, cap) * mul + add) >> shift per plane class (column_lds_kernel<..., RNG>)
struct RangeStep15 { int32_t cap[2], mul[2], add[2], shift[2]; };

// The pad border of one plane class, written by extra workgroups of the SAME launch (they follow the scaler's
// workgroups in the logical block order).  Only the border is tiled: the full-width rows above and below the
// picture (`chunks` x `rowgroups` workgroups per plane), then the columns left and right of it over the picture's
// rows (`chunks_sd` x `rgs_sd`); a workgroup takes blockDim / 64 * kPadRowsPerWave rows x 256 columns of its region
// and stores the fill byte (already range-mapped).  (Tiling the whole canvas and leaving where a tile held no
// border cost a large canvas thousands of idle workgroups: 1080p letterboxed into 1920x1280 ran 2.5x slower.)
struct PadClass {
    int32_t w, h, stride;              // canvas of this plane class
    int32_t px, py, pw, ph;            // picture rectangle that is left alone
    int32_t planes, plane0;
    int32_t chunks, rowgroups;         // rows above / below the picture: 256-column chunks, row groups
    int32_t chunks_sd, rgs_sd;         // columns left / right of it
    uint32_t origin;                   // bytes from the canvas origin to the picture origin (frame offsets point at the picture)
    uint32_t nblocks;
    int32_t fill[3];
};

struct StripArgs {
    StripClass c[2];
    uint64_t src_span, dst_span;   // bytes reachable from the base pointers (buffer range check)
    RangeStep range;
    PadClass pad[2];
    uint32_t scaler_blocks;        // logical blocks [0, scaler_blocks) scale, the rest fill pad borders
    // strip_lds_kernel as a work queue (TTS_PERSIST, an experiment: DESIGN.md 7 finding 2): the launch holds grid_scaler
    // (<= scaler_blocks, a multiple of 8) scaler workgroups; one that has finished logical block b takes the next of ITS
    // XCD's blocks (b = x mod 8 stays on XCD x, whose run of logical blocks xcd_remap keeps contiguous) from queue[x].
    // queue == nullptr: one workgroup per logical block, grid_scaler == scaler_blocks.
    uint32_t grid_scaler;
    uint32_t *queue;
    RangeStep15 r15;               // RNG instantiations only (strip_lds_kernel, column_lds_kernel)
    int32_t  no_union;             // strip_lds_kernel, two columns a 12-byte window: never read a lane's four windows as one run (A/B knob TTS_NO_UNION)
    int32_t  y16;                  // column walk only: class 0 (luma) stores its UNCLIPPED vertical sums as int16 (2 bytes a column): the
                                   // intermediate of the packed-RGB thumbnail path (ttscale_rgb.hip rgb_scratch_convert_kernel)
    uint64_t *dbg;                 // -DTTS_EXPERIMENTS builds: (start, end) clock of every workgroup of strip_lds_kernel; else null, unused
    FrameAddr addr;
};

constexpr int kPadRowsPerWave = 8;     // canvas rows a wave of a pad workgroup fills (rowgroups of blockDim / 64 * 8 rows)
__device__ __forceinline__ void pad_fill_blocks(const StripArgs &a, uint8_t *dst, uint32_t fb)
{
    const int k = fb >= a.pad[0].nblocks ? 1 : 0;
    if (k) fb -= a.pad[0].nblocks;
    const PadClass &c = a.pad[k];
    const uint32_t tb = uint32_t(c.chunks) * uint32_t(c.rowgroups), per_fp = tb + uint32_t(c.chunks_sd) * uint32_t(c.rgs_sd);
    const int fp = int(fb / per_fp);
    uint32_t b = fb % per_fp;
    const bool side = b >= tb;                                     // workgroup-uniform
    if (side) b -= tb;
    const uint32_t nch = uint32_t(side ? c.chunks_sd : c.chunks);
    const int chunk = int(b % nch), rg = int(b / nch);
    // the region as a rectangle of its own: `vw` x `vh`, mapped back to canvas columns and rows around the picture
    const int vw = side ? c.w - c.pw : c.w, vh = side ? c.ph : c.h - c.ph;
    const int u0 = chunk * 256 + int(threadIdx.x & 63) * 4;
    if (u0 >= vw) return;
    const int plane = c.plane0 + fp % c.planes, frame = fp / c.planes;
    const size_t doff = (a.addr.use_table ? size_t(a.addr.dst_tab[plane][frame])
                                          : size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane]) - c.origin;
    const uint8_t f = uint8_t(c.fill[plane]);
    const int waves = int(blockDim.x >> 6);
    for (int it = 0; it < kPadRowsPerWave; ++it) {
        const int v = (rg * kPadRowsPerWave + it) * waves + int(threadIdx.x >> 6);
        if (v >= vh) return;
        const int y = side ? c.py + v : (v < c.py ? v : v + c.ph);
        uint8_t *row = dst + doff + size_t(y) * size_t(c.stride);
        for (int u = u0; u < min(u0 + 4, vw); ++u) row[side ? (u < c.px ? u : u + c.pw) : u] = f;
    }
}

// the range step on four packed bytes ((p * mul + add) >> shift, clipped by the pack)
__device__ __forceinline__ uint32_t range_bytes4(uint32_t o, int mul, int add, int shift)
{
    const int v0 = __mul24(int(o & 255u), mul) + add, v1 = __mul24(int((o >> 8) & 255u), mul) + add;
    const int v2 = __mul24(int((o >> 16) & 255u), mul) + add, v3 = __mul24(int(o >> 24), mul) + add;
    return pack4_by(v0, v1, v2, v3, shift);                  // (>> shift and the clip in the pack)
}

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

    // (pad workgroups come last in dispatch order and are left out of the XCD remap: remapped with the
    // rest they would fill whole XCDs by themselves and leave the scaler the others -- 4.5x slower thumbnails)
    if (blockIdx.x >= a.scaler_blocks) { pad_fill_blocks(a, dst, blockIdx.x - a.scaler_blocks); return; }
    uint32_t lb = xcd_remap(blockIdx.x, a.scaler_blocks);
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
                if (a.range.on) {                               // wave-uniform
#pragma unroll
                    for (int q = 0; q < N / 4; ++q) o[q] = range_bytes4(o[q], a.range.mul[ci], a.range.add[ci], a.range.shift[ci]);
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

// LW: window bytes per column (4, 6, 8 for <= 4, 5-6, 7-8 horizontal taps); R: ring rows (>= vertical
// taps); GC: columns that share one LDS window (0 = none, see the static_assert below; "G4" in the body
// means GC != 0); NDMA: 1 KiB LDS-DMAs per source row (the wave's runs hold <= 64 * NDMA sixteen-byte
// chunks); D: source rows in flight per wave, one LDS slot each.  Eight columns per lane.
// Y16: with a.y16 set, slot 0 (luma) stores its sums UNCLIPPED as int16 (the RGB scratch planes, see the column
// walk); a parameter rather than a test so that the planar instantiations keep their registers.
// RNG: the converting context's own range step on the 15-bit rows (StripArgs.r15; swscale's lumConvertRange / chrConvertRange),
// between the horizontal finish and the ring: (min(t, cap) * mul + add) >> shift as ONE v_mad_i32_i24 whose product carries
// 1 << (16 - shift), so that the sample lands in the high half and the ring insert picks it from there (every constant pair
// of swscale.c's four functions keeps (min(t, cap) * mul + add) << (16 - shift) inside int32: 2,147,432,768 at most).
template <int LW, int R, int GC, int NDMA, int D, bool Y16 = false, bool RNG = false>
__global__ __launch_bounds__(kMaxLdsBlock, (GC == 4 && R == 4 && LW == 4) ? 8 : 4) void strip_lds_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                               const uint32_t *__restrict__ tab, const StripArgs a)
{
    // GC: columns that share one window (0: every column its own LW-byte window; 4 with LW = 4: 8-byte
    // windows; 4 with LW = 6, 2 with LW = 6 or 8: 12-byte windows)
    [[maybe_unused]] constexpr bool G4 = GC != 0;
    static_assert(R % 2 == 0 && (LW == 4 || LW == 6 || LW == 8) && (NDMA == 1 || NDMA == 2 || NDMA == 4), "see strip_kernel");
    static_assert(GC == 0 || (GC == 4 && (LW == 4 || LW == 6)) || (GC == 2 && (LW == 6 || LW == 8)), "window shapes on offer");
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

    if (blockIdx.x >= a.grid_scaler) { pad_fill_blocks(a, dst, blockIdx.x - a.grid_scaler); return; }   // (whole workgroups: no barrier is skipped; see strip_kernel)
    __shared__ uint32_t next_item;
    for (uint32_t item = blockIdx.x; item < a.scaler_blocks;) {    // (one pass unless the launch is a work queue: a.queue)
    uint32_t lb = xcd_remap(item, a.scaler_blocks);
#ifdef TTS_EXPERIMENTS
    if (a.dbg && threadIdx.x == 0) a.dbg[2 * item] = __builtin_amdgcn_s_memrealtime();
#endif
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
    [[maybe_unused]] const bool y16 = Y16 && a.y16 != 0 && ci == 0;       // wave-uniform
    doff += uint32_t(min(strip * N, c.ow - N)) << (y16 ? 1 : 0);  // the last strip is slid left, not cut short
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

    // Exact 2x shrinks (4K -> 1080p): a lane's four 12-byte windows lie ONE dword apart, so its 16 aligned window dwords are 7
    // distinct ones, and neighbouring lanes read 4 dwords apart -- a 4-way bank conflict on every read, 80 % of the launch on
    // the LDS pipe (profiles/r03/q_planar_lds_counters.txt).  A wave all of whose lanes have their windows on that lattice
    // (every wave but those that hold a strip at a row's end) reads the run of 8 dwords once: half the LDS instructions, and
    // 6 v_alignbyte instead of 12.
    [[maybe_unused]] bool one_run = false;
    if constexpr (GC == 2 && LW == 8) {
        bool mine = true;
#pragma unroll
        for (int j = 1; j < NW; ++j) mine &= lpos[j] == lpos[0] + 4u * uint32_t(j) && ((phase[j] ^ phase[0]) & 3u) == 0;
        one_run = !a.no_union && __builtin_amdgcn_ballot_w64(!mine) == 0;   // wave-uniform
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
    const bool rng = a.range.on != 0;
    const int rmul = a.range.mul[ci], radd = a.range.add[ci], rsh = a.range.shift[ci];
    [[maybe_unused]] int r15cap = 0, r15mul = 0, r15add = 0;
    if constexpr (RNG) {
        const int up = 1 << (16 - a.r15.shift[ci]);
        r15cap = a.r15.cap[ci]; r15mul = a.r15.mul[ci] * up; r15add = a.r15.add[ci] * up;
    }

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
            [[maybe_unused]] bool aligned_already = false;
            if constexpr (G4 && LW >= 6) {
                bool done = false;
                if constexpr (GC == 2 && LW == 8) {
                    if (one_run) {                               // wave-uniform: the four windows as one run of 8 dwords
                        u32x2_t uv[2][2];
                        uint32_t up[2] = {lpos[0] + uint32_t(s % D) * uint32_t(kSlotBytes), lpos[0] + uint32_t(s % D) * uint32_t(kSlotBytes) + 16u};
                        TTS_LDS_READ_WIN4_X2(uv, up);
                        const uint32_t u[8] = {uv[0][0].x, uv[0][0].y, uv[0][1].x, uv[0][1].y, uv[1][0].x, uv[1][0].y, uv[1][1].x, uv[1][1].y};
                        uint32_t sh[6];
#pragma unroll
                        for (int m = 0; m < 6; ++m) sh[m] = __builtin_amdgcn_alignbyte(u[m + 1], u[m], phase[0]);
#pragma unroll
                        for (int j = 0; j < NW; ++j) { wk[j][0] = sh[j]; wk[j][1] = sh[j + 1]; wk[j][2] = sh[j + 2]; wk[j][3] = 0; }
                        done = true; aligned_already = true;
                    }
                }
                if (!done) {
                    u32x2_t wv[NW][2];
                    // (ds_read2's 8-bit dword offsets cannot reach the second slot: its addresses are a few VALU adds)
                    uint32_t lp[NW];
#pragma unroll
                    for (int j = 0; j < NW; ++j) lp[j] = lpos[j] + uint32_t(s % D) * uint32_t(kSlotBytes);
                    if constexpr (NW == 2) TTS_LDS_READ_WIN4_X2(wv, lp); else TTS_LDS_READ_WIN4_X4(wv, lp);
#pragma unroll
                    for (int j = 0; j < NW; ++j) { wk[j][0] = wv[j][0].x; wk[j][1] = wv[j][0].y; wk[j][2] = wv[j][1].x; wk[j][3] = wv[j][1].y; }
                }
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
            if (G4 && !aligned_already) {                       // rebuild the two 8- / 12-byte windows
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
                if constexpr (RNG) {
                    // (every cap is <= 32767: the horizontal pass's own clip rides in the same v_min)
                    const uint32_t t2 = uint32_t(__mul24(min(acc >> (HB ? 15 : 7), r15cap), r15mul) + r15add);
                    ring[i][s / 2] = (s & 1) ? __builtin_amdgcn_perm(t2, ring[i][s / 2], 0x07060100)
                                             : __builtin_amdgcn_perm(ring[i][s / 2], t2, 0x07060302);
                } else if constexpr (HB) {
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
                if constexpr (Y16) {
                    if (y16) {
                        uint32_t w[N / 2];
#pragma unroll
                        for (int q = 0; q < N / 2; ++q) {
                            int v[2];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                int acc = 64 << 12;
#pragma unroll
                                for (int p = 0; p < R / 2; ++p) acc = dot2(ring[2 * q + j][p], vk[p], acc);
                                v[j] = acc;
                            }
                            w[q] = (uint32_t(v[0] >> 19) & 0xffffu) | (uint32_t(v[1] >> 19) << 16);
                        }
                        if (bar) asm volatile("s_barrier" ::: "memory");
                        store_b128(u32x4_t{w[0], w[1], w[2], w[3]}, drs, doff, y * dst_stride);
                        ++y;
                        continue;
                    }
                }
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
                if (rng) { o[0] = range_bytes4(o[0], rmul, radd, rsh); o[1] = range_bytes4(o[1], rmul, radd, rsh); }   // wave-uniform
                if (bar) asm volatile("s_barrier" ::: "memory");
                store_b64(u32x2_t{o[0], o[1]}, drs, doff, y * dst_stride);
                ++y;
            }
            hist[s % D] = cnt;
        }
    }
#ifdef TTS_EXPERIMENTS
    if (a.dbg && threadIdx.x == 0) a.dbg[2 * item + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (!a.queue) break;
    __syncthreads();                                           // (every wave has consumed next_item of the previous round)
    if (threadIdx.x == 0) next_item = a.grid_scaler + 8u * atomicAdd(&a.queue[blockIdx.x & 7u], 1u) + (blockIdx.x & 7u);
    __syncthreads();
    item = next_item;
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
// NDW aligned dwords starting at LDS address p (NDW = 4..17: windows of 6..32 tap pairs at any byte phase):
// ds_read2_b32 pairs (+ one ds_read_b32) and their lgkmcnt wait in ONE asm statement, like the strip readers
// above: as far as the compiler can tell no output register exists before its data has landed, so it cannot
// place a copy or a spill of one between a read and the wait.  (The compiler knows nothing of the reads
// themselves, so it adds no waits of its own.)  The statement is spelt once per pair count by macros.
#define TTS_RDP(k, a, b) "ds_read2_b32 %" #k ", %[p] offset0:" #a " offset1:" #b "\n\t"
#define TTS_RD_2 TTS_RDP(0, 0, 1) TTS_RDP(1, 2, 3)
#define TTS_RD_3 TTS_RD_2 TTS_RDP(2, 4, 5)
#define TTS_RD_4 TTS_RD_3 TTS_RDP(3, 6, 7)
#define TTS_RD_5 TTS_RD_4 TTS_RDP(4, 8, 9)
#define TTS_RD_6 TTS_RD_5 TTS_RDP(5, 10, 11)
#define TTS_RD_7 TTS_RD_6 TTS_RDP(6, 12, 13)
#define TTS_RD_8 TTS_RD_7 TTS_RDP(7, 14, 15)
#define TTS_RO_2 "=&v"(t[0]), "=&v"(t[1])
#define TTS_RO_3 TTS_RO_2, "=&v"(t[2])
#define TTS_RO_4 TTS_RO_3, "=&v"(t[3])
#define TTS_RO_5 TTS_RO_4, "=&v"(t[4])
#define TTS_RO_6 TTS_RO_5, "=&v"(t[5])
#define TTS_RO_7 TTS_RO_6, "=&v"(t[6])
#define TTS_RO_8 TTS_RO_7, "=&v"(t[7])
// NP pairs, then (odd NDW) the last dword at byte offset 8 * NP as operand %NP
#define TTS_RD_EVEN(NP) asm volatile(TTS_RD_##NP "s_waitcnt lgkmcnt(0)" : TTS_RO_##NP : [p] "v"(p) : "memory")
#define TTS_RD_ODD(NP, LASTOP, LASTOFF) \
    asm volatile(TTS_RD_##NP "ds_read_b32 %" #LASTOP ", %[p] offset:" #LASTOFF "\n\ts_waitcnt lgkmcnt(0)" : TTS_RO_##NP, "=&v"(last) : [p] "v"(p) : "memory")
template <int NDW>
__device__ __forceinline__ void lds_read_dwords(uint32_t (&w)[NDW], uint32_t p)
{
    static_assert(NDW >= 4 && NDW <= 17, "one asm statement per dword count on offer");
    u32x2_t t[NDW / 2];
    [[maybe_unused]] uint32_t last = 0;
    if constexpr (NDW == 4) TTS_RD_EVEN(2);
    else if constexpr (NDW == 5) TTS_RD_ODD(2, 2, 16);
    else if constexpr (NDW == 6) TTS_RD_EVEN(3);
    else if constexpr (NDW == 7) TTS_RD_ODD(3, 3, 24);
    else if constexpr (NDW == 8) TTS_RD_EVEN(4);
    else if constexpr (NDW == 9) TTS_RD_ODD(4, 4, 32);
    else if constexpr (NDW == 10) TTS_RD_EVEN(5);
    else if constexpr (NDW == 11) TTS_RD_ODD(5, 5, 40);
    else if constexpr (NDW == 12) TTS_RD_EVEN(6);
    else if constexpr (NDW == 13) TTS_RD_ODD(6, 6, 48);
    else if constexpr (NDW == 14) TTS_RD_EVEN(7);
    else if constexpr (NDW == 15) TTS_RD_ODD(7, 7, 56);
    else if constexpr (NDW == 16) TTS_RD_EVEN(8);
    else TTS_RD_ODD(8, 8, 64);
#pragma unroll
    for (int i = 0; i < NDW / 2; ++i) { w[2 * i] = t[i].x; w[2 * i + 1] = t[i].y; }
    if constexpr (NDW % 2 != 0) w[NDW - 1] = last;
}

// NTP: tap pairs per column (window = 2 * NTP bytes); R: ring depth >= vertical taps; NDMA as above;
// NC: adjacent output columns per lane (1, 2 or 4: the per-row bookkeeping -- DMA, waits, loop control,
// the store -- is the same for 1 byte or 4, so small ratios, where it weighs most, take the widest
// strip their registers allow; the last strip of a row is slid left like strip_lds_kernel's).
//
// D4: the horizontal sum through v_dot4_i32_i8 on the packed bytes themselves.  A 14-bit tap is split into two
// signed-byte digits, t = 128 th + tl (tl in 0..127), and the pixels are biased to signed bytes (one v_xor
// 0x80808080 per window dword; 128 * sum(taps) comes back through the accumulator's start value), so that
// sum(p t) = 128 dot4(p', th) + dot4(p', tl) + bias -- an identity on integers, nothing is rounded.  The window's
// byte phase in LDS is folded into the digit tables (one table per phase, picked once per walk), which removes
// every v_perm: per window dword 1 xor + 2 dot4 for 4 taps where the v_dot2 form spends 2 v_perm + 2 v_dot2.
// With NC = 2 both columns dot the SAME span of NDW + 2 dwords (adjacent windows overlap by three quarters at 6x):
// one LDS read and one xor pass for both; column 1's digits sit `delta` bytes further on.  (Its dword 0 holds
// taps only where swscale folded windows at a row's edge -- but two waves in five hold such a strip, a branch
// around that pair of v_dot4 cost 50 % when taken, and if-converted it cost six instructions: it just runs.)
// RNG: the context's own range step on the 15-bit rows (StripArgs.r15), between hfinish and the ring.
template <int NTP, int R, int NDMA, int NC, int D = 1, bool D4 = false, bool RNG = false>
__global__ __launch_bounds__(kBlock, (D4 && NC == 2 && NTP == 12 && NDMA == 1) ? 6 : 1) void column_lds_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                            const uint32_t *__restrict__ tab, const StripArgs a)
{
    static_assert(NTP % 2 == 0 && R % 2 == 0 && (NDMA == 1 || NDMA == 2), "window is whole dwords; ring holds row pairs");
    static_assert(NC == 1 || NC == 2 || NC == 4, "one byte, short or dword store per lane and row");
    static_assert(D >= 1 && R % D == 0, "D source rows in flight, one LDS slot each, rotating with the ring");
    static_assert(!D4 || NC <= 2, "digit tables exist for one and two columns per lane");
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin exists in the device pass only
    constexpr int NDW = NTP / 2 + 1;                 // aligned dwords that hold a 2 * NTP-byte window at any byte phase
    constexpr int NSP = NC == 2 ? NDW + 2 : NDW;     // D4: dwords of the span that holds every column's window (delta <= 9 bytes: host-checked)
    constexpr int kSlotBytes = NDMA * 1024 + 16;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kBlock / 64][D][kSlotBytes];

    // (pad workgroups come last in dispatch order and are left out of the XCD remap: remapped with the
    // rest they would fill whole XCDs by themselves and leave the scaler the others -- 4.5x slower thumbnails)
    if (blockIdx.x >= a.scaler_blocks) { pad_fill_blocks(a, dst, blockIdx.x - a.scaler_blocks); return; }
    uint32_t lb = xcd_remap(blockIdx.x, a.scaler_blocks);
#ifdef TTS_EXPERIMENTS
    if (a.dbg && threadIdx.x == 0) a.dbg[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
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
    const bool y16 = a.y16 != 0 && ci == 0;        // (wave-uniform)
    doff += uint32_t(col) << (y16 ? 1 : 0);
    if (!live) doff = kDiscard;                    // see strip_lds_kernel

    [[maybe_unused]] uint32_t hc[D4 ? 1 : NC][D4 ? 1 : NTP];
    uint32_t off[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        if constexpr (!D4) {
#pragma unroll
            for (int k = 0; k < NTP; ++k) hc[j][k] = tab[c.hcoef + (col + j) * NTP + k];
        }
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
        const uint32_t pos = off[j] + row0_off + lbase + uint32_t(reinterpret_cast<uintptr_t>((lds_void_t *)&lds[wave][0][0]));
        const uint32_t o = pos & 3u;
        lpos[j] = pos & ~3u;
        // taps 2tp, 2tp+1 (as 16-bit lanes) sit at bytes o + 2 (tp % 2), +1 of dwords tp / 2, tp / 2 + 1
        psel[j][0] = 0x0c000c00u | o | ((o + 1u) << 16);
        psel[j][1] = 0x0c000c00u | (o + 2u) | ((o + 3u) << 16);
    }

    // D4: digit tables of this lane's strip for the byte phase its span has in LDS
    //   NC = 1: [bias][th x NDW][tl x NDW]   NC = 2: [bias0][bias1][th0 x NDW][tl0 x NDW][th1 x NSP][tl1 x NSP]
    [[maybe_unused]] uint32_t th[NC][D4 ? NSP : 1], tl[NC][D4 ? NSP : 1];
    [[maybe_unused]] int bias[NC];
    if constexpr (D4) {
        constexpr int ESZ = NC == 2 ? 2 + 2 * NDW + 2 * NSP : 1 + 2 * NDW;
        const uint32_t o0 = (off[0] + row0_off + lbase + uint32_t(reinterpret_cast<uintptr_t>((lds_void_t *)&lds[wave][0][0]))) & 3u;
        const uint32_t *e = tab + c.hcoef + (uint32_t(strip) * 4u + o0) * uint32_t(ESZ);
#pragma unroll
        for (int j = 0; j < NC; ++j) bias[j] = int(e[j]);
#pragma unroll
        for (int i = 0; i < NDW; ++i) { th[0][i] = e[NC + i]; tl[0][i] = e[NC + NDW + i]; }
        if constexpr (NC == 2) {
#pragma unroll
            for (int i = 0; i < NSP; ++i) { th[1][i] = e[2 + 2 * NDW + i]; tl[1][i] = e[2 + 2 * NDW + NSP + i]; }
        }
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
    const bool rng = a.range.on != 0;
    const int rmul = a.range.mul[ci], radd = a.range.add[ci], rsh = a.range.shift[ci];
    [[maybe_unused]] const int r15cap = a.r15.cap[ci], r15mul = a.r15.mul[ci], r15add = a.r15.add[ci], r15sh = a.r15.shift[ci];

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
    int hist[D];                                   // stores issued in each of the last D iterations
#pragma unroll
    for (int r = 0; r < D; ++r) hist[r] = 0;

    for (int k0 = 0; k0 < nrows; k0 += R) {
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int k = k0 + s;
            if (k >= nrows) continue;                           // wave-uniform (no early exit: the loop must unroll fully)
            const int cnt = cnt_next;
            cnt_next = int(uni(emit[min(k + 1, nrows - 1)]));
            // row k's DMA is followed by the DMAs of D-1 younger rows and the stores of the last D iterations
            int younger = 0;
#pragma unroll
            for (int r = 0; r < D; ++r) younger += hist[r];
            wait_vm_plus<(D - 1) * NDMA>(younger);
            [[maybe_unused]] uint32_t w[D4 ? 1 : NC][D4 ? 1 : NDW];
            [[maybe_unused]] uint32_t ws[D4 ? NSP : 1];
            if constexpr (D4) lds_read_dwords<NSP>(ws, lpos[0] + uint32_t(s % D) * uint32_t(kSlotBytes));
            else {
#pragma unroll
                for (int j = 0; j < NC; ++j) lds_read_dwords<NDW>(w[j], lpos[j] + uint32_t(s % D) * uint32_t(kSlotBytes));
            }
            // the slot is free again (its bytes sit in registers): request row k + D into it
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_void_t *)&lds[wave][s % D][d * 1024], 16, csrc[d],
                                                         min(k + D, nrows - 1) * src_stride, 0, 0);
            if constexpr (D4) {
#pragma unroll
                for (int i = 0; i < NSP; ++i) ws[i] ^= 0x80808080u;
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                int acc = 0;
                if constexpr (D4) {
                    // (the first v_dot4 of a chain in its VOP3P form -- clamp = true, as in dot2(): the sums stay far
                    // from the int32 range -- whose accumulator input is a free operand: no v_mov of 0 / bias first)
                    int hi = __builtin_amdgcn_sdot4(int(ws[0]), int(th[j][0]), 0, true);
                    int lo = __builtin_amdgcn_sdot4(int(ws[0]), int(tl[j][0]), bias[j], true);
#pragma unroll
                    for (int i = 1; i < (j ? NSP : NDW); ++i) {
                        hi = __builtin_amdgcn_sdot4(int(ws[i]), int(th[j][i]), hi, false);
                        lo = __builtin_amdgcn_sdot4(int(ws[i]), int(tl[j][i]), lo, false);
                    }
                    acc = (hi << 7) + lo;
                } else {
#pragma unroll
                    for (int tp = 0; tp < NTP; ++tp)
                        acc = dot2(__builtin_amdgcn_perm(w[j][tp / 2 + 1], w[j][tp / 2], psel[j][tp % 2]), hc[j][tp], acc);
                }
                uint32_t t = hfinish(acc);
                if constexpr (RNG) t = uint32_t((min(int(t), r15cap) * r15mul + r15add) >> r15sh);
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
                if (y16) {                                      // wave-uniform: unclipped 16-bit sums for the RGB conversion pass
                    if constexpr (NC == 2) store_b32((uint32_t(v[0] >> 19) & 0xffffu) | (uint32_t(v[1] >> 19) << 16), drs, doff, yoff);
                    else if constexpr (NC == 1) store_b16(uint32_t(v[0] >> 19), drs, doff, yoff);
                    ++y;
                    continue;
                }
                if constexpr (NC == 4) {
                    uint32_t o4 = pack4(v[0], v[1], v[2], v[3]);
                    if (rng) o4 = range_bytes4(o4, rmul, radd, rsh);
                    store_b32(o4, drs, doff, yoff);
                } else {
                    uint32_t o2 = pack2(v[0], v[NC - 1]);       // bits 16..31 undefined: the stores below take 8 / 16 bits
                    if (rng) o2 = pack2_by(__mul24(int(o2 & 255u), rmul) + radd, __mul24(int((o2 >> 8) & 255u), rmul) + radd, rsh);   // wave-uniform
                    if constexpr (NC == 2) store_b16(o2, drs, doff, yoff);
                    else store_b8(o2, drs, doff, yoff);
                }
                ++y;
            }
            hist[s % D] = cnt;
        }
    }
#ifdef TTS_EXPERIMENTS
    if (a.dbg && threadIdx.x == 0) a.dbg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
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
    int32_t range15;                   // TTS_RANGE_*: the context's own range step on the 15-bit rows between the passes (0: none)
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
        // swscale.c lumRangeToJpeg_c / chrRangeToJpeg_c / lumRangeFromJpeg_c / chrRangeFromJpeg_c (c->lumConvertRange, chrConvertRange)
        if (a.range15 == TTS_RANGE_TO_FULL) acc = plane ? (min(acc, 30775) * 4663 - 9289992) >> 12 : (min(acc, 30189) * 19077 - 39057361) >> 14;
        else if (a.range15 == TTS_RANGE_TO_LIMITED) acc = plane ? (acc * 1799 + 4081085) >> 11 : (acc * 14071 + 33561947) >> 14;
        val += acc * int(vc[j]);
    }
    val >>= 19;                                            // (a one-tap vertical filter gives (t + 64) >> 7: the same value)
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

// Range step (tts_geom.out_range) as a pass of its own over the finished canvas, in place: behind the general kernel and
// behind the kernels fed with packed RGB (the walks of this file carry the step in their store epilogue instead).  Sixteen
// bytes a lane through the closed form of filter.cpp range_affine() where the row allows dword access, byte by byte at
// ragged ends and on rows off the dword grid; the 256-entry table only if the step has no closed form (it always has).
struct RangeArgs {
    uint8_t *dst;
    int32_t w, h, stride;              // canvas of this plane class
    int32_t planes, plane0;
    int32_t frame0;
    uint32_t origin;                   // bytes from the canvas origin to the picture origin
    int32_t affine, mul, add, shift;   // affine: (p * mul + add) >> shift, clipped, equals the table
    FrameAddr addr;
    uint8_t lut[256];
};

__global__ __launch_bounds__(kBlock) void range_kernel(const RangeArgs a)
{
    // (h == 1: a plane whose rows lie back to back, taken as one row by the host -- all four waves of a workgroup side by side)
    const int x  = a.h == 1 ? (int(blockIdx.x) * kBlock + int(threadIdx.x)) * 16 : (int(blockIdx.x) * 64 + int(threadIdx.x & 63)) * 16;
    const int y  = a.h == 1 ? 0 : int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    const int fp = int(blockIdx.z);
    if (x >= a.w || y >= a.h) return;
    const int plane = a.plane0 + fp % a.planes, frame = a.frame0 + fp / a.planes;
    const size_t doff = (a.addr.use_table ? size_t(a.addr.dst_tab[plane][frame])
                                          : size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane]) - a.origin;
    uint8_t *row = a.dst + doff + size_t(y) * size_t(a.stride);
    if (a.affine && x + 16 <= a.w && (reinterpret_cast<uintptr_t>(row) & 3u) == 0) {
        u32x4_t v = *reinterpret_cast<const u32x4_t *>(row + x);
        v.x = range_bytes4(v.x, a.mul, a.add, a.shift); v.y = range_bytes4(v.y, a.mul, a.add, a.shift);
        v.z = range_bytes4(v.z, a.mul, a.add, a.shift); v.w = range_bytes4(v.w, a.mul, a.add, a.shift);
        *reinterpret_cast<u32x4_t *>(row + x) = v;
        return;
    }
    for (int i = x; i < min(x + 16, a.w); ++i) {
        if (a.affine) { const int v = (int(row[i]) * a.mul + a.add) >> a.shift; row[i] = uint8_t(v < 0 ? 0 : v > 255 ? 255 : v); }
        else row[i] = a.lut[row[i]];
    }
}

}  // namespace
