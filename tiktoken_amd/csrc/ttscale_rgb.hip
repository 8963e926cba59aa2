// ttscale_rgb.hip -- packed-RGB output of libttscale: the yuv420p -> rgb24 / bgr24 /
// rgba / bgra conversion swscale performs INSIDE its scaling pass, fused here into the scaler's vertical
// pass so that no yuv intermediate ever reaches HBM and every output row is written once, as RGB.
//
// The reference's chains end in yuv420p / mjpeg (src/services/video.js:329-334); BASELINE.json's north_star
// lists the colour conversion as part of the path, so `format=rgb24` (bgr24, rgba, bgra) ending a chain
// selects what libswscale's C path computes for such a context (default flags: no full_chroma_int, no
// accurate_rnd):
//   * chroma is scaled to ceil(dst_w / 2) columns and dst_h rows (two neighbouring pixels share a chroma
//     sample; vertically chroma is interpolated to every output row), luma to dst_w x dst_h, both through
//     hScale8To15 (15-bit intermediate, truncating >> 7);
//   * output.c yuv2rgb_X_c_template(): Y1, Y2, U, V = (1 << 18 + sum of taps x intermediates) >> 19, NOT
//     clipped, index libswscale/yuv2rgb.c's tables: r = table_rV[V], g = table_gU[U] + table_gV[V],
//     b = table_bU[U], pixel = r[Y], g[Y], b[Y];
//   * the tables (ff_yuv2rgb_c_init_tables, ITU-R BT.601, limited range) are one clipped ramp
//     y_table[i] = clip_u8((i * cy - (384 << 16) - (16 << 16) + 0x8000) >> 16), entered at 326 + Y + the
//     chroma's offset ((clip_u8(C) * inc) >> 16) - (inc >> 9) -- so a pixel is a closed form,
//         clip_u8(((Y + off) * cy + K) >> 16),   K = 326 cy - (400 << 16) + 0x8000,
//     which is what the kernels evaluate (the oracle keeps swscale's tables; the two must agree);
//   * same size in and out with an even height: the unscaled converter (yuv2rgb_c_24_rgb & co.), which
//     indexes the same tables with the 8-bit samples, chroma row y >> 1 for both luma rows.
// Constants recalled from FFmpeg 4.x-6.x, UNVERIFIED like the rest.
//
// Kernels: rgb_strip_kernel (every axis <= 4 taps: the portrait upscales), rgb_generic_kernel (any taps),
// rgb_unscaled_kernel.  All three cover the whole canvas: pad borders are written by the same launch.
#include "ttscale_rgb.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <tuple>
#include <type_traits>
#include <vector>

#include "filter.h"

namespace tts {

// (RgbCoef -- the tables of ff_yuv2rgb_c_init_tables() as the closed form above -- is declared in ttscale_rgb.h)
// full: the source is yuvj420p (ff_yuv2rgb_c_init_tables(fullRange = 1)): no luma stretch (cy = 1 << 16, no 16 << 16
// offset), chroma coefficients x 224 / 255, table origin 384 instead of 326 -- the ramp is then y_table[384 + i] = i
static RgbCoef rgb_coef(bool full = false)
{
    RgbCoef c;
    const int64_t cy = full ? (int64_t(1) << 16) : (int64_t(1) << 16) * 255 / 219;
    auto scaled = [&](int64_t v) { if (full) v = v * 224 / 255; return int32_t((v * (1 << 16) + 0x8000) / cy); };   // C division: truncates toward zero
    c.cy = int32_t(cy);
    c.crv = scaled(104597); c.cbu = scaled(132201); c.cgu = scaled(-25675); c.cgv = scaled(-53279);
    c.k = full ? 0x8000 : int32_t(326 * cy - (int64_t(400) << 16) + 0x8000);
    c.off_r = -(c.crv >> 9); c.off_b = -(c.cbu >> 9); c.off_g = -(c.cgu >> 9) - (c.cgv >> 9);
    return c;
}

// one pixel, host and device: Y is the (unclipped) vertical-pass value, U and V likewise
__host__ __device__ inline void rgb_pixel(const RgbCoef &c, int Y, int U, int V, int &r, int &g, int &b)
{
    U = U < 0 ? 0 : U > 255 ? 255 : U;          // fill_table(): av_clip_uint8(i - YUVRGB_TABLE_HEADROOM)
    V = V < 0 ? 0 : V > 255 ? 255 : V;
    const int tr = Y + ((V * c.crv) >> 16) + c.off_r;
    const int tg = Y + ((U * c.cgu) >> 16) + ((V * c.cgv) >> 16) + c.off_g;      // (>> of a negative product: floor, as in C on int64)
    const int tb = Y + ((U * c.cbu) >> 16) + c.off_b;
    r = (tr * c.cy + c.k) >> 16; g = (tg * c.cy + c.k) >> 16; b = (tb * c.cy + c.k) >> 16;
    r = r < 0 ? 0 : r > 255 ? 255 : r; g = g < 0 ? 0 : g > 255 ? 255 : g; b = b < 0 ? 0 : b > 255 ? 255 : b;
}

}  // namespace tts

namespace {

using tts::RgbCoef;

// What every RGB kernel needs to know of the chain: the canvas, where the picture sits in it, which part of
// the context's output (the virtual scaled image) the picture is, the border colour, the frame addresses.
struct RgbFrameArgs {
    const uint8_t *src;
    uint8_t       *dst;
    int32_t canvas_w, canvas_h;        // output frame
    int32_t px, py, pw, ph;            // picture rectangle inside the canvas
    int32_t ox, oy;                    // its origin in the context's output (crop offset behind the scale)
    int32_t src_stride_y, src_stride_c, dst_stride;
    int32_t bpp, swap;                 // 3 / 4 bytes a pixel; swap: B first
    int32_t border[3];                 // border pixel, r g b
    // The pad ran on yuv420p, ahead of the conversion (TTS_RGB_AT_END): luma and chroma planes were padded
    // separately, on their own grids -- an odd picture height leaves a luma border row over a picture chroma
    // row -- so a canvas pixel takes Y from the luma rectangle and U, V from the chroma rectangle.
    int32_t yuv_pad;
    int32_t cpx, cpy, cpw, cph;        // yuv_pad: picture rectangle of the chroma planes
    int32_t fill[3];                   // yuv_pad: pad colour y, u, v
    int32_t frame0;                    // first frame of this launch (grid.z holds at most 65535 frames)
    // `format=yuvj420p,format=rgb24`: the conversion pass reads the canvas the range pass left, sample by sample
    // clip8((p * mul + add) >> shift) (filter.h range_affine), luma / chroma; coef then holds the full-range tables
    int32_t rng_on, rng[2][3];
    RgbCoef coef;
    FrameAddr addr;
};

__device__ __forceinline__ int rgb_range(const RgbFrameArgs &a, int chroma, int p)
{
    if (!a.rng_on) return p;
    const int v = (p * a.rng[chroma][0] + a.rng[chroma][1]) >> a.rng[chroma][2];
    return v < 0 ? 0 : v > 255 ? 255 : v;
}

// base pointers of the frame a workgroup works on (grid.z = frame): 64-bit, per workgroup
__device__ __forceinline__ void rgb_frame_bases(const RgbFrameArgs &a, int frame, const uint8_t *(&sp)[3], uint8_t *&dp)
{
    if (a.addr.use_table) {
        for (int k = 0; k < 3; ++k) sp[k] = a.src + a.addr.src_tab[k][frame];
        dp = a.dst + a.addr.dst_tab[0][frame];
    } else {
        for (int k = 0; k < 3; ++k) sp[k] = a.src + size_t(frame) * a.addr.src_pitch + a.addr.src_plane[k];
        dp = a.dst + size_t(frame) * a.addr.dst_pitch;
    }
}

__device__ __forceinline__ void rgb_store_pixel(uint8_t *d, int bpp, int swap, int r, int g, int b)
{
    d[0] = uint8_t(swap ? b : r); d[1] = uint8_t(g); d[2] = uint8_t(swap ? r : b);
    if (bpp == 4) d[3] = 255;
}

// ---- same size in and out, even height: swscale's unscaled converter --------------------------------
// Four adjacent pixels a lane (two chroma samples), 12 or 16 bytes stored as dwords where the row allows.
__global__ __launch_bounds__(kBlock) void rgb_unscaled_kernel(const RgbFrameArgs a)
{
    const int x0 = (int(blockIdx.x) * 64 + int(threadIdx.x & 63)) * 4;
    const int y  = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    if (x0 >= a.canvas_w || y >= a.canvas_h) return;
    const uint8_t *sp[3];
    uint8_t *dp;
    rgb_frame_bases(a, a.frame0 + int(blockIdx.z), sp, dp);
    uint8_t px[16];
    const int n = min(4, a.canvas_w - x0);
    const bool in_rows = y >= a.py && y < a.py + a.ph;
    for (int i = 0; i < n; ++i) {
        const int x = x0 + i;
        int r = a.border[0], g = a.border[1], b = a.border[2];
        const bool in_luma = in_rows && x >= a.px && x < a.px + a.pw;
        if (a.yuv_pad) {
            int Y = a.fill[0], U = a.fill[1], V = a.fill[2];
            if (in_luma) Y = sp[0][size_t(a.oy + y - a.py) * a.src_stride_y + (a.ox + x - a.px)];
            const int cx = x >> 1, cy = y >> 1;
            if (cy >= a.cpy && cy < a.cpy + a.cph && cx >= a.cpx && cx < a.cpx + a.cpw) {
                const size_t o = size_t((a.oy >> 1) + cy - a.cpy) * a.src_stride_c + ((a.ox >> 1) + cx - a.cpx);
                U = sp[1][o]; V = sp[2][o];
            }
            tts::rgb_pixel(a.coef, rgb_range(a, 0, Y), rgb_range(a, 1, U), rgb_range(a, 1, V), r, g, b);
        } else if (in_luma) {
            const int sx = a.ox + x - a.px, sy = a.oy + y - a.py;
            const int Y = sp[0][size_t(sy) * a.src_stride_y + sx];
            const int U = sp[1][size_t(sy >> 1) * a.src_stride_c + (sx >> 1)];
            const int V = sp[2][size_t(sy >> 1) * a.src_stride_c + (sx >> 1)];
            tts::rgb_pixel(a.coef, rgb_range(a, 0, Y), rgb_range(a, 1, U), rgb_range(a, 1, V), r, g, b);
        }
        rgb_store_pixel(px + i * a.bpp, a.bpp, a.swap, r, g, b);
    }
    uint8_t *d = dp + size_t(y) * a.dst_stride + size_t(x0) * a.bpp;
    const int nb = n * a.bpp;
    if (n == 4 && (reinterpret_cast<uintptr_t>(d) & 3) == 0) {
        uint32_t w[4];
        for (int i = 0; i < 4; ++i) w[i] = uint32_t(px[4 * i]) | uint32_t(px[4 * i + 1]) << 8 | uint32_t(px[4 * i + 2]) << 16 | uint32_t(px[4 * i + 3]) << 24;
        for (int i = 0; i < nb / 4; ++i) reinterpret_cast<uint32_t *>(d)[i] = w[i];
    } else {
        for (int i = 0; i < nb; ++i) d[i] = px[i];
    }
}

// ---- any tap counts: one lane per output pixel --------------------------------------------------------
struct RgbGenericArgs {
    RgbFrameArgs f;
    const int32_t *lhpos, *lvpos, *chpos, *cvpos;     // luma / chroma, horizontal / vertical: first tap
    const int16_t *lhtaps, *lvtaps, *chtaps, *cvtaps;
    int32_t lht, lvt, cht, cvt;                        // taps per sample
    // two_stage: the chain's scale wrote yuv420p (a `format=yuv420p` pins its output) and a conversion pass
    // follows at the end.  Same kernel, other arithmetic: every plane is scaled as yuv420p scales it (chroma
    // to ceil(w / 2) x ceil(h / 2)), clipped to 8 bits, padded with the yuv fill on the chroma grid, and the
    // finished canvas goes through the unscaled converter (chroma sample x >> 1, y >> 1).
    int32_t two_stage;
    int32_t cox, coy;                                  // two_stage: origin of the chroma picture rectangle (f.cp*) in the scaled chroma plane
};

// hScale8To15_c for one output column of one source row (rng: the sample is read through the range step first)
__device__ __forceinline__ int rgb_hrow(const RgbFrameArgs &f, int chroma, const uint8_t *row, const int16_t *hc, int ht)
{
    int acc = 0;
    for (int k = 0; k < ht; ++k) acc += rgb_range(f, chroma, int(row[k])) * int(hc[k]);
    acc >>= 7;
    return acc < 32767 ? acc : 32767;
}

__device__ __forceinline__ int rgb_hv(const RgbFrameArgs &f, int chroma, const uint8_t *plane, int stride, const int32_t *hpos, const int16_t *htaps, int ht,
                                      const int32_t *vpos, const int16_t *vtaps, int vt, int vx, int vy)
{
    const int hp = hpos[vx], vp = vpos[vy];
    const int16_t *hc = htaps + size_t(vx) * ht, *vc = vtaps + size_t(vy) * vt;
    int val = 1 << 18;
    for (int j = 0; j < vt; ++j) val += rgb_hrow(f, chroma, plane + size_t(vp + j) * size_t(stride) + hp, hc, ht) * int(vc[j]);
    return val >> 19;
}

// vscale.c packed_vscale(): two taps qualify for the 1- / 2-line writers when they sum to 4096 and the second is not negative
__device__ __forceinline__ bool rgb_two_tap_ok(const int16_t *t) { return int(t[0]) + int(t[1]) == 4096 && uint16_t(t[1]) <= 4096u; }

__global__ __launch_bounds__(kBlock) void rgb_generic_kernel(const RgbGenericArgs a)
{
    const RgbFrameArgs &f = a.f;
    const int x = int(blockIdx.x) * 64 + int(threadIdx.x & 63);
    const int y = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    if (x >= f.canvas_w || y >= f.canvas_h) return;
    const uint8_t *sp[3];
    uint8_t *dp;
    rgb_frame_bases(f, f.frame0 + int(blockIdx.z), sp, dp);
    int r = f.border[0], g = f.border[1], b = f.border[2];
    if (a.two_stage) {
        // (the scaling pass is the planar one: no range step inside it; the range pass, if any, runs on its clipped bytes)
        RgbFrameArgs plain = f;
        plain.rng_on = 0;
        auto clip8 = [](int v) { return v < 0 ? 0 : v > 255 ? 255 : v; };
        int Y = f.fill[0], U = f.fill[1], V = f.fill[2];
        if (y >= f.py && y < f.py + f.ph && x >= f.px && x < f.px + f.pw)
            Y = clip8(rgb_hv(plain, 0, sp[0], f.src_stride_y, a.lhpos, a.lhtaps, a.lht, a.lvpos, a.lvtaps, a.lvt, f.ox + x - f.px, f.oy + y - f.py));
        const int cx = x >> 1, cy = y >> 1;
        if (cy >= f.cpy && cy < f.cpy + f.cph && cx >= f.cpx && cx < f.cpx + f.cpw) {
            U = clip8(rgb_hv(plain, 1, sp[1], f.src_stride_c, a.chpos, a.chtaps, a.cht, a.cvpos, a.cvtaps, a.cvt, a.cox + cx - f.cpx, a.coy + cy - f.cpy));
            V = clip8(rgb_hv(plain, 1, sp[2], f.src_stride_c, a.chpos, a.chtaps, a.cht, a.cvpos, a.cvtaps, a.cvt, a.cox + cx - f.cpx, a.coy + cy - f.cpy));
        }
        tts::rgb_pixel(f.coef, rgb_range(f, 0, Y), rgb_range(f, 1, U), rgb_range(f, 1, V), r, g, b);
    } else if (y >= f.py && y < f.py + f.ph && x >= f.px && x < f.px + f.pw) {
        const int vx = f.ox + x - f.px, vy = f.oy + y - f.py;
        // which of swscale's packed writers this output row takes (vscale.c packed_vscale(), decided per row from its taps)
        const int16_t *lvc = a.lvtaps + size_t(vy) * a.lvt, *cvc = a.cvtaps + size_t(vy) * a.cvt;
        int writer = 0, yalpha = 0, uvalpha = 0;
        if (a.lvt == 1 && a.cvt == 1) writer = 1;
        else if (a.lvt == 1 && a.cvt == 2 && rgb_two_tap_ok(cvc)) { writer = 1; uvalpha = cvc[1]; }
        else if (a.lvt == 2 && a.cvt == 2 && rgb_two_tap_ok(lvc) && rgb_two_tap_ok(cvc)) { writer = 2; yalpha = lvc[1]; uvalpha = cvc[1]; }
        int Y, U, V;
        if (writer == 0) {                                     // yuv2rgb_X_c_template
            Y = rgb_hv(f, 0, sp[0], f.src_stride_y, a.lhpos, a.lhtaps, a.lht, a.lvpos, a.lvtaps, a.lvt, vx, vy);
            U = rgb_hv(f, 1, sp[1], f.src_stride_c, a.chpos, a.chtaps, a.cht, a.cvpos, a.cvtaps, a.cvt, vx >> 1, vy);
            V = rgb_hv(f, 1, sp[2], f.src_stride_c, a.chpos, a.chtaps, a.cht, a.cvpos, a.cvtaps, a.cvt, vx >> 1, vy);
        } else {
            const int cxv = vx >> 1, lp = a.lvpos[vy], cp = a.cvpos[vy];
            const int16_t *lh = a.lhtaps + size_t(vx) * a.lht, *chh = a.chtaps + size_t(cxv) * a.cht;
            const uint8_t *ly = sp[0] + size_t(lp) * size_t(f.src_stride_y) + a.lhpos[vx];
            const size_t co = size_t(cp) * size_t(f.src_stride_c) + a.chpos[cxv];
            const int y0 = rgb_hrow(f, 0, ly, lh, a.lht);
            const int u0 = rgb_hrow(f, 1, sp[1] + co, chh, a.cht), v0 = rgb_hrow(f, 1, sp[2] + co, chh, a.cht);
            const bool two_c = writer == 2 || uvalpha >= 2048;  // (the second chroma line is read only where the writer uses it)
            const int u1 = two_c ? rgb_hrow(f, 1, sp[1] + co + f.src_stride_c, chh, a.cht) : 0, v1 = two_c ? rgb_hrow(f, 1, sp[2] + co + f.src_stride_c, chh, a.cht) : 0;
            if (writer == 1) {                                 // yuv2rgb_1_c_template: chroma from the first line (uvalpha < 2048) or the average of two
                Y = (y0 + 64) >> 7;
                U = two_c ? (u0 + u1 + 128) >> 8 : (u0 + 64) >> 7;
                V = two_c ? (v0 + v1 + 128) >> 8 : (v0 + 64) >> 7;
            } else {                                           // yuv2rgb_2_c_template: no rounding term
                const int y1 = rgb_hrow(f, 0, ly + f.src_stride_y, lh, a.lht);
                Y = (y0 * (4096 - yalpha) + y1 * yalpha) >> 19;
                U = (u0 * (4096 - uvalpha) + u1 * uvalpha) >> 19;
                V = (v0 * (4096 - uvalpha) + v1 * uvalpha) >> 19;
            }
        }
        tts::rgb_pixel(f.coef, Y, U, V, r, g, b);
    }
    rgb_store_pixel(dp + size_t(y) * f.dst_stride + size_t(x) * f.bpp, f.bpp, f.swap, r, g, b);
}


// ---- every axis <= 4 taps (the portrait upscales): a lane owns 8 adjacent output pixels --------------
// Same walk as the planar strip kernels (ttscale_kernels.h): a lane takes 8 adjacent output pixels of one
// frame -- 8 luma columns and the 4 chroma columns under them, for U and for V -- and goes down a segment of
// output rows.  Source rows arrive as grouped 8-byte windows (one per 4 columns: 2 luma, 1 U, 1 V per lane)
// through plain buffer loads issued one source row ahead; the 15-bit horizontal results sit in three
// register rings of 4 rows (shift registers: the newest row enters at the top, so the vertical taps keep
// their natural order).  Luma and chroma advance at different rates -- chroma has half the source rows but
// is interpolated to EVERY output row -- so the host tabulates, per output row, how many new luma and chroma
// source rows it needs first.  Then 16 two-instruction vertical sums, the yuv -> RGB closed form (per
// chroma sample: three products shared by its two pixels; per pixel and channel one add) and 24 (32) bytes
// stored per lane: a wave writes 1.5 (2) KiB of one RGB row, contiguous.  No yuv intermediate exists.
struct RgbStripArgs {
    uint32_t hoffl, sell, coefl;       // luma tables (dword offsets into tab): [strips][2], [strips][8][2], [strips][8][2]
    uint32_t hoffc, selc, coefc;       // chroma: [strips], [strips][4][2], [strips][4][2]
    uint32_t vrow;                     // [ph][6]: new luma rows, new chroma rows, luma taps (2 dwords), chroma taps (2 dwords)
    uint32_t seginfo;                  // [nseg][2]: first luma / chroma source row of the segment's ring fill (may be < 0: clamped, zero taps)
    int32_t strips, n_items, nseg, seg_rows, ph, pw;
    int32_t src_h_l, src_h_c;          // rows of the (pre-cropped) source planes
    uint32_t end_rel[2];               // luma / chroma: bytes from a plane's (pre-cropped) origin to one past its last byte
    int32_t src_stride_y, src_stride_c, dst_stride;
    int32_t swap;
    uint64_t src_span, dst_span;
    RgbCoef coef;
    FrameAddr addr;
};

// two 32-bit sums -> their high halves (sum >> 16) clipped to bytes, in bits 0..15
[[maybe_unused]] __device__ __forceinline__ uint32_t hi2_u8(int a, int b)
{
    return sat_pk_u8(__builtin_amdgcn_perm(uint32_t(b), uint32_t(a), 0x07060302));
}
[[maybe_unused]] __device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }   // (v_med3_i32)

// ---- LDS staging of the source rows (LDS = true) --------------------------------------------------------
// With plain per-lane window loads the kernel ran at 0.89 ms per 300 frames of 1080p -> portrait -> rgb24
// although its instruction stream alone takes 0.50 ms, its loads alone 0.54 and its stores alone 0.55
// (zero-record descriptors, experiments build): the window gathers and the stores fight over the vector L1.
// So, as in the planar strip_lds_kernel, a wave fetches the run of source bytes that holds all of its lanes'
// windows with ONE coalesced LDS-DMA per plane and source row (`buffer_load_dwordx4 ... lds`, 16 bytes a
// lane, into the wave's own 1 KiB slot; one slot per plane), and the lanes pick their windows out of LDS as
// aligned dwords.  Nothing in flight lives in VGPRs; the waits are ours and exact: before a row is read, wait
// until only the vector-memory operations issued AFTER its DMA are outstanding (counted at run time; the
// counter retires in issue order; a smaller count is always safe).
typedef __attribute__((address_space(3))) void rgb_lds_t;
[[maybe_unused]] constexpr int kRgbSlot = 1024 + 16;
constexpr int kRgbMaxPieces = 8;

[[maybe_unused]] __device__ __forceinline__ void rgb_wait_vm(int outstanding)
{
    switch (outstanding < 7 ? outstanding : 7) {
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

// two 8-byte windows at any byte phase: the three aligned dwords that hold each, and the wait, in ONE asm
// statement (no output register exists, as far as the compiler can tell, before its data has landed)
[[maybe_unused]] __device__ __forceinline__ void rgb_lds_read2(uint32_t (&w)[2][3], uint32_t p0, uint32_t p1)
{
    asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:4\n\tds_read_b32 %2, %6 offset:8\n\t"
                 "ds_read_b32 %3, %7\n\tds_read_b32 %4, %7 offset:4\n\tds_read_b32 %5, %7 offset:8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0][0]), "=&v"(w[0][1]), "=&v"(w[0][2]), "=&v"(w[1][0]), "=&v"(w[1][1]), "=&v"(w[1][2])
                 : "v"(p0), "v"(p1) : "memory");
}

// The source run of one plane for this wave: one piece per frame row its 64 lanes touch (a new piece starts
// at every lane whose strip is the first of a row, `new_row`), each rounded out to 16-byte chunks, packed
// one after another into the wave's 64-chunk slot; chunk c is fetched by lane c.  In: this lane's bytes
// [first, last) of the plane's row 0 (the DMA adds row * stride as its scalar offset), the plane's origin
// `soff`, `end_rel` = bytes from that origin to one past the plane's last byte, `last_rowoff` = offset of the
// plane's last row.  A run that would pass the plane's end on that last row is slid down to end exactly on
// the piece's last byte (at the end of a launch's span the range check zero-fills whole dwords).  Packed
// batches only: the bytes before a plane are the previous plane's.  Out: what this lane fetches; slot
// position minus source offset for this lane's piece.
[[maybe_unused]] __device__ __forceinline__ void rgb_plan_run(uint32_t first, uint32_t last, uint32_t soff, uint32_t end_rel, uint32_t last_rowoff,
                                             bool new_row, int lane, uint32_t &csrc, uint32_t &lbase)
{
    uint64_t pm = __builtin_amdgcn_ballot_w64(new_row) | 1ull;
    uint32_t prefix = 0;
    lbase = 0;
    csrc = ~0u;
    for (int p = 0; p < kRgbMaxPieces && pm; ++p) {
        const int l0 = __builtin_ctzll(pm);
        pm &= pm - 1;
        const int l1 = pm ? __builtin_ctzll(pm) : 64;
        const uint32_t hi_abs = uint32_t(__builtin_amdgcn_readlane(int(soff), l0)) + end_rel;
        const uint32_t end = uint32_t(__builtin_amdgcn_readlane(int(last), l1 - 1));
        uint32_t start = uint32_t(__builtin_amdgcn_readlane(int(first), l0)) & ~15u;
        const uint32_t n = (end - start + 15u) >> 4;
        if (start + 16u * n + last_rowoff > hi_abs && end >= 16u * n) start = end - 16u * n;
        if (lane >= l0 && lane < l1) lbase = 16u * prefix - start;
        if (p == 0) csrc = start;                   // spare lanes re-fetch the first piece's first chunk
        if (uint32_t(lane) >= prefix && uint32_t(lane) < prefix + n) csrc = start + 16u * (uint32_t(lane) - prefix);
        prefix += n;
    }
    // the host picks the LDS variant only for geometries whose runs fit; anything else here would be silent corruption
    if (pm || prefix > 64u) __builtin_trap();
}

// ---- whole-sector stores (XP = true) --------------------------------------------------------------------
// A lane's 8 pixels are 24 (32) bytes; stored lane by lane (16 + 8 bytes at a stride of 24) every store
// instruction touches every 64-byte sector of the wave's row piece and leaves it partly written: PMC showed
// 24.9 L2 write requests per store instruction, twice what the bytes need, and the kernel ran at the speed of
// those requests.  So the wave's output row goes through a small LDS exchange first: every lane writes its
// bytes at lane * 24 (32), then lane j reads 16-byte chunk j (and chunk 64 + j) of that 1.5 (2) KiB string
// and stores it: 1 KiB of whole sectors per instruction.  Chunk c of rgb24 comes from lane 16c / 24 and, when
// it starts at byte 16 of that lane, ends in the next lane; where the two lanes' pixels are not neighbours in
// memory (a wave's 64 strips wrap over frames once; the host sends other shapes to XP = false) that ONE chunk
// is held back and stored as two 8-byte halves.
[[maybe_unused]] __device__ __forceinline__ void rgb_lds_exchange3(uint32_t waddr, const uint32_t (&o)[6], uint32_t r1, uint32_t r2, u32x4_t &v1, u32x4_t &v2)
{
    asm volatile("ds_write_b64 %2, %3\n\tds_write_b64 %2, %4 offset:8\n\tds_write_b64 %2, %5 offset:16\n\t"
                 "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v1), "=&v"(v2)
                 : "v"(waddr), "v"(u32x2_t{o[0], o[1]}), "v"(u32x2_t{o[2], o[3]}), "v"(u32x2_t{o[4], o[5]}), "v"(r1), "v"(r2) : "memory");
}
[[maybe_unused]] __device__ __forceinline__ void rgb_lds_exchange4(uint32_t waddr, const uint32_t (&o)[8], uint32_t r1, uint32_t r2, u32x4_t &v1, u32x4_t &v2)
{
    asm volatile("ds_write_b128 %2, %3\n\tds_write_b128 %2, %4 offset:16\n\t"
                 "ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v1), "=&v"(v2)
                 : "v"(waddr), "v"(u32x4_t{o[0], o[1], o[2], o[3]}), "v"(u32x4_t{o[4], o[5], o[6], o[7]}), "v"(r1), "v"(r2) : "memory");
}

// ---- same size in and out, the fast form: eight pixels of a row PAIR a lane --------------------------------
// The conversion alone (`format=rgb24` with no scale in the chain, or behind a crop: yuv2rgb_c_24_rgb) is a
// streaming job: 1.5 bytes in, 3 (4) out per pixel.  A lane takes 8 adjacent pixels of two rows that share a chroma
// row -- 8 + 8 bytes of luma, 4 + 4 of chroma, as dwords -- and produces 2 x 24 (32) bytes; the chroma terms
// (three products per sample) serve four pixels, a pixel costs one multiply and three adds, and four channel
// values become four clipped bytes in two instructions (pack4<16>).  A wave's row goes through the LDS exchange of
// the strip kernel above, so that every store instruction writes whole sectors.  The host sends here only what
// fits: no border, a picture that is a multiple of 8 wide, dword-aligned source and output rows; everything else
// stays with rgb_unscaled_kernel.
template <int BPP>
__global__ __launch_bounds__(kBlock) void rgb_unscaled8_kernel(const RgbFrameArgs a)
{
    static_assert(BPP == 3 || BPP == 4, "rgb24 / bgr24 or rgba / bgra");
    __shared__ __attribute__((aligned(16))) uint8_t xlds[kBlock / 64][64 * 8 * BPP];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = int(threadIdx.x & 63);
    const int units = a.pw >> 3;                                   // 8-pixel units a row
    const int wpr = (units + 63) >> 6;                             // waves a row pair
    const int wid = int(blockIdx.x) * (kBlock / 64) + wave;
    const int rp = wid / wpr, wx = wid - rp * wpr;
    if (rp >= (a.ph >> 1)) return;                                 // (whole waves; the kernel has no barrier)
    const int u = wx * 64 + lane;
    const bool live = u < units;
    const int nlive = min(64, units - wx * 64);                    // wave-uniform
    const uint8_t *sp[3];
    uint8_t *dp;
    rgb_frame_bases(a, a.frame0 + int(blockIdx.z), sp, dp);
    const int sy = a.oy + 2 * rp, sx = a.ox + 8 * u;
    uint32_t yw[2][2] = {{0, 0}, {0, 0}}, uw = 0, vw = 0;
    if (live) {
        const uint32_t *y0 = reinterpret_cast<const uint32_t *>(sp[0] + size_t(sy) * a.src_stride_y + sx);
        const uint32_t *y1 = reinterpret_cast<const uint32_t *>(sp[0] + size_t(sy + 1) * a.src_stride_y + sx);
        yw[0][0] = y0[0]; yw[0][1] = y0[1]; yw[1][0] = y1[0]; yw[1][1] = y1[1];
        const size_t co = size_t(sy >> 1) * a.src_stride_c + (sx >> 1);
        uw = *reinterpret_cast<const uint32_t *>(sp[1] + co);
        vw = *reinterpret_cast<const uint32_t *>(sp[2] + co);
    }
    const RgbCoef &c = a.coef;
    int CR[4], CG[4], CB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int U = int((uw >> (8 * j)) & 255u), V = int((vw >> (8 * j)) & 255u);
        const int cr = __mul24((__mul24(V, c.crv) >> 16) + c.off_r, c.cy) + c.k;
        const int cb = __mul24((__mul24(U, c.cbu) >> 16) + c.off_b, c.cy) + c.k;
        CG[j] = __mul24((__mul24(U, c.cgu) >> 16) + (__mul24(V, c.cgv) >> 16) + c.off_g, c.cy) + c.k;
        CR[j] = a.swap ? cb : cr;                                  // (bgr: the first byte of a pixel is B)
        CB[j] = a.swap ? cr : cb;
    }
    const uint32_t xbase = uint32_t(reinterpret_cast<uintptr_t>((rgb_lds_t *)&xlds[wave][0]));
    const uint32_t xw = xbase + uint32_t(lane) * (8u * BPP);
    constexpr int kChunks = 64 * 8 * BPP / 16;                     // 16-byte chunks of a full wave's row: 96 / 128
    const int live_bytes = nlive * 8 * BPP, live_chunks = live_bytes / 16;   // an odd count of rgb24 units ends in half a chunk: stored by itself
    const bool half = (live_bytes & 8) != 0;
    const int c2 = 64 + lane;
    const uint32_t xr1 = xbase + 16u * uint32_t(lane), xr2 = xbase + 16u * uint32_t(c2 < kChunks ? c2 : 0);
    uint8_t *drow = dp + size_t(2 * rp) * a.dst_stride + size_t(wx) * (64 * 8 * BPP);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int P[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) P[i] = __mul24(int((yw[r][i >> 2] >> (8 * (i & 3))) & 255u), c.cy);
        u32x4_t v1, v2;
        if constexpr (BPP == 3) {
            int v[24];
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[3 * i] = P[i] + CR[i / 2]; v[3 * i + 1] = P[i] + CG[i / 2]; v[3 * i + 2] = P[i] + CB[i / 2]; }
            uint32_t o[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) o[q] = pack4<16>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            rgb_lds_exchange3(xw, o, xr1, xr2, v1, v2);
        } else {
            uint32_t o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = pack4<16>(P[i] + CR[i / 2], P[i] + CG[i / 2], P[i] + CB[i / 2], 0x7fff0000);   // alpha 255
            rgb_lds_exchange4(xw, o, xr1, xr2, v1, v2);
        }
        uint8_t *d = drow + size_t(r) * a.dst_stride;
        typedef uint32_t u32x4_dw __attribute__((ext_vector_type(4), aligned(4)));     // (rows start on a dword, not more)
        typedef uint32_t u32x2_dw __attribute__((ext_vector_type(2), aligned(4)));
        if (lane < live_chunks) *reinterpret_cast<u32x4_dw *>(d + 16 * lane) = v1;
        if (c2 < live_chunks) *reinterpret_cast<u32x4_dw *>(d + 16 * c2) = v2;
        if (half && lane == live_chunks) *reinterpret_cast<u32x2_dw *>(d + 16 * lane) = u32x2_dw{v1.x, v1.y};
        if (half && c2 == live_chunks) *reinterpret_cast<u32x2_dw *>(d + 16 * c2) = u32x2_dw{v2.x, v2.y};
    }
}

template <int BPP, bool LDS, bool XP>
__global__ __launch_bounds__(kBlock, 4) void rgb_strip_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                           const uint32_t *__restrict__ tab, const RgbStripArgs a)
{
    static_assert(BPP == 3 || BPP == 4, "rgb24 / bgr24 or rgba / bgra");
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtin exists in the device pass only
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS ? kBlock / 64 : 1][3][LDS ? kRgbSlot : 16];
    __shared__ __attribute__((aligned(16))) uint8_t xlds[XP ? kBlock / 64 : 1][XP ? 64 * 8 * BPP : 16];   // the waves' output rows in exchange
    const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = int(lb % uint32_t(a.nseg));
    const int sb  = int(lb / uint32_t(a.nseg));
    int g = sb * kBlock + int(threadIdx.x);
    const bool live = g < a.n_items;
    if (!live) g = a.n_items - 1;                  // dead lanes (a suffix of the last workgroup) shadow the last live one
    const int strip = g % a.strips;
    const int frame = g / a.strips;

    // this wave's bases and the lane's 32-bit offsets from them (see frame_offsets in ttscale_kernels.h)
    const uint8_t *wsrc = src;
    uint8_t *wdst = dst;
    uint32_t so[3], doff, src_rec, dst_rec;
    if (a.addr.use_table) {
#pragma unroll
        for (int k = 0; k < 3; ++k) so[k] = a.addr.src_tab[k][frame];
        doff = a.addr.dst_tab[0][frame];
        src_rec = uint32_t(a.src_span); dst_rec = uint32_t(a.dst_span);
    } else {
        const uint32_t f0 = uni(uint32_t(frame));
        const uint64_t sbase = uint64_t(f0) * a.addr.src_pitch, dbase = uint64_t(f0) * a.addr.dst_pitch;
        wsrc += sbase; wdst += dbase;
        src_rec = a.src_span > sbase ? uint32_t(min(a.src_span - sbase, uint64_t(kMaxSpan))) : 0u;
        dst_rec = a.dst_span > dbase ? uint32_t(min(a.dst_span - dbase, uint64_t(kMaxSpan))) : 0u;
#pragma unroll
        for (int k = 0; k < 3; ++k) so[k] = (uint32_t(frame) - f0) * a.addr.src_pitch + a.addr.src_plane[k];
        doff = (uint32_t(frame) - f0) * a.addr.dst_pitch;
    }
    doff += uint32_t(min(strip * 8, a.pw - 8)) * BPP;        // the last strip is slid left, not cut short
    if (!live) doff = kDiscard;                               // the descriptor's range check drops these stores
    // XP: where this lane's two chunks go (row offsets come as the stores' scalar offset), the held-back chunk
    [[maybe_unused]] uint32_t dest1 = kDiscard, dest2 = kDiscard, dest_a = kDiscard, dest_b = kDiscard, xw = 0, xr1 = 0, xr2 = 0;
    [[maybe_unused]] bool split_second = false, any_split = false;
    if constexpr (XP) {
        const int lane = int(threadIdx.x) & 63, xwave = int(threadIdx.x) >> 6;
        constexpr int kChunks = 64 * 8 * BPP / 16;             // 96 (rgb24) or 128 (rgba)
        const uint32_t xbase = uint32_t(reinterpret_cast<uintptr_t>((rgb_lds_t *)&xlds[xwave][0]));
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
    uint32_t offl[2], offu, offv, sell[8][2], coefl[8][2], selc[4][2], coefc[4][2];
    {
        offl[0] = tab[a.hoffl + strip * 2] + so[0];
        offl[1] = tab[a.hoffl + strip * 2 + 1] + so[0];
        const uint32_t oc = tab[a.hoffc + strip];
        offu = oc + so[1]; offv = oc + so[2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) { sell[i][k] = tab[a.sell + (strip * 8 + i) * 2 + k]; coefl[i][k] = tab[a.coefl + (strip * 8 + i) * 2 + k]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) { selc[i][k] = tab[a.selc + (strip * 4 + i) * 2 + k]; coefc[i][k] = tab[a.coefc + (strip * 4 + i) * 2 + k]; }
    }

    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(wsrc), 0, src_rec, 0x00020000);
    const uint64_t dbase64 = reinterpret_cast<uint64_t>(wdst);
    const u32x4_t drs = {uni(uint32_t(dbase64)), uni(uint32_t(dbase64 >> 32) & 0xffffu), uni(dst_rec), 0x00020000u};

    uint32_t ry[8][2], ru[4][2], rv[4][2];           // rings: rows (oldest, .) (., newest) as int16 pairs
#pragma unroll
    for (int i = 0; i < 8; ++i) ry[i][0] = ry[i][1] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) ru[i][0] = ru[i][1] = rv[i][0] = rv[i][1] = 0;

    int lrow = int(uni(tab[a.seginfo + 2 * seg])), crow = int(uni(tab[a.seginfo + 2 * seg + 1]));
    const int lmax = a.src_h_l - 1, cmax = a.src_h_c - 1;
    const int sy = int(uni(uint32_t(a.src_stride_y))), sc = int(uni(uint32_t(a.src_stride_c))), ds = int(uni(uint32_t(a.dst_stride)));
    u32x2_t wl[2], wu, wv;                           // register variant: the next source rows, requested one row ahead
    // LDS variant: what this lane fetches in each plane's DMA, where its windows lie in the slots, and the
    // vector-memory operations issued since the pending luma DMA / since the pending chroma DMAs
    [[maybe_unused]] uint32_t cs[3] = {0, 0, 0}, lposl[2] = {0, 0}, lposu = 0, lposv = 0;
    [[maybe_unused]] int since_l = 0, since_c = 0;
    [[maybe_unused]] const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    {
        const int lo = min(max(lrow, 0), lmax) * sy, co = min(max(crow, 0), cmax) * sc;
        if constexpr (LDS) {
            const int lane = int(threadIdx.x) & 63;
            const bool new_row = lane > 0 && strip == 0 && live;
            uint32_t lbase[3];
            rgb_plan_run(offl[0], offl[1] + 8u, so[0], a.end_rel[0], uint32_t(lmax) * uint32_t(sy), new_row, lane, cs[0], lbase[0]);
            rgb_plan_run(offu, offu + 8u, so[1], a.end_rel[1], uint32_t(cmax) * uint32_t(sc), new_row, lane, cs[1], lbase[1]);
            rgb_plan_run(offv, offv + 8u, so[2], a.end_rel[1], uint32_t(cmax) * uint32_t(sc), new_row, lane, cs[2], lbase[2]);
            auto slot = [&](int k) { return uint32_t(reinterpret_cast<uintptr_t>((rgb_lds_t *)&lds[wave][k][0])); };
            lposl[0] = offl[0] + lbase[0] + slot(0); lposl[1] = offl[1] + lbase[0] + slot(0);
            lposu = offu + lbase[1] + slot(1); lposv = offv + lbase[2] + slot(2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (rgb_lds_t *)&lds[wave][0][0], 16, cs[0], lo, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (rgb_lds_t *)&lds[wave][1][0], 16, cs[1], co, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (rgb_lds_t *)&lds[wave][2][0], 16, cs[2], co, 0, 0);
            since_l = 2; since_c = 0;
        } else {
            wl[0] = __builtin_amdgcn_raw_buffer_load_b64(srs, offl[0], lo, 0);
            wl[1] = __builtin_amdgcn_raw_buffer_load_b64(srs, offl[1], lo, 0);
            wu = __builtin_amdgcn_raw_buffer_load_b64(srs, offu, co, 0);
            wv = __builtin_amdgcn_raw_buffer_load_b64(srs, offv, co, 0);
        }
    }
    // the closed form's constants (wave-uniform)
    const int cy = a.coef.cy, crv = a.coef.crv, cbu = a.coef.cbu, cgu = a.coef.cgu, cgv = a.coef.cgv;
    const int kr = a.coef.off_r * cy + a.coef.k, kg = a.coef.off_g * cy + a.coef.k, kb = a.coef.off_b * cy + a.coef.k;
    const bool swap = a.swap != 0;

    const int y0 = seg * a.seg_rows, y1 = min(y0 + a.seg_rows, a.ph);
    for (int y = y0; y < y1; ++y) {
        const uint32_t *vr = tab + a.vrow + 6 * y;
        const int nl = y == y0 ? 4 : int(uni(vr[0])), nc = y == y0 ? 4 : int(uni(vr[1]));
        for (int n = 0; n < nl; ++n) {                         // wave-uniform
            if constexpr (LDS) {
                rgb_wait_vm(since_l);
                uint32_t w3[2][3];
                rgb_lds_read2(w3, lposl[0] & ~3u, lposl[1] & ~3u);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    wl[j].x = __builtin_amdgcn_alignbyte(w3[j][1], w3[j][0], lposl[j]);
                    wl[j].y = __builtin_amdgcn_alignbyte(w3[j][2], w3[j][1], lposl[j]);
                }
                // the slot is free again (its bytes sit in registers): request the next row into it
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (rgb_lds_t *)&lds[wave][0][0], 16, cs[0], min(max(lrow + 1, 0), lmax) * sy, 0, 0);
                since_l = 0; since_c += 1;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x2_t w = wl[i / 4];
                int acc = dot2(__builtin_amdgcn_perm(w.y, w.x, sell[i][0]), coefl[i][0], 0);
                acc = dot2(__builtin_amdgcn_perm(w.y, w.x, sell[i][1]), coefl[i][1], acc);
                const uint32_t t = hfinish(acc);
                ry[i][0] = __builtin_amdgcn_alignbit(ry[i][1], ry[i][0], 16);
                ry[i][1] = __builtin_amdgcn_alignbit(t, ry[i][1], 16);
            }
            ++lrow;
            if constexpr (!LDS) {
                const int lo = min(max(lrow, 0), lmax) * sy;
                wl[0] = __builtin_amdgcn_raw_buffer_load_b64(srs, offl[0], lo, 0);
                wl[1] = __builtin_amdgcn_raw_buffer_load_b64(srs, offl[1], lo, 0);
            }
        }
        for (int n = 0; n < nc; ++n) {                         // wave-uniform
            if constexpr (LDS) {
                rgb_wait_vm(since_c);
                uint32_t w3[2][3];
                rgb_lds_read2(w3, lposu & ~3u, lposv & ~3u);
                wu.x = __builtin_amdgcn_alignbyte(w3[0][1], w3[0][0], lposu); wu.y = __builtin_amdgcn_alignbyte(w3[0][2], w3[0][1], lposu);
                wv.x = __builtin_amdgcn_alignbyte(w3[1][1], w3[1][0], lposv); wv.y = __builtin_amdgcn_alignbyte(w3[1][2], w3[1][1], lposv);
                const int co = min(max(crow + 1, 0), cmax) * sc;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (rgb_lds_t *)&lds[wave][1][0], 16, cs[1], co, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (rgb_lds_t *)&lds[wave][2][0], 16, cs[2], co, 0, 0);
                since_c = 0; since_l += 2;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int au = dot2(__builtin_amdgcn_perm(wu.y, wu.x, selc[i][0]), coefc[i][0], 0);
                au = dot2(__builtin_amdgcn_perm(wu.y, wu.x, selc[i][1]), coefc[i][1], au);
                int av = dot2(__builtin_amdgcn_perm(wv.y, wv.x, selc[i][0]), coefc[i][0], 0);
                av = dot2(__builtin_amdgcn_perm(wv.y, wv.x, selc[i][1]), coefc[i][1], av);
                const uint32_t tu = hfinish(au), tv = hfinish(av);
                ru[i][0] = __builtin_amdgcn_alignbit(ru[i][1], ru[i][0], 16);
                ru[i][1] = __builtin_amdgcn_alignbit(tu, ru[i][1], 16);
                rv[i][0] = __builtin_amdgcn_alignbit(rv[i][1], rv[i][0], 16);
                rv[i][1] = __builtin_amdgcn_alignbit(tv, rv[i][1], 16);
            }
            ++crow;
            if constexpr (!LDS) {
                const int co = min(max(crow, 0), cmax) * sc;
                wu = __builtin_amdgcn_raw_buffer_load_b64(srs, offu, co, 0);
                wv = __builtin_amdgcn_raw_buffer_load_b64(srs, offv, co, 0);
            }
        }

        // vertical pass: yuv2rgb_X_c_template's 1 << 18 rounding, >> 19, no clip
        const uint32_t vl0 = uni(vr[2]), vl1 = uni(vr[3]), vc0 = uni(vr[4]), vc1 = uni(vr[5]);
        int P[8], CR[4], CG[4], CB[4];
#pragma unroll
        // (every product below has operands within 24 bits -- |Y|, |offsets| < 2^11, constants < 2^18 -- so the
        // full-rate v_mul_i32_i24 / v_mad_i32_i24 serve; a plain `*` costs a quarter-rate v_mul_lo_u32 each)
        for (int i = 0; i < 8; ++i) P[i] = __mul24(dot2(ry[i][1], vl1, dot2(ry[i][0], vl0, 1 << 18)) >> 19, cy);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int U = clip255(dot2(ru[j][1], vc1, dot2(ru[j][0], vc0, 1 << 18)) >> 19);
            const int V = clip255(dot2(rv[j][1], vc1, dot2(rv[j][0], vc0, 1 << 18)) >> 19);
            const int cr = __mul24(__mul24(V, crv) >> 16, cy) + kr;
            const int cb = __mul24(__mul24(U, cbu) >> 16, cy) + kb;
            CG[j] = __mul24((__mul24(U, cgu) >> 16) + (__mul24(V, cgv) >> 16), cy) + kg;
            CR[j] = swap ? cb : cr;                              // (bgr: the first byte of a pixel is B)
            CB[j] = swap ? cr : cb;
        }
        const int yoff = y * ds;
        if constexpr (BPP == 3) {
            int v[24];
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[3 * i] = P[i] + CR[i / 2]; v[3 * i + 1] = P[i] + CG[i / 2]; v[3 * i + 2] = P[i] + CB[i / 2]; }
            uint32_t o[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) o[q] = pack4<16>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);      // (>> 16, clip, pack: two instructions)
            if constexpr (XP) {
                u32x4_t v1, v2;
                rgb_lds_exchange3(xw, o, xr1, xr2, v1, v2);
                store_b128(v1, drs, dest1, yoff);
                store_b128(v2, drs, dest2, yoff);
                if (any_split) {                               // wave-uniform: the one chunk whose halves lie apart
                    const u32x4_t h = split_second ? v2 : v1;
                    store_b64(u32x2_t{h.x, h.y}, drs, dest_a, yoff);
                    store_b64(u32x2_t{h.z, h.w}, drs, dest_b, yoff);
                    since_l += 2; since_c += 2;
                }
            } else {
                store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, drs, doff, yoff);
                store_b64(u32x2_t{o[4], o[5]}, drs, doff + 16u, yoff);
            }
        } else {
            uint32_t o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i] = pack4<16>(P[i] + CR[i / 2], P[i] + CG[i / 2], P[i] + CB[i / 2], 0x7fff0000);   // alpha 255
            if constexpr (XP) {
                u32x4_t v1, v2;
                rgb_lds_exchange4(xw, o, xr1, xr2, v1, v2);
                store_b128(v1, drs, dest1, yoff);
                store_b128(v2, drs, dest2, yoff);
            } else {
                store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, drs, doff, yoff);
                store_b128(u32x4_t{o[4], o[5], o[6], o[7]}, drs, doff + 16u, yoff);
            }
        }
        since_l += 2; since_c += 2;                              // two stores a row, either format
    }
#endif
}

// ---- packed-RGB shrinks: in-place conversion of the rows the column walk left in the frame ----------------
// The many-tap shrinks of the reference's thumbnail chain (src/services/video.js:329-334 ends in an image
// encoder) ran through rgb_generic_kernel at taps^2 multiply-adds per plane and pixel (29 k frames/s for
// 1080p -> 320x180 -> rgb24).  Instead the planar column walk (column_lds_kernel, luma stored unclipped as
// int16, chroma at ceil(w / 2) x h as the RGB-writing context computes it) writes into the destination frame
// itself, ROW BY ROW where the pixels of that row will stand: the w * bpp bytes of picture row y hold
// [w int16 luma | ceil(w / 2) U | ceil(w / 2) V] -- 3 w bytes for an even w.  A row is then converted by one
// workgroup that reads all of it (every thread its pixels' luma and chroma, into registers), meets at a
// barrier, and only then writes: nothing outside the row is touched, so rows and frames are independent
// workgroups, the picture may be any size, rows may be padded, and no memory beyond the output frame is used.
struct RgbScratchArgs {
    uint8_t *dst;
    int32_t  pw, ph, cw;               // picture, chroma columns
    int32_t  canvas_w, canvas_h, px, py, dst_stride;
    int32_t  bpp, swap;
    int32_t  border[3];
    int32_t  frame0;
    int32_t  tpr;                      // threads per canvas row: 64, 128 or 256
    RgbCoef  coef;
    FrameAddr addr;                    // dst side: frame f's canvas origin (dst_plane[0] / dst_tab[0])
};

constexpr int kScratchThreads = 256, kScratchMaxPerThread = 64;     // pictures up to 16,384 columns

__global__ __launch_bounds__(kScratchThreads) void rgb_scratch_convert_kernel(const RgbScratchArgs a)
{
    // a.tpr threads (64, 128 or 256: the host picks by the picture's width) share a canvas row, so a workgroup takes
    // 256 / tpr rows (one tiny workgroup per 320-pixel row was twice as slow as one per four rows)
    const int frame = a.frame0 + int(blockIdx.y), t = int(threadIdx.x) & (a.tpr - 1);
    uint8_t *fb = a.dst + (a.addr.use_table ? size_t(a.addr.dst_tab[0][frame]) : size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[0]);
    const int cy = int(blockIdx.x) * (kScratchThreads / a.tpr) + int(threadIdx.x) / a.tpr;   // canvas row
    const bool row_ok = cy < a.canvas_h;
    uint8_t *crow = fb + size_t(row_ok ? cy : 0) * a.dst_stride;
    const bool in_rows = row_ok && cy >= a.py && cy < a.py + a.ph;
    uint8_t *prow = crow + size_t(a.px) * a.bpp;                   // the picture's bytes of this row: its scratch, then its pixels
    const uint8_t *urow = prow + 2 * size_t(a.pw), *vrow = urow + a.cw;
    const bool aligned = (reinterpret_cast<uintptr_t>(prow) & 1u) == 0;
    int yv[kScratchMaxPerThread], uv[kScratchMaxPerThread];
#pragma unroll
    for (int it = 0; it < kScratchMaxPerThread; ++it) {
        if (it * a.tpr >= a.pw) break;                             // (uniform)
        const int x = it * a.tpr + t;                              // consecutive lanes, consecutive pixels
        int v = 0, c = 0;
        if (in_rows && x < a.pw) {
            const uint8_t *q = prow + 2 * size_t(x);
            v = aligned ? int(*reinterpret_cast<const int16_t *>(q)) : int(int16_t(uint16_t(q[0]) | uint16_t(q[1]) << 8));
            c = int(urow[x >> 1]) | int(vrow[x >> 1]) << 8;
        }
        yv[it] = v; uv[it] = c;
    }
    __syncthreads();                                               // the workgroup's rows sit in registers: they may be overwritten
    if (!row_ok) return;
    if (in_rows) {
#pragma unroll
        for (int it = 0; it < kScratchMaxPerThread; ++it) {
            if (it * a.tpr >= a.pw) break;
            const int x = it * a.tpr + t;
            if (x < a.pw) {
                int r, g, b;
                tts::rgb_pixel(a.coef, yv[it], uv[it] & 255, uv[it] >> 8, r, g, b);
                uint8_t *d = prow + size_t(x) * a.bpp;
                if (a.bpp == 4 && (reinterpret_cast<uintptr_t>(prow) & 3u) == 0)
                    *reinterpret_cast<uint32_t *>(d) = uint32_t(a.swap ? b : r) | uint32_t(g) << 8 | uint32_t(a.swap ? r : b) << 16 | 0xff000000u;
                else rgb_store_pixel(d, a.bpp, a.swap, r, g, b);
            }
        }
    }
    // the pad border of this canvas row
    if (a.canvas_w != a.pw || !in_rows)
        for (int x = t; x < a.canvas_w; x += a.tpr)
            if (!in_rows || x < a.px || x >= a.px + a.pw) rgb_store_pixel(crow + size_t(x) * a.bpp, a.bpp, a.swap, a.border[0], a.border[1], a.border[2]);
}

// The same for rows whose picture bytes start on a dword and hold a multiple of four pixels (the host checks:
// the thumbnail chain does): a thread takes FOUR adjacent pixels -- 8 bytes of luma, 2 + 2 of chroma in, 12 or 16
// bytes of pixels out, as dwords -- instead of one pixel through byte and halfword accesses, which kept the
// texture addresser busier than the memory (82 -> 3x fewer memory instructions).
// PX = 4 or 8 pixels a thread (8 where the picture is a multiple of eight wide: 16 bytes of luma, 4 + 4 of chroma in,
// 24 or 32 bytes of pixels out).
template <int PX>
__global__ __launch_bounds__(kScratchThreads) void rgb_scratch_convert4_kernel(const RgbScratchArgs a)
{
    static_assert(PX == 4 || PX == 8, "pixels a thread");
    const int frame = a.frame0 + int(blockIdx.y), t = int(threadIdx.x) & (a.tpr - 1);
    uint8_t *fb = a.dst + (a.addr.use_table ? size_t(a.addr.dst_tab[0][frame]) : size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[0]);
    const int cy = int(blockIdx.x) * (kScratchThreads / a.tpr) + int(threadIdx.x) / a.tpr;   // canvas row
    const bool row_ok = cy < a.canvas_h;
    uint8_t *crow = fb + size_t(row_ok ? cy : 0) * a.dst_stride;
    const bool in_rows = row_ok && cy >= a.py && cy < a.py + a.ph;
    uint8_t *prow = crow + size_t(a.px) * a.bpp;
    const uint8_t *urow = prow + 2 * size_t(a.pw), *vrow = urow + a.cw;
    const int units = a.pw / PX;
    constexpr int kIters = kScratchMaxPerThread / PX, LW = PX / 2, CW = PX / 4;   // dwords of luma / of each chroma plane a unit
    uint32_t yw[kIters][LW], uw[kIters][CW], vw[kIters][CW];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        if (it * a.tpr >= units) break;                            // (uniform)
        const int q = it * a.tpr + t;
#pragma unroll
        for (int k = 0; k < LW; ++k) yw[it][k] = 0;
#pragma unroll
        for (int k = 0; k < CW; ++k) uw[it][k] = vw[it][k] = 0;
        if (in_rows && q < units) {
            const uint32_t *yp = reinterpret_cast<const uint32_t *>(prow + 2 * PX * size_t(q));
#pragma unroll
            for (int k = 0; k < LW; ++k) yw[it][k] = yp[k];
            if constexpr (PX == 8) {
                uw[it][0] = *reinterpret_cast<const uint32_t *>(urow + 4 * size_t(q));
                vw[it][0] = *reinterpret_cast<const uint32_t *>(vrow + 4 * size_t(q));
            } else {
                uw[it][0] = *reinterpret_cast<const uint16_t *>(urow + 2 * size_t(q));
                vw[it][0] = *reinterpret_cast<const uint16_t *>(vrow + 2 * size_t(q));
            }
        }
    }
    __syncthreads();                                               // the workgroup's rows sit in registers: they may be overwritten
    if (!row_ok) return;
    // Eight pixels a thread: the arithmetic of rgb_unscaled8_kernel (chroma terms
    // shared by a sample's pixels, two-instruction packs) and its whole-sector stores through an LDS exchange -- a
    // wave is 64 adjacent units of ONE row (tpr >= 64), its pixels one string of 1.5 (2) KiB.
    if (PX == 8) {
        if constexpr (PX == 8) {
            __shared__ __attribute__((aligned(16))) uint8_t xlds[kScratchThreads / 64][64 * 8 * 4];
            if (in_rows) {
                const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = int(threadIdx.x & 63);
                const RgbCoef &c = a.coef;
                const uint32_t xbase = uint32_t(reinterpret_cast<uintptr_t>((rgb_lds_t *)&xlds[wave][0]));
                const int ubytes = 8 * a.bpp;                      // bytes of a unit's pixels
                const uint32_t xw = xbase + uint32_t(lane * ubytes);
#pragma unroll
                for (int it = 0; it < kIters; ++it) {
                    const int base = it * a.tpr + (t - lane);      // the wave's first unit (wave-uniform)
                    if (base >= units) break;
                    const int live_bytes = min(64, units - base) * ubytes, live_chunks = live_bytes / 16;
                    const bool half = (live_bytes & 8) != 0;           // an odd count of rgb24 units: the last chunk is half a chunk
                    int CR[4], CG[4], CB[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int U = int((uw[it][0] >> (8 * j)) & 255u), V = int((vw[it][0] >> (8 * j)) & 255u);
                        const int cr = __mul24((__mul24(V, c.crv) >> 16) + c.off_r, c.cy) + c.k;
                        const int cb = __mul24((__mul24(U, c.cbu) >> 16) + c.off_b, c.cy) + c.k;
                        CG[j] = __mul24((__mul24(U, c.cgu) >> 16) + (__mul24(V, c.cgv) >> 16) + c.off_g, c.cy) + c.k;
                        CR[j] = a.swap ? cb : cr;
                        CB[j] = a.swap ? cr : cb;
                    }
                    int P[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t w = yw[it][i >> 1];
                        P[i] = __mul24(int(int16_t(i & 1 ? w >> 16 : w & 0xffffu)), c.cy);      // (unclipped luma, |Y| < 2^12)
                    }
                    u32x4_t v1, v2;
                    const int c2 = 64 + lane;
                    uint8_t *d = prow + size_t(base) * ubytes;
                    if (a.bpp == 3) {                              // (uniform)
                        int v[24];
#pragma unroll
                        for (int i = 0; i < 8; ++i) { v[3 * i] = P[i] + CR[i / 2]; v[3 * i + 1] = P[i] + CG[i / 2]; v[3 * i + 2] = P[i] + CB[i / 2]; }
                        uint32_t o[6];
#pragma unroll
                        for (int q = 0; q < 6; ++q) o[q] = pack4<16>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                        rgb_lds_exchange3(xw, o, xbase + 16u * uint32_t(lane), xbase + 16u * uint32_t(c2 < 96 ? c2 : 0), v1, v2);
                    } else {
                        uint32_t o[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] = pack4<16>(P[i] + CR[i / 2], P[i] + CG[i / 2], P[i] + CB[i / 2], 0x7fff0000);   // alpha 255
                        rgb_lds_exchange4(xw, o, xbase + 16u * uint32_t(lane), xbase + 16u * uint32_t(c2), v1, v2);
                    }
                    typedef uint32_t u32x4_dw __attribute__((ext_vector_type(4), aligned(4)));     // (picture rows start on a dword, not more)
                    typedef uint32_t u32x2_dw __attribute__((ext_vector_type(2), aligned(4)));
                    if (lane < live_chunks) *reinterpret_cast<u32x4_dw *>(d + 16 * lane) = v1;
                    if (c2 < live_chunks) *reinterpret_cast<u32x4_dw *>(d + 16 * c2) = v2;
                    if (half && lane == live_chunks) *reinterpret_cast<u32x2_dw *>(d + 16 * lane) = u32x2_dw{v1.x, v1.y};
                    if (half && c2 == live_chunks) *reinterpret_cast<u32x2_dw *>(d + 16 * c2) = u32x2_dw{v2.x, v2.y};
                }
            }
        }
    } else if (in_rows) {
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            if (it * a.tpr >= units) break;
            const int q = it * a.tpr + t;
            if (q < units) {
                int r[PX], g[PX], b[PX];
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    const uint32_t w = yw[it][i >> 1];
                    const int Y = int(int16_t(i & 1 ? w >> 16 : w & 0xffffu));
                    const int U = int((uw[it][0] >> (8 * (i >> 1))) & 255u), V = int((vw[it][0] >> (8 * (i >> 1))) & 255u);
                    tts::rgb_pixel(a.coef, Y, U, V, r[i], g[i], b[i]);
                    if (a.swap) { const int s = r[i]; r[i] = b[i]; b[i] = s; }
                }
                if (a.bpp == 4) {
                    uint32_t *d = reinterpret_cast<uint32_t *>(prow + 4 * PX * size_t(q));
#pragma unroll
                    for (int i = 0; i < PX; ++i) d[i] = uint32_t(r[i]) | uint32_t(g[i]) << 8 | uint32_t(b[i]) << 16 | 0xff000000u;
                } else {
                    uint32_t *d = reinterpret_cast<uint32_t *>(prow + 3 * PX * size_t(q));
#pragma unroll
                    for (int h = 0; h < PX / 4; ++h) {
                        const int *R = r + 4 * h, *G = g + 4 * h, *B = b + 4 * h;
                        d[3 * h]     = uint32_t(R[0]) | uint32_t(G[0]) << 8 | uint32_t(B[0]) << 16 | uint32_t(R[1]) << 24;
                        d[3 * h + 1] = uint32_t(G[1]) | uint32_t(B[1]) << 8 | uint32_t(R[2]) << 16 | uint32_t(G[2]) << 24;
                        d[3 * h + 2] = uint32_t(B[2]) | uint32_t(R[3]) << 8 | uint32_t(G[3]) << 16 | uint32_t(B[3]) << 24;
                    }
                }
            }
        }
    }
    if (a.canvas_w != a.pw || !in_rows)
        for (int x = t; x < a.canvas_w; x += a.tpr)
            if (!in_rows || x < a.px || x >= a.px + a.pw) rgb_store_pixel(crow + size_t(x) * a.bpp, a.bpp, a.swap, a.border[0], a.border[1], a.border[2]);
}

// ---- packed rgb24 in, yuv420p out: any tap counts, one lane per output sample ---------------------------
// swscale's input side (input.c rgb24ToY_c / rgb24ToUV_half_c, BT.601 limited range, coefficients in 15
// fractional bits) evaluated on the fly in front of the scaler's arithmetic: samples enter the horizontal
// pass as 14-bit values (Y << 6, C << 6; chroma from the SUM of two neighbouring pixels), the pass is
// hScale16To15_c's (>> 13, clip to 32767), the vertical pass the planar one.  Chroma has one row per SOURCE
// row (RGB is not subsampled vertically), so its vertical filter maps src_h rows to ceil(dst_h / 2).
struct RgbInArgs {
    const uint8_t *src;
    uint8_t       *dst;
    const int32_t *lhpos, *lvpos, *chpos, *cvpos;
    const int16_t *lhtaps, *lvtaps, *chtaps, *cvtaps;
    int32_t lht, lvt, cht, cvt;
    int32_t canvas_w[2], canvas_h[2];  // output planes, luma / chroma
    int32_t px[2], py[2], pw[2], ph[2];   // picture rectangle inside them
    int32_t ox[2], oy[2];              // its origin in the scaled plane (crop behind the scale)
    int32_t dst_stride[2], src_stride;
    int32_t fill[3];
    int32_t frame0;
    int32_t coef[9];                   // ry gy by, ru gu bu, rv gv bv
    int32_t  only_plane;               // >= 0: the launch produces this destination plane alone
    int32_t  chr_half;                 // 1: chroma samples are pixel-pair sums (rgb24ToUV_half_c), 0: one per pixel (rgb24ToUV_c); filter.h rgb_in_half_chroma()
    int32_t  pb;                       // bytes a source pixel: 3 (rgb24, bgr24) or 4 (rgba, bgra: the fourth is ignored); coef is by byte of a pixel
    uint32_t src_origin, src_extent;   // bytes from a frame's first byte to the pre-cropped origin / to one past its last byte
    int32_t  range15;                  // 1: the context writes yuvj420p -- lumRangeToJpeg_c / chrRangeToJpeg_c on the 15-bit rows between the passes
    int32_t  vs_on, vs_x, vs_y, vs_w, vs_h, vs_fill[3];   // a pad on RGB pixels in front of the conversion: the picture's rectangle in the canvas the conversion reads, the colour by byte of a pixel
    FrameAddr addr;
};

// Pixel (x, y) of the picture the conversion reads: the pre-cropped source's or, behind a pad on RGB pixels, the canvas's
// (the picture where it was put, the pad colour around it)
__device__ __forceinline__ void rgbin_pixel(const RgbInArgs &a, const uint8_t *sp, int x, int y, int &c0, int &c1, int &c2)
{
    if (a.vs_on) {
        x -= a.vs_x; y -= a.vs_y;
        if (x < 0 || y < 0 || x >= a.vs_w || y >= a.vs_h) { c0 = a.vs_fill[0]; c1 = a.vs_fill[1]; c2 = a.vs_fill[2]; return; }
    }
    const uint8_t *q = sp + size_t(y) * size_t(a.src_stride) + a.pb * x;
    c0 = q[0]; c1 = q[1]; c2 = q[2];
}

// Where frame `frame` lies: its first source byte (dword loads must stay inside [fb, fb + src_extent)) and plane
// `plane` of its output -- at a fixed pitch from the bases, or through the launch's offset tables (tts_run).
__device__ __forceinline__ void rgbin_frame(const RgbInArgs &a, int frame, int plane, const uint8_t *&fb, uint8_t *&dp)
{
    if (a.addr.use_table) {
        fb = a.src + a.addr.src_tab[0][frame];                  // (at most kMaxTableFrames a launch: frame0 is 0)
        dp = a.dst + a.addr.dst_tab[plane][frame];
    } else {
        fb = a.src + size_t(frame) * a.addr.src_pitch + a.addr.src_plane[0];
        dp = a.dst + size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane];
    }
}

__global__ __launch_bounds__(kBlock) void rgbin_generic_kernel(const RgbInArgs a)
{
    const int plane = int(blockIdx.z) % 3, frame = a.frame0 + int(blockIdx.z) / 3;
    if (a.only_plane >= 0 && plane != a.only_plane) return;
    const int ci = plane ? 1 : 0;
    const int x = int(blockIdx.x) * 64 + int(threadIdx.x & 63);
    const int y = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    if (x >= a.canvas_w[ci] || y >= a.canvas_h[ci]) return;
    const uint8_t *fb;
    uint8_t *dp;
    rgbin_frame(a, frame, plane, fb, dp);
    const uint8_t *sp = fb + a.src_origin;
    int out = a.fill[plane];
    if (y >= a.py[ci] && y < a.py[ci] + a.ph[ci] && x >= a.px[ci] && x < a.px[ci] + a.pw[ci]) {
        const int vx = a.ox[ci] + x - a.px[ci], vy = a.oy[ci] + y - a.py[ci];
        const int32_t *hpos = ci ? a.chpos : a.lhpos, *vpos = ci ? a.cvpos : a.lvpos;
        const int ht = ci ? a.cht : a.lht, vt = ci ? a.cvt : a.lvt;
        const int16_t *hc = (ci ? a.chtaps : a.lhtaps) + size_t(vx) * ht, *vc = (ci ? a.cvtaps : a.lvtaps) + size_t(vy) * vt;
        const int hp = hpos[vx], vp = vpos[vy];
        const int cr = a.coef[3 * plane], cg = a.coef[3 * plane + 1], cb = a.coef[3 * plane + 2];
        int val = 64 << 12;
        for (int j = 0; j < vt; ++j) {
            int acc = 0;
            for (int k = 0; k < ht; ++k) {
                int s14, q0, q1, q2;
                if (ci == 0) {
                    rgbin_pixel(a, sp, hp + k, vp + j, q0, q1, q2);
                    s14 = (cr * q0 + cg * q1 + cb * q2 + (32 << 14) + (1 << 8)) >> 9;
                } else if (a.chr_half) {
                    int r0, r1, r2;
                    rgbin_pixel(a, sp, 2 * (hp + k), vp + j, q0, q1, q2);
                    rgbin_pixel(a, sp, 2 * (hp + k) + 1, vp + j, r0, r1, r2);
                    s14 = (cr * (q0 + r0) + cg * (q1 + r1) + cb * (q2 + r2) + (256 << 15) + (1 << 9)) >> 10;
                } else {
                    rgbin_pixel(a, sp, hp + k, vp + j, q0, q1, q2);
                    s14 = (cr * q0 + cg * q1 + cb * q2 + (256 << 14) + (1 << 8)) >> 9;
                }
                acc += s14 * int(hc[k]);
            }
            acc >>= 13;
            acc = acc < 32767 ? acc : 32767;
            if (a.range15) acc = ci ? (min(acc, 30775) * 4663 - 9289992) >> 12 : (min(acc, 30189) * 19077 - 39057361) >> 14;
            val += acc * int(vc[j]);
        }
        val >>= 19;                                            // (a 1-tap vertical filter gives (t + 64) >> 7: the same value)
        out = val < 0 ? 0 : val > 255 ? 255 : val;
    }
    dp[size_t(y) * size_t(a.dst_stride[ci]) + x] = uint8_t(out);
}

// ---- bgr24 in at the same size: swscale's special converter -------------------------------------------------
// swscale_unscaled.c "bgr24toYV12" -> rgb2rgb_template.c ff_rgb24toyv12_c() (filter.h bgr24_special_converter()): no
// 14-bit samples, no filters -- luma ((ry r + gy g + by b) >> 15) + 16 per pixel, chroma ((ru r + gu g + bu b) >> 15)
// + 128 from the left pixel of each pair of the even source rows, sums unsigned, the byte store keeps the low 8 bits.
// One lane per output sample, crop / pad behind it as in the generic kernel.
__global__ __launch_bounds__(kBlock) void bgr24_yv12_kernel(const RgbInArgs a)
{
    const int plane = int(blockIdx.z) % 3, frame = a.frame0 + int(blockIdx.z) / 3;
    if (a.only_plane >= 0 && plane != a.only_plane) return;
    const int ci = plane ? 1 : 0;
    const int x = int(blockIdx.x) * 64 + int(threadIdx.x & 63);
    const int y = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    if (x >= a.canvas_w[ci] || y >= a.canvas_h[ci]) return;
    const uint8_t *fb;
    uint8_t *dp;
    rgbin_frame(a, frame, plane, fb, dp);
    uint32_t out = uint32_t(a.fill[plane]);
    if (y >= a.py[ci] && y < a.py[ci] + a.ph[ci] && x >= a.px[ci] && x < a.px[ci] + a.pw[ci]) {
        const int vx = (a.ox[ci] + x - a.px[ci]) << ci, vy = (a.oy[ci] + y - a.py[ci]) << ci;   // chroma: pixel (2 x, 2 y)
        int q0, q1, q2;
        rgbin_pixel(a, fb + a.src_origin, vx, vy, q0, q1, q2);
        const uint32_t s = uint32_t(a.coef[3 * plane]) * uint32_t(q0) + uint32_t(a.coef[3 * plane + 1]) * uint32_t(q1) + uint32_t(a.coef[3 * plane + 2]) * uint32_t(q2);   // (coef: by byte of a pixel)
        out = ((s >> 15) + (plane ? 128u : 16u)) & 255u;
    }
    dp[size_t(y) * size_t(a.dst_stride[ci]) + x] = uint8_t(out);
}

// The same converter where nothing follows it (no crop or pad behind, width >= 8): a lane owns 8 pixels of two rows --
// 24 source bytes a row as six dwords at any byte address, 8 luma bytes a row, 4 + 4 chroma bytes from the upper row.  The
// last lane of a row is slid left to end with the row (the width is even: so is its first pixel; same values, written twice).
__global__ __launch_bounds__(kBlock) void bgr24_yv12_rows_kernel(const RgbInArgs a)
{
    typedef uint32_t u32_any __attribute__((aligned(1)));
    const int frame = a.frame0 + int(blockIdx.z);
    const int w = a.canvas_w[0], h = a.canvas_h[0];
    const int gx = int(blockIdx.x) * 64 + int(threadIdx.x & 63), yp = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    if (gx * 8 >= w || 2 * yp >= h) return;
    const int x0 = min(gx * 8, w - 8);
    const uint8_t *fb;
    uint8_t *dy, *du, *dv;
    rgbin_frame(a, frame, 0, fb, dy);
    rgbin_frame(a, frame, 1, fb, du);
    rgbin_frame(a, frame, 2, fb, dv);
    for (int rr = 0; rr < 2; ++rr) {
        const int y = 2 * yp + rr;
        if (y >= h) break;                                       // (an odd height: the last row is an upper row)
        const u32_any *q = reinterpret_cast<const u32_any *>(fb + a.src_origin + size_t(y) * size_t(a.src_stride) + 3 * size_t(x0));
        uint32_t d[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = q[k];
        auto byte = [&](int i) { return (d[i >> 2] >> (8 * (i & 3))) & 255u; };
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t s = uint32_t(a.coef[0]) * byte(3 * i) + uint32_t(a.coef[1]) * byte(3 * i + 1) + uint32_t(a.coef[2]) * byte(3 * i + 2);
            o[i >> 2] |= ((s >> 15) + 16u) << (8 * (i & 3));     // (luma: at most 235)
        }
        u32_any *py = reinterpret_cast<u32_any *>(dy + size_t(y) * size_t(a.dst_stride[0]) + x0);
        py[0] = o[0]; py[1] = o[1];
        if (rr == 0) {
            uint32_t cu = 0, cv = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t b = byte(6 * i), g = byte(6 * i + 1), r = byte(6 * i + 2);
                cu |= ((((uint32_t(a.coef[3]) * b + uint32_t(a.coef[4]) * g + uint32_t(a.coef[5]) * r) >> 15) + 128u) & 255u) << (8 * i);
                cv |= ((((uint32_t(a.coef[6]) * b + uint32_t(a.coef[7]) * g + uint32_t(a.coef[8]) * r) >> 15) + 128u) & 255u) << (8 * i);
            }
            *reinterpret_cast<u32_any *>(du + size_t(yp) * size_t(a.dst_stride[1]) + (x0 >> 1)) = cu;
            *reinterpret_cast<u32_any *>(dv + size_t(yp) * size_t(a.dst_stride[1]) + (x0 >> 1)) = cv;
        }
    }
}

// ---- packed rgb24 in, tiled: a workgroup owns 64 x TH output samples of one plane ---------------------------
// The kernel above converts every source pixel once per TAP PAIR that touches it (taps_h x taps_v times per
// output sample).  Here a workgroup converts the source samples its tile needs ONCE into LDS (rgb24ToY_c /
// rgb24ToUV_half_c, 14-bit values), runs hScale16To15_c over them into a second LDS array (one 15-bit row per
// source row of the tile), and the vertical pass reads that: the same arithmetic in the same order per
// sample, so the same bytes.  The host sizes the tile (rows `th`) so that both arrays fit `lds_bytes`, from
// the worst tile of the plane class (`sx_max` source samples a row, `sy_max` source rows).
struct RgbInTileArgs {
    RgbInArgs g;
    int32_t th[2];                     // output rows per tile, luma / chroma
    int32_t sx_max[2], sy_max[2];      // LDS array extents
    int32_t tiles_y[2];                // tiles down the canvas
    int32_t raw_w[2];                  // bytes of a row of the raw-bytes LDS array (a multiple of 4)
};

// MAXT: an upper bound of every tap count known at compile time (loops unrolled under a predicate), 0 = any count
template <int MAXT>
__global__ __launch_bounds__(kBlock) void rgbin_tile_kernel(const RgbInTileArgs t)
{
    extern __shared__ __attribute__((aligned(16))) int16_t rgbin_lds[];
    const RgbInArgs &a = t.g;
    const int plane = int(blockIdx.z) % 3, frame = a.frame0 + int(blockIdx.z) / 3;
    if (a.only_plane >= 0 && plane != a.only_plane) return;        // (whole workgroups)
    const int ci = plane ? 1 : 0;
    if (int(blockIdx.y) >= t.tiles_y[ci]) return;                  // (the grid is sized for the luma class; whole workgroups leave)
    const int x0 = int(blockIdx.x) * 64, y0 = int(blockIdx.y) * t.th[ci];
    if (x0 >= a.canvas_w[ci]) return;
    const int x1 = min(x0 + 64, a.canvas_w[ci]), y1 = min(y0 + t.th[ci], a.canvas_h[ci]);
    const uint8_t *fb;
    uint8_t *dp;
    rgbin_frame(a, frame, plane, fb, dp);
    const uint8_t *sp = fb + a.src_origin, *src_lo = fb, *src_hi = fb + a.src_extent;
    // the part of the tile that lies inside the picture (workgroup-uniform)
    const int ix0 = max(x0, a.px[ci]), ix1 = min(x1, a.px[ci] + a.pw[ci]);
    const int iy0 = max(y0, a.py[ci]), iy1 = min(y1, a.py[ci] + a.ph[ci]);
    const bool any = ix0 < ix1 && iy0 < iy1;
    const int32_t *hpos = ci ? a.chpos : a.lhpos, *vpos = ci ? a.cvpos : a.lvpos;
    const int ht = ci ? a.cht : a.lht, vt = ci ? a.cvt : a.lvt;
    const int16_t *htaps = ci ? a.chtaps : a.lhtaps, *vtaps = ci ? a.cvtaps : a.lvtaps;
    const int sxm = t.sx_max[ci], sym = t.sy_max[ci], raww = t.raw_w[ci];
    int16_t *s14 = rgbin_lds;                                      // [sy_max][sx_max]
    int16_t *hres = s14 + sym * sxm;                               // [sy_max][64]
    int16_t *vrow = hres + sym * 64;                               // [th][1 + vt]: first source row (tile-relative) and taps of each output row
    uint8_t *raw = reinterpret_cast<uint8_t *>(rgbin_lds) + ((size_t(sym) * (sxm + 64) + size_t(t.th[ci]) * (vt + 1)) * 2 + 3 & ~size_t(3));   // [sy_max][raw_w] source bytes
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = int(threadIdx.x & 63);
    const int waves = int(blockDim.x >> 6);
    const int nb = ci && a.chr_half ? 2 * a.pb : a.pb;              // bytes per source sample: a pixel, a pixel pair
    const int bias = ci ? (256 << 14) + (1 << 8) : (32 << 14) + (1 << 8);   // rgb24ToUV_c / rgb24ToY_c
    int sxlo = 0, nsx = 0, sylo = 0, nsy = 0;
    if (any) {
        const int vxa = a.ox[ci] + ix0 - a.px[ci], vxb = a.ox[ci] + ix1 - 1 - a.px[ci];
        const int vya = a.oy[ci] + iy0 - a.py[ci], vyb = a.oy[ci] + iy1 - 1 - a.py[ci];
        sxlo = hpos[vxa]; nsx = hpos[vxb] + ht - sxlo;            // (positions do not decrease along an axis: host-checked)
        sylo = vpos[vya]; nsy = vpos[vyb] + vt - sylo;
        // the vertical taps of the tile's rows, for phase C
        for (int yr = wave; yr < iy1 - iy0; yr += waves) {
            const int vy = a.oy[ci] + iy0 + yr - a.py[ci];
            for (int j = lane; j <= vt; j += 64) vrow[(iy0 - y0 + yr) * (vt + 1) + j] = j ? vtaps[size_t(vy) * vt + j - 1] : int16_t(vpos[vy] - sylo);
        }
        // A1: the tile's source bytes as aligned dwords, coalesced (byte loads of 3-byte pixels kept the texture
        // addresser 60 % busy by themselves); a dword that would reach outside the frame is assembled from bytes
        for (int r = wave; r < nsy; r += waves) {
            const uint8_t *g0 = sp + size_t(sylo + r) * size_t(a.src_stride) + nb * sxlo;
            const int mis = int(reinterpret_cast<uintptr_t>(g0) & 3u);
            const int ndw = (mis + nb * nsx + 3) >> 2;
            uint32_t *lrow = reinterpret_cast<uint32_t *>(raw + r * raww);
            for (int i = lane; i < ndw; i += 64) {
                const uint8_t *q = g0 - mis + 4 * i;
                uint32_t w;
                if (q >= src_lo && q + 4 <= src_hi) w = *reinterpret_cast<const uint32_t *>(q);
                else {
                    w = 0;
                    for (int k = 0; k < 4; ++k) if (q + k >= src_lo && q + k < src_hi) w |= uint32_t(q[k]) << (8 * k);
                }
                lrow[i] = w;
            }
        }
    }
    __syncthreads();
    if (any) {
        // A2: rgb24ToY_c / rgb24ToUV_half_c once per source sample of the tile
        const int cr = a.coef[3 * plane], cg = a.coef[3 * plane + 1], cb = a.coef[3 * plane + 2];
        for (int r = wave; r < nsy; r += waves) {
            const uint8_t *g0 = sp + size_t(sylo + r) * size_t(a.src_stride) + nb * sxlo;
            const uint8_t *q0 = raw + r * raww + int(reinterpret_cast<uintptr_t>(g0) & 3u);
            for (int c = lane; c < nsx; c += 64) {
                const uint8_t *q = q0 + nb * c;
                const int v = nb == a.pb ? (cr * q[0] + cg * q[1] + cb * q[2] + bias) >> 9
                                         : (cr * (q[0] + q[a.pb]) + cg * (q[1] + q[a.pb + 1]) + cb * (q[2] + q[a.pb + 2]) + (256 << 15) + (1 << 9)) >> 10;
                s14[r * sxm + c] = int16_t(v);
            }
        }
    }
    __syncthreads();
    const int x = x0 + lane;
    const bool inx = x >= ix0 && x < ix1;
    if (any && inx) {
        // B: hScale16To15_c for every (source row, output column) of the tile; a lane keeps its column and its taps
        const int vx = a.ox[ci] + x - a.px[ci];
        const int16_t *hc = htaps + size_t(vx) * ht;
        const int hp = hpos[vx] - sxlo;
        if constexpr (MAXT > 0) {
            int tap[MAXT];
#pragma unroll
            for (int k = 0; k < MAXT; ++k) tap[k] = k < ht ? int(hc[k]) : 0;
            for (int r = wave; r < nsy; r += waves) {
                const int16_t *srow = s14 + r * sxm + hp;
                int acc = 0;
#pragma unroll
                for (int k = 0; k < MAXT; ++k) if (k < ht) acc += int(srow[k]) * tap[k];      // (workgroup-uniform predicate)
                acc >>= 13;
                hres[r * 64 + lane] = int16_t(acc < 32767 ? acc : 32767);
            }
        } else {
            for (int r = wave; r < nsy; r += waves) {
                const int16_t *srow = s14 + r * sxm + hp;
                int acc = 0;
                for (int k = 0; k < ht; ++k) acc += int(srow[k]) * int(hc[k]);
                acc >>= 13;
                hres[r * 64 + lane] = int16_t(acc < 32767 ? acc : 32767);
            }
        }
    }
    __syncthreads();
    // C: vertical pass, clip, store: a wave takes an output row; border samples take the pad colour
    if (x < x1) {
        for (int y = y0 + wave; y < y1; y += waves) {
            int out = a.fill[plane];
            if (inx && y >= iy0 && y < iy1) {
                const int16_t *vc = vrow + (y - y0) * (vt + 1) + 1;
                const int16_t *hcol = hres + int(vc[-1]) * 64 + lane;
                int val = 64 << 12;
                if constexpr (MAXT > 0) {
#pragma unroll
                    for (int j = 0; j < MAXT; ++j) if (j < vt) val += int(hcol[j * 64]) * int(vc[j]);
                } else {
                    for (int j = 0; j < vt; ++j) val += int(hcol[j * 64]) * int(vc[j]);
                }
                val >>= 19;
                out = val < 0 ? 0 : val > 255 ? 255 : val;
            }
            dp[size_t(y) * size_t(a.dst_stride[ci]) + x] = uint8_t(out);
        }
    }
}

// ---- packed rgb24 in, same size out: the conversion alone as a streaming walk ------------------------------
// `-pix_fmt rgb24` in front of a chain with no scale in it is swscale's ordinary scaler with one-tap filters in
// three of its four passes -- luma is converted sample by sample -- but the chroma planes still shrink 2 : 1 DOWN
// the picture (an RGB source has a chroma sample on every row), an 8-tap bicubic.  A lane owns 8 adjacent pixel
// columns (4 chroma columns) and walks a segment of source rows: 24 bytes in per row, 8 luma bytes out at once, the
// row's 4 + 4 chroma samples (15-bit, as hScale16To15_c leaves them) into a register ring of 8 rows, kept as row
// pairs per dword; a chroma output row is emitted when its last source row has entered: 8 sums of 4 v_dot2 over
// the ring with the row's taps arranged by ring slot (host table), two packs, 4 + 4 bytes out.  Segments start up
// to 7 rows early to fill the ring.  The host sends here only what fits (rgbin_plan_create / rgbin_launch).
struct RgbInWalkArgs {
    // (the table -- [ny] last source row of chroma output row y, then [ny][4] its taps by ring slot (source row & 7) as
    // int16 pairs -- and the frames are __restrict__ kernel arguments of their own: wave-uniform table reads are then
    // scalar loads instead of a vector load, a full wait and a v_readfirstlane per chroma output row)
    int32_t w, h, ny, vt;              // picture, chroma output rows, vertical chroma taps (<= 8)
    int32_t seg_rows, nseg;
    int32_t src_stride, dst_stride[2];
    uint32_t src_origin;
    int32_t coef[9];
    int32_t frame0;
    FrameAddr addr;
};

__device__ __forceinline__ int rgbin_lower_bound(const int32_t *last, int n, int v)    // first y with last[y] >= v (uniform)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (last[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// RNG: the context writes yuvj420p (tts_geom.sc_range): swscale.c lumRangeToJpeg_c / chrRangeToJpeg_c on the 15-bit samples
template <bool RNG>
__global__ __launch_bounds__(kBlock) void rgbin_walk_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int32_t *__restrict__ tab, const RgbInWalkArgs a)
{
    const int32_t *last = tab;
    const uint32_t *vco = reinterpret_cast<const uint32_t *>(tab + a.ny);
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = int(threadIdx.x & 63);
    const int units = a.w >> 3, wpr = (units + 63) >> 6;           // 8-pixel units, waves across the picture
    const int wid = int(blockIdx.x) * (kBlock / 64) + wave;
    const int seg = wid / wpr, wx = wid - seg * wpr;
    if (seg >= a.nseg) return;                                     // (whole waves; no barrier in this kernel)
    const int u = min(wx * 64 + lane, units - 1);                  // dead lanes shadow the last unit and store nothing
    const bool live = wx * 64 + lane < units;
    const int frame = a.frame0 + int(blockIdx.z);
    const uint8_t *sp;                                             // (wave-uniform: the frame's pre-cropped origin; lanes add 32-bit offsets)
    uint8_t *dy, *du, *dv;
    if (a.addr.use_table) {                                        // tts_run: per-frame offsets (at most kMaxTableFrames a launch: frame0 is 0)
        sp = src + a.addr.src_tab[0][frame];
        dy = dst + a.addr.dst_tab[0][frame]; du = dst + a.addr.dst_tab[1][frame]; dv = dst + a.addr.dst_tab[2][frame];
    } else {
        sp = src + size_t(frame) * a.addr.src_pitch + a.addr.src_plane[0];
        uint8_t *dpf = dst + size_t(frame) * a.addr.dst_pitch;
        dy = dpf + a.addr.dst_plane[0]; du = dpf + a.addr.dst_plane[1]; dv = dpf + a.addr.dst_plane[2];
    }
    sp += a.src_origin;
    // a lane's 24 bytes of a row as three 8-byte loads the compiler cannot merge (it is not told that the offsets are 8
    // apart): what counts for the waits below is the NUMBER of loads issued behind a row's own
    uint32_t lo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = 24u * uint32_t(u) + 8u * k; asm volatile("" : "+v"(lo[k])); }
    dy += 8 * size_t(u); du += 4 * size_t(u); dv += 4 * size_t(u);
    const int ra = seg * a.seg_rows, rb = min(ra + a.seg_rows, a.h);
    int y = rgbin_lower_bound(last, a.ny, ra);
    const int yend = rgbin_lower_bound(last, a.ny, rb);
    const int rstart = y < yend ? min(int(last[y]) - (a.vt - 1), ra) : ra;
    int next_last = y < yend ? int(last[y]) : -1;
    const int ry = a.coef[0], gy = a.coef[1], by = a.coef[2], ru = a.coef[3], gu = a.coef[4], bu = a.coef[5], rv = a.coef[6], gv = a.coef[7], bv = a.coef[8];
    uint32_t ring[2][4][4];                                        // plane, chroma column, row pair (slots 2k, 2k + 1)
#pragma unroll
    for (int i = 0; i < 32; ++i) (&ring[0][0][0])[i] = 0;
    // Two rows in flight and the output stores hidden from the compiler (ttscale_device.h gstore_b64): the wait in front of a
    // row's bytes is then vmcnt(3) -- the loads of the row behind it by the compiler's count, in fact the youngest three of
    // those and of the stores issued since.  (One row in flight, stores the compiler sees inside the variable-trip loop
    // below: vmcnt(0) every row, i.e. the acknowledgement of the stores just issued.)  Every pass of the unrolled loop
    // issues its loads, run or skipped, so that no path joins another with loads in flight on one side only.
    u32x2_t nx[2][3];
    auto fetch = [&](int r, u32x2_t (&raw)[3]) {
        const uint8_t *row = sp + size_t(r) * a.src_stride;        // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 3; ++k) raw[k] = *reinterpret_cast<const u32x2_t *>(row + lo[k]);
    };
    fetch(min(rstart + (rstart & 1), rb - 1), nx[0]);
    fetch(min(rstart + 1 - (rstart & 1), rb - 1), nx[1]);
    for (int r0 = rstart & ~7; r0 < rb; r0 += 8) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int r = r0 + s;
            const bool skip = r < rstart || r >= rb;               // wave-uniform
            uint32_t w[6];
            if (!skip) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { w[2 * k] = nx[s & 1][k].x; w[2 * k + 1] = nx[s & 1][k].y; }
            }
            fetch(min(max(r + 2, rstart + ((r - rstart) & 1)), rb - 1), nx[s & 1]);
            if (skip) continue;
            int R[8], G[8], B[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b0 = 3 * i;
                R[i] = int((w[b0 >> 2] >> (8 * (b0 & 3))) & 255u);
                G[i] = int((w[(b0 + 1) >> 2] >> (8 * ((b0 + 1) & 3))) & 255u);
                B[i] = int((w[(b0 + 2) >> 2] >> (8 * ((b0 + 2) & 3))) & 255u);
            }
            if (r >= ra) {
                // rgb24ToY_c, hScale16To15_c with its one tap of 1 << 14 (x 2, clip), yuv2plane1_8
                int v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int s14 = (__mul24(ry, R[i]) + __mul24(gy, G[i]) + __mul24(by, B[i]) + (32 << 14) + (1 << 8)) >> 9;
                    int t15 = min(2 * s14, 32767);
                    if constexpr (RNG) t15 = (__mul24(min(t15, 30189), 19077) - 39057361) >> 14;
                    v[i] = t15 + 64;
                }
                const u32x2_t o = {pack4<7>(v[0], v[1], v[2], v[3]), pack4<7>(v[4], v[5], v[6], v[7])};
                if (live) gstore_b64(dy + size_t(r) * a.dst_stride[0], o);
            }
            // rgb24ToUV_half_c on the four pixel pairs, hScale16To15_c (one tap), into ring slot r & 7 (= s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sr = R[2 * j] + R[2 * j + 1], sg = G[2 * j] + G[2 * j + 1], sb = B[2 * j] + B[2 * j + 1];
                const int u14 = (__mul24(ru, sr) + __mul24(gu, sg) + __mul24(bu, sb) + (256 << 15) + (1 << 9)) >> 10;
                const int v14 = (__mul24(rv, sr) + __mul24(gv, sg) + __mul24(bv, sb) + (256 << 15) + (1 << 9)) >> 10;
                int cu = min(2 * u14, 32767), cv = min(2 * v14, 32767);
                if constexpr (RNG) { cu = (__mul24(min(cu, 30775), 4663) - 9289992) >> 12; cv = (__mul24(min(cv, 30775), 4663) - 9289992) >> 12; }
                const uint32_t tu = uint32_t(cu), tv = uint32_t(cv);
                ring[0][j][s / 2] = (s & 1) ? __builtin_amdgcn_perm(tu, ring[0][j][s / 2], 0x05040100) : __builtin_amdgcn_perm(ring[0][j][s / 2], tu, 0x07060100);
                ring[1][j][s / 2] = (s & 1) ? __builtin_amdgcn_perm(tv, ring[1][j][s / 2], 0x05040100) : __builtin_amdgcn_perm(ring[1][j][s / 2], tv, 0x07060100);
            }
            while (next_last == r) {                               // wave-uniform: chroma output rows this source row completes
                uint32_t vk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) vk[k] = uni(vco[4 * y + k]);
                int acc[2][4];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int t = 64 << 12;
#pragma unroll
                        for (int k = 0; k < 4; ++k) t = dot2(ring[pl][j][k], vk[k], t);
                        acc[pl][j] = t;
                    }
                const uint32_t ou = pack4<19>(acc[0][0], acc[0][1], acc[0][2], acc[0][3]), ov = pack4<19>(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
                if (live) {
                    gstore_b32(du + size_t(y) * a.dst_stride[1], ou);
                    gstore_b32(dv + size_t(y) * a.dst_stride[1], ov);
                }
                ++y;
                next_last = y < yend ? int(last[y]) : -1;
            }
        }
    }
}

// ---- packed rgb24 in, enlarged: the strip walk on 16-bit samples -----------------------------------------------
// rgbin_tile_kernel spends its time between barriers.  For horizontal enlargements (at most 4 luma taps
// per axis: the portrait chains and scale=w=720:h=1280 from an RGB source) a lane owns 8 luma output columns and the 4
// chroma columns under them and walks a segment of SOURCE rows, as rgbin_walk_kernel does.  An enlargement reads chroma
// at FULL width (filter.h rgb_in_half_chroma(): rgb24ToUV_c, one sample per pixel), so the chroma planes shrink -- by up
// to 2 : 1 across (5-8 taps) and, an RGB source having chroma on every row, by 2 : 1 more than luma down the picture:
//   lanes are the flattened (frame, strip) index -- a 1080-column picture has 135 strips, 2.1 waves a row: per row a
//   third of the lanes would idle -- so a wave's 64 strips may lie in two (narrow pictures: several) frames; what each
//   lane converts and where its strip's windows start in the wave's LDS rows comes from a host table per wave start;
//   per source row the wave converts the runs of pixels its lanes' windows need ONCE, lane i PXL pixels of them, into
//   three int16 rows (Y, U, V) in the wave's own LDS -- in float: 8-bit channels times 15-bit coefficients, scaled by
//   2^-9, are exact in fp32's 24 bits, so rgb24ToY_c / rgb24ToUV_c's (sum + bias) >> 9 is three v_pk_fma_f32 per pixel
//   PAIR and plane, and adding 2^23 - 1/2 to (2 sum + 1) / 1024 leaves floor(sum / 512) in the low mantissa bits (the
//   1 / 1024 keeps the sum off the rounding ties); each row is stored TWICE, the second copy one sample on, so that
//   every tap pair of every window is one aligned dword whatever the window's parity;
//   hScale16To15_c is then two v_dot2 per luma column and CHD per chroma column (>> 13, clip) into register rings kept
//   by ring slot (luma: source row & 3; chroma: & 7);
//   output rows whose last source row this was leave: luma 8 sums of 2 v_dot2, chroma 4 + 4 sums of 4, packed.
// No barrier: a wave's LDS rows are its own, its LDS operations complete in order.
struct RgbInStripArgs {
    uint32_t o_strip, o_wave, o_lastl, o_vcol, o_lastc, o_vcoc;        // dword offsets into tab
    int32_t strips, total, nframes;                                    // 8-column strips a row; lanes = nframes x strips
    int32_t oh, och, lvt, cvt, h;                                      // output rows; vertical taps; source rows
    int32_t seg_rows, nseg;
    int32_t src_stride, dst_stride[2];
    uint32_t src_origin;
    float cf[3][4];                                                    // per plane: coefficients of R, G, B (x 2^-9) and (2 bias + 1) / 1024
    int32_t range15;                                                   // the context writes yuvj420p (tts_geom.sc_range): the range step on its 15-bit rows
    FrameAddr addr;
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// PXL: pixels of the wave's runs a lane converts per row (the host takes the smallest of 4 / 6 / 8 that covers every
// wave's runs: all 64 lanes execute the conversion); CHD: tap pairs of a chroma window (taps <= 4, 6, 8).
// src / dst / tab are __restrict__ kernel arguments so that the wave-uniform table reads (vertical taps, last source
// row of an output row) are scalar loads: as vector loads each output row waited out two dependent round trips to
// memory -- `s_waitcnt vmcnt(0)` in front of a v_readfirstlane -- which is where the first version spent 65 % of its
// wave cycles.
template <int PXL, int CHD>
__global__ __launch_bounds__(kBlock, 3) void rgbin_strip_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                const int32_t *__restrict__ tab, const RgbInStripArgs a)
{
    static_assert(PXL == 4 || PXL == 6 || PXL == 8, "pixels a lane converts");
    constexpr bool SPLIT = true;                                     // (false: merged loads, four rows in flight -- 1-3 % slower, and three of the nine instantiations spill)
    constexpr int NP = PXL / 2, NDW = (3 * PXL + 6) / 4;             // pixel pairs; aligned dwords that hold a lane's 3 PXL bytes at any phase
    constexpr int CB = 32 * PXL + 8;                                 // dwords of one copy of a row (64 PXL samples + slack: windows read whole tap pairs; the odd copy starts one dword in)
    constexpr int ST = 28 + 4 * CHD;                                 // words of a strip's table entry
    // every row twice: as pairs (s0 s1)(s2 s3).. and, one halfword on, as (s1 s2)(s3 s4)..: a window that starts on an
    // odd sample reads the second copy, so that every tap pair of every window is ONE aligned dword (no v_perm)
    __shared__ uint32_t lds[kBlock / 64][3][2 * CB];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = int(threadIdx.x & 63);
    const int wid = int(blockIdx.x) * (kBlock / 64) + wave;
    if (wid > (a.total - 1) / 64) return;                          // (whole waves; no barrier in this kernel)
    const int g0 = wid * 64, seg = int(blockIdx.y);
    const int nlive = min(64, a.total - g0);
    const bool live = lane < nlive;
    const int lc = min(lane, nlive - 1);                           // dead lanes shadow the last live one and store nothing
    const int fA = g0 / a.strips, sA = g0 - fA * a.strips;         // wave-uniform: the wave's first frame, its first strip there
    const int32_t *WC = tab + a.o_wave + 4 * (size_t(sA) * 64 + lane), *WS = tab + a.o_wave + 4 * (size_t(sA) * 64 + lc);
    const int cpx = WC[0], clds = WC[1] >> 8, cfr = fA + (WC[1] & 255);   // what this lane converts: first pixel (< 0: nothing), where it goes in the LDS rows, frame
    const int frame = fA + WS[2], adj = WS[3];                     // this lane's strip: its frame, LDS sample of pixel 0 of that frame's row
    const int strip = sA + lc - WS[2] * a.strips;
    const bool conv = cpx >= 0 && cfr < a.nframes;
    // source bytes are addressed from the wave's first frame (a wave-uniform base) by 32-bit lane offsets
    const uint8_t *sbase;
    uint32_t loff;
    uint8_t *dy, *du, *dv;
    if (a.addr.use_table) {
        sbase = src;
        loff = a.addr.src_tab[0][conv ? cfr : fA];
        dy = dst + a.addr.dst_tab[0][frame]; du = dst + a.addr.dst_tab[1][frame]; dv = dst + a.addr.dst_tab[2][frame];
    } else {
        sbase = src + size_t(fA) * a.addr.src_pitch + a.addr.src_plane[0];
        loff = uint32_t(conv ? cfr - fA : 0) * a.addr.src_pitch;
        uint8_t *dpf = dst + size_t(frame) * a.addr.dst_pitch;
        dy = dpf + a.addr.dst_plane[0]; du = dpf + a.addr.dst_plane[1]; dv = dpf + a.addr.dst_plane[2];
    }
    dy += 8 * size_t(strip); du += 4 * size_t(strip); dv += 4 * size_t(strip);
    // byte offset of this lane's first pixel of row 0 from the dword-aligned base (idle lanes read pixel 0: mapped memory)
    const uint32_t mis = uint32_t(reinterpret_cast<uintptr_t>(sbase) & 3u);
    const uint8_t *sbase4 = sbase - mis;
    const uint32_t off0 = mis + loff + a.src_origin + 3u * uint32_t(conv ? cpx : 0);
    // this lane's windows and taps
    int li[8], ci[4];                                              // first dword of each window, in the copy its parity picks
    uint32_t lt[8][2], ct[4][CHD];
    {
        const int32_t *T = tab + a.o_strip + size_t(ST) * strip;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int h = T[c] + adj;
            li[c] = (h & 1) ? CB + 1 + (h >> 1) : (h >> 1);
            lt[c][0] = uint32_t(T[8 + 2 * c]); lt[c][1] = uint32_t(T[9 + 2 * c]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int h = T[24 + j] + adj;
            ci[j] = (h & 1) ? CB + 1 + (h >> 1) : (h >> 1);
#pragma unroll
            for (int k = 0; k < CHD; ++k) ct[j][k] = uint32_t(T[28 + CHD * j + k]);
        }
    }
    const int32_t *lastl = tab + a.o_lastl, *lastc = tab + a.o_lastc;
    const uint32_t *vcol = reinterpret_cast<const uint32_t *>(tab + a.o_vcol), *vcoc = reinterpret_cast<const uint32_t *>(tab + a.o_vcoc);
    const int ra = seg * a.seg_rows, rb = min(ra + a.seg_rows, a.h);
    int yl = rgbin_lower_bound(lastl, a.oh, ra), yc = rgbin_lower_bound(lastc, a.och, ra);
    const int yle = rgbin_lower_bound(lastl, a.oh, rb), yce = rgbin_lower_bound(lastc, a.och, rb);
    if (yl >= yle && yc >= yce) return;                            // (wave-uniform) no output row ends in this segment
    int rstart = ra, rlast = ra;
    if (yl < yle) { rstart = min(rstart, int(lastl[yl]) - (a.lvt - 1)); rlast = max(rlast, int(lastl[yle - 1])); }
    if (yc < yce) { rstart = min(rstart, int(lastc[yc]) - (a.cvt - 1)); rlast = max(rlast, int(lastc[yce - 1])); }
    rstart = max(rstart, 0);
    // the next output row of each plane class: its last source row and its taps, read (scalar loads) one output row ahead
    int nl = -1, nc = -1;
    uint32_t kl[2] = {0, 0}, kc[4] = {0, 0, 0, 0};
    auto next_luma = [&]() {
        nl = -1;
        if (yl < yle) { nl = lastl[yl]; kl[0] = vcol[2 * yl]; kl[1] = vcol[2 * yl + 1]; }
    };
    auto next_chroma = [&]() {
        nc = -1;
        if (yc < yce) {
            nc = lastc[yc];
#pragma unroll
            for (int k = 0; k < 4; ++k) kc[k] = vcoc[4 * yc + k];
        }
    };
    next_luma(); next_chroma();
    f32x2_t cR[3], cG[3], cB[3], c0[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { cR[p] = a.cf[p][0]; cG[p] = a.cf[p][1]; cB[p] = a.cf[p][2]; c0[p] = a.cf[p][3]; }
    uint32_t ringl[8][2], ringu[4][4], ringv[4][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) { (&ringl[0][0])[i] = 0; (&ringu[0][0])[i] = 0; (&ringv[0][0])[i] = 0; }
    // A lane's 3 PXL bytes of a source row as the aligned dwords that hold them.  Every dword read holds at least one byte
    // of the lane's pixels (the host keeps those inside the frame), so it lies in a mapped page: the last of the NDW is
    // needed only at the larger byte phases, elsewhere the one before it is read again.
    // SPLIT: one load instruction per dword (the offsets are made opaque so that the compiler cannot merge them): what counts
    // for the waits is the NUMBER of loads issued behind a row's own (see D below)
    auto fetch = [&](int r, uint32_t (&raw)[NDW]) {
        const uint32_t o = off0 + uint32_t(r) * uint32_t(a.src_stride), oa = o & ~3u;
#pragma unroll
        for (int k = 0; k < NDW - 1; ++k) {
            uint32_t ok = oa + 4u * k;
            if (SPLIT) asm volatile("" : "+v"(ok));
            raw[k] = *reinterpret_cast<const uint32_t *>(sbase4 + ok);
        }
        const bool need_last = (o & 3u) + 3u * PXL > 4u * (NDW - 1);
        raw[NDW - 1] = *reinterpret_cast<const uint32_t *>(sbase4 + (oa + (need_last ? 4u * (NDW - 1) : 4u * (NDW - 2))));
    };
    // D rows ahead, and the output stores below hidden from the compiler (ttscale_device.h gstore_b64): the wait in
    // front of a row's bytes then leaves the loads of the D - 1 rows behind it -- in fact the stores issued since and the
    // youngest of those loads -- in flight
    constexpr int D = SPLIT ? 2 : 4;                               // rows in flight (a power of two that divides the unrolling)
    uint32_t nx[D][NDW];
#pragma unroll
    for (int k = 0; k < D; ++k) fetch(min(rstart + ((k - rstart) & (D - 1)), rlast), nx[k]);   // (slot k: the first row r >= rstart with r % D == k; static register indices)
    for (int r0 = rstart & ~7; r0 <= rlast; r0 += 8) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int r = r0 + s;
            const bool skip = r < rstart || r > rlast;             // wave-uniform
            // the row's bytes out of their registers, then the loads of the row two on into the same registers: EVERY
            // pass issues them (a skipped one whatever row its registers have to hold when the walk gets there), so that no
            // path joins another with loads in flight on one side only -- the copies such a join needs wait for vmcnt(0)
            uint32_t w[NDW - 1];
            if (!skip) {
                const uint32_t ph = (off0 + uint32_t(r) * uint32_t(a.src_stride)) & 3u;
#pragma unroll
                for (int k = 0; k < NDW - 1; ++k) w[k] = __builtin_amdgcn_alignbyte(nx[s & (D - 1)][k + 1], nx[s & (D - 1)][k], ph);
            }
            fetch(min(max(r + D, rstart + ((r - rstart) & (D - 1))), rlast), nx[s & (D - 1)]);
            if (skip) continue;
            // ---- convert this lane's PXL pixels of the wave's runs
            {
                auto ub = [&](int b) { return float((w[b >> 2] >> (8 * (b & 3))) & 255u); };   // v_cvt_f32_ubyteN
                uint32_t pk[3][NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const f32x2_t R = {ub(6 * i), ub(6 * i + 3)}, G = {ub(6 * i + 1), ub(6 * i + 4)}, B = {ub(6 * i + 2), ub(6 * i + 5)};
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        f32x2_t t = R * cR[p] + (G * cG[p] + (B * cB[p] + c0[p]));      // exact: (2 sum + 1) / 1024
                        t += 8388607.5f;                                              // 2^23 + floor(sum / 512), exactly
                        const float t0 = t.x, t1 = t.y;                               // (__builtin_bit_cast of a vector ELEMENT reads element 0 whichever is named: copy out first)
                        pk[p][i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, t1), __builtin_bit_cast(uint32_t, t0), 0x05040100);
                    }
                }
                if (conv) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        // even copy, then the odd one: dword m = (s[2m + 1], s[2m + 2]); its first and last halfword belong to the neighbours' dwords
                        uint32_t *ea = &lds[wave][p][clds], *ob = &lds[wave][p][CB + 1 + clds];
#pragma unroll
                        for (int k = 0; k < NP; ++k) ea[k] = pk[p][k];
                        reinterpret_cast<uint16_t *>(ob - 1)[1] = uint16_t(pk[p][0]);
#pragma unroll
                        for (int k = 0; k < NP - 1; ++k) ob[k] = __builtin_amdgcn_perm(pk[p][k + 1], pk[p][k], 0x05040302);
                        reinterpret_cast<uint16_t *>(ob + NP - 1)[0] = uint16_t(pk[p][NP - 1] >> 16);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- hScale16To15_c from the wave's rows into the rings (RNG: the context writes yuvj420p -- swscale.c lumRangeToJpeg_c /
            // chrRangeToJpeg_c on the 15-bit samples; the loops are spelt twice so that the usual path pays nothing for it)
            const uint32_t *Y = lds[wave][0], *U = lds[wave][1], *V = lds[wave][2];
            auto hpass = [&](auto RNG) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    int acc = min(dot2(Y[li[c]], lt[c][0], dot2(Y[li[c] + 1], lt[c][1], 0)) >> 13, 32767);
                    if constexpr (decltype(RNG)::value) acc = (__mul24(min(acc, 30189), 19077) - 39057361) >> 14;
                    const uint32_t t = uint32_t(acc);
                    ringl[c][(s & 3) / 2] = (s & 1) ? __builtin_amdgcn_perm(t, ringl[c][(s & 3) / 2], 0x05040100) : __builtin_amdgcn_perm(ringl[c][(s & 3) / 2], t, 0x07060100);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int au = 0, av = 0;
#pragma unroll
                    for (int k = 0; k < CHD; ++k) { au = dot2(U[ci[j] + k], ct[j][k], au); av = dot2(V[ci[j] + k], ct[j][k], av); }
                    au = min(au >> 13, 32767); av = min(av >> 13, 32767);
                    if constexpr (decltype(RNG)::value) { au = (__mul24(min(au, 30775), 4663) - 9289992) >> 12; av = (__mul24(min(av, 30775), 4663) - 9289992) >> 12; }
                    const uint32_t tu = uint32_t(au), tv = uint32_t(av);
                    ringu[j][s / 2] = (s & 1) ? __builtin_amdgcn_perm(tu, ringu[j][s / 2], 0x05040100) : __builtin_amdgcn_perm(ringu[j][s / 2], tu, 0x07060100);
                    ringv[j][s / 2] = (s & 1) ? __builtin_amdgcn_perm(tv, ringv[j][s / 2], 0x05040100) : __builtin_amdgcn_perm(ringv[j][s / 2], tv, 0x07060100);
                }
            };
            if (a.range15) hpass(std::true_type{}); else hpass(std::false_type{});   // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (the next row's conversion overwrites the wave's LDS rows)
            __builtin_amdgcn_wave_barrier();
            // ---- output rows this source row completes
            while (nl == r) {                                      // wave-uniform
                int v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = dot2(ringl[c][1], kl[1], dot2(ringl[c][0], kl[0], 64 << 12));
                const u32x2_t o = {pack4<19>(v[0], v[1], v[2], v[3]), pack4<19>(v[4], v[5], v[6], v[7])};
                if (live) gstore_b64(dy + size_t(yl) * a.dst_stride[0], o);
                ++yl;
                next_luma();
            }
            while (nc == r) {
                int au[4], av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int tu = 64 << 12, tv = 64 << 12;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { tu = dot2(ringu[j][k], kc[k], tu); tv = dot2(ringv[j][k], kc[k], tv); }
                    au[j] = tu; av[j] = tv;
                }
                const uint32_t ou = pack4<19>(au[0], au[1], au[2], au[3]), ov = pack4<19>(av[0], av[1], av[2], av[3]);
                if (live) {
                    gstore_b32(du + size_t(yc) * a.dst_stride[1], ou);
                    gstore_b32(dv + size_t(yc) * a.dst_stride[1], ov);
                }
                ++yc;
                next_chroma();
            }
        }
    }
}

template <class T>
bool upload_vec(T *&dptr, const std::vector<T> &host)
{
    if (hipMalloc(reinterpret_cast<void **>(&dptr), std::max<size_t>(host.size(), 1) * sizeof(T)) != hipSuccess) return false;
    return hipMemcpy(dptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

template <class... P, class... A>
int launch_rgb(const char *what, void (*fn)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A &&...args)
{
    std::tuple<P...> held(std::forward<A>(args)...);
    void *ptrs[sizeof...(P)];
    size_t i = 0;
    std::apply([&](auto &...v) { ((ptrs[i++] = const_cast<void *>(static_cast<const void *>(&v))), ...); }, held);
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(fn), grid, block, ptrs, lds, st);
    if (e == hipSuccess) return TTS_OK;
    tts::set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    return TTS_ERR_PROCESSING_FAILED;
}

}  // namespace

namespace tts {

struct RgbPlan {
    tts_geom g{};
    int n_cus = 256;
    int bpp = 3, swap = 0;
    enum Mode { kUnscaled, kGeneric } mode = kGeneric;
    // the fused strip walk (every axis <= 4 taps, no pad, even crop): tables in HBM, one set per segment length
    bool strip_ok = false;
    bool unscaled8 = false;               // kUnscaled: rgb_unscaled8_kernel where a launch's addresses allow
    bool convert4 = false;                // in-place conversion: four pixels a thread even where eight fit (TTS_RGB_CONVERT4, A/B)
    bool strip_lds = false;            // ... with the source rows staged through LDS (the runs of any 64 strips fit a 1 KiB slot per plane)
    uint32_t *d_tab = nullptr;
    RgbStripArgs strip{};              // table offsets and constants; per-launch fields are filled in by rgb_launch
    int strip_occ[4] = {1, 1, 1, 1};   // resident workgroups per CU of the plan's rgb_strip_kernel instantiations [lds * 2 + xp], asked at plan creation on the plan's device
    bool no_xp = false;                // TTS_NO_XP at plan creation (A/B knob: no whole-sector store exchange)
    struct Seg { int seg_rows, nseg; uint32_t seginfo; };
    std::vector<Seg> segs;
    bool two_stage = false;            // kGeneric: yuv420p scale, then the unscaled converter over the finished canvas
    int cpx = 0, cpy = 0, cpw = 0, cph = 0, cox = 0, coy = 0;
    AxisFilter lh, lv, ch, cv;
    int32_t *d_pos[4] = {nullptr, nullptr, nullptr, nullptr};      // lh, lv, ch, cv
    int16_t *d_taps[4] = {nullptr, nullptr, nullptr, nullptr};
    int px = 0, py = 0, pw = 0, ph = 0, ox = 0, oy = 0;
    int border[3] = {0, 0, 0};
    RgbCoef coef{};
    bool full = false;                 // the conversion pass reads a yuvj420p canvas: range step per sample (rng), full-range tables (coef)
    int rng[2][3] = {{1, 0, 0}, {1, 0, 0}};
    bool packed12 = false;             // some output rows take swscale's 1- / 2-line writers (vertical filters of 1-2 taps): the general kernel only
    RgbWalkPlan *walk = nullptr;       // 5-8 taps an axis: the fused walk of ttscale_rgbwalk.hip (packed batches)
    char name[96] = "";
};

typedef void (*rgb_strip_fn)(const uint8_t *, uint8_t *, const uint32_t *, const RgbStripArgs);
static const rgb_strip_fn rgb_strip_fns[8] = {rgb_strip_kernel<3, false, false>, rgb_strip_kernel<3, false, true>, rgb_strip_kernel<3, true, false>, rgb_strip_kernel<3, true, true>,
                                              rgb_strip_kernel<4, false, false>, rgb_strip_kernel<4, false, true>, rgb_strip_kernel<4, true, false>, rgb_strip_kernel<4, true, true>};

// Tables of the fused strip walk, if this geometry can take it.
static bool build_rgb_strip(RgbPlan *p)
{
    const tts_geom &g = p->g;
    const AxisFilter &lh = p->lh, &lv = p->lv, &ch = p->ch, &cv = p->cv;
    if (std::getenv("TTS_FORCE_GENERIC") && std::atoi(std::getenv("TTS_FORCE_GENERIC"))) return false;
    if (std::getenv("TTS_NO_RGB_STRIP") && std::atoi(std::getenv("TTS_NO_RGB_STRIP"))) return false;     // (A/B: the 5-8 tap walk takes the enlargements too)
    if (g.rgb_stage != TTS_RGB_AT_SCALE || g.do_pad || p->two_stage) return false;
    if (lh.size > 4 || lv.size > 4 || ch.size > 4 || cv.size > 4) return false;
    if ((g.crop_x & 1) || (g.crop_w & 1) || g.crop_w < 8) return false;           // pixel pairs must not straddle a strip
    if (lh.src < 8 || ch.src < 8) return false;
    for (size_t i = 1; i < lv.pos.size(); ++i) if (lv.pos[i] < lv.pos[i - 1]) return false;
    for (size_t i = 1; i < cv.pos.size(); ++i) if (cv.pos[i] < cv.pos[i - 1]) return false;
    const int pw = g.crop_w, ph = g.crop_h, strips = (pw + 7) / 8;
    std::vector<uint32_t> tab;
    RgbStripArgs &a = p->strip;
    auto pack16 = [](int16_t lo, int16_t hi) { return uint32_t(uint16_t(lo)) | (uint32_t(uint16_t(hi)) << 16); };
    // one 8-byte window per 4 columns; each column picks its taps with a byte selector (taps <= 4 and 4
    // columns within 8 bytes: every enlargement)
    auto group = [&](const AxisFilter &f, int first_col, uint32_t &hoff_out, uint32_t *sel, uint32_t *coef) {
        const int glo = std::min(f.pos[first_col], f.src - 8);
        for (int j = 0; j < 4; ++j) {
            const int col = std::min(first_col + j, f.dst - 1);
            const int d = f.pos[col] - glo;
            if (d < 0 || d + f.size > 8) return false;
            int16_t t4[4] = {0, 0, 0, 0};
            for (int k = 0; k < f.size; ++k) t4[k] = f.taps[size_t(col) * f.size + k];
            sel[2 * j] = 0x0c000c00u | uint32_t(d) | (uint32_t(d + 1) << 16);
            sel[2 * j + 1] = 0x0c000c00u | uint32_t(d + 2) | (uint32_t(d + 3) << 16);
            coef[2 * j] = pack16(t4[0], t4[1]); coef[2 * j + 1] = pack16(t4[2], t4[3]);
        }
        hoff_out = uint32_t(glo);
        return true;
    };
    a.hoffl = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips) * 2);
    a.sell = uint32_t(tab.size());  tab.resize(tab.size() + size_t(strips) * 16);
    a.coefl = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips) * 16);
    a.hoffc = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips));
    a.selc = uint32_t(tab.size());  tab.resize(tab.size() + size_t(strips) * 8);
    a.coefc = uint32_t(tab.size()); tab.resize(tab.size() + size_t(strips) * 8);
    for (int s0 = 0; s0 < strips; ++s0) {
        const int vx = g.crop_x + std::min(s0 * 8, pw - 8);                      // even: crop_x and pw are
        for (int q = 0; q < 2; ++q)
            if (!group(lh, vx + 4 * q, tab[a.hoffl + s0 * 2 + q], &tab[a.sell + (size_t(s0) * 8 + 4 * q) * 2], &tab[a.coefl + (size_t(s0) * 8 + 4 * q) * 2])) return false;
        if (!group(ch, vx >> 1, tab[a.hoffc + s0], &tab[a.selc + size_t(s0) * 8], &tab[a.coefc + size_t(s0) * 8])) return false;
    }
    // per output row: source rows that enter the rings first; the 4 ring rows end at the filter's last tap,
    // so shorter filters sit right-aligned behind zero taps
    a.vrow = uint32_t(tab.size()); tab.resize(tab.size() + size_t(ph) * 6);
    for (int y = 0; y < ph; ++y) {
        const int vy = g.crop_y + y;
        uint32_t *r = &tab[a.vrow + size_t(y) * 6];
        r[0] = y ? uint32_t(lv.pos[vy] - lv.pos[vy - 1]) : 0u;
        r[1] = y ? uint32_t(cv.pos[vy] - cv.pos[vy - 1]) : 0u;
        int16_t tl[4] = {0, 0, 0, 0}, tc[4] = {0, 0, 0, 0};
        for (int k = 0; k < lv.size; ++k) tl[4 - lv.size + k] = lv.taps[size_t(vy) * lv.size + k];
        for (int k = 0; k < cv.size; ++k) tc[4 - cv.size + k] = cv.taps[size_t(vy) * cv.size + k];
        r[2] = pack16(tl[0], tl[1]); r[3] = pack16(tl[2], tl[3]);
        r[4] = pack16(tc[0], tc[1]); r[5] = pack16(tc[2], tc[3]);
    }
    for (int sr : {16, 24, 32, 48, 64, 96, 128, 192, 256}) {
        RgbPlan::Seg sg;
        sg.seg_rows = sr; sg.nseg = (ph + sr - 1) / sr; sg.seginfo = uint32_t(tab.size());
        for (int k = 0; k < sg.nseg; ++k) {
            const int vy = g.crop_y + k * sr;
            tab.push_back(uint32_t(lv.pos[vy] + lv.size - 4));
            tab.push_back(uint32_t(cv.pos[vy] + cv.size - 4));
        }
        p->segs.push_back(sg);
        if (sg.nseg == 1) break;
    }
    // LDS staging: any 64 consecutive strips (they wrap over frames: one piece per row touched) must need at
    // most kRgbMaxPieces runs and 64 sixteen-byte chunks per plane, whatever the runs' alignment
    auto fits = [&](uint32_t hoff, int windows) {
        if (std::getenv("TTS_NO_LDS") && std::atoi(std::getenv("TTS_NO_LDS"))) return false;
        if (strips < 2) return false;
        for (int s0 = 0; s0 < strips; ++s0) {
            uint32_t chunks = 0;
            int pieces = 0, s1 = s0, left = 64;
            while (left > 0) {
                const int take = std::min(left, strips - s1);
                const uint32_t b = tab[hoff + size_t(s1) * windows], e = tab[hoff + size_t(s1 + take - 1) * windows + windows - 1] + 8u;
                chunks += (e - b + 30u) >> 4;
                ++pieces; left -= take; s1 = 0;
            }
            if (pieces > kRgbMaxPieces || chunks > 64u) return false;
        }
        return true;
    };
    p->strip_lds = fits(a.hoffl, 2) && fits(a.hoffc, 1);
    a.strips = strips; a.ph = ph; a.pw = pw;
    a.src_h_l = lv.src; a.src_h_c = cv.src;
    a.swap = p->swap; a.coef = p->coef;
    return upload_vec(p->d_tab, tab);
}

int rgb_plan_create(const tts_geom &g, int n_cus, RgbPlan **out)
{
    *out = nullptr;
    if (g.out_fmt < TTS_FMT_RGB24 || g.out_fmt > TTS_FMT_BGRA || (g.rgb_stage != TTS_RGB_AT_SCALE && g.rgb_stage != TTS_RGB_AT_END)) {
        set_last_error("not a packed-RGB geometry");
        return TTS_ERR_INVALID_INPUT;
    }
    RgbPlan *p = new (std::nothrow) RgbPlan();
    if (!p) return TTS_ERR_INITIALIZATION_FAILED;
    p->g = g; p->n_cus = n_cus;
    p->bpp = g.out_fmt >= TTS_FMT_RGBA ? 4 : 3;
    p->swap = g.out_fmt == TTS_FMT_BGR24 || g.out_fmt == TTS_FMT_BGRA;
    const bool full = g.out_range == TTS_RANGE_TO_FULL;            // `format=yuvj420p,format=<rgb>`: range pass, then a conversion pass over a full-range source (AT_END only: check_geom)
    p->coef = rgb_coef(full);
    p->full = full;
    if (full)
        for (int k = 0; k < 2; ++k)
            if (!range_affine(g.out_range, k == 1, p->rng[k][0], p->rng[k][1], p->rng[k][2])) { set_last_error("range step has no closed form"); delete p; return TTS_ERR_UNSUPPORTED; }
    p->convert4 = std::getenv("TTS_RGB_CONVERT4") && std::atoi(std::getenv("TTS_RGB_CONVERT4"));
    p->px = g.do_pad ? g.pad_x : 0; p->py = g.do_pad ? g.pad_y : 0;
    p->pw = g.crop_w; p->ph = g.crop_h; p->ox = g.crop_x; p->oy = g.crop_y;

    // the swscale context that writes the RGB: source size, destination size
    int ctx_sw, ctx_sh, ctx_dw, ctx_dh;
    if (g.rgb_stage == TTS_RGB_AT_SCALE) {
        ctx_sw = g.pre_w; ctx_sh = g.pre_h; ctx_dw = g.sc_w; ctx_dh = g.sc_h;
        p->border[0] = (g.fill_rgb >> 16) & 255; p->border[1] = (g.fill_rgb >> 8) & 255; p->border[2] = g.fill_rgb & 255;
    } else {
        // conversion pass at the end of a chain without scale: its source is the finished yuv420p canvas
        // (crop = an offset into the source; pad = fill colour around it, on the chroma grid; an odd picture
        // width loses its last column to the fill, see vf_pad in the oracle)
        ctx_sw = ctx_dw = g.out_w; ctx_sh = ctx_dh = g.out_h;
        if (g.do_pad) p->pw = g.crop_w & ~1;
        p->cpx = p->px >> 1; p->cpy = p->py >> 1; p->cpw = (p->pw + 1) >> 1; p->cph = (g.crop_h + 1) >> 1;
        {   // the border as the conversion pass sees it: the yuv fill, through the range pass if there is one
            auto rg = [&](int k, int v) { if (!full) return v; const int t = (v * p->rng[k][0] + p->rng[k][1]) >> p->rng[k][2]; return t < 0 ? 0 : t > 255 ? 255 : t; };
            rgb_pixel(p->coef, rg(0, g.fill_y), rg(1, g.fill_u), rg(1, g.fill_v), p->border[0], p->border[1], p->border[2]);
        }
    }
    int rc = TTS_OK;
    auto cs = [](int v) { return (v + 1) >> 1; };
    if (g.rgb_stage == TTS_RGB_AT_END && g.do_scale) {
        // scale (+ crop, pad) on yuv420p first, then the conversion pass over the finished canvas.  The
        // unscaled converter is a function of one pixel's Y and of chroma sample (x >> 1, y >> 1), so both
        // passes fit one kernel; a canvas of odd height would take swscale's generic path instead, which
        // interpolates the SCALED chroma rows: that needs the intermediate frame in memory.
        if ((g.out_w | g.out_h) & 1) {
            set_last_error("yuv420p scale followed by an RGB conversion pass over a canvas of odd width or height is not supported");
            rc = TTS_ERR_UNSUPPORTED;
        }
        if (!rc && (!build_axis_filter(g.pre_w, g.sc_w, 1 << 14, g.alg, p->lh) || !build_axis_filter(g.pre_h, g.sc_h, 1 << 12, g.alg, p->lv) ||
                    !build_axis_filter(cs(g.pre_w), cs(g.sc_w), 1 << 14, g.alg, p->ch) || !build_axis_filter(cs(g.pre_h), cs(g.sc_h), 1 << 12, g.alg, p->cv))) {
            set_last_error("filter tables"); rc = TTS_ERR_INVALID_INPUT;
        }
        p->two_stage = true;
        p->cox = g.crop_x >> 1; p->coy = g.crop_y >> 1;
        const AxisFilter *f[4] = {&p->lh, &p->lv, &p->ch, &p->cv};
        for (int i = 0; i < 4 && !rc; ++i)
            if (!upload_vec(p->d_pos[i], f[i]->pos) || !upload_vec(p->d_taps[i], f[i]->taps)) { set_last_error("table upload failed (hipMalloc / hipMemcpy)"); rc = TTS_ERR_INITIALIZATION_FAILED; }
        std::snprintf(p->name, sizeof p->name, "rgb_generic<%d,%d,%d,%d>+unscaled", p->lh.size, p->lv.size, p->ch.size, p->cv.size);
    } else if (ctx_sw == ctx_dw && ctx_sh == ctx_dh && !(ctx_dh & 1)) {
        p->mode = RgbPlan::kUnscaled;
        if (ctx_dw & 1) { set_last_error("swscale's unscaled yuv420p -> RGB converter never writes the last column of an odd width: not supported"); rc = TTS_ERR_UNSUPPORTED; }
        // the streaming form where the geometry fits it (the addresses are checked per launch): no border, a
        // picture that is a multiple of 8 wide, its source rows on the dword grid
        p->unscaled8 = !full && !p->convert4 && (g.rgb_stage != TTS_RGB_AT_END || (p->cpx == 0 && p->cpy == 0 && 2 * p->cpw == p->pw && 2 * p->cph == p->ph)) && p->px == 0 && p->py == 0 && p->pw == g.out_w && p->ph == g.out_h &&
                       p->pw % 8 == 0 && p->ph % 2 == 0 && p->ox % 8 == 0 && p->oy % 2 == 0 && g.pre_x % 8 == 0;
        std::snprintf(p->name, sizeof p->name, p->unscaled8 ? "rgb_unscaled8" : "rgb_unscaled");
    } else {
        if (g.rgb_stage == TTS_RGB_AT_END && (g.do_pad || g.crop_w != g.sc_w || g.crop_h != g.sc_h)) {
            // (cannot happen: crop and pad make the height even, which selects the unscaled converter)
            set_last_error("conversion pass over a cropped or padded canvas of odd height"); rc = TTS_ERR_UNSUPPORTED;
        }
        const int csw = (ctx_sw + 1) >> 1, csh = (ctx_sh + 1) >> 1, cdw = (ctx_dw + 1) >> 1;
        if (!rc && (!build_axis_filter(ctx_sw, ctx_dw, 1 << 14, g.alg, p->lh) || !build_axis_filter(ctx_sh, ctx_dh, 1 << 12, g.alg, p->lv) ||
                    !build_axis_filter(csw, cdw, 1 << 14, g.alg, p->ch) || !build_axis_filter(csh, ctx_dh, 1 << 12, g.alg, p->cv))) {
            set_last_error("filter tables"); rc = TTS_ERR_INVALID_INPUT;
        }
        // vscale.c packed_vscale(): rows whose vertical filters have 1-2 taps on both luma and chroma go to yuv2packed1 /
        // yuv2packed2, whose arithmetic differs (no 19-bit sum): the general kernel decides per row, the fused ones stay out
        p->packed12 = p->lv.size <= 2 && p->cv.size <= 2;
        const AxisFilter *f[4] = {&p->lh, &p->lv, &p->ch, &p->cv};
        for (int i = 0; i < 4 && !rc; ++i)
            if (!upload_vec(p->d_pos[i], f[i]->pos) || !upload_vec(p->d_taps[i], f[i]->taps)) { set_last_error("table upload failed (hipMalloc / hipMemcpy)"); rc = TTS_ERR_INITIALIZATION_FAILED; }
        std::snprintf(p->name, sizeof p->name, "rgb_generic<%d,%d,%d,%d>", p->lh.size, p->lv.size, p->ch.size, p->cv.size);
        if (!rc && !p->packed12 && !full && (p->strip_ok = build_rgb_strip(p))) {
            std::snprintf(p->name, sizeof p->name, "rgb_strip<%d,%s>", p->bpp, p->strip_lds ? "lds" : "regs");
            // per-plan, per-device launch facts (they used to sit in function statics of rgb_launch, shared by every thread and GPU)
            for (int v = 0; v < 4; ++v)
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p->strip_occ[v], rgb_strip_fns[(p->bpp == 4) * 4 + v], kBlock, 0) != hipSuccess || p->strip_occ[v] < 1) p->strip_occ[v] = 1;
            p->no_xp = std::getenv("TTS_NO_XP") && std::atoi(std::getenv("TTS_NO_XP"));
        }
        if (!rc && !p->strip_ok && !p->packed12 && !full) p->walk = rgb_walk_create(g, p->lh, p->lv, p->ch, p->cv, p->bpp, p->swap, p->coef, n_cus);
    }
    if (rc) { rgb_plan_destroy(p); return rc; }
    *out = p;
    return TTS_OK;
}

void rgb_plan_destroy(RgbPlan *p)
{
    if (!p) return;
    for (int i = 0; i < 4; ++i) { (void)hipFree(p->d_pos[i]); (void)hipFree(p->d_taps[i]); }
    (void)hipFree(p->d_tab);
    rgb_walk_destroy(p->walk);
    delete p;
}

const char *rgb_plan_kernel_name(const RgbPlan *p) { return !p ? "" : p->walk ? rgb_walk_name(p->walk) : p->name; }
bool rgb_plan_walks(const RgbPlan *p, int n, uint64_t src_span, uint32_t src_pitch, uint32_t dst_pitch)
{
    FrameAddr addr{};
    addr.src_pitch = src_pitch; addr.dst_pitch = dst_pitch;
    return p && p->walk && rgb_walk_takes(p->walk, n, src_span, addr);
}
bool rgb_plan_is_strip(const RgbPlan *p) { return p && p->strip_ok; }
bool rgb_plan_general_only(const RgbPlan *p) { return p && (p->packed12 || p->full); }

int rgb_launch(const RgbPlan *p, const uint8_t *src, uint8_t *dst, int n, const int src_stride[2], int dst_stride,
               uint64_t src_span, uint64_t dst_span, const FrameAddr &addr, hipStream_t st)
{
    if (p->strip_ok) {
        // lanes are the flattened (frame, strip) index and address from their wave's first frame with 32-bit
        // offsets: the frames 64 lanes can touch must lie within kMaxSpan of it (else: the generic kernel)
        const size_t wave_frames = size_t(63 / p->strip.strips + 2);
        const bool fits = addr.use_table || (wave_frames * addr.src_pitch <= kMaxSpan && wave_frames * addr.dst_pitch <= kMaxSpan);
        const int64_t items = int64_t(n) * p->strip.strips;
        if (fits && items < (int64_t(1) << 30)) {
            // (per-frame plane pointers keep the register variant: a staged run may round into bytes around a
            // plane, which only a packed batch is known to own)
            const bool lds = p->strip_lds && !addr.use_table && src_span >= 4096u;
            // whole-sector stores: rgba always (a chunk never leaves its lane); rgb24 where a wave's strips wrap
            // over frames at most once and no last strip is slid over its neighbour (one held-back chunk at most)
            const bool xp = !p->no_xp && (p->bpp == 4 || (p->strip.pw % 8 == 0 && p->strip.strips >= 64));
            rgb_strip_fn fn = rgb_strip_fns[(p->bpp == 4) * 4 + lds * 2 + xp];
            const int occ = p->strip_occ[lds * 2 + xp];
            // the longest segment <= 96 rows that still leaves 3 workgroups per resident slot, else the shortest
            const double slots = double(p->n_cus) * occ;
            const int64_t sblocks = (items + kBlock - 1) / kBlock;
            size_t best = 0;
            for (size_t i = 0; i < p->segs.size() && p->segs[i].seg_rows <= 96; ++i)
                if (i == 0 || double(sblocks) * p->segs[i].nseg >= 3.0 * slots) best = i;
            RgbStripArgs a = p->strip;
            a.seginfo = p->segs[best].seginfo; a.nseg = p->segs[best].nseg; a.seg_rows = p->segs[best].seg_rows;
            a.n_items = int(items);
            a.src_stride_y = src_stride[0]; a.src_stride_c = src_stride[1]; a.dst_stride = dst_stride;
            a.src_span = src_span; a.dst_span = dst_span; a.addr = addr;
            {   // the source planes, seen from the pre-cropped origin the frame offsets point at
                const tts_geom &g = p->g;
                for (int k = 0; k < 2; ++k) {
                    const int pw_ = (g.in_w + k) >> k, ph_ = (g.in_h + k) >> k, px_ = g.pre_x >> k, py_ = g.pre_y >> k;
                    a.end_rel[k] = uint32_t(ph_ - py_ - 1) * uint32_t(src_stride[k]) + uint32_t(pw_ - px_);
                }
            }
#ifdef TTS_EXPERIMENTS
            // timing experiments (outputs are wrong): a zero-record descriptor drops that side's traffic, the instruction stream stays
            if (std::getenv("TTS_EXP_NOSRC")) a.src_span = 0;
            if (std::getenv("TTS_EXP_NODST")) a.dst_span = 0;
#endif
            return launch_rgb("rgb_strip_kernel", fn, dim3(uint32_t(sblocks) * uint32_t(a.nseg)), dim3(kBlock), 0, st, src, dst, p->d_tab, a);
        }
    }
    if (p->walk && rgb_walk_takes(p->walk, n, src_span, addr))
        return rgb_walk_launch(p->walk, src, dst, n, src_stride, dst_stride, src_span, dst_span, addr, st);
    RgbFrameArgs f{};
    f.src = src; f.dst = dst;
    f.canvas_w = p->g.out_w; f.canvas_h = p->g.out_h;
    f.px = p->px; f.py = p->py; f.pw = p->pw; f.ph = p->ph; f.ox = p->ox; f.oy = p->oy;
    f.src_stride_y = src_stride[0]; f.src_stride_c = src_stride[1]; f.dst_stride = dst_stride;
    f.bpp = p->bpp; f.swap = p->swap;
    for (int i = 0; i < 3; ++i) f.border[i] = p->border[i];
    f.coef = p->coef; f.addr = addr;
    f.rng_on = p->full ? 1 : 0;
    for (int k = 0; k < 2; ++k) for (int i = 0; i < 3; ++i) f.rng[k][i] = p->rng[k][i];
    f.yuv_pad = p->g.rgb_stage == TTS_RGB_AT_END;
    f.cpx = p->cpx; f.cpy = p->cpy; f.cpw = p->cpw; f.cph = p->cph;
    f.fill[0] = p->g.fill_y; f.fill[1] = p->g.fill_u; f.fill[2] = p->g.fill_v;
    int rc = TTS_OK;
    for (int f0 = 0; f0 < n && !rc; f0 += 65535) {          // grid.z
        const int nf = std::min(65535, n - f0);
        f.frame0 = f0;
        if (p->mode == RgbPlan::kUnscaled) {
            // the streaming form where the geometry and the addresses allow (see the kernel)
            bool fast = p->unscaled8 && reinterpret_cast<uintptr_t>(src) % 4 == 0 && src_stride[0] % 4 == 0 && src_stride[1] % 4 == 0 &&
                        reinterpret_cast<uintptr_t>(dst) % 4 == 0 && dst_stride % 4 == 0;
            if (fast && addr.use_table) {                            // per-frame pointers: every frame's planes
                for (int i = f0; i < f0 + nf && fast; ++i)
                    fast = addr.src_tab[0][i] % 4 == 0 && addr.src_tab[1][i] % 4 == 0 && addr.src_tab[2][i] % 4 == 0 && addr.dst_tab[0][i] % 4 == 0;
            } else if (fast)
                fast = addr.src_pitch % 4 == 0 && addr.src_plane[0] % 4 == 0 && addr.src_plane[1] % 4 == 0 && addr.src_plane[2] % 4 == 0 && addr.dst_pitch % 4 == 0;
            if (fast) {
                const int waves = ((f.pw / 8 + 63) / 64) * (f.ph / 2);
                rc = launch_rgb("rgb_unscaled8_kernel", p->bpp == 3 ? rgb_unscaled8_kernel<3> : rgb_unscaled8_kernel<4>,
                                dim3((waves + kBlock / 64 - 1) / (kBlock / 64), 1, nf), dim3(kBlock), 0, st, f);
            } else
            rc = launch_rgb("rgb_unscaled_kernel", rgb_unscaled_kernel, dim3((f.canvas_w + 255) / 256, (f.canvas_h + 3) / 4, nf), dim3(kBlock), 0, st, f);
        } else {
            RgbGenericArgs a{};
            a.f = f;
            a.lhpos = p->d_pos[0]; a.lvpos = p->d_pos[1]; a.chpos = p->d_pos[2]; a.cvpos = p->d_pos[3];
            a.lhtaps = p->d_taps[0]; a.lvtaps = p->d_taps[1]; a.chtaps = p->d_taps[2]; a.cvtaps = p->d_taps[3];
            a.lht = p->lh.size; a.lvt = p->lv.size; a.cht = p->ch.size; a.cvt = p->cv.size;
            a.two_stage = p->two_stage;
            a.cox = p->cox; a.coy = p->coy;
            rc = launch_rgb("rgb_generic_kernel", rgb_generic_kernel, dim3((f.canvas_w + 63) / 64, (f.canvas_h + 3) / 4, nf), dim3(kBlock), 0, st, a);
        }
    }
    return rc;
}

}  // namespace tts

namespace tts {

bool rgb_scratch_layout(const tts_geom &g, int dst_stride, RgbScratchLayout &out)
{
    if (g.out_fmt == TTS_FMT_YUV420P || g.rgb_stage != TTS_RGB_AT_SCALE || !g.do_scale || ((g.crop_x | g.crop_y) & 1)) return false;
    const int pw = g.crop_w, cw = (pw + 1) >> 1, bpp = g.out_fmt >= TTS_FMT_RGBA ? 4 : 3;
    // (an odd width at 3 bytes a pixel is one byte short: it may borrow from the border pixels to its right in the
    // same canvas row, which the row's workgroup writes after its barrier like everything else)
    const int right = g.do_pad ? g.out_w - g.pad_x - pw : 0;
    if (2 * pw + 2 * cw > (pw + right) * bpp || pw > kScratchThreads * kScratchMaxPerThread) return false;
    const uint32_t origin = uint32_t(g.do_pad ? g.pad_y : 0) * uint32_t(dst_stride) + uint32_t(g.do_pad ? g.pad_x : 0) * uint32_t(bpp);
    out.off[0] = origin; out.off[1] = origin + 2u * uint32_t(pw); out.off[2] = out.off[1] + uint32_t(cw);
    out.stride[0] = out.stride[1] = dst_stride;                    // every plane has one row per canvas row
    return true;
}

// frames [0, n): addr.dst_plane[0] / dst_tab[0] = a frame's canvas origin
int rgb_scratch_convert(const RgbPlan *p, uint8_t *dst, int dst_stride, int n, const FrameAddr &addr, hipStream_t st)
{
    RgbScratchArgs a{};
    a.dst = dst;
    a.pw = p->pw; a.ph = p->ph; a.cw = (p->pw + 1) >> 1;
    a.canvas_w = p->g.out_w; a.canvas_h = p->g.out_h; a.px = p->px; a.py = p->py; a.dst_stride = dst_stride;
    a.bpp = p->bpp; a.swap = p->swap ? 1 : 0;
    for (int i = 0; i < 3; ++i) a.border[i] = p->border[i];
    a.coef = p->coef;
    a.addr = addr;
    // four pixels a thread through dword accesses where every picture row starts on a dword and holds whole quads
    bool quad = p->pw % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0 && dst_stride % 4 == 0 && (p->px * p->bpp) % 4 == 0;
    if (quad && addr.use_table) {                                  // per-frame pointers (at most kMaxTableFrames a launch)
        for (int i = 0; i < n && quad; ++i) quad = addr.dst_tab[0][i] % 4 == 0;
    } else if (quad)
        quad = addr.dst_pitch % 4 == 0 && addr.dst_plane[0] % 4 == 0;
    const int px_unit = !quad ? 1 : p->pw % 8 == 0 && !p->convert4 ? 8 : 4;
    const int units = p->pw / px_unit;                            // per row; about two units a thread (six pixels without dword accesses) on narrow pictures
    a.tpr = units <= 64 * (quad ? 2 : 6) ? 64 : units <= 128 * (quad ? 2 : 6) ? 128 : 256;
    const int rows_per_wg = kScratchThreads / a.tpr;
    int rc = TTS_OK;
    for (int f0 = 0; f0 < n && !rc; f0 += 65535) {               // grid.y = frames
        a.frame0 = f0;
        rc = launch_rgb("rgb_scratch_convert_kernel", px_unit == 8 ? rgb_scratch_convert4_kernel<8> : px_unit == 4 ? rgb_scratch_convert4_kernel<4> : rgb_scratch_convert_kernel,
                        dim3(unsigned((p->g.out_h + rows_per_wg - 1) / rows_per_wg), unsigned(std::min(65535, n - f0))), dim3(kScratchThreads), 0, st, a);
    }
    return rc;
}

}  // namespace tts

namespace tts {

struct RgbInPlan {
    tts_geom g{};
    bool half = true;                  // chroma read as pixel-pair sums (no horizontal enlargement), else one sample per pixel
    int pb = 3;                        // bytes a source pixel (rgba, bgra: 4 -- the general kernels only)
    bool blue_first = false;           // bgr24, bgra: the coefficient table is swapped, the kernels read bytes 0, 1, 2 as ever
    AxisFilter lh, lv, ch, cv;
    int32_t *d_pos[4] = {nullptr, nullptr, nullptr, nullptr};
    int16_t *d_taps[4] = {nullptr, nullptr, nullptr, nullptr};
    // rgbin_tile_kernel: output rows per tile and LDS extents per plane class; tiled == false: the generic kernel
    bool tiled = false;
    int th[2] = {0, 0}, sx_max[2] = {0, 0}, sy_max[2] = {0, 0}, tiles_y[2] = {0, 0};
    size_t lds_bytes = 0;
    // rgbin_walk_kernel (the conversion alone): per chroma output row its last source row, then its taps by ring slot
    bool walk = false;
    int32_t *d_walk = nullptr;
    // rgbin_strip_kernel (horizontal enlargements): per-strip windows and taps, per-wave-start conversion / placement
    // table, vertical tables
    bool strip = false;
    int32_t *d_strip = nullptr;
    uint32_t so_strip = 0, so_wave = 0, so_lastl = 0, so_vcol = 0, so_lastc = 0, so_vcoc = 0;
    int strips = 0, pxl = 8, chd = 2;
    int seg_knob = 0;                  // TTS_RGBIN_SEG at plan creation: row segments a launch (0: the launch decides)
    RgbInShrinkPlan *shrink = nullptr; // shrinks with at most 8 taps across: the source-row walk of ttscale_rgbin_shrink.hip
    bool yv12 = false;                 // bgr24 at the same size: swscale's special converter (bgr24_yv12_kernel), none of the above
    bool yv12_rows = false;            // ... with nothing behind it: bgr24_yv12_rows_kernel
    bool range15 = false;              // into yuvj420p: the context's own range step (RgbInArgs.range15), general kernel only
    bool canvas = false;               // a pad on RGB pixels in front of the conversion (TTS_RGB_AT_END): the general kernels read the canvas through vs_*
    char name[96] = "";
};

// Picture rectangle of plane class k inside its canvas, and its origin in the scaled plane (as rgbin_launch fills them in)
struct RgbInRect { int canvas_w, canvas_h, px, py, pw, ph, ox, oy; };
static RgbInRect rgbin_rect(const tts_geom &g, int k)
{
    const int content_w = g.do_pad ? (g.crop_w & ~1) : g.crop_w;     // vf_pad: an odd picture loses its last column to the fill
    return {(g.out_w + k) >> k, (g.out_h + k) >> k, g.do_pad ? g.pad_x >> k : 0, g.do_pad ? g.pad_y >> k : 0,
            (content_w + k) >> k, (g.crop_h + k) >> k, g.crop_x >> k, g.crop_y >> k};
}

// Tile height of rgbin_tile_kernel for one plane class: the tallest of the heights on offer whose two LDS arrays
// (converted source samples, horizontal results) fit `budget` bytes for the plane's worst tile.  false: none does.
static int rgbin_raw_w(int sx_max, int nb) { return (nb * sx_max + 6) & ~3; }          // (up to 3 bytes of misalignment in front, rounded up to dwords)
static size_t rgbin_tile_lds(int sy, int sx, int th, int vt, int nb)
{
    return (((size_t(sy) * size_t(sx + 64) + size_t(th) * size_t(vt + 1)) * 2 + 3) & ~size_t(3)) + size_t(sy) * size_t(rgbin_raw_w(sx, nb));
}
static bool rgbin_tile_shape(const RgbInRect &r, const AxisFilter &h, const AxisFilter &v, int nb, size_t budget, int &th, int &sx_max, int &sy_max)
{
    auto monotone = [](const std::vector<int32_t> &p) { for (size_t i = 1; i < p.size(); ++i) if (p[i] < p[i - 1]) return false; return true; };
    if (!monotone(h.pos) || !monotone(v.pos) || r.pw < 1 || r.ph < 1) return false;
    sx_max = 0;
    for (int x0 = 0; x0 < r.canvas_w; x0 += 64) {
        const int a = std::max(x0, r.px), b = std::min(std::min(x0 + 64, r.canvas_w), r.px + r.pw);
        if (a < b) sx_max = std::max(sx_max, h.pos[r.ox + b - 1 - r.px] + h.size - h.pos[r.ox + a - r.px]);
    }
    for (int cand : {64, 48, 32, 24, 16, 12, 8, 6, 4, 3, 2, 1}) {
        int sy = 0;
        for (int y0 = 0; y0 < r.canvas_h; y0 += cand) {
            const int a = std::max(y0, r.py), b = std::min(std::min(y0 + cand, r.canvas_h), r.py + r.ph);
            if (a < b) sy = std::max(sy, v.pos[r.oy + b - 1 - r.py] + v.size - v.pos[r.oy + a - r.py]);
        }
        if (rgbin_tile_lds(sy, sx_max, cand, v.size, nb) <= budget) { th = cand; sy_max = sy; return true; }
    }
    return false;
}

int rgbin_plan_create(const tts_geom &g, RgbInPlan **out)
{
    *out = nullptr;
    // A pad on RGB pixels in front of the conversion (no scale in the chain, include/ttscale.h TTS_RGB_AT_END): the context
    // converts the padded canvas at its own size; the kernels read it through RgbInArgs' vs_* rectangle, never stored
    const bool canvas = g.rgb_stage == TTS_RGB_AT_END;
    const int src_w = canvas ? g.out_w : g.pre_w, src_h = canvas ? g.out_h : g.pre_h, dst_w = canvas ? g.out_w : g.sc_w, dst_h = canvas ? g.out_h : g.sc_h;
    const bool half = rgb_in_half_chroma(src_w, dst_w);              // sws_init_context()'s chrSrcHSubSample for this context
    if (g.in_fmt == TTS_FMT_YUV420P || g.out_fmt != TTS_FMT_YUV420P || g.sc_range == TTS_RANGE_TO_LIMITED || (half && (src_w & 1))) {
        set_last_error("packed RGB in: the chain must hand out yuv420p or yuvj420p and, unless it widens the picture, read an even width");
        return TTS_ERR_UNSUPPORTED;
    }
    RgbInPlan *p = new (std::nothrow) RgbInPlan();
    if (!p) return TTS_ERR_INITIALIZATION_FAILED;
    p->g = g;
    p->half = half;
    p->pb = packed_pixel_bytes(g.in_fmt);
    p->blue_first = packed_blue_first(g.in_fmt);
    p->canvas = canvas;
    p->range15 = g.sc_range == TTS_RANGE_TO_FULL;            // the converting context writes yuvj420p: the source-row walk and the general kernel carry its range step
    p->yv12 = bgr24_special_converter(g);
    if (p->yv12 && (src_w & 1)) {
        set_last_error("swscale's bgr24 -> yuv420p converter for pictures of the same size never writes the last column of an odd width: not supported");
        delete p;
        return TTS_ERR_UNSUPPORTED;
    }
    const bool three = p->pb == 3 && !p->yv12 && !canvas;    // the conversion walk and the strip walk read 3-byte pixels
    int rc = TTS_OK;
    if (!build_axis_filter(src_w, dst_w, 1 << 14, g.alg, p->lh) || !build_axis_filter(src_h, dst_h, 1 << 12, g.alg, p->lv) ||
        !build_axis_filter(half ? src_w / 2 : src_w, (dst_w + 1) >> 1, 1 << 14, g.alg, p->ch) || !build_axis_filter(src_h, (dst_h + 1) >> 1, 1 << 12, g.alg, p->cv)) {
        set_last_error("filter tables"); rc = TTS_ERR_INVALID_INPUT;
    }
    const AxisFilter *f[4] = {&p->lh, &p->lv, &p->ch, &p->cv};
    for (int i = 0; i < 4 && !rc; ++i)
        if (!upload_vec(p->d_pos[i], f[i]->pos) || !upload_vec(p->d_taps[i], f[i]->taps)) { set_last_error("table upload failed (hipMalloc / hipMemcpy)"); rc = TTS_ERR_INITIALIZATION_FAILED; }
    if (!rc && !p->yv12 && !canvas && !p->range15 && !std::getenv("TTS_FORCE_GENERIC")) {
        // 32 KiB a workgroup if some tile height fits that (five workgroups on a CU's 160 KiB), else 64 KiB
        const RgbInRect r0 = rgbin_rect(g, 0), r1 = rgbin_rect(g, 1);
        for (size_t budget : {size_t(32768), size_t(65536)})        // (64 KiB: the most a workgroup gets without opting in)
            if (!p->tiled)
                p->tiled = rgbin_tile_shape(r0, p->lh, p->lv, p->pb, budget, p->th[0], p->sx_max[0], p->sy_max[0]) &&
                           rgbin_tile_shape(r1, p->ch, p->cv, half ? 2 * p->pb : p->pb, budget, p->th[1], p->sx_max[1], p->sy_max[1]);
        if (p->tiled) {
            p->tiles_y[0] = (r0.canvas_h + p->th[0] - 1) / p->th[0]; p->tiles_y[1] = (r1.canvas_h + p->th[1] - 1) / p->th[1];
            const AxisFilter *vf[2] = {&p->lv, &p->cv};
            for (int k = 0; k < 2; ++k) p->lds_bytes = std::max(p->lds_bytes, rgbin_tile_lds(p->sy_max[k], p->sx_max[k], p->th[k], vf[k]->size, k && half ? 2 * p->pb : p->pb));
        }
    }
    // The conversion alone (nothing in the chain resizes, crops or pads): the streaming walk.  Its arithmetic has the
    // one-tap filters folded in, so they are checked to be exactly that: position i, the full weight.
    if (!rc && three && !std::getenv("TTS_FORCE_GENERIC") && !(std::getenv("TTS_NO_RGBIN_WALK") && std::atoi(std::getenv("TTS_NO_RGBIN_WALK")))) {
        auto identity = [](const AxisFilter &f, int one) {
            if (f.size != 1) return false;
            for (size_t i = 0; i < f.pos.size(); ++i) if (f.pos[i] != int(i) || f.taps[i] != one) return false;
            return true;
        };
        auto monotone = [](const std::vector<int32_t> &v) { for (size_t i = 1; i < v.size(); ++i) if (v[i] < v[i - 1]) return false; return true; };
        const int ny = (g.sc_h + 1) >> 1;
        if (!g.do_pad && g.sc_w == g.pre_w && g.sc_h == g.pre_h && g.crop_x == 0 && g.crop_y == 0 && g.crop_w == g.sc_w && g.crop_h == g.sc_h &&
            g.out_w == g.sc_w && g.out_h == g.sc_h && g.pre_w % 8 == 0 && g.pre_h >= 8 && identity(p->lh, 1 << 14) && identity(p->lv, 1 << 12) &&
            identity(p->ch, 1 << 14) && p->cv.size <= 8 && int(p->cv.pos.size()) == ny && monotone(p->cv.pos)) {
            const int vt = p->cv.size;
            std::vector<int32_t> tabw(size_t(ny) * 5);
            for (int y = 0; y < ny; ++y) {
                tabw[y] = p->cv.pos[y] + vt - 1;
                int16_t slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int j = 0; j < vt; ++j) slot[(p->cv.pos[y] + j) & 7] = p->cv.taps[size_t(y) * vt + j];
                for (int k = 0; k < 4; ++k) tabw[size_t(ny) + 4 * size_t(y) + k] = int32_t(uint32_t(uint16_t(slot[2 * k])) | uint32_t(uint16_t(slot[2 * k + 1])) << 16);
            }
            if (!upload_vec(p->d_walk, tabw)) { set_last_error("table upload failed (hipMalloc / hipMemcpy)"); rc = TTS_ERR_INITIALIZATION_FAILED; }
            else p->walk = true;
        }
    }
    // Horizontal enlargements (full-width chroma; <= 4 luma taps per axis, <= 8 chroma taps per axis): the strip walk on
    // 16-bit samples, if every wave's runs of source pixels fit what its 64 lanes convert per row
    if (!rc && three && !p->walk && !std::getenv("TTS_FORCE_GENERIC") && !(std::getenv("TTS_NO_RGBIN_STRIP") && std::atoi(std::getenv("TTS_NO_RGBIN_STRIP")))) {
        auto monotone = [](const std::vector<int32_t> &v) { for (size_t i = 1; i < v.size(); ++i) if (v[i] < v[i - 1]) return false; return true; };
        const int lht = p->lh.size, cht = p->ch.size, lvt = p->lv.size, cvt = p->cv.size;
        bool ok = !half && !(g.pre_w & 1) && !g.do_pad && g.out_w == g.crop_w && g.out_h == g.crop_h && g.out_w % 8 == 0 && g.out_w >= 64 && lht <= 4 && cht <= 8 && lvt <= 4 && cvt <= 8 &&
                  g.pre_h >= 8 && monotone(p->lh.pos) && monotone(p->ch.pos) && monotone(p->lv.pos) && monotone(p->cv.pos);
        if (ok) {
            const int ox = g.crop_x, oy = g.crop_y, oxc = g.crop_x >> 1, oyc = g.crop_y >> 1;
            const int strips = g.out_w / 8, oh = g.out_h, och = (g.out_h + 1) >> 1;
            const int chd = cht <= 4 ? 2 : cht <= 6 ? 3 : 4, st = 28 + 4 * chd;
            ok = ox + g.out_w <= int(p->lh.pos.size()) && oxc + g.out_w / 2 <= int(p->ch.pos.size()) && oy + oh <= int(p->lv.pos.size()) && oyc + och <= int(p->cv.pos.size());
            std::vector<int32_t> t;
            std::vector<int> slo(size_t(strips), 0), shi(size_t(strips), 0);     // source pixels a strip's windows touch: [lo, hi), lo even
            if (ok) {
                auto pk = [](int a0, int a1) { return int32_t(uint32_t(uint16_t(int16_t(a0))) | uint32_t(uint16_t(int16_t(a1))) << 16); };
                p->so_strip = 0;
                t.resize(size_t(st) * strips, 0);
                for (int sidx = 0; sidx < strips; ++sidx) {
                    int32_t *T = t.data() + size_t(st) * sidx;
                    int lo = INT32_MAX, hi = 0;
                    for (int c = 0; c < 8; ++c) {
                        const int vx = ox + 8 * sidx + c;
                        int tp[4] = {0, 0, 0, 0};
                        for (int k = 0; k < lht; ++k) tp[k] = p->lh.taps[size_t(vx) * lht + k];
                        T[c] = p->lh.pos[vx];                                    // first sample of the window (a pixel of the pre-cropped row)
                        T[8 + 2 * c] = pk(tp[0], tp[1]); T[9 + 2 * c] = pk(tp[2], tp[3]);
                        lo = std::min(lo, p->lh.pos[vx]); hi = std::max(hi, p->lh.pos[vx] + lht);
                    }
                    for (int j = 0; j < 4; ++j) {
                        const int vx = oxc + 4 * sidx + j;
                        int tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int k = 0; k < cht; ++k) tp[k] = p->ch.taps[size_t(vx) * cht + k];
                        T[24 + j] = p->ch.pos[vx];
                        for (int k = 0; k < chd; ++k) T[28 + chd * j + k] = pk(tp[2 * k], tp[2 * k + 1]);
                        lo = std::min(lo, p->ch.pos[vx]); hi = std::max(hi, p->ch.pos[vx] + cht);
                    }
                    slo[sidx] = lo & ~1; shi[sidx] = hi;
                }
                // per wave start sA (the wave's first strip inside its first frame) and lane: what the lane converts
                // {first pixel or -1, frame - first frame} and where its strip lies {frame - first frame, LDS sample of
                // pixel 0 of that frame's row}: a wave's 64 strips are cut into one piece per frame they lie in, every
                // piece's run of pixels is dealt PXL a lane, piece after piece
                p->so_wave = uint32_t(t.size());
                t.resize(t.size() + size_t(4) * 64 * strips, 0);
                int pxl = 0;
                const int avail = g.in_w - g.pre_x;                              // pixels from the pre-cropped origin to the end of a source row: a lane's PXL pixels stay inside them
                for (int cand : {4, 6, 8}) {
                    bool fits = avail >= cand;
                    for (int sA = 0; sA < strips && fits; ++sA) {
                        int32_t *W = t.data() + p->so_wave + size_t(4) * 64 * sA;
                        for (int i = 0; i < 64; ++i) { W[4 * i] = -1; W[4 * i + 1] = 0; }
                        int used = 0;                                          // conversion lanes dealt so far
                        for (int i = 0, fd = 0; i < 64 && fits; ++fd) {
                            const int first = fd ? 0 : sA, count = std::min(64 - i, strips - first), lastS = first + count - 1;
                            // (a piece shorter than a lane's pixels at the end of a row starts early enough for its first lane to stay inside the row)
                            const int lo = std::min(slo[first], (avail - cand) & ~1);
                            int hi = 0;
                            for (int q = first; q <= lastS; ++q) hi = std::max(hi, shi[q]);
                            const int lanes = (hi - lo + cand - 1) / cand;
                            if (used + lanes > 64 || fd > 255) { fits = false; break; }
                            for (int q = 0; q < lanes; ++q) {
                                // a lane whose pixels would pass the end of the row is slid back over its neighbour's (same values, written twice)
                                const int px = std::min(lo + cand * q, (avail - cand) & ~1);
                                W[4 * (used + q)] = px;
                                W[4 * (used + q) + 1] = int32_t(uint32_t((cand * used + (px - lo)) / 2) << 8 | uint32_t(fd));   // first LDS dword of the lane's samples | frame - first frame
                            }
                            for (int q = 0; q < count; ++q) { W[4 * (i + q) + 2] = fd; W[4 * (i + q) + 3] = cand * used - lo; }
                            used += lanes; i += count;
                        }
                    }
                    if (fits) { pxl = cand; break; }
                }
                ok = pxl != 0;
                if (ok) {
                    p->so_lastl = uint32_t(t.size()); t.resize(t.size() + oh);
                    p->so_vcol = uint32_t(t.size()); t.resize(t.size() + size_t(2) * oh);
                    for (int y = 0; y < oh; ++y) {
                        const int pos = p->lv.pos[oy + y];
                        int slot[4] = {0, 0, 0, 0};
                        for (int j = 0; j < lvt; ++j) slot[(pos + j) & 3] = p->lv.taps[size_t(oy + y) * lvt + j];
                        t[p->so_lastl + y] = pos + lvt - 1;
                        t[p->so_vcol + 2 * size_t(y)] = pk(slot[0], slot[1]); t[p->so_vcol + 2 * size_t(y) + 1] = pk(slot[2], slot[3]);
                    }
                    p->so_lastc = uint32_t(t.size()); t.resize(t.size() + och);
                    p->so_vcoc = uint32_t(t.size()); t.resize(t.size() + size_t(4) * och);
                    for (int y = 0; y < och; ++y) {
                        const int pos = p->cv.pos[oyc + y];
                        int slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int j = 0; j < cvt; ++j) slot[(pos + j) & 7] = p->cv.taps[size_t(oyc + y) * cvt + j];
                        t[p->so_lastc + y] = pos + cvt - 1;
                        for (int k = 0; k < 4; ++k) t[p->so_vcoc + 4 * size_t(y) + k] = pk(slot[2 * k], slot[2 * k + 1]);
                    }
                    if (!upload_vec(p->d_strip, t)) { set_last_error("table upload failed (hipMalloc / hipMemcpy)"); rc = TTS_ERR_INITIALIZATION_FAILED; }
                    else { p->strip = true; p->strips = strips; p->pxl = pxl; p->chd = chd; p->seg_knob = std::getenv("TTS_RGBIN_SEG") ? std::atoi(std::getenv("TTS_RGBIN_SEG")) : 0; }
                }
            }
        }
    }
    if (!rc && !p->walk && !p->strip && !p->yv12 && !canvas) {   // (rgba / bgra too: the walk skips the fourth byte; and it carries the range step of a context that writes yuvj420p)
        int coef[9];
        rgbin_coefficients(coef, p->blue_first);
        p->shrink = rgbin_shrink_create(g, p->lh, p->lv, p->ch, p->cv, coef, p->pb);
    }
    p->yv12_rows = p->yv12 && !g.do_pad && g.crop_x == 0 && g.crop_y == 0 && g.crop_w == g.sc_w && g.crop_h == g.sc_h && g.out_w == g.sc_w && g.out_h == g.sc_h &&
                   g.pre_w >= 8 && !std::getenv("TTS_FORCE_GENERIC");
    if (p->yv12) std::snprintf(p->name, sizeof p->name, p->yv12_rows ? "bgr24_yv12_rows" : "bgr24_yv12");
    else if (p->shrink) std::snprintf(p->name, sizeof p->name, "%s", rgbin_shrink_name(p->shrink));
    else if (p->walk) std::snprintf(p->name, sizeof p->name, "rgbin_walk<%d>%s", p->cv.size, p->range15 ? "+range15" : "");
    else if (p->strip) std::snprintf(p->name, sizeof p->name, "rgbin_strip<%d,%d,%d,%d,%dpx>%s", p->lh.size, p->lv.size, p->ch.size, p->cv.size, p->pxl, p->range15 ? "+range15" : "");   // luma taps across / down, chroma taps across / down
    else if (p->tiled) std::snprintf(p->name, sizeof p->name, "rgbin_tile<%d,%d>", p->th[0], p->th[1]);
    else std::snprintf(p->name, sizeof p->name, "rgbin_generic<%d,%d,%d,%d>%s%s", p->lh.size, p->lv.size, p->ch.size, p->cv.size, p->range15 ? "+range15" : "", canvas ? "+rgb_pad" : "");
    if (p->yv12 && canvas) std::snprintf(p->name, sizeof p->name, "bgr24_yv12+rgb_pad");
    if (rc) { rgbin_plan_destroy(p); return rc; }
    *out = p;
    return TTS_OK;
}

void rgbin_plan_destroy(RgbInPlan *p)
{
    if (!p) return;
    for (int i = 0; i < 4; ++i) { (void)hipFree(p->d_pos[i]); (void)hipFree(p->d_taps[i]); }
    (void)hipFree(p->d_walk); (void)hipFree(p->d_strip);
    rgbin_shrink_destroy(p->shrink);
    delete p;
}

const char *rgbin_plan_kernel_name(const RgbInPlan *p) { return p ? p->name : ""; }

int rgbin_launch(const RgbInPlan *p, const uint8_t *src, uint8_t *dst, int n, int src_stride, const int dst_stride[2],
                 const FrameAddr &addr, hipStream_t st, int only_plane)
{
    const tts_geom &g = p->g;
    RgbInArgs a{};
    a.src = src; a.dst = dst;
    a.lhpos = p->d_pos[0]; a.lvpos = p->d_pos[1]; a.chpos = p->d_pos[2]; a.cvpos = p->d_pos[3];
    a.lhtaps = p->d_taps[0]; a.lvtaps = p->d_taps[1]; a.chtaps = p->d_taps[2]; a.cvtaps = p->d_taps[3];
    a.lht = p->lh.size; a.lvt = p->lv.size; a.cht = p->ch.size; a.cvt = p->cv.size;
    for (int k = 0; k < 2; ++k) {
        RgbInRect r = rgbin_rect(g, k);
        if (p->canvas) r = {(g.out_w + k) >> k, (g.out_h + k) >> k, 0, 0, (g.out_w + k) >> k, (g.out_h + k) >> k, 0, 0};   // every output sample comes out of the conversion
        a.canvas_w[k] = r.canvas_w; a.canvas_h[k] = r.canvas_h;
        a.px[k] = r.px; a.py[k] = r.py; a.pw[k] = r.pw; a.ph[k] = r.ph; a.ox[k] = r.ox; a.oy[k] = r.oy;
        a.dst_stride[k] = dst_stride[k];
    }
    a.src_stride = src_stride;
    a.only_plane = only_plane;
    a.chr_half = p->half ? 1 : 0;
    a.pb = p->pb;
    a.range15 = p->range15 ? 1 : 0;
    a.src_origin = uint32_t(g.pre_y) * uint32_t(src_stride) + uint32_t(g.pre_x) * uint32_t(p->pb);
    a.src_extent = uint32_t(size_t(g.in_h - 1) * size_t(src_stride) + size_t(g.in_w) * size_t(p->pb));
    a.fill[0] = g.fill_y; a.fill[1] = g.fill_u; a.fill[2] = g.fill_v;
    if (p->canvas) {
        a.vs_on = 1; a.vs_x = g.pad_x; a.vs_y = g.pad_y; a.vs_w = g.pre_w; a.vs_h = g.pre_h;
        const int fr = (g.fill_rgb >> 16) & 255, fg = (g.fill_rgb >> 8) & 255, fb = g.fill_rgb & 255;
        a.vs_fill[0] = p->blue_first ? fb : fr; a.vs_fill[1] = fg; a.vs_fill[2] = p->blue_first ? fr : fb;
    }
    int coef[9];
    rgbin_coefficients(coef, p->blue_first);
    for (int i = 0; i < 9; ++i) a.coef[i] = coef[i];
    a.addr = addr;
    int rc = TTS_OK;
    if (p->yv12 && !(p->yv12_rows && only_plane < 0)) {
        for (int f0 = 0; f0 < n && !rc; f0 += 21845) {      // grid.z = frames x 3 planes
            a.frame0 = f0;
            rc = launch_rgb("bgr24_yv12_kernel", bgr24_yv12_kernel, dim3((g.out_w + 63) / 64, (g.out_h + 3) / 4, std::min(21845, n - f0) * 3), dim3(kBlock), 0, st, a);
        }
        return rc;
    }
    if (p->yv12_rows && only_plane < 0) {
        const int groups = (g.out_w + 7) / 8, pairs = (g.out_h + 1) / 2;
        for (int f0 = 0; f0 < n && !rc; f0 += 65535) {
            a.frame0 = f0;
            rc = launch_rgb("bgr24_yv12_rows_kernel", bgr24_yv12_rows_kernel, dim3((groups + 63) / 64, (pairs + 3) / 4, std::min(65535, n - f0)), dim3(kBlock), 0, st, a);
        }
        return rc;
    }
    // the conversion alone, frames at a fixed pitch, rows where 8-byte loads and stores may go: the streaming walk
    bool walk = p->walk && only_plane < 0 && reinterpret_cast<uintptr_t>(src) % 8 == 0 && src_stride % 8 == 0 && a.src_origin % 8 == 0 &&
                reinterpret_cast<uintptr_t>(dst) % 8 == 0 && dst_stride[0] % 8 == 0 && dst_stride[1] % 4 == 0;
    if (walk && addr.use_table) {
        for (int i = 0; i < n && walk; ++i)
            walk = addr.src_tab[0][i] % 8 == 0 && addr.dst_tab[0][i] % 8 == 0 && addr.dst_tab[1][i] % 4 == 0 && addr.dst_tab[2][i] % 4 == 0;
    } else if (walk)
        walk = addr.src_pitch % 8 == 0 && addr.src_plane[0] % 8 == 0 && addr.dst_pitch % 8 == 0 && addr.dst_plane[0] % 8 == 0 &&
               addr.dst_plane[1] % 4 == 0 && addr.dst_plane[2] % 4 == 0;
    if (walk) {
        RgbInWalkArgs w{};
        w.ny = (g.sc_h + 1) >> 1;
        w.w = g.pre_w; w.h = g.pre_h; w.vt = p->cv.size;
        w.src_stride = src_stride; w.dst_stride[0] = dst_stride[0]; w.dst_stride[1] = dst_stride[1];
        w.src_origin = a.src_origin;
        for (int i = 0; i < 9; ++i) w.coef[i] = coef[i];
        w.addr = addr;
        // row segments: enough waves for many rounds of the chip, none shorter than 32 rows (each starts 7 rows early)
        const int wpr = (g.pre_w / 8 + 63) / 64;
        const int64_t want = 16 * 2048;                              // (many short rounds: 4 x 2048 left 1.6 rounds at 300 frames of 1080p and ran 6 % slower)
        int nseg = int(std::min<int64_t>(std::max<int64_t>(1, (want + int64_t(wpr) * n - 1) / (int64_t(wpr) * n)), std::max(1, g.pre_h / 32)));
        w.seg_rows = (g.pre_h + nseg - 1) / nseg;
        w.nseg = (g.pre_h + w.seg_rows - 1) / w.seg_rows;
        for (int f0 = 0; f0 < n && !rc; f0 += 65535) {
            w.frame0 = f0;
            rc = launch_rgb("rgbin_walk_kernel", p->range15 ? rgbin_walk_kernel<true> : rgbin_walk_kernel<false>, dim3(unsigned((wpr * w.nseg + kBlock / 64 - 1) / (kBlock / 64)), 1, unsigned(std::min(65535, n - f0))),
                            dim3(kBlock), 0, st, src, dst, static_cast<const int32_t *>(p->d_walk), w);
        }
        return rc;
    }
    // enlargements: the strip walk on 16-bit samples, where the output rows lie on the grid its stores need
    bool strip = p->strip && only_plane < 0 && reinterpret_cast<uintptr_t>(dst) % 8 == 0 && dst_stride[0] % 8 == 0 && dst_stride[1] % 4 == 0 && int64_t(n) * p->strips < (int64_t(1) << 30);
    if (strip && addr.use_table) {
        for (int i = 0; i < n && strip; ++i) strip = addr.dst_tab[0][i] % 8 == 0 && addr.dst_tab[1][i] % 4 == 0 && addr.dst_tab[2][i] % 4 == 0;
    } else if (strip)
        strip = addr.dst_pitch % 8 == 0 && addr.dst_plane[0] % 8 == 0 && addr.dst_plane[1] % 4 == 0 && addr.dst_plane[2] % 4 == 0;
    if (strip) {
        RgbInStripArgs w{};
        w.o_strip = p->so_strip; w.o_wave = p->so_wave; w.o_lastl = p->so_lastl; w.o_vcol = p->so_vcol; w.o_lastc = p->so_lastc; w.o_vcoc = p->so_vcoc;
        w.strips = p->strips; w.total = n * p->strips; w.nframes = n;
        w.oh = g.out_h; w.och = (g.out_h + 1) >> 1; w.lvt = p->lv.size; w.cvt = p->cv.size; w.h = g.pre_h;
        w.src_stride = src_stride; w.dst_stride[0] = dst_stride[0]; w.dst_stride[1] = dst_stride[1];
        w.src_origin = a.src_origin;
        w.range15 = p->range15 ? 1 : 0;
        // rgb24ToY_c: (ry r + gy g + by b + (32 << 14) + (1 << 8)) >> 9, rgb24ToUV_c: (... + (256 << 14) + (1 << 8)) >> 9, in the
        // kernel's exact float form: coefficients x 2^-9, constant (2 bias + 1) / 1024
        for (int pl = 0; pl < 3; ++pl) {
            for (int k = 0; k < 3; ++k) w.cf[pl][k] = float(coef[3 * pl + k]) / 512.0f;
            w.cf[pl][3] = float(2 * ((pl ? 256 << 14 : 32 << 14) + (1 << 8)) + 1) / 1024.0f;
        }
        w.addr = addr;
        // row segments: enough workgroups for about four rounds of the chip's resident waves (3 a SIMD), none shorter than
        // 16 source rows (each starts up to 7 rows early to fill the chroma ring)
        const int64_t waves = (int64_t(w.total) + 63) / 64;
        const int seg_env = p->seg_knob;
        int nseg = int(std::max<int64_t>(1, std::min<int64_t>((int64_t(4) * 3072 + waves - 1) / waves, std::max(1, g.pre_h / 16))));
        if (seg_env > 0) nseg = std::max(1, std::min(seg_env, g.pre_h / 8));
        w.seg_rows = (g.pre_h + nseg - 1) / nseg;
        w.nseg = (g.pre_h + w.seg_rows - 1) / w.seg_rows;
        auto fn = p->pxl == 4 ? (p->chd == 2 ? rgbin_strip_kernel<4, 2> : p->chd == 3 ? rgbin_strip_kernel<4, 3> : rgbin_strip_kernel<4, 4>)
                : p->pxl == 6 ? (p->chd == 2 ? rgbin_strip_kernel<6, 2> : p->chd == 3 ? rgbin_strip_kernel<6, 3> : rgbin_strip_kernel<6, 4>)
                              : (p->chd == 2 ? rgbin_strip_kernel<8, 2> : p->chd == 3 ? rgbin_strip_kernel<8, 3> : rgbin_strip_kernel<8, 4>);
        return launch_rgb("rgbin_strip_kernel", fn, dim3(unsigned((waves + kBlock / 64 - 1) / (kBlock / 64)), unsigned(w.nseg), 1), dim3(kBlock), 0, st, src, dst, static_cast<const int32_t *>(p->d_strip), w);
    }
    // shrinks: the source-row walk, where the output planes lie on the halfword grid its stores need
    bool shrink = p->shrink && only_plane < 0 && reinterpret_cast<uintptr_t>(dst) % 2 == 0 && dst_stride[0] % 2 == 0 && int64_t(n) * ((g.out_w + 3) / 4) < (int64_t(1) << 30);
    if (shrink && addr.use_table) {
        for (int i = 0; i < n && shrink; ++i) shrink = addr.dst_tab[0][i] % 2 == 0;
    } else if (shrink)
        shrink = addr.dst_pitch % 2 == 0 && addr.dst_plane[0] % 2 == 0;
    if (shrink) return rgbin_shrink_launch(p->shrink, src, dst, n, src_stride, dst_stride, addr, st);
    for (int f0 = 0; f0 < n && !rc; f0 += 21845) {          // grid.z = frames x 3 planes
        const int nf = std::min(21845, n - f0);
        a.frame0 = f0;
        if (p->tiled) {
            RgbInTileArgs t{};
            t.g = a;
            for (int k = 0; k < 2; ++k) { t.th[k] = p->th[k]; t.sx_max[k] = p->sx_max[k]; t.sy_max[k] = p->sy_max[k]; t.tiles_y[k] = p->tiles_y[k]; }
            for (int k = 0; k < 2; ++k) t.raw_w[k] = rgbin_raw_w(p->sx_max[k], k && p->half ? 2 * p->pb : p->pb);
            const bool t8 = std::max(std::max(a.lht, a.lvt), std::max(a.cht, a.cvt)) <= 8;
            rc = launch_rgb("rgbin_tile_kernel", t8 ? rgbin_tile_kernel<8> : rgbin_tile_kernel<0>,
                            dim3((g.out_w + 63) / 64, std::max(p->tiles_y[0], p->tiles_y[1]), nf * 3), dim3(kBlock), p->lds_bytes, st, t);
        } else
            rc = launch_rgb("rgbin_generic_kernel", rgbin_generic_kernel, dim3((g.out_w + 63) / 64, (g.out_h + 3) / 4, nf * 3), dim3(kBlock), 0, st, a);
    }
    return rc;
}

}  // namespace tts
