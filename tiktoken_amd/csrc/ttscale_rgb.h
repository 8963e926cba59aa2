// ttscale_rgb.h -- what ttscale.hip (the C ABI, the planar scaler) and ttscale_rgb.hip (the packed-RGB
// output paths: SURVEY.md 8f-1) know of each other.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

#include "../../include/ttscale.h"
#include "ttscale_device.h"

namespace tts {

// the calling thread's tts_last_error() text (defined in ttscale.hip)
void set_last_error(const std::string &msg);

// BT.601 limited-range (or, for a yuvj420p source, full-range) tables of ff_yuv2rgb_c_init_tables() as a closed form:
// a channel is clip_u8(((Y + off) * cy + k) >> 16) with off = ((clip_u8(C) * inc) >> 16) - (inc >> 9) per chroma term
struct RgbCoef {
    int32_t cy, crv, cbu, cgu, cgv;     // cy = 65536 * 255 / 219; the chroma increments "scaled by cy"
    int32_t k;                          // 326 cy - (400 << 16) + 0x8000
    int32_t off_r, off_g, off_b;        // -(crv >> 9), -(cgu >> 9) - (cgv >> 9), -(cbu >> 9)
};

struct RgbPlan;        // tables of one geometry whose output is packed RGB, on the current device

// TTS_OK, or why this geometry cannot go out as packed RGB (message through set_last_error)
int  rgb_plan_create(const tts_geom &g, int n_cus, RgbPlan **out);
void rgb_plan_destroy(RgbPlan *p);
const char *rgb_plan_kernel_name(const RgbPlan *p);
bool rgb_plan_is_strip(const RgbPlan *p);
bool rgb_plan_general_only(const RgbPlan *p);   // rows through swscale's 1- / 2-line writers, or a full-range source: no fused or in-place path          // the fused strip kernel takes this geometry (<= 4 taps an axis)

// One launch set: frames [0, n) addressed by `addr` relative to (src, dst); src_stride = luma / chroma row
// pitch of the yuv420p source, dst_stride = bytes between RGB rows.  Asynchronous on st.
int rgb_launch(const RgbPlan *p, const uint8_t *src, uint8_t *dst, int n, const int src_stride[2], int dst_stride,
               uint64_t src_span, uint64_t dst_span, const FrameAddr &addr, hipStream_t st);

// ---- packed-RGB output of a scale with 5-8 taps an axis: one fused walk (ttscale_rgbwalk.hip) ---------------
struct AxisFilter;
struct RgbWalkPlan;
// nullptr: the geometry does not fit the walk (more than 8 taps, a pad, windows that do not group, runs too long for the LDS slots)
RgbWalkPlan *rgb_walk_create(const tts_geom &g, const AxisFilter &lh, const AxisFilter &lv, const AxisFilter &ch, const AxisFilter &cv,
                             int bpp, int swap, const RgbCoef &coef, int n_cus);
void rgb_walk_destroy(RgbWalkPlan *p);
const char *rgb_walk_name(const RgbWalkPlan *p);
bool rgb_walk_takes(const RgbWalkPlan *p, int n, uint64_t src_span, const FrameAddr &addr);    // packed batches whose waves fit 32-bit offsets
int rgb_walk_launch(const RgbWalkPlan *p, const uint8_t *src, uint8_t *dst, int n, const int src_stride[2], int dst_stride,
                    uint64_t src_span, uint64_t dst_span, const FrameAddr &addr, hipStream_t st);
// the plan's walk, if it has one and it takes this packed batch (then no in-place scheme, no general kernel)
bool rgb_plan_walks(const RgbPlan *p, int n, uint64_t src_span, uint32_t src_pitch, uint32_t dst_pitch);

// ---- packed-RGB shrinks: planar column walk into the destination frame, converted in place ---------------
// Where the planar pass leaves its rows inside an RGB frame whose rows are dst_stride bytes apart (offsets from
// the frame's canvas origin): picture row y holds, in the bytes its pixels will take, crop_w int16 of unclipped
// luma, then ceil(crop_w / 2) bytes of U and of V.  false: the geometry does not qualify (conversion not inside
// the chain's scale, a crop at an odd offset, an odd width at 3 bytes a pixel with no border to its right).
struct RgbScratchLayout { uint32_t off[3]; int stride[2]; };
bool rgb_scratch_layout(const tts_geom &g, int dst_stride, RgbScratchLayout &out);
// frames [0, n) addressed by addr's destination side (dst_plane[0] / dst_tab[0] = a frame's canvas origin): one
// workgroup per canvas row reads the row's planes, then writes its pixels and border.  Asynchronous on st.
int rgb_scratch_convert(const RgbPlan *p, uint8_t *dst, int dst_stride, int n, const FrameAddr &addr, hipStream_t st);

// ---- packed rgb24 INPUT, yuv420p out (tts_geom.in_fmt == TTS_FMT_RGB24) ----------------------------------
struct RgbInPlan;
int  rgbin_plan_create(const tts_geom &g, RgbInPlan **out);
void rgbin_plan_destroy(RgbInPlan *p);
const char *rgbin_plan_kernel_name(const RgbInPlan *p);
// frames [0, n): addr.src_plane[0] = the pre-cropped origin inside a source frame, addr.dst_plane[k] = the
// origin of canvas plane k inside an output frame (packed mode only); src_stride = bytes between RGB rows
// only_plane >= 0: that destination plane alone (dst and the tables then belong to it)
int rgbin_launch(const RgbInPlan *p, const uint8_t *src, uint8_t *dst, int n, int src_stride, const int dst_stride[2],
                 const FrameAddr &addr, hipStream_t st, int only_plane = -1);

// ---- packed rgb24 in, shrunk (at most 8 taps across): one walk over the source rows (ttscale_rgbin_shrink.hip) ----
struct RgbInShrinkPlan;
// nullptr: the geometry does not fit (an enlargement across, more than 8 taps across, a pad, odd widths, ...)
RgbInShrinkPlan *rgbin_shrink_create(const tts_geom &g, const AxisFilter &lh, const AxisFilter &lv, const AxisFilter &ch, const AxisFilter &cv, const int coef[9], int pb = 3);
void rgbin_shrink_destroy(RgbInShrinkPlan *p);
const char *rgbin_shrink_name(const RgbInShrinkPlan *p);
int rgbin_shrink_launch(const RgbInShrinkPlan *p, const uint8_t *src, uint8_t *dst, int n, int src_stride, const int dst_stride[2],
                        const FrameAddr &addr, hipStream_t st);

inline int  packed_pixel_bytes(int fmt) { return fmt >= TTS_FMT_RGBA ? 4 : 3; }
inline bool packed_blue_first(int fmt) { return fmt == TTS_FMT_BGR24 || fmt == TTS_FMT_BGRA; }

// utils.c fill_rgb2yuv_table(): the default matrix's table (BT.601 limited range), 15 fractional bits: ry gy by, ru gu bu, rv gv bv
// -- by BYTE of a pixel: blue_first (bgr24, bgra) swaps the first and the third of each row, and every kernel reads
// bytes 0, 1, 2 of a pixel as it does for rgb24
inline void rgbin_coefficients(int coef[9], bool blue_first = false)
{
    const double S = 32768.0;
    const int c[9] = {int(0.299 * 219 / 255 * S + 0.5), int(0.587 * 219 / 255 * S + 0.5), int(0.114 * 219 / 255 * S + 0.5),
                      -int(0.169 * 224 / 255 * S + 0.5), -int(0.331 * 224 / 255 * S + 0.5), int(0.500 * 224 / 255 * S + 0.5),
                      int(0.500 * 224 / 255 * S + 0.5), -int(0.419 * 224 / 255 * S + 0.5), -int(0.081 * 224 / 255 * S + 0.5)};
    for (int i = 0; i < 9; ++i) coef[i] = c[blue_first ? 3 * (i / 3) + 2 - i % 3 : i];
}

}  // namespace tts
