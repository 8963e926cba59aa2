// Host-side planning for libttscale: integer filter tables and chain geometry.
// Product code (never includes or links anything under oracle/).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/ttscale.h"

namespace tts {

// One axis of one plane: taps[i*size + j] weighs source sample pos[i] + j.
struct AxisFilter {
    int src = 0, dst = 0, size = 0;
    std::vector<int16_t> taps;
    std::vector<int32_t> pos;
};

// swscale-compatible table for resampling `src` samples to `dst` samples
// (SURVEY.md S1-S3, S6).  one = 1<<14 horizontally, 1<<12 vertically.
bool build_axis_filter(int src, int dst, int one, int alg, AxisFilter &out);

// Geometry helpers.
bool scale_output_size(int iw, int ih, int w, int h, int foar, int fdb, int &ow, int &oh);
int  crop_origin(int in, int out, double value, bool has_value, int sub = 2);   // sub: chroma grid of the link (2 yuv420p, 1 RGB)

// Packed-RGB source into a 4:2:0 destination (sws_init_context(): "drop every other pixel for chroma calculation
// unless user wants full chroma"): chroma is read at half width (rgb24ToUV_half_c, pixel-pair sums) only when the
// chroma output is no wider than half the source, (dst_w >> 1) <= (src_w >> 1) -- the conversion alone and every
// horizontal shrink; an enlargement reads one chroma sample per pixel (rgb24ToUV_c) and shrinks src_w -> ceil(dst_w / 2).
inline bool rgb_in_half_chroma(int src_w, int dst_w) { return (dst_w >> 1) <= (src_w >> 1); }

// bgr24 into yuv420p at the SAME size (no accurate_rnd) is not that path at all: swscale_unscaled.c's "bgr24toYV12" entry
// hands the frame to rgb2rgb_template.c ff_rgb24toyv12_c() (truncating sums, chroma from the left pixel of each pair of
// the even rows only).  Its chromWidth = width >> 1 pairs leave the last column of an odd width unwritten: refused.
// (Ranges that differ -- bgr24 into yuvj420p -- keep the context on the general path: sws_init_context() only asks for an unscaled
// converter when srcRange == dstRange.)
inline bool bgr24_special_converter(const tts_geom &g) { return g.in_fmt == TTS_FMT_BGR24 && g.sc_w == g.pre_w && g.sc_h == g.pre_h && g.sc_range == TTS_RANGE_KEEP; }

// Validate a resolved geometry; returns TTS_OK or an error.
int  check_geom(const tts_geom &g);

// Range step on a finished 8-bit plane as swscale's unscaled-size generic path computes it
// (1-tap H pass: v = p << 7; lum/chrRangeToJpeg_c or ...FromJpeg_c on the 15-bit value; 1-tap V
// pass: (v + 64) >> 7, clipped).  A function of one byte, so a 256-entry table per plane class.
void build_range_lut(int out_range, bool chroma, unsigned char lut[256]);
bool range_affine(int out_range, bool chroma, int &mul, int &add, int &shift);   // the same step in closed form, verified against the table

// "-vf" chain parser; err receives a one-line reason on failure.
int  parse_vf(const char *vf, int iw, int ih, tts_geom &g, std::string &err, int in_fmt = TTS_FMT_YUV420P);

}  // namespace tts
