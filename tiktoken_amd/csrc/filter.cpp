// Integer resampling tables and chain geometry for libttscale (host side).
//
// The tables reproduce what libswscale's C path computes once per geometry
// (SURVEY.md 8a rows S1-S3, S6): 64-bit fixed-point kernel weights, trimming of
// negligible taps, folding of out-of-range taps into the edge sample, and
// normalisation with error diffusion so every row sums to exactly `one`.
// Reference call sites that select these geometries:
//   scripts/processVideos.js:94-97, src/services/video.js:329-332,
//   web/src/tests/video/portrait.test.js:78,95-96.
#include "filter.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace tts {
namespace {

using i64 = int64_t;

constexpr double kTrimCutoff = 0.002;   // share of unity below which edge taps are dropped

inline int floor_log2(unsigned v) { int n = 0; while (v >>= 1) ++n; return n; }
inline i64 iabs(i64 v) { return v < 0 ? -v : v; }
inline i64 div_round(i64 a, i64 b) { return (a >= 0 ? a + (b >> 1) : a - (b >> 1)) / b; }

// Keys cubic with B = 0, C = 0.6 in 2^24 fixed point; d is the tap distance with 2^30 == 1 px.
i64 cubic_weight(i64 d)
{
    const i64 U = 1 << 24;
    const i64 C = static_cast<i64>(0.6 * (1 << 24));
    if (d >= (i64(1) << 31)) return 0;
    const i64 d2 = (d * d) >> 30;
    const i64 d3 = (d2 * d) >> 30;
    if (d < (i64(1) << 30))
        return (12 * U - 6 * C) * d3 + (-18 * U + 6 * C) * d2 + 6 * U * (i64(1) << 30);
    return (-6 * C) * d3 + (30 * C) * d2 + (-48 * C) * d + (24 * C) * (i64(1) << 30);
}

// Box-overlap weight of the area scaler; inc is the 16.16 source step.
i64 area_weight(i64 d, i64 inc)
{
    const i64 half = i64(1) << (29 + 16);
    const i64 e = (d - (1 << 29)) * inc;
    if (e < -half) return i64(1) << (30 + 16);
    if (e < half) return half - e;
    return 0;
}

struct Row {
    int pos;
    std::vector<i64> w;
};

}  // namespace

bool build_axis_filter(int src, int dst, int one, int alg, AxisFilter &out)
{
    if (src < 1 || dst < 1) return false;
    const i64 inc = ((i64(src) << 16) + (dst >> 1)) / dst;
    const int lg = std::min(floor_log2(unsigned(src / dst)), 8);
    const i64 unity = i64(1) << (54 - lg);
    std::vector<Row> rows(dst);
    int width;

    if (std::llabs(inc - 0x10000) < 10) {
        // same size: one unit tap per sample
        width = 1;
        for (int i = 0; i < dst; ++i) rows[i] = Row{i, {unity}};
    } else if (inc <= (1 << 16) && alg == TTS_ALG_AREA) {
        // area scaler asked to enlarge: two-tap linear interpolation
        width = 2;
        i64 centre = inc / 2 - 0x8000;                      // (128*inc >> 8) - (128*0x8000 >> 7)
        for (int i = 0; i < dst; ++i, centre += inc) {
            int x = int((centre - (i64(width - 1) << 15) + (1 << 15)) >> 16);
            rows[i].pos = x;
            rows[i].w.resize(width);
            for (int j = 0; j < width; ++j, ++x) {
                i64 c = unity - iabs(i64(x) * (1 << 16) - centre) * (unity >> 16);
                rows[i].w[j] = c < 0 ? 0 : c;
            }
        }
    } else {
        const int support = alg == TTS_ALG_BICUBIC ? 4 : alg == TTS_ALG_AREA ? 1 : 2;
        const bool shrink = inc > (1 << 16);
        width = shrink ? 1 + (support * src + dst - 1) / dst : 1 + support;
        width = std::max(std::min(width, src - 2), 1);
        // sample centres in 1/2^17 source pixels; both grids are centre-aligned
        i64 centre = inc - 0x10000;
        for (int i = 0; i < dst; ++i, centre += 2 * inc) {
            int x = int((centre - i64(width - 2) * (1 << 16)) / (1 << 17));   // truncating division
            rows[i].pos = x;
            rows[i].w.resize(width);
            for (int j = 0; j < width; ++j, ++x) {
                i64 d = iabs(i64(x) * (1 << 17) - centre) << 13;
                if (shrink) d = d * dst / src;                                // widen the kernel
                i64 c;
                switch (alg) {
                case TTS_ALG_BICUBIC: c = cubic_weight(d) / ((i64(1) << 54) / unity); break;
                case TTS_ALG_AREA:    c = area_weight(d, inc) * (unity >> (30 + 16)); break;
                default:              c = std::max<i64>((i64(1) << 30) - d, 0) * (unity >> 30); break;
                }
                rows[i].w[j] = c;
            }
        }
    }

    // Drop negligible taps: leading ones by sliding the row left (keeping positions
    // monotonic), trailing ones by shrinking the common width.
    const double cutoff = kTrimCutoff * double(unity);
    int keep = 0;
    for (int i = dst - 1; i >= 0; --i) {
        Row &r = rows[i];
        i64 mass = 0;
        for (int j = 0; j < width; ++j) {
            mass += iabs(r.w[0]);
            if (double(mass) > cutoff) break;
            if (i < dst - 1 && r.pos >= rows[i + 1].pos) break;
            r.w.erase(r.w.begin());
            r.w.push_back(0);
            ++r.pos;
        }
        int live = width;
        mass = 0;
        for (int j = width - 1; j > 0; --j) {
            mass += iabs(r.w[j]);
            if (double(mass) > cutoff) break;
            --live;
        }
        keep = std::max(keep, live);
    }
    for (Row &r : rows) r.w.resize(keep);
    width = keep;

    // Fold taps that fall outside [0, src) into the nearest edge sample.
    for (Row &r : rows) {
        if (r.pos < 0) {
            for (int j = 1; j < width; ++j) {
                const int to = std::max(j + r.pos, 0);
                r.w[to] += r.w[j];
                r.w[j] = 0;
            }
            r.pos = 0;
        }
        if (r.pos + width > src) {
            const int over = r.pos + std::min(width - src, 0);
            i64 spill = 0;
            for (int j = width - 1; j >= 0; --j)
                if (r.pos + j >= src) { spill += r.w[j]; r.w[j] = 0; }
            for (int j = width - 1; j >= 0; --j)
                r.w[j] = j < over ? 0 : r.w[j - over];
            r.pos -= over;
            r.w[src - 1 - r.pos] += spill;
        }
    }

    // Quantise to int16 so each row sums to `one`, carrying the rounding error forward.
    out.src = src; out.dst = dst; out.size = width;
    out.taps.assign(size_t(dst) * width, 0);
    out.pos.resize(dst);
    for (int i = 0; i < dst; ++i) {
        const Row &r = rows[i];
        i64 total = 0;
        for (i64 c : r.w) total += c;
        total = (total + one / 2) / one;
        if (total == 0) total = 1;
        i64 carry = 0;
        for (int j = 0; j < width; ++j) {
            const i64 v = r.w[j] + carry;
            const int q = int(div_round(v, total));
            out.taps[size_t(i) * width + j] = int16_t(q);
            carry = v - q * total;
        }
        out.pos[i] = r.pos;
    }
    return true;
}

bool scale_output_size(int iw, int ih, int w, int h, int foar, int fdb, int &ow, int &oh)
{
    auto rescale = [](i64 a, i64 b, i64 c) { return (a * b + c / 2) / c; };   // round half up, positives
    // (sizes far outside anything a frame can be: refused here, before -w, w * fdb or the casts below can overflow)
    constexpr int kLimit = 1 << 20;
    if (iw < 1 || ih < 1 || iw > kLimit || ih > kLimit || w < -kLimit || w > kLimit || h < -kLimit || h > kLimit || fdb > kLimit) return false;
    if (w == 0) w = iw;
    if (h == 0) h = ih;
    const int fw = w < -1 ? -w : 1, fh = h < -1 ? -h : 1;
    if (w < 0 && h < 0) { w = iw; h = ih; }
    if (w < 0) w = int(rescale(h, iw, i64(ih) * fw) * fw);
    if (h < 0) h = int(rescale(w, ih, i64(iw) * fh) * fh);
    if (foar) {
        // libavfilter scale_eval.c ff_scale_adjust_dimensions(): the aspect-derived sizes are first rounded to
        // the NEAREST multiple of force_divisible_by, then min / max'ed against the requested ones, then
        // rounded down (decrease) or up (increase): 720x1280, 320:180, decrease, fdb 2 -> 102x180.
        fdb = std::max(fdb, 1);
        const int tw = int(rescale(h, iw, i64(ih) * fdb)) * fdb, th = int(rescale(w, ih, i64(iw) * fdb)) * fdb;
        if (foar == 1) {
            w = std::min(tw, w) / fdb * fdb;
            h = std::min(th, h) / fdb * fdb;
        } else {
            w = (std::max(tw, w) + fdb - 1) / fdb * fdb;
            h = (std::max(th, h) + fdb - 1) / fdb * fdb;
        }
    }
    if (w <= 0 || h <= 0) return false;
    ow = w; oh = h;
    return true;
}

int crop_origin(int in, int out, double value, bool has_value, int sub)
{
    const double v = has_value ? value : (in - out) / 2.0;
    int x = int(std::lrint(v));                 // round half to even, as lrint does
    x = std::max(x, 0);
    if (x + out > in) x = in - out;
    return x & ~(sub - 1);                      // the link's chroma grid: 2 on yuv420p, 1 (none) on RGB
}

int check_geom(const tts_geom &g)
{
    bool (*even)(int) = [](int v) { return (v & 1) == 0; };
    if (g.in_w < 1 || g.in_h < 1 || g.in_w > 16384 || g.in_h > 16384) return TTS_ERR_INVALID_INPUT;
    if (g.pre_w < 1 || g.pre_h < 1 || g.pre_x < 0 || g.pre_y < 0 ||
        g.pre_x + g.pre_w > g.in_w || g.pre_y + g.pre_h > g.in_h) return TTS_ERR_INVALID_INPUT;
    const bool has_pre = g.pre_x || g.pre_y || g.pre_w != g.in_w || g.pre_h != g.in_h;
    if (g.in_fmt < TTS_FMT_YUV420P || g.in_fmt > TTS_FMT_BGRA) return TTS_ERR_INVALID_INPUT;
    if (g.in_fmt != TTS_FMT_YUV420P) {
        // a crop in front of the conversion works on RGB pixels (any offset); rgb24ToUV_half_c reads pixel pairs
        const bool canvas = g.rgb_stage == TTS_RGB_AT_END;      // (the conversion reads a padded RGB canvas: its width is checked below)
        // (sc_range TO_FULL on a chain fed with RGB: the converting context itself writes yuvj420p -- the range step on its 15-bit rows;
        // out_range: a range pass of its own over the finished frame, as behind any other chain)
        if (g.out_fmt != TTS_FMT_YUV420P || g.sc_range == TTS_RANGE_TO_LIMITED || (!canvas && (g.pre_w & 1) && rgb_in_half_chroma(g.pre_w, g.sc_w))) return TTS_ERR_UNSUPPORTED;
        if (!canvas && bgr24_special_converter(g) && (g.pre_w & 1)) return TTS_ERR_UNSUPPORTED;   // (already covered: the same size is never an enlargement)
    } else if (has_pre && !(even(g.pre_x) && even(g.pre_y) && even(g.pre_w) && even(g.pre_h))) return TTS_ERR_INVALID_INPUT;
    if (g.sc_w < 1 || g.sc_h < 1 || g.sc_w > 16384 || g.sc_h > 16384) return TTS_ERR_INVALID_INPUT;
    if (!g.do_scale && (g.sc_w != g.pre_w || g.sc_h != g.pre_h)) return TTS_ERR_INVALID_INPUT;
    if (g.alg < TTS_ALG_BICUBIC || g.alg > TTS_ALG_BILINEAR) return TTS_ERR_INVALID_INPUT;
    if (g.out_range < TTS_RANGE_KEEP || g.out_range > TTS_RANGE_TO_LIMITED) return TTS_ERR_INVALID_INPUT;
    if (g.out_fmt < TTS_FMT_YUV420P || g.out_fmt > TTS_FMT_BGRA) return TTS_ERR_INVALID_INPUT;
    // (fed with packed RGB: TTS_RGB_AT_END marks a pad on RGB pixels in front of the conversion pass: no scale, nothing else behind it)
    const bool rgb_pad = g.in_fmt != TTS_FMT_YUV420P && g.rgb_stage == TTS_RGB_AT_END;
    if (rgb_pad && (!g.do_pad || g.do_scale || g.sc_w != g.pre_w || g.sc_h != g.pre_h || g.crop_x || g.crop_y || g.crop_w != g.sc_w || g.crop_h != g.sc_h)) return TTS_ERR_INVALID_INPUT;
    if (rgb_pad && (g.out_w & 1)) return TTS_ERR_UNSUPPORTED;      // (the pair-summing chroma reader / bgr24's own converter over an odd canvas)
    // (sc_range: a range step inside the converting context -- planar chains: the scale's; it needs one that resizes, and yuv out)
    if (g.sc_range < TTS_RANGE_KEEP || g.sc_range > TTS_RANGE_TO_LIMITED) return TTS_ERR_INVALID_INPUT;
    if (g.sc_range != TTS_RANGE_KEEP && (g.out_fmt != TTS_FMT_YUV420P || (g.in_fmt == TTS_FMT_YUV420P && !g.do_scale))) return TTS_ERR_INVALID_INPUT;
    if (!rgb_pad && ((g.out_fmt == TTS_FMT_YUV420P) != (g.rgb_stage == TTS_RGB_NONE) || g.rgb_stage < 0 || g.rgb_stage > TTS_RGB_AT_END)) return TTS_ERR_INVALID_INPUT;
    // packed RGB from a full-range link (`format=yuvj420p,format=rgb24`): the range pass, then a conversion pass of its own
    if (g.out_fmt != TTS_FMT_YUV420P && g.out_range != TTS_RANGE_KEEP && !(g.out_range == TTS_RANGE_TO_FULL && g.rgb_stage == TTS_RGB_AT_END)) return TTS_ERR_UNSUPPORTED;
    if (g.fill_rgb & ~0xffffff) return TTS_ERR_INVALID_INPUT;
    // crop and pad behind an RGB-writing scale work on RGB pixels: no chroma grid
    if (g.rgb_stage == TTS_RGB_AT_SCALE || rgb_pad) even = [](int) { return true; };
    if (g.crop_w < 1 || g.crop_h < 1 || g.crop_x < 0 || g.crop_y < 0 ||
        g.crop_x + g.crop_w > g.sc_w || g.crop_y + g.crop_h > g.sc_h) return TTS_ERR_INVALID_INPUT;
    const bool has_crop = g.crop_x || g.crop_y || g.crop_w != g.sc_w || g.crop_h != g.sc_h;
    if (has_crop && !(even(g.crop_x) && even(g.crop_y) && even(g.crop_w) && even(g.crop_h))) return TTS_ERR_INVALID_INPUT;
    if (g.out_w > 16384 || g.out_h > 16384) return TTS_ERR_INVALID_INPUT;     // (a pad canvas: the same limit as every other size here)
    if (g.do_pad) {
        if (!(even(g.pad_x) && even(g.pad_y) && even(g.out_w) && even(g.out_h))) return TTS_ERR_INVALID_INPUT;
        // vf_pad.c config_input(): "Input area ... not within the padded area" once the canvas has been rounded down to the chroma
        // grid (an odd picture that fits keeps all its rows; its last COLUMN goes to the fill when the right border is drawn)
        if (g.pad_x < 0 || g.pad_y < 0 || g.pad_x + g.crop_w > g.out_w || g.pad_y + g.crop_h > g.out_h) return TTS_ERR_INVALID_INPUT;
        if ((g.fill_y | g.fill_u | g.fill_v) & ~255) return TTS_ERR_INVALID_INPUT;
    } else if (g.out_w != g.crop_w || g.out_h != g.crop_h) {
        return TTS_ERR_INVALID_INPUT;
    }
    return TTS_OK;
}

void build_range_lut(int out_range, bool chroma, unsigned char lut[256])
{
    for (int p = 0; p < 256; ++p) {
        int v = p << 7;
        if (out_range == TTS_RANGE_TO_FULL)
            v = chroma ? (std::min(v, 30775) * 4663 - 9289992) >> 12 : (std::min(v, 30189) * 19077 - 39057361) >> 14;
        else if (out_range == TTS_RANGE_TO_LIMITED)
            v = chroma ? (v * 1799 + 4081085) >> 11 : (v * 14071 + 33561947) >> 14;
        v = (v + 64) >> 7;
        lut[p] = static_cast<unsigned char>(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

// The same step as clip_u8((p * mul + add) >> shift): p << 7 folds into the multiplier, and
// floor((floor(X / 2^a) + 64) / 128) == floor((X + 64 * 2^a) / 2^(a + 7)); the input clamp of the *ToJpeg forms
// only acts where the result is 255 already.  Checked against the table for every byte; false = no such form.
bool range_affine(int out_range, bool chroma, int &mul, int &add, int &shift)
{
    if (out_range == TTS_RANGE_TO_FULL) {
        if (chroma) { mul = 128 * 4663; add = -9289992 + (64 << 12); shift = 19; }
        else        { mul = 128 * 19077; add = -39057361 + (64 << 14); shift = 21; }
    } else if (out_range == TTS_RANGE_TO_LIMITED) {
        if (chroma) { mul = 128 * 1799; add = 4081085 + (64 << 11); shift = 18; }
        else        { mul = 128 * 14071; add = 33561947 + (64 << 14); shift = 21; }
    } else {
        return false;
    }
    unsigned char lut[256];
    build_range_lut(out_range, chroma, lut);
    for (int p = 0; p < 256; ++p) {
        int v = (p * mul + add) >> shift;
        v = v < 0 ? 0 : v > 255 ? 255 : v;
        if (v != lut[p]) return false;
    }
    return true;
}

}  // namespace tts
