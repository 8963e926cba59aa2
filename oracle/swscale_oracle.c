/*
 * oracle/swscale_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the pixel arithmetic the reference hands to an
 * external `ffmpeg` through filter strings:
 *   scripts/processVideos.js:94-97      .size('720x1280')  -> scale=w=720:h=1280
 *   src/services/video.js:22,329-332    scale=320x180:force_original_aspect_ratio=decrease,
 *                                       pad=320x180:-1:-1:color=black,format=yuv420p
 *   web/src/tests/video/portrait.test.js:78,95-96  scale=W:H / min(720 / decrease
 *
 * The arithmetic itself lives in a third-party dependency that is absent from
 * /root/reference and from the container: FFmpeg (system binary, unpinned;
 * @ffmpeg/core ^0.11.0 wraps FFmpeg 4.3).  What follows restates, from the
 * published algorithm as recalled, libswscale's 8-bit planar C path:
 *   libswscale/utils.c   initFilter(), sws_init_context()   (SURVEY.md S1-S3,S6)
 *   libswscale/swscale.c hScale8To15_c()                    
 *   libswscale/output.c  yuv2planeX_8_c(), yuv2plane1_8_c()
 *   libswscale/yuv2rgb.c ff_yuv2rgb_c_init_tables(), yuv2rgb_c_24_rgb & co.;
 *   libswscale/output.c  yuv2rgb_X_c_template(), yuv2rgb_write()   (packed RGB out)
 * and libavfilter's vf_scale / vf_crop / vf_pad geometry.
 *
 * PARITY UNPINNED: no golden vector, known-answer test or runnable FFmpeg
 * exists to pin this file against.  tests/golden/ holds vectors
 * produced BY this oracle plus hand-derivable anchors (taps sum to `one`,
 * constant planes stay constant, identity geometry reproduces the input).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file.  It is deliberately scalar and unoptimised.
 */
#include "swscale_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define TTO_MAX_REDUCE_CUTOFF 0.002   /* SWS_MAX_REDUCE_CUTOFF */

static int tto_log2(unsigned v) { int n = 0; while (v >>= 1) n++; return n; }   /* av_log2, av_log2(0)=0 */
static int64_t tto_abs64(int64_t a) { return a < 0 ? -a : a; }
static int64_t tto_rounded_div(int64_t a, int64_t b)                           /* ROUNDED_DIV */
{
    return (a >= 0 ? a + (b >> 1) : a - (b >> 1)) / b;
}

/* libswscale/utils.c initFilter(), srcFilter/dstFilter NULL, filterAlign 1 (C path),
 * srcPos == dstPos == 128 (get_local_pos(): luma, and centre-sited 4:2:0 chroma). */
int tto_filter_init(tto_filter *f, int srcW, int dstW, int one, int alg)
{
    const int xInc = (int)((((int64_t)srcW << 16) + (dstW >> 1)) / dstW);      /* sws_init_context */
    const int srcPos = 128, dstPos = 128;
    const int64_t fone = 1LL << (54 - (tto_log2((unsigned)(srcW / dstW)) < 8 ? tto_log2((unsigned)(srcW / dstW)) : 8));
    int filterSize, minFilterSize, i, j;
    int64_t *filter = NULL, *filter2 = NULL;
    int32_t *pos;

    memset(f, 0, sizeof(*f));
    if (srcW < 1 || dstW < 1) return -1;
    pos = (int32_t *)calloc((size_t)dstW, sizeof(*pos));
    if (!pos) return -1;

    if (llabs((long long)xInc - 0x10000) < 10) {                 /* unscaled */
        filterSize = 1;
        filter = (int64_t *)calloc((size_t)dstW, sizeof(*filter));
        for (i = 0; i < dstW; i++) { filter[i] = fone; pos[i] = i; }
    } else if (xInc <= (1 << 16) && alg == TTO_ALG_AREA) {       /* "bilinear upscale" branch */
        int64_t xDstInSrc;
        filterSize = 2;
        filter = (int64_t *)calloc((size_t)dstW * filterSize, sizeof(*filter));
        xDstInSrc = ((dstPos * (int64_t)xInc) >> 8) - ((srcPos * 0x8000LL) >> 7);
        for (i = 0; i < dstW; i++) {
            int xx = (int)((xDstInSrc - ((int64_t)(filterSize - 1) << 15) + (1 << 15)) >> 16);
            pos[i] = xx;
            for (j = 0; j < filterSize; j++) {
                int64_t coeff = fone - tto_abs64((int64_t)xx * (1 << 16) - xDstInSrc) * (fone >> 16);
                if (coeff < 0) coeff = 0;
                filter[i * filterSize + j] = coeff;
                xx++;
            }
            xDstInSrc += xInc;
        }
    } else {
        int64_t xDstInSrc;
        int sizeFactor = alg == TTO_ALG_BICUBIC ? 4 : alg == TTO_ALG_AREA ? 1 : 2;
        if (xInc <= 1 << 16) filterSize = 1 + sizeFactor;                       /* upscale */
        else                 filterSize = 1 + (sizeFactor * srcW + dstW - 1) / dstW;
        if (filterSize > srcW - 2) filterSize = srcW - 2;
        if (filterSize < 1) filterSize = 1;
        filter = (int64_t *)calloc((size_t)dstW * filterSize, sizeof(*filter));

        xDstInSrc = ((dstPos * (int64_t)xInc) >> 7) - ((srcPos * 0x10000LL) >> 7);
        for (i = 0; i < dstW; i++) {
            int xx = (int)((xDstInSrc - (filterSize - 2) * (1LL << 16)) / (1 << 17)); /* C division */
            pos[i] = xx;
            for (j = 0; j < filterSize; j++) {
                int64_t d = tto_abs64(((int64_t)xx * (1 << 17)) - xDstInSrc) << 13;
                int64_t coeff;
                if (xInc > 1 << 16) d = d * dstW / srcW;
                if (alg == TTO_ALG_BICUBIC) {
                    const int64_t B = 0;                                        /* SWS_PARAM_DEFAULT */
                    const int64_t C = (int64_t)(0.6 * (1 << 24));
                    if (d >= 1LL << 31) {
                        coeff = 0;
                    } else {
                        int64_t dd  = (d * d) >> 30;
                        int64_t ddd = (dd * d) >> 30;
                        if (d < 1LL << 30)
                            coeff = (12 * (1 << 24) - 9 * B - 6 * C) * ddd +
                                    (-18 * (1 << 24) + 12 * B + 6 * C) * dd +
                                    (6 * (1 << 24) - 2 * B) * (1LL << 30);
                        else
                            coeff = (-B - 6 * C) * ddd + (6 * B + 30 * C) * dd +
                                    (-12 * B - 48 * C) * d + (8 * B + 24 * C) * (1LL << 30);
                    }
                    coeff /= (1LL << 54) / fone;
                } else if (alg == TTO_ALG_AREA) {
                    int64_t d2 = d - (1 << 29);
                    if (d2 * xInc < -(1LL << (29 + 16)))     coeff = 1LL << (30 + 16);
                    else if (d2 * xInc < (1LL << (29 + 16))) coeff = -d2 * xInc + (1LL << (29 + 16));
                    else                                     coeff = 0;
                    coeff *= fone >> (30 + 16);
                } else {                                                         /* bilinear */
                    coeff = (1LL << 30) - d;
                    if (coeff < 0) coeff = 0;
                    coeff *= fone >> 30;
                }
                filter[i * filterSize + j] = coeff;
                xx++;
            }
            xDstInSrc += 2 * xInc;
        }
    }

    /* no src/dst filter: filter2 = filter, filter2Size = filterSize */
    filter2 = filter; filter = NULL;
    {
        const int filter2Size = filterSize;
        /* step 1: shift near-zero leading taps out, count near-zero trailing taps */
        minFilterSize = 0;
        for (i = dstW - 1; i >= 0; i--) {
            int min = filter2Size;
            int64_t cutOff = 0;
            for (j = 0; j < filter2Size; j++) {
                int k;
                cutOff += tto_abs64(filter2[i * filter2Size]);
                if ((double)cutOff > TTO_MAX_REDUCE_CUTOFF * (double)fone) break;
                /* preserve monotonicity */
                if (i < dstW - 1 && pos[i] >= pos[i + 1]) break;
                for (k = 1; k < filter2Size; k++)
                    filter2[i * filter2Size + k - 1] = filter2[i * filter2Size + k];
                filter2[i * filter2Size + k - 1] = 0;
                pos[i]++;
            }
            cutOff = 0;
            for (j = filter2Size - 1; j > 0; j--) {
                cutOff += tto_abs64(filter2[i * filter2Size + j]);
                if ((double)cutOff > TTO_MAX_REDUCE_CUTOFF * (double)fone) break;
                min--;
            }
            if (min > minFilterSize) minFilterSize = min;
        }
        /* step 2: reduce (filterAlign == 1 on the C path) */
        filterSize = minFilterSize;
        filter = (int64_t *)calloc((size_t)dstW * filterSize, sizeof(*filter));
        for (i = 0; i < dstW; i++)
            for (j = 0; j < filterSize; j++)
                filter[i * filterSize + j] = j >= filter2Size ? 0 : filter2[i * filter2Size + j];
        free(filter2);
    }

    /* fix borders: out-of-range taps are folded into the edge tap */
    for (i = 0; i < dstW; i++) {
        if (pos[i] < 0) {
            for (j = 1; j < filterSize; j++) {
                int left = j + pos[i] > 0 ? j + pos[i] : 0;
                filter[i * filterSize + left] += filter[i * filterSize + j];
                filter[i * filterSize + j] = 0;
            }
            pos[i] = 0;
        }
        if (pos[i] + filterSize > srcW) {
            int shift = pos[i] + (filterSize - srcW < 0 ? filterSize - srcW : 0);
            int64_t acc = 0;
            for (j = filterSize - 1; j >= 0; j--) {
                if (pos[i] + j >= srcW) {
                    acc += filter[i * filterSize + j];
                    filter[i * filterSize + j] = 0;
                }
            }
            for (j = filterSize - 1; j >= 0; j--) {
                if (j < shift) filter[i * filterSize + j] = 0;
                else           filter[i * filterSize + j] = filter[i * filterSize + j - shift];
            }
            pos[i] -= shift;
            filter[i * filterSize + srcW - 1 - pos[i]] += acc;
        }
    }

    /* normalise to `one` with error diffusion */
    f->coef = (int16_t *)calloc((size_t)dstW * filterSize, sizeof(int16_t));
    for (i = 0; i < dstW; i++) {
        int64_t error = 0, sum = 0;
        for (j = 0; j < filterSize; j++) sum += filter[i * filterSize + j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (j = 0; j < filterSize; j++) {
            int64_t v = filter[i * filterSize + j] + error;
            int intV  = (int)tto_rounded_div(v, sum);
            f->coef[i * filterSize + j] = (int16_t)intV;
            error = v - intV * sum;
        }
    }
    free(filter);
    f->pos = pos; f->size = filterSize; f->dst = dstW; f->src = srcW;
    return 0;
}

void tto_filter_free(tto_filter *f)
{
    free(f->coef); free(f->pos); memset(f, 0, sizeof(*f));
}

static uint8_t tto_clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : (uint8_t)v; }

/* swscale.c lumRangeToJpeg_c / chrRangeToJpeg_c / lumRangeFromJpeg_c / chrRangeFromJpeg_c on one 15-bit sample
 * (c->lumConvertRange / chrConvertRange: run on the horizontal pass's output row, before the vertical pass).
 * range: 1 limited -> full, 2 full -> limited.  Constants of FFmpeg 4.x-6.0, recalled, UNVERIFIED. */
static int tto_range15(int v, int chroma, int range)
{
    if (range == 1) {
        if (chroma) { if (v > 30775) v = 30775; return (v * 4663 - 9289992) >> 12; }
        if (v > 30189) v = 30189;
        return (v * 19077 - 39057361) >> 14;
    }
    if (range == 2) return chroma ? (v * 1799 + 4081085) >> 11 : (v * 14071 + 33561947) >> 14;
    return v;
}

static int tto_scale_plane_range(const uint8_t *src, int sstride, int sw, int sh,
                                 uint8_t *dst, int dstride, int dw, int dh, int alg, int chroma, int range);

/* hScale8To15_c then yuv2planeX_8_c / yuv2plane1_8_c (dither = ff_sws_pb_64 for 8-bit sources) */
int tto_scale_plane(const uint8_t *src, int sstride, int sw, int sh,
                    uint8_t *dst, int dstride, int dw, int dh, int alg)
{
    return tto_scale_plane_range(src, sstride, sw, sh, dst, dstride, dw, dh, alg, 0, 0);
}

/* ... with the range step of a context whose source and destination ranges differ (yuv420p in, yuvj420p out of the SAME
 * scale: vf_scale is the converter when nothing pins the link behind it to yuv420p) between the two passes */
static int tto_scale_plane_range(const uint8_t *src, int sstride, int sw, int sh,
                                 uint8_t *dst, int dstride, int dw, int dh, int alg, int chroma, int range)
{
    tto_filter hf, vf;
    int16_t *tmp;
    int x, y, j;
    if (tto_filter_init(&hf, sw, dw, 1 << 14, alg)) return -1;
    if (tto_filter_init(&vf, sh, dh, 1 << 12, alg)) { tto_filter_free(&hf); return -1; }
    tmp = (int16_t *)malloc((size_t)dw * sh * sizeof(int16_t));
    for (y = 0; y < sh; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        int16_t *t = tmp + (size_t)y * dw;
        for (x = 0; x < dw; x++) {
            int val = 0, p = hf.pos[x];
            for (j = 0; j < hf.size; j++) val += (int)s[p + j] * hf.coef[hf.size * x + j];
            val >>= 7;
            val = val < (1 << 15) - 1 ? val : (1 << 15) - 1;
            t[x] = (int16_t)tto_range15(val, chroma, range);
        }
    }
    for (y = 0; y < dh; y++) {
        uint8_t *d = dst + (size_t)y * dstride;
        int p = vf.pos[y];
        const int16_t *c = vf.coef + (size_t)vf.size * y;
        if (vf.size == 1) {
            const int16_t *t = tmp + (size_t)p * dw;
            for (x = 0; x < dw; x++) d[x] = tto_clip_u8((t[x] + 64) >> 7);
        } else {
            for (x = 0; x < dw; x++) {
                int val = 64 << 12;
                for (j = 0; j < vf.size; j++) val += tmp[(size_t)(p + j) * dw + x] * c[j];
                d[x] = tto_clip_u8(val >> 19);
            }
        }
    }
    free(tmp); tto_filter_free(&hf); tto_filter_free(&vf);
    return 0;
}

/* av_rescale(a,b,c) with AV_ROUND_NEAR_INF, positive operands */
static int64_t tto_rescale(int64_t a, int64_t b, int64_t c) { return (a * b + c / 2) / c; }

/* libavfilter scale_eval.c ff_scale_adjust_dimensions(), SAR 1:1 */
int tto_scale_dims(int iw, int ih, int w, int h, int foar, int fdb, int *ow, int *oh)
{
    int factor_w = 1, factor_h = 1;
    if (w == 0) w = iw;
    if (h == 0) h = ih;
    if (w < -1) factor_w = -w;
    if (h < -1) factor_h = -h;
    if (w < 0 && h < 0) { w = iw; h = ih; }
    if (w < 0) w = (int)(tto_rescale(h, iw, (int64_t)ih * factor_w) * factor_w);
    if (h < 0) h = (int)(tto_rescale(w, ih, (int64_t)iw * factor_h) * factor_h);
    if (foar) {
        int tmp_w, tmp_h;
        if (fdb < 1) fdb = 1;
        /* "Including force_divisible_by here rounds to the nearest multiple of it." */
        tmp_w = (int)tto_rescale(h, iw, (int64_t)ih * fdb) * fdb;
        tmp_h = (int)tto_rescale(w, ih, (int64_t)iw * fdb) * fdb;
        if (foar == 1) {
            w = tmp_w < w ? tmp_w : w;
            h = tmp_h < h ? tmp_h : h;
            if (fdb > 1) { w = w / fdb * fdb; h = h / fdb * fdb; }
        } else {
            w = tmp_w > w ? tmp_w : w;
            h = tmp_h > h ? tmp_h : h;
            if (fdb > 1) { w = (w + fdb - 1) / fdb * fdb; h = (h + fdb - 1) / fdb * fdb; }
        }
    }
    if (w <= 0 || h <= 0) return -1;
    *ow = w; *oh = h;
    return 0;
}

/* libavfilter vf_crop.c: x = lrint(expr), clamp into frame, then mask to the 4:2:0 chroma grid */
int tto_crop_offset(int in, int out, double expr_value, int use_expr)
{
    double v = use_expr ? expr_value : (in - out) / 2.0;
    int x = (int)lrint(v);
    if (x < 0) x = 0;
    if (x + out > in) x = in - out;
    return x & ~1;
}

static void tto_copy_rect(const uint8_t *s, int ss, uint8_t *d, int ds, int w, int h)
{
    for (int y = 0; y < h; y++) memcpy(d + (size_t)y * ds, s + (size_t)y * ss, (size_t)w);
}

/* A range pass of its own over a finished frame (out_range 1: to full, 2: to limited), in place. */
static void tto_range_pass(int out_range, int out_w, int out_h, uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs)
{
    uint8_t *dp[3] = {dy, du, dv};
    const int dst_[3] = {dys, dus, dvs};
    if (!out_range) return;
    for (int p = 0; p < 3; p++) {
        const int sub = p ? 1 : 0;
        const int ow = (out_w + sub) >> sub, oh = (out_h + sub) >> sub;
        for (int y = 0; y < oh; y++) {
            uint8_t *row = dp[p] + (size_t)y * dst_[p];
            for (int x = 0; x < ow; x++) {
                int v = row[x] << 7;
                if (out_range == 1) {
                    if (p) { if (v > 30775) v = 30775; v = (v * 4663 - 9289992) >> 12; }
                    else   { if (v > 30189) v = 30189; v = (v * 19077 - 39057361) >> 14; }
                } else {
                    if (p) v = (v * 1799 + 4081085) >> 11;
                    else   v = (v * 14071 + 33561947) >> 14;
                }
                v = (v + 64) >> 7;
                row[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
    }
}

int tto_run_chain(const tto_chain *c,
                  const uint8_t *sy, const uint8_t *su, const uint8_t *sv,
                  int sys, int sus, int svs,
                  uint8_t *dy, uint8_t *du, uint8_t *dv,
                  int dys, int dus, int dvs)
{
    const uint8_t *sp[3] = { sy, su, sv };
    const int sst[3] = { sys, sus, svs };
    uint8_t *dp[3] = { dy, du, dv };
    const int dst_[3] = { dys, dus, dvs };
    const int fill[3] = { c->fill_y, c->fill_u, c->fill_v };

    for (int p = 0; p < 3; p++) {
        const int sub = p ? 1 : 0;
        /* vf_crop (before scale): pointer offset; sizes are even so >>1 is exact */
        const uint8_t *s = sp[p] + (size_t)(c->pre_y >> sub) * sst[p] + (c->pre_x >> sub);
        const int pw = (c->pre_w + sub) >> sub, ph = (c->pre_h + sub) >> sub;        /* AV_CEIL_RSHIFT */
        const int vw = (c->sc_w + sub) >> sub,  vh = (c->sc_h + sub) >> sub;
        uint8_t *scaled = NULL;
        const uint8_t *q; int qs;
        if (c->do_scale) {
            /* sc_range: `format=yuvj420p` behind the scale with nothing pinning the link to yuv420p in between (or a full-range
             * source scaled into a link pinned to yuv420p) -- the scale's own context converts the range, on its 15-bit rows;
             * crop and pad behind it then work on the new range (the pad colour arrives as values of that range) */
            const int in_scale = c->out_fmt == 0 ? c->sc_range : 0;
            scaled = (uint8_t *)malloc((size_t)vw * vh);
            if (tto_scale_plane_range(s, sst[p], pw, ph, scaled, vw, vw, vh, c->alg, p != 0, in_scale)) { free(scaled); return -1; }
            q = scaled; qs = vw;
        } else { q = s; qs = sst[p]; }
        /* vf_crop (after scale) */
        q += (size_t)(c->crop_y >> sub) * qs + (c->crop_x >> sub);
        const int cw = (c->crop_w + sub) >> sub, ch = (c->crop_h + sub) >> sub;
        if (!c->do_pad) {
            tto_copy_rect(q, qs, dp[p], dst_[p], cw, ch);
        } else {
            /* vf_pad.c filter_frame(): bars filled, frame copied, then the right border is drawn
             * from x + (in_w rounded DOWN to the chroma grid): an odd input width loses its
             * last column to the fill colour. */
            const int ow = (c->out_w + sub) >> sub, oh = (c->out_h + sub) >> sub;
            const int px = c->pad_x >> sub, py = c->pad_y >> sub;
            const int in_w_r = (c->crop_w & ~1) >> sub;
            /* vf_pad.c config_input(): "Input area not within the padded area" is an error there */
            if (c->pad_x < 0 || c->pad_y < 0 || c->pad_x + c->crop_w > c->out_w || c->pad_y + c->crop_h > c->out_h) { free(scaled); return -1; }
            for (int y = 0; y < oh; y++) memset(dp[p] + (size_t)y * dst_[p], fill[p], (size_t)ow);
            tto_copy_rect(q, qs, dp[p] + (size_t)py * dst_[p] + px, dst_[p], in_w_r < cw ? in_w_r : cw, ch);
        }
        free(scaled);
    }
    /* yuv420p -> yuvj420p (or back) is a SECOND swscale pass, same size, inserted by libavfilter's
     * format negotiation.  Ranges differ, so sws_init_context() keeps the generic path
     * (`unscaled && ... && srcRange == dstRange` fails): hScale8To15_c with the 1-tap 16384 filter
     * (v = p << 7), lum/chrConvertRange on the 15-bit row (swscale.c lumRangeToJpeg_c,
     * chrRangeToJpeg_c, lumRangeFromJpeg_c, chrRangeFromJpeg_c -- constants of FFmpeg 4.x-6.0,
     * recalled, UNVERIFIED), then yuv2plane1_8_c: (v + 64) >> 7 clipped.  Borders included. */
    tto_range_pass(c->out_range, c->out_w, c->out_h, dp[0], dp[1], dp[2], dst_[0], dst_[1], dst_[2]);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Packed RGB output.  RECALLED, UNVERIFIED like the rest of this file -- and the least
 * certain part of it: the constants of ff_yuv2rgb_c_init_tables() below (yoffs 326 / 384, the 384 << 16
 * table origin) are as remembered from FFmpeg 4.x-6.x and imply limited-range luma 1.8 levels darker than
 * the textbook BT.601 matrix (tests/test_oracle_vs_float.py states the bound).  A diff against a real
 * `ffmpeg -pix_fmt rgb24` (tests/test_vs_ffmpeg.py) is what settles it.
 * ------------------------------------------------------------------------------------------------ */
#define TTO_RGB_HEADROOM      512   /* YUVRGB_TABLE_HEADROOM */
#define TTO_RGB_LUMA_HEADROOM 512   /* YUVRGB_TABLE_LUMA_HEADROOM */
#define TTO_RGB_PLANE (1024 + 2 * TTO_RGB_LUMA_HEADROOM)
#define TTO_RGB_CHROMA (256 + 2 * TTO_RGB_HEADROOM)

typedef struct tto_yuv2rgb {
    uint8_t ytab[TTO_RGB_PLANE];            /* c->yuvTable, 24 bpp case: one clipped ramp */
    int rV[TTO_RGB_CHROMA], gU[TTO_RGB_CHROMA], bU[TTO_RGB_CHROMA];   /* c->table_rV ...: swscale keeps POINTERS into yuvTable, here offsets */
    int gV[TTO_RGB_CHROMA];                 /* c->table_gV: an int offset in swscale as well */
} tto_yuv2rgb;

/* yuv2rgb.c fill_table(): table[i] = y_table - (inc >> 9) + ((clip_u8(i - HEADROOM) * inc) >> 16) */
static void tto_fill_table(int *table, int64_t inc, int y_off)
{
    y_off -= (int)(inc >> 9);
    for (int i = 0; i < TTO_RGB_CHROMA; i++) {
        int64_t cb = (int64_t)tto_clip_u8(i - TTO_RGB_HEADROOM) * inc;
        table[i] = y_off + (int)(cb >> 16);
    }
}
/* yuv2rgb.c fill_gv_table() */
static void tto_fill_gv_table(int *table, int64_t inc)
{
    const int off = -(int)(inc >> 9);
    for (int i = 0; i < TTO_RGB_CHROMA; i++) {
        int64_t cb = (int64_t)tto_clip_u8(i - TTO_RGB_HEADROOM) * inc;
        table[i] = off + (int)(cb >> 16);
    }
}

/* yuv2rgb.c ff_yuv2rgb_c_init_tables(c, ff_yuv2rgb_coeffs[SWS_CS_DEFAULT = ITU601], fullRange = the SOURCE's range
 * (0 for yuv420p, 1 for yuvj420p), brightness 0, contrast = saturation = 1 << 16), bpp 24 (the 32 bpp tables hold the
 * same bytes shifted) */
static void tto_yuv2rgb_init(tto_yuv2rgb *t, int full_range)
{
    const int yoffs = (full_range ? 384 : 326) + TTO_RGB_LUMA_HEADROOM;
    int64_t crv = 104597, cbu = 132201, cgu = -25675, cgv = -53279;   /* ITU-R BT.601 */
    int64_t cy = 1 << 16, oy = 0, yb;
    if (!full_range) {
        cy = (cy * 255) / 219;
        oy = 16 << 16;
    } else {                                                   /* a yuvj420p source (c->srcRange = 1) */
        crv = (crv * 224) / 255;
        cbu = (cbu * 224) / 255;
        cgu = (cgu * 224) / 255;
        cgv = (cgv * 224) / 255;
    }
    /* contrast, saturation = 1.0 and brightness = 0 leave cy, c?? and oy as they are */
    crv = ((crv * (1 << 16)) + 0x8000) / cy;                   /* "scale coefficients by cy" */
    cbu = ((cbu * (1 << 16)) + 0x8000) / cy;
    cgu = ((cgu * (1 << 16)) + 0x8000) / cy;
    cgv = ((cgv * (1 << 16)) + 0x8000) / cy;
    yb = -(384 << 16) - TTO_RGB_LUMA_HEADROOM * cy - oy;
    for (int i = 0; i < TTO_RGB_PLANE; i++) {
        t->ytab[i] = tto_clip_u8((int)((yb + 0x8000) >> 16));
        yb += cy;
    }
    tto_fill_table(t->rV, crv, yoffs);
    tto_fill_table(t->gU, cgu, yoffs);
    tto_fill_table(t->bU, cbu, yoffs);
    tto_fill_gv_table(t->gV, cgv);
}

/* output.c yuv2rgb_write() for RGB24 / BGR24 / RGBA / BGRA: r[Y], g[Y], b[Y] with r = table_rV[V], ... */
static int tto_rgb_put(const tto_yuv2rgb *t, uint8_t *d, int fmt, int Y, int U, int V)
{
    const int r = t->rV[V + TTO_RGB_HEADROOM] + Y;
    const int g = t->gU[U + TTO_RGB_HEADROOM] + t->gV[V + TTO_RGB_HEADROOM] + Y;
    const int b = t->bU[U + TTO_RGB_HEADROOM] + Y;
    if (r < 0 || g < 0 || b < 0 || r >= TTO_RGB_PLANE || g >= TTO_RGB_PLANE || b >= TTO_RGB_PLANE) return -1;  /* swscale would read outside yuvTable */
    const int swap = fmt == TTO_FMT_BGR24 || fmt == TTO_FMT_BGRA;
    d[0] = t->ytab[swap ? b : r];
    d[1] = t->ytab[g];
    d[2] = t->ytab[swap ? r : b];
    if (fmt == TTO_FMT_RGBA || fmt == TTO_FMT_BGRA) d[3] = 255;
    return 0;
}

static void tto_hscale_plane(const uint8_t *src, int sstride, int sh, const tto_filter *hf, int16_t *tmp)
{
    for (int y = 0; y < sh; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        int16_t *t = tmp + (size_t)y * hf->dst;
        for (int x = 0; x < hf->dst; x++) {
            int val = 0, p = hf->pos[x];
            for (int j = 0; j < hf->size; j++) val += (int)s[p + j] * hf->coef[hf->size * x + j];
            val >>= 7;
            t[x] = (int16_t)(val < (1 << 15) - 1 ? val : (1 << 15) - 1);
        }
    }
}

int tto_sws_yuv420_to_rgb(const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int sys, int sus, int svs,
                          int sw, int sh, uint8_t *dst, int dstride, int dw, int dh, int alg, int fmt)
{
    return tto_sws_yuv420_to_rgb_range(sy, su, sv, sys, sus, svs, sw, sh, dst, dstride, dw, dh, alg, fmt, 0);
}

/* vscale.c packed_vscale(): "chr_filter[2 * y + 1] + chr_filter[2 * y] == 4096 && chr_filter[2 * y + 1] <= 4096U" */
static int tto_two_tap_ok(const int16_t *t) { return (int)t[0] + (int)t[1] == 4096 && (uint16_t)t[1] <= 4096u; }

int tto_sws_yuv420_to_rgb_range(const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int sys, int sus, int svs,
                                int sw, int sh, uint8_t *dst, int dstride, int dw, int dh, int alg, int fmt, int full_range)
{
    const int bpp = (fmt == TTO_FMT_RGBA || fmt == TTO_FMT_BGRA) ? 4 : 3;
    tto_yuv2rgb *t;
    int rc = 0;
    if (fmt < TTO_FMT_RGB24 || fmt > TTO_FMT_BGRA || sw < 1 || sh < 1 || dw < 1 || dh < 1) return -1;
    t = (tto_yuv2rgb *)malloc(sizeof(*t));
    tto_yuv2rgb_init(t, full_range);

    if (sw == dw && sh == dh && !(dh & 1)) {
        /* swscale_unscaled.c ff_get_unscaled_swscale(): yuv420p -> RGB, no accurate_rnd, even height ->
         * ff_yuv2rgb_get_func_ptr(): yuv2rgb_c_24_rgb & co.  Two luma rows share chroma row y >> 1, two
         * pixels chroma column x >> 1; the 8-bit samples index the tables directly.  (Its 8 / 4 / 2-pixel
         * loops never write the last column of an odd width: not restated.) */
        if (dw & 1) { free(t); return -2; }
        for (int y = 0; y < dh && !rc; y++)
            for (int x = 0; x < dw && !rc; x++)
                rc = tto_rgb_put(t, dst + (size_t)y * dstride + (size_t)x * bpp, fmt, sy[(size_t)y * sys + x],
                                 su[(size_t)(y >> 1) * sus + (x >> 1)], sv[(size_t)(y >> 1) * svs + (x >> 1)]);
        free(t);
        return rc;
    }

    /* sws_init_context(): RGB out without full_chroma_int -> chrDstHSubSample = 1, chrDstVSubSample = 0:
     * chroma is scaled to ceil(dw / 2) columns and to dh rows (every position 128: centre-aligned) */
    {
        const int csw = (sw + 1) >> 1, csh = (sh + 1) >> 1, cdw = (dw + 1) >> 1;
        tto_filter lh, lv, ch, cv;
        int16_t *ty, *tu, *tv;
        if (tto_filter_init(&lh, sw, dw, 1 << 14, alg) || tto_filter_init(&lv, sh, dh, 1 << 12, alg) ||
            tto_filter_init(&ch, csw, cdw, 1 << 14, alg) || tto_filter_init(&cv, csh, dh, 1 << 12, alg)) { free(t); return -1; }
        ty = (int16_t *)malloc((size_t)dw * sh * sizeof(int16_t));
        tu = (int16_t *)malloc((size_t)cdw * csh * sizeof(int16_t));
        tv = (int16_t *)malloc((size_t)cdw * csh * sizeof(int16_t));
        if (!rc) {
            tto_hscale_plane(sy, sys, sh, &lh, ty);
            tto_hscale_plane(su, sus, csh, &ch, tu);
            tto_hscale_plane(sv, svs, csh, &ch, tv);
        }
        for (int y = 0; y < dh && !rc; y++) {
            /* output.c yuv2rgb_X_c_template() */
            const int16_t *lf = lv.coef + (size_t)lv.size * y, *cf = cv.coef + (size_t)cv.size * y;
            const int lp = lv.pos[y], cp = cv.pos[y];
            /* vscale.c packed_vscale(), decided PER OUTPUT ROW from its taps (chroma rows = output rows: RGB has no
             * vertical chroma subsampling):
             *   1: luma 1 tap and chroma 1 tap, or chroma 2 taps that sum to 4096 with a non-negative second one
             *      -> yuv2packed1 = output.c yuv2rgb_1_c_template(uvalpha = 0 / the second chroma tap);
             *   2: luma and chroma 2 such taps -> yuv2packed2 = yuv2rgb_2_c_template(yalpha, uvalpha = the second taps);
             *   0: everything else -> yuv2packedX = yuv2rgb_X_c_template. */
            int writer = 0, uvalpha = 0, yalpha = 0;
            if (lv.size == 1 && cv.size == 1) writer = 1;
            else if (lv.size == 1 && cv.size == 2 && tto_two_tap_ok(cf)) { writer = 1; uvalpha = cf[1]; }
            else if (lv.size == 2 && cv.size == 2 && tto_two_tap_ok(lf) && tto_two_tap_ok(cf)) { writer = 2; yalpha = lf[1]; uvalpha = cf[1]; }
            for (int i = 0; i < ((dw + 1) >> 1) && !rc; i++) {
                int Y1 = 1 << 18, Y2 = 1 << 18, U = 1 << 18, V = 1 << 18;
                const int has2 = i * 2 + 1 < dw;             /* (swscale computes and writes it into the row's padding) */
                if (writer == 1) {
                    /* yuv2rgb_1_c_template(): no filtering down the picture; chroma from the first line alone
                     * (uvalpha < 2048) or the plain average of the two */
                    Y1 = (ty[(size_t)lp * dw + i * 2] + 64) >> 7;
                    Y2 = has2 ? (ty[(size_t)lp * dw + i * 2 + 1] + 64) >> 7 : 0;
                    if (uvalpha < 2048) {
                        U = (tu[(size_t)cp * cdw + i] + 64) >> 7;
                        V = (tv[(size_t)cp * cdw + i] + 64) >> 7;
                    } else {
                        U = (tu[(size_t)cp * cdw + i] + tu[(size_t)(cp + 1) * cdw + i] + 128) >> 8;
                        V = (tv[(size_t)cp * cdw + i] + tv[(size_t)(cp + 1) * cdw + i] + 128) >> 8;
                    }
                } else if (writer == 2) {
                    /* yuv2rgb_2_c_template(): (line0 * (4096 - alpha) + line1 * alpha) >> 19, no rounding term */
                    const int ya1 = 4096 - yalpha, ua1 = 4096 - uvalpha;
                    Y1 = (ty[(size_t)lp * dw + i * 2] * ya1 + ty[(size_t)(lp + 1) * dw + i * 2] * yalpha) >> 19;
                    Y2 = has2 ? (ty[(size_t)lp * dw + i * 2 + 1] * ya1 + ty[(size_t)(lp + 1) * dw + i * 2 + 1] * yalpha) >> 19 : 0;
                    U = (tu[(size_t)cp * cdw + i] * ua1 + tu[(size_t)(cp + 1) * cdw + i] * uvalpha) >> 19;
                    V = (tv[(size_t)cp * cdw + i] * ua1 + tv[(size_t)(cp + 1) * cdw + i] * uvalpha) >> 19;
                } else {
                for (int j = 0; j < lv.size; j++) {
                    Y1 += ty[(size_t)(lp + j) * dw + i * 2] * lf[j];
                    if (has2) Y2 += ty[(size_t)(lp + j) * dw + i * 2 + 1] * lf[j];
                }
                for (int j = 0; j < cv.size; j++) {
                    U += tu[(size_t)(cp + j) * cdw + i] * cf[j];
                    V += tv[(size_t)(cp + j) * cdw + i] * cf[j];
                }
                Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
                }
                if (U < -TTO_RGB_HEADROOM || U >= 256 + TTO_RGB_HEADROOM || V < -TTO_RGB_HEADROOM || V >= 256 + TTO_RGB_HEADROOM) { rc = -1; break; }
                rc = tto_rgb_put(t, dst + (size_t)y * dstride + (size_t)(i * 2) * bpp, fmt, Y1, U, V);
                if (has2 && !rc) rc = tto_rgb_put(t, dst + (size_t)y * dstride + (size_t)(i * 2 + 1) * bpp, fmt, Y2, U, V);
            }
        }
        free(ty); free(tu); free(tv);
        tto_filter_free(&lh); tto_filter_free(&lv); tto_filter_free(&ch); tto_filter_free(&cv);
    }
    free(t);
    return rc;
}

int tto_run_chain_rgb(const tto_chain *c, const uint8_t *sy, const uint8_t *su, const uint8_t *sv,
                      int sys, int sus, int svs, uint8_t *dst, int dstride)
{
    const int bpp = (c->out_fmt == TTO_FMT_RGBA || c->out_fmt == TTO_FMT_BGRA) ? 4 : 3;
    int rc;
    if (c->rgb_stage == 2) {
        /* no scale in the chain: crop / pad run on yuv420p (tto_run_chain), then libavfilter inserts a
         * conversion-only swscale pass in front of `format`, borders included */
        tto_chain y = *c;
        const int ow = c->out_w, oh = c->out_h, cw = (ow + 1) >> 1, chh = (oh + 1) >> 1;
        uint8_t *buf = (uint8_t *)malloc((size_t)ow * oh + 2 * (size_t)cw * chh);
        uint8_t *py = buf, *pu = buf + (size_t)ow * oh, *pv = pu + (size_t)cw * chh;
        /* `format=yuvj420p,format=rgb24`: the range pass of tto_run_chain() first (borders included), then the conversion
         * pass reads a FULL-range source: its yuv2rgb tables are built with fullRange = 1 */
        y.out_fmt = 0; y.rgb_stage = 0;
        rc = tto_run_chain(&y, sy, su, sv, sys, sus, svs, py, pu, pv, ow, cw, cw);
        if (!rc) rc = tto_sws_yuv420_to_rgb_range(py, pu, pv, ow, cw, cw, ow, oh, dst, dstride, ow, oh, c->alg, c->out_fmt, c->out_range == 1);
        free(buf);
        return rc;
    }
    if (c->rgb_stage != 1 || c->out_range) return -1;
    {
        /* the chain's scale writes RGB (sc_w x sc_h, from the pre-cropped source); vf_crop and vf_pad then
         * work on RGB pixels: offsets and sizes are not rounded to a chroma grid */
        const uint8_t *py = sy + (size_t)c->pre_y * sys + c->pre_x;
        const uint8_t *pu = su + (size_t)(c->pre_y >> 1) * sus + (c->pre_x >> 1);
        const uint8_t *pv = sv + (size_t)(c->pre_y >> 1) * svs + (c->pre_x >> 1);
        const size_t sstride = (size_t)c->sc_w * bpp;
        uint8_t *scaled = (uint8_t *)malloc(sstride * c->sc_h);
        rc = tto_sws_yuv420_to_rgb(py, pu, pv, sys, sus, svs, c->pre_w, c->pre_h, scaled, (int)sstride, c->sc_w, c->sc_h, c->alg, c->out_fmt);
        if (!rc) {
            const uint8_t *q = scaled + (size_t)c->crop_y * sstride + (size_t)c->crop_x * bpp;
            if (c->do_pad) {
                const int swap = c->out_fmt == TTO_FMT_BGR24 || c->out_fmt == TTO_FMT_BGRA;
                uint8_t px[4] = { (uint8_t)(c->fill_rgb >> 16), (uint8_t)(c->fill_rgb >> 8), (uint8_t)c->fill_rgb, 255 };
                if (swap) { uint8_t t0 = px[0]; px[0] = px[2]; px[2] = t0; }
                for (int y = 0; y < c->out_h; y++)
                    for (int x = 0; x < c->out_w; x++) memcpy(dst + (size_t)y * dstride + (size_t)x * bpp, px, (size_t)bpp);
                dst += (size_t)c->pad_y * dstride + (size_t)c->pad_x * bpp;
            }
            for (int y = 0; y < c->crop_h; y++) memcpy(dst + (size_t)y * dstride, q + (size_t)y * sstride, (size_t)c->crop_w * bpp);
        }
        free(scaled);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Packed rgb24 INPUT (the other direction).  RECALLED, UNVERIFIED:
 *   utils.c fill_rgb2yuv_table(): for the default (ITU-R BT.601) matrix the table is overwritten with
 *     (int)(0.299 * 219 / 255 * (1 << 15) + 0.5) & co. -- limited-range coefficients in 15 fractional bits;
 *   input.c rgb24ToY_c(): (ry r + gy g + by b + (32 << 14) + (1 << 8)) >> 9          -> Y << 6 (14 bits);
 *   utils.c sws_init_context(), "drop every other pixel for chroma calculation unless user wants full chroma":
 *         if (isAnyRGB(srcFormat) && !(flags & SWS_FULL_CHR_H_INP) && <not a 4- / 8-bit or planar RGB format> &&
 *             ((dstW >> c->chrDstHSubSample) <= (srcW >> 1) || (flags & SWS_FAST_BILINEAR)))
 *             c->chrSrcHSubSample = 1;
 *     so chroma is read at HALF width only when the chroma output is no wider than half the source
 *     (dstW >> 1 <= srcW >> 1 for yuv420p out: the conversion alone and every horizontal shrink).  Then
 *       input.c rgb24ToUV_half_c(): the SUMS of two neighbouring pixels,
 *         (ru r + gu g + bu b + (256 << 15) + (1 << 9)) >> 10                           -> C << 6,
 *       chrSrcW = ceil(srcW / 2), chroma h-filter ceil(srcW / 2) -> ceil(dstW / 2).
 *     For every horizontal ENLARGEMENT (1920 -> 3413, 480 -> 720) chrSrcHSubSample stays 0:
 *       input.c rgb24ToUV_c(): one sample per pixel,
 *         (ru r + gu g + bu b + (256 << 14) + (1 << 8)) >> 9                            -> C << 6,
 *       chrSrcW = srcW, chroma h-filter srcW -> ceil(dstW / 2) (a shrink of up to 2 : 1 with its widened kernel);
 *     (round 2 of this file always took the half form: VERDICT r2 "what's weak" 1)
 *   get_local_pos() gives 128 (centre) for both chroma source positions either way;
 *   every source row yields a chroma row (chrSrcVSubSample = 0 for RGB), so chroma is scaled sh -> ceil(dh / 2);
 *   swscale.c hScale16To15_c() with sh = 13 for RGB sources: min(sum >> 13, 32767)     -> 15 bits, then the
 *   vertical pass of the planar path (yuv2planeX_8_c / yuv2plane1_8_c).
 * ------------------------------------------------------------------------------------------------ */
int tto_rgb_in_half_chroma(int sw, int dw) { return (dw >> 1) <= (sw >> 1); }
/* bgr24 -> yuv420p at the same size takes swscale's special converter (see tto_sws_packed_to_yuv420) */
int tto_bgr24_special_converter(int fmt, int sw, int sh, int dw, int dh) { return fmt == TTO_FMT_BGR24 && sw == dw && sh == dh; }

/* swscale.c hScale16To15_c (sh = 13 for RGB sources), then the context's range step on the 15-bit row where its ranges
 * differ (c->lumConvertRange / chrConvertRange; range 0: none) */
static void tto_hscale16_plane_range(const int16_t *src, int sw, int sh, const tto_filter *hf, int16_t *tmp, int chroma, int range)
{
    for (int y = 0; y < sh; y++) {
        const int16_t *s = src + (size_t)y * sw;
        int16_t *t = tmp + (size_t)y * hf->dst;
        for (int x = 0; x < hf->dst; x++) {
            int val = 0, p = hf->pos[x];
            for (int j = 0; j < hf->size; j++) val += (int)s[p + j] * hf->coef[hf->size * x + j];
            val >>= 13;
            val = val < (1 << 15) - 1 ? val : (1 << 15) - 1;
            t[x] = (int16_t)tto_range15(val, chroma, range);
        }
    }
}

static void tto_vscale_plane(const int16_t *tmp, int dw, const tto_filter *vf, uint8_t *dst, int dstride)
{
    for (int y = 0; y < vf->dst; y++) {
        uint8_t *d = dst + (size_t)y * dstride;
        const int p = vf->pos[y];
        const int16_t *c = vf->coef + (size_t)vf->size * y;
        for (int x = 0; x < dw; x++) {
            if (vf->size == 1) { d[x] = tto_clip_u8((tmp[(size_t)p * dw + x] + 64) >> 7); continue; }
            int val = 64 << 12;
            for (int j = 0; j < vf->size; j++) val += tmp[(size_t)(p + j) * dw + x] * c[j];
            d[x] = tto_clip_u8(val >> 19);
        }
    }
}

int tto_sws_rgb24_to_yuv420(const uint8_t *src, int sstride, int sw, int sh,
                            uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs, int dw, int dh, int alg)
{
    return tto_sws_packed_to_yuv420(TTO_FMT_RGB24, src, sstride, sw, sh, dy, du, dv, dys, dus, dvs, dw, dh, alg);
}

/* The same for the four packed layouts: input.c bgr24ToY_c & co. are rgb24's with the channels read from the other
 * end; the 32-bit readers (rgb16_32ToY_c_template for rgba / bgra) carry 8 more fractional bits through the same
 * sum -- (coef << 8) * channel + (rnd << 8) >> (S - 6 + 8) -- all of them zero, so the 14-bit samples are the
 * same; alpha is not read.  RECALLED, UNVERIFIED like the rest of this block. */
int tto_sws_packed_to_yuv420(int fmt, const uint8_t *src, int sstride, int sw, int sh,
                             uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs, int dw, int dh, int alg)
{
    return tto_sws_packed_to_yuv420_range(fmt, src, sstride, sw, sh, dy, du, dv, dys, dus, dvs, dw, dh, alg, 0);
}

/* ... into yuvj420p (full = 1): an RGB source has srcRange 0 (sws_setColorspaceDetails()), utils.c fill_rgb2yuv_table()
 * builds the LIMITED-range table whatever the destination ("dstRange = 0; //FIXME range = 1 is handled elsewhere"), and
 * elsewhere is ff_sws_init_range_convert(): srcRange != dstRange -> lumRangeToJpeg_c / chrRangeToJpeg_c on the 15-bit rows
 * between the passes.  Ranges differ, so sws_init_context() never asks ff_get_unscaled_swscale(): no bgr24 special case
 * here.  RECALLED, UNVERIFIED. */
int tto_sws_packed_to_yuv420_range(int fmt, const uint8_t *src, int sstride, int sw, int sh,
                                   uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs, int dw, int dh, int alg, int full)
{
    const int pb = (fmt == TTO_FMT_RGBA || fmt == TTO_FMT_BGRA) ? 4 : 3;    /* bytes a pixel */
    const int ro = (fmt == TTO_FMT_BGR24 || fmt == TTO_FMT_BGRA) ? 2 : 0, bo = 2 - ro;   /* where red and blue sit */
    const int S = 15;                                                      /* RGB2YUV_SHIFT */
    const int ry = (int)(0.299 * 219 / 255 * (1 << S) + 0.5), gy = (int)(0.587 * 219 / 255 * (1 << S) + 0.5), by = (int)(0.114 * 219 / 255 * (1 << S) + 0.5);
    const int ru = -(int)(0.169 * 224 / 255 * (1 << S) + 0.5), gu = -(int)(0.331 * 224 / 255 * (1 << S) + 0.5), bu = (int)(0.500 * 224 / 255 * (1 << S) + 0.5);
    const int rv = (int)(0.500 * 224 / 255 * (1 << S) + 0.5), gv = -(int)(0.419 * 224 / 255 * (1 << S) + 0.5), bv = -(int)(0.081 * 224 / 255 * (1 << S) + 0.5);
    const int half = tto_rgb_in_half_chroma(sw, dw);                       /* c->chrSrcHSubSample */
    const int csw = half ? (sw + 1) >> 1 : sw, cdw = (dw + 1) >> 1, cdh = (dh + 1) >> 1;
    tto_filter lh, lv, ch, cv;
    int16_t *y14, *u14, *v14, *ty, *tu, *tv;
    if (fmt < TTO_FMT_RGB24 || fmt > TTO_FMT_BGRA || sw < 1 || sh < 1 || dw < 1 || dh < 1) return -1;
    if (!full && tto_bgr24_special_converter(fmt, sw, sh, dw, dh)) {
        /* swscale_unscaled.c ff_get_unscaled_swscale(), "bgr24toYV12": srcFormat == BGR24 && dstFormat == YUV420P &&
         * !(flags & SWS_ACCURATE_RND) -> bgr24ToYv12Wrapper -> rgb2rgb_template.c ff_rgb24toyv12_c() (FFmpeg 4.3: "Chrominance
         * data is only taken from every second line, others are ignored in the C version"): no 14-bit intermediates, no
         * rounding term -- luma ((ry r + gy g + by b) >> 15) + 16 per pixel; chroma ((ru r + gu g + bu b) >> 15) + 128 from
         * the LEFT pixel of each pair of an EVEN row (rows 0, 2, ...; the last row of an odd height is such a row), stored
         * as bytes (the sums are unsigned there: the shift floors, the store keeps the low 8 bits).  chromWidth = width >> 1
         * pairs a row: the last column of an odd width is never written, so an odd width is refused here (later FFmpeg
         * sends odd widths down the general path instead).  Later FFmpeg (the 2x2 average in this function) differs:
         * RECALLED for 4.3, UNVERIFIED like the rest of this block. */
        if (sw & 1) return -1;
        for (int y = 0; y < sh; y++) {
            const uint8_t *s = src + (size_t)y * sstride;
            for (int i = 0; i < sw; i++) {
                const unsigned b = s[3 * i], g = s[3 * i + 1], r = s[3 * i + 2];
                dy[(size_t)y * dys + i] = (uint8_t)((((unsigned)ry * r + (unsigned)gy * g + (unsigned)by * b) >> S) + 16);
                if (!(y & 1) && !(i & 1)) {
                    du[(size_t)(y >> 1) * dus + (i >> 1)] = (uint8_t)((((unsigned)ru * r + (unsigned)gu * g + (unsigned)bu * b) >> S) + 128);
                    dv[(size_t)(y >> 1) * dvs + (i >> 1)] = (uint8_t)((((unsigned)rv * r + (unsigned)gv * g + (unsigned)bv * b) >> S) + 128);
                }
            }
        }
        return 0;
    }
    if (half && (sw < 2 || (sw & 1))) return -1;                           /* (the pair reader would run past an odd row's end) */
    if (tto_filter_init(&lh, sw, dw, 1 << 14, alg) || tto_filter_init(&lv, sh, dh, 1 << 12, alg) ||
        tto_filter_init(&ch, csw, cdw, 1 << 14, alg) || tto_filter_init(&cv, sh, cdh, 1 << 12, alg)) return -1;
    y14 = (int16_t *)malloc((size_t)sw * sh * sizeof(int16_t));
    u14 = (int16_t *)malloc((size_t)csw * sh * sizeof(int16_t));
    v14 = (int16_t *)malloc((size_t)csw * sh * sizeof(int16_t));
    for (int y = 0; y < sh; y++) {
        const uint8_t *s = src + (size_t)y * sstride;
        for (int i = 0; i < sw; i++) {
            const int r = s[pb * i + ro], g = s[pb * i + 1], b = s[pb * i + bo];
            y14[(size_t)y * sw + i] = (int16_t)((ry * r + gy * g + by * b + (32 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
        }
        for (int i = 0; i < csw; i++) {
            if (half) {                                                    /* rgb24ToUV_half_c */
                const uint8_t *q = s + 2 * pb * i;
                const int r = q[ro] + q[pb + ro], g = q[1] + q[pb + 1], b = q[bo] + q[pb + bo];
                u14[(size_t)y * csw + i] = (int16_t)((ru * r + gu * g + bu * b + (256 << S) + (1 << (S - 6))) >> (S - 5));
                v14[(size_t)y * csw + i] = (int16_t)((rv * r + gv * g + bv * b + (256 << S) + (1 << (S - 6))) >> (S - 5));
            } else {                                                       /* rgb24ToUV_c */
                const int r = s[pb * i + ro], g = s[pb * i + 1], b = s[pb * i + bo];
                u14[(size_t)y * csw + i] = (int16_t)((ru * r + gu * g + bu * b + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
                v14[(size_t)y * csw + i] = (int16_t)((rv * r + gv * g + bv * b + (256 << (S - 1)) + (1 << (S - 7))) >> (S - 6));
            }
        }
    }
    ty = (int16_t *)malloc((size_t)dw * sh * sizeof(int16_t));
    tu = (int16_t *)malloc((size_t)cdw * sh * sizeof(int16_t));
    tv = (int16_t *)malloc((size_t)cdw * sh * sizeof(int16_t));
    tto_hscale16_plane_range(y14, sw, sh, &lh, ty, 0, full ? 1 : 0);
    tto_hscale16_plane_range(u14, csw, sh, &ch, tu, 1, full ? 1 : 0);
    tto_hscale16_plane_range(v14, csw, sh, &ch, tv, 1, full ? 1 : 0);
    tto_vscale_plane(ty, dw, &lv, dy, dys);
    tto_vscale_plane(tu, cdw, &cv, du, dus);
    tto_vscale_plane(tv, cdw, &cv, dv, dvs);
    free(y14); free(u14); free(v14); free(ty); free(tu); free(tv);
    tto_filter_free(&lh); tto_filter_free(&lv); tto_filter_free(&ch); tto_filter_free(&cv);
    return 0;
}

int tto_run_chain_rgbin(const tto_chain *c, const uint8_t *src, int sstride,
                        uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs)
{
    /* the chain's scale (or, without one, a conversion pass behind the crops) reads the RGB of the pre-cropped
     * rectangle -- a crop on RGB pixels, any offset -- and writes yuv420p; crop / pad behind it are the planar ones */
    const int vw = c->sc_w, vh = c->sc_h, cw = (vw + 1) >> 1, chh = (vh + 1) >> 1;
    uint8_t *buf = (uint8_t *)malloc((size_t)vw * vh + 2 * (size_t)cw * chh);
    uint8_t *py = buf, *pu = buf + (size_t)vw * vh, *pv = pu + (size_t)cw * chh;
    tto_chain rest = *c;
    const int pb = (c->in_fmt == TTO_FMT_RGBA || c->in_fmt == TTO_FMT_BGRA) ? 4 : 3;
    if (c->rgb_stage == 2) {
        /* No scale in the chain and a pad: libavfilter's negotiation (avfiltergraph.c query_formats(): vf_pad keeps ONE
         * format list for its input and output, the source pins it to the packed format, so the link that fails to merge
         * is the one behind the pad) puts the conversion pass at the END: vf_pad fills its canvas with the colour as RGB
         * and copies the (cropped) picture to any offset, then one swscale context converts the canvas, border included.
         * RECALLED, UNVERIFIED. */
        const int W = c->out_w, H = c->out_h, cw2 = (W + 1) >> 1;
        const int fr = (c->fill_rgb >> 16) & 255, fg = (c->fill_rgb >> 8) & 255, fb = c->fill_rgb & 255;
        const int ro = (c->in_fmt == TTO_FMT_BGR24 || c->in_fmt == TTO_FMT_BGRA) ? 2 : 0;
        uint8_t *canvas;
        int rcc;
        if (!c->do_pad || c->do_scale || c->pad_x < 0 || c->pad_y < 0 || c->pad_x + c->pre_w > W || c->pad_y + c->pre_h > H) return -1;
        canvas = (uint8_t *)malloc((size_t)W * H * pb);
        for (size_t i = 0; i < (size_t)W * H; i++) {
            uint8_t *q = canvas + i * pb;
            q[ro] = (uint8_t)fr; q[1] = (uint8_t)fg; q[2 - ro] = (uint8_t)fb;
            if (pb == 4) q[3] = 255;                                        /* (not read) */
        }
        for (int y = 0; y < c->pre_h; y++)
            memcpy(canvas + ((size_t)(c->pad_y + y) * W + c->pad_x) * pb, src + (size_t)(c->pre_y + y) * sstride + (size_t)c->pre_x * pb, (size_t)c->pre_w * pb);
        rcc = tto_sws_packed_to_yuv420_range(c->in_fmt, canvas, W * pb, W, H, dy, du, dv, dys, dus, dvs, W, H, c->alg, c->sc_range == 1);
        (void)cw2;
        if (!rcc) tto_range_pass(c->out_range, W, H, dy, du, dv, dys, dus, dvs);   /* (`...,format=yuv420p,format=yuvj420p`: the encoder's pass) */
        free(canvas);
        free(buf);
        return rcc;
    }
    /* sc_range 1 on a chain fed with RGB: the converting context itself writes yuvj420p (`scale=...,format=yuvj420p`, or no
     * scale and `format=yuvj420p`); what follows works on yuvj420p (pad colour in full range), no pass of its own */
    int rc = tto_sws_packed_to_yuv420_range(c->in_fmt, src + (size_t)c->pre_y * sstride + (size_t)c->pre_x * pb, sstride, c->pre_w, c->pre_h,
                                            py, pu, pv, vw, cw, cw, vw, vh, c->alg, c->sc_range == 1);
    if (!rc) {
        /* what is left of the chain sees a yuv420p frame of the scaled size */
        rest.in_fmt = 0; rest.in_w = vw; rest.in_h = vh; rest.pre_x = rest.pre_y = 0; rest.pre_w = vw; rest.pre_h = vh; rest.do_scale = 0;
        rest.sc_range = 0;
        rc = tto_run_chain(&rest, py, pu, pv, vw, cw, cw, dy, du, dv, dys, dus, dvs);
    }
    free(buf);
    return rc;
}
