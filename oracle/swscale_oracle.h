/*
 * oracle/swscale_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the third-party pixel engine behind the reference's
 * filter strings (libswscale 8-bit planar C path + libavfilter scale/crop/pad
 * geometry).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this.  The shipped library (libttscale.so) never links it.
 *
 * PARITY UNPINNED: FFmpeg is not in /root/reference nor in the container and
 * the reference's own tests hold no pixel vectors, so this
 * restatement is anchored only on the published algorithm as recalled.
 */
#ifndef TTO_SWSCALE_ORACLE_H
#define TTO_SWSCALE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TTO_ALG_BICUBIC = 0, TTO_ALG_AREA = 1, TTO_ALG_BILINEAR = 2 };

/* one axis of one plane: int16 taps, first-tap positions, taps per sample */
typedef struct tto_filter {
    int16_t *coef;   /* dst * size */
    int32_t *pos;    /* dst */
    int size;
    int dst;
    int src;
} tto_filter;

int  tto_filter_init(tto_filter *f, int src, int dst, int one, int alg);
void tto_filter_free(tto_filter *f);

/* scale one 8-bit plane (hScale8To15 then yuv2planeX_8 / yuv2plane1_8) */
int tto_scale_plane(const uint8_t *src, int sstride, int sw, int sh,
                    uint8_t *dst, int dstride, int dw, int dh, int alg);

/* geometry of a whole chain, all in luma pixels */
typedef struct tto_chain {
    int in_w, in_h;
    int pre_x, pre_y, pre_w, pre_h;   /* crop before scale (pre_w = in_w if none) */
    int do_scale, alg, sc_w, sc_h;    /* virtual scaled size */
    int crop_x, crop_y, crop_w, crop_h; /* crop after scale */
    int do_pad, pad_x, pad_y, out_w, out_h; /* pad canvas */
    int fill_y, fill_u, fill_v;
    int out_range;                    /* 0 keep, 1 limited->full, 2 full->limited: a second, unscaled swscale pass */
    int out_fmt;                      /* TTO_FMT_*: packed RGB written by the chain's swscale pass */
    int rgb_stage;                    /* 0 none, 1 inside the chain's scale (crop / pad then work on RGB), 2 a pass of its own at the end */
    int fill_rgb;                     /* pad colour 0xRRGGBB for pads that work on RGB */
    int in_fmt;                       /* TTO_FMT_YUV420P, or a packed layout (rgb24, bgr24, rgba, bgra): the chain's swscale pass reads packed RGB */
    int sc_range;                     /* 0 none, 1 limited->full, 2 full->limited: a range step INSIDE the converting context (on its 15-bit rows, between the passes) */
} tto_chain;

enum { TTO_FMT_YUV420P = 0, TTO_FMT_RGB24 = 1, TTO_FMT_BGR24 = 2, TTO_FMT_RGBA = 3, TTO_FMT_BGRA = 4 };

/* vf_scale output-size rule; foar: 0 disable, 1 decrease, 2 increase */
int tto_scale_dims(int iw, int ih, int w, int h, int foar, int fdb, int *ow, int *oh);
/* vf_crop default/explicit offset -> chroma-aligned int; pass x<0 for centred */
int tto_crop_offset(int in, int out, double expr_value, int use_expr);

/* run a yuv420p frame through the chain, unfused (scale whole, crop, pad) */
int tto_run_chain(const tto_chain *c,
                  const uint8_t *sy, const uint8_t *su, const uint8_t *sv,
                  int sys, int sus, int svs,
                  uint8_t *dy, uint8_t *du, uint8_t *dv,
                  int dys, int dus, int dvs);

/* One swscale context yuv420p (sw x sh) -> packed RGB (dw x dh), C path, default flags (no
 * full_chroma_int, no accurate_rnd): libswscale/yuv2rgb.c tables + output.c yuv2rgb_X_c_template, or the
 * unscaled converter yuv2rgb_c_24_rgb & co. when the sizes match and dh is even.
 * Rows whose vertical filters have 1 - 2 taps go through yuv2rgb_1_c_template / yuv2rgb_2_c_template as
 * vscale.c packed_vscale() picks them, per output row.
 * Returns 0, -1 on bad arguments, -2 where swscale's result is not defined (the unscaled converter never writes
 * the last column of an odd width: the byte is whatever the destination held). */
int tto_sws_yuv420_to_rgb(const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int sys, int sus, int svs,
                          int sw, int sh, uint8_t *dst, int dstride, int dw, int dh, int alg, int fmt);

/* the same for a source of the given range: full_range = 1 for yuvj420p (the tables of ff_yuv2rgb_c_init_tables() are
 * then built with fullRange = 1: no luma stretch, chroma coefficients x 224 / 255, table origin 384) */
int tto_sws_yuv420_to_rgb_range(const uint8_t *sy, const uint8_t *su, const uint8_t *sv, int sys, int sus, int svs,
                                int sw, int sh, uint8_t *dst, int dstride, int dw, int dh, int alg, int fmt, int full_range);

/* a chain that ends in packed RGB (c->out_fmt != 0): dst holds out_h rows of out_w * bpp bytes */
int tto_run_chain_rgb(const tto_chain *c, const uint8_t *sy, const uint8_t *su, const uint8_t *sv,
                      int sys, int sus, int svs, uint8_t *dst, int dstride);

/* sws_init_context()'s chrSrcHSubSample for a packed-RGB source and a 4:2:0 destination, default flags:
 * 1 (chroma read at half width, rgb24ToUV_half_c) iff (dw >> 1) <= (sw >> 1), else 0 (rgb24ToUV_c per pixel) */
int tto_rgb_in_half_chroma(int sw, int dw);

/* 1 when the context takes swscale_unscaled.c's "bgr24toYV12" converter instead (bgr24 in, same size out, no
 * accurate_rnd: rgb2rgb_template.c ff_rgb24toyv12_c -- truncating sums, chroma from the left pixel of every pair of
 * the even rows); tto_sws_packed_to_yuv420 then refuses an odd width (-1: its last column is never written) */
int tto_bgr24_special_converter(int fmt, int sw, int sh, int dw, int dh);

/* One swscale context packed rgb24 (sw x sh) -> yuv420p (dw x dh), C path, default flags:
 * libswscale/input.c rgb24ToY_c / rgb24ToUV_half_c or rgb24ToUV_c (14-bit samples), swscale.c hScale16To15_c,
 * output.c yuv2planeX_8_c.  Chroma: ceil(sw / 2) x sh samples (sw even) when the width does not grow, else
 * sw x sh, scaled to ceil(dw / 2) x ceil(dh / 2). */
int tto_sws_packed_to_yuv420(int fmt, const uint8_t *src, int sstride, int sw, int sh,
                             uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs, int dw, int dh, int alg);
/* ... into yuvj420p when full = 1: the same limited-range samples, lumRangeToJpeg_c / chrRangeToJpeg_c between the passes */
int tto_sws_packed_to_yuv420_range(int fmt, const uint8_t *src, int sstride, int sw, int sh,
                                   uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs, int dw, int dh, int alg, int full);
int tto_sws_rgb24_to_yuv420(const uint8_t *src, int sstride, int sw, int sh,
                            uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs, int dw, int dh, int alg);

/* a chain fed with packed rgb24 (c->in_fmt == TTO_FMT_RGB24) that hands out yuv420p */
int tto_run_chain_rgbin(const tto_chain *c, const uint8_t *src, int sstride,
                        uint8_t *dy, uint8_t *du, uint8_t *dv, int dys, int dus, int dvs);

#ifdef __cplusplus
}
#endif
#endif
