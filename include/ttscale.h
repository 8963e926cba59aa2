/*
 * ttscale.h -- C ABI of libttscale.so: yuv420p 8-bit scale / crop / pad on MI355X.
 *
 * The reference has no FFI for this path: its boundary is the argv of an external
 * `ffmpeg` process.  Each entry point below therefore replaces one
 * piece of that process boundary and cites the reference line that drives it.
 * Plain pointers and sizes only; no torch / C++ types cross this header.
 * All functions return 0 on success or a negative tts_status; nothing throws.
 */
#ifndef TTSCALE_H
#define TTSCALE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tts_status {
    TTS_OK = 0,
    TTS_ERR_INVALID_INPUT = -1,         /* web spec string INVALID_INPUT  (compression.test.js:62-90) */
    TTS_ERR_INVALID_FORMAT = -2,        /* INVALID_FORMAT: filter string not understood */
    TTS_ERR_INITIALIZATION_FAILED = -3, /* INITIALIZATION_FAILED: no HIP device / allocation failed */
    TTS_ERR_PROCESSING_FAILED = -4,     /* PROCESSING_FAILED: launch or copy failed */
    TTS_ERR_UNSUPPORTED = -5            /* geometry outside what the kernels cover */
} tts_status;

enum { TTS_ALG_BICUBIC = 0, TTS_ALG_AREA = 1, TTS_ALG_BILINEAR = 2 };
enum { TTS_MEM_DEVICE = 0, TTS_MEM_HOST = 1 };   /* tts_frame.mem: see tts_run */
/* Range step applied to the finished 8-bit frame, borders included: what the swscale pass that
 * ffmpeg auto-inserts between a yuv420p chain and an encoder wanting yuvj420p does -- the mjpeg
 * thumbnail of src/services/video.js:329-334 (`-q:v 2 out.jpg`).  SURVEY.md S7 / 8f-1. */
enum { TTS_RANGE_KEEP = 0, TTS_RANGE_TO_FULL = 1, TTS_RANGE_TO_LIMITED = 2 };
/* Pixel format of the frames a chain hands out: `format=rgb24 | bgr24 | rgba | bgra` ending it makes the
 * chain's swscale pass write packed RGB (BT.601, limited-range source, swscale's table-driven C path:
 * libswscale/yuv2rgb.c + output.c yuv2rgb_X_c_template; two neighbouring pixels share one chroma sample,
 * as without full_chroma_int).  The stage right after the thumbnail filters of
 * src/services/video.js:329-334 and the colour conversion BASELINE.json's north_star names.
 * Frames of a packed format are ONE plane of out_w * bpp bytes a row (bpp 3 or 4; alpha = 255). */
enum { TTS_FMT_YUV420P = 0, TTS_FMT_RGB24 = 1, TTS_FMT_BGR24 = 2, TTS_FMT_RGBA = 3, TTS_FMT_BGRA = 4,
       TTS_FMT_YUVJ420P = 5 };   /* tts_parse_vf_fmt's in_fmt only: yuv420p planes that hold FULL-range samples (`-pix_fmt yuvj420p -i`, what some
                                  * phone cameras record).  The geometry it gives has in_fmt TTS_FMT_YUV420P -- the planes are the same -- and
                                  * says what differs through sc_range / out_range and the pad colour */
/* Where the conversion sits, as libavfilter's format negotiation places it: inside the chain's `scale`
 * (everything after it -- crop, pad -- then works on RGB pixels: no chroma-grid rounding, pad colour in
 * RGB), or, in a chain without one, in a pass of its own at the very end (after pad, borders included). */
enum { TTS_RGB_NONE = 0, TTS_RGB_AT_SCALE = 1, TTS_RGB_AT_END = 2 };
/* A chain FED with packed RGB (in_fmt != 0) converts inside its `scale` too (rgb_stage 0).  Without a scale the
 * conversion pass is inserted at the end, which only matters behind a `pad`: the pad then runs on RGB pixels (any
 * offset, colour in RGB) and the conversion reads the padded canvas, border included -- rgb_stage TTS_RGB_AT_END. */

/*
 * Resolved geometry of one filter chain, in luma pixels of a yuv420p frame:
 *   [crop] -> [scale] -> [crop] -> [pad]   (every stage optional)
 * Produced by tts_parse_vf() or filled in by hand.
 */
typedef struct tts_geom {
    int32_t in_w, in_h;                       /* source frame */
    int32_t pre_x, pre_y, pre_w, pre_h;       /* crop applied before scale (whole frame if none) */
    int32_t do_scale, alg;                    /* scale stage present; TTS_ALG_* */
    int32_t sc_w, sc_h;                       /* (virtual) scaled size */
    int32_t crop_x, crop_y, crop_w, crop_h;   /* crop applied after scale (whole scaled frame if none) */
    int32_t do_pad, pad_x, pad_y;             /* pad stage present; where the picture sits */
    int32_t out_w, out_h;                     /* final frame size */
    int32_t fill_y, fill_u, fill_v;           /* pad colour */
    int32_t out_range;                        /* TTS_RANGE_*: `format=yuvj420p` / `format=yuv420p` ending the chain */
    int32_t out_fmt;                          /* TTS_FMT_*: what the chain hands out (0 = yuv420p) */
    int32_t rgb_stage;                        /* TTS_RGB_*: where the colour conversion sits (0 when the chain stays yuv420p, or is fed RGB and has a scale or no pad) */
    int32_t fill_rgb;                         /* pad colour as 0xRRGGBB, used when the pad stage works on RGB pixels */
    int32_t in_fmt;                           /* TTS_FMT_*: what the chain is fed (tts_parse_vf_fmt); 0 = yuv420p */
    int32_t sc_range;                         /* TTS_RANGE_*: a range step INSIDE the chain's converting context -- the scale writes the other
                                               * range itself (`scale=...,format=yuvj420p` with no format=yuv420p between the two; a packed-RGB
                                               * source converted straight into yuvj420p): swscale runs it on its 15-bit rows BETWEEN the
                                               * horizontal and the vertical pass; crop and pad behind it work on the new range (fill_y/u/v are
                                               * values of that range).  out_range is a pass of its own over the finished frame (the reference's
                                               * thumbnail chain pins format=yuv420p before the mjpeg encoder asks for yuvj420p); both may be set */
} tts_geom;

/* One frame: Y, U, V plane pointers and byte strides (yuv420p), or plane[0] / stride[0] alone (packed RGB). */
typedef struct tts_frame {
    void   *plane[3];
    int32_t stride[3];
    int32_t width, height;
    int32_t mem;                               /* TTS_MEM_DEVICE or TTS_MEM_HOST */
} tts_frame;

typedef struct tts_plan tts_plan;

/*
 * Parse a `-vf` chain exactly as the reference writes them and resolve it for an
 * iw x ih input.  Replaces libavfilter's option parsing for:
 *   scripts/processVideos.js:94-97   `.size('720x1280')` -> "scale=w=720:h=1280"
 *   src/services/video.js:329-332    "scale=320x180:force_original_aspect_ratio=decrease,
 *                                     pad=320x180:-1:-1:color=black,format=yuv420p"
 *   web/src/tests/video/portrait.test.js:78,95-96,146  "scale=W:H", "min(720,...", decrease
 * Accepted filters: scale (bicubic | area | bilinear), crop, pad, setsar,
 * format=yuv420p | yuvj420p | rgb24 | bgr24 | rgba | bgra.
 */
int tts_parse_vf(const char *vf, int iw, int ih, tts_geom *out);

/*
 * The same for a chain fed with packed rgb24 frames (`-f rawvideo -pix_fmt rgb24 -i ...`, in_fmt =
 * TTS_FMT_RGB24; TTS_FMT_YUV420P is tts_parse_vf): the other direction of the colour conversion BASELINE.json's
 * north_star names.  The chain must hand out yuv420p.  As libavfilter negotiates it, the
 * chain's `scale` reads the RGB (a crop in front of it works on RGB pixels: no chroma grid) and writes yuv420p:
 * swscale's input side (libswscale/input.c rgb24ToY_c, rgb24ToUV_half_c: BT.601 limited range, two
 * neighbouring pixels averaged into one chroma sample, every source row its own chroma row) feeding the scaler's
 * 16-bit horizontal pass.  Without a scale the conversion is a pass of its own behind the crops.
 * Source frames are ONE plane of in_w * 3 bytes a row; in_w (after a crop: its width) must be even unless the chain
 * widens the picture (chroma is then read per pixel: rgb24ToUV_c).
 * in_fmt = TTS_FMT_BGR24, TTS_FMT_RGBA, TTS_FMT_BGRA: the same arithmetic on the other byte orders (input.c bgr24ToY_c
 * & co., and the 32-bit readers, whose extra fractional bits are all zero; alpha is ignored); 4 bytes a pixel for the last two.
 */
int tts_parse_vf_fmt(const char *vf, int iw, int ih, int in_fmt, tts_geom *out);

/*
 * Build the per-geometry state on the current HIP device (coefficient tables in HBM).
 * Replaces sws_getContext()/sws_init_context() inside the ffmpeg child that
 * scripts/processVideos.js:108 (`.save()`) and src/services/video.js:334 spawn.
 * The plan is immutable afterwards and may be shared between host threads.  It belongs to the
 * device it was made on: the run calls below return TTS_ERR_INVALID_INPUT (tts_last_error names
 * both devices) when another device is current -- one plan per GPU in a process that drives several.
 */
int tts_plan_create(const tts_geom *g, tts_plan **out);
void tts_plan_destroy(tts_plan *p);
int tts_plan_geom(const tts_plan *p, tts_geom *out);
int tts_plan_device(const tts_plan *p);        /* index of the HIP device the plan belongs to (negative status if p is NULL) */
/* Bytes of one packed frame going in (which = 0: I420 of in_w x in_h) or coming out (which = 1: I420 of
 * out_w x out_h, or out_w * out_h * bpp for a packed-RGB chain): the frame pitches tts_run_batch expects. */
size_t tts_plan_frame_bytes(const tts_plan *p, int which);

/*
 * Run n_frames frames through the plan, asynchronously on hip_stream (a hipStream_t,
 * NULL = default stream).  Replaces the per-frame scale/crop/pad loop of the ffmpeg
 * child (scripts/processVideos.js:71-108).  TTS_MEM_DEVICE frames: the caller owns every
 * buffer and nothing is allocated here.  Planes may lie anywhere (every plane its
 * own allocation, any stride >= width; U and V strides equal); nothing outside a plane is read.
 * TTS_MEM_HOST frames (every frame of the call, both sides; any strides): copied plane by plane
 * through the staging pool of tts_run_host, synchronously -- hip_stream is not used, and the
 * frames are finished when the call returns.
 * Packed formats: an RGB destination frame is plane[0] / stride[0] alone (>= out_w * bpp); an rgb24 SOURCE
 * frame (tts_parse_vf_fmt) likewise (stride[0] >= in_w * 3; device frames only).
 */
int tts_run(const tts_plan *p, const tts_frame *src, const tts_frame *dst, int n_frames, void *hip_stream);

/*
 * Same, for a batch laid out as consecutive packed I420 frames (Y, U, V back to back,
 * strides == widths) at a fixed byte pitch between frames -- what a raw-video demuxer
 * (`-f rawvideo -pix_fmt yuv420p`) hands over.  Device pointers.  A packed-RGB chain writes
 * consecutive frames of out_w * out_h * bpp bytes (`-f rawvideo -pix_fmt rgb24`).
 */
int tts_run_batch(const tts_plan *p, const void *src, size_t src_frame_pitch,
                  void *dst, size_t dst_frame_pitch, int n_frames, void *hip_stream);

/*
 * The same with padded rows, as decoders and encoders hand frames over (AVFrame.linesize[]): inside
 * a frame Y, U, V still follow each other, rows `*_stride_y` (luma) and `*_stride_c` (both chroma
 * planes) bytes apart, each at least the plane's width (packed RGB out: dst_stride_y = bytes between
 * RGB rows, dst_stride_c ignored).  Padding bytes are never written.  (Packed
 * output rows are the fastest to write: 1080-column rows at a stride of 1088 measured 17 % slower
 * than at 1080 -- the 8-byte holes leave a partly written 64-byte sector at every row's end.)
 */
int tts_run_batch_strided(const tts_plan *p, const void *src, size_t src_frame_pitch, int src_stride_y, int src_stride_c,
                          void *dst, size_t dst_frame_pitch, int dst_stride_y, int dst_stride_c, int n_frames, void *hip_stream);

/*
 * Time-domain steps as frame indices (host only, no device call).  A frame i of a constant-rate clip
 * has timestamp i * fps_den / fps_num and belongs to `-ss start -t duration` when that lies in
 * [start, start + duration); duration < 0 = to the end.
 *   tts_trim_range  replaces `.seekInput(start).duration(d)` / `-ss`, `-t`
 *                   (scripts/processVideos.js:84-92, scripts/loopVideos.js:42-52).
 *   tts_loop_order  replaces createLoopedVideo (scripts/loopVideos.js:54-100): initial [0, start) +
 *                   repeats x [start, start + duration) + end [start + duration, n), joined by the concat
 *                   demuxer.  order == NULL: returns the number of output frames; else fills order[0..cap)
 *                   and returns the count.  (The script's `crossfade: 0.5` is never read -- :55 -- so
 *                   there is no blend step to replace.)
 */
int tts_trim_range(int n_frames, int fps_num, int fps_den, double start, double duration, int *first, int *count);
int tts_loop_order(int n_frames, int fps_num, int fps_den, double extract_start, double extract_duration,
                   int repeats, int32_t *order, int cap);

/*
 * tts_run_batch through a frame order (host array): output frame j = the plan applied to source frame
 * order[j].  A trimmed or looped clip is scaled straight from the original frames; no segment is
 * copied out first (the reference writes three temporary files, scripts/loopVideos.js:56-74).
 */
int tts_run_batch_order(const tts_plan *p, const void *src, size_t src_frame_pitch, int n_src_frames,
                        void *dst, size_t dst_frame_pitch, const int32_t *order, int n_out, void *hip_stream);

/*
 * Convenience for callers that hold host memory (the CLI, the Node addon): copies in, runs,
 * copies out, synchronises.  Packed frames as for tts_run_batch.  The staging buffers and streams
 * belong to a per-device pool that grows to the largest call seen and is reused by later calls
 * (one thumbnail per call, as src/services/video.js:315-368 works, pays no allocation after the
 * first); tts_trim() frees the pool of the current device.  Calls on one device take turns.
 */
int tts_run_host(const tts_plan *p, const void *src, void *dst, int n_frames);
void tts_trim(void);

/*
 * tts_run_host for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Stages through a 64 MiB
 * `static __device__` array inside the library (each GPU the library runs on holds its copy until the
 * process ends) instead of pool buffers: calls no hipMalloc / hipFree / hipStreamCreate, runs on the
 * caller's hip_stream, and does NOT synchronise -- dst is complete once hip_stream has passed the
 * call's last copy (pinned host buffers make the copies truly asynchronous).  Batches larger than
 * the array go through it in several passes; TTS_ERR_UNSUPPORTED if a single frame (source + output,
 * each rounded up to 256 bytes) exceeds 64 MiB.  Every call on a GPU shares that GPU's array: calls
 * that overlap in time on one GPU (two streams, two threads) race and corrupt each other's frames.
 */
int tts_run_host_static(const tts_plan *p, const void *src, void *dst, int n_frames, void *hip_stream);

/*
 * Read back the integer filter tables of one plane class (0 = luma, 1 = chroma) and
 * axis (0 = horizontal, 1 = vertical) so tests can compare them with the oracle.
 * coef/pos may be NULL to query sizes.  Replaces nothing in the reference: swscale
 * keeps these private.
 */
int tts_plan_filter(const tts_plan *p, int plane_class, int axis,
                    int *taps, int *n_out, int16_t *coef, int32_t *pos);

/*
 * Host-only: the integer table that resamples `src` samples to `dst` samples
 * (one = 1<<14 horizontal, 1<<12 vertical; alg = TTS_ALG_*).  No device is touched.
 * Call with coef == NULL to learn *taps, then with coef[dst * *taps], pos[dst].
 * Replaces nothing in the reference (libswscale's initFilter is private).
 */
int tts_build_filter(int src, int dst, int one, int alg, int *taps, int16_t *coef, int32_t *pos);

/* Which kernel the plan dispatches to for a plane class: 0 = generic, 1 = strip walk with register
 * gathers, 2 / 3 = LDS-staged strip walk with one / more (two or four) LDS-DMAs per source row, 4 / 5 = column walk
 * (9..64 taps per axis) with one / two LDS-DMAs per source row. */
int tts_plan_kernel_kind(const tts_plan *p, int plane_class);
/* The same with the template arguments, for logs and tests: "strip_lds<6,6,grouped,1,1>" = window bytes,
 * ring rows, grouped (one window per 4 columns) / grouped2 (per 2) or percolumn windows, LDS-DMAs per row,
 * rows in flight; "column_lds<6,12,1,2,4>" = tap pairs, ring rows, LDS-DMAs per row, columns per lane,
 * rows in flight (above 8 tap pairs small launches take one column per lane; the name is the large-launch
 * variant); "strip<...>" the register walk, "generic".
 * Valid until the calling thread's next call. */
const char *tts_plan_kernel_name(const tts_plan *p, int plane_class);

size_t tts_frame_bytes(int w, int h);          /* packed I420 size of a w x h frame */
const char *tts_strerror(int status);
const char *tts_last_error(void);          /* detail of the calling thread's last tts_parse_vf failure */
const char *tts_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TTSCALE_H */
