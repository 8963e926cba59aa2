// libttscale: yuv420p 8-bit scale / crop / pad for MI355X (gfx950), HIP kernels + C ABI.
//
// Replaces the per-frame `scale`/`crop`/`pad` loop that the reference hands to an
// external ffmpeg (scripts/processVideos.js:71-108, src/services/video.js:329-334).
// Arithmetic contract: horizontal pass to a 15-bit intermediate
// with a truncating >>7 and a clip to 32767, vertical pass with +64<<12 rounding,
// >>19 and a clip to 8 bits; integer only, bit-exact against the CPU oracle.
//
// One translation unit in three files:
//   ttscale_kernels.h   the kernels (device code) and their argument structs
//   ttscale_plan.h      tap tables per geometry, kernel choice, launch sets
//   ttscale.hip         this file: the C ABI of include/ttscale.h
// filter.cpp / vf_parse.cpp / timeline.cpp hold the host-only parts (tap generation, filter
// string parsing, frame orders) and are separate translation units.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <string>
#include <tuple>
#include <vector>

#include "filter.h"
#include "ttscale_rgb.h"

#include "ttscale_kernels.h"
#include "ttscale_plan.h"

namespace tts {
void set_last_error(const std::string &msg) { g_last_error = msg; }
}  // namespace tts

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

int tts_parse_vf_fmt(const char *vf, int iw, int ih, int in_fmt, tts_geom *out)
{
    if (!out) return TTS_ERR_INVALID_INPUT;
    std::string err;
    tts_geom g;
    const int rc = tts::parse_vf(vf, iw, ih, g, err, in_fmt);
    g_last_error = err;
    if (rc == TTS_OK) *out = g;
    return rc;
}

// The two plane classes of a planar plan and their table buffer.  full_v: see build_class.
static int build_classes(tts_plan *p, const tts_geom &g, bool full_v, bool force_column = false)
{
    std::vector<uint32_t> tab;
    int rc = TTS_OK;
    // (a range step inside the scale's context: strip classes whose windows do not group have no instantiation for it and
    // take the column walk, whatever their tap count)
    const bool range_in_scale = g.in_fmt == TTS_FMT_YUV420P && g.out_fmt == TTS_FMT_YUV420P && g.sc_range != TTS_RANGE_KEEP;
    for (int ci = 0; ci < 2 && rc == TTS_OK; ++ci) {
        rc = build_class(g, ci, p->cls[ci], tab, 0, full_v && ci, force_column);   // (full_v: a scratch plan)
        if (rc == TTS_OK && range_in_scale && p->cls[ci].kind == 1 && !strip_has_rng(strip_choice(p->cls[ci], p->knobs))) {
            free_class(p->cls[ci]);
            p->cls[ci] = ClassPlan();
            rc = build_class(g, ci, p->cls[ci], tab, 0, full_v && ci, true);
        }
    }
    // Column walk: luma and chroma ratios differ by a rounding, so their tap counts may straddle an
    // instantiation boundary (720x1280 -> 101x180: 16 tap pairs luma, 14 chroma) and the chain would take two
    // launches.  The class with fewer is rebuilt with the other's window: a seventh more arithmetic on a third
    // of the samples against one launch less on a path that is one thumbnail per call in the reference.
    if (rc == TTS_OK && p->cls[0].kind == 2 && p->cls[1].kind == 2 && p->cls[0].NTP != p->cls[1].NTP &&
        std::abs(p->cls[0].NTP - p->cls[1].NTP) <= 2 && !p->knobs.no_merge) {
        const int lo = p->cls[0].NTP < p->cls[1].NTP ? 0 : 1, want = p->cls[1 - lo].NTP;
        free_class(p->cls[lo]);
        p->cls[lo] = ClassPlan();
        rc = build_class(g, lo, p->cls[lo], tab, want, full_v && lo, force_column);
        if (rc == TTS_OK && p->cls[lo].kind != 2) {         // (the wider window does not fit this class's rows: as it was)
            free_class(p->cls[lo]);
            p->cls[lo] = ClassPlan();
            rc = build_class(g, lo, p->cls[lo], tab, 0, full_v && lo, force_column);
        }
    }
    if (rc == TTS_OK && !upload(p->d_tab, tab)) rc = TTS_ERR_INITIALIZATION_FAILED;
    if (rc == TTS_OK && p->knobs.persist && !p->d_queue && hipMalloc(reinterpret_cast<void **>(&p->d_queue), 64) != hipSuccess) rc = TTS_ERR_INITIALIZATION_FAILED;
    return rc;
}

namespace { int scratch_for_device(int dev, unsigned char **out); }

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
    // tts_run_host_static's scratch: its device address is looked up here, at set-up, so that the entry point itself does no
    // first-call runtime work (a failure is kept for that call to report)
    { unsigned char *scratch_base = nullptr; (void)scratch_for_device(dev, &scratch_base); }
    if (g->in_fmt != TTS_FMT_YUV420P) {
        rc = tts::rgbin_plan_create(*g, &p->rgbin);            // (sets the error text itself)
        if (rc) { tts_plan_destroy(p); return rc; }
        *out = p;
        return TTS_OK;
    }
    if (g->out_fmt != TTS_FMT_YUV420P) {
        rc = tts::rgb_plan_create(*g, p->n_cus, &p->rgb);      // (sets the error text itself)
        if (rc) { tts_plan_destroy(p); return rc; }
        // Thumbnail-sized pictures written as RGB by a many-tap scale: a planar plan of the same scale (chroma at
        // every output row, luma unclipped) whose column walk writes into the RGB frame, converted in place.
        // Anything that does not qualify -- or whose classes do not both take the column walk -- stays with p->rgb.
        tts::RgbScratchLayout lay;
        if (!p->knobs.no_rgb_scratch && !tts::rgb_plan_is_strip(p->rgb) && !tts::rgb_plan_general_only(p->rgb) && tts::rgb_scratch_layout(*g, g->out_w * (g->out_fmt >= TTS_FMT_RGBA ? 4 : 3), lay)) {
            tts_plan *sp = new (std::nothrow) tts_plan();
            if (sp) {
                tts_geom gp = *g;
                gp.out_fmt = TTS_FMT_YUV420P; gp.rgb_stage = TTS_RGB_NONE; gp.fill_rgb = 0;
                gp.do_pad = 0; gp.pad_x = gp.pad_y = 0; gp.out_w = g->crop_w; gp.out_h = g->crop_h;
                sp->g = gp; sp->device = dev; sp->knobs = p->knobs; sp->n_cus = p->n_cus; sp->y16 = true;
                // 5-8 taps: the strip kernels where both classes get a grouped-window instantiation (those have the
                // 16-bit luma store), else -- and above 8 taps -- the column walk
                auto strips16 = [&](const ClassPlan &c) { return c.kind == 1 && strip_has_y16(strip_choice(c, sp->knobs)); };
                bool ok = build_classes(sp, gp, true) == TTS_OK;
                if (ok && !(strips16(sp->cls[0]) && strips16(sp->cls[1])) && (sp->cls[0].kind != 2 || sp->cls[1].kind != 2)) {
                    free_class(sp->cls[0]); free_class(sp->cls[1]);
                    sp->cls[0] = ClassPlan(); sp->cls[1] = ClassPlan();
                    (void)hipFree(sp->d_tab); sp->d_tab = nullptr;
                    ok = build_classes(sp, gp, true, true) == TTS_OK && sp->cls[0].kind == 2 && sp->cls[1].kind == 2;
                }
                if (ok) p->scratch = sp;
                else tts_plan_destroy(sp);
            }
        }
        *out = p;
        return TTS_OK;
    }
    rc = build_classes(p, *g, false);
    if (rc) {
        if (rc == TTS_ERR_INITIALIZATION_FAILED) g_last_error = "table upload failed (hipMalloc / hipMemcpy)";
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
    (void)hipFree(p->d_queue);
    tts::rgb_plan_destroy(p->rgb);
    tts::rgbin_plan_destroy(p->rgbin);
    tts_plan_destroy(p->scratch);
    delete p;
}

int tts_plan_geom(const tts_plan *p, tts_geom *out)
{
    if (!p || !out) return TTS_ERR_INVALID_INPUT;
    *out = p->g;
    return TTS_OK;
}

int tts_plan_device(const tts_plan *p) { return p ? p->device : TTS_ERR_INVALID_INPUT; }

size_t tts_plan_frame_bytes(const tts_plan *p, int which)
{
    if (!p) return 0;
    if (which == 0) return p->g.in_fmt != TTS_FMT_YUV420P ? size_t(p->g.in_w) * p->g.in_h * tts::packed_pixel_bytes(p->g.in_fmt) : tts_frame_bytes(p->g.in_w, p->g.in_h);
    if (p->g.out_fmt == TTS_FMT_YUV420P) return tts_frame_bytes(p->g.out_w, p->g.out_h);
    return size_t(p->g.out_w) * p->g.out_h * (p->g.out_fmt >= TTS_FMT_RGBA ? 4 : 3);
}

int tts_plan_filter(const tts_plan *p, int plane_class, int axis, int *taps, int *n_out, int16_t *coef, int32_t *pos)
{
    if (!p || plane_class < 0 || plane_class > 1 || axis < 0 || axis > 1) return TTS_ERR_INVALID_INPUT;
    if (p->rgb || p->rgbin) return TTS_ERR_UNSUPPORTED;   // (the RGB paths keep four tables: luma and chroma of one swscale context)
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
    if (p->rgb) return 6;
    if (p->rgbin) return 7;
    const ClassPlan &c = p->cls[plane_class];
    if (c.kind == 0) return 0;
    if (c.kind == 2) return 3 + c.col_dmas;
    const int lds = strip_choice(c, p->knobs).lds;
    return lds == 0 ? 1 : lds == 1 ? 2 : 3;
}

const char *tts_plan_kernel_name(const tts_plan *p, int plane_class)
{
    static thread_local char buf[96];
    if (!p || plane_class < 0 || plane_class > 1) return "";
    if (p->rgb && tts::rgb_plan_kernel_name(p->rgb)[4] == 'w') return tts::rgb_plan_kernel_name(p->rgb);   // "rgb_walk<...>": what packed batches run
    if (p->scratch) {                                       // (what packed batches run; pointers and padded rows: the generic kernel)
        static thread_local char sbuf[128];
        std::snprintf(sbuf, sizeof sbuf, "%s+rgb_convert", tts_plan_kernel_name(p->scratch, plane_class));
        return sbuf;
    }
    if (p->rgb) return tts::rgb_plan_kernel_name(p->rgb);
    if (p->rgbin) {
        if (p->g.out_range == TTS_RANGE_KEEP) return tts::rgbin_plan_kernel_name(p->rgbin);
        static thread_local char rbuf[128];                 // (the range pass at the end: a launch of its own behind these kernels)
        std::snprintf(rbuf, sizeof rbuf, "%s+range_pass", tts::rgbin_plan_kernel_name(p->rgbin));
        return rbuf;
    }
    const ClassPlan &c = p->cls[plane_class];
    if (c.kind == 0) std::snprintf(buf, sizeof buf, "generic");
    else if (c.kind == 2) {
        bool d4 = true;                                  // (as launch_all decides: every column class of the launch)
        for (int ci = 0; ci < 2; ++ci)
            if (p->cls[ci].kind == 2) d4 &= column_dot4(p->cls[ci], p->knobs);
        const bool rng15 = p->g.in_fmt == TTS_FMT_YUV420P && p->g.out_fmt == TTS_FMT_YUV420P && p->g.sc_range != TTS_RANGE_KEEP;   // the scale's own range step
        std::snprintf(buf, sizeof buf, "column_lds<%d,%d,%d,%d,%d%s%s>", c.NTP, c.Rc, c.col_dmas, c.col_nc, column_depth(c.NTP), d4 ? ",dot4" : "", rng15 ? ",range15" : "");
    }
    else {
        const StripChoice ch = strip_choice(c, p->knobs);
        const bool rng15 = p->g.in_fmt == TTS_FMT_YUV420P && p->g.out_fmt == TTS_FMT_YUV420P && p->g.sc_range != TTS_RANGE_KEEP;
        if (ch.lds) std::snprintf(buf, sizeof buf, "strip_lds<%d,%d,%s,%d,%d%s>", ch.LW, ch.R, ch.g12 == 2 ? "grouped2" : ch.g4 ? "grouped" : "percolumn", ch.lds, ch.depth, rng15 ? ",range15" : "");
        else std::snprintf(buf, sizeof buf, "strip<%d,%d,%d,%s>", ch.LW, ch.R, ch.N, ch.g4 ? "grouped" : "percolumn");
    }
    return buf;
}

size_t tts_frame_bytes(int w, int h)
{
    if (w < 1 || h < 1) return 0;
    return size_t(w) * h + 2 * size_t((w + 1) / 2) * ((h + 1) / 2);
}

}  // extern "C"

namespace {

// A plan's tables live in the HBM of the device that was current when it was made; run on another
// device the kernels would read them across the fabric, or fault where peer access is off.
int check_device(const tts_plan *p)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { g_last_error = "hipGetDevice failed"; return TTS_ERR_PROCESSING_FAILED; }
    if (dev != p->device) {
        g_last_error = "plan was created on device " + std::to_string(p->device) + ", current device is " + std::to_string(dev);
        return TTS_ERR_INVALID_INPUT;
    }
    return TTS_OK;
}

// Packed RGB in, tts_geom.out_range set (`...,format=yuv420p,format=yuvj420p`: the reference's thumbnail chain and its mjpeg
// encoder, fed with RGB frames): the range pass over the finished canvas, in place, behind the kernels of ttscale_rgb.hip.
// `addr` holds the canvas origins of the destination planes; `only` as in rgbin_launch.
int rgbin_range_pass(const tts_plan *p, uint8_t *dst, int n, const int dstride[2], const FrameAddr &addr, hipStream_t st, int only = -1)
{
    const tts_geom &g = p->g;
    if (g.out_range == TTS_RANGE_KEEP) return TTS_OK;
    for (int ci = 0; ci < 2; ++ci) {
        if (only >= 0 && ci != (only ? 1 : 0)) continue;
        RangeArgs r{};
        r.dst = dst; r.w = ceil_shift(g.out_w, ci); r.h = ceil_shift(g.out_h, ci); r.stride = dstride[ci];
        r.planes = only >= 0 ? 1 : ci ? 2 : 1; r.plane0 = only >= 0 ? only : ci;
        r.origin = 0;
        r.addr = addr;
        tts::build_range_lut(g.out_range, ci != 0, r.lut);
        r.affine = tts::range_affine(g.out_range, ci != 0, r.mul, r.add, r.shift) ? 1 : 0;
        if (r.stride == r.w && r.h > 1) { r.w *= r.h; r.h = 1; }       // rows back to back: the plane as one row
        const int per = 65535 / r.planes;
        for (int f0 = 0; f0 < n; f0 += per) {
            r.frame0 = f0;
            dim3 grid(r.h == 1 ? (r.w + 16 * kBlock - 1) / (16 * kBlock) : (r.w + 1023) / 1024, (r.h + 3) / 4, std::min(per, n - f0) * r.planes);
            if (const int rc = launch_checked("range_kernel", range_kernel, grid, dim3(kBlock), 0, st, r)) return rc;
        }
    }
    return TTS_OK;
}

// Frames at a fixed pitch, Y / U / V back to back inside a frame, rows `sstride[class]` / `dstride[class]`
// bytes apart (class 0 = luma, 1 = both chroma planes).
int run_batch_impl(const tts_plan *p, const void *src, size_t src_pitch, const int sstride[2], void *dst, size_t dst_pitch,
                   const int dstride[2], int n_frames, void *hip_stream)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    const tts_geom &g = p->g;
    if (p->rgbin) {
        // packed rgb24 in (one plane, rows sstride[0] bytes apart), packed or strided I420 out; the kernel covers the canvas
        if (sstride[0] < g.in_w * tts::packed_pixel_bytes(g.in_fmt) || dstride[0] < g.out_w || dstride[1] < ceil_shift(g.out_w, 1)) return TTS_ERR_INVALID_INPUT;
        const size_t sbytes = size_t(sstride[0]) * g.in_h;
        const size_t dbytes = size_t(dstride[0]) * g.out_h + 2 * size_t(dstride[1]) * ceil_shift(g.out_h, 1);
        if (src_pitch < sbytes || dst_pitch < dbytes) return TTS_ERR_INVALID_INPUT;
        if (src_pitch >= (size_t(1) << 31) || dst_pitch >= (size_t(1) << 31)) return TTS_ERR_UNSUPPORTED;
        if (const int rc = check_device(p)) return rc;
        FrameAddr addr{};
        addr.src_pitch = uint32_t(src_pitch); addr.dst_pitch = uint32_t(dst_pitch);
        addr.src_plane[0] = 0;                         // (the pre-crop origin is rgbin_launch's business: it knows the stride)
        addr.dst_plane[0] = 0;
        addr.dst_plane[1] = uint32_t(dstride[0]) * uint32_t(g.out_h);
        addr.dst_plane[2] = addr.dst_plane[1] + uint32_t(dstride[1]) * uint32_t(ceil_shift(g.out_h, 1));
        if (const int rc = tts::rgbin_launch(p->rgbin, static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), n_frames, sstride[0], dstride,
                                             addr, static_cast<hipStream_t>(hip_stream))) return rc;
        return rgbin_range_pass(p, static_cast<uint8_t *>(dst), n_frames, dstride, addr, static_cast<hipStream_t>(hip_stream));
    }
    const int bpp = g.out_fmt == TTS_FMT_YUV420P ? 0 : g.out_fmt >= TTS_FMT_RGBA ? 4 : 3;   // packed RGB out: one plane
    for (int ci = 0; ci < 2; ++ci)
        if (sstride[ci] < ceil_shift(g.in_w, ci) || (!bpp && dstride[ci] < ceil_shift(g.out_w, ci))) return TTS_ERR_INVALID_INPUT;
    if (bpp && dstride[0] < g.out_w * bpp) return TTS_ERR_INVALID_INPUT;
    const size_t sbytes = size_t(sstride[0]) * g.in_h + 2 * size_t(sstride[1]) * ceil_shift(g.in_h, 1);
    const size_t dbytes = bpp ? size_t(dstride[0]) * g.out_h : size_t(dstride[0]) * g.out_h + 2 * size_t(dstride[1]) * ceil_shift(g.out_h, 1);
    if (src_pitch < sbytes || dst_pitch < dbytes) return TTS_ERR_INVALID_INPUT;
    if (src_pitch >= (size_t(1) << 31) || dst_pitch >= (size_t(1) << 31)) return TTS_ERR_UNSUPPORTED;
    if (const int rc = check_device(p)) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    tts::RgbScratchLayout lay{};
    // (sources under 4 KiB go to the one-lane-per-sample kernel, which has no 16-bit store: not through the scratch pass)
    bool use_scratch = p->rgb && p->scratch && sbytes >= 4096 && tts::rgb_scratch_layout(g, dstride[0], lay);
    // (5-8 taps an axis, packed batches: the fused walk writes the pixels at once)
    if (use_scratch && tts::rgb_plan_walks(p->rgb, n_frames, uint64_t(n_frames - 1) * src_pitch + sbytes, uint32_t(src_pitch), uint32_t(dst_pitch))) use_scratch = false;
rgb_generic:
    if (p->rgb && !use_scratch) {
        // The RGB kernels cover the whole canvas (borders included) and address every frame from its own base
        // (per workgroup or per wave), so one launch set takes the batch whatever its span.
        if (sbytes > kMaxSpan || dbytes > kMaxSpan) return TTS_ERR_UNSUPPORTED;
        FrameAddr addr{};
        addr.src_pitch = uint32_t(src_pitch); addr.dst_pitch = uint32_t(dst_pitch); addr.wave_base = 1;
        uint32_t splane = 0;
        for (int k = 0; k < 3; ++k) {
            const int ci = k ? 1 : 0;
            addr.src_plane[k] = splane + uint32_t(g.pre_y >> ci) * sstride[ci] + uint32_t(g.pre_x >> ci);
            splane += uint32_t(sstride[ci]) * ceil_shift(g.in_h, ci);
        }
        return launch_all(p, static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), n_frames, sstride, dstride,
                          uint64_t(n_frames - 1) * src_pitch + sbytes, uint64_t(n_frames - 1) * dst_pitch + dbytes, addr, st);
    }

    // (packed-RGB thumbnails: the planar pass of p->scratch writes its planes into the RGB frames, see tts_plan)
    const tts_plan *pp = use_scratch ? p->scratch : p;
    const int pds[2] = {use_scratch ? lay.stride[0] : dstride[0], use_scratch ? lay.stride[1] : dstride[1]};
    FrameAddr addr{};
    addr.src_pitch = uint32_t(src_pitch); addr.dst_pitch = uint32_t(dst_pitch);
    uint32_t splane = 0, dplane = 0;
    for (int k = 0; k < 3; ++k) {
        const int ci = k ? 1 : 0;
        const ClassPlan &c = pp->cls[ci];
        addr.src_plane[k] = splane + uint32_t(g.pre_y >> ci) * sstride[ci] + uint32_t(g.pre_x >> ci);
        addr.dst_plane[k] = use_scratch ? lay.off[k] : dplane + uint32_t(c.pad_y) * dstride[ci] + uint32_t(c.pad_x);
        splane += uint32_t(sstride[ci]) * ceil_shift(g.in_h, ci);
        dplane += uint32_t(dstride[ci]) * ceil_shift(g.out_h, ci);
    }
    // Lane offsets are 32-bit and count from the first frame of the lane's WAVE (frame_offsets), so a
    // launch may cover any number of frames as long as the frames one wave can touch stay within
    // kMaxSpan: a wave holds 64 items, a frame at least `ipf` of them.  (Pitches so large that they do
    // not -- hundreds of megabytes between frames -- fall back to launches whose whole span fits.)
    if (sbytes > kMaxSpan || dbytes > kMaxSpan) return TTS_ERR_UNSUPPORTED;
    const int ipf = std::max(1, (std::min(pp->cls[0].ow, pp->cls[1].ow) + 15) / 16);
    const size_t wave_frames = size_t(63 / ipf + 2);
    const bool any_size = wave_frames * src_pitch + sbytes <= kMaxSpan && wave_frames * dst_pitch + dbytes <= kMaxSpan && !p->knobs.span_chunks;
    // items and workgroups are counted in 32 bits: at most 2^30 output columns a launch
    const size_t cols_per_frame = size_t(pp->cls[0].ow) + 2 * size_t(pp->cls[1].ow);
    const size_t max_frames = std::max<size_t>(1, (size_t(1) << 30) / std::max<size_t>(cols_per_frame, 1));
    const size_t by_span = 1 + std::min((kMaxSpan - sbytes) / src_pitch, (kMaxSpan - dbytes) / dst_pitch);
    const int chunk = int(std::min(any_size ? max_frames : std::min(by_span, max_frames), size_t(1) << 30));
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int n = std::min(chunk, n_frames - f0);
        const uint8_t *s = static_cast<const uint8_t *>(src) + size_t(f0) * src_pitch;
        uint8_t *d = static_cast<uint8_t *>(dst) + size_t(f0) * dst_pitch;
        const uint64_t sspan = uint64_t(n - 1) * src_pitch + sbytes, dspan = uint64_t(n - 1) * dst_pitch + dbytes;
        addr.wave_base = sspan > kMaxSpan || dspan > kMaxSpan || p->knobs.wave_base;
        const int rc = launch_all(pp, s, d, n, sstride, pds, sspan, dspan, addr, st);
        if (rc == TTS_ERR_UNSUPPORTED && use_scratch) { use_scratch = false; goto rgb_generic; }   // (whatever the pass wrote, the generic kernel overwrites)
        if (rc) return rc;
    }
    if (use_scratch) {
        FrameAddr ca{};
        ca.dst_pitch = uint32_t(dst_pitch);
        return tts::rgb_scratch_convert(p->rgb, static_cast<uint8_t *>(dst), dstride[0], n_frames, ca, st);
    }
    return TTS_OK;
}

// Packed rgb24 frames through per-frame pointers (tts_run): one source plane, rows stride[0] bytes apart, three
// destination planes.  Frames are grouped like the planar path's: up to kMaxTableFrames whose planes lie within
// kMaxSpan of a common base ride in one launch as 32-bit offsets.
int run_rgbin_frames(const tts_plan *p, const tts_frame *src, const tts_frame *dst, int n_frames, void *hip_stream)
{
    const tts_geom &g = p->g;
    for (int i = 0; i < n_frames; ++i) {
        if (src[i].mem != TTS_MEM_DEVICE || dst[i].mem != TTS_MEM_DEVICE) { g_last_error = "rgb24 in: device frames only (host buffers: tts_run_host)"; return TTS_ERR_UNSUPPORTED; }
        if (src[i].width != g.in_w || src[i].height != g.in_h || dst[i].width != g.out_w || dst[i].height != g.out_h) return TTS_ERR_INVALID_INPUT;
        if (!src[i].plane[0] || src[i].stride[0] < g.in_w * tts::packed_pixel_bytes(g.in_fmt)) return TTS_ERR_INVALID_INPUT;
        for (int k = 0; k < 3; ++k)
            if (!dst[i].plane[k] || dst[i].stride[k] < ceil_shift(g.out_w, k ? 1 : 0)) return TTS_ERR_INVALID_INPUT;
        if (dst[i].stride[1] != dst[i].stride[2]) return TTS_ERR_UNSUPPORTED;
    }
    if (const int drc = check_device(p)) return drc;
    // `only` < 0: the three destination planes of a frame share one base; `only` = k: plane k alone (planes that lie
    // more than kMaxSpan apart -- every plane its own far-away allocation -- go one plane class of work at a time)
    auto run_groups = [&](int only) -> int {
        const int k0 = only < 0 ? 0 : only, k1 = only < 0 ? 3 : only + 1;
        int f0 = 0;
        while (f0 < n_frames) {
            uintptr_t smin = UINTPTR_MAX, dmin = UINTPTR_MAX, smax = 0, dmax = 0;
            int n = 0;
            while (f0 + n < n_frames && n < kMaxTableFrames) {
                const tts_frame &sf = src[f0 + n], &df = dst[f0 + n];
                if (n && (sf.stride[0] != src[f0].stride[0] || df.stride[0] != dst[f0].stride[0] || df.stride[1] != dst[f0].stride[1])) break;
                const uintptr_t a = std::min(smin, uintptr_t(sf.plane[0]));
                const uintptr_t b = std::max(smax, uintptr_t(sf.plane[0]) + size_t(g.in_h - 1) * size_t(sf.stride[0]) + size_t(g.in_w) * tts::packed_pixel_bytes(g.in_fmt));
                uintptr_t c2 = dmin, e = dmax;
                for (int k = k0; k < k1; ++k) {
                    const int ci = k ? 1 : 0;
                    c2 = std::min(c2, uintptr_t(df.plane[k]));
                    e = std::max(e, uintptr_t(df.plane[k]) + size_t(df.stride[k]) * (ceil_shift(g.out_h, ci) - 1) + ceil_shift(g.out_w, ci));
                }
                if (b - a > kMaxSpan || e - c2 > kMaxSpan) break;
                smin = a; smax = b; dmin = c2; dmax = e;
                ++n;
            }
            if (n == 0) return TTS_ERR_UNSUPPORTED;
            FrameAddr addr{};
            addr.use_table = 1;
            for (int i = 0; i < n; ++i) {
                addr.src_tab[0][i] = uint32_t(uintptr_t(src[f0 + i].plane[0]) - smin);
                for (int k = k0; k < k1; ++k) addr.dst_tab[k][i] = uint32_t(uintptr_t(dst[f0 + i].plane[k]) - dmin);
            }
            const int dstride[2] = {dst[f0].stride[0], dst[f0].stride[1]};
            const int rc = tts::rgbin_launch(p->rgbin, reinterpret_cast<const uint8_t *>(smin), reinterpret_cast<uint8_t *>(dmin), n,
                                             src[f0].stride[0], dstride, addr, static_cast<hipStream_t>(hip_stream), only);
            if (rc) return rc;
            if (const int rrc = rgbin_range_pass(p, reinterpret_cast<uint8_t *>(dmin), n, dstride, addr, static_cast<hipStream_t>(hip_stream), only)) return rrc;
            f0 += n;
        }
        return TTS_OK;
    };
    int rc = p->knobs.planewise ? TTS_ERR_UNSUPPORTED : run_groups(-1);   // (TTS_PLANEWISE: tests force the fallback)
    if (rc == TTS_ERR_UNSUPPORTED)                       // (a repeat of frames already done is harmless: same inputs, same outputs)
        for (int k = 0; k < 3 && (rc == TTS_ERR_UNSUPPORTED || rc == TTS_OK); ++k) rc = run_groups(k);
    if (rc == TTS_ERR_UNSUPPORTED) g_last_error = "rgb24 in: a plane does not fit 32-bit offsets from a common base";
    return rc;
}

}  // namespace

extern "C" {

int tts_run_batch(const tts_plan *p, const void *src, size_t src_pitch, void *dst, size_t dst_pitch,
                  int n_frames, void *hip_stream)
{
    if (!p) return TTS_ERR_INVALID_INPUT;
    const int bpp = p->g.out_fmt == TTS_FMT_YUV420P ? 1 : p->g.out_fmt >= TTS_FMT_RGBA ? 4 : 3;
    const int sstride[2] = {p->g.in_fmt != TTS_FMT_YUV420P ? p->g.in_w * tts::packed_pixel_bytes(p->g.in_fmt) : p->g.in_w, ceil_shift(p->g.in_w, 1)}, dstride[2] = {p->g.out_w * bpp, ceil_shift(p->g.out_w, 1)};
    return run_batch_impl(p, src, src_pitch, sstride, dst, dst_pitch, dstride, n_frames, hip_stream);
}

int tts_run_batch_strided(const tts_plan *p, const void *src, size_t src_frame_pitch, int src_stride_y, int src_stride_c,
                          void *dst, size_t dst_frame_pitch, int dst_stride_y, int dst_stride_c, int n_frames, void *hip_stream)
{
    const int sstride[2] = {src_stride_y, src_stride_c}, dstride[2] = {dst_stride_y, dst_stride_c};
    return run_batch_impl(p, src, src_frame_pitch, sstride, dst, dst_frame_pitch, dstride, n_frames, hip_stream);
}

static int run_host_frames(const tts_plan *p, const tts_frame *src, const tts_frame *dst, int n_frames);

int tts_run(const tts_plan *p, const tts_frame *src, const tts_frame *dst, int n_frames, void *hip_stream)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    const tts_geom &g = p->g;
    if (p->rgbin) return run_rgbin_frames(p, src, dst, n_frames, hip_stream);
    const int bpp = g.out_fmt == TTS_FMT_YUV420P ? 0 : g.out_fmt >= TTS_FMT_RGBA ? 4 : 3;
    const int ndp = bpp ? 1 : 3;                       // destination planes: packed RGB frames have one
    // Host frames (TTS_MEM_HOST, every frame of the call, both sides): staged through the per-device pool
    // of tts_run_host, synchronous, on the pool's own streams (hip_stream is not used).
    int n_host = 0;
    for (int i = 0; i < n_frames; ++i) n_host += (src[i].mem == TTS_MEM_HOST) + (dst[i].mem == TTS_MEM_HOST);
    if (n_host == 2 * n_frames) return run_host_frames(p, src, dst, n_frames);
    for (int i = 0; i < n_frames; ++i) {
        if (src[i].mem != TTS_MEM_DEVICE || dst[i].mem != TTS_MEM_DEVICE) { g_last_error = "a call takes device frames or host frames, not both"; return TTS_ERR_INVALID_INPUT; }
        if (src[i].width != g.in_w || src[i].height != g.in_h || dst[i].width != g.out_w || dst[i].height != g.out_h) return TTS_ERR_INVALID_INPUT;
        for (int k = 0; k < 3; ++k) {
            if (!src[i].plane[k] || (k < ndp && !dst[i].plane[k])) return TTS_ERR_INVALID_INPUT;
            if (src[i].stride[k] < ceil_shift(g.in_w, k ? 1 : 0)) return TTS_ERR_INVALID_INPUT;
            if (k < ndp && dst[i].stride[k] < (bpp ? g.out_w * bpp : ceil_shift(g.out_w, k ? 1 : 0))) return TTS_ERR_INVALID_INPUT;
        }
        if (src[i].stride[1] != src[i].stride[2] || (!bpp && dst[i].stride[1] != dst[i].stride[2])) return TTS_ERR_UNSUPPORTED;
    }
    if (const int drc = check_device(p)) return drc;
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
                          d.stride[0] != dst[f0].stride[0] || (!bpp && d.stride[1] != dst[f0].stride[1]))) break;
                uintptr_t a = smin, b = smax, c2 = dmin, e = dmax;
                for (int k = k0; k < k1; ++k) {
                    const int ci = k ? 1 : 0;
                    const size_t sext = size_t(s.stride[k]) * (ceil_shift(g.in_h, ci) - 1) + ceil_shift(g.in_w, ci);
                    a = std::min(a, uintptr_t(s.plane[k])); b = std::max(b, uintptr_t(s.plane[k]) + sext);
                    if (k >= ndp) continue;
                    const size_t dext = bpp ? size_t(d.stride[0]) * (g.out_h - 1) + size_t(g.out_w) * bpp
                                            : size_t(d.stride[k]) * (ceil_shift(g.out_h, ci) - 1) + ceil_shift(g.out_w, ci);
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
                    const size_t so = size_t(uintptr_t(src[f0 + i].plane[k]) - smin);
                    addr.src_tab[k][i] = uint32_t(so) + uint32_t(g.pre_y >> ci) * sstride[ci] + uint32_t(g.pre_x >> ci);
                    if (k >= ndp) continue;
                    const size_t dof = size_t(uintptr_t(dst[f0 + i].plane[k]) - dmin);
                    addr.dst_tab[k][i] = bpp ? uint32_t(dof) : uint32_t(dof) + uint32_t(c.pad_y) * dstride[ci] + uint32_t(c.pad_x);
                }
            // packed-RGB shrinks: the planar pass of p->scratch into the frames' own rows, then the in-place conversion
            // (see tts_plan); planes the column walk cannot stage through per-frame pointers fall back below
            tts::RgbScratchLayout lay{};
            if (p->rgb && p->scratch && only < 0 && tts::rgb_scratch_layout(g, dstride[0], lay)) {
                FrameAddr sa = addr;
                for (int i = 0; i < n; ++i)
                    for (int k = 0; k < 3; ++k) sa.dst_tab[k][i] = addr.dst_tab[0][i] + lay.off[k];
                const int pds[2] = {lay.stride[0], lay.stride[1]};
                int rc = launch_all(p->scratch, reinterpret_cast<const uint8_t *>(smin), reinterpret_cast<uint8_t *>(dmin), n,
                                    sstride, pds, uint32_t(smax - smin), uint32_t(dmax - dmin), sa, st, -1);
                if (rc == TTS_OK) rc = tts::rgb_scratch_convert(p->rgb, reinterpret_cast<uint8_t *>(dmin), dstride[0], n, addr, st);
                if (rc != TTS_ERR_UNSUPPORTED) { if (rc) return rc; f0 += n; continue; }
            }
            const int rc = launch_all(p, reinterpret_cast<const uint8_t *>(smin), reinterpret_cast<uint8_t *>(dmin), n,
                                      sstride, dstride, uint32_t(smax - smin), uint32_t(dmax - dmin), addr, st, only);
            if (rc) return rc;
            f0 += n;
        }
        return TTS_OK;
    };
    int rc = p->knobs.planewise && !bpp ? TTS_ERR_UNSUPPORTED : run_groups(0, n_frames, -1);
    if (rc == TTS_ERR_UNSUPPORTED && !bpp) {
        // some frame's planes are spread over more than kMaxSpan: go plane by plane (a repeat of the
        // frames already done above is harmless -- same inputs, same outputs)
        for (int k = 0; k < 3 && (rc == TTS_ERR_UNSUPPORTED || rc == TTS_OK); ++k) rc = run_groups(0, n_frames, k);
    }
    return rc;
}

}  // extern "C"

namespace {

// ---- staging for callers that hold host memory -----------------------------------------------------
// tts_run_host used to hipMalloc four buffers and create two streams on every call (one thumbnail per
// call through the Node addon paid ~1 ms of allocation for microseconds of work).  The buffers and
// streams now live in a per-device pool that grows to the largest chunk seen and is reused by every later
// call on that device; tts_trim() gives it back.  One call at a time per device holds the pool (host
// calls are synchronous and share one PCIe link anyway).
struct HostPool {
    std::mutex mu;
    void *src[2] = {nullptr, nullptr}, *dst[2] = {nullptr, nullptr};
    size_t src_cap[2] = {0, 0}, dst_cap[2] = {0, 0};
    hipStream_t st[2] = {nullptr, nullptr};
    bool reserve(int slot, size_t sbytes, size_t dbytes)
    {
        auto grow = [](void *&ptr, size_t &cap, size_t want) {
            if (want <= cap) return true;
            (void)hipFree(ptr); ptr = nullptr; cap = 0;
            if (hipMalloc(&ptr, want) != hipSuccess) return false;
            cap = want;
            return true;
        };
        if (!st[slot] && hipStreamCreateWithFlags(&st[slot], hipStreamNonBlocking) != hipSuccess) return false;
        return grow(src[slot], src_cap[slot], sbytes) && grow(dst[slot], dst_cap[slot], dbytes);
    }
    void release()
    {
        for (int s = 0; s < 2; ++s) {
            (void)hipFree(src[s]); (void)hipFree(dst[s]);
            src[s] = dst[s] = nullptr; src_cap[s] = dst_cap[s] = 0;
            if (st[s]) (void)hipStreamDestroy(st[s]);
            st[s] = nullptr;
        }
    }
};
constexpr int kMaxDevices = 64;
HostPool g_host_pool[kMaxDevices];

// ---- the scratch of tts_run_host_static ------------------------------------------------------------
// Device memory that belongs to the library's code object: every GPU the library runs on holds its own
// copy from module load until the process ends.  In host code the array's NAME is a host-side stand-in,
// not a device address: only the pointer hipGetSymbolAddress returns -- asked once per device, with that
// device current -- may be handed to kernels and copies.
constexpr size_t kScratchBytes = size_t(64) << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
std::mutex g_scratch_mu;
unsigned char *g_scratch_ptr[kMaxDevices] = {};

int scratch_for_device(int dev, unsigned char **out)
{
    if (dev < 0 || dev >= kMaxDevices) { g_last_error = "device index out of range"; return TTS_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));       // (the caller has made `dev` current)
        if (e != hipSuccess || !ptr) {
            g_last_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
            return TTS_ERR_INITIALIZATION_FAILED;
        }
        g_scratch_ptr[dev] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return TTS_OK;
}

}  // namespace

extern "C" {

int tts_run_host_static(const tts_plan *p, const void *src, void *dst, int n_frames, void *hip_stream)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    if (const int drc = check_device(p)) return drc;
    unsigned char *base = nullptr;
    if (const int rc = scratch_for_device(p->device, &base)) return rc;
    const size_t sb = tts_plan_frame_bytes(p, 0), db = tts_plan_frame_bytes(p, 1);
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    if (up(sb) + up(db) > kScratchBytes) {
        g_last_error = "one frame in (" + std::to_string(sb) + " B) and out (" + std::to_string(db) + " B) exceeds the " +
                       std::to_string(kScratchBytes >> 20) + " MiB static scratch of tts_run_host_static";
        return TTS_ERR_UNSUPPORTED;
    }
    // frames per pass: the source chunk at the start of the array, the output chunk behind it
    const size_t per = std::min<size_t>(size_t(n_frames), std::max<size_t>(1, (kScratchBytes - 256) / (sb + db)));
    unsigned char *dsrc = base, *ddst = base + up(per * sb);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    for (size_t f0 = 0; f0 < size_t(n_frames); f0 += per) {
        const size_t cnt = std::min(per, size_t(n_frames) - f0);
        if (hipMemcpyAsync(dsrc, static_cast<const uint8_t *>(src) + f0 * sb, sb * cnt, hipMemcpyHostToDevice, st) != hipSuccess) {
            g_last_error = "hipMemcpyAsync (host to scratch) failed"; return TTS_ERR_PROCESSING_FAILED;
        }
        if (const int rc = tts_run_batch(p, dsrc, sb, ddst, db, int(cnt), st)) return rc;
        if (hipMemcpyAsync(static_cast<uint8_t *>(dst) + f0 * db, ddst, db * cnt, hipMemcpyDeviceToHost, st) != hipSuccess) {
            g_last_error = "hipMemcpyAsync (scratch to host) failed"; return TTS_ERR_PROCESSING_FAILED;
        }
    }
    return TTS_OK;
}

void tts_trim(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return;
    HostPool &pool = g_host_pool[dev];
    std::lock_guard<std::mutex> lock(pool.mu);
    pool.release();
}

// tts_run with TTS_MEM_HOST frames: plane by plane (rows may be padded) into packed frames in the pool's
// device buffers, one batch launch set, plane by plane back.
static int run_host_frames(const tts_plan *p, const tts_frame *src, const tts_frame *dst, int n_frames)
{
    const tts_geom &g = p->g;
    const int bpp = g.out_fmt == TTS_FMT_YUV420P ? 0 : g.out_fmt >= TTS_FMT_RGBA ? 4 : 3;
    const int ndp = bpp ? 1 : 3;
    for (int i = 0; i < n_frames; ++i) {
        if (src[i].width != g.in_w || src[i].height != g.in_h || dst[i].width != g.out_w || dst[i].height != g.out_h) return TTS_ERR_INVALID_INPUT;
        for (int k = 0; k < 3; ++k) {
            if (!src[i].plane[k] || src[i].stride[k] < ceil_shift(g.in_w, k ? 1 : 0)) return TTS_ERR_INVALID_INPUT;
            if (k < ndp && (!dst[i].plane[k] || dst[i].stride[k] < (bpp ? g.out_w * bpp : ceil_shift(g.out_w, k ? 1 : 0)))) return TTS_ERR_INVALID_INPUT;
        }
    }
    if (const int drc = check_device(p)) return drc;
    if (p->device < 0 || p->device >= kMaxDevices) return TTS_ERR_UNSUPPORTED;
    const size_t sb = tts_plan_frame_bytes(p, 0), db = tts_plan_frame_bytes(p, 1);
    const int per = int(std::max<size_t>(1, std::min<size_t>(size_t(n_frames), (size_t(p->knobs.host_chunk_mb) << 20) / (sb + db))));
    HostPool &pool = g_host_pool[p->device];
    std::lock_guard<std::mutex> lock(pool.mu);
    if (!pool.reserve(0, sb * per, db * per)) { g_last_error = "staging pool: hipMalloc / hipStreamCreate failed"; return TTS_ERR_INITIALIZATION_FAILED; }
    hipStream_t st = pool.st[0];
    uint8_t *ds = static_cast<uint8_t *>(pool.src[0]), *dd = static_cast<uint8_t *>(pool.dst[0]);
    int rc = TTS_OK;
    for (int f0 = 0; f0 < n_frames && !rc; f0 += per) {
        const int cnt = std::min(per, n_frames - f0);
        for (int i = 0; i < cnt && !rc; ++i) {
            size_t off = 0;
            for (int k = 0; k < 3 && !rc; ++k) {
                const int w = ceil_shift(g.in_w, k ? 1 : 0), h = ceil_shift(g.in_h, k ? 1 : 0);
                if (hipMemcpy2DAsync(ds + size_t(i) * sb + off, size_t(w), src[f0 + i].plane[k], size_t(src[f0 + i].stride[k]), size_t(w), size_t(h),
                                     hipMemcpyHostToDevice, st) != hipSuccess) rc = TTS_ERR_PROCESSING_FAILED;
                off += size_t(w) * h;
            }
        }
        if (!rc) rc = tts_run_batch(p, ds, sb, dd, db, cnt, st);
        for (int i = 0; i < cnt && !rc; ++i) {
            size_t off = 0;
            for (int k = 0; k < ndp && !rc; ++k) {
                const size_t w = bpp ? size_t(g.out_w) * bpp : size_t(ceil_shift(g.out_w, k ? 1 : 0));
                const size_t h = bpp ? size_t(g.out_h) : size_t(ceil_shift(g.out_h, k ? 1 : 0));
                if (hipMemcpy2DAsync(dst[f0 + i].plane[k], size_t(dst[f0 + i].stride[k]), dd + size_t(i) * db + off, w, w, h,
                                     hipMemcpyDeviceToHost, st) != hipSuccess) rc = TTS_ERR_PROCESSING_FAILED;
                off += w * h;
            }
        }
        // (one slot: the next chunk's uploads must not overtake this chunk's downloads; same stream, so they cannot)
    }
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = TTS_ERR_PROCESSING_FAILED;
    if (rc == TTS_ERR_PROCESSING_FAILED && g_last_error.empty()) g_last_error = "copy between host frames and the staging pool failed";
    return rc;
}

int tts_run_host(const tts_plan *p, const void *src, void *dst, int n_frames)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    if (const int drc = check_device(p)) return drc;
    const size_t sb = tts_plan_frame_bytes(p, 0), db = tts_plan_frame_bytes(p, 1);
    // The clip goes through in chunks over two slots (a source and a destination buffer in HBM each) on
    // two streams: chunk i + 1 is on its way up while chunk i is scaled and sent back, both directions of
    // the link busy, device memory bounded whatever the clip's length.  Buffers the caller has pinned
    // (hipHostMalloc / hipHostRegister) move by DMA at link speed -- 10,100 frames/s of 1080p -> portrait,
    // 31.6 GB/s each way at once, against 8,900 without the overlap; pageable ones are staged by the
    // runtime and run at host-memory speed (2,400 - 3,600 frames/s; registering them here for the call
    // cost more than it saved, so that is left to the caller).
    const size_t chunk_bytes = size_t(p->knobs.host_chunk_mb) << 20;
    const int per = int(std::max<size_t>(1, std::min<size_t>(size_t(n_frames), chunk_bytes / (sb + db))));
    const int nchunks = (n_frames + per - 1) / per;
    const int slots = std::min(2, nchunks);
    if (p->device < 0 || p->device >= kMaxDevices) return TTS_ERR_UNSUPPORTED;
    HostPool &pool = g_host_pool[p->device];
    std::lock_guard<std::mutex> lock(pool.mu);
    void **ds = pool.src, **dd = pool.dst;
    hipStream_t *st = pool.st;
    int rc = TTS_OK;
    for (int s = 0; s < slots && !rc; ++s)
        if (!pool.reserve(s, sb * per, db * per)) { g_last_error = "staging pool: hipMalloc / hipStreamCreate failed"; rc = TTS_ERR_INITIALIZATION_FAILED; }
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
