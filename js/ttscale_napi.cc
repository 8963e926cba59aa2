// ttscale.node -- N-API binding of libttscale.so for the reference's Node scripts.
//
// scripts/processVideos.js:3,71-108 drives the scale step through fluent-ffmpeg and a spawned
// `ffmpeg`; this addon puts the same step behind a function call on raw yuv420p buffers.
//   parseVf(vf, iw, ih)            -> { in_w, in_h, sc_w, ..., out_w, out_h }
//   frameBytes(w, h)               -> bytes of a packed I420 frame
//   runHost(vf, iw, ih, srcBuffer) -> Buffer with the processed frames (packed I420)
//   trimRange(nFrames, fpsNum, fpsDen, start, duration)             -> { first, count }   (duration < 0: to the end)
//   loopOrder(nFrames, fpsNum, fpsDen, start, duration, repeats)    -> [source frame of every output frame]
// Errors are thrown as Error whose message carries INVALID_INPUT / INVALID_FORMAT /
// INITIALIZATION_FAILED / PROCESSING_FAILED (web/src/tests/video/compression.test.js:62-90).
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <string>

#include "../include/ttscale.h"

namespace {

napi_value throw_status(napi_env env, const char *what, int rc)
{
    std::string msg = std::string(tts_strerror(rc)) + ": " + what;
    if (*tts_last_error()) msg += std::string(" (") + tts_last_error() + ")";
    napi_throw_error(env, tts_strerror(rc), msg.c_str());
    return nullptr;
}

bool get_string(napi_env env, napi_value v, std::string &out)
{
    size_t len = 0;
    if (napi_get_value_string_utf8(env, v, nullptr, 0, &len) != napi_ok) return false;
    out.resize(len);
    return napi_get_value_string_utf8(env, v, &out[0], len + 1, &len) == napi_ok;
}

bool parse_common(napi_env env, napi_callback_info info, size_t want, napi_value *argv, std::string &vf, int32_t &iw, int32_t &ih)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) return false;
    return get_string(env, argv[0], vf) && napi_get_value_int32(env, argv[1], &iw) == napi_ok &&
           napi_get_value_int32(env, argv[2], &ih) == napi_ok;
}

// optional trailing argument: the input's pixel format, as `-pix_fmt` in front of `-i` ("yuv420p", the default, or "rgb24")
bool get_in_fmt(napi_env env, napi_callback_info info, size_t index, int &in_fmt)
{
    size_t argc = 8; napi_value argv[8];
    in_fmt = TTS_FMT_YUV420P;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc <= index) return true;
    napi_valuetype t;
    if (napi_typeof(env, argv[index], &t) != napi_ok || t == napi_undefined || t == napi_null) return true;
    std::string f;
    if (!get_string(env, argv[index], f)) return false;
    if (f == "rgb24") in_fmt = TTS_FMT_RGB24;
    else if (f == "bgr24") in_fmt = TTS_FMT_BGR24;
    else if (f == "rgba") in_fmt = TTS_FMT_RGBA;
    else if (f == "bgra") in_fmt = TTS_FMT_BGRA;
    else if (f == "yuvj420p") in_fmt = TTS_FMT_YUVJ420P;   // the same planes, full-range samples
    else if (f != "yuv420p") return false;
    return true;
}

napi_value ParseVf(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    std::string vf; int32_t iw, ih;
    int in_fmt;
    if (!parse_common(env, info, 3, argv, vf, iw, ih) || !get_in_fmt(env, info, 3, in_fmt)) return throw_status(env, "parseVf(vf, iw, ih[, 'yuv420p' | 'rgb24'])", TTS_ERR_INVALID_INPUT);
    tts_geom g;
    const int rc = tts_parse_vf_fmt(vf.c_str(), iw, ih, in_fmt, &g);
    if (rc) return throw_status(env, "filter chain", rc);
    napi_value obj;
    napi_create_object(env, &obj);
    const char *names[] = {"in_w", "in_h", "pre_x", "pre_y", "pre_w", "pre_h", "do_scale", "alg", "sc_w", "sc_h",
                           "crop_x", "crop_y", "crop_w", "crop_h", "do_pad", "pad_x", "pad_y", "out_w", "out_h",
                           "fill_y", "fill_u", "fill_v", "out_range", "out_fmt", "rgb_stage", "fill_rgb", "in_fmt", "sc_range"};
    static_assert(sizeof(tts_geom) == sizeof(names) / sizeof(names[0]) * sizeof(int32_t), "names[] lists every field of tts_geom, in order");
    const int32_t *vals = &g.in_w;
    for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); ++i) {
        napi_value v;
        napi_create_int32(env, vals[i], &v);
        napi_set_named_property(env, obj, names[i], v);
    }
    return obj;
}

napi_value FrameBytes(napi_env env, napi_callback_info info)
{
    size_t argc = 2; napi_value argv[2]; int32_t w = 0, h = 0;
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    if (argc < 2 || napi_get_value_int32(env, argv[0], &w) != napi_ok || napi_get_value_int32(env, argv[1], &h) != napi_ok)
        return throw_status(env, "frameBytes(w, h)", TTS_ERR_INVALID_INPUT);
    napi_value out;
    napi_create_double(env, double(tts_frame_bytes(w, h)), &out);
    return out;
}

// Plans are kept between calls (a script converts many clips of a few geometries, the app one
// thumbnail after another): building one costs about a millisecond, a frame far less.  Node runs
// addon calls on one thread; the map is never shared.
std::map<std::string, tts_plan *> plans;

// videoProcessor.terminate() (web/src/tests/video/compression.test.js:27-29): give back what is cached between
// calls -- the plans (tables in HBM) and the library's host-path staging pool
napi_value Terminate(napi_env env, napi_callback_info)
{
    const size_t n = plans.size();
    for (auto &kv : plans) tts_plan_destroy(kv.second);
    plans.clear();
    tts_trim();
    napi_value out;
    napi_create_uint32(env, uint32_t(n), &out);
    return out;
}

tts_plan *cached_plan(const std::string &vf, int32_t iw, int32_t ih, const tts_geom &g, int *rc)
{
    const std::string key = vf + "|" + std::to_string(iw) + "x" + std::to_string(ih) + "|" + std::to_string(g.in_fmt);
    auto it = plans.find(key);
    if (it != plans.end()) { *rc = TTS_OK; return it->second; }
    if (plans.size() >= 32) {                       // bounded: drop everything rather than track ages
        for (auto &kv : plans) tts_plan_destroy(kv.second);
        plans.clear();
    }
    tts_plan *plan = nullptr;
    *rc = tts_plan_create(&g, &plan);
    if (*rc) return nullptr;
    plans[key] = plan;
    return plan;
}

napi_value RunHost(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    std::string vf; int32_t iw, ih;
    int in_fmt;
    if (!parse_common(env, info, 4, argv, vf, iw, ih) || !get_in_fmt(env, info, 4, in_fmt)) return throw_status(env, "runHost(vf, iw, ih, buffer[, 'yuv420p' | 'rgb24'])", TTS_ERR_INVALID_INPUT);
    void *src = nullptr; size_t len = 0;
    if (napi_get_buffer_info(env, argv[3], &src, &len) != napi_ok) return throw_status(env, "frames must be a Buffer", TTS_ERR_INVALID_INPUT);
    tts_geom g;
    int rc = tts_parse_vf_fmt(vf.c_str(), iw, ih, in_fmt, &g);
    if (rc) return throw_status(env, "filter chain", rc);
    tts_plan *plan = cached_plan(vf, iw, ih, g, &rc);
    if (rc) return throw_status(env, "plan", rc);
    const size_t sb = tts_plan_frame_bytes(plan, 0);      // packed I420, or in_w * in_h * 3 for an rgb24 source
    if (len == 0 || len % sb) return throw_status(env, "buffer is not a whole number of frames", TTS_ERR_INVALID_INPUT);
    const int n = int(len / sb);
    const size_t db = tts_plan_frame_bytes(plan, 1);      // packed I420, or out_w * out_h * bpp behind format=rgb24 & co.
    napi_value out; void *dst = nullptr;
    if (napi_create_buffer(env, db * n, &dst, &out) != napi_ok) return throw_status(env, "allocation", TTS_ERR_PROCESSING_FAILED);
    rc = tts_run_host(plan, src, dst, n);
    if (rc) return throw_status(env, "run", rc);
    return out;
}

// scripts/loopVideos.js:42-52 (`.seekInput(start).duration(d)`) and :54-100 (createLoopedVideo) as frame numbers
bool timeline_args(napi_env env, napi_callback_info info, size_t want, int32_t *ints, double *dbl)
{
    size_t argc = want; napi_value argv[6];
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) return false;
    for (int i = 0; i < 3; ++i) if (napi_get_value_int32(env, argv[i], &ints[i]) != napi_ok) return false;
    for (int i = 0; i < 2; ++i) if (napi_get_value_double(env, argv[3 + i], &dbl[i]) != napi_ok) return false;
    if (want > 5 && napi_get_value_int32(env, argv[5], &ints[3]) != napi_ok) return false;
    return true;
}

napi_value TrimRange(napi_env env, napi_callback_info info)
{
    int32_t a[4]; double d[2];
    if (!timeline_args(env, info, 5, a, d)) return throw_status(env, "trimRange(nFrames, fpsNum, fpsDen, start, duration)", TTS_ERR_INVALID_INPUT);
    int first = 0, count = 0;
    const int rc = tts_trim_range(a[0], a[1], a[2], d[0], d[1], &first, &count);
    if (rc) return throw_status(env, "trim range", rc);
    napi_value obj, v;
    napi_create_object(env, &obj);
    napi_create_int32(env, first, &v); napi_set_named_property(env, obj, "first", v);
    napi_create_int32(env, count, &v); napi_set_named_property(env, obj, "count", v);
    return obj;
}

napi_value LoopOrder(napi_env env, napi_callback_info info)
{
    int32_t a[4]; double d[2];
    if (!timeline_args(env, info, 6, a, d)) return throw_status(env, "loopOrder(nFrames, fpsNum, fpsDen, start, duration, repeats)", TTS_ERR_INVALID_INPUT);
    const int n = tts_loop_order(a[0], a[1], a[2], d[0], d[1], a[3], nullptr, 0);
    if (n < 0) return throw_status(env, "loop order", n);
    std::string buf(size_t(n) * sizeof(int32_t), '\0');
    int32_t *order = reinterpret_cast<int32_t *>(&buf[0]);
    const int got = tts_loop_order(a[0], a[1], a[2], d[0], d[1], a[3], order, n);
    if (got < 0) return throw_status(env, "loop order", got);
    napi_value arr;
    napi_create_array_with_length(env, size_t(got), &arr);
    for (int i = 0; i < got; ++i) {
        napi_value v;
        napi_create_int32(env, order[i], &v);
        napi_set_element(env, arr, uint32_t(i), v);
    }
    return arr;
}

napi_value Init(napi_env env, napi_value exports)
{
    napi_property_descriptor props[] = {
        {"parseVf", nullptr, ParseVf, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"frameBytes", nullptr, FrameBytes, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"runHost", nullptr, RunHost, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"trimRange", nullptr, TrimRange, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"loopOrder", nullptr, LoopOrder, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"terminate", nullptr, Terminate, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
    return exports;
}

}  // namespace

NAPI_MODULE(ttscale, Init)
