// Time-domain steps around the per-frame path: which source frame each output frame comes from.
//
// The reference trims clips (`-ss`/`-t`: scripts/processVideos.js:84-92) and extends them by repeating
// a middle segment (scripts/loopVideos.js:54-100: three `-ss/-t` extractions, then the concat
// demuxer).  Neither touches a pixel -- the `crossfade: 0.5` of loopVideos.js:20,31 is declared and
// never read (:55 destructures only extractStart, extractDuration, repeats) -- so here they are an
// index list, and the scaler consumes the list directly: a looped or trimmed clip costs no copy.
// Host code only; no device call in this file.
#include <cmath>
#include <cstdint>

#include "../../include/ttscale.h"

namespace {

// First frame whose timestamp i * den / num is >= t (ffmpeg keeps a frame of a `-ss t` input when its
// pts is not before t, and stops a `-t d` output at the first frame whose pts - start reaches d).
int64_t first_frame_at(double t, int num, int den)
{
    if (t <= 0) return 0;
    const double x = t * double(num) / double(den);
    if (!(x < 4e18)) return INT64_MAX;                         // (infinite or absurd times: past any clip, and no out-of-range cast)
    return int64_t(std::ceil(x - 1e-9));
}

}  // namespace

extern "C" {

int tts_trim_range(int n_frames, int fps_num, int fps_den, double start, double duration, int *first, int *count)
{
    if (n_frames < 0 || fps_num < 1 || fps_den < 1 || !first || !count || !(start >= 0) || std::isnan(duration)) return TTS_ERR_INVALID_INPUT;
    int64_t a = first_frame_at(start, fps_num, fps_den);
    int64_t b = duration < 0 ? n_frames : first_frame_at(start + duration, fps_num, fps_den);
    if (a > n_frames) a = n_frames;
    if (b > n_frames) b = n_frames;
    if (b < a) b = a;
    *first = int(a);
    *count = int(b - a);
    return TTS_OK;
}

int tts_loop_order(int n_frames, int fps_num, int fps_den, double extract_start, double extract_duration,
                   int repeats, int32_t *order, int cap)
{
    if (repeats < 0 || !(extract_duration >= 0)) return TTS_ERR_INVALID_INPUT;
    int first = 0, count = 0;
    const int rc = tts_trim_range(n_frames, fps_num, fps_den, extract_start, extract_duration, &first, &count);
    if (rc) return rc;
    // initial [0, first) + repeats x [first, first+count) + end [first+count, n)   (loopVideos.js:77-81)
    const int64_t total = int64_t(first) + int64_t(repeats) * count + (n_frames - first - count);
    if (total > INT32_MAX) return TTS_ERR_UNSUPPORTED;
    if (!order) return int(total);
    if (cap < total) return TTS_ERR_INVALID_INPUT;
    int n = 0;
    for (int i = 0; i < first; ++i) order[n++] = i;
    for (int r = 0; r < repeats; ++r)
        for (int i = 0; i < count; ++i) order[n++] = first + i;
    for (int i = first + count; i < n_frames; ++i) order[n++] = i;
    return n;
}

int tts_run_batch_order(const tts_plan *p, const void *src, size_t src_frame_pitch, int n_src_frames,
                        void *dst, size_t dst_frame_pitch, const int32_t *order, int n_out, void *hip_stream)
{
    if (!p || n_out < 0 || n_src_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_out == 0) return TTS_OK;
    if (!src || !dst || !order) return TTS_ERR_INVALID_INPUT;
    for (int j = 0; j < n_out; ++j)
        if (order[j] < 0 || order[j] >= n_src_frames) return TTS_ERR_INVALID_INPUT;
    // maximal runs of consecutive source frames: each is one ordinary batch launch set
    for (int j = 0; j < n_out;) {
        int len = 1;
        while (j + len < n_out && order[j + len] == order[j] + len) ++len;
        const int rc = tts_run_batch(p, static_cast<const uint8_t *>(src) + size_t(order[j]) * src_frame_pitch, src_frame_pitch,
                                     static_cast<uint8_t *>(dst) + size_t(j) * dst_frame_pitch, dst_frame_pitch, len, hip_stream);
        if (rc) return rc;
        j += len;
    }
    return TTS_OK;
}

}  // extern "C"
