// ttscale -- raw-YUV stand-in for the `ffmpeg` executable on the scale/crop/pad step.
//
// The reference spawns `ffmpeg` with a filter chain (scripts/processVideos.js:71-108 through
// fluent-ffmpeg, src/services/video.js:329-334 through ffmpeg-kit).  Decode/encode are out of
// scope, so this CLI speaks rawvideo only:
//   ttscale -s 1920x1080 [-pix_fmt yuv420p|rgb24] [-f rawvideo] -i in.yuv -vf "<chain>" [-pix_fmt <fmt>] [-y] out.yuv
// As with ffmpeg, -pix_fmt in front of -i names the input's format; behind it, the output's: yuv420p, yuvj420p,
// rgb24, bgr24, rgba, bgra append a `format=` to the chain (packed RGB out).
// Exit status 0 on success; otherwise one line on stderr and a non-zero status, like ffmpeg.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ttscale.h"

static int fail(const char *what, int rc)
{
    std::fprintf(stderr, "ttscale: %s: %s%s%s\n", what, tts_strerror(rc), *tts_last_error() ? ": " : "", tts_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    std::string in, out, vf = "format=yuv420p", size, out_fmt;
    int in_fmt = TTS_FMT_YUV420P;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-s" || a == "-video_size") size = next();
        else if (a == "-i") in = next();
        else if (a == "-vf" || a == "-filter:v") vf = next();
        else if (a == "-pix_fmt") {
            const std::string f = next();
            if (in.empty()) {                                // in front of -i: the input
                if (f == "rgb24") in_fmt = TTS_FMT_RGB24;
                else if (f == "bgr24") in_fmt = TTS_FMT_BGR24;
                else if (f == "rgba") in_fmt = TTS_FMT_RGBA;
                else if (f == "bgra") in_fmt = TTS_FMT_BGRA;
                else if (f == "yuvj420p") in_fmt = TTS_FMT_YUVJ420P;   // the same planes, full-range samples
                else if (f != "yuv420p") { std::fprintf(stderr, "ttscale: input -pix_fmt yuv420p, yuvj420p, rgb24, bgr24, rgba or bgra\n"); return 1; }
            } else out_fmt = f;
        }
        else if (a == "-f") { if (std::strcmp(next(), "rawvideo")) { std::fprintf(stderr, "ttscale: only -f rawvideo\n"); return 1; } }
        else if (a == "-y" || a == "-an" || a == "-hide_banner") {}
        else if (a == "-version") { std::printf("%s\n", tts_version()); return 0; }
        else if (!a.empty() && a[0] != '-') out = a;
        else { std::fprintf(stderr, "ttscale: unsupported option %s\n", a.c_str()); return 1; }
    }
    int iw = 0, ih = 0;
    if (std::sscanf(size.c_str(), "%dx%d", &iw, &ih) != 2 || in.empty() || out.empty()) {
        std::fprintf(stderr, "usage: ttscale -s WxH -i in.yuv -vf \"<chain>\" out.yuv\n");
        return 1;
    }
    if (!out_fmt.empty()) vf += ",format=" + out_fmt;
    tts_geom g;
    int rc = tts_parse_vf_fmt(vf.c_str(), iw, ih, in_fmt, &g);
    if (rc) return fail("filter chain", rc);
    tts_plan *plan = nullptr;
    rc = tts_plan_create(&g, &plan);
    if (rc) return fail("plan", rc);

    const size_t sb = tts_plan_frame_bytes(plan, 0), db = tts_plan_frame_bytes(plan, 1);   // (packed RGB behind format=rgb24 & co.)
    FILE *fi = std::fopen(in.c_str(), "rb"), *fo = std::fopen(out.c_str(), "wb");
    if (!fi || !fo) { std::fprintf(stderr, "ttscale: cannot open %s\n", fi ? out.c_str() : in.c_str()); return 1; }
    const int batch = 64;
    std::vector<unsigned char> src(sb * batch), dst(db * batch);
    long frames = 0;
    for (;;) {
        const size_t got = std::fread(src.data(), 1, src.size(), fi);
        const int n = int(got / sb);
        if (n == 0) break;
        rc = tts_run_host(plan, src.data(), dst.data(), n);
        if (rc) return fail("run", rc);
        if (std::fwrite(dst.data(), db, n, fo) != size_t(n)) { std::fprintf(stderr, "ttscale: short write\n"); return 1; }
        frames += n;
        if (got < src.size()) break;
    }
    std::fclose(fi); std::fclose(fo);
    tts_plan_destroy(plan);
    std::fprintf(stderr, "ttscale: %ld frame(s) %dx%d -> %dx%d\n", frames, g.in_w, g.in_h, g.out_w, g.out_h);
    return 0;
}
