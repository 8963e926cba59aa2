// Host-only robustness check of the filter-string parser and the tap-table builder: random and mutated -vf
// strings, random frame sizes, under AddressSanitizer + UBSan (the GPU pool has no device sanitizer; this code
// runs on the host in the product too).  Whatever parse_vf accepts must pass check_geom and build tap tables of
// sane sizes; whatever it rejects must name a reason.
// Build + run: tools/hostfuzz/run.sh [iterations]
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../tiktoken_amd/csrc/filter.h"

int main(int argc, char **argv)
{
    const long iters = argc > 1 ? std::atol(argv[1]) : 200000;
    std::mt19937_64 rng(0x5EED);
    const std::vector<std::string> seeds = {
        "scale=w=720:h=1280",
        "scale=320x180:force_original_aspect_ratio=decrease,pad=320x180:-1:-1:color=black,format=yuv420p",
        "scale='min(720,iw)':'min(1280,ih)':force_original_aspect_ratio=decrease",
        "scale=1080:1920:force_original_aspect_ratio=increase,crop=1080:1920",
        "scale=320:568:flags=area", "crop=100:80:11:7,scale=200:160,format=rgb24", "format=yuvj420p",
        "scale=iw/2:-2:force_divisible_by=4,pad=iw+20:ih+20:(ow-iw)/2:(oh-ih)/2:color=red,format=bgra",
        "scale=trunc(iw*1.5/2)*2:ih*2:flags=bilinear,crop=in_w-10:in_h-10",
    };
    const char alphabet[] = "0123456789:=,x*/+-()'._abcdefghijklmnopqrstuvwxyzWH \t";
    long accepted = 0, rejected = 0;
    for (long it = 0; it < iters; ++it) {
        std::string s = seeds[rng() % seeds.size()];
        const int edits = int(rng() % 6);
        for (int e = 0; e < edits && !s.empty(); ++e) {
            const size_t at = rng() % s.size();
            switch (rng() % 5) {
            case 0: s.erase(at, 1 + rng() % 3); break;
            case 1: s.insert(at, 1, alphabet[rng() % (sizeof alphabet - 1)]); break;
            case 2: s[at] = alphabet[rng() % (sizeof alphabet - 1)]; break;
            case 3: s.insert(at, std::to_string(int64_t(rng() % 5 ? rng() % 5000 : rng()))); break;
            default: s.insert(at, seeds[rng() % seeds.size()].substr(0, rng() % 12)); break;
            }
        }
        if (rng() % 50 == 0) { s.assign(rng() % 40, 'x'); for (auto &c : s) c = alphabet[rng() % (sizeof alphabet - 1)]; }
        const int iw = int(rng() % 3 ? 2 + rng() % 4000 : rng() % 70000) , ih = int(rng() % 3 ? 2 + rng() % 2400 : rng() % 70000);
        const int in_fmt = rng() % 5 == 0 ? TTS_FMT_RGB24 : TTS_FMT_YUV420P;
        tts_geom g{};
        std::string err;
        const int rc = tts::parse_vf(s.c_str(), iw, ih, g, err, in_fmt);
        if (rc != TTS_OK) {
            ++rejected;
            if (err.empty() && rc != TTS_ERR_INVALID_INPUT && rc != TTS_ERR_UNSUPPORTED) { std::printf("rejected without a reason (%d): %s\n", rc, s.c_str()); return 1; }
            continue;
        }
        ++accepted;
        if (tts::check_geom(g) != TTS_OK) { std::printf("accepted by the parser, refused by check_geom: %s (%dx%d)\n", s.c_str(), iw, ih); return 1; }
        if (g.out_w < 1 || g.out_h < 1 || g.out_w > 16384 || g.out_h > 16384) { std::printf("output %dx%d from %s\n", g.out_w, g.out_h, s.c_str()); return 1; }
        if (it % 16 == 0 && g.do_scale && int64_t(g.pre_w) * g.sc_w < (int64_t(1) << 26)) {   // tap tables of the accepted geometry
            tts::AxisFilter f;
            if (tts::build_axis_filter(g.pre_w, g.sc_w, 1 << 14, g.alg, f)) {
                if (f.size < 1 || f.pos.size() != size_t(g.sc_w) || f.taps.size() != size_t(g.sc_w) * f.size) { std::printf("tap table shape: %s\n", s.c_str()); return 1; }
                for (int i = 0; i < g.sc_w; ++i) {
                    int sum = 0;
                    for (int k = 0; k < f.size; ++k) sum += f.taps[size_t(i) * f.size + k];
                    if (sum != (1 << 14) || f.pos[i] < 0 || f.pos[i] + f.size > std::max(g.pre_w, f.size)) { std::printf("tap row %d of %s: sum %d pos %d size %d\n", i, s.c_str(), sum, f.pos[i], f.size); return 1; }
                }
            }
        }
    }
    std::printf("parse_fuzz: %ld strings, %ld accepted, %ld rejected, no finding\n", iters, accepted, rejected);
    return 0;
}
