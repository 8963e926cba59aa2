// "-vf" chain parser for libttscale.
//
// Accepts the filter strings the reference builds for the external ffmpeg:
//   scripts/processVideos.js:94-97   .size('720x1280')  (fluent-ffmpeg emits scale=w=720:h=1280)
//   src/services/video.js:329-332    scale=320x180:force_original_aspect_ratio=decrease,
//                                    pad=320x180:-1:-1:color=black,format=yuv420p
//   web/src/tests/video/portrait.test.js:78,95-96,146   scale=W:H with min(720,..)/min(1280,..)
// and resolves them with libavfilter's sizing rules into a tts_geom.
#include "filter.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <map>
#include <stdexcept>

namespace tts {
namespace {

using Vars = std::map<std::string, double>;

// Small arithmetic evaluator: + - * /, unary minus, parentheses, named values and
// the handful of functions that appear in sizing expressions.
class Expr {
public:
    Expr(const std::string &s, const Vars &v) : s_(s), v_(v) {}
    double run()
    {
        double r = sum();
        skip();
        if (i_ != s_.size()) throw std::runtime_error("trailing characters in expression '" + s_ + "'");
        return r;
    }

private:
    const std::string &s_;
    const Vars &v_;
    size_t i_ = 0;

    void skip() { while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_; }
    bool eat(char c) { skip(); if (i_ < s_.size() && s_[i_] == c) { ++i_; return true; } return false; }

    double sum()
    {
        double a = product();
        for (;;) {
            if (eat('+')) a += product();
            else if (eat('-')) a -= product();
            else return a;
        }
    }
    double product()
    {
        double a = unary();
        for (;;) {
            if (eat('*')) a *= unary();
            else if (eat('/')) a /= unary();
            else return a;
        }
    }
    double unary()
    {
        if (eat('-')) return -unary();
        if (eat('+')) return unary();
        return atom();
    }
    double atom()
    {
        skip();
        if (eat('(')) { double r = sum(); if (!eat(')')) throw std::runtime_error("missing ')'"); return r; }
        if (i_ < s_.size() && (std::isdigit((unsigned char)s_[i_]) || s_[i_] == '.')) {
            size_t used = 0;
            double r = std::stod(s_.substr(i_), &used);
            i_ += used;
            return r;
        }
        size_t b = i_;
        while (i_ < s_.size() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '_')) ++i_;
        if (b == i_) throw std::runtime_error("unexpected character in expression '" + s_ + "'");
        const std::string name = s_.substr(b, i_ - b);
        if (eat('(')) {
            std::vector<double> a;
            if (!eat(')')) {
                do a.push_back(sum()); while (eat(','));
                if (!eat(')')) throw std::runtime_error("missing ')' after " + name);
            }
            return call(name, a);
        }
        auto it = v_.find(name);
        if (it == v_.end()) throw std::runtime_error("unknown name '" + name + "'");
        return it->second;
    }
    static double call(const std::string &f, const std::vector<double> &a)
    {
        auto need = [&](size_t n) { if (a.size() != n) throw std::runtime_error(f + "() takes " + std::to_string(n) + " arguments"); };
        if (f == "min")   { need(2); return std::fmin(a[0], a[1]); }
        if (f == "max")   { need(2); return std::fmax(a[0], a[1]); }
        if (f == "trunc") { need(1); return std::trunc(a[0]); }
        if (f == "floor") { need(1); return std::floor(a[0]); }
        if (f == "ceil")  { need(1); return std::ceil(a[0]); }
        if (f == "round") { need(1); return std::round(a[0]); }
        if (f == "abs")   { need(1); return std::fabs(a[0]); }
        if (f == "gt")    { need(2); return a[0] >  a[1]; }
        if (f == "gte")   { need(2); return a[0] >= a[1]; }
        if (f == "lt")    { need(2); return a[0] <  a[1]; }
        if (f == "lte")   { need(2); return a[0] <= a[1]; }
        if (f == "eq")    { need(2); return a[0] == a[1]; }
        if (f == "mod")   { need(2); return std::fmod(a[0], a[1]); }
        if (f == "if")    { if (a.size() == 2) return a[0] != 0 ? a[1] : 0; need(3); return a[0] != 0 ? a[1] : a[2]; }
        throw std::runtime_error("unknown function '" + f + "'");
    }
};

double eval(const std::string &s, const Vars &v) { return Expr(s, v).run(); }

// Split on `sep` outside quotes and parentheses; strips quotes and backslash escapes.
std::vector<std::string> split(const std::string &s, char sep)
{
    std::vector<std::string> out(1);
    int depth = 0;
    bool quoted = false;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (c == '\\' && i + 1 < s.size()) { out.back() += s[++i]; continue; }
        if (c == '\'') { quoted = !quoted; continue; }
        if (!quoted) {
            if (c == '(') ++depth;
            if (c == ')') --depth;
            if (c == sep && depth == 0) { out.emplace_back(); continue; }
        }
        out.back() += c;
    }
    return out;
}

std::string trim(const std::string &s)
{
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

struct Options {
    std::map<std::string, std::string> kv;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    std::string get(const std::string &k, const std::string &d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
};

// positional[i] names the i-th bare argument; alias maps long names onto the short ones.
Options parse_options(const std::string &args, const std::vector<std::string> &positional,
                      const std::map<std::string, std::string> &alias)
{
    Options o;
    if (trim(args).empty()) return o;
    size_t next = 0;
    for (const std::string &raw : split(args, ':')) {
        const std::string a = trim(raw);
        size_t eq = a.find('=');
        bool keyed = false;
        if (eq != std::string::npos) {
            keyed = true;
            for (size_t i = 0; i < eq; ++i)
                if (!(std::isalnum((unsigned char)a[i]) || a[i] == '_')) keyed = false;
        }
        std::string k, v;
        if (keyed) {
            k = a.substr(0, eq); v = a.substr(eq + 1);
            auto it = alias.find(k);
            if (it != alias.end()) k = it->second;
        } else {
            if (next >= positional.size()) throw std::runtime_error("too many arguments");
            k = positional[next++]; v = a;
        }
        o.kv[k] = v;
    }
    return o;
}

bool parse_size(const std::string &s, int &w, int &h)
{
    size_t x = s.find('x');
    if (x == std::string::npos || x == 0 || x + 1 >= s.size()) return false;
    for (size_t i = 0; i < s.size(); ++i)
        if (i != x && !std::isdigit((unsigned char)s[i])) return false;
    w = std::stoi(s.substr(0, x)); h = std::stoi(s.substr(x + 1));
    return true;
}

// RGB -> limited-range BT.601, the conversion pad's draw helper applies to its colour.
void rgb_to_yuv(int r, int g, int b, tts_geom &o)
{
    auto fix = [](double v) { return int(v * 1024 + 0.5); };
    o.fill_y = (fix(0.29900 * 219.0 / 255.0) * r + fix(0.58700 * 219.0 / 255.0) * g +
                fix(0.11400 * 219.0 / 255.0) * b + (512 + (16 << 10))) >> 10;
    o.fill_u = ((-fix(0.16874 * 224.0 / 255.0) * r - fix(0.33126 * 224.0 / 255.0) * g +
                 fix(0.50000 * 224.0 / 255.0) * b + 511) >> 10) + 128;
    o.fill_v = ((fix(0.50000 * 224.0 / 255.0) * r - fix(0.41869 * 224.0 / 255.0) * g -
                 fix(0.08131 * 224.0 / 255.0) * b + 511) >> 10) + 128;
}

// RGB -> full-range BT.601 (a pad on a yuvj420p link): libavutil/colorspace.h RGB_TO_Y_JPEG / RGB_TO_U_JPEG / RGB_TO_V_JPEG
void rgb_to_yuv_full(int r, int g, int b, tts_geom &o)
{
    auto fix = [](double v) { return int(v * 1024 + 0.5); };
    o.fill_y = std::min((fix(0.29900) * r + fix(0.58700) * g + fix(0.11400) * b + 512) >> 10, 255);
    o.fill_u = ((-fix(0.16874) * r - fix(0.33126) * g + fix(0.50000) * b + 511) >> 10) + 128;
    o.fill_v = ((fix(0.50000) * r - fix(0.41869) * g - fix(0.08131) * b + 511) >> 10) + 128;
}

int parse_pix_fmt(const std::string &f)
{
    if (f == "yuv420p" || f == "yuvj420p") return TTS_FMT_YUV420P;
    if (f == "rgb24") return TTS_FMT_RGB24;
    if (f == "bgr24") return TTS_FMT_BGR24;
    if (f == "rgba") return TTS_FMT_RGBA;
    if (f == "bgra") return TTS_FMT_BGRA;
    return -1;
}

std::string format_arg(const std::string &args)
{
    std::string f = trim(args);
    if (f.rfind("pix_fmts=", 0) == 0) f = f.substr(9);
    return f;
}

bool parse_color(std::string c, tts_geom &o)
{
    size_t at = c.find('@');
    if (at != std::string::npos) c = c.substr(0, at);
    for (char &ch : c) ch = char(std::tolower((unsigned char)ch));
    static const std::map<std::string, unsigned> named = {
        {"black", 0x000000}, {"white", 0xffffff}, {"red", 0xff0000}, {"green", 0x008000},
        {"blue", 0x0000ff}, {"gray", 0x808080}, {"grey", 0x808080}, {"yellow", 0xffff00}};
    unsigned rgb;
    auto it = named.find(c);
    if (it != named.end()) rgb = it->second;
    else {
        if (c.rfind("0x", 0) == 0) c = c.substr(2);
        else if (!c.empty() && c[0] == '#') c = c.substr(1);
        if (c.size() != 6) return false;
        for (char ch : c) if (!std::isxdigit((unsigned char)ch)) return false;
        rgb = unsigned(std::stoul(c, nullptr, 16));
    }
    rgb_to_yuv((rgb >> 16) & 255, (rgb >> 8) & 255, rgb & 255, o);
    o.fill_rgb = int32_t(rgb);
    return true;
}

int parse_alg(const std::string &flags, int &alg)
{
    bool seen = false;
    alg = TTS_ALG_BICUBIC;
    for (const std::string &raw : split(flags, '+')) {
        const std::string f = trim(raw);
        if (f.empty()) continue;
        if (f == "bicubic")  { alg = TTS_ALG_BICUBIC; seen = true; }
        else if (f == "area") { alg = TTS_ALG_AREA; seen = true; }
        else if (f == "bilinear") { alg = TTS_ALG_BILINEAR; seen = true; }
        else if (f == "accurate_rnd" || f == "bitexact" || f == "full_chroma_int" || f == "full_chroma_inp" || f == "print_info") {}
        else return TTS_ERR_UNSUPPORTED;
    }
    (void)seen;
    return TTS_OK;
}

}  // namespace

int parse_vf(const char *vf, int iw, int ih, tts_geom &g, std::string &err, int in_fmt)
{
    if (!vf || iw < 1 || ih < 1) { err = "bad arguments"; return TTS_ERR_INVALID_INPUT; }
    if (in_fmt < TTS_FMT_YUV420P || in_fmt > TTS_FMT_YUVJ420P) { err = "input format: yuv420p, yuvj420p, rgb24, bgr24, rgba or bgra"; return TTS_ERR_UNSUPPORTED; }
    const bool src_full = in_fmt == TTS_FMT_YUVJ420P;     // the same planes, full-range samples: only the range steps (and a pad's colour) differ
    if (src_full) in_fmt = TTS_FMT_YUV420P;
    std::memset(&g, 0, sizeof(g));
    g.in_fmt = in_fmt;
    g.in_w = iw; g.in_h = ih;
    g.pre_w = iw; g.pre_h = ih;
    g.alg = TTS_ALG_BICUBIC;
    g.fill_y = 16; g.fill_u = 128; g.fill_v = 128;

    enum { kBeforeScale, kAfterScale, kAfterPad } stage = kBeforeScale;
    bool converted_by_format = false;       // packed RGB in: a format=yuv420p stood where a scale would have converted
    int cw = iw, ch = ih;                   // size of the picture at the current point of the chain
    int cx = 0, cy = 0;                     // accumulated crop origin at the current stage

    // A packed-RGB format ending the chain: libavfilter's format negotiation makes the chain's `scale` (the
    // one filter that can change the pixel format) write it, unless a `format=yuv...` between the two pins
    // that link to yuv, or the chain has no scale; then a conversion pass of its own is inserted in front of
    // the `format` filter, i.e. at the very end.  First pass over the chain: find that out, because every
    // filter between the conversion and the end then works on RGB pixels (no chroma grid to round to).
    // The same negotiation decides where RANGE steps sit.  Every link has a range -- the source's (yuv420p: limited, yuvj420p:
    // full) until a filter changes it: a scale writes the range of the first yuv `format` behind it (it is the converter:
    // the step runs inside its context, on its 15-bit rows between the two passes: sc_range), any other change of range is
    // a pass of its own over the frame as it stands (out_range; here: at the end of the chain).  The reference's thumbnail
    // chain pins `format=yuv420p` behind its scale and the mjpeg encoder then asks for yuvj420p: a limited source gets one
    // pass at the end, a full-range one (some phone cameras) both steps.  A pad takes its colour in the range of its link.
    int sc_step = TTS_RANGE_KEEP, end_step = TTS_RANGE_KEEP;
    bool pad_full = src_full;
    try {
        bool seen_scale = false, yuv_pinned = false;                      // (the packed-RGB-out bookkeeping)
        bool cur_full = src_full, range_pass_seen = false;
        bool rgb_converted = false;                                       // packed RGB in: the converting context has been passed
        const std::vector<std::string> filters = split(vf, ',');
        auto yuv_format_of = [&](const std::string &raw, bool &is_format, bool &full) {   // true: a yuv format filter
            const std::string f = trim(raw);
            const size_t eq = f.find('=');
            is_format = trim(f.substr(0, eq)) == "format";
            if (!is_format) return false;
            const std::string pf = format_arg(eq == std::string::npos ? "" : f.substr(eq + 1));
            full = pf == "yuvj420p";
            return pf == "yuv420p" || pf == "yuvj420p";
        };
        for (size_t fi = 0; fi < filters.size(); ++fi) {
            const std::string f = trim(filters[fi]);
            if (f.empty()) continue;
            const size_t eq = f.find('=');
            const std::string name = trim(f.substr(0, eq));
            if (g.out_fmt != TTS_FMT_YUV420P && name != "setsar") { err = "an RGB format must end the chain"; return TTS_ERR_UNSUPPORTED; }
            if (name == "scale") {
                if (g.in_fmt == TTS_FMT_YUV420P) {
                    if (range_pass_seen) { err = "a range step in front of the scale (it would run first, at the source's size): not supported"; return TTS_ERR_UNSUPPORTED; }
                    bool mid = cur_full;                       // the scale's output link: the first yuv format behind it says
                    for (size_t fj = fi + 1; fj < filters.size(); ++fj) {
                        bool is_format, full;
                        if (yuv_format_of(filters[fj], is_format, full)) { mid = full; break; }
                        if (is_format) break;                  // (a packed-RGB format: the scale converts to that; no range of its own)
                    }
                    if (mid != cur_full) sc_step = mid ? TTS_RANGE_TO_FULL : TTS_RANGE_TO_LIMITED;
                    cur_full = mid;
                } else if (!rgb_converted) {
                    // packed RGB in: the scale is the converter; it writes the range of the first yuv format behind it (yuv420p
                    // where nothing says), through the limited-range table + lum/chrRangeToJpeg_c for yuvj420p (sc_range)
                    bool mid = false;
                    for (size_t fj = fi + 1; fj < filters.size(); ++fj) {
                        bool is_format, full;
                        if (yuv_format_of(filters[fj], is_format, full)) { mid = full; break; }
                        if (is_format) break;
                    }
                    if (mid) sc_step = TTS_RANGE_TO_FULL;
                    cur_full = mid; rgb_converted = true;
                }
                seen_scale = true; yuv_pinned = false;
            }
            if (name == "pad") pad_full = cur_full;
            if (name != "format") continue;
            const std::string pf = format_arg(eq == std::string::npos ? "" : f.substr(eq + 1));
            const int fmt = parse_pix_fmt(pf);
            if (fmt < 0) { err = "format: only yuv420p, yuvj420p, rgb24, bgr24, rgba, bgra are supported"; return TTS_ERR_UNSUPPORTED; }
            if (g.in_fmt == TTS_FMT_YUV420P && fmt == TTS_FMT_YUV420P && (pf == "yuvj420p") != cur_full) {
                if (end_step != TTS_RANGE_KEEP) { err = "two range passes in one chain: not supported"; return TTS_ERR_UNSUPPORTED; }
                end_step = pf == "yuvj420p" ? TTS_RANGE_TO_FULL : TTS_RANGE_TO_LIMITED;
                cur_full = pf == "yuvj420p";
                range_pass_seen = true;
            }
            if (g.in_fmt != TTS_FMT_YUV420P && fmt == TTS_FMT_YUV420P) {
                // fed with packed RGB: the first yuv format is written by the context that converts (the scale in front of it, or
                // the pass inserted here); one of the other range behind it is a range pass of its own (the reference's thumbnail
                // chain pins yuv420p and its mjpeg encoder asks for yuvj420p: `...,format=yuv420p,format=yuvj420p`)
                const bool full = pf == "yuvj420p";
                if (!rgb_converted) { if (full) sc_step = TTS_RANGE_TO_FULL; cur_full = full; rgb_converted = true; }
                else if (full != cur_full) {
                    if (end_step != TTS_RANGE_KEEP) { err = "two range passes in one chain: not supported"; return TTS_ERR_UNSUPPORTED; }
                    end_step = full ? TTS_RANGE_TO_FULL : TTS_RANGE_TO_LIMITED;
                    cur_full = full;
                }
            }
            if (fmt == TTS_FMT_YUV420P) { yuv_pinned = true; continue; }
            if (sc_step != TTS_RANGE_KEEP) { err = "packed RGB behind a scale that changes the range: pin the scale's output first (format=yuv420p,format=yuvj420p: a range pass of its own)"; return TTS_ERR_UNSUPPORTED; }
            if (g.in_fmt == TTS_FMT_YUV420P && cur_full && end_step != TTS_RANGE_TO_FULL) { err = "packed RGB out of a full-range source: not supported"; return TTS_ERR_UNSUPPORTED; }
            // (RGB from a full-range link, `format=yuvj420p,format=rgb24`: the range pass stays in out_range and the conversion
            // pass behind it reads a yuvj420p source -- full-range yuv2rgb tables; yuv_pinned is set, so it sits at the end)
            g.out_fmt = fmt;
            g.rgb_stage = seen_scale && !yuv_pinned ? TTS_RGB_AT_SCALE : TTS_RGB_AT_END;
        }
    } catch (const std::exception &e) {
        err = e.what();
        return TTS_ERR_INVALID_FORMAT;
    }
    // chroma subsampling of the link a filter sits on: 2 on yuv420p, 1 once the pixels are RGB
    // (rgb24 in: everything in front of the scale -- or, without one, the whole chain -- works on RGB pixels)
    auto grid = [&]() {
        if (g.in_fmt != TTS_FMT_YUV420P) return stage == kBeforeScale ? 1 : 2;
        return g.rgb_stage == TTS_RGB_AT_SCALE && stage != kBeforeScale ? 1 : 2;
    };
    if (g.in_fmt != TTS_FMT_YUV420P && g.out_fmt != TTS_FMT_YUV420P) { err = "packed RGB in and packed RGB out is not supported"; return TTS_ERR_UNSUPPORTED; }

    try {
        for (const std::string &raw : split(vf, ',')) {
            const std::string f = trim(raw);
            if (f.empty()) continue;
            const size_t eq = f.find('=');
            const std::string name = trim(f.substr(0, eq));
            const std::string args = eq == std::string::npos ? "" : f.substr(eq + 1);

            if (name == "format" || name == "setsar") {
                if (name == "format") {
                    const std::string pf = format_arg(args);
                    // fed with packed RGB: a `format=yuv420p` in front of every scale is where the conversion pass goes (a context
                    // of its own, same size) -- the rest of the chain sees yuv420p, exactly as behind a scale to the same size
                    if (g.in_fmt != TTS_FMT_YUV420P && stage == kBeforeScale && (pf == "yuv420p" || pf == "yuvj420p")) {
                        g.pre_x = cx; g.pre_y = cy; g.pre_w = cw; g.pre_h = ch;
                        g.do_scale = 0; g.sc_w = cw; g.sc_h = ch;
                        cx = cy = 0;
                        stage = kAfterScale;
                        converted_by_format = true;
                    }
                    // (which range steps the yuv formats stand for: the first pass above has decided, see below)
                }
                continue;
            }
            if (stage == kAfterPad) { err = "no filter may follow pad"; return TTS_ERR_UNSUPPORTED; }

            if (name == "scale") {
                if (converted_by_format) { err = "packed RGB in: a scale behind the format=yuv420p that converts (two swscale contexts): not supported"; return TTS_ERR_UNSUPPORTED; }
                if (stage != kBeforeScale) { err = "only one scale per chain"; return TTS_ERR_UNSUPPORTED; }
                Options o = parse_options(args, {"w", "h", "flags"},
                    {{"width", "w"}, {"height", "h"}, {"size", "s"}, {"sws_flags", "flags"}});
                for (auto &kv : o.kv) {
                    static const char *known[] = {"w", "h", "s", "flags", "force_original_aspect_ratio",
                                                  "force_divisible_by", "interl", "eval"};
                    bool ok = false;
                    for (const char *k : known) ok |= kv.first == k;
                    if (!ok) { err = "scale: unsupported option '" + kv.first + "'"; return TTS_ERR_INVALID_FORMAT; }
                }
                if (o.get("interl", "0") != "0") { err = "interlaced scaling unsupported"; return TTS_ERR_UNSUPPORTED; }
                int w = 0, h = 0, sw, sh;
                std::string ws = o.get("w", ""), hs = o.get("h", "");
                if (o.has("s")) { if (!parse_size(o.get("s", ""), w, h)) { err = "scale: bad size"; return TTS_ERR_INVALID_FORMAT; } }
                else if (!ws.empty() && hs.empty() && parse_size(ws, sw, sh)) { w = sw; h = sh; }
                else {
                    Vars v = {{"in_w", cw}, {"iw", cw}, {"in_h", ch}, {"ih", ch}, {"a", double(cw) / ch},
                              {"dar", double(cw) / ch}, {"sar", 1}, {"hsub", 2}, {"vsub", 2}, {"ohsub", 2}, {"ovsub", 2}};
                    if (ws.empty()) ws = "iw";
                    if (hs.empty()) hs = "ih";
                    // width first (may not know oh yet), then height, then width again
                    v["out_w"] = v["ow"] = cw; v["out_h"] = v["oh"] = ch;
                    double rw = NAN;
                    try { rw = eval(ws, v); } catch (...) {}
                    if (!std::isnan(rw)) v["out_w"] = v["ow"] = int(rw) == 0 ? cw : int(rw);
                    double rh = eval(hs, v);
                    v["out_h"] = v["oh"] = int(rh) == 0 ? ch : int(rh);
                    rw = eval(ws, v);
                    w = int(rw); h = int(rh);
                }
                int foar = 0;
                const std::string fo = o.get("force_original_aspect_ratio", "disable");
                if (fo == "disable" || fo == "0") foar = 0;
                else if (fo == "decrease" || fo == "1") foar = 1;
                else if (fo == "increase" || fo == "2") foar = 2;
                else { err = "scale: bad force_original_aspect_ratio"; return TTS_ERR_INVALID_FORMAT; }
                const int fdb = int(eval(o.get("force_divisible_by", "1"), {}));
                int rc = parse_alg(o.get("flags", "bicubic"), g.alg);
                if (rc) { err = "scale: unsupported flags '" + o.get("flags", "") + "'"; return rc; }
                int ow, oh;
                if (!scale_output_size(cw, ch, w, h, foar, fdb, ow, oh)) { err = "scale: invalid output size"; return TTS_ERR_INVALID_INPUT; }
                g.pre_x = cx; g.pre_y = cy; g.pre_w = cw; g.pre_h = ch;
                g.do_scale = (ow != cw || oh != ch);
                g.sc_w = ow; g.sc_h = oh;
                cw = ow; ch = oh; cx = cy = 0;
                stage = kAfterScale;
            } else if (name == "crop") {
                Options o = parse_options(args, {"w", "h", "x", "y", "keep_aspect", "exact"},
                    {{"out_w", "w"}, {"out_h", "h"}});
                if (o.get("exact", "0") != "0" || o.get("keep_aspect", "0") != "0") { err = "crop: exact/keep_aspect unsupported"; return TTS_ERR_UNSUPPORTED; }
                const int sub = grid();                 // vf_crop rounds sizes and offsets to the link's chroma grid
                Vars v = {{"in_w", cw}, {"iw", cw}, {"in_h", ch}, {"ih", ch}, {"a", double(cw) / ch},
                          {"sar", 1}, {"dar", double(cw) / ch}, {"hsub", sub}, {"vsub", sub}, {"n", 0}, {"t", 0},
                          {"x", 0}, {"y", 0}};
                v["out_w"] = v["ow"] = cw; v["out_h"] = v["oh"] = ch;
                int w = int(eval(o.get("w", "iw"), v));
                v["out_w"] = v["ow"] = w;
                int h = int(eval(o.get("h", "ih"), v));
                v["out_h"] = v["oh"] = h;
                w = int(eval(o.get("w", "iw"), v));
                w &= ~(sub - 1); h &= ~(sub - 1);
                if (w <= 0 || h <= 0 || w > cw || h > ch) { err = "crop: invalid too big or non positive size"; return TTS_ERR_INVALID_INPUT; }
                v["out_w"] = v["ow"] = w; v["out_h"] = v["oh"] = h;
                double xv = 0, yv = 0;
                const bool hx = o.has("x"), hy = o.has("y");
                if (hx) xv = eval(o.get("x", "0"), v);
                v["x"] = hx ? xv : (cw - w) / 2.0;
                if (hy) yv = eval(o.get("y", "0"), v);
                v["y"] = hy ? yv : (ch - h) / 2.0;
                if (hx) xv = eval(o.get("x", "0"), v);   // x may be written in terms of y
                const int x = crop_origin(cw, w, xv, hx, sub), y = crop_origin(ch, h, yv, hy, sub);
                cx += x; cy += y; cw = w; ch = h;
                if (stage == kBeforeScale) { g.pre_x = cx; g.pre_y = cy; g.pre_w = cw; g.pre_h = ch; }
            } else if (name == "pad") {
                Options o = parse_options(args, {"w", "h", "x", "y", "color"},
                    {{"width", "w"}, {"height", "h"}});
                int w = 0, h = 0, sw, sh;
                std::string ws = o.get("w", "iw"), hs = o.get("h", "ih"), xs = o.get("x", "0"), ys = o.get("y", "0");
                // The reference writes pad=320x180:-1:-1 (src/services/video.js:331): a WxH size in the
                // first slot, then x, y.  Read it the way it is meant.
                if (parse_size(ws, sw, sh)) { w = sw; h = sh; xs = o.get("h", "0"); ys = o.get("x", "0"); if (o.has("y") && !o.has("color")) o.kv["color"] = o.get("y", "black"); }
                else {
                    Vars v = {{"in_w", cw}, {"iw", cw}, {"in_h", ch}, {"ih", ch}, {"a", double(cw) / ch},
                              {"sar", 1}, {"dar", double(cw) / ch}, {"hsub", 2}, {"vsub", 2}};
                    v["out_w"] = v["ow"] = cw; v["out_h"] = v["oh"] = ch;
                    w = int(eval(ws, v)); if (!w) w = cw;
                    v["out_w"] = v["ow"] = w;
                    h = int(eval(hs, v)); if (!h) h = ch;
                    v["out_h"] = v["oh"] = h;
                    w = int(eval(ws, v)); if (!w) w = cw;
                }
                Vars v = {{"in_w", cw}, {"iw", cw}, {"in_h", ch}, {"ih", ch}, {"out_w", w}, {"ow", w},
                          {"out_h", h}, {"oh", h}, {"a", double(cw) / ch}, {"sar", 1}, {"dar", double(cw) / ch},
                          {"hsub", 2}, {"vsub", 2}, {"x", 0}, {"y", 0}};
                int x = int(eval(xs, v)); v["x"] = x;
                int y = int(eval(ys, v)); v["y"] = y;
                x = int(eval(xs, v));
                if (w < cw || h < ch) { err = "pad: padded size smaller than input"; return TTS_ERR_INVALID_INPUT; }
                if (x < 0 || x + cw > w) x = (w - cw) / 2;
                if (y < 0 || y + ch > h) y = (h - ch) / 2;
                if (grid() == 2) { w &= ~1; h &= ~1; x &= ~1; y &= ~1; }   // round down to the chroma grid (none on RGB)
                if (!parse_color(o.get("color", "black"), g)) { err = "pad: unsupported colour"; return TTS_ERR_INVALID_FORMAT; }
                // packed RGB in and no scale in front: the pad runs on RGB pixels and the conversion pass libavfilter inserts
                // behind it reads the padded canvas, border included (include/ttscale.h, TTS_RGB_AT_END)
                if (stage == kBeforeScale && g.in_fmt != TTS_FMT_YUV420P) g.rgb_stage = TTS_RGB_AT_END;
                if (stage == kBeforeScale) { g.pre_x = cx; g.pre_y = cy; g.pre_w = cw; g.pre_h = ch; g.sc_w = cw; g.sc_h = ch; cx = cy = 0; }
                g.crop_x = cx; g.crop_y = cy; g.crop_w = cw; g.crop_h = ch;
                g.do_pad = 1; g.pad_x = x; g.pad_y = y; g.out_w = w; g.out_h = h;
                stage = kAfterPad;
            } else {
                err = "unsupported filter '" + name + "'";
                return TTS_ERR_INVALID_FORMAT;
            }
        }
    } catch (const std::exception &e) {
        err = e.what();
        return TTS_ERR_INVALID_FORMAT;
    }

    if (stage == kBeforeScale) { g.pre_x = cx; g.pre_y = cy; g.pre_w = cw; g.pre_h = ch; g.sc_w = cw; g.sc_h = ch; cx = cy = 0; }
    if (stage != kAfterPad) {
        g.crop_x = cx; g.crop_y = cy; g.crop_w = cw; g.crop_h = ch;
        g.out_w = cw; g.out_h = ch;
    }
    // fed with packed RGB into yuvj420p: a pad behind the converting context sits on a yuvj420p link (a pad on the RGB pixels in
    // front of it keeps its RGB colour)
    if (g.in_fmt != TTS_FMT_YUV420P) {
        g.sc_range = sc_step; g.out_range = end_step;                     // (the converting context's own step; a pass at the end)
        if (g.do_pad && g.rgb_stage != TTS_RGB_AT_END && pad_full) rgb_to_yuv_full((g.fill_rgb >> 16) & 255, (g.fill_rgb >> 8) & 255, g.fill_rgb & 255, g);
    }
    if (g.in_fmt == TTS_FMT_YUV420P) {
        g.out_range = end_step;
        if (sc_step != TTS_RANGE_KEEP) {
            if (g.do_scale) g.sc_range = sc_step;
            else if (end_step == TTS_RANGE_KEEP && !g.do_pad) g.out_range = sc_step;   // (a scale to the same size: one-tap filters around the step -- the bytes of a pass of its own)
            else { err = "a scale to the same size that changes the range, then a pad or another range step: not supported"; return TTS_ERR_UNSUPPORTED; }
        }
        // the pad's colour in the range of its link: drawutils.c ff_draw_color() with full_range, libavutil/colorspace.h RGB_TO_Y_JPEG & co.
        if (g.do_pad && pad_full && g.rgb_stage != TTS_RGB_AT_SCALE) rgb_to_yuv_full((g.fill_rgb >> 16) & 255, (g.fill_rgb >> 8) & 255, g.fill_rgb & 255, g);
    }
    const int rc = check_geom(g);
    if (rc) err = "resolved geometry is invalid";
    return rc;
}

}  // namespace tts
