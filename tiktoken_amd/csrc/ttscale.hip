// libttscale: yuv420p 8-bit scale / crop / pad for MI355X (gfx950), HIP kernels + C ABI.
//
// Replaces the per-frame `scale`/`crop`/`pad` loop that the reference hands to an
// external ffmpeg (scripts/processVideos.js:71-108, src/services/video.js:329-334).
// Arithmetic contract: horizontal pass to a 15-bit intermediate
// with a truncating >>7 and a clip to 32767, vertical pass with +64<<12 rounding,
// >>19 and a clip to 8 bits; integer only, bit-exact against the CPU oracle.
//
// Kernel families
//   strip_kernel<LW,R,N>  the hot path.  One lane owns N adjacent output columns of one
//       plane of one frame and walks down a segment of output rows.  Per source row it
//       gathers one LW-byte window per column straight through the vector L1 (the
//       windows of a wave are a contiguous run of the row), runs the horizontal taps
//       with v_dot2 on the window, and drops the 15-bit result into a register ring of
//       R source rows kept as packed row pairs.  Every output row whose last source
//       row has arrived is produced from the ring with v_dot2 against wave-uniform
//       vertical taps (scalar loads) and stored as N contiguous bytes.  The
//       intermediate never leaves registers: HBM sees the source window once and the
//       output once.  No LDS, no barriers, no inter-workgroup traffic.
//   generic_kernel        any tap counts (e.g. the 29-tap 720->101 thumbnail), one
//       lane per output sample; correctness path for shapes the strip kernel does not
//       cover.
//   fill_kernel           pad colour outside the picture.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "filter.h"

namespace {

// ------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------

constexpr int kBlock = 256;
constexpr int kMaxTableFrames = 32;   // frames addressed through the in-kernarg offset table

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

struct PlaneAddr {
    // item -> byte offset of its plane origin, relative to the base pointer of the launch
    uint32_t src_pitch, dst_pitch;            // packed mode: bytes between frames
    uint32_t src_plane[2], dst_plane[2];      // packed mode: plane origin inside a frame
    int32_t  use_table;                        // table mode: explicit per-frame offsets
    uint32_t src_tab[2][kMaxTableFrames];
    uint32_t dst_tab[2][kMaxTableFrames];
};

struct StripArgs {
    uint32_t src_span, dst_span;      // bytes reachable from the base pointers (buffer range check)
    int32_t src_stride, dst_stride;
    int32_t ow, oh;
    int32_t strips, planes, n_items;
    int32_t seg_rows, nseg;
    PlaneAddr addr;
};

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), c, false);
}

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of
// logical blocks so neighbouring strips and row segments share one L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

template <int LW>
__device__ __forceinline__ int hpass(const uint32_t (&w)[LW / 4], const uint32_t (&c)[LW / 2])
{
    int acc = 0;
#pragma unroll
    for (int k = 0; k < LW / 4; ++k) {
        acc = dot2(__builtin_amdgcn_perm(0, w[k], 0x0c010c00), c[2 * k], acc);
        acc = dot2(__builtin_amdgcn_perm(0, w[k], 0x0c030c02), c[2 * k + 1], acc);
    }
    acc >>= 7;                         // truncating, keeps the sign
    return acc < 32767 ? acc : 32767;
}

// four vertical sums -> four clipped bytes (val >> 19, clip to 0..255)
__device__ __forceinline__ uint32_t pack4(int v0, int v1, int v2, int v3)
{
    // high halves of each pair, then >>3 on the packed int16s == >>19 on the int32s
    const uint32_t h01 = __builtin_amdgcn_perm(uint32_t(v1), uint32_t(v0), 0x07060302);
    const uint32_t h23 = __builtin_amdgcn_perm(uint32_t(v3), uint32_t(v2), 0x07060302);
    const short2_t s01 = __builtin_bit_cast(short2_t, h01) >> short2_t{3, 3};
    const short2_t s23 = __builtin_bit_cast(short2_t, h23) >> short2_t{3, 3};
    uint32_t b01, b23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b01) : "v"(s01));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(b23) : "v"(s23));
    return b01 | (b23 << 16);
}

// hoff  [strips*N]        first byte of each column's window
// hcoef [strips*N][LW/2]  window taps as int16 pairs
// vtab  [oh][2+R/2]       first row, last row, taps by ring slot as int16 pairs
// The tables are __restrict__ kernel arguments so the wave-uniform reads become scalar loads.
template <int LW, int R, int N>
__global__ __launch_bounds__(kBlock) void strip_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                       const uint32_t *__restrict__ hoff,
                                                       const uint32_t *__restrict__ hcoef,
                                                       const uint32_t *__restrict__ vtab, const StripArgs a)
{
    static_assert(LW == 4 || LW == 8, "window is one or two dwords");
    static_assert(R % 2 == 0 && N % 4 == 0, "ring holds row pairs; stores are dwords");
    constexpr int VS = 2 + R / 2;

    const uint32_t lb  = xcd_remap(blockIdx.x, gridDim.x);
    const int      seg = int(lb % uint32_t(a.nseg));
    const int      sb  = int(lb / uint32_t(a.nseg));
    int            g   = sb * kBlock + int(threadIdx.x);
    const bool     live = g < a.n_items;
    if (!live) g = a.n_items - 1;

    const int strip = g % a.strips;
    const int fp    = g / a.strips;
    const int plane = fp % a.planes;
    const int frame = fp / a.planes;
    uint32_t soff, doff;
    if (a.addr.use_table) {
        soff = a.addr.src_tab[plane][frame];
        doff = a.addr.dst_tab[plane][frame];
    } else {
        soff = uint32_t(frame) * a.addr.src_pitch + a.addr.src_plane[plane];
        doff = uint32_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane];
    }
    doff += uint32_t(strip) * N;

    // per-lane horizontal tables stay in registers for the whole walk
    uint32_t off[N];
    uint32_t hc[N][LW / 2];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        off[i] = hoff[strip * N + i] + soff;
#pragma unroll
        for (int k = 0; k < LW / 2; ++k) hc[i][k] = hcoef[(strip * N + i) * (LW / 2) + k];
    }

    uint32_t ring[N][R / 2];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < R / 2; ++k) ring[i][k] = 0;

    const int y0 = seg * a.seg_rows;
    const int y1 = min(a.oh, y0 + a.seg_rows);
    int y = y0;
    const int rstart = __builtin_amdgcn_readfirstlane(int(vtab[y0 * VS]));
    const int rend   = __builtin_amdgcn_readfirstlane(int(vtab[(y1 - 1) * VS + 1]));
    int ylast = __builtin_amdgcn_readfirstlane(int(vtab[y * VS + 1]));
    const int cols = live ? min(N, a.ow - strip * N) : 0;     // valid columns of this strip

    // Buffer addressing: wave-uniform descriptor + per-lane 32-bit byte offset + scalar row
    // offset, so a row step costs no vector instructions; reads past the span return 0.
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, a.src_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, a.dst_span, 0x00020000);

    for (int r0 = rstart - rstart % R; r0 <= rend; r0 += R) {
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int r = r0 + s;
            if (r < rstart || r > rend) continue;             // wave-uniform
            const int rowoff = r * a.src_stride;
            uint32_t w[N][LW / 4];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if constexpr (LW == 4) {
                    w[i][0] = __builtin_amdgcn_raw_buffer_load_b32(srs, off[i], rowoff, 0);
                } else {
                    const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(srs, off[i], rowoff, 0);
                    w[i][0] = t.x; w[i][1] = t.y;
                }
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t t = uint32_t(hpass<LW>(w[i], hc[i]));
                ring[i][s / 2] = (s & 1) ? __builtin_amdgcn_perm(t, ring[i][s / 2], 0x05040100)
                                         : __builtin_amdgcn_perm(ring[i][s / 2], t, 0x07060100);
            }
            while (y < y1 && ylast == r) {                     // wave-uniform
                const uint32_t *vt = vtab + y * VS + 2;
                uint32_t vk[R / 2];
#pragma unroll
                for (int k = 0; k < R / 2; ++k) vk[k] = uint32_t(__builtin_amdgcn_readfirstlane(int(vt[k])));
                uint32_t o[N / 4];
#pragma unroll
                for (int q = 0; q < N / 4; ++q) {
                    int v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int acc = 64 << 12;
#pragma unroll
                        for (int k = 0; k < R / 2; ++k) acc = dot2(ring[4 * q + e][k], vk[k], acc);
                        v[e] = acc;
                    }
                    o[q] = pack4(v[0], v[1], v[2], v[3]);
                }
                const int outoff = y * a.dst_stride;
                if (cols == N) {
                    if constexpr (N == 16) {
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{o[0], o[1], o[2], o[3]}, drs, doff, outoff, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{o[0], o[1]}, drs, doff, outoff, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i < cols) __builtin_amdgcn_raw_buffer_store_b8(uint8_t(o[i / 4] >> (8 * (i % 4))), drs, doff + i, outoff, 0);
                }
                ++y;
                ylast = __builtin_amdgcn_readfirstlane(int(vtab[min(y, a.oh - 1) * VS + 1]));
            }
        }
    }
}

struct GenericArgs {
    const uint8_t *src;
    uint8_t       *dst;
    const int32_t *hpos;  const int16_t *htaps;   // indexed by virtual column
    const int32_t *vpos;  const int16_t *vtaps;   // indexed by virtual row
    int32_t ht, vt;
    int32_t x0, y0, ow, oh;
    int32_t src_stride, dst_stride;
    int32_t planes, n_fp;                          // frames*planes
    PlaneAddr addr;
};

__global__ __launch_bounds__(kBlock) void generic_kernel(const GenericArgs a)
{
    const int x  = int(blockIdx.x) * 64 + int(threadIdx.x & 63);
    const int y  = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    const int fp = int(blockIdx.z);
    if (x >= a.ow || y >= a.oh) return;
    const int plane = fp % a.planes, frame = fp / a.planes;
    uint32_t soff, doff;
    if (a.addr.use_table) {
        soff = a.addr.src_tab[plane][frame];
        doff = a.addr.dst_tab[plane][frame];
    } else {
        soff = uint32_t(frame) * a.addr.src_pitch + a.addr.src_plane[plane];
        doff = uint32_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane];
    }
    const int vx = a.x0 + x, vy = a.y0 + y;
    const int hp = a.hpos[vx], vp = a.vpos[vy];
    const int16_t *hc = a.htaps + size_t(vx) * a.ht;
    const int16_t *vc = a.vtaps + size_t(vy) * a.vt;
    int val = 64 << 12;
    for (int j = 0; j < a.vt; ++j) {
        const uint8_t *row = a.src + soff + size_t(vp + j) * size_t(a.src_stride) + hp;
        int acc = 0;
        for (int k = 0; k < a.ht; ++k) acc += int(row[k]) * int(hc[k]);
        acc >>= 7;
        acc = acc < 32767 ? acc : 32767;
        val += acc * int(vc[j]);
    }
    val >>= 19;
    val = val < 0 ? 0 : val > 255 ? 255 : val;
    a.dst[doff + size_t(y) * size_t(a.dst_stride) + x] = uint8_t(val);
}

struct FillArgs {
    uint8_t *dst;
    int32_t w, h, stride;              // canvas of this plane
    int32_t px, py, pw, ph;            // picture rectangle that is left alone
    int32_t planes, n_fp;
    int32_t fill[2];
    PlaneAddr addr;                    // dst side only
};

__global__ __launch_bounds__(kBlock) void fill_kernel(const FillArgs a)
{
    const int x  = (int(blockIdx.x) * 64 + int(threadIdx.x & 63)) * 4;
    const int y  = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    const int fp = int(blockIdx.z);
    if (x >= a.w || y >= a.h) return;
    const int plane = fp % a.planes, frame = fp / a.planes;
    const uint32_t doff = a.addr.use_table ? a.addr.dst_tab[plane][frame]
                                           : uint32_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane];
    uint8_t *row = a.dst + doff + size_t(y) * size_t(a.stride);
    const uint8_t f = uint8_t(a.fill[plane]);
    const bool in_rows = y >= a.py && y < a.py + a.ph;
    for (int i = x; i < min(x + 4, a.w); ++i)
        if (!(in_rows && i >= a.px && i < a.px + a.pw)) row[i] = f;
}

}  // namespace

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------

namespace {

struct ClassPlan {
    tts::AxisFilter h, v;
    int sub = 0;
    int src_w = 0, src_h = 0;          // (pre-cropped) source plane
    int x0 = 0, y0 = 0, ow = 0, oh = 0; // region of the virtual scaled plane that is produced
    int canvas_w = 0, canvas_h = 0;    // output plane
    int pad_x = 0, pad_y = 0;
    int kind = 0;                      // 0 generic, 1 strip
    int LW = 0, R = 0, N = 0, strips = 0;
    // device tables
    uint32_t *d_hoff = nullptr, *d_hcoef = nullptr, *d_vtab = nullptr;
    int32_t *d_hpos = nullptr, *d_vpos = nullptr;
    int16_t *d_htaps = nullptr, *d_vtaps = nullptr;
};

thread_local std::string g_last_error;

template <class T>
bool upload(T *&dptr, const std::vector<T> &host)
{
    if (hipMalloc(reinterpret_cast<void **>(&dptr), std::max<size_t>(host.size(), 1) * sizeof(T)) != hipSuccess) return false;
    return hipMemcpy(dptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

inline int ceil_shift(int v, int s) { return (v + (1 << s) - 1) >> s; }

int env_int(const char *name, int dflt)
{
    const char *e = std::getenv(name);
    return e && *e ? std::atoi(e) : dflt;
}

}  // namespace

struct tts_plan {
    tts_geom g;
    ClassPlan cls[2];
    int device = 0;
};

namespace {

bool monotone(const std::vector<int32_t> &p)
{
    for (size_t i = 1; i < p.size(); ++i) if (p[i] < p[i - 1]) return false;
    return true;
}

int build_class(const tts_geom &g, int sub, ClassPlan &c)
{
    c.sub = sub;
    c.src_w = ceil_shift(g.pre_w, sub); c.src_h = ceil_shift(g.pre_h, sub);
    const int vw = ceil_shift(g.sc_w, sub), vh = ceil_shift(g.sc_h, sub);
    if (!tts::build_axis_filter(c.src_w, vw, 1 << 14, g.alg, c.h)) return TTS_ERR_INVALID_INPUT;
    if (!tts::build_axis_filter(c.src_h, vh, 1 << 12, g.alg, c.v)) return TTS_ERR_INVALID_INPUT;
    const int content_w = g.do_pad ? (g.crop_w & ~1) : g.crop_w;   // pad's right border eats an odd last column
    c.x0 = g.crop_x >> sub; c.y0 = g.crop_y >> sub;
    c.ow = ceil_shift(content_w, sub); c.oh = ceil_shift(g.crop_h, sub);
    c.canvas_w = ceil_shift(g.out_w, sub); c.canvas_h = ceil_shift(g.out_h, sub);
    c.pad_x = g.do_pad ? g.pad_x >> sub : 0; c.pad_y = g.do_pad ? g.pad_y >> sub : 0;
    if (c.ow < 1 || c.oh < 1) return TTS_ERR_INVALID_INPUT;

    // generic tables (also what tts_plan_filter reports)
    if (!upload(c.d_hpos, c.h.pos) || !upload(c.d_htaps, c.h.taps) ||
        !upload(c.d_vpos, c.v.pos) || !upload(c.d_vtaps, c.v.taps)) return TTS_ERR_INITIALIZATION_FAILED;

    const int HT = c.h.size, VT = c.v.size;
    const int LW = HT <= 4 ? 4 : 8;
    const int R = VT <= 4 ? 4 : VT <= 6 ? 6 : 8;
    const bool strip_ok = HT <= 8 && VT <= 8 && c.src_w >= LW && monotone(c.v.pos) &&
                          !env_int("TTS_FORCE_GENERIC", 0);
    if (!strip_ok) { c.kind = 0; return TTS_OK; }
    c.kind = 1; c.LW = LW; c.R = R;
    c.N = (LW == 4 && R == 4) ? 16 : 8;
    c.strips = (c.ow + c.N - 1) / c.N;

    std::vector<uint32_t> hoff(size_t(c.strips) * c.N), hcoef(size_t(c.strips) * c.N * (LW / 2), 0);
    for (int i = 0; i < c.strips * c.N; ++i) {
        const int vx = c.x0 + std::min(i, c.ow - 1);
        const int pos = c.h.pos[vx];
        const int lo = std::min(pos, c.src_w - LW);
        int16_t slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < HT; ++j) slot[pos - lo + j] = c.h.taps[size_t(vx) * HT + j];
        hoff[i] = uint32_t(lo);
        for (int k = 0; k < LW / 2; ++k)
            hcoef[size_t(i) * (LW / 2) + k] = uint32_t(uint16_t(slot[2 * k])) | (uint32_t(uint16_t(slot[2 * k + 1])) << 16);
    }
    const int VS = 2 + R / 2;
    std::vector<uint32_t> vtab(size_t(c.oh) * VS, 0);
    for (int y = 0; y < c.oh; ++y) {
        const int vy = c.y0 + y;
        const int first = c.v.pos[vy];
        int16_t slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < VT; ++j) slot[(first + j) % R] = c.v.taps[size_t(vy) * VT + j];
        vtab[size_t(y) * VS + 0] = uint32_t(first);
        vtab[size_t(y) * VS + 1] = uint32_t(first + VT - 1);
        for (int k = 0; k < R / 2; ++k)
            vtab[size_t(y) * VS + 2 + k] = uint32_t(uint16_t(slot[2 * k])) | (uint32_t(uint16_t(slot[2 * k + 1])) << 16);
    }
    if (!upload(c.d_hoff, hoff) || !upload(c.d_hcoef, hcoef) || !upload(c.d_vtab, vtab)) return TTS_ERR_INITIALIZATION_FAILED;
    return TTS_OK;
}

void free_class(ClassPlan &c)
{
    (void)hipFree(c.d_hoff); (void)hipFree(c.d_hcoef); (void)hipFree(c.d_vtab);
    (void)hipFree(c.d_hpos); (void)hipFree(c.d_vpos); (void)hipFree(c.d_htaps); (void)hipFree(c.d_vtaps);
}

typedef void (*strip_fn)(const uint8_t *, uint8_t *, const uint32_t *, const uint32_t *, const uint32_t *, const StripArgs);

strip_fn pick_strip(int LW, int R, int N)
{
    if (LW == 4 && R == 4 && N == 16) return strip_kernel<4, 4, 16>;
    if (LW == 4 && R == 6) return strip_kernel<4, 6, 8>;
    if (LW == 4 && R == 8) return strip_kernel<4, 8, 8>;
    if (LW == 8 && R == 4) return strip_kernel<8, 4, 8>;
    if (LW == 8 && R == 6) return strip_kernel<8, 6, 8>;
    if (LW == 8 && R == 8) return strip_kernel<8, 8, 8>;
    return nullptr;
}

// One launch set: frames [0, n) addressed by `addr` relative to (src, dst).
int launch_class(const tts_plan *p, int ci, const uint8_t *src, uint8_t *dst, int n,
                 int src_stride, int dst_stride, uint32_t src_span, uint32_t dst_span,
                 const PlaneAddr &addr, hipStream_t st)
{
    const ClassPlan &c = p->cls[ci];
    const int planes = ci == 0 ? 1 : 2;

    if (p->g.do_pad) {
        FillArgs f{};
        f.dst = dst; f.w = c.canvas_w; f.h = c.canvas_h; f.stride = dst_stride;
        f.px = c.pad_x; f.py = c.pad_y; f.pw = c.ow; f.ph = c.oh;
        f.planes = planes; f.n_fp = n * planes;
        f.fill[0] = ci == 0 ? p->g.fill_y : p->g.fill_u;
        f.fill[1] = p->g.fill_v;
        f.addr = addr;
        // the picture origin is folded into dst_plane/dst_tab for the scaler; undo it for the canvas
        const uint32_t shift = uint32_t(c.pad_y) * uint32_t(dst_stride) + uint32_t(c.pad_x);
        for (int k = 0; k < 2; ++k) {
            f.addr.dst_plane[k] -= shift;
            for (int i = 0; i < kMaxTableFrames; ++i) f.addr.dst_tab[k][i] -= shift;
        }
        dim3 grid((c.canvas_w + 255) / 256, (c.canvas_h + 3) / 4, n * planes);
        hipLaunchKernelGGL(fill_kernel, grid, dim3(kBlock), 0, st, f);
    }

    if (c.kind == 1) {
        StripArgs a{};
        a.src_span = src_span; a.dst_span = dst_span;
        a.src_stride = src_stride; a.dst_stride = dst_stride;
        a.ow = c.ow; a.oh = c.oh;
        a.strips = c.strips; a.planes = planes; a.n_items = n * planes * c.strips;
        const int nsb = (a.n_items + kBlock - 1) / kBlock;
        int seg_rows = env_int("TTS_SEG_ROWS", 0);
        if (seg_rows <= 0) {
            // enough row segments to put >= ~8 waves on every SIMD, but not shorter than 16 rows
            seg_rows = 64;
            while (seg_rows > 16 && int64_t(nsb) * ((c.oh + seg_rows - 1) / seg_rows) < 256 * 8) seg_rows >>= 1;
        }
        a.seg_rows = seg_rows; a.nseg = (c.oh + seg_rows - 1) / seg_rows;
        a.addr = addr;
        strip_fn fn = pick_strip(c.LW, c.R, c.N);
        if (!fn) return TTS_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(fn, dim3(uint32_t(nsb) * uint32_t(a.nseg)), dim3(kBlock), 0, st, src, dst, c.d_hoff, c.d_hcoef, c.d_vtab, a);
    } else {
        GenericArgs a{};
        a.src = src; a.dst = dst;
        a.hpos = c.d_hpos; a.htaps = c.d_htaps; a.vpos = c.d_vpos; a.vtaps = c.d_vtaps;
        a.ht = c.h.size; a.vt = c.v.size;
        a.x0 = c.x0; a.y0 = c.y0; a.ow = c.ow; a.oh = c.oh;
        a.src_stride = src_stride; a.dst_stride = dst_stride;
        a.planes = planes; a.n_fp = n * planes;
        a.addr = addr;
        dim3 grid((c.ow + 63) / 64, (c.oh + 3) / 4, n * planes);
        hipLaunchKernelGGL(generic_kernel, grid, dim3(kBlock), 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? TTS_OK : TTS_ERR_PROCESSING_FAILED;
}

}  // namespace

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
    for (int ci = 0; ci < 2 && rc == TTS_OK; ++ci) rc = build_class(*g, ci, p->cls[ci]);
    if (rc) {
        if (rc == TTS_ERR_INITIALIZATION_FAILED) g_last_error = std::string("table upload: ") + hipGetErrorString(hipGetLastError());
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
    delete p;
}

int tts_plan_geom(const tts_plan *p, tts_geom *out)
{
    if (!p || !out) return TTS_ERR_INVALID_INPUT;
    *out = p->g;
    return TTS_OK;
}

int tts_plan_filter(const tts_plan *p, int plane_class, int axis, int *taps, int *n_out, int16_t *coef, int32_t *pos)
{
    if (!p || plane_class < 0 || plane_class > 1 || axis < 0 || axis > 1) return TTS_ERR_INVALID_INPUT;
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
    return p->cls[plane_class].kind;
}

size_t tts_frame_bytes(int w, int h)
{
    if (w < 1 || h < 1) return 0;
    return size_t(w) * h + 2 * size_t((w + 1) / 2) * ((h + 1) / 2);
}

int tts_run_batch(const tts_plan *p, const void *src, size_t src_pitch, void *dst, size_t dst_pitch,
                  int n_frames, void *hip_stream)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    const tts_geom &g = p->g;
    if (src_pitch < tts_frame_bytes(g.in_w, g.in_h) || dst_pitch < tts_frame_bytes(g.out_w, g.out_h)) return TTS_ERR_INVALID_INPUT;
    if (src_pitch >= (size_t(1) << 31) || dst_pitch >= (size_t(1) << 31)) return TTS_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // 32-bit lane offsets: keep each launch's span under 4 GiB
    const size_t span = size_t(0xF0000000u);
    const int chunk = int(std::max<size_t>(1, std::min(span / src_pitch, span / dst_pitch)));
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int n = std::min(chunk, n_frames - f0);
        const uint8_t *s = static_cast<const uint8_t *>(src) + size_t(f0) * src_pitch;
        uint8_t *d = static_cast<uint8_t *>(dst) + size_t(f0) * dst_pitch;
        for (int ci = 0; ci < 2; ++ci) {
            const ClassPlan &c = p->cls[ci];
            const int sstride = ceil_shift(g.in_w, ci), dstride = ceil_shift(g.out_w, ci);
            PlaneAddr addr{};
            addr.src_pitch = uint32_t(src_pitch); addr.dst_pitch = uint32_t(dst_pitch);
            const uint32_t s_y = uint32_t(g.in_w) * g.in_h, s_c = uint32_t(ceil_shift(g.in_w, 1)) * ceil_shift(g.in_h, 1);
            const uint32_t d_y = uint32_t(g.out_w) * g.out_h, d_c = uint32_t(ceil_shift(g.out_w, 1)) * ceil_shift(g.out_h, 1);
            const uint32_t s_in = uint32_t(g.pre_y >> ci) * sstride + uint32_t(g.pre_x >> ci);
            const uint32_t d_in = uint32_t(c.pad_y) * dstride + uint32_t(c.pad_x);
            if (ci == 0) { addr.src_plane[0] = s_in; addr.dst_plane[0] = d_in; }
            else {
                addr.src_plane[0] = s_y + s_in;       addr.dst_plane[0] = d_y + d_in;
                addr.src_plane[1] = s_y + s_c + s_in; addr.dst_plane[1] = d_y + d_c + d_in;
            }
            const uint32_t sspan = uint32_t(size_t(n - 1) * src_pitch + tts_frame_bytes(g.in_w, g.in_h));
            const uint32_t dspan = uint32_t(size_t(n - 1) * dst_pitch + tts_frame_bytes(g.out_w, g.out_h));
            const int rc = launch_class(p, ci, s, d, n, sstride, dstride, sspan, dspan, addr, st);
            if (rc) return rc;
        }
    }
    return TTS_OK;
}

int tts_run(const tts_plan *p, const tts_frame *src, const tts_frame *dst, int n_frames, void *hip_stream)
{
    if (!p || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    if (!src || !dst) return TTS_ERR_INVALID_INPUT;
    const tts_geom &g = p->g;
    for (int i = 0; i < n_frames; ++i) {
        if (src[i].mem != TTS_MEM_DEVICE || dst[i].mem != TTS_MEM_DEVICE) return TTS_ERR_INVALID_INPUT;
        if (src[i].width != g.in_w || src[i].height != g.in_h || dst[i].width != g.out_w || dst[i].height != g.out_h) return TTS_ERR_INVALID_INPUT;
        for (int k = 0; k < 3; ++k) {
            if (!src[i].plane[k] || !dst[i].plane[k]) return TTS_ERR_INVALID_INPUT;
            if (src[i].stride[k] < ceil_shift(g.in_w, k ? 1 : 0) || dst[i].stride[k] < ceil_shift(g.out_w, k ? 1 : 0)) return TTS_ERR_INVALID_INPUT;
        }
        if (src[i].stride[1] != src[i].stride[2] || dst[i].stride[1] != dst[i].stride[2]) return TTS_ERR_UNSUPPORTED;
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    int f0 = 0;
    while (f0 < n_frames) {
        // group consecutive frames with equal strides whose planes lie within 4 GiB of a common base
        uintptr_t smin = UINTPTR_MAX, dmin = UINTPTR_MAX, smax = 0, dmax = 0;
        int n = 0;
        while (f0 + n < n_frames && n < kMaxTableFrames) {
            const tts_frame &s = src[f0 + n], &d = dst[f0 + n];
            if (n && (s.stride[0] != src[f0].stride[0] || s.stride[1] != src[f0].stride[1] ||
                      d.stride[0] != dst[f0].stride[0] || d.stride[1] != dst[f0].stride[1])) break;
            uintptr_t a = smin, b = smax, c2 = dmin, e = dmax;
            for (int k = 0; k < 3; ++k) {
                a = std::min(a, uintptr_t(s.plane[k])); b = std::max(b, uintptr_t(s.plane[k]));
                c2 = std::min(c2, uintptr_t(d.plane[k])); e = std::max(e, uintptr_t(d.plane[k]));
            }
            if (n && (b - a >= 0xE0000000u || e - c2 >= 0xE0000000u)) break;
            smin = a; smax = b; dmin = c2; dmax = e;
            ++n;
        }
        for (int ci = 0; ci < 2; ++ci) {
            const ClassPlan &c = p->cls[ci];
            const int sstride = src[f0].stride[ci], dstride = dst[f0].stride[ci];
            PlaneAddr addr{};
            addr.use_table = 1;
            const uint32_t s_in = uint32_t(g.pre_y >> ci) * sstride + uint32_t(g.pre_x >> ci);
            const uint32_t d_in = uint32_t(c.pad_y) * dstride + uint32_t(c.pad_x);
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < (ci ? 2 : 1); ++k) {
                    addr.src_tab[k][i] = uint32_t(uintptr_t(src[f0 + i].plane[ci + k]) - smin) + s_in;
                    addr.dst_tab[k][i] = uint32_t(uintptr_t(dst[f0 + i].plane[ci + k]) - dmin) + d_in;
                }
            // bytes reachable from the common bases: furthest plane end of the group
            size_t sspan = 0, dspan = 0;
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) {
                    const int sub = k ? 1 : 0;
                    sspan = std::max(sspan, size_t(uintptr_t(src[f0 + i].plane[k]) - smin) +
                                            size_t(src[f0 + i].stride[k]) * (ceil_shift(g.in_h, sub) - 1) + ceil_shift(g.in_w, sub));
                    dspan = std::max(dspan, size_t(uintptr_t(dst[f0 + i].plane[k]) - dmin) +
                                            size_t(dst[f0 + i].stride[k]) * (ceil_shift(g.out_h, sub) - 1) + ceil_shift(g.out_w, sub));
                }
            const int rc = launch_class(p, ci, reinterpret_cast<const uint8_t *>(smin), reinterpret_cast<uint8_t *>(dmin),
                                        n, sstride, dstride, uint32_t(std::min<size_t>(sspan, 0xFFFFFFFFu)),
                                        uint32_t(std::min<size_t>(dspan, 0xFFFFFFFFu)), addr, st);
            if (rc) return rc;
        }
        f0 += n;
    }
    return TTS_OK;
}

int tts_run_host(const tts_plan *p, const void *src, void *dst, int n_frames)
{
    if (!p || !src || !dst || n_frames < 0) return TTS_ERR_INVALID_INPUT;
    if (n_frames == 0) return TTS_OK;
    const size_t sb = tts_frame_bytes(p->g.in_w, p->g.in_h), db = tts_frame_bytes(p->g.out_w, p->g.out_h);
    void *ds = nullptr, *dd = nullptr;
    int rc = TTS_OK;
    if (hipMalloc(&ds, sb * n_frames) != hipSuccess || hipMalloc(&dd, db * n_frames) != hipSuccess) rc = TTS_ERR_INITIALIZATION_FAILED;
    if (!rc && hipMemcpy(ds, src, sb * n_frames, hipMemcpyHostToDevice) != hipSuccess) rc = TTS_ERR_PROCESSING_FAILED;
    if (!rc) rc = tts_run_batch(p, ds, sb, dd, db, n_frames, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = TTS_ERR_PROCESSING_FAILED;
    if (!rc && hipMemcpy(dst, dd, db * n_frames, hipMemcpyDeviceToHost) != hipSuccess) rc = TTS_ERR_PROCESSING_FAILED;
    (void)hipFree(ds); (void)hipFree(dd);
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
