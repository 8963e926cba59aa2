// ttscale_rgbin_shrink.hip -- packed rgb24 in, yuv420p out, the picture SHRUNK (any ratio down the picture; up to 50 taps
// across: bicubic down to a twelfth, the reference's thumbnail chain on a 4K source): one walk over the source rows.
//
// swscale reads chroma at half width here (filter.h rgb_in_half_chroma(): rgb24ToUV_half_c on pixel pairs), at every
// source row: the chroma planes shrink by the picture's ratio across and by TWICE that down the picture (16 and more
// vertical taps), which is why the tiled kernel spent its time between barriers (DESIGN.md 7 finding 5).  Here, as in
// rgbin_strip_kernel, lanes are the flattened (frame, strip) index and a wave converts the runs of source pixels its
// strips need ONCE per source row into its own LDS rows (Y per pixel, U / V per pixel pair; exact float forms, each row
// stored twice so that every tap pair of every window is one aligned dword); a lane then owns 4 luma columns and the
// 2 + 2 chroma columns under them: hScale16To15_c is TPH v_dot2 per column.  The vertical pass is OUTPUT-STATIONARY: a
// source row's 15-bit result is multiplied into the sums of the (at most 8) output rows whose filters hold that row --
// wave-uniform taps from a host table per source row, v_mad_i32_i24 -- and an output row leaves, (sum + 64 << 12) >> 19
// clipped, when its last source row has gone in; the sums then move up one place.  No ring, so no limit on the number
// of vertical taps and no unrolling by a ring's depth.  No barrier: a wave's LDS rows are its own.
// More than 8 taps across (NL = 2): a lane owns 2 luma columns and the U and the V column under them, converts 16 pixels a
// row, and keeps 3 TPH tap pairs in registers; a pad behind the scale (src/services/video.js:331) is a border-fill launch of
// its own in front (rgbin_pad_fill_kernel) and an offset on the picture's rows.
#include "ttscale_rgb.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <tuple>
#include <vector>

#include "filter.h"

namespace {

constexpr int kPend = 8;               // output rows a source row may contribute to, at most (host-checked; the kernel keeps KP <= kPend sums a column)
constexpr int kRowEntry = 4 + 2 * kPend;   // dwords of a source row's table entry

struct RgbInShrinkArgs {
    uint32_t o_strip, o_wave, o_rows;  // dword offsets into tab
    int32_t strips, spw, total, nframes;   // 4-column strips a row; strips a wave; lanes = nframes x strips
    int32_t h0, h1;                    // source rows [h0, h1) of the pre-cropped picture are needed
    int32_t seg_rows, nseg;
    int32_t src_stride, dst_stride[2];
    uint32_t src_origin;
    uint32_t dst_origin[2];            // the picture's origin inside the canvas (a pad), luma / chroma planes, in bytes
    int32_t w128;                      // every lane's piece of a row starts on a multiple of 8 samples (8 pixels a lane): vector LDS writes
    int32_t range15;                   // the converting context writes yuvj420p (tts_geom.sc_range): lum / chrRangeToJpeg_c on the 15-bit rows
    float cy[4];                       // rgb24ToY_c: coefficients of R, G, B x 2^-9 and (2 bias + 1) / 1024
    float cu[4], cv[4];                // rgb24ToUV_half_c: coefficients x 2^-10 and 2^-11 (see the kernel)
    FrameAddr addr;
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) int32_t *ctab_t;   // wave-uniform reads of it are scalar loads

// PXL: pixels of the wave's runs a lane converts per row (4 or 8); TPH: tap pairs of a horizontal window (taps <= 6, 8);
// KP: sums kept per column = the most output rows any source row of this geometry contributes to (6: 16 registers fewer
// than 8, which is the fourth wave a SIMD)
// NL: luma columns a lane owns (4; 2 for the many-tap windows), NL / 2 columns of each chroma plane under them
// PB: bytes a source pixel (3; 4 for rgba / bgra: the fourth byte is not read)
template <int PXL, int TPH, int KP, int NL = 4, int PB = 3>
__global__ __launch_bounds__(kBlock, NL == 2 ? 2 : KP == 6 ? 4 : 3) void rgbin_shrink_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                 const int32_t *__restrict__ tab, const RgbInShrinkArgs a)
{
    static_assert((PXL == 4 || PXL == 8 || PXL == 16) && TPH >= 3 && TPH <= 25 && (NL == 4 || NL == 2), "pixels a lane converts; tap pairs a window; luma columns a lane");
    constexpr int NC = NL / 2;                                       // chroma columns a lane (of each plane)
    static_assert(PB == 3 || PB == 4, "packed pixels of 3 or 4 bytes");
    constexpr int NP = PXL / 2, NPC = PXL / 4, NDW = (PB * PXL + 6) / 4;   // pixel pairs = luma dwords; chroma dwords; aligned dwords that hold a lane's PB PXL bytes at any phase
    constexpr int CBL = 32 * PXL + 8, CBC = 16 * PXL + 8;            // dwords of one copy of a luma / chroma row (+ slack: windows read whole tap pairs; the odd copy starts one dword in)
    constexpr int ST = NL + NC + (NL + NC) * TPH + 1;                // words of a strip's table entry
    // every row twice: as pairs (s0 s1)(s2 s3).. and, one halfword on, as (s1 s2)(s3 s4)..: a window that starts on an
    // odd sample reads the second copy, so that every tap pair of every window is ONE aligned dword (no v_perm)
    __shared__ __attribute__((aligned(16))) uint32_t ldsy[kBlock / 64][2 * CBL];
    __shared__ __attribute__((aligned(16))) uint32_t ldsc[kBlock / 64][2][2 * CBC];
    static_assert((2 * CBL) % 4 == 0 && CBL % 4 == 0 && (2 * CBC) % 4 == 0 && CBC % 2 == 0, "a lane's 16-byte (luma) and 8-byte (chroma) row pieces stay aligned in both copies");
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), lane = int(threadIdx.x & 63);
    const int wid = int(blockIdx.x) * (kBlock / 64) + wave;
    const int g0 = wid * a.spw, seg = int(blockIdx.y);
    if (g0 >= a.total) return;                                       // (whole waves; no barrier in this kernel)
    const int nlive = min(a.spw, a.total - g0);
    const bool live = lane < nlive;
    const int lc = min(lane, nlive - 1);                             // dead lanes shadow the last live one and store nothing
    const int fA = g0 / a.strips, sA = g0 - fA * a.strips;           // wave-uniform: the wave's first frame, its first strip there
    const int32_t *WC = tab + a.o_wave + 4 * (size_t(sA) * 64 + lane), *WS = tab + a.o_wave + 4 * (size_t(sA) * 64 + lc);
    const int cpx = WC[0], cs = WC[1] >> 8, cfr = fA + (WC[1] & 255);  // what this lane converts: first pixel (< 0: nothing), its LDS sample index (a multiple of 4), frame
    const int frame = fA + WS[2], adj = WS[3];                       // this lane's strip: its frame, LDS sample of pixel 0 of that frame's row (even)
    const int strip = sA + lc - WS[2] * a.strips;
    const bool conv = cpx >= 0 && cfr < a.nframes;
    // source bytes are addressed from the wave's first frame (a wave-uniform base) by 32-bit lane offsets
    const uint8_t *sbase;
    uint32_t loff;
    uint8_t *dy, *du, *dv;
    if (a.addr.use_table) {
        sbase = src;
        loff = a.addr.src_tab[0][conv ? cfr : fA];
        dy = dst + a.addr.dst_tab[0][frame]; du = dst + a.addr.dst_tab[1][frame]; dv = dst + a.addr.dst_tab[2][frame];
    } else {
        sbase = src + size_t(fA) * a.addr.src_pitch + a.addr.src_plane[0];
        loff = uint32_t(conv ? cfr - fA : 0) * a.addr.src_pitch;
        uint8_t *dpf = dst + size_t(frame) * a.addr.dst_pitch;
        dy = dpf + a.addr.dst_plane[0]; du = dpf + a.addr.dst_plane[1]; dv = dpf + a.addr.dst_plane[2];
    }
    const uint32_t mis = uint32_t(reinterpret_cast<uintptr_t>(sbase) & 3u);
    const uint8_t *sbase4 = sbase - mis;
    const uint32_t off0 = mis + loff + a.src_origin + uint32_t(PB) * uint32_t(conv ? cpx : 0);
    // this lane's windows and taps
    int li[NL], ci[NC];                                              // first dword of each window, in the copy its parity picks
    uint32_t lt[NL][TPH], ct[NC][TPH];
    {
        const int32_t *T = tab + a.o_strip + size_t(ST) * strip;
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            const int h = T[c] + adj;
            li[c] = (h & 1) ? CBL + 1 + (h >> 1) : (h >> 1);
#pragma unroll
            for (int k = 0; k < TPH; ++k) lt[c][k] = uint32_t(T[NL + TPH * c + k]);
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int h = T[NL + NL * TPH + j] + (adj >> 1);
            ci[j] = (h & 1) ? CBC + 1 + (h >> 1) : (h >> 1);
#pragma unroll
            for (int k = 0; k < TPH; ++k) ct[j][k] = uint32_t(T[NL + NC + NL * TPH + TPH * j + k]);
        }
        // first output column of the strip: the last strip of a row is slid left, not cut short (the host tabulates it)
        const int col = T[ST - 1];
        dy += a.dst_origin[0] + uint32_t(col); du += a.dst_origin[1] + uint32_t(col >> 1); dv += a.dst_origin[1] + uint32_t(col >> 1);
    }
    const int ra = a.h0 + seg * a.seg_rows, rb = min(ra + a.seg_rows, a.h1);
    ctab_t rows = (ctab_t)(tab + a.o_rows);
    const int rstart = rows[size_t(kRowEntry) * (ra - a.h0) + 3];
    const f32x2_t yR = a.cy[0], yG = a.cy[1], yB = a.cy[2], y0 = a.cy[3];
    const f32x2_t cR = {a.cu[0], a.cv[0]}, cG = {a.cu[1], a.cv[1]}, cB = {a.cu[2], a.cv[2]}, c0 = {a.cu[3], a.cv[3]};
    constexpr int kBias = 64 << 12;
    int accl[NL][KP], accu[NC][KP], accv[NC][KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
#pragma unroll
        for (int c = 0; c < NL; ++c) accl[c][k] = kBias;
#pragma unroll
        for (int j = 0; j < NC; ++j) accu[j][k] = accv[j][k] = kBias;
    }
    // A lane's 3 PXL bytes of a source row as the aligned dwords that hold them.  Every dword read holds at least one byte
    // of the lane's pixels (the host keeps those inside the frame), so it lies in a mapped page: the last of the NDW is
    // needed only at the larger byte phases, elsewhere the one before it is read again.
    auto fetch = [&](int r, uint32_t (&raw)[NDW]) {
        const uint32_t o = off0 + uint32_t(r) * uint32_t(a.src_stride), oa = o & ~3u;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(sbase4 + oa);
#pragma unroll
        for (int k = 0; k < NDW - 1; ++k) raw[k] = q[k];
        const bool need_last = (o & 3u) + uint32_t(PB) * PXL > 4u * (NDW - 1);
        raw[NDW - 1] = *reinterpret_cast<const uint32_t *>(sbase4 + (oa + (need_last ? 4u * (NDW - 1) : 4u * (NDW - 2))));
    };
    uint32_t nx[NDW];
    fetch(rstart, nx);
    int32_t E[kRowEntry];                                            // the row's table entry (scalars), read one row ahead
    {
        ctab_t q = rows + size_t(kRowEntry) * (rstart - a.h0);
#pragma unroll
        for (int k = 0; k < kRowEntry; ++k) E[k] = q[k];
    }
    uint32_t *Y = ldsy[wave], *U = ldsc[wave][0], *V = ldsc[wave][1];
    for (int r = rstart; r < rb; ++r) {
        int32_t e[kRowEntry];
#pragma unroll
        for (int k = 0; k < kRowEntry; ++k) e[k] = E[k];
        {
            ctab_t q = rows + size_t(kRowEntry) * (min(r + 1, a.h1 - 1) - a.h0);
#pragma unroll
            for (int k = 0; k < kRowEntry; ++k) E[k] = q[k];
        }
        // ---- convert this lane's PXL pixels of the wave's runs
        {
            const uint32_t ph = (off0 + uint32_t(r) * uint32_t(a.src_stride)) & 3u;
            uint32_t w[NDW - 1];
#pragma unroll
            for (int k = 0; k < NDW - 1; ++k) w[k] = __builtin_amdgcn_alignbyte(nx[k + 1], nx[k], ph);
            fetch(min(r + 1, rb - 1), nx);
            auto ub = [&](int b) { return float((w[b >> 2] >> (8 * (b & 3))) & 255u); };   // v_cvt_f32_ubyteN
            uint32_t pky[NP], pku[NPC], pkv[NPC];
            float tu[NP], tv[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const f32x2_t R = {ub(2 * PB * i), ub(2 * PB * i + PB)}, G = {ub(2 * PB * i + 1), ub(2 * PB * i + PB + 1)}, B = {ub(2 * PB * i + 2), ub(2 * PB * i + PB + 2)};
                // rgb24ToY_c: (ry r + gy g + by b + (32 << 14) + (1 << 8)) >> 9: 8-bit channels times 15-bit coefficients x 2^-9
                // are exact in fp32; adding 2^23 - 1/2 to (2 sum + 1) / 1024 leaves floor(sum / 512) in the low mantissa bits
                f32x2_t t = R * yR + (G * yG + (B * yB + y0));
                t += 8388607.5f;
                const float t0 = t.x, t1 = t.y;                      // (__builtin_bit_cast of a vector ELEMENT reads element 0 whichever is named: copy out first)
                pky[i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, t1), __builtin_bit_cast(uint32_t, t0), 0x05040100);
                // rgb24ToUV_half_c on the pair: (ru (r0 + r1) + gu (g0 + g1) + bu (b0 + b1) + (256 << 15) + (1 << 9)) >> 10
                // = 8192 + round-half-up(S / 1024).  S / 1024 + 2^-11 is exact in fp32 (|S| / 1024 <= 7168, steps of 2^-11)
                // and never on a tie, so adding 1.5 x 2^23 + 8192 rounds it to the nearest integer, which for k / 1024 = 1/2
                // is the upper one: the sample sits in the low mantissa bits.
                const float sr = R.x + R.y, sg = G.x + G.y, sb = B.x + B.y;
                f32x2_t c = f32x2_t{sr, sr} * cR + (f32x2_t{sg, sg} * cG + (f32x2_t{sb, sb} * cB + c0));
                c += 12591104.0f;
                tu[i] = c.x; tv[i] = c.y;
            }
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                pku[i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, tu[2 * i + 1]), __builtin_bit_cast(uint32_t, tu[2 * i]), 0x05040100);
                pkv[i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, tv[2 * i + 1]), __builtin_bit_cast(uint32_t, tv[2 * i]), 0x05040100);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the previous row's window reads are done)
            __builtin_amdgcn_wave_barrier();
            if (PXL >= 8 && a.w128) {
                // Every lane's piece of the row starts on a multiple of 8 samples (the host says so: a.w128): ds_write_b128 for each
                // four dwords of a copy (b64 for the chroma rows) instead of dword writes 4 or 8 dwords apart, a 4- / 8-way bank
                // conflict each.  The odd copy's first dword is (s[-1], s[0]): the low half written here is not this lane's --
                // the neighbour's halfword store, the NEXT instruction of every lane, puts the right sample there.
                if (conv) {
                    if constexpr (PXL >= 8) {
                        uint32_t *ea = Y + (cs >> 1), *ob = Y + CBL + (cs >> 1);
#pragma unroll
                        for (int q = 0; q < NP / 4; ++q)
                            *static_cast<u32x4_t *>(__builtin_assume_aligned(ea + 4 * q, 16)) = u32x4_t{pky[4 * q], pky[4 * q + 1], pky[4 * q + 2], pky[4 * q + 3]};
                        uint32_t od[NP];                                 // the odd copy's dwords of this lane: (s[2m - 1], s[2m])
                        od[0] = pky[0] << 16;
#pragma unroll
                        for (int k = 1; k < NP; ++k) od[k] = __builtin_amdgcn_perm(pky[k], pky[k - 1], 0x05040302);
#pragma unroll
                        for (int q = 0; q < NP / 4; ++q)
                            *static_cast<u32x4_t *>(__builtin_assume_aligned(ob + 4 * q, 16)) = u32x4_t{od[4 * q], od[4 * q + 1], od[4 * q + 2], od[4 * q + 3]};
                        asm volatile("" ::: "memory");               // (the halfword store after the vector store that covers its neighbour's)
                        reinterpret_cast<uint16_t *>(ob + NP)[0] = uint16_t(pky[NP - 1] >> 16);
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const uint32_t (&pk)[NPC] = p ? pkv : pku;
                            uint32_t *ce = (p ? V : U) + (cs >> 2), *co = (p ? V : U) + CBC + (cs >> 2);
                            uint32_t oc[NPC];
                            oc[0] = pk[0] << 16;
#pragma unroll
                            for (int k = 1; k < NPC; ++k) oc[k] = __builtin_amdgcn_perm(pk[k], pk[k - 1], 0x05040302);
#pragma unroll
                            for (int q = 0; q < NPC / 2; ++q) {
                                *static_cast<u32x2_t *>(__builtin_assume_aligned(ce + 2 * q, 8)) = u32x2_t{pk[2 * q], pk[2 * q + 1]};
                                *static_cast<u32x2_t *>(__builtin_assume_aligned(co + 2 * q, 8)) = u32x2_t{oc[2 * q], oc[2 * q + 1]};
                            }
                            asm volatile("" ::: "memory");
                            reinterpret_cast<uint16_t *>(co + NPC)[0] = uint16_t(pk[NPC - 1] >> 16);
                        }
                    }
                }
            } else if (conv) {
                // even copy, then the odd one: dword m = (s[2m + 1], s[2m + 2]); its first and last halfword belong to the neighbours' dwords
                {
                    uint32_t *ea = Y + (cs >> 1), *ob = Y + CBL + 1 + (cs >> 1);
#pragma unroll
                    for (int k = 0; k < NP; ++k) ea[k] = pky[k];
                    reinterpret_cast<uint16_t *>(ob - 1)[1] = uint16_t(pky[0]);
#pragma unroll
                    for (int k = 0; k < NP - 1; ++k) ob[k] = __builtin_amdgcn_perm(pky[k + 1], pky[k], 0x05040302);
                    reinterpret_cast<uint16_t *>(ob + NP - 1)[0] = uint16_t(pky[NP - 1] >> 16);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const uint32_t (&pk)[NPC] = p ? pkv : pku;
                    uint32_t *ea = (p ? V : U) + (cs >> 2), *ob = (p ? V : U) + CBC + 1 + (cs >> 2);
#pragma unroll
                    for (int k = 0; k < NPC; ++k) ea[k] = pk[k];
                    reinterpret_cast<uint16_t *>(ob - 1)[1] = uint16_t(pk[0]);
#pragma unroll
                    for (int k = 0; k < NPC - 1; ++k) ob[k] = __builtin_amdgcn_perm(pk[k + 1], pk[k], 0x05040302);
                    reinterpret_cast<uint16_t *>(ob + NPC - 1)[0] = uint16_t(pk[NPC - 1] >> 16);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- hScale16To15_c from the wave's rows
        int hl[NL], hu[NC], hv[NC];
#pragma unroll
        for (int c = 0; c < NL; ++c) {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < TPH; ++k) acc = dot2(Y[li[c] + k], lt[c][k], acc);
            hl[c] = min(acc >> 13, 32767);
        }
        if (a.range15) {                                             // wave-uniform; swscale.c lumRangeToJpeg_c
#pragma unroll
            for (int c = 0; c < NL; ++c) hl[c] = (__mul24(min(hl[c], 30189), 19077) - 39057361) >> 14;
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            int au = 0, av = 0;
#pragma unroll
            for (int k = 0; k < TPH; ++k) { au = dot2(U[ci[j] + k], ct[j][k], au); av = dot2(V[ci[j] + k], ct[j][k], av); }
            hu[j] = min(au >> 13, 32767); hv[j] = min(av >> 13, 32767);
        }
        if (a.range15) {                                             // chrRangeToJpeg_c
#pragma unroll
            for (int j = 0; j < NC; ++j) { hu[j] = (__mul24(min(hu[j], 30775), 4663) - 9289992) >> 12; hv[j] = (__mul24(min(hv[j], 30775), 4663) - 9289992) >> 12; }
        }
        // ---- vertical pass, output-stationary: this row into the sums of the output rows that hold it
        const int npl = (e[2] >> 16) & 255, npc = (e[2] >> 24) & 255;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (k < 4 || k < npl) {                                  // wave-uniform (the first four always: zero taps where nothing is pending)
#pragma unroll
                for (int c = 0; c < NL; ++c) accl[c][k] = __mul24(hl[c], e[4 + k]) + accl[c][k];
            }
            if (k < 4 || k < npc) {
#pragma unroll
                for (int j = 0; j < NC; ++j) { accu[j][k] = __mul24(hu[j], e[4 + kPend + k]) + accu[j][k]; accv[j][k] = __mul24(hv[j], e[4 + kPend + k]) + accv[j][k]; }
            }
        }
        // ---- output rows this source row completes (stored by the segment that owns the row)
        const int ncl = e[2] & 255, ncc = (e[2] >> 8) & 255;
        for (int i = 0; i < ncl; ++i) {                              // wave-uniform
            uint32_t o;
            if constexpr (NL == 4) o = pack4<19>(accl[0][0], accl[1][0], accl[2][0], accl[3][0]);
            else o = pack2(accl[0][0], accl[1][0]);
            if (live && r >= ra) {
                uint8_t *q = dy + size_t(e[0] + i) * a.dst_stride[0];
                typedef uint32_t u32_b __attribute__((aligned(2)));   // (an output width of 4 k + 2: the slid last strip starts on a halfword)
                if constexpr (NL == 4) *reinterpret_cast<u32_b *>(q) = o;
                else *reinterpret_cast<uint16_t *>(q) = uint16_t(o);
            }
#pragma unroll
            for (int c = 0; c < NL; ++c) {
#pragma unroll
                for (int k = 0; k < KP - 1; ++k) accl[c][k] = accl[c][k + 1];
                accl[c][KP - 1] = kBias;
            }
        }
        for (int i = 0; i < ncc; ++i) {
            const uint32_t ou = pack2(accu[0][0], accu[NC - 1][0]), ov = pack2(accv[0][0], accv[NC - 1][0]);
            if (live && r >= ra) {
                const size_t ro = size_t(e[1] + i) * a.dst_stride[1];
                typedef uint16_t u16_b __attribute__((aligned(1)));   // (chroma columns: any byte)
                if constexpr (NC == 2) { *reinterpret_cast<u16_b *>(du + ro) = uint16_t(ou); *reinterpret_cast<u16_b *>(dv + ro) = uint16_t(ov); }
                else { du[ro] = uint8_t(ou); dv[ro] = uint8_t(ov); }
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
#pragma unroll
                for (int k = 0; k < KP - 1; ++k) { accu[j][k] = accu[j][k + 1]; accv[j][k] = accv[j][k + 1]; }
                accu[j][KP - 1] = accv[j][KP - 1] = kBias;
            }
        }
    }
}

// The border of a padded canvas (a pad behind the scale), every plane of every frame: four bytes a lane, the picture's
// rectangle left alone.
struct RgbInPadFillArgs {
    uint8_t *dst;
    int32_t w[2], h[2], stride[2];     // canvas of a luma / chroma plane
    int32_t px[2], py[2], pw[2], ph[2];   // the picture's rectangle in it
    int32_t fill[3];
    int32_t frame0;
    FrameAddr addr;
};

__global__ __launch_bounds__(kBlock) void rgbin_pad_fill_kernel(const RgbInPadFillArgs a)
{
    const int plane = int(blockIdx.z) % 3, frame = a.frame0 + int(blockIdx.z) / 3, k = plane ? 1 : 0;
    const int x = (int(blockIdx.x) * 64 + int(threadIdx.x & 63)) * 4, y = int(blockIdx.y) * 4 + int(threadIdx.x >> 6);
    if (x >= a.w[k] || y >= a.h[k]) return;
    uint8_t *row = a.dst + (a.addr.use_table ? size_t(a.addr.dst_tab[plane][frame]) : size_t(frame) * a.addr.dst_pitch + a.addr.dst_plane[plane]) + size_t(y) * size_t(a.stride[k]);
    const bool in_rows = y >= a.py[k] && y < a.py[k] + a.ph[k];
    const uint8_t f = uint8_t(a.fill[plane]);
    for (int i = x; i < min(x + 4, a.w[k]); ++i)
        if (!(in_rows && i >= a.px[k] && i < a.px[k] + a.pw[k])) row[i] = f;
}

template <class... P, class... A>
int launch_shrink(const char *what, void (*fn)(P...), dim3 grid, dim3 block, hipStream_t st, A &&...args)
{
    std::tuple<P...> held(std::forward<A>(args)...);
    void *ptrs[sizeof...(P)];
    size_t i = 0;
    std::apply([&](auto &...v) { ((ptrs[i++] = const_cast<void *>(static_cast<const void *>(&v))), ...); }, held);
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(fn), grid, block, ptrs, 0, st);
    if (e == hipSuccess) return TTS_OK;
    tts::set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    return TTS_ERR_PROCESSING_FAILED;
}

}  // namespace

namespace tts {

struct RgbInShrinkPlan {
    tts_geom g{};
    int32_t *d_tab = nullptr;
    RgbInShrinkArgs args{};            // table offsets and constants; per-launch fields are filled in by rgbin_shrink_launch
    int pxl = 8, tph = 4, kp = 8, nl = 4, pb = 3;
    bool pad = false;                  // a pad behind the scale: rgbin_pad_fill_kernel in front, the picture at its offset
    int vt_max = 0;                    // the taller of the two vertical filters
    int seg_knob = 0;                  // TTS_RGBIN_SEG at plan creation: row segments a launch (0: the launch decides)
    char name[96] = "";
};

RgbInShrinkPlan *rgbin_shrink_create(const tts_geom &g, const AxisFilter &lh, const AxisFilter &lv, const AxisFilter &ch, const AxisFilter &cv, const int coef[9], int pb)
{
    if (pb != 3 && pb != 4) return nullptr;
    auto env = [](const char *k) { const char *v = std::getenv(k); return v ? std::atoi(v) : 0; };
    if (env("TTS_FORCE_GENERIC") || env("TTS_NO_RGBIN_SHRINK")) return nullptr;
    if (!rgb_in_half_chroma(g.pre_w, g.sc_w) || (g.pre_w & 1)) return nullptr;
    if (!g.do_pad && (g.out_w != g.crop_w || g.out_h != g.crop_h)) return nullptr;
    // the picture inside its canvas (vf_pad: an odd picture loses its last column to the fill; offsets are even)
    const int ow = g.do_pad ? (g.crop_w & ~1) : g.out_w, oh = g.do_pad ? g.crop_h : g.out_h, och = (oh + 1) >> 1;
    if (g.do_pad && ((g.pad_x | g.pad_y) & 1)) return nullptr;
    if ((ow & 1) || ow < 16 || (g.crop_x & 1) || g.pre_h < 8 || ((g.in_w - g.pre_x) & 3)) return nullptr;   // (lanes convert on a grid of 4 pixels that must end with the row)
    if (lh.size > 50 || ch.size > 50) return nullptr;
    if (lh.size == 1 && lv.size == 1 && ch.size == 1 && pb == 3 && g.sc_range == TTS_RANGE_KEEP) return nullptr;   // (the conversion alone: rgbin_walk_kernel -- which reads 3-byte pixels and has no range step)
    auto monotone = [](const std::vector<int32_t> &v) { for (size_t i = 1; i < v.size(); ++i) if (v[i] < v[i - 1]) return false; return true; };
    if (!monotone(lh.pos) || !monotone(ch.pos) || !monotone(lv.pos) || !monotone(cv.pos)) return nullptr;
    const int ox = g.crop_x, oy = g.crop_y, oxc = g.crop_x >> 1, oyc = g.crop_y >> 1;
    if (ox + ow > int(lh.pos.size()) || oxc + ow / 2 > int(ch.pos.size()) || oy + oh > int(lv.pos.size()) || oyc + och > int(cv.pos.size())) return nullptr;
    const int lht = lh.size, cht = ch.size, lvt = lv.size, cvt = cv.size, mt = std::max(lht, cht);
    // <= 8 taps across: 4 luma columns a lane, windows of 3 or 4 tap pairs; more: 2 columns a lane, 7 .. 25 tap pairs
    const int nl = mt <= 8 ? 4 : 2, nc = nl / 2;
    const int tph = mt <= 6 ? 3 : mt <= 8 ? 4 : mt <= 14 ? 7 : mt <= 20 ? 10 : mt <= 26 ? 13 : mt <= 38 ? 19 : 25;
    const int strips = (ow + nl - 1) / nl, st = nl + nc + (nl + nc) * tph + 1;
    RgbInShrinkPlan *p = new (std::nothrow) RgbInShrinkPlan();
    if (!p) return nullptr;
    p->g = g; p->tph = tph; p->nl = nl; p->pad = g.do_pad != 0; p->pb = pb;
    RgbInShrinkArgs &a = p->args;
    std::vector<int32_t> t;
    auto pk = [](int a0, int a1) { return int32_t(uint32_t(uint16_t(int16_t(a0))) | uint32_t(uint16_t(int16_t(a1))) << 16); };
    // per strip: its 4 luma windows (first sample = a pixel of the pre-cropped row) and taps, its 2 chroma windows (first
    // sample = a pixel PAIR) and taps, its first output column
    std::vector<int> slo(size_t(strips), 0), shi(size_t(strips), 0);             // source pixels a strip's windows touch: [lo, hi), lo even
    a.o_strip = 0;
    t.resize(size_t(st) * strips, 0);
    for (int s = 0; s < strips; ++s) {
        int32_t *T = t.data() + size_t(st) * s;
        const int c0 = std::min(nl * s, ow - nl);                                // (even: ow is)
        int lo = INT_MAX, hi = 0;
        for (int c = 0; c < nl; ++c) {
            const int vx = ox + c0 + c;
            int tp[52] = {0};
            for (int k = 0; k < lht; ++k) tp[k] = lh.taps[size_t(vx) * lht + k];
            T[c] = lh.pos[vx];
            for (int k = 0; k < tph; ++k) T[nl + tph * c + k] = pk(tp[2 * k], tp[2 * k + 1]);
            lo = std::min(lo, lh.pos[vx]); hi = std::max(hi, lh.pos[vx] + lht);   // (a window reads whole tap pairs: past its taps, whatever the LDS row holds meets zero taps)
        }
        for (int j = 0; j < nc; ++j) {
            const int vx = oxc + c0 / 2 + j;
            int tp[52] = {0};
            for (int k = 0; k < cht; ++k) tp[k] = ch.taps[size_t(vx) * cht + k];
            T[nl + nl * tph + j] = ch.pos[vx];
            for (int k = 0; k < tph; ++k) T[nl + nc + nl * tph + tph * j + k] = pk(tp[2 * k], tp[2 * k + 1]);
            lo = std::min(lo, 2 * ch.pos[vx]); hi = std::max(hi, 2 * (ch.pos[vx] + cht));
        }
        T[st - 1] = c0;
        slo[s] = lo & ~3; shi[s] = std::min((hi + 1) & ~1, g.pre_w);
    }
    // per wave start sA (the wave's first strip inside its first frame) and lane: what the lane converts {first pixel or
    // -1, LDS sample index << 8 | frame - first frame} and where its strip lies {frame - first frame, LDS sample of pixel 0
    // of that frame's row}: a wave's strips are cut into one piece per frame they lie in, every piece's run of pixels is
    // dealt PXL a lane, piece after piece.  The wave holds as many strips (spw <= 64) as its 64 lanes can convert for.
    a.o_wave = uint32_t(t.size());
    t.resize(t.size() + size_t(4) * 64 * strips, 0);
    const int avail = g.in_w - g.pre_x;                                          // pixels from the pre-cropped origin to the end of a source row: a lane's PXL pixels stay inside them
    bool aligned8 = true;
    auto deal = [&](int pxl, int spw, bool write) {
        if (avail < pxl) return false;
        for (int sA = 0; sA < strips; ++sA) {
            int32_t *W = t.data() + a.o_wave + size_t(4) * 64 * sA;
            if (write) for (int i = 0; i < 64; ++i) { W[4 * i] = -1; W[4 * i + 1] = 0; W[4 * i + 2] = 0; W[4 * i + 3] = 0; }
            int used = 0;                                                        // conversion lanes dealt so far
            for (int i = 0, fd = 0; i < spw; ++fd) {
                const int first = fd ? 0 : sA, count = std::min(spw - i, strips - first), lastS = first + count - 1;
                // (a piece whose run ends near the end of a row starts early enough for its last lane to stay inside the row)
                const int top = (avail - pxl) & ~3;
                int lo = std::min(slo[first], top);
                // (8 or 16 pixels a lane: pieces start so that every lane, the slid last one included, lies a multiple of 8 pixels
                // from the piece's first -- the kernel then writes its LDS rows as whole vectors)
                if (pxl >= 8 && ((top - lo) & 7)) { if (lo >= 4) lo -= 4; else if (write) aligned8 = false; }
                int hi = 0;
                for (int q = first; q <= lastS; ++q) hi = std::max(hi, shi[q]);
                const int lanes = (hi - lo + pxl - 1) / pxl;
                if (used + lanes > 64 || fd > 255) return false;
                if (write) {
                    for (int q = 0; q < lanes; ++q) {
                        // a lane whose pixels would pass the end of the row is slid back over its neighbour's (same values, written twice)
                        const int px = std::min(lo + pxl * q, lo + ((top - lo) & ~3));
                        W[4 * (used + q)] = px;
                        W[4 * (used + q) + 1] = int32_t(uint32_t(pxl * used + (px - lo)) << 8 | uint32_t(fd));
                    }
                    for (int q = 0; q < count; ++q) { W[4 * (i + q) + 2] = fd; W[4 * (i + q) + 3] = pxl * used - lo; }
                }
                used += lanes; i += count;
            }
        }
        return true;
    };
    int best_pxl = 0, best_spw = 0;
    double best_cost = 1e30;
    for (int pxl : {4, 8, 16}) {
        if ((nl == 4 && pxl == 16) || (nl == 2 && pxl == 4) || (pb == 4 && pxl == 4)) continue;   // (the instantiations on offer: 4 or 8 pixels a lane with 4 columns -- 8 of four bytes --, 8 or 16 with 2)
        int spw = 0;
        for (int cand = 64; cand >= (nl == 4 ? 16 : 4) && !spw; --cand) if (deal(pxl, cand, false)) spw = cand;
        if (!spw) continue;
        // instructions a source row: conversion + the rest, per strip.  (Two columns a lane: 16 pixels a lane measures about
        // 1.25x its instruction count against 8 -- one wave a SIMD fewer, and b128 LDS writes 32 bytes apart; before those
        // vector writes it was 1.6x: profiles/r03/q_rgbin_many_tap_8_vs_16_pixels_a_lane.txt, q_rgbin_lds_vector_writes.txt)
        const double cost = (11.0 * pxl + 50.0 + 24.0 * tph) / spw * (nl == 2 && pxl == 16 ? 1.25 : 1.0);
        if (cost < best_cost) { best_cost = cost; best_pxl = pxl; best_spw = spw; }
    }
    if ((nl == 4 && pb == 3 && (env("TTS_RGBIN_SHRINK_PXL") == 4 || env("TTS_RGBIN_SHRINK_PXL") == 8)) ||
        (nl == 2 && (env("TTS_RGBIN_SHRINK_PXL") == 8 || env("TTS_RGBIN_SHRINK_PXL") == 16))) {
        best_pxl = env("TTS_RGBIN_SHRINK_PXL"); best_spw = 0;
        for (int cand = 64; cand >= (nl == 4 ? 16 : 4) && !best_spw; --cand) if (deal(best_pxl, cand, false)) best_spw = cand;
    }
    bool ok = best_pxl != 0 && best_spw != 0 && deal(best_pxl, best_spw, true);
    a.w128 = best_pxl >= 8 && aligned8 && !env("TTS_RGBIN_SHRINK_NO_W128") ? 1 : 0;
    // per source row: the output rows whose filters hold it (a run of rows: positions do not decrease), its tap in each,
    // how many of them it completes, and where a segment that starts at this row has to begin
    std::vector<int> pl(static_cast<size_t>(oh), 0), ll(static_cast<size_t>(oh), 0), pc(static_cast<size_t>(och), 0), lcv(static_cast<size_t>(och), 0);
    for (int y = 0; y < oh; ++y) { pl[y] = lv.pos[oy + y]; ll[y] = pl[y] + lvt - 1; }
    for (int y = 0; y < och; ++y) { pc[y] = cv.pos[oyc + y]; lcv[y] = pc[y] + cvt - 1; }
    const int h0 = std::min(pl[0], pc[0]), h1 = std::max(ll[oh - 1], lcv[och - 1]) + 1;
    a.o_rows = uint32_t(t.size());
    t.resize(t.size() + size_t(kRowEntry) * (h1 - h0), 0);
    {
        int yl = 0, yc = 0, most = 0;
        for (int r = h0; r < h1 && ok; ++r) {
            int32_t *E = t.data() + a.o_rows + size_t(kRowEntry) * (r - h0);
            while (yl < oh && ll[yl] < r) ++yl;
            while (yc < och && lcv[yc] < r) ++yc;
            int npl = 0, npc = 0, ncl = 0, ncc = 0;
            for (int y = yl; y < oh && pl[y] <= r; ++y, ++npl) {
                if (npl >= kPend) { ok = false; break; }
                E[4 + npl] = lv.taps[size_t(oy + y) * lvt + (r - pl[y])];
                ncl += ll[y] == r;
            }
            for (int y = yc; y < och && pc[y] <= r && ok; ++y, ++npc) {
                if (npc >= kPend) { ok = false; break; }
                E[4 + kPend + npc] = cv.taps[size_t(oyc + y) * cvt + (r - pc[y])];
                ncc += lcv[y] == r;
            }
            most = std::max(most, std::max(npl, npc));
            E[0] = yl; E[1] = yc;
            E[2] = ncl | ncc << 8 | npl << 16 | npc << 24;
            E[3] = std::max(h0, std::min(yl < oh ? pl[yl] : r, yc < och ? pc[yc] : r));
            if (ncl > 255 || ncc > 255) ok = false;
        }
        p->kp = most <= 6 && !env("TTS_RGBIN_SHRINK_KP8") ? 6 : 8;
    }
    if (ok) {
        const size_t bytes = std::max<size_t>(t.size(), 1) * sizeof(int32_t);
        ok = hipMalloc(reinterpret_cast<void **>(&p->d_tab), bytes) == hipSuccess && hipMemcpy(p->d_tab, t.data(), t.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { rgbin_shrink_destroy(p); return nullptr; }
    a.strips = strips; a.spw = best_spw; a.h0 = h0; a.h1 = h1;
    p->pxl = best_pxl;
    // rgb24ToY_c in the kernel's exact float form: coefficients x 2^-9, constant (2 bias + 1) / 1024; rgb24ToUV_half_c:
    // coefficients x 2^-10, constant 2^-11 (the rest of its bias, 8192, rides in the rounding constant)
    for (int k = 0; k < 3; ++k) { a.cy[k] = float(coef[k]) / 512.0f; a.cu[k] = float(coef[3 + k]) / 1024.0f; a.cv[k] = float(coef[6 + k]) / 1024.0f; }
    a.cy[3] = float(2 * ((32 << 14) + (1 << 8)) + 1) / 1024.0f;
    a.cu[3] = a.cv[3] = 1.0f / 2048.0f;
    a.range15 = g.sc_range == TTS_RANGE_TO_FULL ? 1 : 0;
    p->seg_knob = env("TTS_RGBIN_SEG");
    p->vt_max = std::max(lvt, cvt);
    std::snprintf(p->name, sizeof p->name, "rgbin_shrink<%d,%d,%d,%d,%dpx%s,%d strips a wave,%d sums>%s%s", lht, lvt, cht, cvt, best_pxl, pb == 4 ? " of 4 bytes" : "", best_spw, p->kp, a.range15 ? "+range15" : "", p->pad ? "+pad" : "");
    return p;
}

void rgbin_shrink_destroy(RgbInShrinkPlan *p)
{
    if (!p) return;
    (void)hipFree(p->d_tab);
    delete p;
}

const char *rgbin_shrink_name(const RgbInShrinkPlan *p) { return p ? p->name : ""; }

int rgbin_shrink_launch(const RgbInShrinkPlan *p, const uint8_t *src, uint8_t *dst, int n, int src_stride, const int dst_stride[2],
                        const FrameAddr &addr, hipStream_t st)
{
    const tts_geom &g = p->g;
    RgbInShrinkArgs w = p->args;
    w.total = n * w.strips; w.nframes = n;
    w.src_stride = src_stride; w.dst_stride[0] = dst_stride[0]; w.dst_stride[1] = dst_stride[1];
    w.src_origin = uint32_t(g.pre_y) * uint32_t(src_stride) + uint32_t(g.pre_x) * uint32_t(p->pb);
    w.addr = addr;
    if (p->pad) {
        // the border first (the picture's rectangle is left alone: the walk below writes all of it), then the picture at its offset
        RgbInPadFillArgs f{};
        f.dst = dst; f.addr = addr;
        const int cw = g.crop_w & ~1;
        for (int k = 0; k < 2; ++k) {
            f.w[k] = (g.out_w + k) >> k; f.h[k] = (g.out_h + k) >> k; f.stride[k] = dst_stride[k];
            f.px[k] = g.pad_x >> k; f.py[k] = g.pad_y >> k; f.pw[k] = (cw + k) >> k; f.ph[k] = (g.crop_h + k) >> k;
        }
        f.fill[0] = g.fill_y; f.fill[1] = g.fill_u; f.fill[2] = g.fill_v;
        for (int f0 = 0; f0 < n; f0 += 21845) {
            f.frame0 = f0;
            if (const int rc = launch_shrink("rgbin_pad_fill_kernel", rgbin_pad_fill_kernel, dim3(unsigned((g.out_w + 255) / 256), unsigned((g.out_h + 3) / 4), unsigned(std::min(21845, n - f0) * 3)),
                                             dim3(kBlock), st, f)) return rc;
        }
        w.dst_origin[0] = uint32_t(g.pad_y) * uint32_t(dst_stride[0]) + uint32_t(g.pad_x);
        w.dst_origin[1] = uint32_t(g.pad_y >> 1) * uint32_t(dst_stride[1]) + uint32_t(g.pad_x >> 1);
    }
    // row segments: enough waves for about four rounds of the chip's resident waves (3 a SIMD), none shorter than 48 source
    // rows (each starts as many rows early as the chroma filter is tall)
    const int64_t waves = (int64_t(w.total) + w.spw - 1) / w.spw;
    const int rows = w.h1 - w.h0;
    // (... nor than four times the tallest vertical filter: a segment starts that many rows early)
    const int min_rows = std::max(48, 4 * p->vt_max);
    int nseg = int(std::max<int64_t>(1, std::min<int64_t>((int64_t(4) * 3072 + waves - 1) / waves, std::max(1, rows / min_rows))));
    if (p->nl == 2) {
        // Many taps: a segment starts a whole vertical filter early (49 rows for the reference's thumbnails), so short segments
        // cost real work, and the launch is a round or two of the chip's resident waves (2 to 4 a SIMD by the instantiation's
        // registers), so a last round that is mostly empty costs as much as a full one.  Take the count with the least
        // rounds x rows a wave (segments no shorter than the tallest filter).
        const double wps = double(p->pxl <= 8 ? (p->tph > 13 ? 3 : 4) : (p->tph > 19 ? 2 : 3));
        const int64_t slots = int64_t(wps) * 4 * 256;
        double best = 1e30;
        for (int cand = 1; cand <= std::max(1, rows / std::max(16, p->vt_max)); ++cand) {
            const int sr = (rows + cand - 1) / cand;
            const double cost = double((waves * cand + slots - 1) / slots) * double(sr + p->vt_max);
            if (cost < best) { best = cost; nseg = cand; }
        }
        // A launch that leaves most SIMDs with no wave at all -- one thumbnail a call, the reference's use: what counts is how
        // few rows a wave walks, and a wave alone on its SIMD walks a row in half the time one of three does (measured on R2
        // from 1080p RGB frames: 1.3 us a row at 0.4 waves a SIMD, 1.6 at 1.2, 1.9 at 1.5, 2.5 at 2.9: about 1.1 + 0.5 k).
        // Segments down to 8 rows: 122 -> 72 us a call.
        if (waves * nseg < 4 * 256) {
            best = 1e30;
            for (int cand = 1; cand <= std::max(1, rows / 8); ++cand) {
                const int sr = (rows + cand - 1) / cand;
                const double k = double(waves) * cand / (4.0 * 256.0);
                const double cost = double(sr + p->vt_max) * (1.1 + 0.5 * std::min(k, wps)) * std::max(1.0, k / wps);
                if (cost < best) { best = cost; nseg = cand; }
            }
        }
    }
    if (p->seg_knob > 0) nseg = std::max(1, std::min(p->seg_knob, rows / 8));
    w.seg_rows = (rows + nseg - 1) / nseg;
    w.nseg = (rows + w.seg_rows - 1) / w.seg_rows;
    if (p->nl == 2) {
#define TTS_SHRINK2P(PXL_, TPH_) (p->pb == 4 ? (p->kp == 6 ? rgbin_shrink_kernel<PXL_, TPH_, 6, 2, 4> : rgbin_shrink_kernel<PXL_, TPH_, 8, 2, 4>) \
                                             : (p->kp == 6 ? rgbin_shrink_kernel<PXL_, TPH_, 6, 2> : rgbin_shrink_kernel<PXL_, TPH_, 8, 2>))
#define TTS_SHRINK2(TPH_) (p->pxl == 8 ? TTS_SHRINK2P(8, TPH_) : TTS_SHRINK2P(16, TPH_))
        auto fn2 = p->tph == 7 ? TTS_SHRINK2(7) : p->tph == 10 ? TTS_SHRINK2(10) : p->tph == 13 ? TTS_SHRINK2(13) : p->tph == 19 ? TTS_SHRINK2(19) : TTS_SHRINK2(25);
#undef TTS_SHRINK2
#undef TTS_SHRINK2P
        return launch_shrink("rgbin_shrink_kernel", fn2, dim3(unsigned((waves + kBlock / 64 - 1) / (kBlock / 64)), unsigned(w.nseg), 1), dim3(kBlock), st,
                             src, dst, static_cast<const int32_t *>(p->d_tab), w);
    }
    if (p->pb == 4) {
        auto fn4 = p->kp == 6 ? (p->tph == 3 ? rgbin_shrink_kernel<8, 3, 6, 4, 4> : rgbin_shrink_kernel<8, 4, 6, 4, 4>) : (p->tph == 3 ? rgbin_shrink_kernel<8, 3, 8, 4, 4> : rgbin_shrink_kernel<8, 4, 8, 4, 4>);
        return launch_shrink("rgbin_shrink_kernel", fn4, dim3(unsigned((waves + kBlock / 64 - 1) / (kBlock / 64)), unsigned(w.nseg), 1), dim3(kBlock), st,
                             src, dst, static_cast<const int32_t *>(p->d_tab), w);
    }
    auto fn = p->kp == 6 ? (p->pxl == 4 ? (p->tph == 3 ? rgbin_shrink_kernel<4, 3, 6> : rgbin_shrink_kernel<4, 4, 6>) : (p->tph == 3 ? rgbin_shrink_kernel<8, 3, 6> : rgbin_shrink_kernel<8, 4, 6>))
                         : (p->pxl == 4 ? (p->tph == 3 ? rgbin_shrink_kernel<4, 3, 8> : rgbin_shrink_kernel<4, 4, 8>) : (p->tph == 3 ? rgbin_shrink_kernel<8, 3, 8> : rgbin_shrink_kernel<8, 4, 8>));
    return launch_shrink("rgbin_shrink_kernel", fn, dim3(unsigned((waves + kBlock / 64 - 1) / (kBlock / 64)), unsigned(w.nseg), 1), dim3(kBlock), st,
                         src, dst, static_cast<const int32_t *>(p->d_tab), w);
}

}  // namespace tts
