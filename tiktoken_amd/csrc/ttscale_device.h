// ttscale_device.h -- what the translation units of libttscale share on the device side: launch-wide
// constants, the frame address table that rides in the kernel arguments, and the small gfx950 helpers
// (packed dot products, wave-uniform reads, the XCD remap, output stores issued from inline asm).
// Included by ttscale_kernels.h (the planar scaler, ttscale.hip) and by ttscale_rgb.hip (packed RGB).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace tts {

constexpr int kBlock = 256;
constexpr int kMaxLdsBlock = 1024;    // largest workgroup of the LDS-staged kernel
constexpr size_t kMaxSpan = 0xFF000000u;   // bytes one launch may address from its base pointers
constexpr uint32_t kDiscard = 0xFFFFFF00u; // a lane offset the buffer range check always rejects (> kMaxSpan)
constexpr int kMaxTableFrames = 120;  // frames addressed through the in-kernarg offset table (kernel arguments stay under 4 KB)

// Where the planes of frame f live, relative to the base pointers of the launch.  (Named namespace: the
// struct crosses from ttscale.hip to ttscale_rgb.hip.)  A packed-RGB destination has one plane, [0].
struct FrameAddr {
    uint32_t src_pitch, dst_pitch;            // packed mode: bytes between frames
    uint32_t src_plane[3], dst_plane[3];      // packed mode: plane origin inside a frame
    int32_t  use_table;                        // table mode: explicit per-frame offsets
    int32_t  wave_base;                        // packed mode: spans pass kMaxSpan, waves address from their own first frame
    uint32_t src_tab[3][kMaxTableFrames];
    uint32_t dst_tab[3][kMaxTableFrames];
};

}  // namespace tts

namespace {

using tts::FrameAddr;
using tts::kBlock;
using tts::kMaxLdsBlock;
using tts::kMaxSpan;
using tts::kDiscard;
using tts::kMaxTableFrames;

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// clamp = true selects the VOP3P form (v_dot2_i32_i16 d, a, b, c) whose accumulator input is a
// free operand; the sums here never reach the int32 range, so the clamp never acts.
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), c, true);
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of
// logical blocks so neighbouring strips and row segments share one L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// 15-bit intermediate of swscale's horizontal pass: truncating >>7, clip to 32767
__device__ __forceinline__ uint32_t hfinish(int acc)
{
    acc >>= 7;
    return uint32_t(acc < 32767 ? acc : 32767);
}

// two int16 -> two clipped bytes in bits 0..15 (bits 16..31 zero).  Inline asm on purpose: hipcc matches
// "shift right, clamp to 0..255, pack" written in C++ to gfx950's v_ashr_pk_u8_i32, whose result carried
// non-zero bits 16..31 on MI355X (seen as bytes 2 and 3 of a packed dword OR-ed with garbage in the range
// epilogue's first version); an asm statement is opaque to that pattern.
__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t packed_i16)
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(packed_i16));
    return r;
}

// four vertical sums -> four clipped bytes (val >> 19, clip to 0..255) in two instructions: gfx950's
// v_ashr_pk_u8_i32 shifts two int32, clips both to 0..255 and writes them as two bytes into ONE HALF of the
// destination -- the low half, or with op_sel:[0,0,0,1] the high half -- leaving the other half as it was (that
// is the "garbage" of the note above).  tools/micro/ashr_pk.hip checks both forms against the host over edge
// values.  (Was: two v_perm, two packed shifts, two v_sat_pk_u8_i16 and an or.)  The s_nop is the wait state a
// VALU result written through op_sel's high half needs before the next VALU reads it (gfx940-family forwarding
// hazard; the compiler inserts it for its own instructions and cannot see into this one).
template <int SH = 19>
__device__ __forceinline__ uint32_t pack4(int v0, int v1, int v2, int v3)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(d) : "v"(v0), "v"(v1), "n"(SH));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]\n\ts_nop 0" : "+v"(d) : "v"(v2), "v"(v3), "n"(SH));
    return d;
}
// two sums -> two clipped bytes in bits 0..15; bits 16..31 are whatever the register held
__device__ __forceinline__ uint32_t pack2(int v0, int v1)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 19" : "=v"(d) : "v"(v0), "v"(v1));
    return d;
}
__device__ __forceinline__ uint32_t pack2_by(int v0, int v1, int sh)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(d) : "v"(v0), "v"(v1), "s"(sh));
    return d;
}
// the same with the shift in a register (the range step's)
__device__ __forceinline__ uint32_t pack4_by(int v0, int v1, int v2, int v3, int sh)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(d) : "v"(v0), "v"(v1), "s"(sh));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]\n\ts_nop 0" : "+v"(d) : "v"(v2), "v"(v3), "s"(sh));
    return d;
}

// Output stores are issued from inline asm.  hipcc's s_waitcnt pass counts the vector-memory
// operations it can see; with stores visible inside the variable-trip emission loop it can no
// longer count and falls back to vmcnt(0) before every use of a prefetched row, which drains the
// rows still in flight AND every store acknowledgement once per source row.  Stores it cannot see
// leave it counting the row loads alone, vmcnt((D-1)*loads), which (the counter retires in issue
// order) is never too weak: the hidden stores only make the wait retire a few more operations.
__device__ __forceinline__ void store_b128(u32x4_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
__device__ __forceinline__ void store_b64(u32x2_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
__device__ __forceinline__ void store_b8(uint32_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_byte %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
[[maybe_unused]] __device__ __forceinline__ void store_b16(uint32_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_short %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}
[[maybe_unused]] __device__ __forceinline__ void store_b32(uint32_t v, u32x4_t rs, uint32_t off, int soff)
{
    asm volatile("buffer_store_dword %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(rs), "s"(soff));
}

// The same through a 64-bit address (the kernels that walk SOURCE rows with the next rows' loads in flight: with these
// stores hidden and the loads two rows ahead, the compiler's wait for a row is vmcnt(<loads of the row behind it>), which
// leaves the stores just issued in flight; with one row ahead, or with stores it sees inside the variable-trip emission
// loops, every row waited out the acknowledgement of the stores in front of it)
[[maybe_unused]] __device__ __forceinline__ void gstore_b64(void *p, u32x2_t v) { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v)); }
[[maybe_unused]] __device__ __forceinline__ void gstore_b32(void *p, uint32_t v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v)); }

}  // namespace
