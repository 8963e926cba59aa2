// ttscale_plan.h -- host side of libttscale: per-geometry tables, kernel choice, launches.
// Included by ttscale.hip only, after ttscale_kernels.h.
#pragma once
// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------

namespace {

constexpr int kSegOptions[] = {16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160, 192, 256};
constexpr int kNumSegOptions = int(sizeof(kSegOptions) / sizeof(kSegOptions[0]));

struct SegTables {
    int seg_rows = 0, nseg = 0, maxrows = 0;
    uint32_t seginfo = 0, emit = 0, vcoef = 0;     // dword offsets into the plan's table buffer
    int total_src_rows = 0;                        // sum over segments (halo cost model)
    bool spans_plane = false;                      // some segment needs both the first and the last source row
};

// Horizontal tables of one plane class for one strip width (dword offsets into the table buffer).
struct HTables {
    int strips = 0;
    uint32_t hoff = 0, hcoef = 0;                  // one window per column
    bool g4 = false;                               // grouped 8-byte windows usable (taps <= 4, 4 columns span <= 8 bytes)
    uint32_t hoff4 = 0, hsel4 = 0, hcoef4 = 0;
    int lds_g4 = 0, lds_col = 0;                   // LDS-staged kernel: 1 KiB DMAs per row it needs with grouped /
                                                   // per-column windows (0 = not usable)
    uint32_t hoff6 = 0, hcoef6 = 0;                // 5-6 taps: 6-byte windows for the LDS-staged kernel (3 tap pairs a column)
    // grouped 12-byte windows (LDS-staged kernel only): 5-6 taps with 4 or 2 columns a window, 7-8 taps with 2
    struct Grouped12 { int lw = 0, gc = 0; uint32_t hoff = 0, hsel = 0, hcoef = 0; int lds = 0; };
    Grouped12 g12;                                 // the variant this class takes (lds == 0: none)
};

struct ClassPlan {
    tts::AxisFilter h, v;
    int sub = 0;
    int src_w = 0, src_h = 0;          // (pre-cropped) source plane
    int x0 = 0, y0 = 0, ow = 0, oh = 0; // region of the virtual scaled plane that is produced
    int canvas_w = 0, canvas_h = 0;    // output plane
    int pad_x = 0, pad_y = 0;
    int kind = 0;                      // 0 generic, 1 strip, 2 column
    int LW = 0, R = 0;
    int LWlds = 0;                     // window width of the LDS-staged kernel's per-column mode (6 when 5-6 taps)
    // kind 2: column walk (9..64 taps): one output column per lane
    int NTP = 0, Rc = 0, col_dmas = 0; // tap pairs per column, ring depth, LDS-DMAs per source row
    int col_nc = 1;                    // output columns per lane (1, 2), the most a launch may take
    int col_dmas1 = 0;                 // LDS-DMAs per source row at one column per lane (what small launches take above 8 tap pairs)
    uint32_t hoff1 = 0, hcoef1 = 0;    // window start per column, taps per column [ow][NTP]
    uint32_t d4tab[3] = {0, 0, 0};     // [nc]: digit tables of the v_dot4 form for nc columns per lane (0 = none: see build_d4_tables)
    HTables ht[2];                     // horizontal tables for N = 8 and N = 16 columns per lane
    std::vector<SegTables> segs;
    // generic tables
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

// One kernel launch whose OWN status is returned: hipLaunchKernel reports the failure of this launch, so a
// stale error left by the host application's earlier HIP calls is neither blamed on this library nor
// cleared from under the application (hipGetLastError would do both).
template <class... P, class... A>
int launch_checked(const char *what, void (*fn)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A &&...args)
{
    std::tuple<P...> held(std::forward<A>(args)...);
    void *ptrs[sizeof...(P)];
    size_t i = 0;
    std::apply([&](auto &...v) { ((ptrs[i++] = const_cast<void *>(static_cast<const void *>(&v))), ...); }, held);
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(fn), grid, block, ptrs, lds, st);
    if (e == hipSuccess) return TTS_OK;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return TTS_ERR_PROCESSING_FAILED;
}

}  // namespace

// Tuning knobs (environment), read ONCE when a plan is created so that tts_run* never calls getenv.
// Defaults are the product behaviour; none of them changes a pixel (tests/test_gpu_parity.py checks
// that); tools/sweep.py sets them per plan.  The TTS_EXP_* timing experiments of DESIGN.md 4.5 DO
// produce wrong output (they drop or alias traffic to see what a launch costs without it) and exist
// only in builds made with -DTTS_EXPERIMENTS.
struct Knobs {
    int strip_n = 8, strip_d = 0, paced = -1, lds_block = 0, no_merge = 0;
    int exp_nosrc = 0, exp_nodst = 0, exp_srcpitch0 = 0, exp_dstpitch0 = 0, exp_stride0 = 0;
    int planewise = 0;                // tts_run: always take the plane-by-plane path (test knob)
    int span_chunks = 0;              // tts_run_batch: always cut batches into launches whose whole span fits 32 bits (test knob)
    int wave_base = 0;                // tts_run_batch: per-wave descriptors even when the batch fits one base (test knob)
    int col_nc_launch = 0;            // column walk above 8 tap pairs: 1 / 2 columns per lane whatever the launch's size (0 = by size)
    int col_fill = 0;                 // column walk: workgroups per resident slot (in tenths) a segment length must leave; 0 = by tap count
    int no_rgb_scratch = 0;           // packed-RGB thumbnails through the generic kernel instead of column walk + in-place conversion (A/B, tests)
    int col_d4 = 1;                   // column walk: v_dot4 on packed bytes with two-digit taps instead of v_perm + v_dot2: 0 never, 1 above 8 tap pairs, 2 wherever the tables exist
    int col_model = 1;                // column walk segment length: 0 = fill rule, 1 = busiest-CU model above 8 tap pairs, 2 = everywhere
    int host_chunk_mb = 128;          // tts_run_host: bytes (source + output) of one pipelined chunk
    int no_union = 0;                 // strip_lds_kernel (two columns a window): never the one-run read of a lane's four windows (A/B)
    int persist = 0;                  // strip_lds_kernel as a work queue of resident workgroups (experiment; tenths of the resident slots to launch, 10 = all)
    void read()
    {
        strip_n = env_int("TTS_STRIP_N", 8); strip_d = env_int("TTS_STRIP_D", 0);
        paced = env_int("TTS_PACED", -1); lds_block = env_int("TTS_LDS_BLOCK", 0); no_merge = env_int("TTS_NO_MERGE", 0);
#ifdef TTS_EXPERIMENTS
        exp_nosrc = env_int("TTS_EXP_NOSRC", 0); exp_nodst = env_int("TTS_EXP_NODST", 0);
        exp_srcpitch0 = env_int("TTS_EXP_SRCPITCH0", 0); exp_dstpitch0 = env_int("TTS_EXP_DSTPITCH0", 0);
        exp_stride0 = env_int("TTS_EXP_STRIDE0", 0);
#endif
        no_union = env_int("TTS_NO_UNION", 0);
        planewise = env_int("TTS_PLANEWISE", 0);
        span_chunks = env_int("TTS_SPAN_CHUNKS", 0);
        wave_base = env_int("TTS_WAVE_BASE", 0);
        col_fill = std::max(0, env_int("TTS_COL_FILL", 0));
        col_model = env_int("TTS_COL_MODEL", 1);
        col_d4 = env_int("TTS_COL_D4", 1);
        no_rgb_scratch = env_int("TTS_NO_RGB_SCRATCH", 0);
        col_nc_launch = env_int("TTS_COL_NC_LAUNCH", 0);
        host_chunk_mb = std::max(1, env_int("TTS_HOST_CHUNK_MB", 128));
        persist = std::max(0, env_int("TTS_PERSIST", 0));
    }
};

#ifdef TTS_EXPERIMENTS
// workgroup timeline of the last strip launch (tools/timeline.py): 2 clocks per workgroup, 100 MHz
static uint64_t *g_exp_dbg = nullptr;
static uint32_t g_exp_dbg_blocks = 0;
static uint64_t *exp_dbg_buffer(uint32_t blocks)
{
    constexpr uint32_t kCap = 1u << 20;
    if (!std::getenv("TTS_EXP_TIMELINE") || blocks > kCap) { g_exp_dbg_blocks = 0; return nullptr; }
    if (!g_exp_dbg && hipMalloc(reinterpret_cast<void **>(&g_exp_dbg), size_t(kCap) * 16) != hipSuccess) return nullptr;
    g_exp_dbg_blocks = blocks;
    return g_exp_dbg;
}
extern "C" int tts_exp_timeline(uint64_t *out, uint32_t cap)
{
    if (!g_exp_dbg || !g_exp_dbg_blocks) return 0;
    const uint32_t n = std::min(cap, g_exp_dbg_blocks);
    if (hipMemcpy(out, g_exp_dbg, size_t(n) * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return int(n);
}
#endif

struct tts_plan {
    tts_geom g;
    Knobs knobs;
    ClassPlan cls[2];
    uint32_t *d_tab = nullptr;        // strip tables of both classes
    uint32_t *d_queue = nullptr;      // TTS_PERSIST: per-XCD block counters of a work-queue launch (experiment: launches of one plan then take turns)
    tts::RgbPlan *rgb = nullptr;      // packed-RGB output (g.out_fmt != yuv420p): the plan of ttscale_rgb.hip; cls[] stay empty
    tts::RgbInPlan *rgbin = nullptr;  // packed rgb24 input (g.in_fmt): likewise
    // Packed-RGB thumbnails (many-tap shrinks written as RGB by the chain's scale): a planar plan of the same
    // scale whose column walk leaves unclipped 16-bit luma and chroma at ceil(w / 2) x h in the destination frame
    // itself, converted in place by rgb_scratch_convert_kernel (tts_run_batch on packed rows; `rgb` stays the
    // path of everything else).  Null when the geometry does not qualify.
    tts_plan *scratch = nullptr;
    bool y16 = false;                 // this IS such a scratch plan: luma class stores int16
    int device = 0;
    int n_cus = 256;
};

namespace {

bool monotone(const std::vector<int32_t> &p)
{
    for (size_t i = 1; i < p.size(); ++i) if (p[i] < p[i - 1]) return false;
    return true;
}

inline uint32_t pack16(int16_t lo, int16_t hi) { return uint32_t(uint16_t(lo)) | (uint32_t(uint16_t(hi)) << 16); }

void build_seg_tables(const ClassPlan &c, int R, int seg_rows, std::vector<uint32_t> &tab, SegTables &st)
{
    const int VT = c.v.size;
    st.seg_rows = seg_rows;
    st.nseg = (c.oh + seg_rows - 1) / seg_rows;
    st.maxrows = 0; st.total_src_rows = 0;
    std::vector<int> first(st.nseg), rows(st.nseg);
    for (int s = 0; s < st.nseg; ++s) {
        const int y0 = s * seg_rows, y1 = std::min(c.oh, y0 + seg_rows);
        first[s] = c.v.pos[c.y0 + y0];
        rows[s] = c.v.pos[c.y0 + y1 - 1] + VT - first[s];
        st.maxrows = std::max(st.maxrows, rows[s]);
        st.total_src_rows += rows[s];
        st.spans_plane |= first[s] == 0 && first[s] + rows[s] >= c.src_h;
    }
    st.seginfo = uint32_t(tab.size());
    for (int s = 0; s < st.nseg; ++s) { tab.push_back(uint32_t(first[s])); tab.push_back(uint32_t(rows[s])); }
    st.emit = uint32_t(tab.size());
    tab.resize(tab.size() + size_t(st.nseg) * st.maxrows, 0);
    st.vcoef = uint32_t(tab.size());
    tab.resize(tab.size() + size_t(c.oh) * (R / 2), 0);
    for (int y = 0; y < c.oh; ++y) {
        const int s = y / seg_rows, vy = c.y0 + y;
        const int p = c.v.pos[vy];
        tab[st.emit + size_t(s) * st.maxrows + (p + VT - 1 - first[s])] += 1;
        int16_t slot[64] = {0};
        for (int j = 0; j < VT; ++j) slot[(p + j - first[s]) % R] = c.v.taps[size_t(vy) * VT + j];
        for (int k = 0; k < R / 2; ++k) tab[st.vcoef + size_t(y) * (R / 2) + k] = pack16(slot[2 * k], slot[2 * k + 1]);
    }
}

// LDS-staged kernel: how many 1 KiB LDS-DMAs per source row does a wave need?  Any 64 consecutive
// strips of the flattened (plane, strip) order -- they wrap over plane rows -- need one run of source
// bytes per row touched, each rounded out to 16-byte chunks (worst case over the run's alignment);
// the runs must fit 64 chunks per DMA and kMaxPieces runs.  `lo` holds the first byte of each of the
// `nw` windows of a strip, `win` their width.  0 = not usable.
int lds_dmas(const ClassPlan &c, const HTables &t, int N, const int *lo, int nw, int win, bool four = false)
{
    if ((N != 8 && N != 4 && N != 2 && N != 1) || c.ow < N || t.strips < 2 || env_int("TTS_NO_LDS", 0)) return 0;
    auto chunks = [&](int s0, int s1) {   // strips [s0, s1] of one row
        const uint32_t b = uint32_t(lo[size_t(s0) * nw]), e = uint32_t(lo[size_t(s1) * nw + nw - 1] + win);
        return (e - b + 30u) >> 4;
    };
    uint32_t worst = 0;
    for (int s0 = 0; s0 < t.strips; ++s0) {
        uint32_t n = 0;
        int pieces = 0, s = s0, left = 64;
        while (left > 0) {
            const int take = std::min(left, t.strips - s);
            n += chunks(s, s + take - 1);
            ++pieces; left -= take; s = 0;
        }
        if (pieces > kMaxPieces) return 0;
        worst = std::max(worst, n);
    }
    // (four DMAs a row: the per-column strip kernels only -- 4x..7x `area` thumbnails, whose 64 strips read 2-4 KB a row)
    return worst <= 64u ? 1 : worst <= 128u ? 2 : (four && worst <= 256u) ? 4 : 0;
}

// Digit tables of column_lds_kernel's v_dot4 form, one entry per (strip, byte phase of its span in LDS):
//   nc = 1: [bias][th x NDW][tl x NDW]      nc = 2: [bias0][bias1][th0 x NDW][tl0 x NDW][th1 x NSP][tl1 x NSP]
// th / tl: the taps' signed-byte digits (t = 128 th + tl, tl in 0..127), byte (phase + k) of the span for tap k of
// column 0's window, (phase + delta + k) for column 1's, delta = the distance of the two windows; bias = 128 x
// the column's tap sum (the pixels enter as p - 128).  0 if some tap has no such digits or some delta is too wide.
uint32_t build_d4_tables(const ClassPlan &c, int NTP, int nc, const std::vector<int> &lo, const std::vector<int16_t> &slots,
                         std::vector<uint32_t> &tab)
{
    const int NDW = NTP / 2 + 1, NSP = nc == 2 ? NDW + 2 : NDW;
    const int esz = nc == 2 ? 2 + 2 * NDW + 2 * NSP : 1 + 2 * NDW;
    const int strips = (c.ow + nc - 1) / nc;
    if (c.ow < nc) return 0;
    std::vector<uint32_t> out(size_t(strips) * 4 * esz, 0);
    for (int s = 0; s < strips; ++s) {
        const int col0 = std::min(s * nc, c.ow - nc);
        for (int o = 0; o < 4; ++o) {
            uint32_t *e = out.data() + (size_t(s) * 4 + o) * esz;
            for (int j = 0; j < nc; ++j) {
                const int col = col0 + j;
                const int shift = o + (lo[col] - lo[col0]);
                const int ndw = j ? NSP : NDW;
                if (shift < 0 || shift + 2 * NTP > 4 * ndw) return 0;
                uint8_t hi[4 * 20] = {0}, lw[4 * 20] = {0};
                int sum = 0;
                for (int k = 0; k < 2 * NTP; ++k) {
                    const int t = slots[size_t(col) * 64 + k];
                    const int l = t & 127, h = (t - l) >> 7;
                    if (h < -128 || h > 127) return 0;
                    hi[shift + k] = uint8_t(int8_t(h)); lw[shift + k] = uint8_t(l);
                    sum += t;
                }
                e[j] = uint32_t(128 * sum);
                uint32_t *th = e + nc + (j ? 2 * NDW : 0), *tl = th + ndw;
                for (int i = 0; i < ndw; ++i) {
                    th[i] = uint32_t(hi[4 * i]) | uint32_t(hi[4 * i + 1]) << 8 | uint32_t(hi[4 * i + 2]) << 16 | uint32_t(hi[4 * i + 3]) << 24;
                    tl[i] = uint32_t(lw[4 * i]) | uint32_t(lw[4 * i + 1]) << 8 | uint32_t(lw[4 * i + 2]) << 16 | uint32_t(lw[4 * i + 3]) << 24;
                }
            }
        }
    }
    const uint32_t at = uint32_t(tab.size());
    tab.insert(tab.end(), out.begin(), out.end());
    return at;
}

// min_ntp: column walk only -- at least this many tap pairs (so that both classes of a chain can run the same
// instantiation, i.e. one launch: see tts_plan_create)
// full_v: the chroma class keeps every OUTPUT row (its source rows are still subsampled): what a swscale context
// that writes packed RGB computes (chroma at ceil(w / 2) x h) -- the scratch planes of the RGB thumbnail path.
// force_column: the column walk for few taps too (it is the kernel with the 16-bit luma store that plan needs).
int build_class(const tts_geom &g, int sub, ClassPlan &c, std::vector<uint32_t> &tab, int min_ntp = 0, bool full_v = false, bool force_column = false)
{
    c.sub = sub;
    const int vsub = full_v ? 0 : sub;                 // subsampling of the output rows
    c.src_w = ceil_shift(g.pre_w, sub); c.src_h = ceil_shift(g.pre_h, sub);
    const int vw = ceil_shift(g.sc_w, sub), vh = ceil_shift(g.sc_h, vsub);
    if (!tts::build_axis_filter(c.src_w, vw, 1 << 14, g.alg, c.h)) return TTS_ERR_INVALID_INPUT;
    if (!tts::build_axis_filter(c.src_h, vh, 1 << 12, g.alg, c.v)) return TTS_ERR_INVALID_INPUT;
    const int content_w = g.do_pad ? (g.crop_w & ~1) : g.crop_w;   // pad's right border eats an odd last column
    c.x0 = g.crop_x >> sub; c.y0 = g.crop_y >> vsub;
    c.ow = ceil_shift(content_w, sub); c.oh = ceil_shift(g.crop_h, vsub);
    c.canvas_w = ceil_shift(g.out_w, sub); c.canvas_h = ceil_shift(g.out_h, vsub);
    c.pad_x = g.do_pad ? g.pad_x >> sub : 0; c.pad_y = g.do_pad ? g.pad_y >> vsub : 0;
    if (c.ow < 1 || c.oh < 1) return TTS_ERR_INVALID_INPUT;

    // generic tables (also what tts_plan_filter reports)
    if (!upload(c.d_hpos, c.h.pos) || !upload(c.d_htaps, c.h.taps) ||
        !upload(c.d_vpos, c.v.pos) || !upload(c.d_vtaps, c.v.taps)) return TTS_ERR_INITIALIZATION_FAILED;

    const int HT = c.h.size, VT = c.v.size;
    const int LW = HT <= 4 ? 4 : 8;
    const int R = VT <= 4 ? 4 : VT <= 6 ? 6 : 8;
    // (A range step INSIDE the scale's context -- tts_geom.sc_range -- converts the 15-bit rows between the two passes: the
    // grouped-window strip kernels and the column walk have RNG instantiations for it, the general kernel GenericArgs.range15;
    // a class whose strip tables come out ungrouped is rebuilt for the column walk by the caller: force_column)
    const bool strip_ok = !force_column && HT <= 8 && VT <= 8 && c.src_w >= LW && monotone(c.v.pos) &&
                          !env_int("TTS_FORCE_GENERIC", 0);
    if (!strip_ok) {
        // Many taps: the column walk, if the taps fit its windows and ring and the runs fit its slot.
        c.kind = 0;
        // one instantiation per 4 taps of the larger tap count of the two axes: 6, 8, ... 32 tap pairs
        // (windows of 2 NTP bytes, ring of 2 NTP rows) -- the work is one v_perm + v_dot2 per tap pair per
        // source row, so a window cut to fit the taps is that much less arithmetic
        const int mt = std::max(HT, VT);
        const int NTP = std::max(std::max(6, ((mt + 1) / 2 + 1) & ~1), min_ntp), Rc = 2 * NTP;
        if (HT > 64 || VT > 64 || c.src_w < 2 * NTP || !monotone(c.v.pos) || env_int("TTS_FORCE_GENERIC", 0)) return TTS_OK;
        HTables t;                                     // (only `strips` is used by lds_dmas)
        t.strips = c.ow;
        std::vector<int> lo(static_cast<size_t>(c.ow));
        std::vector<int16_t> slots(size_t(c.ow) * 64, 0);       // every column's window as taps per byte
        const uint32_t hoff1 = uint32_t(tab.size());
        tab.resize(tab.size() + c.ow);
        const uint32_t hcoef1 = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(c.ow) * NTP, 0);
        for (int i = 0; i < c.ow; ++i) {
            const int vx = c.x0 + i, pos = c.h.pos[vx];
            lo[i] = std::min(pos, c.src_w - 2 * NTP);           // keep the window inside the row
            int16_t slot[64] = {0};
            for (int j = 0; j < HT; ++j) slot[pos - lo[i] + j] = c.h.taps[size_t(vx) * HT + j];
            tab[hoff1 + i] = uint32_t(lo[i]);
            for (int k = 0; k < NTP; ++k) tab[hcoef1 + size_t(i) * NTP + k] = pack16(slot[2 * k], slot[2 * k + 1]);
            std::copy(slot, slot + 64, slots.begin() + size_t(i) * 64);
        }
        // columns per lane: two up to 12 tap pairs (2160x3840 -> 720x1280: 25 -> 41 % of peak; four measured the
        // same there and 5 % slower on 4K -> 720p, at 110 VGPRs; at 10-12 tap pairs, with four rows in flight,
        // two pay on large launches only and launch_all decides), one above, if the 64-strip runs fit the LDS slot
        int dmas = 0, nc = NTP <= 12 ? 2 : 1;
        const int forced_nc = env_int("TTS_COL_NC", 0);
        if (forced_nc == 1 || forced_nc == 2) nc = std::min(nc, forced_nc);
        for (; nc >= 1 && !dmas; nc = dmas ? nc : nc / 2) {
            t.strips = (c.ow + nc - 1) / nc;
            std::vector<int> lo_nc(size_t(t.strips) * nc);
            for (int s = 0; s < t.strips; ++s)
                for (int j = 0; j < nc; ++j) lo_nc[size_t(s) * nc + j] = lo[std::min(std::min(s * nc, std::max(c.ow - nc, 0)) + j, c.ow - 1)];
            dmas = lds_dmas(c, t, nc, lo_nc.data(), nc, 2 * NTP);
            if (dmas) break;
        }
        if (!dmas) return TTS_OK;
        c.kind = 2; c.NTP = NTP; c.Rc = Rc; c.col_dmas = dmas; c.col_nc = nc; c.hoff1 = hoff1; c.hcoef1 = hcoef1;
        c.col_dmas1 = dmas;
        if (nc == 2) { t.strips = c.ow; c.col_dmas1 = lds_dmas(c, t, 1, lo.data(), 1, 2 * NTP); }
        c.d4tab[1] = build_d4_tables(c, NTP, 1, lo, slots, tab);
        if (nc == 2) c.d4tab[2] = build_d4_tables(c, NTP, 2, lo, slots, tab);
        // Segment lengths on offer: the luma plane cut into 1..32 EQUAL parts (outputs here are small --
        // a ragged last segment would be most of the tail), the same list for both classes, ascending.
        std::vector<int> opts;
        for (int parts : {32, 24, 16, 12, 10, 8, 6, 5, 4, 3, 2, 1}) {
            const int sr = std::max(8, (g.crop_h + parts - 1) / parts);
            if (opts.empty() || sr > opts.back()) opts.push_back(sr);
        }
        const int forced = env_int("TTS_SEG_ROWS", 0);
        if (forced > 0) opts.assign(1, forced);
        for (int sr : opts) {
            SegTables st;
            build_seg_tables(c, Rc, sr, tab, st);
            c.segs.push_back(st);
        }
        return TTS_OK;
    }
    c.kind = 1; c.LW = LW; c.R = R;

    // Horizontal tables, one set per strip width N.  Strip s covers columns [xs, xs + N) with
    // xs = min(s * N, ow - N): the last strip is slid left over its neighbour instead of being cut
    // short (the shared columns are simply produced twice, identically), so every lane stores N
    // full bytes and the kernels need no ragged-edge path unless the plane is narrower than N.
    for (int v = 0; v < 2; ++v) {
        HTables &t = c.ht[v];
        const int N = v ? 16 : 8, NL = N / 4;
        t.strips = (c.ow + N - 1) / N;
        const int ncols = t.strips * N;
        auto column = [&](int i) { return c.x0 + std::min(std::min((i / N) * N, std::max(c.ow - N, 0)) + i % N, c.ow - 1); };
        t.hoff = uint32_t(tab.size());
        tab.resize(tab.size() + ncols);
        t.hcoef = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(ncols) * (LW / 2), 0);
        for (int i = 0; i < ncols; ++i) {
            const int vx = column(i);
            const int pos = c.h.pos[vx];
            const int lo = std::min(pos, c.src_w - LW);           // keep the window inside the row
            int16_t slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < HT; ++j) slot[pos - lo + j] = c.h.taps[size_t(vx) * HT + j];
            tab[t.hoff + i] = uint32_t(lo);
            for (int k = 0; k < LW / 2; ++k) tab[t.hcoef + size_t(i) * (LW / 2) + k] = pack16(slot[2 * k], slot[2 * k + 1]);
        }
        // grouped windows: one 8-byte window per 4 columns, taps picked by per-lane byte selectors
        t.g4 = LW == 4 && c.src_w >= 8 && !env_int("TTS_NO_G4", 0);
        std::vector<int> glo(ncols / 4);
        for (int q = 0; q < ncols / 4 && t.g4; ++q) {
            const int vx0 = column(4 * q), vx3 = column(4 * q + 3);
            glo[q] = std::min(c.h.pos[vx0], c.src_w - 8);
            if (c.h.pos[vx3] + HT - glo[q] > 8 || c.h.pos[vx0] < glo[q]) t.g4 = false;
        }
        if (!t.g4) continue;
        t.hoff4 = uint32_t(tab.size());
        tab.resize(tab.size() + ncols / 4);
        t.hsel4 = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(ncols) * 2);
        t.hcoef4 = uint32_t(tab.size());
        tab.resize(tab.size() + size_t(ncols) * 2, 0);
        for (int i = 0; i < ncols; ++i) {
            const int vx = column(i);
            const uint32_t d = uint32_t(c.h.pos[vx] - glo[i / 4]);          // 0..4: first tap's byte in the window
            int16_t t4[4] = {0, 0, 0, 0};
            for (int j = 0; j < HT; ++j) t4[j] = c.h.taps[size_t(vx) * HT + j];
            if (i % 4 == 0) tab[t.hoff4 + i / 4] = uint32_t(glo[i / 4]);
            tab[t.hsel4 + 2 * size_t(i)]     = 0x0c000c00u | d | ((d + 1) << 16);
            tab[t.hsel4 + 2 * size_t(i) + 1] = 0x0c000c00u | (d + 2) | ((d + 3) << 16);
            tab[t.hcoef4 + 2 * size_t(i)]     = pack16(t4[0], t4[1]);
            tab[t.hcoef4 + 2 * size_t(i) + 1] = pack16(t4[2], t4[3]);
        }
        t.lds_g4 = lds_dmas(c, t, N, glo.data(), NL, 8);
    }
    c.LWlds = LW == 8 && HT <= 6 && c.src_w >= 6 ? 6 : LW;
    for (int v = 0; v < 2; ++v) {
        HTables &t = c.ht[v];
        const int N = v ? 16 : 8;
        const int ncols = t.strips * N;
        std::vector<int> lo(static_cast<size_t>(ncols));
        if (c.LWlds == 6) {
            auto column = [&](int i) { return c.x0 + std::min(std::min((i / N) * N, std::max(c.ow - N, 0)) + i % N, c.ow - 1); };
            t.hoff6 = uint32_t(tab.size());
            tab.resize(tab.size() + ncols);
            t.hcoef6 = uint32_t(tab.size());
            tab.resize(tab.size() + size_t(ncols) * 3, 0);
            for (int i = 0; i < ncols; ++i) {
                const int vx = column(i), pos = c.h.pos[vx];
                lo[i] = std::min(pos, c.src_w - 6);
                int16_t slot[6] = {0, 0, 0, 0, 0, 0};
                for (int j = 0; j < HT; ++j) slot[pos - lo[i] + j] = c.h.taps[size_t(vx) * HT + j];
                tab[t.hoff6 + i] = uint32_t(lo[i]);
                for (int k = 0; k < 3; ++k) tab[t.hcoef6 + size_t(i) * 3 + k] = pack16(slot[2 * k], slot[2 * k + 1]);
            }
        } else {
            for (int i = 0; i < ncols; ++i) lo[i] = int(tab[t.hoff + i]);
        }
        t.lds_col = lds_dmas(c, t, N, lo.data(), N, c.LWlds, !env_int("TTS_NO_DMA4", 0));
        // Grouped 12-byte windows: GC columns share one window, read as four aligned dwords -- 8 (GC = 4) or
        // 16 (GC = 2) LDS dwords per lane and row instead of 24.  Tap j of a column whose first tap is byte d
        // of the window is picked out of the dword pair (j/4, j/4 + 1): d + j - 4 (j/4) must stay <= 7, and
        // d + taps <= 12.  5-6 taps: four columns up to ratios of about 1.5 (4K -> portrait is 1.125), two
        // beyond (the 3.4x area thumbnails); 7-8 taps: two columns (1.5x .. 2x: 4K -> 1080p).
        if (c.src_w >= 12 && HT >= 5 && !env_int("TTS_NO_G4", 0) && !env_int("TTS_NO_G6", 0)) {
            auto column = [&](int i) { return c.x0 + std::min(std::min((i / N) * N, std::max(c.ow - N, 0)) + i % N, c.ow - 1); };
            const int lw = HT <= 6 ? 6 : 8;
            const int forced_gc = env_int("TTS_G12_GC", 0);
            for (int gc : {4, 2}) {
                if ((lw == 8 && gc == 4) || t.g12.lds || (forced_gc && gc != forced_gc)) continue;
                std::vector<int> glo(ncols / gc);
                bool ok = true;
                for (int q = 0; q < ncols / gc && ok; ++q) {
                    glo[q] = c.h.pos[column(gc * q)];   // not pulled back from the row's end: see the kernel's run bounds
                    for (int j = 0; j < gc && ok; ++j) {
                        const int d = c.h.pos[column(gc * q + j)] - glo[q];
                        if (d < 0 || d + HT > 12) ok = false;
                        for (int k = 0; k < HT && ok; ++k)
                            if (d + k - 4 * (k / 4) > 7) ok = false;
                    }
                }
                if (!ok) continue;
                const int lds = lds_dmas(c, t, N, glo.data(), N / gc, 12, gc == 2 && !env_int("TTS_NO_DMA4", 0));
                if (!lds || (gc == 4 && lds != 1)) continue;            // (the 4-column kernels exist for one DMA only)
                HTables::Grouped12 &gr = t.g12;
                gr.lw = lw; gr.gc = gc; gr.lds = lds;
                gr.hoff = uint32_t(tab.size());
                tab.resize(tab.size() + ncols / gc);
                gr.hsel = uint32_t(tab.size());
                tab.resize(tab.size() + size_t(ncols) * 2);
                gr.hcoef = uint32_t(tab.size());
                tab.resize(tab.size() + size_t(ncols) * (lw / 2), 0);
                for (int i = 0; i < ncols; ++i) {
                    const int vx = column(i);
                    const uint32_t d = uint32_t(c.h.pos[vx] - glo[i / gc]);
                    int16_t t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int k = 0; k < HT; ++k) t8[k] = c.h.taps[size_t(vx) * HT + k];
                    if (i % gc == 0) tab[gr.hoff + i / gc] = uint32_t(glo[i / gc]);
                    tab[gr.hsel + 2 * size_t(i)]     = 0x0c000c00u | d | ((d + 1) << 16);
                    tab[gr.hsel + 2 * size_t(i) + 1] = 0x0c000c00u | (d + 2) | ((d + 3) << 16);
                    for (int k = 0; k < lw / 2; ++k) tab[gr.hcoef + size_t(lw / 2) * i + k] = pack16(t8[2 * k], t8[2 * k + 1]);
                }
            }
        }
    }
    std::vector<int> opts(kSegOptions, kSegOptions + kNumSegOptions);
    const int forced = env_int("TTS_SEG_ROWS", 0);
    if (forced > 0) opts.assign(1, forced);
    for (int sr : opts) {
        SegTables st;
        build_seg_tables(c, c.R, sr, tab, st);
        c.segs.push_back(st);
    }
    return TTS_OK;
}

void free_class(ClassPlan &c)
{
    (void)hipFree(c.d_hpos); (void)hipFree(c.d_vpos); (void)hipFree(c.d_htaps); (void)hipFree(c.d_vtaps);
}

typedef void (*strip_fn)(const uint8_t *, uint8_t *, const uint32_t *, const StripArgs);

// lds: LDS-DMAs per row of the LDS-staged kernel, 0 = register kernel; depth: its source rows in flight
struct StripChoice { int LW, R, N; bool g4; int lds; int depth; int g12 = 0; };   // g12: columns per 12-byte window, 0 = none

StripChoice strip_choice(const ClassPlan &c, const Knobs &kn)
{
    StripChoice ch{c.LW, c.R, 8, false, 0, 1};
    if (c.LW == 4 && c.R == 4) ch.N = kn.strip_n == 16 ? 16 : 8;
    const HTables &t = c.ht[ch.N == 16];
    ch.g4 = t.g4;
    ch.lds = t.g4 ? t.lds_g4 : t.lds_col;
    if (t.g4 && ch.lds != 1) { ch.lds = t.lds_col; if (ch.lds) ch.g4 = false; }   // grouped windows ride in one DMA only
    if (ch.lds && !ch.g4) ch.LW = c.LWlds;
    if (!ch.g4 && t.g12.lds) { ch.g4 = true; ch.g12 = t.g12.gc; ch.lds = t.g12.lds; ch.LW = t.g12.lw; }   // (LDS-staged kernel only)
    if (ch.lds && ch.g4 && ch.R == 4) ch.depth = kn.strip_d == 2 ? 2 : 1;
    // The 8-tap 2x downscales (4K -> 1080p) are read-bound at 4 waves a SIMD: four source rows in flight
    // per wave instead of one, +5..10 %.  (Two rows made no robust difference to the 5-6 tap kernels:
    // thumbnails +1.8 % at 1,200 frames, 0 at 300; 4K -> portrait and the web box within box-to-box noise.)
    if (ch.lds == 2 && ch.g12 == 2 && ch.LW == 8 && ch.R == 8) ch.depth = kn.strip_d == 1 ? 1 : 4;
    return ch;
}

// y16: the instantiation with the 16-bit luma store (scratch plans; the grouped 12-byte-window kernels only)
bool strip_has_y16(const StripChoice &ch) { return ch.lds && ch.g12; }
// rng: the instantiation with the range step on the 15-bit rows (tts_geom.sc_range; the grouped-window LDS kernels only:
// every enlargement and the usual shrinks up to 2x -- what they do not cover takes the column walk, decided when the plan is built)
bool strip_has_rng(const StripChoice &ch) { return ch.lds && ch.g4; }

strip_fn pick_strip(const StripChoice &ch, const Knobs &kn, bool y16 = false, bool rng = false)
{
    const int D = kn.strip_d;                          // rows in flight of the register kernels (tuning knob)
    if (y16 && !strip_has_y16(ch)) return nullptr;
    if (rng && (y16 || !strip_has_rng(ch))) return nullptr;
#define TTS_R(LW_, R_, GC_, NDMA_, D_) (rng ? strip_lds_kernel<LW_, R_, GC_, NDMA_, D_, false, true> : strip_lds_kernel<LW_, R_, GC_, NDMA_, D_>)
    if (ch.lds && ch.g4 && !ch.g12) {
        if (ch.R == 4) return ch.depth == 2 ? TTS_R(4, 4, 4, 1, 2) : TTS_R(4, 4, 4, 1, 1);
        return ch.R == 6 ? TTS_R(4, 6, 4, 1, 1) : TTS_R(4, 8, 4, 1, 1);
    }
#define TTS_Y(LW_, R_, GC_, NDMA_, D_) (y16 ? strip_lds_kernel<LW_, R_, GC_, NDMA_, D_, true> : TTS_R(LW_, R_, GC_, NDMA_, D_))
    if (ch.lds && ch.g12 == 4) return ch.R == 4 ? TTS_Y(6, 4, 4, 1, 1) : ch.R == 6 ? TTS_Y(6, 6, 4, 1, 1) : TTS_Y(6, 8, 4, 1, 1);
    if (ch.lds && ch.g12 == 2) {
#define TTS_G2_PICK(LW_, R_) (ch.lds == 1 ? TTS_Y(LW_, R_, 2, 1, 1) : ch.lds == 2 ? TTS_Y(LW_, R_, 2, 2, 1) : TTS_Y(LW_, R_, 2, 4, 1))
        if (ch.depth == 4) return TTS_Y(8, 8, 2, 2, 4);
        if (ch.LW == 6) return ch.R == 4 ? TTS_G2_PICK(6, 4) : ch.R == 6 ? TTS_G2_PICK(6, 6) : TTS_G2_PICK(6, 8);
        return ch.R == 4 ? TTS_G2_PICK(8, 4) : ch.R == 6 ? TTS_G2_PICK(8, 6) : TTS_G2_PICK(8, 8);
#undef TTS_G2_PICK
    }
#undef TTS_Y
#undef TTS_R
    if (ch.lds) {
#define TTS_LDS_PICK(LW_, R_) (ch.lds == 1 ? strip_lds_kernel<LW_, R_, 0, 1, 1> : ch.lds == 2 ? strip_lds_kernel<LW_, R_, 0, 2, 1> : strip_lds_kernel<LW_, R_, 0, 4, 1>)
        if (ch.LW == 4) return ch.R == 4 ? TTS_LDS_PICK(4, 4) : ch.R == 6 ? TTS_LDS_PICK(4, 6) : TTS_LDS_PICK(4, 8);
        if (ch.LW == 6) return ch.R == 4 ? TTS_LDS_PICK(6, 4) : ch.R == 6 ? TTS_LDS_PICK(6, 6) : TTS_LDS_PICK(6, 8);
        return ch.R == 4 ? TTS_LDS_PICK(8, 4) : ch.R == 6 ? TTS_LDS_PICK(8, 6) : TTS_LDS_PICK(8, 8);
#undef TTS_LDS_PICK
    }
    if (ch.LW == 4 && ch.R == 4 && ch.g4) {
        if (ch.N == 16) return strip_kernel<4, 4, 16, 1, true>;
        return D == 4 ? strip_kernel<4, 4, 8, 4, true> : D == 2 ? strip_kernel<4, 4, 8, 2, true> : strip_kernel<4, 4, 8, 1, true>;
    }
    if (ch.LW == 4 && ch.R == 4) return ch.N == 16 ? strip_kernel<4, 4, 16, 1, false> : strip_kernel<4, 4, 8, 2, false>;
    if (ch.LW == 4 && ch.R == 6) return ch.g4 ? strip_kernel<4, 6, 8, 2, true> : strip_kernel<4, 6, 8, 2, false>;
    if (ch.LW == 4 && ch.R == 8) return ch.g4 ? strip_kernel<4, 8, 8, 2, true> : strip_kernel<4, 8, 8, 2, false>;
    if (ch.LW == 8 && ch.R == 4) return strip_kernel<8, 4, 8, 2, false>;
    if (ch.LW == 8 && ch.R == 6) return strip_kernel<8, 6, 8, 2, false>;
    if (ch.LW == 8 && ch.R == 8) return strip_kernel<8, 8, 8, 2, false>;
    return nullptr;
}

void fill_strip_class(const tts_plan *p, int ci, int n, int N, bool g4, int lw, int src_stride, int dst_stride,
                      const SegTables &st, int block, int paced, int planes, int plane0, StripClass &sc)
{
    const ClassPlan &c = p->cls[ci];
    const HTables &t = c.ht[N == 16];
    sc.hoff = g4 ? t.hoff4 : t.hoff; sc.hcoef = g4 ? t.hcoef4 : t.hcoef; sc.hsel = t.hsel4;
    if (!g4 && lw == 6) { sc.hoff = t.hoff6; sc.hcoef = t.hcoef6; }
    if (g4 && lw >= 6) { sc.hoff = t.g12.hoff; sc.hcoef = t.g12.hcoef; sc.hsel = t.g12.hsel; }
    sc.seginfo = st.seginfo; sc.emit = st.emit; sc.vcoef = st.vcoef;
    sc.maxrows = st.maxrows; sc.nseg = st.nseg; sc.seg_rows = st.seg_rows;
    sc.strips = t.strips;
    sc.planes = planes; sc.plane0 = plane0;
    sc.n_items = n * sc.planes * sc.strips;
    sc.ow = c.ow; sc.oh = c.oh; sc.src_stride = p->knobs.exp_stride0 ? 0 : src_stride; sc.dst_stride = dst_stride;
    sc.nblocks = uint32_t((sc.n_items + block - 1) / block) * uint32_t(st.nseg);
    sc.paced = paced;
    {   // the source plane of this class, seen from the pre-cropped origin the frame offsets point at
        const int sub = ci ? 1 : 0;
        const int plane_w = ceil_shift(p->g.in_w, sub), plane_h = ceil_shift(p->g.in_h, sub);
        sc.src_begin_rel = uint32_t(p->g.pre_y >> sub) * uint32_t(src_stride) + uint32_t(p->g.pre_x >> sub);
        sc.src_end_rel = uint32_t(plane_h - (p->g.pre_y >> sub) - 1) * uint32_t(src_stride) + uint32_t(plane_w - (p->g.pre_x >> sub));
    sc.src_row_rel = uint32_t(plane_w - (p->g.pre_x >> sub));
        sc.src_row_rel = uint32_t(plane_w - (p->g.pre_x >> sub));
    }
}

// Resident workgroups per CU of a strip kernel (VGPR- and LDS-limited); asked once per kernel.
int blocks_per_cu(strip_fn fn, int block, size_t lds_bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<std::pair<strip_fn, int>, int>> cache;
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<strip_fn, int> key(fn, block);
    for (auto &e : cache) if (e.first == key) return e.second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, lds_bytes) != hipSuccess || nb < 1) nb = 1;
    cache.emplace_back(key, nb);
    return nb;
}

// Workgroup shape of a strip launch.  The register kernels run 256 threads.  The LDS-staged kernel
// takes any multiple of 64 up to 1024 (one LDS slot per wave) and has a PACED mode for jobs that
// are bound by their writes, large enough to fill the chip with such workgroups (~100 1080p frames),
// while the output rows are not sector-aligned (1080-byte rows: a wave's
// 512-byte chunk then ends inside a 64-byte sector it shares with the next wave on 7 rows of 8):
// 1024-thread workgroups of neighbouring strips, dealt strip-major so that neighbours also share
// an XCD, whose waves meet at a barrier before every output row's store.  The halves of a shared
// sector then reach L2 within one row's time instead of whenever.  Measured (300 frames):
// 1080p->portrait 0.370 -> 0.330 ms, 720p->portrait 0.380 -> 0.327 ms; jobs that read more than they
// write (4K->portrait) or write aligned rows lose 3-10 % to the barrier and stay unpaced.
struct BlockShape { int threads; size_t lds_bytes; int paced; };
BlockShape block_shape(const tts_plan *p, int ci, const StripChoice &ch, int dst_stride, int max_items, int64_t total_items)
{
    if (!ch.lds) return {kBlock, 0, 0};
    const ClassPlan &c = p->cls[ci];
    const bool misaligned = dst_stride % 64 != 0 || c.ow % 64 != 0 || c.pad_x % 64 != 0;
    const int src_rows = c.v.pos[c.y0 + c.oh - 1] + c.v.size - c.v.pos[c.y0];
    const int src_cols = c.h.pos[c.x0 + c.ow - 1] + c.h.size - c.h.pos[c.x0];
    const bool write_bound = 2.0 * c.ow * c.oh >= 3.0 * src_rows * src_cols;     // writes >= 60 % of the traffic
    // (below ~100 strips per CU there are too few 1024-thread workgroups to keep the chip busy: 16
    // 1080p frames take 44 us paced, 29 us not; the gain starts at ~100 frames)
    int paced = misaligned && write_bound && total_items >= int64_t(96) * p->n_cus;
    if (p->knobs.paced >= 0) paced = p->knobs.paced;
    // (a paced workgroup is no larger than the biggest class needs: idle waves still walk every row)
    int t = p->knobs.lds_block > 0 ? p->knobs.lds_block : paced ? std::max(kBlock, std::min(kMaxLdsBlock, (max_items + 63) / 64 * 64)) : kBlock;
    t = std::min(std::max(t / 64 * 64, 64), kMaxLdsBlock);
    return {t, size_t(t / 64) * ch.depth * (size_t(ch.lds) * 1024 + 16), paced};
}

// Pick the row-segment length (one for both plane classes).  Long segments amortise the (taps-1)-row
// halo repeated at every segment start; short ones give enough workgroups to fill the chip and to
// keep the tail short when the last ones drain.  Measured on MI355X the time is flat from 64 to 96
// rows once there are >= ~3 workgroups per resident slot, so: the longest segment <= 96 rows that
// still leaves that many workgroups, else the shortest on offer.
void choose_segs(const tts_plan *p, const int *cis, int ncls, int n, int N, strip_fn fn, const BlockShape &bs, int min_segs, size_t out_idx[2])
{
    out_idx[0] = out_idx[1] = 0;
    const size_t nopt = p->cls[cis[0]].segs.size();
    if (nopt == 1) return;
    const double slots = double(p->n_cus) * blocks_per_cu(fn, bs.threads, bs.lds_bytes);
    size_t best = 0;
    for (size_t i = 0; i < nopt; ++i) {
        if (p->cls[cis[0]].segs[i].seg_rows > 96) break;
        double blocks = 0;
        for (int k = 0; k < ncls; ++k) {
            const ClassPlan &c = p->cls[cis[k]];
            const double sb = double((int64_t(n) * (cis[k] ? 2 : 1) * c.ht[N == 16].strips + bs.threads - 1) / bs.threads);
            blocks += sb * c.segs[i].nseg;
        }
        if (i == 0 || blocks >= (bs.paced ? 2.0 : 3.0) * slots) best = i;   // (paced workgroups are 4x as large)
    }
    for (bool ok = false; !ok && best > 0; ) {         // pointer mode + LDS-staged kernel: at least min_segs segments per plane
        ok = true;
        for (int k = 0; k < ncls; ++k) ok &= p->cls[cis[k]].segs[best].nseg >= min_segs && !(min_segs > 1 && p->cls[cis[k]].segs[best].spans_plane);
        if (!ok) --best;
    }
    out_idx[0] = out_idx[1] = best;
}

// The v_dot4 form (see column_lds_kernel) pays above 8 tap pairs -- 300 frames, A/B in one process: the 1080p
// thumbnail chain 47.4 -> 49.9 % of the roofline, 1080x1920 -> 101x180 45.1 -> 48.6 %, the 4K chain 56.9 -> 59.5 % --
// and costs below (a window of 5 dwords gains little from dropping v_perm and pays the xor and the second
// column's two extra dwords: 4K -> 720p 54.5 -> 51.8 %, 2160x3840 -> 720x1280 52.7 -> 51.5 %).
bool column_dot4(const ClassPlan &c, const Knobs &k)
{
    if (!(k.col_d4 == 2 || (k.col_d4 == 1 && c.NTP > 8))) return false;
    return c.d4tab[1] && (c.col_nc != 2 || c.d4tab[2]);
}

constexpr int column_depth(int) { return 4; }    // source rows in flight per wave, every tap count (see pick_column)

struct ColMode { int nc, dmas, d4, rng = 0; };     // columns per lane, LDS-DMAs per row, v_dot4 form, range step on the 15-bit rows: of one launch (per class)

strip_fn pick_column(const ClassPlan &c, const ColMode m)
{
    // A wave has one or two columns' arithmetic per source row against a memory latency of microseconds,
    // so four rows are kept in flight (four LDS slots): the 1080p thumbnail chain 35 -> 39-41 % of peak,
    // 4K -> 720p 46 -> 51 %, the 48-tap 4K thumbnail chain 48 -> 54 %.  Eight measured worse than four
    // (1080p chain 34 %, 4K chain 51 %).
#define TTS_COL_PICK_(NTP_, NC_, RNG_) (m.d4 ? (m.dmas == 1 ? column_lds_kernel<NTP_, 2 * NTP_, 1, NC_, column_depth(NTP_), true, RNG_> \
                                                            : column_lds_kernel<NTP_, 2 * NTP_, 2, NC_, column_depth(NTP_), true, RNG_>) \
                                       : (m.dmas == 1 ? column_lds_kernel<NTP_, 2 * NTP_, 1, NC_, column_depth(NTP_), false, RNG_> \
                                                            : column_lds_kernel<NTP_, 2 * NTP_, 2, NC_, column_depth(NTP_), false, RNG_>))
#define TTS_COL_PICK(NTP_, NC_) (m.rng ? TTS_COL_PICK_(NTP_, NC_, true) : TTS_COL_PICK_(NTP_, NC_, false))
    if (c.Rc != 2 * c.NTP) return nullptr;
    if (m.nc == 2)
        switch (c.NTP) {
        case 6: return TTS_COL_PICK(6, 2);
        case 8: return TTS_COL_PICK(8, 2);
        case 10: return TTS_COL_PICK(10, 2);
        case 12: return TTS_COL_PICK(12, 2);
        default: return nullptr;
        }
    switch (c.NTP) {
    case 6: return TTS_COL_PICK(6, 1);
    case 8: return TTS_COL_PICK(8, 1);
    case 10: return TTS_COL_PICK(10, 1);
    case 12: return TTS_COL_PICK(12, 1);
    case 14: return TTS_COL_PICK(14, 1);
    case 16: return TTS_COL_PICK(16, 1);
    case 18: return TTS_COL_PICK(18, 1);
    case 20: return TTS_COL_PICK(20, 1);
    case 22: return TTS_COL_PICK(22, 1);
    case 24: return TTS_COL_PICK(24, 1);
    case 26: return TTS_COL_PICK(26, 1);
    case 28: return TTS_COL_PICK(28, 1);
    case 30: return TTS_COL_PICK(30, 1);
    case 32: return TTS_COL_PICK(32, 1);
    default: return nullptr;
    }
#undef TTS_COL_PICK
#undef TTS_COL_PICK_
}

void fill_column_class(const tts_plan *p, int ci, const ColMode m, int n, int src_stride, int dst_stride, const SegTables &st,
                       int planes, int plane0, StripClass &sc)
{
    const ClassPlan &c = p->cls[ci];
    sc = StripClass{};
    sc.hoff = c.hoff1; sc.hcoef = m.d4 ? c.d4tab[m.nc] : c.hcoef1;
    sc.seginfo = st.seginfo; sc.emit = st.emit; sc.vcoef = st.vcoef;
    sc.maxrows = st.maxrows; sc.nseg = st.nseg; sc.seg_rows = st.seg_rows;
    sc.strips = (c.ow + m.nc - 1) / m.nc;              // nc columns per lane
    sc.planes = planes; sc.plane0 = plane0;
    sc.n_items = n * planes * sc.strips;
    sc.ow = c.ow; sc.oh = c.oh; sc.src_stride = src_stride; sc.dst_stride = dst_stride;
    sc.nblocks = uint32_t((sc.n_items + kBlock - 1) / kBlock) * uint32_t(st.nseg);
    const int sub = ci ? 1 : 0;
    const int plane_w = ceil_shift(p->g.in_w, sub), plane_h = ceil_shift(p->g.in_h, sub);
    sc.src_begin_rel = uint32_t(p->g.pre_y >> sub) * uint32_t(src_stride) + uint32_t(p->g.pre_x >> sub);
    sc.src_end_rel = uint32_t(plane_h - (p->g.pre_y >> sub) - 1) * uint32_t(src_stride) + uint32_t(plane_w - (p->g.pre_x >> sub));
}

// Row-segment length of a column-walk launch: the longest segment on offer that leaves >= 1.5
// workgroups per resident slot (every segment start re-reads taps - 1 source rows: 24 of 120 at
// 16-row segments of the 1080p thumbnail) -- >= 3 up to 8 tap pairs, where a start costs little
// (4K -> 720p 50.7 -> 54.7 % of peak at 300 frames, 2160x3840 -> 720x1280 48.1 -> 52.8 %) -- else the
// shortest; at least `min_segs` segments, and two wherever there is a choice: whole-plane segments
// measured 15-20 % slower than halves at any fill (4K -> 720p, 1,200 frames: 43.8 vs 55.3 %; the 4K
// thumbnail chain, 900 frames: 46.9 vs 56.3 %; the 1080p one, 1,200 frames: 40.1 vs 50.1 %).
size_t choose_column_segs(const tts_plan *p, const int *cis, const ColMode *cm, int ncls, int n, int planes_each[2], strip_fn fn, int min_segs)
{
    const size_t nopt = p->cls[cis[0]].segs.size();
    const double slots = double(p->n_cus) * blocks_per_cu(fn, kBlock, 0);
    const double fill = p->knobs.col_fill ? 0.1 * p->knobs.col_fill : p->cls[cis[0]].NTP <= 8 ? 3.0 : 1.5;
    size_t best = 0;
    double best_cost = 0;
    for (size_t i = 0; i < nopt; ++i) {
        double blocks = 0;
        int rows = 0;
        for (int k = 0; k < ncls; ++k) {
            const ClassPlan &c = p->cls[cis[k]];
            blocks += double((int64_t(n) * planes_each[k] * ((c.ow + cm[cis[k]].nc - 1) / cm[cis[k]].nc) + kBlock - 1) / kBlock) * c.segs[i].nseg;
            rows = std::max(rows, c.segs[i].maxrows);
        }
        if (p->knobs.col_model == 2 || (p->knobs.col_model == 1 && p->cls[cis[0]].NTP > 8)) {
            // Many taps: the walk is bound by its own instruction stream (the 1080p thumbnail chain runs in 210 us
            // with all memory traffic dropped, 261 us with it: tools/decompose.py), so a launch lasts about as long
            // as the busiest CU's share of source rows, halo rows included: ceil(workgroups / CUs) x rows per
            // segment.  At least 3 workgroups per CU (= waves per SIMD), or nothing covers the memory latency.
            // Against the fill rule below, 300 frames: 1080x1920 -> 101x180 40.2 -> 48.9 % of the roofline.
            const double per_cu = std::ceil(blocks / p->n_cus);
            const double cost = per_cu * rows;
            const bool ok = per_cu >= 3 && (p->cls[cis[0]].segs[i].nseg >= 2 || nopt == 1);
#ifdef TTS_EXPERIMENTS
            if (std::getenv("TTS_EXP_LOG_SEGS")) std::fprintf(stderr, "  seg option %zu: rows/seg %d nseg %d blocks %.0f per CU %.0f rows %d cost %.0f ok %d\n", i, p->cls[cis[0]].segs[i].seg_rows, p->cls[cis[0]].segs[i].nseg, blocks, per_cu, rows, cost, int(ok));
#endif
            if (i == 0 || (ok && (best_cost == 0 || cost <= best_cost))) { best = i; if (ok) best_cost = cost; }
            continue;
        }
        if (i == 0 || (blocks >= fill * slots && p->cls[cis[0]].segs[i].nseg >= 2)) best = i;
    }
    for (bool ok = false; !ok && best > 0; ) {
        ok = true;
        for (int k = 0; k < ncls; ++k) ok &= p->cls[cis[k]].segs[best].nseg >= min_segs && !(min_segs > 1 && p->cls[cis[k]].segs[best].spans_plane);
        if (!ok) --best;
    }
    return best;
}

// One launch set: frames [0, n) addressed by `addr` relative to (src, dst).
// `only_plane` >= 0 restricts the launch set to that one plane (per-frame pointers whose planes lie
// too far apart for one base + 32-bit offsets are processed plane by plane).
int launch_all(const tts_plan *p, const uint8_t *src, uint8_t *dst, int n, const int src_stride[2],
               const int dst_stride[2], uint64_t src_span, uint64_t dst_span, const FrameAddr &addr, hipStream_t st,
               int only_plane = -1)
{
    // the plain kernels index frames by grid.z (at most 65535 planes a launch)
    if (p->rgb) {
        if (only_plane >= 0) { g_last_error = "packed RGB output needs the planes of a frame within 4 GiB of each other"; return TTS_ERR_UNSUPPORTED; }
        return tts::rgb_launch(p->rgb, src, dst, n, src_stride, dst_stride[0], src_span, dst_span, addr, st);
    }
    int lrc = TTS_OK;                                  // first failed launch, if any
    auto z_launches = [&](int planes, auto &&launch) {
        const int per = 65535 / planes;
        for (int f0 = 0; f0 < n && !lrc; f0 += per) lrc = launch(f0, std::min(per, n - f0));
    };
    auto active  = [&](int ci) { return only_plane < 0 || ci == (only_plane ? 1 : 0); };
    auto nplanes = [&](int ci) { return only_plane < 0 ? (ci ? 2 : 1) : 1; };
    auto plane0  = [&](int ci) { return only_plane < 0 ? (ci ? 1 : 0) : only_plane; };
    // Which classes take the generic kernel in this launch set?  Their pad border and range step stay kernels
    // of their own (fill_kernel before, range_kernel after); the strip and column walks carry both themselves:
    // the border is written by extra workgroups of the scaler's launch, the range step sits in its store
    // epilogue, so the reference's thumbnail chain (src/services/video.js:329-334) is ONE launch.
    // (sc_range, a range step inside the scale's context: the RNG instantiations of the grouped strip kernels and of the column
    // walk and the general kernel carry it; out_range, the pass at the end, is independent of it -- a full-range source
    // thumbnailed the reference's way has both)
    const bool range_in_scale = p->g.in_fmt == TTS_FMT_YUV420P && p->g.out_fmt == TTS_FMT_YUV420P && p->g.sc_range != TTS_RANGE_KEEP;
    bool to_generic[2] = {false, false};
    if (range_in_scale) {
        // the register strip kernels (tiny sources, per-frame pointers into short planes: see the strip classes below) have no
        // such instantiation: then every strip class of the launch set takes the general kernel
        bool short_plane = false, no_rng = false;
        for (int ci = 0; ci < 2; ++ci) {
            const ClassPlan &c = p->cls[ci];
            if (c.kind != 1 || !active(ci)) continue;
            short_plane |= c.segs[0].nseg < 2 || c.segs[0].spans_plane || c.src_w < 48 || c.src_h < 32;
            no_rng |= !strip_has_rng(strip_choice(c, p->knobs));
        }
        if (no_rng || src_span < 4096u || (addr.use_table && short_plane))
            for (int ci = 0; ci < 2; ++ci) if (p->cls[ci].kind == 1) to_generic[ci] = true;
    }
    for (int ci = 0; ci < 2; ++ci) {
        const ClassPlan &c = p->cls[ci];
        if (c.kind == 0) to_generic[ci] = true;
        if (c.kind == 2 && (src_span < 4096u || (addr.use_table && (c.segs[0].nseg < 2 || c.segs[0].spans_plane || c.src_w < 48 || c.src_h < 32)))) to_generic[ci] = true;
        // (the scratch plan of the packed-RGB shrinks: only the column walk and the grouped-window strip kernels
        // store int16 luma; the caller falls back)
        if (p->y16 && (c.kind == 0 || to_generic[ci]) && active(ci)) { g_last_error = "scratch pass needs the column walk or an LDS strip kernel"; return TTS_ERR_UNSUPPORTED; }
    }
    const int range_at_end = p->g.out_range;
    RangeStep range{};
    if (range_at_end != TTS_RANGE_KEEP) {
        range.on = 1;
        for (int ci = 0; ci < 2; ++ci)
            if (!tts::range_affine(p->g.out_range, ci != 0, range.mul[ci], range.add[ci], range.shift[ci])) range.on = 0;
    }
    const bool fuse_range = range.on != 0;             // (always, with the constants of filter.cpp; else every class keeps range_kernel)
    auto pad_class = [&](int ci, int threads, PadClass &pc) {
        const ClassPlan &c = p->cls[ci];
        pc = PadClass{};
        if (!p->g.do_pad) return;
        if (c.pad_x == 0 && c.pad_y == 0 && c.ow == c.canvas_w && c.oh == c.canvas_h) return;   // the picture fills the canvas (1080p -> 320x180): no border
        pc.w = c.canvas_w; pc.h = c.canvas_h; pc.stride = dst_stride[ci];
        pc.px = c.pad_x; pc.py = c.pad_y; pc.pw = c.ow; pc.ph = c.oh;
        pc.planes = nplanes(ci); pc.plane0 = plane0(ci);
        const int rows_wg = threads / 64 * kPadRowsPerWave;        // rows a pad workgroup covers
        pc.chunks = (c.canvas_w + 255) / 256; pc.rowgroups = (c.canvas_h - c.oh + rows_wg - 1) / rows_wg;          // above and below the picture
        pc.chunks_sd = (c.canvas_w - c.ow + 255) / 256; pc.rgs_sd = pc.chunks_sd ? (c.oh + rows_wg - 1) / rows_wg : 0;   // left and right of it
        if (!pc.rowgroups) pc.chunks = 0;
        pc.origin = uint32_t(c.pad_y) * uint32_t(dst_stride[ci]) + uint32_t(c.pad_x);
        const int fill[3] = {p->g.fill_y, p->g.fill_u, p->g.fill_v};
        for (int k = 0; k < 3; ++k) {
            pc.fill[k] = fill[k];
            if (fuse_range) {                          // the border goes through the range step like every other byte
                unsigned char lut[256];
                tts::build_range_lut(p->g.out_range, k != 0, lut);
                pc.fill[k] = lut[fill[k] & 255];
            }
        }
        pc.nblocks = (uint32_t(pc.chunks) * uint32_t(pc.rowgroups) + uint32_t(pc.chunks_sd) * uint32_t(pc.rgs_sd)) * uint32_t(n) * uint32_t(pc.planes);
    };
    if (p->g.do_pad) {
        for (int ci = 0; ci < 2; ++ci) {
            if (!active(ci) || !to_generic[ci]) continue;
            const ClassPlan &c = p->cls[ci];
            FillArgs f{};
            f.dst = dst; f.w = c.canvas_w; f.h = c.canvas_h; f.stride = dst_stride[ci];
            f.px = c.pad_x; f.py = c.pad_y; f.pw = c.ow; f.ph = c.oh;
            f.planes = nplanes(ci); f.plane0 = plane0(ci);
            f.fill[0] = p->g.fill_y; f.fill[1] = p->g.fill_u; f.fill[2] = p->g.fill_v;
            f.origin = uint32_t(c.pad_y) * uint32_t(dst_stride[ci]) + uint32_t(c.pad_x);
            f.addr = addr;
            z_launches(f.planes, [&](int f0, int nf) {
                f.frame0 = f0;
                dim3 grid((c.canvas_w + 255) / 256, (c.canvas_h + 3) / 4, nf * f.planes);
                return launch_checked("fill_kernel", fill_kernel, grid, dim3(kBlock), 0, st, f);
            });
            if (lrc) return lrc;
        }
    }

    // timing experiments (-DTTS_EXPERIMENTS builds; outputs are wrong): a zero-record descriptor makes the
    // range check drop that buffer's traffic while the instruction stream stays
    auto exp_knobs = [&](StripArgs &a) {
        if (p->knobs.exp_nosrc) a.src_span = 0;
        if (p->knobs.exp_srcpitch0) a.addr.src_pitch = 0;    // every frame reads frame 0: source stays in L2
        if (p->knobs.exp_dstpitch0) a.addr.dst_pitch = 0;    // every frame writes frame 0: output stays in L2
        if (p->knobs.exp_nodst) a.dst_span = 0;
    };
    // strip classes: one launch when both classes run the same instantiation, else one each
    int strip_cls[2], ns = 0;
    for (int ci = 0; ci < 2; ++ci) if (p->cls[ci].kind == 1 && active(ci) && !to_generic[ci]) strip_cls[ns++] = ci;
    // swscale.c lumRangeToJpeg_c / chrRangeToJpeg_c / lumRangeFromJpeg_c / chrRangeFromJpeg_c: {cap, mul, add, shift} per class
    static const int32_t r15_to_full[2][4] = {{30189, 19077, -39057361, 14}, {30775, 4663, -9289992, 12}};
    static const int32_t r15_to_limited[2][4] = {{32767, 14071, 33561947, 14}, {32767, 1799, 4081085, 11}};
    auto set_r15 = [&](StripArgs &a, bool both, int ci) {   // (a single class rides in slot 0)
        const int32_t (*t)[4] = p->g.sc_range == TTS_RANGE_TO_FULL ? r15_to_full : r15_to_limited;
        for (int slot = 0; slot < 2; ++slot) {
            const int k = both ? slot : ci;
            a.r15.cap[slot] = t[k][0]; a.r15.mul[slot] = t[k][1]; a.r15.add[slot] = t[k][2]; a.r15.shift[slot] = t[k][3];
        }
    };
    bool merged = false;
    if (ns == 2 && !p->knobs.no_merge) {
        const StripChoice a0 = strip_choice(p->cls[0], p->knobs), a1 = strip_choice(p->cls[1], p->knobs);
        merged = a0.LW == a1.LW && a0.R == a1.R && a0.N == a1.N && a0.g4 == a1.g4 && a0.g12 == a1.g12 && a0.lds == a1.lds && a0.depth == a1.depth;
    }
    for (int k = 0; k < ns; ++k) {
        StripArgs a{};
        a.src_span = src_span; a.dst_span = dst_span; a.addr = addr;
        exp_knobs(a);
        const int ci = strip_cls[k];
        StripChoice ch = strip_choice(p->cls[ci], p->knobs);
        // Per-frame plane pointers: a segment of the LDS-staged kernel must not hold both the first and
        // the last source row of a plane (see the kernel): planes whose shortest segments still do -- a
        // single segment, or so many vertical taps that the first segment reaches the last row -- take
        // the register kernel; others are cut into segments that do not (choose_segs).
        // (source planes under 48 bytes a row or 32 rows likewise: a run slid down from the plane's end
        // must still begin inside the plane)
        bool short_plane = false;
        for (int j = 0; j < ns; ++j) {
            const ClassPlan &cj = p->cls[strip_cls[j]];
            short_plane |= cj.segs[0].nseg < 2 || cj.segs[0].spans_plane || cj.src_w < 48 || cj.src_h < 32;
        }
        if (src_span < 4096u || (addr.use_table && short_plane)) { ch.lds = 0; ch.depth = 1; ch.LW = p->cls[ci].LW; if (ch.g12) { ch.g4 = false; ch.g12 = 0; } }
        a.y16 = p->y16 && (merged || ci == 0);
        if (p->y16 && !strip_has_y16(ch)) return TTS_ERR_UNSUPPORTED;   // (the caller falls back to the RGB kernels)
        strip_fn fn = pick_strip(ch, p->knobs, a.y16 != 0, range_in_scale);
        if (!fn) return TTS_ERR_UNSUPPORTED;
        if (range_in_scale) set_r15(a, merged, ci);
        a.no_union = p->knobs.no_union;
        size_t si[2];
        int max_items = 0;
        int64_t total_items = 0;
        for (int j = 0; j < ns; ++j) {
            const int cj = strip_cls[j];
            if (!(merged || cj == ci)) continue;
            const int items = n * nplanes(cj) * p->cls[cj].ht[ch.N == 16].strips;
            max_items = std::max(max_items, items); total_items += items;
        }
        const BlockShape bs = block_shape(p, ci, ch, dst_stride[ci], max_items, total_items);
        const int min_segs = addr.use_table && ch.lds ? 2 : 1;
        if (merged) {
            choose_segs(p, strip_cls, 2, n, ch.N, fn, bs, min_segs, si);
            for (int j = 0; j < 2; ++j)
                fill_strip_class(p, j, n, ch.N, ch.g4, ch.LW, src_stride[j], dst_stride[j], p->cls[j].segs[si[j]], bs.threads, bs.paced, nplanes(j), plane0(j), a.c[j]);
        } else {
            // a single class rides in slot 0; slot 1 is empty
            choose_segs(p, &ci, 1, n, ch.N, fn, bs, min_segs, si);
            fill_strip_class(p, ci, n, ch.N, ch.g4, ch.LW, src_stride[ci], dst_stride[ci], p->cls[ci].segs[si[0]], bs.threads, bs.paced, nplanes(ci), plane0(ci), a.c[0]);
            a.c[1] = a.c[0]; a.c[1].nblocks = 0;
        }
        a.scaler_blocks = a.c[0].nblocks + (merged ? a.c[1].nblocks : 0);
        a.grid_scaler = a.scaler_blocks; a.queue = nullptr;
        if (p->knobs.persist && ch.lds && p->d_queue) {
            // work queue: as many workgroups as stay resident (x persist / 10), a multiple of 8; the rest of the logical blocks are taken from the per-XCD counters
            const uint32_t slots = uint32_t(p->n_cus) * uint32_t(blocks_per_cu(fn, bs.threads, bs.lds_bytes)) * uint32_t(p->knobs.persist) / 10u & ~7u;
            if (slots >= 8 && a.scaler_blocks > slots) {
                if (hipMemsetAsync(p->d_queue, 0, 64, st) != hipSuccess) return TTS_ERR_PROCESSING_FAILED;
                a.grid_scaler = slots; a.queue = p->d_queue;
            }
        }
#ifdef TTS_EXPERIMENTS
        a.dbg = exp_dbg_buffer(a.scaler_blocks);
#endif
        if (fuse_range) a.range = range;
        if (merged) { pad_class(0, bs.threads, a.pad[0]); pad_class(1, bs.threads, a.pad[1]); }
        else {
            pad_class(ci, bs.threads, a.pad[0]);
            if (fuse_range && ci) { a.range.mul[0] = range.mul[1]; a.range.add[0] = range.add[1]; a.range.shift[0] = range.shift[1]; }   // (a single class rides in slot 0)
        }
        const uint32_t blocks = a.grid_scaler + a.pad[0].nblocks + a.pad[1].nblocks;
        if (const int rc = launch_checked("strip kernel", fn, dim3(blocks), dim3(bs.threads), bs.lds_bytes, st, src, dst, p->d_tab, a)) return rc;
        if (merged) break;
    }

    // column classes (9..64 taps): one launch when both classes run the same instantiation
    {
        int col_cls[2], nc = 0;
        for (int ci = 0; ci < 2; ++ci) {
            const ClassPlan &c = p->cls[ci];
            if (c.kind != 2 || !active(ci)) continue;
            // (per-frame plane pointers: see the strip classes above; tiny sources likewise: to_generic, decided above)
            if (!to_generic[ci]) col_cls[nc++] = ci;
        }
        const int min_segs = addr.use_table ? 2 : 1;
        // the v_dot4 form where every class of the launch has its digit tables (for either column count)
        int d4 = 1;
        for (int k = 0; k < nc; ++k) d4 &= int(column_dot4(p->cls[col_cls[k]], p->knobs));
        const int rng15 = range_in_scale ? 1 : 0;
        ColMode cm[2] = {{p->cls[0].col_nc, p->cls[0].col_dmas, d4, rng15}, {p->cls[1].col_nc, p->cls[1].col_dmas, d4, rng15}};
        auto cmerged_now = [&] { return nc == 2 && !p->knobs.no_merge && pick_column(p->cls[0], cm[0]) == pick_column(p->cls[1], cm[1]); };
        // Above 8 tap pairs two columns per lane pay only on large launches (1080p thumbnail chain: +2 % at
        // 300 frames, +8.5 % at 600, but -10 % at 120 and -5 % at 30): half as many workgroups want shorter
        // row segments to fill the chip, and every segment start re-reads taps - 1 rows.  So such a launch
        // takes two only where that still leaves segments of an eighth of the plane or longer.
        int decided = -1;                                  // one answer for both classes, so that they stay one launch
        for (int k = 0; k < nc; ++k) {
            const int ci = col_cls[k];
            const ClassPlan &c = p->cls[ci];
            if (c.col_nc != 2 || c.NTP <= 8 || !c.col_dmas1 || p->knobs.col_nc_launch == 2) continue;
            bool two = decided > 0;
            if (decided < 0 && p->knobs.col_nc_launch != 1) {
                strip_fn fn2 = pick_column(c, cm[ci]);
                if (fn2) {
                    const bool m2 = cmerged_now();
                    int planes_each[2] = {m2 ? nplanes(0) : nplanes(ci), m2 ? nplanes(1) : 0};
                    const size_t si = m2 ? choose_column_segs(p, col_cls, cm, 2, n, planes_each, fn2, min_segs)
                                         : choose_column_segs(p, &ci, cm, 1, n, planes_each, fn2, min_segs);
                    two = c.segs[si].nseg <= 8;
                }
            }
            decided = two;
            if (!two) cm[ci] = {1, c.col_dmas1, d4, rng15};
        }
        const bool cmerged = cmerged_now();
        for (int k = 0; k < nc; ++k) {
            const int ci = col_cls[k];
            strip_fn fn = pick_column(p->cls[ci], cm[ci]);
            if (!fn) return TTS_ERR_UNSUPPORTED;
            StripArgs a{};
            a.src_span = src_span; a.dst_span = dst_span; a.addr = addr;
            exp_knobs(a);
            if (cmerged) {
                int planes_each[2] = {nplanes(0), nplanes(1)};
                const size_t si = choose_column_segs(p, col_cls, cm, 2, n, planes_each, fn, min_segs);
                for (int j = 0; j < 2; ++j)
                    fill_column_class(p, j, cm[j], n, src_stride[j], dst_stride[j], p->cls[j].segs[si], nplanes(j), plane0(j), a.c[j]);
            } else {
                int planes_each[2] = {nplanes(ci), 0};
                const size_t si = choose_column_segs(p, &ci, cm, 1, n, planes_each, fn, min_segs);
                fill_column_class(p, ci, cm[ci], n, src_stride[ci], dst_stride[ci], p->cls[ci].segs[si], nplanes(ci), plane0(ci), a.c[0]);
                a.c[1] = a.c[0]; a.c[1].nblocks = 0;
            }
            a.scaler_blocks = a.c[0].nblocks + (cmerged ? a.c[1].nblocks : 0);
            a.grid_scaler = a.scaler_blocks; a.queue = nullptr;
            a.y16 = p->y16 && (cmerged || ci == 0);
            if (rng15) set_r15(a, cmerged, ci);
#ifdef TTS_EXPERIMENTS
            a.dbg = exp_dbg_buffer(a.scaler_blocks);
#endif
            if (fuse_range) a.range = range;
            if (cmerged) { pad_class(0, kBlock, a.pad[0]); pad_class(1, kBlock, a.pad[1]); }
            else {
                pad_class(ci, kBlock, a.pad[0]);
                if (fuse_range && ci) { a.range.mul[0] = range.mul[1]; a.range.add[0] = range.add[1]; a.range.shift[0] = range.shift[1]; }
            }
            const uint32_t blocks = a.scaler_blocks + a.pad[0].nblocks + a.pad[1].nblocks;
            if (const int rc = launch_checked("column kernel", fn, dim3(blocks), dim3(kBlock), 0, st, src, dst, p->d_tab, a)) return rc;
            if (cmerged) break;
        }
    }

    for (int ci = 0; ci < 2; ++ci) {
        const ClassPlan &c = p->cls[ci];
        if (!to_generic[ci] || !active(ci)) continue;
        GenericArgs a{};
        a.src = src; a.dst = dst;
        a.hpos = c.d_hpos; a.htaps = c.d_htaps; a.vpos = c.d_vpos; a.vtaps = c.d_vtaps;
        a.ht = c.h.size; a.vt = c.v.size;
        a.x0 = c.x0; a.y0 = c.y0; a.ow = c.ow; a.oh = c.oh;
        a.src_stride = src_stride[ci]; a.dst_stride = dst_stride[ci];
        a.planes = nplanes(ci); a.plane0 = plane0(ci);
        a.range15 = range_in_scale ? p->g.sc_range : 0;
        a.addr = addr;
        z_launches(a.planes, [&](int f0, int nf) {
            a.frame0 = f0;
            dim3 grid((c.ow + 63) / 64, (c.oh + 3) / 4, nf * a.planes);
            return launch_checked("generic_kernel", generic_kernel, grid, dim3(kBlock), 0, st, a);
        });
        if (lrc) return lrc;
    }
    if (range_at_end != TTS_RANGE_KEEP) {
        for (int ci = 0; ci < 2; ++ci) {
            if (!active(ci) || (fuse_range && !to_generic[ci])) continue;
            const ClassPlan &c = p->cls[ci];
            RangeArgs r{};
            r.dst = dst; r.w = c.canvas_w; r.h = c.canvas_h; r.stride = dst_stride[ci];
            r.planes = nplanes(ci); r.plane0 = plane0(ci);
            r.origin = uint32_t(c.pad_y) * uint32_t(dst_stride[ci]) + uint32_t(c.pad_x);
            r.addr = addr;
            tts::build_range_lut(p->g.out_range, ci != 0, r.lut);
            r.affine = tts::range_affine(p->g.out_range, ci != 0, r.mul, r.add, r.shift) ? 1 : 0;
            z_launches(r.planes, [&](int f0, int nf) {
                r.frame0 = f0;
                dim3 grid((c.canvas_w + 1023) / 1024, (c.canvas_h + 3) / 4, nf * r.planes);
                return launch_checked("range_kernel", range_kernel, grid, dim3(kBlock), 0, st, r);
            });
            if (lrc) return lrc;
        }
    }
    return TTS_OK;
}

}  // namespace
