// api_conv.hip -- the convolution state of a handle (ConvState: one stream for ohs_engine, S streams for ohs_batch):
// set_ir (convolution.rs:111-139), the ring of input spectra, and conv_launch, which picks the kernels that serve a call
// (one partition: k_conv_p1 / k_conv_p1_os; several: the time-parallel kernels, k_conv_tp_old after a per-path set_ir,
// k_conv_general as the fallback).
#include "api_internal.h"

using namespace ohs;
using namespace ohs_api;

namespace ohs_api {

void conv_free(ConvState &c)
{
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    for (int p = 0; p < 4; ++p) { if (c.d_H[p]) hipFree(c.d_H[p]); c.d_H[p] = nullptr; }
    if (c.d_hist) hipFree(c.d_hist);
    if (c.d_pre) hipFree(c.d_pre);
    c.d_pre = nullptr; c.pre_cnt = ~0ull; c.pre_n = 0;
    if (c.d_tails) hipFree(c.d_tails);
    if (c.d_tails_alt) hipFree(c.d_tails_alt);
    if (c.d_cd) hipFree(c.d_cd);
    if (c.d_chunk_tails) hipFree(c.d_chunk_tails);
    if (c.d_cdm) hipFree(c.d_cdm);
    if (c.d_W) hipFree(c.d_W);
    if (c.d_W1) hipFree(c.d_W1);
    if (c.d_merged) hipFree(c.d_merged);
    if (c.d_merged_alt) hipFree(c.d_merged_alt);
    if (c.d_last_in) hipFree(c.d_last_in);
    if (c.d_irt) hipFree(c.d_irt);
    if (c.d_cd_os) hipFree(c.d_cd_os);
    if (c.d_irl) hipFree(c.d_irl);
    if (c.d_xhist) hipFree(c.d_xhist);
    if (c.d_xhist_alt) hipFree(c.d_xhist_alt);
    if (c.d_lb_ring) hipFree(c.d_lb_ring);
    if (c.d_lb_cd) hipFree(c.d_lb_cd);
    if (c.d_lb_ab) hipFree(c.d_lb_ab);
    if (c.d_lb_cd_alt) hipFree(c.d_lb_cd_alt);
    if (c.d_xb_cd) hipFree(c.d_xb_cd);
    if (c.d_xb_ab) hipFree(c.d_xb_ab);
    c.d_xb_cd = nullptr; c.d_xb_ab = nullptr; c.xb_p2 = 0; c.xb_cd_gen = 0;
    if (c.d_ptail) hipFree(c.d_ptail);
    if (c.d_ptail_alt) hipFree(c.d_ptail_alt);
    c.d_lb_cd_alt = nullptr; c.lb_cd_alt_p2pad = 0; c.d_ptail = nullptr; c.d_ptail_alt = nullptr; c.pt_len = 0; c.pt_active = false;
    c.d_irl = nullptr; c.irl_len = 0; c.d_xhist = nullptr; c.d_xhist_alt = nullptr; c.xh_len = 0; c.xh_valid = 0;
    c.d_lb_ring = nullptr; c.lb_ring_cap = 0; c.lb_valid = 0; c.d_lb_cd = nullptr; c.d_lb_ab = nullptr; c.lb_p2pad = 0;
    c.lb_cd_valid = false; c.lb_tables_gen++; c.lb_lazy = false;
    c.d_irt = nullptr; c.d_cd_os = nullptr; c.cd_os_valid = false;
    c.d_merged = nullptr; c.d_merged_alt = nullptr; c.d_last_in = nullptr; c.tails_lazy = false; c.lazy_ok = false;
    c.d_cdm = nullptr; c.d_W = nullptr; c.d_W1 = nullptr;
    c.d_hist = nullptr; c.d_tails = nullptr; c.d_tails_alt = nullptr; c.d_cd = nullptr; c.d_chunk_tails = nullptr;
}

int conv_init(ConvState &c, size_t S, hipStream_t st)
{
    c.S = S;
    for (int p = 0; p < 4; ++p) {
        // default path: one all-zero partition (convolution.rs:46-63); FFT(0) == 0
        HIP_TRY(hipMalloc(&c.d_H[p], NF * sizeof(float2)));
        HIP_TRY(hipMemsetAsync(c.d_H[p], 0, NF * sizeof(float2), st));
        c.P[p] = 1;
        c.since[p] = 0;
    }
    c.cap = 1;
    c.cnt = 0;
    HIP_TRY(hipMalloc(&c.d_hist, S * (size_t)c.cap * NF * sizeof(float2)));
    HIP_TRY(hipMemsetAsync(c.d_hist, 0, S * (size_t)c.cap * NF * sizeof(float2), st));
    HIP_TRY(hipMalloc(&c.d_tails, S * 2 * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMemsetAsync(c.d_tails, 0, S * 2 * 8 * 64 * sizeof(float2), st));
    HIP_TRY(hipMalloc(&c.d_tails_alt, S * 2 * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c.d_cd, 2 * NF * sizeof(float2)));
    c.cd_valid = false;
    return OHS_OK;
}

// Batch objects let the P = 1 fast path leave its state behind lazily: a k_conv_p1 launch then ends with the merged
// overlap of its last block and a copy of that block's input instead of a k_conv_p1_state launch (9 us + a launch gap
// per call, hidden nowhere when the convolution runs alone); the next fast launch starts from the merged overlap -- which
// also makes a call boundary invisible in the output bits.  (Tuning::p1_lazy_state = 0: the per-path state after every launch.)
int conv_enable_lazy_state(ConvState &c)
{
    if (!tuning().p1_lazy_state) return OHS_OK;
    HIP_TRY(hipMalloc(&c.d_merged, c.S * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c.d_merged_alt, c.S * 8 * 64 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c.d_last_in, c.S * 2 * BS * sizeof(float)));
    HIP_TRY(hipMalloc(&c.d_irt, 4 * BS * sizeof(float)));
    HIP_TRY(hipMemset(c.d_irt, 0, 4 * BS * sizeof(float)));        // (default paths: one all-zero partition)
    HIP_TRY(hipMalloc(&c.d_cd_os, 2 * NF * sizeof(float4)));
    c.cd_os_valid = false;
    c.lazy_ok = true;
    c.tails_lazy = false;
    return OHS_OK;
}

// the per-path overlaps (d_tails) from the lazy state: whoever reads or edits d_tails calls this first -- a per-path
// set_ir (with the OLD spectra still in place: the overlaps belong to them), the general and the long-IR kernels
// The block-2048 plan leaves no spectra and no overlaps behind: its state is the streams' last input frames (d_xhist).  The
// block-512 state -- the ring's last Pmax spectra and the four per-path overlaps behind the last block -- is what the
// reference's engine would hold after those frames (convolution.rs:258,279-284): replay the last R = min(Pmax, cnt) blocks
// through the sequential kernel from zero overlaps, nothing stored (block t - i of the replay is visible to path p under
// the same `since` rule as ever; what lies further back than R blocks cannot reach the last block's overlap).
static int conv_materialise_from_xhist(ConvState &c, DeviceCtx *ctx, hipStream_t st)
{
    if (!c.lb_lazy) return OHS_OK;
    int Pmax = 1;
    for (int p = 0; p < 4; ++p) Pmax = std::max(Pmax, c.P[p]);
    const long long R = std::min<long long>(std::min<long long>(Pmax, (long long)c.cnt), c.xh_len / BS);
    if (R > 0) {
        int rc = conv_grow_ring(c, (int)R, st);
        if (rc) return rc;
        {   // the ring of input frames in time order
            hipError_t e = launch_conv_lb_xhist_linearise(c.d_xhist, c.d_xhist_alt, c.xh_len, c.xh_head, (int)c.S, st);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("input history: ") + hipGetErrorString(e));
        }
        ConvGeneralArgs a;
        a.in = c.d_xhist_alt + (c.xh_len - R * BS); a.out = c.d_xhist_alt;     // (nothing is stored: skip_store = R)
        a.in_stream_stride = 2 * c.xh_len; a.in_ch_stride = c.xh_len;
        a.out_stream_stride = 2 * c.xh_len; a.out_ch_stride = c.xh_len;
        a.n_blocks = (int)R;
        for (int p = 0; p < 4; ++p) { a.H[p] = c.d_H[p]; a.P[p] = c.P[p]; a.since[p] = c.since[p] - R; }
        a.hist = c.d_hist; a.cap = c.cap; a.cnt = c.cnt - (unsigned long long)R;
        a.tails = c.d_tails; a.tails_out = c.d_tails;
        a.skip_store = (int)R; a.zero_tails_in = 1;
        a.tw = ctx->d_tw; a.gain = 1.0f; a.fp_mode = c.fp_mode;
        hipError_t e = launch_conv_general(a, (int)c.S, st);
        if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("state replay launch: ") + hipGetErrorString(e));
    } else {
        HIP_TRY(hipMemsetAsync(c.d_tails, 0, c.S * 2 * 8 * 64 * sizeof(float2), st));
    }
    c.lb_lazy = false;
    return OHS_OK;
}

int conv_materialise_state(ConvState &c, DeviceCtx *ctx, hipStream_t st)
{
    if (c.lb_lazy || c.tails_lazy) c.pre_cnt = ~0ull;       // (the ring is about to be rewritten)
    if (c.lb_lazy) {
        const int rc = conv_materialise_from_xhist(c, ctx, st);
        if (rc) return rc;
    }
    if (!c.tails_lazy) return OHS_OK;
    ConvP1Args a;
    std::memset(&a, 0, sizeof(a));
    a.in = c.d_last_in; a.in_stream_stride = 2 * (long long)BS; a.in_ch_stride = (long long)BS;
    a.n_blocks = 1; a.n_streams = (int)c.S;
    for (int p = 0; p < 4; ++p) a.H[p] = c.d_H[p];
    a.tails_out = c.d_tails;
    a.tw = ctx->d_tw; a.fp_mode = c.fp_mode;
    a.xcd_lo = 0; a.xcd_n = 8;
    hipError_t e = launch_conv_p1_state(a, st);
    if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_p1_state launch: ") + hipGetErrorString(e));
    c.tails_lazy = false;
    return OHS_OK;
}

// grow the spectrum ring so that cap >= need, keeping the most recent old_cap blocks
int conv_grow_ring(ConvState &c, int need, hipStream_t st)
{
    if (need <= c.cap) return OHS_OK;
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    c.pre_cnt = ~0ull;
    const int ncap = next_pow2(need);
    float2 *nh = nullptr;
    HIP_TRY(hipMalloc(&nh, c.S * (size_t)ncap * NF * sizeof(float2)));
    HIP_TRY(hipMemsetAsync(nh, 0, c.S * (size_t)ncap * NF * sizeof(float2), st));
    for (int i = 0; i < c.cap; ++i) {
        if (c.cnt < (unsigned long long)(i + 1)) break;
        const unsigned long long u = c.cnt - 1 - (unsigned long long)i;
        const size_t so = (size_t)(u & (unsigned long long)(c.cap - 1));
        const size_t sn = (size_t)(u & (unsigned long long)(ncap - 1));
        // one strided copy over all streams
        HIP_TRY(hipMemcpy2DAsync(nh + sn * NF, (size_t)ncap * NF * sizeof(float2),
                                 c.d_hist + so * NF, (size_t)c.cap * NF * sizeof(float2),
                                 NF * sizeof(float2), c.S, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    hipFree(c.d_hist);
    c.d_hist = nh;
    c.cap = ncap;
    return OHS_OK;
}

// ---- block-2048 plan: the host side of its state (api_internal.h: ConvState) -----------------------------------------
static long long lb_p2pad_for(int Pmax) { return ((((long long)Pmax + 3) / 4) + 3) / 4 * 4; }    // 2048-tap partitions, multiple of 4

// the time-domain copies of the four responses (batch objects): [4][irl_len], zero-padded, grown as needed
static int conv_lb_irl_reserve(ConvState &c, long long need, hipStream_t st)
{
    if (need <= c.irl_len) return OHS_OK;
    DeviceWideSection dws;
    float *n = nullptr;
    HIP_TRY(hipMalloc(&n, 4 * (size_t)need * sizeof(float)));
    hipError_t e = hipMemsetAsync(n, 0, 4 * (size_t)need * sizeof(float), st);
    if (e == hipSuccess && c.d_irl)
        e = hipMemcpy2DAsync(n, (size_t)need * sizeof(float), c.d_irl, (size_t)c.irl_len * sizeof(float),
                             (size_t)c.irl_len * sizeof(float), 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { hipFree(n); return fail(OHS_ERR_HIP, std::string("impulse response copy: ") + hipGetErrorString(e)); }
    if (c.d_irl) hipFree(c.d_irl);
    c.d_irl = n; c.irl_len = need;
    c.lb_cd_valid = false; c.lb_tables_gen++;
    return OHS_OK;
}

int conv_lb_store_ir(ConvState &c, int path, const float *d_src, const float *h_src, size_t len, int P, hipStream_t st)
{
    {
        const int rcr = conv_lb_irl_reserve(c, lb_p2pad_for(P) * kLbBlock, st);
        if (rcr) return rcr;
    }
    if (!c.d_irl) return OHS_OK;
    float *dst = c.d_irl + (size_t)path * c.irl_len;
    HIP_TRY(hipMemsetAsync(dst, 0, (size_t)c.irl_len * sizeof(float), st));
    if (len && d_src) HIP_TRY(hipMemcpyAsync(dst, d_src, len * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (len && h_src) HIP_TRY(hipMemcpyAsync(dst, h_src, len * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    c.lb_cd_valid = false; c.lb_tables_gen++;
    return OHS_OK;
}

// after a set_ir has updated P[] and since[]: the input history exists exactly while some response is long enough for the
// block-2048 plan (the one-partition fast path never pays for it); its length follows the longest response
int conv_lb_after_set_ir(ConvState &c, hipStream_t st)
{
    if (!c.lazy_ok) return OHS_OK;
    int Pmax = 1;
    for (int p = 0; p < 4; ++p) Pmax = std::max(Pmax, c.P[p]);
    const long long want = Pmax >= lb_min_p() ? lb_p2pad_for(Pmax) * kLbBlock : 0;
    if (want != c.xh_len) {
        DeviceWideSection dws;
        HIP_TRY(hipStreamSynchronize(st));
        float *n = nullptr, *n2 = nullptr;
        if (want) {
            const size_t bytes = c.S * 2 * (size_t)want * sizeof(float);
            HIP_TRY(hipMalloc(&n, 2 * bytes));      // (a ring of twice the reach)
            if (hipMalloc(&n2, bytes) != hipSuccess) { hipFree(n); return fail(OHS_ERR_HIP, "hipMalloc(input history) failed"); }
            hipError_t e = hipMemsetAsync(n, 0, 2 * bytes, st);
            if (e == hipSuccess && c.d_xhist) {     // keep what is there: the newest frame ends at position `want` of the new ring
                const long long keep = std::min(want, c.xh_len);
                e = launch_conv_lb_xhist_linearise(c.d_xhist, c.d_xhist_alt, c.xh_len, c.xh_head, (int)c.S, st);
                if (e == hipSuccess)
                    e = hipMemcpy2DAsync(n + (want - keep), (size_t)(2 * want) * sizeof(float), c.d_xhist_alt + (c.xh_len - keep),
                                         (size_t)c.xh_len * sizeof(float), (size_t)keep * sizeof(float), c.S * 2,
                                         hipMemcpyDeviceToDevice, st);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { hipFree(n); hipFree(n2); return fail(OHS_ERR_HIP, std::string("input history: ") + hipGetErrorString(e)); }
        }
        if (c.d_xhist) hipFree(c.d_xhist);
        if (c.d_xhist_alt) hipFree(c.d_xhist_alt);
        c.d_xhist = n; c.d_xhist_alt = n2;
        c.xh_valid = std::min(c.xh_valid, want);
        c.xh_len = want;
        c.xh_head = want;           // (the kept frames end at position `want`: the next frame goes there)
        c.lb_valid = 0;         // (the ring's depth follows the partition count)
        c.lb_cd_valid = false; c.lb_tables_gen++;
    }
    if (c.d_xhist && c.since[0] == 0 && c.since[1] == 0 && c.since[2] == 0 && c.since[3] == 0) {
        // every path has forgotten its past (convolution.rs:135-137): so has the input history, and zeros are what every path
        // may see of it
        HIP_TRY(hipMemsetAsync(c.d_xhist, 0, c.S * 2 * (size_t)(2 * c.xh_len) * sizeof(float), st));
        c.xh_valid = c.xh_len;
        c.xh_head = 0;
        c.lb_valid = 0;
    }
    return OHS_OK;
}

// ohs_batch_set_conv_plan(0) for long responses: the block-2048 plan for every call of an object whose longest response has
// lb_min_p() partitions or more.  Its windows sit on the stream's absolute grid, so a call of any length -- the plugin's
// 1024-frame block included -- continues the ring the previous call left (one or two windows to transform, no warm-up), and
// it beats the block-512 kernels at every call length measured (profiles/r05_short_calls_*.jsonl: 64 streams x 16 384 taps,
// 2-block calls 26 vs 35 us even WITH a warm-up per call; 256-block calls 0.138 vs 0.25 ms).
bool conv_plan_auto_is_lb(size_t S, long long n_blocks, int Pmax)
{
    (void)S; (void)n_blocks;
    return Pmax >= lb_min_p();
}

// segment length of a block-2048 launch sequence, in blocks of 2048 (the ring holds a segment's windows: 32 KiB each)
static long long lb_segment_blocks(size_t S)
{
    long long seg = (long long)tuning().lb_seg_windows / (long long)std::max<size_t>(S, 1);
    seg = std::max<long long>(8, std::min<long long>(seg, 256));
    return seg / kLbTile * kLbTile;
}

// tables of the block-2048 plan: the spectra of A = h_lsl + j h_lsr and B = h_rsl + j h_rsr per 2048-tap partition (the forward
// kernel in tables mode, two "streams" of P2pad windows; kept in d_lb_ab), then (C, D).  `a` receives the fields every launch
// of the plan shares.
static int conv_lb_prepare(ConvState &c, DeviceCtx *ctx, hipStream_t st, ConvLbArgs &a)
{
    const int P2pad = (int)(c.xh_len / kLbBlock);
    if (!c.d_lb_cd || c.lb_p2pad != P2pad) {
        DeviceWideSection dws;
        HIP_TRY(hipStreamSynchronize(st));
        if (c.d_lb_cd) hipFree(c.d_lb_cd);
        if (c.d_lb_ab) hipFree(c.d_lb_ab);
        c.d_lb_cd = nullptr; c.d_lb_ab = nullptr; c.lb_p2pad = 0; c.lb_cd_valid = false; c.lb_tables_gen++;
        HIP_TRY(hipMalloc(&c.d_lb_cd, (size_t)P2pad * 2 * kLbBlock * sizeof(float4)));
        HIP_TRY(hipMalloc(&c.d_lb_ab, (size_t)P2pad * 2 * kLbBlock * sizeof(float4)));
        c.lb_p2pad = P2pad;
    }
    std::memset(&a, 0, sizeof(a));
    a.tw = ctx->d_tw; a.tw4096 = ctx->d_tw4096; a.fp_mode = c.fp_mode; a.P2pad = P2pad;
#ifdef OHS_EXPERIMENTS
    a.debug_skip = tuning().lb_skip;
#endif
    if (!c.lb_cd_valid) {
        if (c.irl_len < (long long)P2pad * kLbBlock) return fail(OHS_ERR_HIP, "block-2048 plan: impulse responses not stored");
        ConvLbArgs t = a;
        t.in = c.d_irl; t.in_stream_stride = 2 * c.irl_len; t.in_ch_stride = c.irl_len;
        t.out = nullptr; t.n_streams = 2; t.io_frame0 = 0; t.io_frames = c.irl_len;
        t.u0 = 0; t.n_fwd = P2pad; t.ring = c.d_lb_ab; t.ring_cap = P2pad; t.ring_base = 0; t.tables_mode = 1;
        t.fp_mode = 0;      // (tables are built in IEEE arithmetic, like k_ir_spectrum's)
        t.debug_skip = 0;
        hipError_t e = launch_conv_lb_forward(t, st);
        if (e == hipSuccess) e = launch_conv_lb_build_cd(c.d_lb_ab, P2pad, c.d_lb_cd, st);
        if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("block-2048 tables: ") + hipGetErrorString(e));
        c.lb_cd_valid = true;
    }
    return OHS_OK;
}

static int conv_lb_ring_reserve(ConvState &c, int cap_need, hipStream_t st)
{
    if (cap_need <= c.lb_ring_cap) return OHS_OK;
    DeviceWideSection dws;
    HIP_TRY(hipStreamSynchronize(st));
    if (c.d_lb_ring) hipFree(c.d_lb_ring);
    c.d_lb_ring = nullptr; c.lb_ring_cap = 0; c.lb_valid = 0;
    HIP_TRY(hipMalloc(&c.d_lb_ring, c.S * (size_t)cap_need * kLbBlock * sizeof(float4)));
    c.lb_ring_cap = cap_need;
    return OHS_OK;
}

// A per-path set_ir in mid-stream on an object that keeps the input history (convolution.rs:135-138: THAT path forgets its
// past, the others do not).  Instead of gating partitions per path in every kernel that follows, the reset is carried out
// as "every path forgets its past" plus what the frames in front of the reset still owe the output through each path --
// the tails: the input history followed by xh_len frames of silence through the plan's own two kernels, once with the
// L-speaker pair's tables alone (re = the lsl tail, im = the lsr tail) and once with the R-speaker pair's.  The reset path's
// row is dropped, rows still pending from an earlier reset are carried on, and conv_launch adds the rows to the output of
// the calls that follow, whatever plan serves them.  Returns taken = false when the history cannot vouch for every path's
// reach (then the block-512 kernels' per-path gates serve, as before).
static int conv_lb_tail_route(ConvState &c, DeviceCtx *ctx, int path, hipStream_t st, bool *taken)
{
    *taken = false;
    if (!c.lazy_ok || !c.d_xhist) return OHS_OK;
    if (c.since[0] == 0 && c.since[1] == 0 && c.since[2] == 0 && c.since[3] == 0) return OHS_OK;     // nothing processed since a full reset
    bool deep = true;
    for (int p = 0; p < 4; ++p) deep = deep && c.since[p] >= c.P[p] && c.xh_valid >= (long long)c.P[p] * BS;
    const bool together = c.since[0] == c.since[1] && c.since[1] == c.since[2] && c.since[2] == c.since[3] && c.xh_valid >= c.xh_len;
    if (!deep && !together) return OHS_OK;
    ConvLbArgs a;
    int rc = conv_lb_prepare(c, ctx, st, a);
    if (rc) return rc;
    const int P2pad = a.P2pad;
    const long long Lt = c.xh_len;
    DeviceWideSection dws;
    if (c.lb_cd_alt_p2pad != P2pad) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c.d_lb_cd_alt) hipFree(c.d_lb_cd_alt);
        c.d_lb_cd_alt = nullptr; c.lb_cd_alt_p2pad = 0;
        HIP_TRY(hipMalloc(&c.d_lb_cd_alt, (size_t)P2pad * 2 * kLbBlock * sizeof(float4)));
        c.lb_cd_alt_p2pad = P2pad;
    }
    if (c.pt_len != Lt) {       // (a response of another length since the last tails: rows of the old length are re-cut below)
        HIP_TRY(hipStreamSynchronize(st));
        float *n1 = nullptr, *n2 = nullptr;
        const size_t bytes = c.S * 4 * (size_t)Lt * sizeof(float);
        HIP_TRY(hipMalloc(&n1, bytes));
        if (hipMalloc(&n2, bytes) != hipSuccess) { hipFree(n1); return fail(OHS_ERR_HIP, "hipMalloc(pending tails) failed"); }
        hipError_t e = hipMemsetAsync(n1, 0, bytes, st);
        if (e == hipSuccess && c.pt_active && c.d_ptail) {
            // carry the rest of the pending rows over, from position 0 of the new buffer
            const long long rest = std::min(c.pt_len - c.pt_pos, Lt);
            if (rest > 0)
                e = hipMemcpy2DAsync(n1, (size_t)Lt * sizeof(float), c.d_ptail + c.pt_pos, (size_t)c.pt_len * sizeof(float),
                                     (size_t)rest * sizeof(float), c.S * 4, hipMemcpyDeviceToDevice, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { hipFree(n1); hipFree(n2); return fail(OHS_ERR_HIP, std::string("pending tails: ") + hipGetErrorString(e)); }
        if (c.d_ptail) hipFree(c.d_ptail);
        if (c.d_ptail_alt) hipFree(c.d_ptail_alt);
        c.d_ptail = n1; c.d_ptail_alt = n2; c.pt_len = Lt; c.pt_pos = 0;
    }
    rc = conv_lb_ring_reserve(c, 2 * P2pad - 1, st);
    if (rc) return rc;
    // windows -(P2pad - 1) .. P2pad - 1 of a segment that starts at the reset: the history, then silence (no buffer frames)
    a.n_streams = (int)c.S; a.gain = 1.0f;
    a.ring = c.d_lb_ring; a.ring_cap = c.lb_ring_cap; a.ring_base = 4ll * c.lb_ring_cap;
    a.in = c.d_xhist; a.io_frame0 = 0; a.io_frames = 0;         // (`in` is never read: no frame lies inside the buffer)
    a.xhist = c.d_xhist; a.xh_append = nullptr; a.xh_len = c.xh_len; a.xh_head = c.xh_head; a.xh_cnt = 0;
    a.u0 = -(P2pad - 1); a.n_fwd = 2 * P2pad - 1;
    a.debug_skip = 0;
    hipError_t e = launch_conv_lb_forward(a, st);
    const int keep = 0xF & ~(1 << path);
    for (int pair = 0; pair < 2 && e == hipSuccess; ++pair) {
        e = launch_conv_lb_build_cd(c.d_lb_ab, P2pad, c.d_lb_cd_alt, st, 1 << pair);
        if (e != hipSuccess) break;
        ConvLbArgs m = a;
        m.cd = c.d_lb_cd_alt; m.n_big = P2pad;
        m.out = c.d_xhist_alt; m.out_stream_stride = 2 * Lt; m.out_ch_stride = Lt;      // [S][2][Lt]: re -> ch 0, im -> ch 1
        m.io_frame0 = 0; m.io_frames = Lt;
        e = launch_conv_lb_mac_inverse(m, st);
        if (e == hipSuccess)
            e = launch_conv_lb_tails_merge(c.pt_active ? c.d_ptail : nullptr, c.pt_pos, c.d_xhist_alt, c.d_ptail_alt, Lt, pair, keep,
                                           (int)c.S, st);
    }
    if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("pending tails: ") + hipGetErrorString(e));
    std::swap(c.d_ptail, c.d_ptail_alt);
    c.pt_pos = 0; c.pt_active = true;
    // every path forgets its past: the history, the block-512 ring and overlaps, the window ring
    HIP_TRY(hipMemsetAsync(c.d_xhist, 0, c.S * 2 * (size_t)(2 * c.xh_len) * sizeof(float), st));
    c.xh_head = 0; c.xh_valid = c.xh_len; c.lb_valid = 0; c.lb_lazy = false; c.tails_lazy = false;
    HIP_TRY(hipMemsetAsync(c.d_hist, 0, c.S * (size_t)c.cap * NF * sizeof(float2), st));
    HIP_TRY(hipMemsetAsync(c.d_tails, 0, c.S * 2 * 8 * 64 * sizeof(float2), st));
    for (int p = 0; p < 4; ++p) c.since[p] = 0;
    HIP_TRY(hipStreamSynchronize(st));
    *taken = true;
    return OHS_OK;
}

static int conv_lb_launch(ConvState &c, DeviceCtx *ctx, const float *in, long long in_ss, long long in_cs, float *out,
                          long long out_ss, long long out_cs, int n_blocks, float gain, hipStream_t st)
{
    ConvLbArgs a;
    {
        const int rcp = conv_lb_prepare(c, ctx, st, a);
        if (rcp) return rcp;
    }
    const int P2pad = a.P2pad;
    const long long SEG = lb_segment_blocks(c.S);
    const long long frames_total = (long long)n_blocks * BS;
    const long long nbig_total = ((long long)c.cnt * BS + frames_total - 1) / kLbBlock - (long long)c.cnt * BS / kLbBlock + 1;
    const int cap_need = (int)(std::min(SEG, (nbig_total + kLbTile - 1) / kLbTile * kLbTile) + P2pad - 1);
    {
        const int rcr = conv_lb_ring_reserve(c, cap_need, st);
        if (rcr) return rcr;
    }
    a.n_streams = (int)c.S; a.gain = gain; a.cd = c.d_lb_cd;
    a.ring = c.d_lb_ring; a.ring_cap = c.lb_ring_cap;
    a.in_stream_stride = in_ss; a.in_ch_stride = in_cs; a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
    a.xh_len = c.xh_len;
    // The call's frames on the stream's absolute grid of 2048-frame blocks: [abs0, absE) touches blocks Ba .. Bl.  Segments of
    // at most SEG blocks; the first may start inside a block (its earlier frames come from the input history and its
    // output frames in front of the call are not stored), the last may end inside one (zeros behind the call; recomputed by
    // the next call).
    const long long abs0 = (long long)c.cnt * BS, absE = abs0 + frames_total;
    const long long Ba = abs0 / kLbBlock, Bl = (absE - 1) / kLbBlock;
    int tiles_last = 0, tile_blocks = kLbTile;
    for (long long Bs = Ba; Bs <= Bl; Bs += SEG) {
        const long long Be = std::min<long long>(Bs + SEG, Bl + 1);
        const int n_big = (int)(Be - Bs);
        const long long f_lo = std::max(abs0, Bs * kLbBlock), f_hi = std::min(absE, Be * kLbBlock);    // the call's frames in this segment
        // does the ring hold the complete windows in front of this segment?
        const bool cont = c.lb_end == Bs && c.lb_valid >= P2pad - 1;
        const int warm = cont ? 0 : P2pad - 1;
        a.in = in + (f_lo - abs0); a.out = out + (f_lo - abs0);
        a.io_frame0 = f_lo - Bs * kLbBlock; a.io_frames = f_hi - f_lo;
        // (the forward kernel appends the segment's last frames to the input history itself -- before the inverse kernel may
        // overwrite them in place --, into the half of the ring its windows do not read)
        const long long xcnt = std::min<long long>(a.io_frames, c.xh_len);
        a.xhist = c.d_xhist; a.xh_append = c.d_xhist; a.xh_head = c.xh_head; a.xh_cnt = xcnt;
        a.u0 = -warm; a.n_fwd = warm + n_big; a.n_big = n_big;
        a.ring_base = Bs + 4ll * c.lb_ring_cap;
        hipError_t e = launch_conv_lb_forward(a, st);
        if (e == hipSuccess) {
            c.xh_head = (c.xh_head + xcnt) % (2 * c.xh_len);
            e = launch_conv_lb_mac_inverse(a, st, &tile_blocks);
        }
        if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("block-2048 launch: ") + hipGetErrorString(e));
        const long long complete = f_hi / kLbBlock - Bs;          // blocks of this segment whose last frame is known
        c.lb_valid = std::min<long long>((cont ? c.lb_valid : warm) + complete, c.lb_ring_cap - 1);
        c.lb_end = Bs + complete;
        tiles_last = (n_big + tile_blocks - 1) / tile_blocks;
    }
    c.xh_valid = std::min(c.xh_len, c.xh_valid + frames_total);
    c.lb_lazy = true;
    c.last_kernel = OHS_CONV_KERNEL_BLOCK2048; c.last_ranges = tiles_last;
    return OHS_OK;
}

// Block 8192 / FFT 16384, one kernel (conv_xb_kernels.hip): long out-of-place calls on responses of at most two 8192-tap
// partitions.  It starts from the input history like the block-2048 plan and leaves the same state behind (the history
// advanced; the block-2048 window ring does not see the call's blocks).
// where the block-8192 kernel pays among the calls the large-block plans serve (out of place only: the caller checks)
bool conv_plan_auto_is_xb(size_t S, long long n_blocks, int Pmax)
{
    const Tuning &tn = tuning();
    const int P2x = (Pmax + 15) / 16;
    return P2x <= 2 && Pmax >= tn.xb_min_p && n_blocks >= tn.xb_min_blocks && (P2x == 1 || (long long)S >= tn.xb_min_streams_p2);
}

static int conv_xb_launch(ConvState &c, DeviceCtx *ctx, const float *in, long long in_ss, long long in_cs, float *out,
                          long long out_ss, long long out_cs, int n_blocks, int P2, float gain, hipStream_t st)
{
#ifdef OHS_EXPERIMENTS
    const int form = tuning().xb_form == 1 ? 1 : 2;     // (1: the sixteen-wave first form, experiments/conv_xb16_kernels.hip)
#else
    constexpr int form = 2;
#endif
    if (c.xb_form != form) { c.xb_cd_gen = 0; c.xb_form = form; }
    if (c.xb_p2 != P2 || !c.d_xb_cd) {
        DeviceWideSection dws;
        HIP_TRY(hipStreamSynchronize(st));
        if (c.d_xb_cd) hipFree(c.d_xb_cd);
        if (c.d_xb_ab) hipFree(c.d_xb_ab);
        c.d_xb_cd = nullptr; c.d_xb_ab = nullptr; c.xb_p2 = 0; c.xb_cd_gen = 0;
        HIP_TRY(hipMalloc(&c.d_xb_cd, (size_t)P2 * 2 * kXbBlock * sizeof(float4)));
        HIP_TRY(hipMalloc(&c.d_xb_ab, (size_t)2 * P2 * kXbBlock * sizeof(float4)));
        c.xb_p2 = P2;
    }
    auto launch = [&](const ConvXbArgs &x) -> hipError_t {
#ifdef OHS_EXPERIMENTS
        if (form == 1) return launch_conv_xb16(x, st);
#endif
        return launch_conv_xb(x, st);
    };
    ConvXbArgs a;
    std::memset(&a, 0, sizeof(a));
    a.tw = ctx->d_tw; a.tw16384 = ctx->d_tw16384; a.P2 = P2;
    if (c.xb_cd_gen != c.lb_tables_gen) {
        if (c.irl_len < (long long)P2 * kXbBlock) return fail(OHS_ERR_HIP, "block-8192 plan: impulse responses not stored");
        ConvXbArgs t = a;
        t.in = c.d_irl; t.in_stream_stride = 2 * c.irl_len; t.in_ch_stride = c.irl_len; t.io_frames = c.irl_len;
        t.n_streams = 2; t.n_blk = P2; t.run = 1; t.ab = c.d_xb_ab; t.tables_mode = 1;
        t.fp_mode = 0;      // (tables are built in IEEE arithmetic, like k_ir_spectrum's)
        hipError_t e = launch(t);
#ifdef OHS_EXPERIMENTS
        if (e == hipSuccess && form == 1) e = launch_conv_xb16_build_cd(c.d_xb_ab, P2, c.d_xb_cd, st);
        else
#endif
        if (e == hipSuccess) e = launch_conv_xb_build_cd(reinterpret_cast<const float2 *>(c.d_xb_ab), P2, c.d_xb_cd, st);
        if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("block-8192 tables: ") + hipGetErrorString(e));
        c.xb_cd_gen = c.lb_tables_gen;
    }
    const long long frames = (long long)n_blocks * BS;
    a.in = in; a.out = out;
    a.in_stream_stride = in_ss; a.in_ch_stride = in_cs; a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
    a.xhist = c.d_xhist; a.xh_len = c.xh_len; a.xh_head = c.xh_head;
    a.io_frames = frames; a.n_streams = (int)c.S; a.n_blk = (int)((frames + kXbBlock - 1) / kXbBlock);
    a.run = conv_xb_run_for(a.n_streams, a.n_blk, P2, ctx->num_cus);
    a.cd = c.d_xb_cd; a.gain = gain; a.fp_mode = c.fp_mode; a.stagger = tuning().xb_stagger;
    // the call's last frames join the input history: appended by the kernel itself, into the half of the ring it does not read
    const long long xcnt = std::min(frames, c.xh_len);
    bool own_append = form == 2;
#ifdef OHS_EXPERIMENTS
    if (tuning().xb_append_launch) own_append = false;     // (A/B: the append as a launch of its own behind the kernel)
#endif
    if (own_append) { a.xh_append = c.d_xhist; a.xh_cnt = xcnt; }
#ifdef OHS_EXPERIMENTS
    if (tuning().xb_run > 0) a.run = std::min(tuning().xb_run, a.n_blk);
    a.debug_skip = tuning().xb_skip;
#endif
    hipError_t e = launch(a);
    if (e == hipSuccess && !own_append)     // (the first form does not append: a launch behind it)
        e = launch_conv_lb_xhist_append(in, in_ss, in_cs, c.d_xhist, c.xh_len, c.xh_head, frames, (int)c.S, st);
    if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("block-8192 launch: ") + hipGetErrorString(e));
    c.xh_head = (c.xh_head + xcnt) % (2 * c.xh_len);
    c.xh_valid = std::min(c.xh_len, c.xh_valid + frames);
    c.lb_valid = 0;             // (the block-2048 window ring has not seen these blocks)
    c.lb_lazy = true;
    c.last_kernel = OHS_CONV_KERNEL_BLOCK8192; c.last_ranges = (a.n_blk + a.run - 1) / a.run;
    return OHS_OK;
}

// set_ir (convolution.rs:111-139) for one path; `ir` is a host pointer, or a device pointer on the state's
// device when ir_on_device (the node-level batch hands over what RCCL has just broadcast)
int conv_set_ir(ConvState &c, DeviceCtx *ctx, int path, const float *ir, size_t len, hipStream_t st,
                bool ir_on_device)
{
    if (path < 0 || path > 3) return fail(OHS_ERR_INVALID_ARG, "path must be 0..3");
    if (len && !ir) return fail(OHS_ERR_INVALID_ARG, "ir is NULL");
    if (len > (size_t)1 << 30) return fail(OHS_ERR_INVALID_ARG, "ir too long");
    const int P = len == 0 ? 1 : (int)((len + BS - 1) / BS);
    DeviceWideSection dws;      // (frees / device-wide waits below: no resident kernel may be in their way)
    c.pre_cnt = ~0ull;          // sums computed ahead belong to the old response and the old `since`
    if (c.lazy_ok && !c.d_xhist && c.tails_lazy && c.d_last_in) {
        // The first response of several partitions on an object that has run the one-partition fast path so far: no input
        // history was kept, but one-partition paths reach back 511 frames only, and the fast path's state at rest holds
        // the last block's 512 input frames.  Seeded with those, the history vouches for every path's reach, and the
        // reset below can carry the other paths' past as pending tails.
        int Pmax_after = P;
        for (int p = 0; p < 4; ++p)
            if (p != path) Pmax_after = std::max(Pmax_after, c.P[p]);
        bool all_one = true;
        for (int p = 0; p < 4; ++p) all_one = all_one && c.P[p] == 1;
        if (all_one && Pmax_after >= lb_min_p()) {
            const long long want = lb_p2pad_for(Pmax_after) * kLbBlock;
            const size_t bytes = c.S * 2 * (size_t)want * sizeof(float);
            float *n = nullptr, *n2 = nullptr;
            HIP_TRY(hipMalloc(&n, 2 * bytes));
            if (hipMalloc(&n2, bytes) != hipSuccess) { hipFree(n); return fail(OHS_ERR_HIP, "hipMalloc(input history) failed"); }
            hipError_t e = hipMemsetAsync(n, 0, 2 * bytes, st);
            if (e == hipSuccess)        // frames -512 .. -1 at ring positions 0 .. 511, the next frame goes to 512
                e = hipMemcpy2DAsync(n, (size_t)(2 * want) * sizeof(float), c.d_last_in, (size_t)BS * sizeof(float),
                                     (size_t)BS * sizeof(float), c.S * 2, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { hipFree(n); hipFree(n2); return fail(OHS_ERR_HIP, std::string("input history: ") + hipGetErrorString(e)); }
            c.d_xhist = n; c.d_xhist_alt = n2; c.xh_len = want; c.xh_head = BS; c.xh_valid = BS;
            c.lb_valid = 0; c.lb_cd_valid = false; c.lb_tables_gen++;
            const int rcr = conv_lb_irl_reserve(c, want, st);       // (the tails' tables span the new reach)
            if (rcr) return rcr;
        }
    }
    {   // an object that keeps the input history turns a reset in mid-stream into "all paths forget" + pending tails
        bool taken = false;
        const int rct = conv_lb_tail_route(c, ctx, path, st, &taken);
        if (rct) return rct;
        if (!taken && c.pt_active) {
            // tails still pending from an earlier reset: this path's row is part of the past it now forgets (e.g. the
            // second of four set_ir calls that re-load every path: nothing has been processed since the first)
            HIP_TRY(hipMemset2DAsync(c.d_ptail + (size_t)path * c.pt_len, 4 * (size_t)c.pt_len * sizeof(float), 0,
                                     (size_t)c.pt_len * sizeof(float), c.S, st));
        }
    }
    {   // the overlaps at rest belong to the spectra that are about to be replaced
        const int rcm = conv_materialise_state(c, ctx, st);
        if (rcm) return rcm;
    }
    // every failure path below frees what it allocated: a HIP_TRY inside the lambda returns from the
    // lambda only, the buffers are released after it
    float2 *nH = nullptr;
    float *d_ir = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&nH, (size_t)P * NF * sizeof(float2)));
        if (len == 0) {
            HIP_TRY(hipMemsetAsync(nH, 0, NF * sizeof(float2), st));   // :114-118 mute
        } else {
            const float *src = ir;
            if (!ir_on_device) {
                HIP_TRY(hipMalloc(&d_ir, len * sizeof(float)));
                HIP_TRY(hipMemcpyAsync(d_ir, ir, len * sizeof(float), hipMemcpyHostToDevice, st));
                src = d_ir;
            }
            HIP_TRY(launch_ir_spectrum(src, (int)len, P, nH, ctx->d_tw, st));
            if (c.d_irt && P == 1) {        // the overlap-save plan builds its 2048-point tables from the taps themselves
                HIP_TRY(hipMemsetAsync(c.d_irt + path * BS, 0, BS * sizeof(float), st));
                HIP_TRY(hipMemcpyAsync(c.d_irt + path * BS, src, len * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
        }
        if (c.d_irt && len == 0) HIP_TRY(hipMemsetAsync(c.d_irt + path * BS, 0, BS * sizeof(float), st));
        int rc = conv_grow_ring(c, P, st);
        if (rc) return rc;
        // one wait: the spectrum kernel has read d_ir, and nothing queued earlier on `st` still reads the
        // old d_H[path]
        HIP_TRY(hipStreamSynchronize(st));
        return OHS_OK;
    };
    const int rc0 = body();
    if (d_ir) hipFree(d_ir);
    if (rc0) { if (nH) hipFree(nH); return rc0; }
    if (c.lazy_ok) {
        const int rcl = conv_lb_store_ir(c, path, ir_on_device ? ir : nullptr, ir_on_device ? nullptr : ir, len, P, st);
        if (rcl) { hipFree(nH); return rcl; }
    }
    hipFree(c.d_H[path]);
    c.d_H[path] = nH;
    c.P[path] = P;
    c.spk_set[path] = false;
    c.cd_os_valid = false;
    c.cd_valid = false;
    c.cdm_valid = false;
    c.since[path] = 0;      // :135-137 that path's history is "all zero" again
    if (c.since[0] == 0 && c.since[1] == 0 && c.since[2] == 0 && c.since[3] == 0) {
        // every path has forgotten its past: the ring holds nothing any path may still see.  Zeroing
        // it lets the gate-free time-parallel kernels run right away (all `since` equal).
        HIP_TRY(hipMemsetAsync(c.d_hist, 0, c.S * (size_t)c.cap * NF * sizeof(float2), st));
    }
    {
        const int rcx = conv_lb_after_set_ir(c, st);
        if (rcx) return rcx;
    }
    // :138 that path's overlap := 0.  Tails are (lsl + j lsr) and (rsl + j rsr).
    const int which = path >> 1, comp = path & 1;
    HIP_TRY(launch_zero_tail_component(c.d_tails, (long long)c.S, which, comp, st));
    HIP_TRY(hipStreamSynchronize(st));
    return OHS_OK;
}

// Chunks per stream of the P = 1 kernel: at least ONE round of resident waves (CUs x 16), and a wave count S * K that
// FILLS its last round -- all waves of a launch take the same time, so 2.13 rounds cost 3 (1024 and 2048 streams with
// K = ceil(2 R / S) ran at 71 % for that reason).  K is the smallest value from ceil(R / S) upwards whose last round is
// >= 97 % full.  (Rounds 1 and 2 asked for two rounds; since the chunk lengths follow the waves' age ranks one exact
// round is the faster plan -- config 3 without EQ: 4096 waves 0.537-0.541 ms, 8192 waves 0.549-0.555, 3840 0.563,
// 4352 0.81 -- and the pre-pass has half the boundary tails to compute.  Tuning::p1_rounds = 2 restores the old rule.)
long long conv_p1_chunks(const DeviceCtx *ctx, size_t S, long long n_blocks, long long target_override, int xcd_n)
{
    const long long R = (long long)ctx->num_cus * xcd_n / 8 * conv_p1_waves_per_cu();
    long long K;
    if (target_override > 0) {
        K = (target_override + (long long)S - 1) / (long long)S;
    } else {
        const long long rounds = tuning().p1_rounds;
        const long long k0 = std::max<long long>(1, (rounds * R + (long long)S - 1) / (long long)S);
        K = k0;
        double best = 0.0;
        for (long long k = k0; k < k0 + 32; ++k) {
            const double rd = (double)((long long)S * k) / (double)R;
            const double eff = rd / std::ceil(rd - 1e-9);
            if (eff > best + 1e-9) { best = eff; K = k; }
            if (eff >= 0.97) break;
        }
    }
    // One chunk per stream (4096 streams and more) leaves the age ranks nothing to balance: two chunks, dealt out over
    // ranks {0, 3} / {1, 2} (conv_kernels.hip: p1_wave_job), cost a second round but end together -- 4096 streams: 8.9 ->
    // 8.0 ms of convolution per step.
    if (K == 1 && n_blocks >= 16 && target_override <= 0 && conv_p1_waves_per_cu() == 16) K = 2;
    if (n_blocks < 4 * K && target_override <= 0) {
        // A short launch (the last, 2 % time chunk of an overlapped batch step: 19 blocks), a few blocks per wave at
        // best: every chunk boundary costs the pre-pass a block's work, and a wave count just above a round starts a second,
        // almost empty one.  Cost in block times: rounds x blocks per wave for the main kernel + rounds for the boundary
        // tails (one more block per wave where the chunks' own waves compute them -- K = 2, 4, 8, 16 --, else a pre-pass
        // launch: its rounds + about a block time of launch); the smallest K of minimal cost (19 blocks x 256 streams:
        // K = 16, one round of at most 2 + 1 blocks, no pre-pass).
        const bool own_ok = conv_p1_waves_per_cu() == 16;
        long long best_k = std::min(K, n_blocks), best_cost = -1;
        for (long long k = 1; k <= std::min(K, n_blocks); ++k) {
            const long long rounds_main = ((long long)S * k + R - 1) / R, per_wave = (n_blocks + k - 1) / k;
            long long cost;
            if (k == 1) cost = rounds_main * per_wave;
            else if (own_ok && (k == 2 || k == 4 || k == 8 || k == 16)) cost = rounds_main * (per_wave + 1);
            else cost = rounds_main * per_wave + ((long long)S * (k - 1) + R - 1) / R + 1;
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_k = k; }
        }
        K = best_k;
    }
    if (K > n_blocks) K = n_blocks;
    if (K < 1) K = 1;
    return K;
}

// ohs_batch_set_conv_plan(0): which P = 1 plan the library picks by itself
// The hop-1536 plan does a quarter less work per frame at three waves per SIMD and wins wherever the convolution's time
// is exposed and the launch can fill the chip (profiles/r04_conv_plan_ab_run3.txt, _run4.txt):
//   out of place (the call's hop ranges are independent: any count per stream): 0.944 x the block-512 plan's time at 256
//     streams, 0.87-0.89 x from 384 on; a tie at 128 and below
//   in place (a stream's ranges share a workgroup: 1, 2, 3, 4, 6 or 12 of them): 0.93 x at 512 streams, 0.87-0.92 x beyond;
//     a tie at 384.  The batch's EQ || convolution path is in place: at 256 streams with the EQ on -- BASELINE configs[2] --
//     the convolution hides under the EQ's serial chain either way (1.000-1.004 x), and the block-512 plan keeps its two
//     properties there: output bits that do not depend on where the calls cut the signal, and the recorded golden hash.
// Short calls stay with the block-512 plan as well (a wave's first and last hops run the guarded form of the hop).
bool conv_plan_auto_is_os(size_t S, long long n_blocks, bool in_place)
{
    return n_blocks >= 48 && S >= (in_place ? 512u : 256u);
}

// hop ranges per stream of the hop-1536 plan: rounds of resident waves x hops per wave (+ 1: the tables and the first
// window), the smallest count of minimal cost.  In place all of a stream's ranges must share a workgroup
// (launch_conv_p1_os): a divisor of its wave count; out of place any count goes, and few streams still fill the chip
// (64 streams x 48 ranges).
int conv_os_chunks(const DeviceCtx *ctx, size_t S, long long n_blocks, bool in_place)
{
    const long long R = (long long)ctx->num_cus * conv_os_waves_per_cu();
    const long long H = (n_blocks * BS + 1535) / 1536;
    int K = 1;
    long long best = -1;
    for (int k = 1; k <= (in_place ? conv_os_waves_per_cu() : 96) && k <= H; ++k) {
        if (in_place && conv_os_waves_per_cu() % k) continue;
        const long long cost = (((long long)S * k + R - 1) / R) * ((H + k - 1) / k + 1);
        if (best < 0 || cost < best) { best = cost; K = k; }
    }
    return K;
}

static int conv_launch_impl(ConvState &c, DeviceCtx *ctx, const float *in, long long in_ss, long long in_cs,
                            float *out, long long out_ss, long long out_cs, int n_blocks, float gain, hipStream_t st,
                            bool allow_fast, hipEvent_t ev_start, hipEvent_t ev_stop);

int conv_launch(ConvState &c, DeviceCtx *ctx, const float *in, long long in_ss, long long in_cs,
                float *out, long long out_ss, long long out_cs, int n_blocks, float gain, hipStream_t st,
                bool allow_fast, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const int rc = conv_launch_impl(c, ctx, in, in_ss, in_cs, out, out_ss, out_cs, n_blocks, gain, st, allow_fast, ev_start, ev_stop);
    if (rc == OHS_OK && n_blocks > 0 && c.last_kernel >= 0 && c.last_kernel < 8) c.kernel_calls[c.last_kernel]++;
    if (rc == OHS_OK && n_blocks > 0 && c.pt_active) {
        // what the frames in front of a per-path set_ir still owe these frames (conv_lb_tail_route)
        const long long n = (long long)n_blocks * BS;
        hipError_t e = launch_conv_lb_tails_add(out, out_ss, out_cs, n, c.d_ptail, c.pt_len, c.pt_pos, gain, (int)c.S, st);
        if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("pending tails: ") + hipGetErrorString(e));
        c.pt_pos += n;
        if (c.pt_pos >= c.pt_len) c.pt_active = false;
    }
    return rc;
}

static int conv_launch_impl(ConvState &c, DeviceCtx *ctx, const float *in, long long in_ss, long long in_cs,
                            float *out, long long out_ss, long long out_cs, int n_blocks, float gain, hipStream_t st,
                            bool allow_fast, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    // (ev_start / ev_stop: recorded at the start / completion of the call's launches)
    struct EvScope {        // the paths that do not carry the events in a dispatch: recorded around their launches
        hipStream_t st; hipEvent_t stop; bool armed;
        ~EvScope() { if (armed && stop) hipEventRecord(stop, st); }
    } ev_scope{st, ev_stop, false};
    if (n_blocks <= 0) {                    // nothing to do (the pre-pass would index block -1)
        if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));
        if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, st));
        return OHS_OK;
    }
    const bool p1 = c.P[0] == 1 && c.P[1] == 1 && c.P[2] == 1 && c.P[3] == 1;
    if (p1 && allow_fast) {
        // every path has one partition: time-parallel single-FFT-pair kernel
        if (!c.cd_valid) {
            hipError_t e = launch_build_cd(c.d_H[0], c.d_H[1], c.d_H[2], c.d_H[3], c.d_cd, st);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("build_cd: ") + hipGetErrorString(e));
            c.cd_valid = true;
        }
        const Tuning &tn = tuning();
        // Two plans for this case (ohs_batch_set_conv_plan): block 512 / FFT 1024 overlap-add (k_conv_p1: the plan whose
        // output bits tests/golden/p1_output_sha256.txt holds) and hop 1536 / FFT 2048 overlap-save (k_conv_p1_os: a third
        // less transform and product work per frame; batch objects only -- it hands the lazy state over).
        const bool os_ok = c.lazy_ok && c.d_irt && n_blocks >= 3 && c.xcd_n == 8 && tn.p1_xcd_n == 0 &&
                           ((in_ss | in_cs | out_ss | out_cs) & 1) == 0 &&
                           ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 7) == 0 &&
                           (unsigned long long)n_blocks * BS + 1536ull < (1ull << 31);
        const bool use_os = os_ok && (c.conv_plan == 2 || (c.conv_plan == 0 && conv_plan_auto_is_os(c.S, n_blocks, in == out)));
        if (use_os) {
            if (!c.cd_os_valid) {
                hipError_t e = launch_os_tables(c.d_irt, ctx->d_tw, ctx->d_tw2048, c.d_cd_os, st);
                if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("os tables: ") + hipGetErrorString(e));
                c.cd_os_valid = true;
            }
            const int K = conv_os_chunks(ctx, c.S, n_blocks, in == out);
            ConvOsArgs o;
            o.in = in; o.out = out;
            o.in_stream_stride = in_ss; o.in_ch_stride = in_cs; o.out_stream_stride = out_ss; o.out_ch_stride = out_cs;
            o.n_blocks = n_blocks; o.n_streams = (int)c.S; o.chunks = K;
            o.CD = c.d_cd_os; o.tw2048 = ctx->d_tw2048;
            o.tails = c.d_tails; o.merged_in = c.tails_lazy ? c.d_merged : nullptr;
            o.merged_out = c.d_merged_alt; o.last_in = c.d_last_in;
            o.tw = ctx->d_tw; o.gain = gain; o.fp_mode = c.fp_mode;
            hipError_t e = launch_conv_p1_os(o, st, ev_start, ev_stop);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_p1_os launch: ") + hipGetErrorString(e));
            std::swap(c.d_merged, c.d_merged_alt);
            c.tails_lazy = true;
            c.cnt += (unsigned long long)n_blocks;
            for (int p = 0; p < 4; ++p) c.since[p] += n_blocks;
            c.last_kernel = OHS_CONV_KERNEL_HOP1536_P1; c.last_ranges = K;
            return OHS_OK;
        }
        const long long target_override = tn.p1_target_waves;
        int xcd_lo = c.xcd_lo, xcd_n = c.xcd_n;
        if (tn.p1_xcd_n > 0) { xcd_lo = tn.p1_xcd_lo; xcd_n = tn.p1_xcd_n; }
        const long long K = conv_p1_chunks(ctx, c.S, n_blocks, target_override, xcd_n);
        if ((size_t)K > c.chunk_tails_cap) {
            DeviceWideSection dws;
            HIP_TRY(hipStreamSynchronize(st));
            if (c.d_chunk_tails) hipFree(c.d_chunk_tails);
            c.d_chunk_tails = nullptr; c.chunk_tails_cap = 0;
            HIP_TRY(hipMalloc(&c.d_chunk_tails, c.S * (size_t)K * 512 * sizeof(float2)));
            c.chunk_tails_cap = (size_t)K;
        }
        ConvP1Args a;
        a.in = in; a.out = out;
        a.in_stream_stride = in_ss; a.in_ch_stride = in_cs;
        a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
        a.n_blocks = n_blocks; a.n_streams = (int)c.S; a.chunks = (int)K;
        a.CD = c.d_cd;
        for (int p = 0; p < 4; ++p) a.H[p] = c.d_H[p];
        a.tails = c.d_tails; a.tails_out = c.d_tails_alt; a.chunk_tails = c.d_chunk_tails;
        a.merged_in = c.tails_lazy ? c.d_merged : nullptr;
        a.merged_out = c.lazy_ok ? c.d_merged_alt : nullptr;
        a.last_in = c.lazy_ok ? c.d_last_in : nullptr;
        a.tw = ctx->d_tw; a.gain = gain; a.fp_mode = c.fp_mode;
        a.xcd_lo = xcd_lo; a.xcd_n = xcd_n;
        {   // boundary tails by the chunks' own waves where a stream's chunks share a workgroup (else: the pre-pass)
            const bool allowed = conv_p1_waves_per_cu() == 16 && (K == 2 || K == 4 || K == 8 || K == 16);
            a.own_tails = (allowed && tn.p1_own_tails) ? 1 : 0;
        }
        a.stagger = tn.p1_stagger;
        a.prio_mode = tn.p1_prio;
        // Chunk lengths by the age rank of the wave that runs a chunk (p1_chunk_begin), equal lengths when chunks are
        // short.  In the plans that do not spread a stream's chunks over the ranks, ALL of a stream's chunks can sit on
        // heavy ranks: its shortest chunk is then n * wmin / (K * wmax) blocks, and that one must keep >= 2 blocks (a
        // chunk of zero blocks would make the boundary-tail code index block b0 - 1 = -1).
        const int *kw = tn.p1_weights;
        const int wmin = std::min(std::min(kw[0], kw[1]), std::min(kw[2], kw[3]));
        const int wmax = std::max(std::max(kw[0], kw[1]), std::max(kw[2], kw[3]));
        const bool weighted = (long long)n_blocks * wmin >= 2ll * K * wmax;
        for (int g = 0; g < 4; ++g) a.weights[g] = weighted ? kw[g] : 1;
        hipError_t e = launch_conv_p1(a, st, ev_start, ev_stop);
        if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_p1 launch: ") + hipGetErrorString(e));
        if (c.lazy_ok) {
            std::swap(c.d_merged, c.d_merged_alt);
            c.tails_lazy = true;
        } else {
            std::swap(c.d_tails, c.d_tails_alt);
        }
        c.last_kernel = OHS_CONV_KERNEL_BLOCK512_P1; c.last_ranges = (int)K;
    } else {
        if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));
        ev_scope.armed = true;
        int Pmax = 1;
        for (int p = 0; p < 4; ++p) Pmax = std::max(Pmax, c.P[p]);
        if (allow_fast && c.d_xhist && Pmax >= lb_min_p()) {
            // Block 2048 / FFT 4096 (conv_lb_kernels.hip).  It convolves the input history itself, so every path must be
            // allowed to see all of its own response's reach of it: older than its response (and the history that deep),
            // or reset together with all the others (the history was zeroed then).
            bool deep = true;
            for (int p = 0; p < 4; ++p) deep = deep && c.since[p] >= c.P[p] && c.xh_valid >= (long long)c.P[p] * BS;
            const bool together = c.since[0] == c.since[1] && c.since[1] == c.since[2] && c.since[2] == c.since[3] &&
                                  c.xh_valid >= c.xh_len;
            const bool lb_ok = (deep || together) && ((in_ss | in_cs | out_ss | out_cs) & 3) == 0 &&
                               ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
            if (lb_ok && (c.conv_plan == 2 || (c.conv_plan == 0 && conv_plan_auto_is_lb(c.S, n_blocks, Pmax)))) {
                // Long out-of-place calls on responses of at most two 8192-tap partitions: block 8192 / FFT 16384, one kernel (the
                // buffers must not overlap at all: a run's first windows read frames in front of its own blocks).  Where it pays
                // (profiles/r05_xb_vs_lb_shapes.txt): calls of 128 blocks or more; with two partitions, on 32 streams or more.
                const int P2x = (Pmax + 15) / 16;
                const long long call_frames = (long long)n_blocks * BS;
                const float *in_end = in + (c.S - 1) * in_ss + in_cs + call_frames;
                const float *out_end = out + (c.S - 1) * out_ss + out_cs + call_frames;
                const bool apart = in_end <= out || out_end <= in;
                const bool xb = apart && conv_plan_auto_is_xb(c.S, n_blocks, Pmax) &&
                                c.xh_len >= (long long)P2x * kXbBlock && call_frames < (1ll << 29);
                const int rcl = xb ? conv_xb_launch(c, ctx, in, in_ss, in_cs, out, out_ss, out_cs, n_blocks, P2x, gain, st)
                                   : conv_lb_launch(c, ctx, in, in_ss, in_cs, out, out_ss, out_cs, n_blocks, gain, st);
                if (rcl) return rcl;
                c.cnt += (unsigned long long)n_blocks;
                for (int p = 0; p < 4; ++p) c.since[p] += n_blocks;
                return OHS_OK;
            }
        }
        {   // these kernels read and write the per-path overlaps
            const int rcm = conv_materialise_state(c, ctx, st);
            if (rcm) return rcm;
        }
        c.lb_valid = 0;         // (the block-2048 ring does not see the blocks of this call)
        // The call's frames join the input history of the block-2048 plan (what it starts from when a later call takes it): the
        // time-parallel forward kernel appends them itself, the sequential kernel's calls get a launch for it -- in front of
        // anything that overwrites them in place.
        const long long call_frames = (long long)n_blocks * BS;
        const bool xh_on = c.d_xhist != nullptr;
        const bool xh_ok = xh_on && ((in_ss | in_cs) & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
        const long long xh_cnt = xh_on ? std::min(call_frames, c.xh_len) : 0;
        const long long xh_head0 = c.xh_head;
        bool xh_lost = false;
        auto xh_append_launch = [&]() -> int {
            if (!xh_on) return OHS_OK;
            if (!xh_ok) { xh_lost = true; return OHS_OK; }      // (buffers the 16-byte copy cannot read: the history no longer follows the stream)
            hipError_t e = launch_conv_lb_xhist_append(in, in_ss, in_cs, c.d_xhist, c.xh_len, xh_head0, call_frames, (int)c.S, st);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("input history update: ") + hipGetErrorString(e));
            return OHS_OK;
        };
        auto xh_advance = [&]() {
            if (!xh_on) return;
            if (xh_lost) { c.xh_valid = 0; return; }
            c.xh_head = (xh_head0 + xh_cnt) % (2 * c.xh_len);
            c.xh_valid = std::min(c.xh_len, c.xh_valid + call_frames);
        };
        const bool since_equal = c.since[0] == c.since[1] && c.since[1] == c.since[2] && c.since[2] == c.since[3];
        bool since_old = true;
        for (int p = 0; p < 4; ++p) since_old = since_old && c.since[p] >= c.P[p] - 1;
        // gate-free condition of the time-parallel kernels: no path may have blocks in the ring that
        // it must not see (either every path is older than its own IR -- path p reads P[p] - 1 blocks back --, or all
        // were reset together: conv_set_ir zeroed the ring then)
        const bool gate_free = since_old || since_equal;
        // Otherwise -- a per-path set_ir in mid-stream -- the same three kernels run on a ring whose old blocks they
        // cannot reach (the call starts Ppad - 1 zeroed slots further on) and k_conv_tp_old adds what each path may
        // still see of them to the first Pmax - 1 blocks.  The first segment must reach beyond every old block
        // (n_blocks >= Pmax): the state it leaves behind is computed gate-free.  Shorter calls take the general kernel
        // (one wave per stream, blocks in sequence: 290 x slower on BASELINE config 4, which is why long calls do not).
        // (calls of 1 .. 3 blocks too: until round 5 they went through the sequential kernel -- one wave per stream, 93 us per
        // block on BASELINE config 4's shape, whatever the number of streams: profiles/r05_short_calls_before.jsonl)
        const bool tp_gated = allow_fast && !gate_free && n_blocks >= Pmax;
        const bool tp = allow_fast && (gate_free || tp_gated);
        auto general = [&](const float *gin, float *gout, int nb, unsigned long long cnt, int since_add,
                           int skip_store, int zero_in, float2 *tails_out) -> int {
            ConvGeneralArgs a;
            a.in = gin; a.out = gout;
            a.in_stream_stride = in_ss; a.in_ch_stride = in_cs;
            a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
            a.n_blocks = nb;
            for (int p = 0; p < 4; ++p) { a.H[p] = c.d_H[p]; a.P[p] = c.P[p]; a.since[p] = c.since[p] + since_add; }
            a.hist = c.d_hist; a.cap = c.cap; a.cnt = cnt;
            a.tails = c.d_tails; a.tails_out = tails_out;
            a.skip_store = skip_store; a.zero_tails_in = zero_in;
            a.tw = ctx->d_tw; a.gain = gain; a.fp_mode = c.fp_mode;
            hipError_t e = launch_conv_general(a, (int)c.S, st);
            if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv launch: ") + hipGetErrorString(e));
            return OHS_OK;
        };
        if (!tp) {
            int rc = xh_append_launch();
            if (rc) return rc;
            xh_advance();
            rc = general(in, out, n_blocks, c.cnt, 0, 0, 0, c.d_tails);
            if (rc) return rc;
            c.last_kernel = OHS_CONV_KERNEL_SEQUENTIAL; c.last_ranges = 1;
        } else {
            // segment size: keep the MAC scratch around 128 MiB
            long long SEG = 16384 / (long long)c.S;
            SEG = std::max<long long>(8, std::min<long long>(SEG, 1024));
            if (tp_gated) SEG = std::max<long long>(SEG, Pmax);
            const long long seg_max = std::min<long long>(SEG, n_blocks);
            const int Ppad = Ppad_for_ring(Pmax);
            // (gated: old blocks + the zeroed gap + the segment live in the ring together)
            int rc = conv_grow_ring(c, (int)(seg_max + (tp_gated ? 2 : 1) * (Ppad - 1) + (tp_gated ? Pmax : 0)), st);
            if (rc) return rc;
            const unsigned long long cnt_old = c.cnt;
            if (tp_gated) {
                // the zeroed gap: ring slots cnt .. cnt + Ppad - 2 of every stream (two strided memsets when it wraps)
                const unsigned long long gap = (unsigned long long)(Ppad - 1), cap = (unsigned long long)c.cap;
                const unsigned long long s0 = c.cnt & (cap - 1), n0 = std::min(gap, cap - s0);
                const size_t pitch = (size_t)c.cap * NF * sizeof(float2);
                HIP_TRY(hipMemset2DAsync(c.d_hist + s0 * NF, pitch, 0, (size_t)n0 * NF * sizeof(float2), c.S, st));
                if (n0 < gap)
                    HIP_TRY(hipMemset2DAsync(c.d_hist, pitch, 0, (size_t)(gap - n0) * NF * sizeof(float2), c.S, st));
                c.cnt += gap;
            }
            if (!c.d_cdm || c.cdm_ppad != Ppad) {       // (a set_ir that keeps Ppad re-uses the buffer: no wait, no malloc)
                DeviceWideSection dws;
                HIP_TRY(hipStreamSynchronize(st));
                if (c.d_cdm) hipFree(c.d_cdm);
                c.d_cdm = nullptr; c.cdm_valid = false;
                HIP_TRY(hipMalloc(&c.d_cdm, (size_t)Ppad * 2 * NF * sizeof(float2)));
                c.cdm_ppad = Ppad;
            }
            if (!c.cdm_valid) {
                hipError_t e = launch_build_cd_multi(c.d_H[0], c.d_H[1], c.d_H[2], c.d_H[3], c.P[0], c.P[1],
                                                     c.P[2], c.P[3], Ppad, c.d_cdm, st);
                if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("build_cd_multi: ") + hipGetErrorString(e));
                c.cdm_ppad = Ppad; c.cdm_valid = true;
            }
            if ((size_t)seg_max > c.w_blocks) {
                DeviceWideSection dws;
                HIP_TRY(hipStreamSynchronize(st));
                if (c.d_W) hipFree(c.d_W);
                c.d_W = nullptr; c.w_blocks = 0;
                HIP_TRY(hipMalloc(&c.d_W, c.S * (size_t)seg_max * NF * sizeof(float2)));
                c.w_blocks = (size_t)seg_max;
            }
            if (!c.d_W1) HIP_TRY(hipMalloc(&c.d_W1, c.S * (size_t)NF * sizeof(float2)));
            long long done = 0;
            while (done < n_blocks) {
                const int seg = (int)std::min<long long>(SEG, n_blocks - done);
                ConvTpArgs a;
                a.in = in + done * BS; a.out = out + done * BS;
                a.in_stream_stride = in_ss; a.in_ch_stride = in_cs;
                a.out_stream_stride = out_ss; a.out_ch_stride = out_cs;
                a.n_streams = (int)c.S; a.seg_blocks = seg; a.n_mac = seg; a.Ppad = Ppad;
                a.CD = c.d_cdm; a.hist = c.d_hist; a.cap = c.cap; a.cnt = c.cnt + (unsigned long long)done;
                a.W = c.d_W; a.W1 = c.d_W1; a.tails = c.d_tails; a.tails_out = c.d_tails_alt;
                a.pairs = ctx->d_pairs; a.n_pairs = ctx->n_pairs;
                a.tw = ctx->d_tw; a.gain = gain; a.fp_mode = c.fp_mode;
                // (frame i of the segment is frame done * 512 + i of the call; the call's last xh_cnt frames are appended)
                a.xh_ring = xh_on ? c.d_xhist : nullptr; a.xh_len = c.xh_len; a.xh_head = xh_head0;
                a.xh_frame0 = done * BS - (call_frames - xh_cnt);
                hipError_t e = launch_conv_tp_forward(a, st);
                if (e == hipSuccess) e = launch_conv_tp_mac(a, st);
                if (e == hipSuccess && tp_gated && done == 0 && Pmax > 1) {
                    ConvTpOldArgs o;
                    o.n_streams = (int)c.S; o.n_old = Pmax - 1; o.Pmax = Pmax;
                    for (int p = 0; p < 4; ++p) { o.H[p] = c.d_H[p]; o.P[p] = c.P[p]; o.since[p] = c.since[p]; }
                    o.hist = c.d_hist; o.cap = c.cap; o.cnt_old = cnt_old;
                    o.W = c.d_W; o.n_mac = seg; o.fp_mode = c.fp_mode;
                    e = launch_conv_tp_old(o, st);
                }
                if (e == hipSuccess) e = launch_conv_tp_inverse(a, st);   // block 0 reads the OLD state; new state -> the other slab
                if (e != hipSuccess) return fail(OHS_ERR_HIP, std::string("conv_tp: ") + hipGetErrorString(e));
                std::swap(c.d_tails, c.d_tails_alt);
                done += seg;
                c.last_kernel = OHS_CONV_KERNEL_BLOCK512_TP; c.last_ranges = seg;
            }
            xh_advance();
        }
    }
    c.cnt += (unsigned long long)n_blocks;
    for (int p = 0; p < 4; ++p) c.since[p] += n_blocks;
    return OHS_OK;
}

}  // namespace ohs_api
