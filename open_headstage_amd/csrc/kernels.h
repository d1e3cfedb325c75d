// kernels.h -- host-callable launchers of the gfx950 kernels (internal to the library).
#pragma once
#include "experiments.h"
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace ohs {

// ---- EQ ------------------------------------------------------------------------
// One pass of up to 16 cascaded DF2T bands (parametric_eq.rs:116-122,166-179).
// Only ENABLED bands are handed to the kernel (a disabled band is the identity and keeps its state,
// parametric_eq.rs:118-120), so lane j of a row is the j-th enabled band of this pass and `slot[j]` is
// that band's index in the per-chain state array.
// Denormal handling of the caller: nih-plug runs Plugin::process with the CPU's flush-to-zero bit set
// (ScopedFtz: MXCSR.FTZ -- results that would be denormal become signed zeros; denormal INPUTS are still read), the
// reference's unit tests run with IEEE denormals.  fp_mode of a launch: 0 = IEEE (the default, what the parity tests
// and the unit tests compare), 1 = flush denormal results (FTZ), 2 = flush results and read denormal inputs as zero
// (FTZ | DAZ, what many audio hosts set themselves).  MODE.FP_DENORM bits [5:4] govern f32: 3 = keep both, 1 = keep
// inputs / flush results, 0 = flush both.  The kernels contain no f32 division or square root (their expansions
// switch the mode and restore the compile-time default), so a mode set at kernel entry holds to the end.
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void ohs_set_fp_mode(int fp_mode)
{
    // hwreg(HW_REG_MODE = 1, offset 4, size 2): id | offset << 6 | (size - 1) << 11
    if (fp_mode == 1) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 1);
    else if (fp_mode == 2) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);
}
#endif

struct EqPassTable {
    float b0[16], b1[16], b2[16], a1[16], a2[16];
    int slot[16];
};
// Per-stream tables (batch mode, ohs_batch_set_stream_eq_band_coeffs): the ENABLED bands of one stream in cascade order,
// at most 12 (the ring form); read from device memory by the stream's two rows when their wave starts.
struct EqStreamTable {
    float b0[12], b1[12], b2[12], a1[12], a2[12];
    int slot[12];               // the bands' own indices (state slots); unused entries repeat the last one
    int nb;                     // enabled bands, 0 .. 12
    int pad[3];
};
constexpr int kEqStateSlots = 64;       // = OHS_MAX_EQ_BANDS
// chains = streams * 2 (L, R); chain c -> stream c>>1, channel c&1.
// state: [chain][kEqStateSlots][2] (s1, s2), indexed by the band's own index.
// n_bands = bands of this pass (1..16, lanes >= n_bands are unused).
// exact_specials: use the conveyor form (k_eq_pass: samples move by DPP moves only) also for <= 12 bands; it is
// bit-exact for -0.0 and for samples next to a non-finite input, where the faster ring form is not (its
// pass-on lanes run 1*x + 0).
hipError_t launch_eq_pass(const float *in, float *out, long long stream_stride, long long ch_stride,
                          long long n, int n_chains, const EqPassTable &tab, int n_bands, float *state,
                          hipStream_t st, bool exact_specials = false, int fp_mode = 0, int xcd_lo = 0, int xcd_n = 8,
                          hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, unsigned long long *stamps = nullptr);
// (stamps, optional, device memory [4]: wave 0 of a RING launch writes {real-time counter at start, at end, shader-clock
// counter at start, at end}: s_memrealtime runs at 100 MHz, s_memtime at the shader clock)
// (ev_start / ev_stop: recorded at the start / completion of THIS launch.  The ring kernel's dispatch carries them
// itself -- hipExtLaunchKernelGGL -- which spares the queue the marker packets of hipEventRecord and the gap they open
// between back-to-back launches: 12-13 -> 7 us per chunk boundary of the overlapped batch step with profiling on.)

// the ring form with one table per stream: d_tabs[n_chains / 2] in device memory (every stream <= 12 enabled bands)
hipError_t launch_eq_ring_streams(const float *in, float *out, long long stream_stride, long long ch_stride, long long n,
                                  int n_chains, const EqStreamTable *d_tabs, float *state, hipStream_t st, int fp_mode = 0,
                                  int xcd_lo = 0, int xcd_n = 8, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
bool eq_ring_addressable(long long stream_stride, long long ch_stride, long long n);     // the strides the ring form reaches

// |H(f)| of the enabled bands (parametric_eq.rs:190-209); coeffs [nb][5], en [nb]
hipError_t launch_eq_freq_response(const float *d_coeffs, const int *d_en, int nb, float fs,
                                   const float *d_freqs, int n, float *d_out, hipStream_t st);

// ---- convolution -------------------------------------------------------------
// "General" kernel: faithful per-path semantics of src/dsp/convolution.rs
// (per-path partition count, per-path history reset, four separate overlap tails).
struct ConvGeneralArgs {
    const float *in;            // planar device audio
    float *out;
    long long in_stream_stride, in_ch_stride;     // floats
    long long out_stream_stride, out_ch_stride;   // floats
    int n_blocks;               // 512-frame blocks to process
    const float2 *H[4];         // per path: [P][16][64] spectra in bin layout
    int P[4];                   // partitions per path
    long long since[4];         // blocks processed since that path's last set_ir (before this launch)
    float2 *hist;               // [stream][cap][16][64] packed spectra ring Z = FFT(x_l + j x_r)
    int cap;                    // ring capacity, power of two >= max P
    unsigned long long cnt;     // absolute block counter at launch start
    const float2 *tails;        // [stream][2][8][64] state in: (lsl + j lsr) tail, (rsl + j rsr) tail
    float2 *tails_out;          // state out (may equal `tails`)
    int skip_store;             // leading blocks whose output is NOT stored (time-parallel path hand-over)
    int zero_tails_in;          // start from zero overlap instead of `tails`
    const float2 *tw;           // master twiddle table, 1024 entries
    float gain;
    int fp_mode;                // denormal handling, see ohs_set_fp_mode
    // Single-engine calls on responses of more than one partition (k_engine_call / k_engine_resident): the part of a block's
    // spectrum product that only involves blocks ALREADY in the ring does not wait for the call -- behind call N (the resident
    // kernel: behind its completion word; launch per call: k_engine_presum, queued behind k_engine_call) the next call's
    // block j (0 <= j < pre_next) and half h get
    //     pre[2 j + h] = sum over partitions i = j + 1 .. P - 1 of  history[cnt' + j - i] * H[i]        (cnt', since': after call N)
    // and call N + 1 only adds partitions 0 .. j (its own blocks) on top: a 1 024-frame call on a 16 384-tap response
    // waits for 1 + 2 partition products instead of 2 x 32.  pre_n = how many blocks of THIS call find their sum in `pre`
    // (0: none, every partition is computed in the call, as before).
    float2 *pre = nullptr;      // [8][2][16][64]
    int pre_n = 0, pre_next = 0;
};
hipError_t launch_conv_general(const ConvGeneralArgs &a, int n_streams, hipStream_t st);

// One-launch ConvolutionEngine::process_block for a single stream: FIFO
// adapter, blocks and pop in one kernel; `a` describes the convolution state (in/out fields unused).
struct EngineCallArgs {
    float *in_fifo;             // device [2][in_cap]; holds in_len (< 512) frames from earlier calls
    int in_cap, in_len;
    float *out_fifo;            // device [2][out_cap]; holds out_len frames not yet popped
    int out_cap, out_len;
    const float *src_l, *src_r; // this call's n input frames (mapped pinned host memory or device)
    float *host_l, *host_r;     // this call's n output frames (mapped pinned host memory)
    int n;
    int emit;                   // 1: out_len + 512 k >= n, the call pops n frames; 0: silence (host fills)
    float gain_pop;             // applied to popped frames only (lib.rs:1202-1207)
};
// done / seq: completion word in pinned host memory (nullptr: none), see k_engine_call
hipError_t launch_engine_call(const ConvGeneralArgs &a, const EngineCallArgs &e, hipStream_t st, unsigned *done = nullptr,
                              unsigned seq = 0);

// Mailbox of the resident real-time kernel, in pinned + mapped host memory (one per engine).
// call_word = {seq, n, gain bits, stop}: the host writes n and gain, then seq (release); the device reads all
// four with one aligned 16-byte load, so a new seq always comes with its own arguments.  `conv` and `call`
// describe the engine's state when the kernel starts (it copies them once; they must stay adjacent).
struct RtMailbox {
    alignas(64) unsigned call_word[4];
    alignas(64) unsigned done;   // device -> host: sequence number of the last call completed
    unsigned alive;              // device -> host: 0 once the kernel has decided to exit
    alignas(64) unsigned long long stamps[16];   // diagnostics: s_memrealtime (100 MHz) at the phases of the last call
    alignas(64) ConvGeneralArgs conv;
    EngineCallArgs call;
    // the chain's EQ (lib.rs:1179-1194), run by the kernel in front of the convolution when call_word[1] has bit
    // 31 set: the kernel re-reads this block whenever eq_version differs from the one it holds
    alignas(64) unsigned eq_version;
    int eq_nb;                   // enabled bands (<= 12: the ring form)
    float *eq_state;             // the ohs_eq handle's [2][kEqStateSlots][2] state
    float *eq_buf;               // device [2][eq_buf_stride] staging: the EQ runs in place here, the convolution reads it
    long long eq_buf_stride;
    EqPassTable eq_tab;
};
constexpr int kResidentMaxBlocks = 4;     // 512-blocks per call the resident kernel serves (host blocks of up to 2048 frames and a FIFO remainder)
hipError_t launch_engine_resident(RtMailbox *mb_dev, const float2 *tw, unsigned first_seq, unsigned long long idle_limit,
                                  unsigned sleep_arg, hipStream_t st);

// Fast path when every path has exactly one partition (taps <= 512): time-parallel over
// `chunks` contiguous block ranges per stream, one forward + one inverse FFT per block, the
// combined spectra C = (A - jB)/2, D = (A + jB)/2 (A = H0 + jH1, B = H2 + jH3) held in LDS.
// XCD partition of a launch.  The dispatcher hands workgroup b of a grid to XCD b % 8 (MI355X: 8 XCDs of 32 CUs, each
// with its own L2); a kernel confined to the XCDs [lo, lo + n) is launched with xcd_grid(workgroups, n) workgroups, those
// on other XCDs return at once and the rest renumber themselves ((b >> 3) * n + (b & 7) - lo).  ohs_batch_process gives
// the EQ and the overlapped convolution disjoint XCD sets: an EQ wave saturates the vector unit of its SIMD, and a
// convolution workgroup that shares a CU with one waits for its starved waves (LABNOTES.md, "What sharing a CU is worth").
inline unsigned xcd_grid(unsigned workgroups, int xcd_n)
{
    return xcd_n == 8 ? workgroups : ((workgroups + (unsigned)xcd_n - 1) / (unsigned)xcd_n) * 8u;
}

struct ConvP1Args {
    const float *in;
    float *out;
    long long in_stream_stride, in_ch_stride;
    long long out_stream_stride, out_ch_stride;
    int n_blocks;
    int n_streams;
    int chunks;                 // K contiguous block ranges per stream
    const float2 *CD;           // [2][16][64]: C then D, bin layout
    const float2 *H[4];         // single-partition spectra (used for the last block: split tails)
    const float2 *tails;        // [stream][2][8][64] state in, same format as the general kernel
    float2 *tails_out;          // state out (a different slab; the host swaps them after the launch)
    float2 *chunk_tails;        // [stream][chunks][8][64] scratch: overlap tail ENTERING chunk e (e >= 1), from k_conv_p1_edges
    const float2 *tw;
    float gain;
    int stagger;                // wave w of a workgroup starts its block loop w * stagger * 64 cycles late (0 = off)
    int prio_mode;              // issue priority of the waves: 0 = hardware default (oldest first), 1 = rotating per block
    int weights[4];             // relative length of a chunk by the age rank of the wave that runs it (conv_kernels.hip: p1_chunk_begin)
    int fp_mode;                // denormal handling, see ohs_set_fp_mode
    int xcd_lo, xcd_n;          // XCDs the main and edges kernels may use (xcd_grid); 0, 8 = all
    // Lazy state (batch objects): merged_out != nullptr -- the launch leaves its state as the MERGED overlap of its last
    // block ([stream][8][64], what its own chunks hand each other) plus a copy of that block's input frames
    // (last_in: [stream][2][512]), and k_conv_p1_state is not launched; the per-path overlaps the state at rest
    // consists of are computed from last_in when somebody needs them (api_conv.hip: conv_materialise_state).
    // merged_in != nullptr: the state the launch starts from is such a merged overlap (else tails: the per-path form).
    const float2 *merged_in;
    float2 *merged_out;
    float *last_in;
    int own_tails;              // 1: no boundary-tail pre-pass, every wave runs one dry block in front of its chunk (chunks 2, 4, 8 or
                                // 16: all chunks of a stream in one workgroup -- p1_wave_job); 0: k_conv_p1_edges fills chunk_tails
};
// (ev_start / ev_stop, optional: recorded at the start / completion of the launches; carried by the dispatch itself when
// the call is a single kernel -- see launch_eq_pass)
hipError_t launch_conv_p1(const ConvP1Args &a, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// k_conv_p1_state alone: the per-path overlaps block n_blocks - 1 of a.in leaves behind -> a.tails_out
hipError_t launch_conv_p1_state(const ConvP1Args &a, hipStream_t st);
int conv_p1_waves_per_cu();      // resident waves of k_conv_p1 per CU (one workgroup)
// CD[0][..] = (A - jB)/2, CD[1][..] = (A + jB)/2 from four single-partition spectra
hipError_t launch_build_cd(const float2 *h0, const float2 *h1, const float2 *h2, const float2 *h3,
                           float2 *cd, hipStream_t st);

// ---- P = 1, batch objects: overlap-save with a 2048-point window, hop 1536 (conv_os_kernels.hip) ------------------
// Same inputs, outputs and state hand-over (merged overlap + last input block, the lazy state above) as k_conv_p1; a third
// less transform and product work per output frame.  Results equal k_conv_p1's up to f32 rounding, not bit for bit.
struct ConvOsArgs {
    const float *in;
    float *out;
    long long in_stream_stride, in_ch_stride;       // floats; even (8-byte accesses)
    long long out_stream_stride, out_ch_stride;
    int n_blocks;               // 512-frame blocks of the call
    int n_streams;
    int chunks;                 // contiguous hop ranges per stream; in place (in == out) a divisor of the workgroup's waves
    const float4 *CD;           // [2 lo/hi][16][64] (C, D), paired layout, 1/2048 folded in (launch_os_tables)
    const float2 *tw2048;       // [16][64] w2048^bin, paired layout
    const float2 *tails;        // state in, per-path form (used when merged_in == nullptr)
    const float2 *merged_in;    // state in, merged form
    float2 *merged_out;         // state out
    float *last_in;             // state out: the call's last 512 input frames
    const float2 *tw;
    float gain;
    int fp_mode;
};
hipError_t launch_conv_p1_os(const ConvOsArgs &a, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// (C, D) lo / hi from the four impulse responses in the time domain: d_irt [4][512] floats, zero-padded
hipError_t launch_os_tables(const float *d_irt, const float2 *tw, const float2 *tw2048, float4 *cd_out, hipStream_t st);
void os_build_twiddles(float2 *out);    // host: [16][64]
int conv_os_waves_per_cu();

// ---- time-parallel long-IR path (any P, no per-path gating needed) ----------------------------
// F: Z_t = FFT(x_l + j x_r) for every (stream, block) of the segment -> history ring
// M: W_t = sum_i Z_{t-i} C_i + conj(Z_{t-i}[N-k]) D_i, register-tiled over (time x partition)
// I: out_t = IFFT(W_t + (-1)^k W_{t-1})[0..512) -- the overlap-add folded in the spectral domain
struct ConvTpArgs {
    const float *in;
    float *out;
    long long in_stream_stride, in_ch_stride;
    long long out_stream_stride, out_ch_stride;
    int n_streams;
    int seg_blocks;             // blocks whose spectra F writes
    int n_mac;                  // blocks M / I produce (= seg_blocks)
    int Ppad;                   // partitions, padded to a multiple of 8 (CD zero beyond the real count)
    const float2 *CD;           // [Ppad][2][1024] combined spectra C_i, D_i (bin layout)
    float2 *hist;               // [stream][cap][1024] ring
    int cap;
    unsigned long long cnt;     // absolute index of the segment's first block
    const int *pairs;           // leaders b of the 514 position pairs {b, mirror(b)} (device)
    int n_pairs;
    int mac_tiles;              // set by launch_conv_tp_mac: 8-block time tiles per stream
    int main_wgs;               // set by the launchers: workgroups (MAC) / waves (inverse) of the main job; the grid's
                                // trailing ones compute W1 / the new state
    float2 *W;                  // [stream][n_mac][1024] scratch
    float2 *W1;                 // [stream][1024]: L-speaker part (lsl + j lsr) of the LAST block's spectrum
    const float2 *tails;        // state in (sum of the two packed tails feeds block 0)
    float2 *tails_out;          // state out: the last block's four overlaps, general-kernel format
    const float2 *tw;
    float gain;
    int fp_mode;                // denormal handling, see ohs_set_fp_mode
    // F appends the call's last frames to the input history of the block-2048 plan (ConvLbArgs: a ring of 2 xh_len per row):
    // frame i of this segment goes to (xh_head + xh_frame0 + i) mod 2 xh_len when xh_frame0 + i >= 0.  xh_ring == NULL: no history
    float *xh_ring;
    long long xh_len, xh_head, xh_frame0;
};
hipError_t launch_conv_tp_forward(const ConvTpArgs &a, hipStream_t st);
hipError_t launch_conv_tp_mac(const ConvTpArgs &a, hipStream_t st);
hipError_t launch_conv_tp_inverse(const ConvTpArgs &a, hipStream_t st);     // + the new state (the last block's four overlaps)
// O: the blocks the ring held BEFORE the call, under per-path gating -- what the three kernels above cannot do.  After a
// per-path set_ir in mid-stream path p must not see blocks older than its own reset (convolution.rs:135-138: that
// path's history is all zero again), while the other paths still do.  The host hides the old blocks from F / M / I by
// starting the call Ppad - 1 zeroed ring slots further on; this kernel then adds, for the first n_old = Pmax - 1 blocks
// of the call, W_t += sum_{d >= 1, i = t + d < Pmax} Z_{-d} C_i^(d) + conj(Z_{-d}[N-k]) D_i^(d), where C^(d), D^(d) are
// built on the fly from the paths that may see old block -d (d <= since[p]) and have an i-th partition.  thread = one
// layout position, workgroup = (quarter spectrum, block t, stream); every load is an L2 hit (ring + spectra).
struct ConvTpOldArgs {
    int n_streams, n_old, Pmax;
    const float2 *H[4];
    int P[4];
    long long since[4];         // blocks processed since that path's last set_ir, BEFORE this call
    const float2 *hist;         // the ring; old block -d sits in slot (cnt_old - d) & (cap - 1)
    int cap;
    unsigned long long cnt_old; // absolute index the call's first block would have had without the zeroed gap
    float2 *W;                  // [stream][n_mac][1024], accumulated into (M has written it)
    int n_mac;
    int fp_mode;
};
hipError_t launch_conv_tp_old(const ConvTpOldArgs &a, hipStream_t st);
int conv_tp_build_pairs(int *out);   // host: fills <= 514 leaders, returns the count
// CD[i][0] = (A_i - jB_i)/2, CD[i][1] = (A_i + jB_i)/2 for i < Ppad; H_p,i := 0 for i >= P_p
hipError_t launch_build_cd_multi(const float2 *h0, const float2 *h1, const float2 *h2, const float2 *h3,
                                 int p0, int p1, int p2, int p3, int Ppad, float2 *cd, hipStream_t st);

// ---- long impulse responses, batch objects: block 2048 / FFT 4096, overlap-save (conv_lb_kernels.hip) ------------
// The reference cuts every impulse response into 512-tap partitions (convolution.rs:120-132) and pays P x 1024 complex
// products per 512 frames (:261-273).  An offline call can afford a longer block: with 2048-tap partitions the product
// work and the history reads per frame drop to a quarter (16 384 taps: 8 partitions instead of 32) for 1.2 x the
// transform work.  The 4096-point transform is four of the wave's 1024-point transforms (one per wave of a workgroup)
// and a lane-local radix-4 step; spectra are stored as PAIRS (Z[q], Z[4096 - q]) so that the packed two-channel product
// needs no mirror fetch; the product and the inverse transform are ONE kernel (a workgroup owns four consecutive
// output blocks of a stream: the product's results never travel through HBM).
//   F  k_conv_lb_forward      : window u = frames [(u - 1) 2048, (u + 1) 2048) -> ring slot (ring_base + u) % ring_cap
//   MI k_conv_lb_mac_inverse  : W_t = sum_i Z_{t-i} C_i + conj(Z_{t-i}[N - k]) D_i ; out block t = IFFT(W_t)[2048, 4096)
// Results equal the block-512 kernels' up to f32 rounding, not bit for bit.
struct ConvLbArgs {
    const float *in;            // the segment's audio (planar device buffers); tables mode: the impulse responses
    float *out;
    long long in_stream_stride, in_ch_stride;       // floats; multiples of 4 (16-byte accesses)
    long long out_stream_stride, out_ch_stride;
    // the input history: [stream][2][2 xh_len], a ring of TWICE the reach xh_len: the frame k frames in front of the segment at
    // (xh_head - k) mod 2 xh_len, k <= xh_len -- "the segment" here is its first buffer frame, io_frame0 --; may be NULL.
    // xh_append != NULL: F also appends the buffer's last xh_cnt = min(io_frames, xh_len) frames at xh_head .. (the half of
    // the ring nobody reads), the host then advances the head
    const float *xhist;
    float *xh_append;
    long long xh_len, xh_head, xh_cnt;  // multiples of 512
    int n_streams;
    // Frames are counted from the segment's grid origin (window u = frames [(u - 1) 2048, (u + 1) 2048)).  The buffers `in` /
    // `out` hold the frames [io_frame0, io_frame0 + io_frames): earlier ones come from the input history (F) / are not
    // stored (MI: a call that starts inside a 2048-frame block of the stream's absolute grid recomputes that block and
    // stores its new frames only); later ones are zeros (F) / not stored (MI).  Multiples of 512.
    long long io_frame0, io_frames;
    int u0, n_fwd;              // F: windows u0 .. u0 + n_fwd - 1 (u < 1 reaches into xhist)
    int run;                    // F: consecutive windows per workgroup; 0 = the launcher's choice
    int n_big;                  // MI: output blocks 0 .. n_big - 1
    int P2pad;                  // 2048-tap partitions, padded to a multiple of 4 (zero tables beyond the real count)
    float4 *ring;               // [stream][ring_cap][2048]: slot q = (Z[q], Z[4096 - q]), slot 0 = (Z[0], Z[2048])
    int ring_cap;
    long long ring_base;        // window u lives in ring slot (ring_base + u) % ring_cap; ring_base + u >= 0 for every u touched
    const float4 *cd;           // [P2pad][2 planes][2048]: plane 0 = (C[k], D[k]), plane 1 = (C[N - k], D[N - k]), 1/4096 folded in
    const float2 *tw;           // master twiddle table of the 1024-point transform
    const float2 *tw4096;       // [3][1024]: w4096^(n1 c), c = 1 .. 3
    float gain;
    int fp_mode;
    int tables_mode;            // F: 1 = window i is [block i of `in`, 2048 zeros] (impulse-response partitions)
    int debug_skip;             // experiments build only (Tuning::lb_skip): timing knock-outs, wrong results; 0 in the product
};
hipError_t launch_conv_lb_forward(const ConvLbArgs &a, hipStream_t st);
hipError_t launch_conv_lb_mac_inverse(const ConvLbArgs &a, hipStream_t st, int *tile_blocks = nullptr);   // -> output blocks per workgroup
// (C, D) tables from the spectra of A = h_lsl + j h_lsr (ab[0][i]) and B = h_rsl + j h_rsr (ab[1][i]), i < P2pad
// (mask: bit 0 = the L-speaker pair takes part, bit 1 = the R-speaker pair; 3 = the tables of the convolution)
hipError_t launch_conv_lb_build_cd(const float4 *ab, int P2pad, float4 *cd, hipStream_t st, int mask = 3);
// Pending tails [stream][4 paths][len]: what the frames in front of a per-path set_ir still contribute to the frames behind
// it, per path (a later reset of a path zeroes its row).  merge: dst rows of the pair's two paths = the rest of the pending
// row (from pos on) + the fresh tail fresh[stream][2][len] (re -> path 2 pair, im -> path 2 pair + 1), or zeros where
// keep_mask has the path's bit clear; add: out += gain * (ear's two rows) for the call's frames
hipError_t launch_conv_lb_tails_merge(const float *pend, long long pos, const float *fresh, float *dst, long long len, int pair,
                                      int keep_mask, int n_streams, hipStream_t st);
hipError_t launch_conv_lb_tails_add(float *out, long long out_ss, long long out_cs, long long n_frames, const float *tails,
                                    long long len, long long pos, float gain, int n_streams, hipStream_t st);
// the input history [stream][2][2 len] (a ring of twice its reach `len`): append the call's last cnt = min(n_frames, len)
// frames at head .. head + cnt (mod 2 len); the caller then advances head by cnt.  (The forward kernels do this themselves;
// this launch serves the calls that go through the sequential kernel.)
hipError_t launch_conv_lb_xhist_append(const float *in, long long in_ss, long long in_cs, float *ring, long long len,
                                       long long head, long long n_frames, int n_streams, hipStream_t st);
// dst[row][j] = the frame (len - j) frames in front of the head, j = 0 .. len - 1: oldest first, newest last; dst rows of len
hipError_t launch_conv_lb_xhist_linearise(const float *ring, float *dst, long long len, long long head, int n_streams,
                                          hipStream_t st);
void conv_lb_build_twiddles(float2 *out);       // host: [3][1024]
constexpr int kLbBlock = 2048;                  // frames per block of this plan
constexpr int kLbTile = 4;                      // output blocks per MI workgroup (calls of 1 or 2 blocks: that many)

// ---- long impulse responses, batch objects, long out-of-place calls: block 8192 / FFT 16384, one kernel (conv_xb_kernels.hip) ------
// Responses of at most two 8192-tap partitions: W_i = Z_i CD_0 + carry, carry' = Z_i CD_1 -- no spectrum leaves the
// workgroup (8 waves = sixteen 1024-point transforms + a lane-local radix-16 step) that runs consecutive blocks of a
// stream.  Frames are counted from the call's first frame; earlier ones come from the input history (ConvLbArgs: a ring
// of twice its reach xh_len >= P2 x 8192), later ones are zeros.  Out of place only (a run's first windows read frames
// another workgroup's blocks would overwrite).  Results equal the other plans' up to f32 rounding.
struct ConvXbArgs {
    const float *in;            // the call's audio (planar); tables mode: the impulse responses
    float *out;
    long long in_stream_stride, in_ch_stride, out_stream_stride, out_ch_stride;     // floats; multiples of 4 (16-byte aligned buffers)
    const float *xhist;         // [stream][2][2 xh_len]; the frame k frames in front of the call at (xh_head - k) mod 2 xh_len; may be NULL
    float *xh_append;           // != NULL (k_conv_xb only): the call's last xh_cnt = min(io_frames, xh_len) frames are appended at xh_head ..
                                // (the half of the ring nobody reads) by the workgroups that load them as their blocks' own frames
    long long xh_len, xh_head, xh_cnt;
    long long io_frames;        // frames in `in` / `out` (a multiple of 512)
    int n_streams;
    int n_blk;                  // blocks of 8192 frames (the last one may reach beyond io_frames)
    int run;                    // consecutive blocks per workgroup (conv_xb_run_for)
    int P2;                     // partitions of 8192 taps: 1 or 2
    const float4 *cd;           // [P2][8 waves][2 classes][16 registers][64 lanes]: (C[k], D[k]) of that bin, 1/16384 folded in
    float4 *ab;                 // tables mode: the windows' spectra per bin (float2, same layout), [stream][n_blk][16384]
    const float2 *tw;           // master twiddle table of the 1024-point transform
    const float2 *tw16384;      // [15][1024]: w16384^(n1 c), c = 1 .. 15
    float gain;
    int fp_mode;
    int tables_mode;            // 1 = window i is [block i of `in`, 8192 zeros], spectra to `ab`, nothing else
    int stagger;                // every second workgroup starts this many x 3.5 us late (the CUs' load bursts then do not coincide)
    int debug_skip;             // experiments build only (Tuning::xb_skip): timing knock-outs, wrong results; 0 in the product
};
hipError_t launch_conv_xb(const ConvXbArgs &a, hipStream_t st);
hipError_t launch_conv_xb_build_cd(const float2 *ab, int P2, float4 *cd, hipStream_t st);
int conv_xb_run_for(int n_streams, int n_blk, int P2, int num_cus);
void conv_xb_build_twiddles(float2 *out);       // host: [15][1024]
constexpr int kXbBlock = 8192;                  // frames per block of this plan
#ifdef OHS_EXPERIMENTS
// the plan's first form (experiments/conv_xb16_kernels.hip): sixteen waves, pair tables [P2][2 planes][8192] as the block-2048 plan's
hipError_t launch_conv_xb16(const ConvXbArgs &a, hipStream_t st);
hipError_t launch_conv_xb16_build_cd(const float4 *ab, int P2, float4 *cd, hipStream_t st);
#endif

// IR chunk -> spectrum: dst[part][16][64], one wave per partition.
hipError_t launch_ir_spectrum(const float *d_ir, int len, int n_parts, float2 *dst,
                              const float2 *tw, hipStream_t st);

// set_ir's overlap reset (convolution.rs:138) for every stream: tails is [S][2][512] float2;
// zero the re (comp 0) or im (comp 1) part of tail `which` (0 = L-speaker pair, 1 = R-speaker pair).
hipError_t launch_zero_tail_component(float2 *tails, long long n_streams, int which, int comp, hipStream_t st);

// diagnostic natural-order FFT-1024 (batch transforms)
hipError_t launch_debug_fft(const float2 *in, float2 *out, int inverse, int batch, const float2 *tw,
                            hipStream_t st);

// dst[i] = src[i] * gain  (n floats)
hipError_t launch_scale_copy(const float *src, float *dst, long long n, float gain, hipStream_t st, int fp_mode = 0);
hipError_t launch_scale_copy_done(const float *src, float *dst, long long n, float gain, unsigned *counter, unsigned *done,
                                  unsigned seq, hipStream_t st, int fp_mode = 0);

}  // namespace ohs
