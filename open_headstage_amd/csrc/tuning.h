// tuning.h -- every launch-plan constant of the host layer in one place.
//
// Product build: tuning() is a constant object -- the measured best of rounds 1-3 -- and nothing can change it.
// Experiments build (-DOHS_EXPERIMENTS): the same object starts from the OHS_* environment variables of LABNOTES.md
// and ohs_debug_set_tuning(key, value) edits it in a running process (the tests' plan overrides, the sweep tools).
// The two options a deployment may legitimately set -- how many resident real-time kernels a device hosts and how long
// an idle one stays -- are arguments of the public ohs_rt_configure, not knobs.
#pragma once
#include "experiments.h"

#include <string>
#include <vector>

namespace ohs {

struct Tuning {
    // k_conv_p1 wave plan
    long long p1_target_waves = 0;          // > 0: chunks per stream = ceil(target / streams) instead of the round rule
    int p1_rounds = 1;                      // rounds of resident waves a launch is cut for
    int p1_own_tails = 1;                   // boundary tails by the chunks' own waves where a stream's chunks share a workgroup
    int p1_stagger = 0;                     // per-wave start offset, x 64 cycles
    int p1_prio = 0;                        // 1: rotate the issue priority per block
    int p1_weights[4] = {1425, 1128, 825, 622};     // chunk length by the age rank of the wave that runs it
    int p1_lazy_state = 1;                  // batch objects leave the merged overlap + last input block behind
    int p1_xcd_lo = 0, p1_xcd_n = 0;        // n > 0: every P = 1 launch confined to XCDs [lo, lo + n)
    // EQ launches
    int eq_xcd_lo = 0, eq_xcd_n = 0;        // n > 0: every ring launch confined to XCDs [lo, lo + n)
    int eq_conveyor = 0;                    // k_eq_pass also for <= 12 bands
    int eq_ring_v1 = 0;                     // round 2's 114-VGPR ring kernel (experiments build only)
    int eq_no_prio = 0;                     // no s_setprio 3 in k_eq_ring
    int eq_form = 0;                        // k_eq_ring's form: 0 = the library's choice, 1 = four chains per wave (16-lane rows), 2 = one chain
                                            // per wave (the ring over all 64 lanes: eq_ring64_body.hpp)
    int eq_wg_waves = 0;                    // 0: 1 wave per workgroup below one wave per CU, else 4
    int eq_lds = 0;                         // LDS reservation per EQ workgroup, bytes
    // ohs_batch_process
    int no_overlap = 0, force_overlap = 0;  // EQ || convolution overlap policy overrides
    int xcd_split = 0;                      // x > 0: overlapped calls run the EQ on XCDs [0, x), the convolution on [x, 8)
    std::vector<double> overlap_cuts = {0.0, 0.25, 0.5, 0.75, 0.92, 0.98, 1.0};
    std::vector<unsigned> conv_cu_mask;     // non-empty: the overlapped convolution's stream gets this CU mask
    int mac_no_xcd = 0;                     // plain stream-major workgroup order in k_conv_tp_mac
    int lb_min_p = 2;                       // 512-tap partitions of the longest response from which the block-2048 plan is considered:
                                            // every response longer than one partition (64 streams x 256 blocks: taps 1024 0.122 vs 0.189 ms for
                                            // block 512, taps 3000 0.123 vs 0.191; 256 streams 0.50 vs 0.79 -- LABNOTES round 5)
    int lb_seg_windows = 131072;            // windows (streams x blocks of 2048) per block-2048 launch pair, i.e. a ring of at most 4 GiB (a call's
                                            // own I/O at that size: 4 GiB): 512 streams x 256 blocks 1.16 ms at 4 096 -> 0.84, 2 048 streams 2.97 ms
                                            // (profiles/r05_lb_seg_sweep.txt)
    int xb_min_p = 2;                       // 512-tap partitions of the longest response from which the block-8192 kernel (conv_xb_kernels.hip)
                                            // serves long out-of-place calls (up to 32: two partitions of 8192 taps)
    int xb_min_blocks = 128;                // ... for calls of at least this many 512-frame blocks (64 streams x 128 blocks: a tie with block 2048;
                                            // 256 blocks 0.118 vs 0.140 ms, 1024 blocks 0.38 vs 0.55: profiles/r05_xb_vs_lb_shapes.txt)
    int xb_min_streams_p2 = 32;             // ... and, with two partitions, on at least this many streams (8 streams: 0.046 vs 0.043 ms)
    int xb_stagger = 1;                     // every second workgroup starts this many x 3.5 us late
    int xb_skip = 0;                        // k_conv_xb timing knock-outs (WRONG results): 1 no input loads, 2 no twiddle loads, 4 no forward
                                            // transforms, 8 no table loads, 16 no inverse transforms, 32 no stores, 64 no radix-16 steps
    int xb_form = 2;                        // experiments build: 1 = the sixteen-wave first form (experiments/conv_xb16_kernels.hip)
    int xb_append_launch = 0;               // experiments build: 1 = the history append as a launch of its own (A/B)
    int xb_run = 0;                         // > 0: blocks of 8192 per workgroup (else conv_xb_run_for)
    int lb_skip = 0;                        // k_conv_lb_mac_inverse timing knock-outs (WRONG results; experiments build only): 1 = no
                                            // products, 2 = no inverse transforms, 4 = no ring loads, 8 = no table loads
    // ohs_batch_process_host
    int host_pipe_trace = 0, host_pipe_flat_priorities = 0;
    // ohs_node_batch_create (experiments build only): several device slots may name the SAME device -- a rehearsal of the
    // multi-slot logic (sharding, per-slot unpack, worker threads, per-slot pointers) on a one-GPU box.  RCCL refuses two
    // ranks on one device, so in this mode the tables travel slot to slot by device copies and no communicator exists.
    int node_shared_device_rehearsal = 0;
    int node_inject_table_failure_slot = 0;     // ohs_node_batch_set_tables fails in front of device slot (this - 1): the tear test
    // resident real-time kernels: set through ohs_rt_configure in either build
    int rt_max_resident = -1;               // per device; < 0: a quarter of the CUs
    unsigned long long rt_idle_polls = 3000000ull;      // polls without a doorbell before the kernel leaves (~2 s)
    unsigned rt_poll_sleep = 0;             // pause between poll bursts, x s_sleep 16
};

const Tuning &tuning();
Tuning &tuning_rt();        // the three rt_* fields only (ohs_rt_configure); everything else stays constant in the product
// ohs_rt_configure may be called while other threads' engines start resident kernels: the two fields it writes are
// read and written through these (relaxed atomics on the plain fields -- Tuning stays copyable)
inline int rt_max_resident_load() { return __atomic_load_n(&tuning().rt_max_resident, __ATOMIC_RELAXED); }
inline unsigned long long rt_idle_polls_load() { return __atomic_load_n(&tuning().rt_idle_polls, __ATOMIC_RELAXED); }
inline void rt_max_resident_store(int v) { __atomic_store_n(&tuning_rt().rt_max_resident, v, __ATOMIC_RELAXED); }
inline void rt_idle_polls_store(unsigned long long v) { __atomic_store_n(&tuning_rt().rt_idle_polls, v, __ATOMIC_RELAXED); }
#ifdef OHS_EXPERIMENTS
// "key=value" edits; returns false for an unknown key or an unparsable value
bool tuning_set(const std::string &key, const std::string &value);
#endif

}  // namespace ohs
