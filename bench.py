#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json `metric` on its configs[2] workload.

  workload : 256 concurrent stereo streams per GPU, 4 x 512-tap HRIR partitioned FFT convolution
             + 10-band parametric EQ + output gain, 48 kHz f32, synthetic white noise
             (SURVEY.md section 8d generators), data resident in HBM.
  step     : one pass of the hot path over one batch = every stream advanced by
             `--frames` frames (default 480256 = 938 internal blocks ~ 10 s of audio).
  value    : whole-job stereo frames per second, in Msamples/s (1 sample = one L+R frame).
  scaling  : weak by default -- 256 streams per GPU; `--total-streams 2048` fixes the job size instead
             (strong scaling, SURVEY section 8e: 2048 / N streams per GPU).  Streams are sharded by
             contiguous id range, no data-path collective; the shared HRIR set + EQ table is broadcast
             once from rank 0 (RCCL over xGMI via torch.distributed "nccl").

Launch:  python bench.py [--gpus N] [--steps K] [--warmup W]
             N > 1 without WORLD_SIZE in the environment: this process never touches the GPU -- it starts N child
             ranks of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays
             rank 0's line and exits with the worst child status
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.  At N > 1 it carries `ranks` (per rank: device index, PCI bus id / UUID, streams, ms per
step -- gathered with all_gather_object), `rccl_world_size`, `distinct_devices`, and in strong mode
`expected_strong_scaling`.  Beside the contract's fields it carries, at N = 1:
  roofline      the dominant kernel (HIP-event time inside the timed region) against HBM and, when it is the
                EQ's serial recurrence, against its instruction-issue floor; `conv` = the convolution kernel
                against HBM by the bytes it really moves
  saturation    the same step at 1024 and 2048 streams (where the chip, not one chain's recurrence, is the bound)
  pcie_inclusive  the headline step fed from / returned to pinned host memory (ohs_batch_process_host)
  cpu_baseline  the CPU oracle on this box's host cores

  node_batch    the same job through the C ABI's node API (ohs_node_batch_*: ONE process, N devices, device-resident
                audio, the tables carried by the library's own RCCL broadcast) -- measured in a child run of
                `bench.py --gpus N --node-batch` after the timed region (at N > 1 the other ranks wait on the CPU)

  python bench.py --gpus N --node-batch [--steps K --warmup W]
         ONE process drives N devices through ohs_node_batch_create / _set_tables / _process / _sync only (no
         torch.distributed; torch owns the device buffers and generates the noise).  This is the path a Rust host binds
         (INTEGRATION.md section 7); its record carries `slots` (device, PCI bus id, first stream, streams, device ms per
         step), `rccl_comm_size`, `value` (device-resident) and `pcie_inclusive` (ohs_node_batch_process_host).

Rehearsal of the multi-rank plumbing on a one-GPU box: OHS_BENCH_DIST_BACKEND=gloo (CPU broadcast, every rank on
cuda:0) -- see tests/test_gpu_bench_rehearsal.py.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Msamples/s stereo 48 kHz (4×512-tap HRIR conv + 10-band EQ); % HBM3E peak"      # BASELINE.json, verbatim
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
STREAMS_PER_GPU = 256
TAPS = 512
FS = 48000.0
NOMINAL_CLOCK_GHZ = 2.4        # same guide: max clock; only used to express the EQ's ns per sample in cycles
EQ_ISSUE_FLOOR_CYCLES = 26.7   # 6.5 VALU per sample x 4.1 cycles per instruction of a lone wave: four chains per wave
EQ_ISSUE_FLOOR_CYCLES_WAVE_RING = 24.8     # 6.04 VALU per sample x 4.1: one chain per wave, the ring over all 64 lanes (eq_ring64_body.hpp)
# ohs_batch_last_conv_plan's kernel family -> the kernel that dominates it (the name rocprofv3 shows)
KERNEL_OF_FAMILY = {"block512_p1": "k_conv_p1", "hop1536_p1": "k_conv_p1_os", "block512_tp": "k_conv_tp_mac",
                    "block2048": "k_conv_lb_mac_inverse", "block8192": "k_conv_xb", "sequential": "k_conv_general", "none": "none"}


def cpu_info():
    """(model name, logical CPUs of the box, CPUs this process may really use).  The last one is the smaller of
    the affinity mask and the cgroup CPU quota: a GPU box hands a job a share of a large host (e.g. 16 of 256
    logical CPUs), and threads beyond that share only time-slice."""
    model, total = "unknown", os.cpu_count() or 1
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = total
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(int(q) / int(per) + 0.5))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = int(f1.read()), int(f2.read())
                if q > 0:
                    quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    if quota:
        usable = min(usable, quota)
    return model, total, usable


def cpu_baseline(frames: int, target_seconds: float = 10.0):
    """Times the CPU oracle (oracle/ohs_oracle.c, kind "port": a C restatement that keeps the
    reference's structure -- 4 forward + 4 inverse complex FFT-1024 per block, four per-path
    histories, serial DF2T cascade) on EVERY host core this process may use, built here with
    -march=native.  Bounded sample of the same workload: whole streams of `frames` frames, host block
    1024 (lib.rs process cadence); every thread runs streams until ~target_seconds have passed."""
    from oracle import ohs_oracle as orc
    from open_headstage_amd import synth

    flags = "-O3 -march=native -ffp-contract=off"
    try:
        orc.use_library(orc.build_native())
    except Exception:       # noqa: BLE001 -- no gcc on this box: the portable library
        orc.use_library(None)
        orc.build()
        flags = "-O3 -mavx2 -ffp-contract=off"
    irs = synth.hrir_set(TAPS)
    bands = synth.eq_table()
    coeffs = [orc.biquad_coeffs(int(b.filter_type), FS, b.center_freq, b.q, b.gain_db) for b in bands]
    sample_frames = min(frames, 480256)
    sample_frames -= sample_frames % 1024

    def run_stream(sid: int) -> None:
        eng = orc.ConvolutionEngine()
        for p in range(4):
            eng.set_ir(p, irs[p])
        eq = orc.StereoParametricEQ(len(bands), FS)
        for i, b in enumerate(bands):
            eq.set_band_coeffs(i, coeffs[i], b.enabled)
        x = synth.white_noise([sid], sample_frames)[0]
        for c in range(sample_frames // 1024):
            l = x[0, c * 1024:(c + 1) * 1024]
            r = x[1, c * 1024:(c + 1) * 1024]
            orc.chain_process(eng, eq, l, r, eq_enable=True, gain=1.0)

    t0 = time.perf_counter()
    run_stream(0)
    single = sample_frames / (time.perf_counter() - t0)
    model, total, cores = cpu_info()
    done = [0] * cores
    deadline = time.perf_counter() + target_seconds

    def worker(tid: int) -> None:
        k = 0
        while True:
            run_stream(1 + tid + cores * k)           # ctypes releases the GIL inside the oracle
            k += 1
            done[tid] = k
            if time.perf_counter() >= deadline:
                return

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    tm = time.perf_counter() - t0
    # the plugin's own call on BASELINE configs[3]'s response, one thread: one engine, 4 x 16 384 taps (32 partitions per path,
    # convolution.rs:261-273), 1 024-frame process_block -- beside long_ir.plugin_call of the GPU record
    long_call = None
    try:
        irs_l = synth.hrir_set(16384)
        eng = orc.ConvolutionEngine()
        for p in range(4):
            eng.set_ir(p, irs_l[p])
        xl = synth.white_noise([7], 1024 * 8)[0]
        ts = []
        for c in range(120):
            l = xl[0, (c & 7) * 1024:((c & 7) + 1) * 1024]
            r = xl[1, (c & 7) * 1024:((c & 7) + 1) * 1024]
            t1 = time.perf_counter()
            eng.process_block(l, r)
            ts.append(time.perf_counter() - t1)
        ts = sorted(ts[40:])
        long_call = {"workload": "1 engine, 4 x 16384 taps, 1024-frame process_block, one thread",
                     "us_per_call_median": round(ts[len(ts) // 2] * 1e6, 1), "us_per_call_min": round(ts[0] * 1e6, 1)}
    except Exception as e:      # noqa: BLE001
        long_call = {"error": str(e)[:200]}
    orc.use_library(None)
    return {
        "long_ir_plugin_call": long_call,
        "value": round(sum(done) * sample_frames / tm / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "cores_total": total,
        "cpu_model": model,
        "kind": "port",
        "sample": f"{cores} threads, {sum(done)} whole streams x {sample_frames} frames in {tm:.1f} s "
                  f"(4x{TAPS}-tap + 10-band EQ, host block 1024), oracle/ohs_oracle.c {flags}",
        "single_thread_value": round(single / 1e6, 4),
        "x_realtime_per_core": round(single / FS, 1),
    }


def load_traffic(kernel: str):
    """Fallback for roofline.traffic: HBM bytes per launch from the committed rocprofv3 PMC summary
    (profiles/pmc_latest.json, written by tools/pmc_summary.py from two separate --pmc passes with the gfx950
    FETCH_SIZE x2 correction).  The default run replaces it by measure_traffic_live(); `traffic_source` in the
    record says which one it was."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(p) as f:
            d = json.load(f)
        return d.get(kernel, {}).get("hbm_bytes_per_launch"), "profiles/pmc_latest.json (" + str(d.get("_tag", "?")) + ")"
    except Exception:       # noqa: BLE001
        return None, None


def measure_traffic_live(frames: int, streams: int, timeout_s: float = 240.0):
    """HBM bytes per launch per kernel, measured NOW: two child runs of this script (2 steps, no extras) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes and no trace domain, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes; both counters are KiB, and on gfx950 FETCH_SIZE reports half
    of the bytes of a streaming read (x2).  Returns ({kernel: bytes}, source) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "rocprofv3 not on PATH"
    # never nest: if this very process is being profiled, the children would inherit the tool's preload
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ) or \
            "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "running under a profiler: no nested rocprofv3 runs"
    work = tempfile.mkdtemp(prefix="ohs_pmc_", dir="/tmp")
    per = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, ctr)
            # the SAME workload as this run (frames / streams forwarded); a fresh child process, never a re-exec of
            # this GPU-initialised one; its own session, so that a timeout ends the profiled grandchild as well
            cmd = [rp, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "run", "--", sys.executable,
                   os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--steps", "2", "--warmup", "1",
                   "--frames", str(frames), "--streams", str(streams)]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc_child = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)     # the session this Popen created: rocprofv3 + its child
                except ProcessLookupError:
                    pass
                proc.wait()
                return None, f"rocprofv3 --pmc {ctr}: timed out after {timeout_s:.0f} s (process group ended)"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {ctr}: no counter file (exit {rc_child})"
            agg = collections.defaultdict(list)
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == ctr and "ohs::" in row.get("Kernel_Name", ""):
                            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ohs::", "").split("<")[0]
                            agg[name].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                scale = 2048.0 if ctr == "FETCH_SIZE" else 1024.0
                per.setdefault(k, 0.0)
                per[k] += sum(v) / len(v) * scale
    except Exception as e:      # noqa: BLE001 -- an extra must never break the contract line
        return None, f"live PMC pass failed: {str(e)[:120]}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    if not per:
        return None, "no ohs:: kernels in the counter files"
    return {k: int(round(v)) for k, v in per.items()}, "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child runs of 2 steps"


def make_processor(ohs, S, nb, irs, coeffs, enabled, device_index):
    bp = ohs.BatchProcessor(S, num_bands=nb, device=device_index)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i in range(nb):
        bp.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    bp.set_eq_enabled(True)
    bp.set_gain(0.8912509)      # -1 dB output gain (lib.rs:1202-1207)
    return bp


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N child ranks of this script -- fresh
    processes, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would -- relay
    rank 0's single JSON line (its stdout IS this process's stdout; the other ranks' stdout goes to stderr) and exit
    with the worst child status.  This parent imports no torch and never initialises the GPU; nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OHS_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    worst, failed_at = 0, None
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = max(worst, rc if rc > 0 else 128 - rc)
                if failed_at is None:
                    failed_at = time.monotonic()
                    print(f"bench.py: rank {r} exited with status {rc}; giving the other ranks 30 s", file=sys.stderr)
        if failed_at is not None and pending and time.monotonic() - failed_at > 30.0:
            for r in sorted(pending):           # exactly the processes started above, nothing by pattern
                procs[r].kill()
        time.sleep(0.05)
    return worst



def tables_host(ohs, synth):
    """(irs, coeffs [nb, 5], enabled [nb]) of the headline workload"""
    bands = synth.eq_table()
    irs = synth.hrir_set(TAPS)
    coeffs = np.stack([ohs.biquad_coefficients(b.filter_type, FS, b.center_freq, b.q, b.gain_db) for b in bands])
    enabled = np.array([b.enabled for b in bands])
    return irs, coeffs, enabled


def node_batch_main(args) -> int:
    """`--node-batch`: the job through ohs_node_batch_* in ONE process over args.gpus devices.  One JSON line."""
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus:
        print(f"bench.py --node-batch: {args.gpus} GPU(s) asked for, {torch.cuda.device_count()} visible "
              "(the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    from open_headstage_amd.batch import device_pci_bus_id

    n = args.gpus
    frames = args.frames - args.frames % 512
    n_blocks = frames // 512
    strong = args.total_streams > 0
    total_streams = args.total_streams if strong else args.streams * n
    if total_streams < n:
        print(f"bench.py: {total_streams} streams < {n} devices", file=sys.stderr)
        return 2
    irs, coeffs, enabled = tables_host(ohs, synth)
    nbands = len(coeffs)
    nb = ohs.NodeBatchProcessor(total_streams, num_bands=nbands, n_devices=n)
    comm_size, rccl_loaded = nb.rccl_info()
    t0 = time.perf_counter()
    nb.set_tables(irs, coeffs, enabled)         # host -> first device -> ONE grouped ncclBroadcast -> every device
    bcast_ms = (time.perf_counter() - t0) * 1e3
    nb.set_eq_enabled(True)
    nb.set_gain(0.8912509)
    shards = [nb.shard(i) for i in range(n)]
    xs = [synth.white_noise_torch(first, cnt, frames, torch.device("cuda", dev)) for dev, first, cnt in shards]
    ys = [torch.empty_like(x) for x in xs]
    for dev, _, _ in shards:
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        nb.process(xs, ys)
    nb.sync()
    # the timed region: K steps queued on every device by its own host thread, then one wait for all of them;
    # the per-slot device times come from HIP events on the slots' own streams
    t0 = time.perf_counter()
    nb.timer_begin()
    for _ in range(args.steps):
        nb.process(xs, ys)
    slot_ms = nb.timer_end()
    elapsed = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(y).all().item()) for y in ys)
    if os.environ.get("OHS_BENCH_DUMP"):
        for i, (dev, first, cnt) in enumerate(shards):
            np.save(os.environ["OHS_BENCH_DUMP"] + f".slot{i}.npy",
                    np.concatenate([np.array([first, first + cnt], np.float32), ys[i][:, :, :4096].cpu().numpy().ravel()]))
    value = float(total_streams) * frames * args.steps / elapsed / 1e6
    handle0 = nb.device_batch(0)
    whole_model = sum(nb.device_batch(i).algorithmic_bytes(n_blocks) for i in range(n))
    eq_b, conv_b = handle0.kernel_bytes(n_blocks)
    slots = []
    for i, (dev, first, cnt) in enumerate(shards):
        props = torch.cuda.get_device_properties(dev)
        slots.append({"slot": i, "device": dev, "device_name": props.name, "pci_bus_id": device_pci_bus_id(dev),
                      "uuid": str(getattr(props, "uuid", "")) or None, "first_stream": first, "n_streams": cnt,
                      "ms_per_step": round(slot_ms[i] / args.steps, 4)})
    out = {
        "metric": METRIC, "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "mode": "node_batch",
        "config": {
            "workload": ("BASELINE configs[2]: 256 concurrent stereo streams per GPU, 4x512-tap HRIR partitioned FFT conv "
                         "(block 512, FFT 1024, P=1) + 10-band EQ + gain, 48 kHz" if not strong else
                         f"BASELINE configs[4]: {total_streams} streams sharded over {n} GPU(s), same per-stream work"),
            "streams_per_gpu": shards[0][2], "total_streams": total_streams, "frames_per_stream_per_step": frames,
            "taps": TAPS, "eq_bands": nbands,
            "parallelism": f"ONE process, {n} device(s) through ohs_node_batch_* (C ABI): contiguous stream shards, "
                           "tables by one grouped ncclBroadcast inside the library, no data-path collective",
        },
        "x_realtime_per_gpu": round(value * 1e6 / n / FS, 1),
        "hbm_frac_whole_path": round(whole_model * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * n), 5),
        "rccl_comm_size": comm_size, "rccl_loaded": rccl_loaded, "table_broadcast_ms": round(bcast_ms, 3),
        "slots": slots,
        "distinct_devices": len({s_["uuid"] or s_["pci_bus_id"] for s_ in slots}),
        "slowest_slot_ms_per_step": round(max(slot_ms) / args.steps, 4),
        "finite": ok,
    }
    # the dominant kernel of slot 0 (HIP events inside the library, a few steps after the timed region)
    try:
        handle0.set_profiling(True)
        for _ in range(3):
            nb.process(xs, ys)
        nb.sync()
        eq_ms, conv_ms, n_calls, eq_l, conv_l = handle0.profile_read()
        handle0.set_profiling(False)
        if n_calls and eq_l:
            dom_ms = eq_ms / eq_l
            dom_bytes = eq_b * n_calls / eq_l
            ach = dom_bytes / (dom_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "issue", "kernel": "k_eq_ring", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                               "traffic_source": "not measured in node-batch mode (the one-GPU default run measures it live)",
                               "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                               "kernels_ms_per_step": {"k_eq_ring": round(eq_ms / n_calls, 4),
                                                       "k_conv_p1": round(conv_ms / n_calls, 4)},
                               "slot": 0}
    except Exception as e:      # noqa: BLE001 -- an extra must never break the contract line
        out["roofline"] = {"error": str(e)[:200]}
    if not args.no_extras:
        # PCIe-inclusive: the whole job from / to pinned host memory through ohs_node_batch_process_host (every device
        # runs its three-stage pipeline from its own thread).  Bounded at N > 1 (the host buffers hold ALL streams).
        try:
            pf = frames if n == 1 else min(frames, 120 * 512)
            hx = torch.empty((total_streams, 2, pf), dtype=torch.float32).pin_memory()
            for i, (dev, first, cnt) in enumerate(shards):
                hx[first:first + cnt] = xs[i][:, :, :pf].cpu()
            hy = torch.empty_like(hx).pin_memory()
            nb.reset()
            nb.process_host(hx, out=hy)
            t0 = time.perf_counter()
            for _ in range(2):
                nb.process_host(hx, out=hy)
            dt = (time.perf_counter() - t0) / 2
            out["pcie_inclusive"] = {"ms_per_step": round(dt * 1e3, 3), "frames_per_stream": pf,
                                     "Msamples_s": round(total_streams * pf / dt / 1e6, 1),
                                     "GBps_each_way_whole_node": round(total_streams * pf * 8 / dt / 1e9, 2),
                                     "how": "ohs_node_batch_process_host, pinned host buffers, one 3-stream pipeline per device"}
            del hx, hy
        except Exception as e:      # noqa: BLE001
            out["pcie_inclusive"] = {"error": str(e)[:200]}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(frames)
    print(json.dumps(out), flush=True)
    return 0 if ok else 1


def node_batch_leg(n: int, args, frames: int, headline_value, timeout_s: float = 240.0):
    """The node API measured in a CHILD run of this script (`--gpus n --node-batch`, same steps / frames / streams): a
    fresh process, its own session (a timeout ends it and whatever it started), never an exec of this GPU-initialised
    one.  Returns the summary that goes into the record under "node_batch"."""
    import signal
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--node-batch", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--frames", str(frames), "--no-cpu-baseline"]
    cmd += ["--total-streams", str(args.total_streams)] if args.total_streams > 0 else ["--streams", str(args.streams)]
    if n > 1:
        cmd.append("--no-extras")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "OHS_BENCH_DUMP", "TORCHELASTIC_RUN_ID")}
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            so, se = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            try:
                proc.communicate(timeout=10)        # (a child stuck in the driver does not die at once: do not wait for it)
            except subprocess.TimeoutExpired:
                pass
            return {"error": f"child run timed out after {timeout_s:.0f} s (process group ended)"}
        lines = [l for l in so.splitlines() if l.startswith("{")]
        if proc.returncode != 0 or len(lines) != 1:
            return {"error": f"child exit {proc.returncode}: {(se or so)[-300:]}"}
        rec = json.loads(lines[0])
    except Exception as e:      # noqa: BLE001 -- an extra must never break the contract line
        return {"error": str(e)[:200]}
    keep = {k: rec.get(k) for k in ("value", "unit", "ms_per_step", "n_gpus", "steps", "warmup", "scaling", "rccl_comm_size",
                                    "rccl_loaded", "table_broadcast_ms", "slots", "distinct_devices",
                                    "slowest_slot_ms_per_step", "finite", "pcie_inclusive")}
    keep["what"] = ("the same job through ohs_node_batch_* (C ABI: one process, device-resident audio, tables by the library's "
                    "RCCL broadcast), child run `bench.py --gpus %d --node-batch`" % n)
    if headline_value:
        keep["vs_headline"] = round(rec.get("value", 0.0) / headline_value, 4)
    return keep

# one MI355X, ms per headline step by streams per GPU (profiles/r03_step_by_streams.txt, end of round 3): what strong
# scaling of a 2048-stream job must look like, whatever the fabric does -- the step time is one EQ chain's serial time
# up to ~1000 streams per GPU, so fewer streams per GPU do not make a step shorter
STEP_MS_BY_STREAMS = {128: 5.64, 256: 5.66, 512: 5.78, 1024: 6.40, 1536: 8.09, 2048: 10.16, 4096: 20.4}


def expected_step_ms(streams: int) -> float:
    pts = sorted(STEP_MS_BY_STREAMS.items())
    if streams <= pts[0][0]:
        return pts[0][1]
    for (s0, t0), (s1, t1) in zip(pts, pts[1:]):
        if streams <= s1:
            return t0 + (t1 - t0) * (streams - s0) / (s1 - s0)
    return pts[-1][1] * streams / pts[-1][0]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=480256, help="frames per stream per step (multiple of 512)")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU (weak scaling)")
    ap.add_argument("--total-streams", type=int, default=0,
                    help="fix the whole job's stream count instead (strong scaling; BASELINE configs[4]: 2048)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the saturation sweep and the PCIe-inclusive leg")
    ap.add_argument("--pipelined", action="store_true", help="pipeline consecutive steps (ohs_batch_process_deferred)")
    ap.add_argument("--node-batch", action="store_true",
                    help="ONE process, --gpus devices through the C ABI's ohs_node_batch_* (no torch.distributed)")
    ap.add_argument("--no-node-batch-leg", action="store_true",
                    help="skip the child run that measures the node API next to the headline")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="take roofline.traffic from profiles/pmc_latest.json instead of measuring it in two rocprofv3 child runs")
    args = ap.parse_args()

    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.node_batch:
        return node_batch_main(args)            # one process whatever --gpus says; no torch.distributed
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args.gpus)           # (before torch is imported: the parent never touches the GPU)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    # every rank checks the sharding BEFORE the process group exists, so that a bad job size fails on all ranks
    # together instead of leaving the others in a barrier
    if args.total_streams > 0 and args.total_streams < world:
        if rank == 0:
            print(f"bench.py: --total-streams {args.total_streams} < {world} ranks: a rank would have no stream",
                  file=sys.stderr)
        return 2
    if args.total_streams <= 0 and args.streams < 1:
        if rank == 0:
            print("bench.py: --streams must be >= 1", file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    # OHS_BENCH_DIST_BACKEND=gloo: rehearsal of the multi-rank plumbing on a box with fewer GPUs than ranks
    # (ranks share the devices round-robin, the table broadcast goes over gloo on the CPU).  The driver's
    # multi-GPU runs use the default: one rank per GPU, RCCL.
    backend = os.environ.get("OHS_BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # OHS_BENCH_FORCE_DIST=1 exercises the RCCL init + broadcast path at world size 1
    distributed = world > 1 or os.environ.get("OHS_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
    bcast_device = torch.device("cpu") if backend == "gloo" else device

    def barrier():
        if distributed:
            dist.barrier()

    import open_headstage_amd as ohs
    from open_headstage_amd import sharding, synth

    frames = args.frames - args.frames % 512
    n_blocks = frames // 512
    strong = args.total_streams > 0
    total_streams = args.total_streams if strong else args.streams * world
    ids = sharding.shard_range(total_streams, rank, world)       # this rank's contiguous stream ids
    S = len(ids)
    assert S > 0        # (checked above, before the process group was created)
    bands = synth.eq_table()
    nb = len(bands)

    # ---- shared tables: built on rank 0, one broadcast (HRIR set + EQ coefficients) ----
    if rank == 0:
        irs = synth.hrir_set(TAPS)
        coeffs = np.stack([ohs.biquad_coefficients(b.filter_type, FS, b.center_freq, b.q, b.gain_db)
                           for b in bands])
        enabled = np.array([b.enabled for b in bands])
    else:
        irs = coeffs = enabled = None
    if distributed:
        irs, coeffs, enabled = sharding.broadcast_tables(irs, coeffs, enabled, TAPS, nb, bcast_device, src=0)

    bp = make_processor(ohs, S, nb, irs, coeffs, enabled, dev_index)
    x = synth.white_noise_torch(ids.start, S, frames, device)       # synthetic audio generated in HBM
    y = torch.empty_like(x)
    torch.cuda.synchronize()

    # The steps are queued on a stream of their own, not on torch's default (NULL) stream: the NULL stream synchronises
    # implicitly with every blocking stream of the process and its launches cost more (the same job through
    # ohs_node_batch_*, whose devices have their own streams, ran 1.9 % faster than this loop did on the NULL stream:
    # profiles/r04_v1_bench.json).  torch.cuda.synchronize() on both sides of the timed region waits for every stream.
    proc_stream = torch.cuda.Stream(device=device)
    proc_stream.wait_stream(torch.cuda.current_stream(device))
    hs = proc_stream.cuda_stream

    # --pipelined: consecutive steps through ohs_batch_process_deferred (the next step's EQ starts while this
    # step's last convolution chunk finishes); the default is the strict call.
    def step():
        bp.process(x, out=y, hip_stream=hs, deferred=args.pipelined)

    if distributed or world > 1:
        # before anything that could hang (the first multi-rank run happens on the driver's box, unattended): which device this
        # rank sits on, to stderr -- a hang still leaves it behind
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            print(f"[bench rank {rank}/{world}] pid {os.getpid()} local_rank {local_rank} device {dev_index} {pr.name} "
                  f"pci {getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x} "
                  f"uuid {getattr(pr, 'uuid', '')} streams {ids.start}..{ids.stop} backend {backend} "
                  f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}", file=sys.stderr, flush=True)
        except Exception as e:      # noqa: BLE001
            print(f"[bench rank {rank}/{world}] device evidence failed: {e}", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    bp.join(hs)
    torch.cuda.synchronize()
    if distributed or world > 1:
        print(f"[bench rank {rank}/{world}] warm-up done, entering the timed region", file=sys.stderr, flush=True)

    bp.set_profiling(True)
    # one event in front of every step and one behind the last, on the stream the steps are queued on: the median / min / max
    # step beside the contract's mean over the wall clock (a marker packet per 5.6 ms step)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_ev[i].record(proc_stream)
        step()
    bp.join(hs)
    step_ev[args.steps].record(proc_stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)) if args.steps else [0.0]
    eq_ms, conv_ms, n_calls, eq_launches, conv_launches = bp.profile_read()
    try:
        eq_clock_ghz, eq_wave_us = bp.profile_eq_clock()        # the shader clock the EQ's wave 0 really ran at (last launch)
    except Exception:       # noqa: BLE001 -- no ring-form launch (more than 12 bands): the nominal clock is used and named
        eq_clock_ghz, eq_wave_us = None, None
    conv_kernel_of_step, conv_ranges_of_step = bp.last_conv_plan()
    bp.set_profiling(False)

    rank_records = None
    if distributed:
        # proof of placement: what THIS rank ran on, gathered from every rank (the driver's scaling run must show N
        # distinct devices, not N ranks on one)
        props = torch.cuda.get_device_properties(dev_index)
        bus = getattr(props, "pci_bus_id", None)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device_name": props.name,
                "pci_bus_id": None if bus is None else f"{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:"
                                                       f"{getattr(props, 'pci_device_id', 0):02x}",
                "uuid": str(getattr(props, "uuid", "")) or None, "streams": S, "stream_ids": [ids.start, ids.stop],
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "pid": os.getpid()}
        rank_records = [None] * world
        dist.all_gather_object(rank_records, mine)
        t = torch.tensor([elapsed], dtype=torch.float64, device=bcast_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = bool(torch.isfinite(y).all().item())
    if os.environ.get("OHS_BENCH_DUMP"):        # rehearsal test: this rank's outputs, for the bit-for-bit comparison
        np.save(os.environ["OHS_BENCH_DUMP"] + f".rank{rank}.npy",
                np.concatenate([np.array([ids.start, ids.stop], np.float32), y[:, :, :4096].cpu().numpy().ravel()]))
    total_frames = float(total_streams) * frames * args.steps
    value = total_frames / elapsed / 1e6

    out = None
    if rank == 0:
        eq_b, conv_b = bp.kernel_bytes(n_blocks)
        # what served the step's convolution launches: asked of the library (ohs_batch_last_conv_plan), not restated here
        conv_name = KERNEL_OF_FAMILY.get(conv_kernel_of_step, conv_kernel_of_step)
        # <= 12 enabled bands -> the ring form of the EQ kernel (eq_kernels.hip), else the conveyor form
        eq_name = "k_eq_ring" if int(np.count_nonzero(enabled)) <= 12 else "k_eq_pass"
        # per LAUNCH (a step issues several launches of each kernel when EQ and convolution overlap):
        # (average launch ms, design bytes per launch, total ms per step)
        kernels = {
            eq_name: (eq_ms / max(eq_launches, 1), eq_b * n_calls / max(eq_launches, 1), eq_ms / max(n_calls, 1)),
            conv_name: (conv_ms / max(conv_launches, 1), conv_b * n_calls / max(conv_launches, 1),
                        conv_ms / max(n_calls, 1)),
        }
        dom = max(kernels, key=lambda k: kernels[k][2])
        dom_ms, dom_bytes, _ = kernels[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        conv_launch_ms, conv_launch_bytes, _ = kernels[conv_name]
        conv_achieved = conv_launch_bytes / (conv_launch_ms * 1e-3) / 1e9 if conv_launch_ms > 0 else 0.0
        whole_model = bp.algorithmic_bytes(n_blocks)            # SURVEY 8d per-block model (25 112 B per stream-block)
        whole_design = eq_b + conv_b                              # bytes the kernels are built to move
        traffic, traffic_source = load_traffic(dom)
        conv_traffic, _ = load_traffic(conv_name)
        if traffic_source is not None and (world > 1 or args.no_live_pmc or args.no_extras or strong):
            # the counters are collected live only by the default one-GPU run (two rocprofv3 --pmc child runs)
            traffic_source = "FALLBACK, not measured in this run: " + traffic_source
        ns_per_sample = eq_ms / max(n_calls, 1) * 1e6 / frames
        eq_wave_ring = 2 * S <= 2 * torch.cuda.get_device_properties(device).multi_processor_count and frames // 6 >= 8192
        eq_floor = EQ_ISSUE_FLOOR_CYCLES_WAVE_RING if eq_wave_ring else EQ_ISSUE_FLOOR_CYCLES
        eq_is_dom = dom == eq_name
        out = {
            "metric": METRIC,
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_device": {"median": round(step_ms[len(step_ms) // 2], 4), "min": round(step_ms[0], 4),
                                   "max": round(step_ms[-1], 4), "how": f"HIP events around each of the {args.steps} timed steps"},
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: 256 concurrent stereo streams per GPU, 4x512-tap HRIR "
                            "partitioned FFT conv (block 512, FFT 1024, P=1) + 10-band EQ + gain, 48 kHz"
                            if not strong else
                            f"BASELINE configs[4]: {total_streams} streams sharded over {world} GPU(s), same per-stream work",
                "streams_per_gpu": S,
                "total_streams": total_streams,
                "frames_per_stream_per_step": frames,
                "taps": TAPS,
                "eq_bands": nb,
                "parallelism": f"streams sharded x{world}, HRIR+EQ tables broadcast once ({backend})",
                "steps_pipelined": bool(args.pipelined),
            },
            "x_realtime_per_gpu": round(value * 1e6 / world / FS, 1),
            # whole path against the HBM peak: by SURVEY 8d's per-block byte model, and by the bytes the kernels move
            "hbm_frac_whole_path": round(whole_model * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
            "hbm_frac_whole_path_design_bytes": round(whole_design * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
            "roofline": {
                # the EQ is a serial recurrence: its launch time is (instructions per sample) x (issue interval of a
                # lone wave) x samples, whatever the memory system does -- so it is priced against that floor, with
                # the HBM fraction of the contract beside it
                "bound": "issue" if eq_is_dom else "hbm",
                "kernel": dom,
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": int(dom_bytes),
                "avg_launch_ms": round(dom_ms, 4),
                "launches_per_step": {eq_name: eq_launches // max(n_calls, 1),
                                      conv_name: conv_launches // max(n_calls, 1)},
                "kernels_ms_per_step": {k: round(v[2], 4) for k, v in kernels.items()},
                # cycles per sample at the clock the EQ's wave 0 MEASURED in the timed region's last launch (s_memtime against
                # s_memrealtime, ohs_batch_profile_eq_clock), not at the nominal 2.4 GHz
                "issue": {
                    "ns_per_sample": round(ns_per_sample, 3),
                    "shader_clock_GHz": None if eq_clock_ghz is None else round(eq_clock_ghz, 4),
                    "clock_source": "measured: s_memtime / s_memrealtime of wave 0 of the last EQ launch of the timed region"
                                    if eq_clock_ghz else f"NOMINAL {NOMINAL_CLOCK_GHZ} GHz (no ring-form launch to measure)",
                    "cycles_per_sample": round(ns_per_sample * (eq_clock_ghz or NOMINAL_CLOCK_GHZ), 2),
                    "cycles_per_sample_at_2p4GHz": round(ns_per_sample * NOMINAL_CLOCK_GHZ, 2),
                    # k_eq_ring has two forms (same bits); which one a launch takes is launch_eq_pass's rule (eq_kernels.hip), quoted
                    # here because the C ABI has no entry that reports it: one chain per wave up to two chains per CU
                    # (2 x streams <= 2 x CUs) when the launch is 8 192 samples or longer, else four chains per wave
                    "form": "one chain per wave (ring over 64 lanes), 6.04 VALU per sample" if eq_wave_ring
                            else "four chains per wave (16-lane rows), 6.5 VALU per sample",
                    "cycles_per_sample_floor": eq_floor,
                    "frac_of_floor": round(eq_floor / max(ns_per_sample * (eq_clock_ghz or NOMINAL_CLOCK_GHZ), 1e-9), 4),
                } if eq_is_dom else None,
                # the convolution kernel, which runs underneath the EQ here and sets the rate from ~1000 streams up:
                # bytes it really moves (16 B per frame + chunk tails), NOT the per-block model it does not execute;
                # its time includes sharing the chip with the EQ launches
                "conv": {
                    "bound": "hbm",
                    "kernel": conv_name,
                    "ranges_per_stream": conv_ranges_of_step,
                    "achieved": round(conv_achieved, 3),
                    "frac": round(conv_achieved / HBM_PEAK_GBS, 5),
                    "design_bytes_per_launch": int(conv_launch_bytes),
                    "avg_launch_ms": round(conv_launch_ms, 4),
                    "traffic": conv_traffic,
                },
            },
            "finite": ok,
        }
        if rank_records is not None:
            out["launcher"] = os.environ.get("OHS_BENCH_LAUNCHER", "torch.distributed.run or equivalent")
            out["dist_backend"] = backend + (" (= RCCL on ROCm)" if backend == "nccl" else " (rehearsal: ranks may share a GPU)")
            out["rccl_world_size"] = dist.get_world_size()
            out["ranks"] = rank_records
            out["distinct_devices"] = len({r["uuid"] or r["pci_bus_id"] or (r["device_index"],) for r in rank_records})
        if strong:
            # Strong scaling of this job cannot beat the ratio of two ONE-GPU step times: the whole job's streams on one
            # GPU against this rank's share -- per-GPU step time is one EQ chain's serial recurrence up to ~1000 streams
            # per GPU, a property of the IIR, not of the fabric.  Both are measured here: the share is the timed region
            # itself, the whole job a short run on this rank's GPU right after it (3 + 5 steps, HIP events).
            exp = {"why": "per-GPU step time is one EQ chain's serial recurrence (~5.6 ms) up to ~1000 streams per GPU, so "
                          f"{total_streams} streams on {world} GPU(s) = {S} per GPU cannot beat that floor: a property of "
                          "the IIR, not of the fabric (the data path has no collective)"}
            try:
                tn = elapsed / args.steps * 1e3
                if world == 1:
                    t1, src = tn, "this run IS the one-GPU run"
                else:
                    bp1 = make_processor(ohs, total_streams, nb, irs, coeffs, enabled, dev_index)
                    x1 = synth.white_noise_torch(0, total_streams, frames, device)
                    for _ in range(3):
                        bp1.process(x1, out=x1)
                    torch.cuda.synchronize()
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record()
                    for _ in range(5):
                        bp1.process(x1, out=x1)
                    eb.record()
                    torch.cuda.synchronize()
                    t1 = ea.elapsed_time(eb) / 5
                    src = f"measured in this run on rank 0's GPU: {total_streams} streams, 5 steps after 3, HIP events"
                    del bp1, x1
                    torch.cuda.empty_cache()
                exp.update({"x_vs_one_gpu": round(t1 / tn, 2), "ms_per_step_this_share": round(tn, 3),
                            "one_gpu_ms_per_step_whole_job": round(t1, 3), "source": src})
            except Exception as e:      # noqa: BLE001 -- an extra must never break the contract line
                t1, tn = expected_step_ms(total_streams), expected_step_ms(S)
                exp.update({"x_vs_one_gpu": round(t1 / tn, 2), "source": "FALLBACK table of one-box measurements "
                            "(profiles/r03_step_by_streams.txt): " + str(e)[:120]})
            out["expected_strong_scaling"] = exp

    # ---- extras, N = 1 only, after the timed region ----
    if world == 1 and not args.no_extras and not strong:
        # The convolution by itself (BASELINE config 3 without the EQ): 8 steps back to back.  In the headline step
        # it runs in six slices underneath the EQ and `roofline.conv` shows what it gets there; this is the kernel
        # with the chip to itself, against HBM by its design bytes.
        try:
            bp.set_eq_enabled(False)
            # 40 untimed steps first: after the EQ phase (7/8 of the chip idle) the first full-chip steps run 25 %
            # slower than the steady state -- 0.78 ms with 2 warm-up steps, 0.64 with 40 or 200 (clock / power ramp)
            for _ in range(40):
                bp.process(x, out=y)
            torch.cuda.synchronize()
            def timed_steps(k):
                """-> sorted device ms of k steps, each between two events on the launch stream (EQ off: one kernel per step)"""
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
                for i in range(k):
                    evs[i].record()
                    bp.process(x, out=y)
                evs[k].record()
                torch.cuda.synchronize()
                return sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(k))

            ts = timed_steps(32)
            dt = ts[len(ts) // 2] * 1e-3
            _, cb_alone = bp.kernel_bytes(n_blocks)
            fam, ranges = bp.last_conv_plan()
            out["roofline"]["conv_alone"] = {
                "what": "the same batch with the EQ off, out of place: ONE kernel per step, the plan the library picks for such a "
                        "call; 32 steps after 40 warm-up steps, a HIP event between steps: median, with min and max beside it",
                "kernel": KERNEL_OF_FAMILY.get(fam, fam), "family": fam, "ranges_per_stream": ranges,
                "ms_per_step": round(dt * 1e3, 4), "ms_min": round(ts[0], 4), "ms_max": round(ts[-1], 4),
                "Msamples_s": round(args.streams * frames / dt / 1e6, 1),
                "design_bytes_per_step": int(cb_alone),
                "achieved": round(cb_alone / dt / 1e9, 1), "unit": "GB/s",
                "frac": round(cb_alone / dt / 1e9 / HBM_PEAK_GBS, 5),
                "frac_min_max": [round(cb_alone / (ts[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                 round(cb_alone / (ts[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)]}
            # the block-512 plan forced (k_conv_p1: what serves BASELINE configs[2] underneath the EQ) on the same batch
            try:
                bp.set_conv_plan(1)
                for _ in range(10):
                    bp.process(x, out=y)
                torch.cuda.synchronize()
                ts1 = timed_steps(32)
                dt1 = ts1[len(ts1) // 2] * 1e-3
                _, cb1 = bp.kernel_bytes(n_blocks)
                fam1, ranges1 = bp.last_conv_plan()
                out["roofline"]["conv_alone"]["block_512_plan"] = {
                    "kernel": KERNEL_OF_FAMILY.get(fam1, fam1), "family": fam1, "ranges_per_stream": ranges1,
                    "ms_per_step": round(dt1 * 1e3, 4), "ms_min": round(ts1[0], 4), "ms_max": round(ts1[-1], 4),
                    "design_bytes_per_step": int(cb1),
                    "achieved": round(cb1 / dt1 / 1e9, 1), "unit": "GB/s", "frac": round(cb1 / dt1 / 1e9 / HBM_PEAK_GBS, 5)}
            except Exception as e:      # noqa: BLE001
                out["roofline"]["conv_alone"]["block_512_plan"] = {"error": str(e)[:200]}
            bp.set_conv_plan(0)
            bp.set_eq_enabled(True)
            bp.reset()
        except Exception as e:      # noqa: BLE001
            out["roofline"]["conv_alone"] = {"error": str(e)[:200]}
        # PCIe-inclusive: the same step from / to pinned host memory, H2D || kernels || D2H over time chunks
        S = args.streams
        try:
            hx = synth.white_noise_torch(0, S, frames, device).cpu().pin_memory()
            hy = torch.empty_like(hx).pin_memory()
            bp.reset()
            bp.process_host(hx, out=hy)
            t0 = time.perf_counter()
            for _ in range(2):
                bp.process_host(hx, out=hy)
            dt = (time.perf_counter() - t0) / 2
            out["pcie_inclusive"] = {"ms_per_step": round(dt * 1e3, 3),
                                     "Msamples_s": round(S * frames / dt / 1e6, 1),
                                     "GBps_each_way": round(S * frames * 8 / dt / 1e9, 2),
                                     "how": "ohs_batch_process_host, pinned host buffers, 3 streams / 3 slots"}
            del hx, hy
        except Exception as e:      # noqa: BLE001 -- never let an extra break the contract line
            out["pcie_inclusive"] = {"error": str(e)[:200]}
        # (The PCIe leg runs first: on this platform every hipFree in the process re-rolls the rate of two-way host
        # copies -- plain hipMemcpyAsync pairs measured 29 ... 49 GB/s each way after frees of various sizes,
        # tools/pcie_state_exp3.py -- and empty_cache() below is such a free.)
        del x, y
        torch.cuda.empty_cache()
        sat = []
        for S2 in (1024, 2048):
            bp2 = make_processor(ohs, S2, nb, irs, coeffs, enabled, dev_index)
            x2 = synth.white_noise_torch(0, S2, frames, device)
            y2 = torch.empty_like(x2)
            for _ in range(3):          # (untimed: the chip's first full-load steps are slower, see conv_alone)
                bp2.process(x2, out=y2)
            torch.cuda.synchronize()
            sa, sb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            bp2.conv_plan_counts(reset=True)
            sa.record()                 # (ohs_batch_process joins its second stream back into this one before it returns)
            for _ in range(6):
                bp2.process(x2, out=y2)
            sb.record()
            torch.cuda.synchronize()
            plan_counts = {k: v // 6 for k, v in bp2.conv_plan_counts(reset=True).items()}     # per step, by kernel family
            dt = sa.elapsed_time(sb) * 1e-3 / 6
            eb2, cb2 = bp2.kernel_bytes(n_blocks)
            fam2, ranges2 = bp2.last_conv_plan()         # (of the step's LAST convolution launch: the short last time chunk)
            # the same with the block-512 plan of the one-partition convolution forced (ohs_batch_set_conv_plan(1)); from 512
            # streams on the library picks the hop-1536 plan by itself: profiles/r04_conv_plan_ab_run3.txt
            dt_p1 = None
            try:
                bp2.set_conv_plan(1)
                for _ in range(2):
                    bp2.process(x2, out=y2)
                torch.cuda.synchronize()
                sa.record()
                for _ in range(6):
                    bp2.process(x2, out=y2)
                sb.record()
                torch.cuda.synchronize()
                dt_p1 = sa.elapsed_time(sb) * 1e-3 / 6
            except Exception:       # noqa: BLE001
                pass
            sat.append({"streams": S2, "ms_per_step": round(dt * 1e3, 4), "how": "6 steps after 3, HIP events",
                        "conv_plan": "the library's choice", "conv_launches_per_step_by_family": plan_counts,
                        "conv_kernel_of_last_chunk": KERNEL_OF_FAMILY.get(fam2, fam2),
                        "conv_ranges_per_stream_of_last_chunk": ranges2,
                        "ms_per_step_conv_plan_1": None if dt_p1 is None else round(dt_p1 * 1e3, 4),
                        "Msamples_s": round(S2 * frames / dt / 1e6, 1),
                        "hbm_frac_8d_model": round(bp2.algorithmic_bytes(n_blocks) / dt / 1e9 / HBM_PEAK_GBS, 5),
                        "hbm_frac_design_bytes": round((eb2 + cb2) / dt / 1e9 / HBM_PEAK_GBS, 5)})
            del bp2, x2, y2
            torch.cuda.empty_cache()
        out["saturation"] = sat
        # BASELINE configs[3] ("config 4" of SURVEY 8: 64 streams, 4 x 16 384 taps, no EQ), 256-block calls: the reference's
        # block-512 partitioning (32 partitions) and what the library picks for such calls -- out of place the block-8192 kernel (2
        # partitions, transforms and product in one kernel), in place the block-2048 plan (8 partitions, product and inverse
        # transform in one kernel); per-step HIP events, median of 20 after 3
        try:
            S4, taps4, nb4 = 64, 16384, 256
            x4 = synth.white_noise_torch(0, S4, nb4 * 512, device)
            y4 = torch.empty_like(x4)
            irs4 = synth.hrir_set(taps4)
            long_ir = {"workload": f"{S4} streams x {nb4 * 512} frames, 4 x {taps4} taps, EQ off, out of place",
                       "io_floor_bytes": S4 * nb4 * 512 * 16}
            for plan, key in ((1, "block_512_plan"), (0, "library_choice"), (0, "library_choice_in_place")):
                bp4 = ohs.BatchProcessor(S4, num_bands=nb, device=dev_index)
                for p in range(4):
                    bp4.set_ir(p, irs4[p])
                bp4.set_eq_enabled(False)
                bp4.set_conv_plan(plan)
                o4 = x4 if key.endswith("in_place") else y4       # (in place the single-kernel plan cannot serve: block 2048)
                for _ in range(3):
                    bp4.process(x4, out=o4)
                torch.cuda.synchronize()
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
                for i in range(20):
                    evs[i].record()
                    bp4.process(x4, out=o4)
                evs[20].record()
                torch.cuda.synchronize()
                ts4 = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(20))
                med = ts4[10] * 1e-3
                fam4, ranges4 = bp4.last_conv_plan()
                _, cb4 = bp4.kernel_bytes(nb4)
                if fam4 == "block2048" and key.endswith("in_place"):
                    # (ohs_batch_kernel_bytes has no in-place argument and takes an EQ-less call to be out of place; the block-2048
                    # plan's design bytes, as api_batch.hip states them: audio in, ring written, ring read once, audio out, history)
                    cb4 = S4 * nb4 * (4096 + 8192 + 8192 + 4096) + 2 * S4 * 2 * 16384 * 4
                long_ir[key] = {"kernel": KERNEL_OF_FAMILY.get(fam4, fam4), "family": fam4, "ranges_per_stream": ranges4,
                                "ms_per_step": round(med * 1e3, 4), "ms_min": round(ts4[0], 4), "ms_max": round(ts4[-1], 4),
                                "Msamples_s": round(S4 * nb4 * 512 / med / 1e6, 1),
                                "design_bytes_per_step": int(cb4), "achieved": round(cb4 / med / 1e9, 1), "unit": "GB/s",
                                "frac": round(cb4 / med / 1e9 / HBM_PEAK_GBS, 5)}
                del bp4
            del x4, y4
            torch.cuda.empty_cache()
            # the plugin's own use of that response: ONE engine, 1 024-frame process_block through the blocking C ABI (host
            # buffers in and out; ctypes-driven, so ~5 us of interpreter time are in every figure), resident kernel.  Calls
            # 300 us apart are what a real-time host's look like to the engine (theirs are 21 ms apart): the sums over the
            # partitions behind a call are computed between the calls.  Back to back, every call waits for them.
            try:
                import gc
                eng = ohs.ConvolutionEngine.new(dev_index)
                for p in range(4):
                    eng.set_ir(p, irs4[p])
                eng.set_realtime(True)
                xn = synth.white_noise([0], 1024 * 8)[0]
                bl = [(xn[0, c * 1024:(c + 1) * 1024].copy(), xn[1, c * 1024:(c + 1) * 1024].copy()) for c in range(8)]
                ol, orr = np.empty(1024, np.float32), np.empty(1024, np.float32)
                pc = {"workload": f"1 engine, 4 x {taps4} taps, 1024-frame calls, ctypes-driven; resident kernel unless named otherwise"}
                gc.collect()
                gc.disable()
                for key, gap in (("us_per_call_300us_apart", 300e-6), ("us_per_call_back_to_back", 0.0),
                                 ("launch_per_call_us_back_to_back", 0.0)):
                    if key.startswith("launch"):        # real-time mode off: what an offline render through process_block sees
                        eng.set_realtime(False)
                    ts = []
                    for c in range(340):
                        if gap:
                            tw = time.perf_counter() + gap
                            while time.perf_counter() < tw:
                                pass
                        l, r = bl[c & 7]
                        t0 = time.perf_counter()
                        eng.process_block(l, r, ol, orr)
                        ts.append(time.perf_counter() - t0)
                    ts = sorted(ts[40:])
                    pc[key] = {"median": round(ts[len(ts) // 2] * 1e6, 1), "p99": round(ts[int(len(ts) * 0.99)] * 1e6, 1)}
                gc.enable()
                eng.set_realtime(False)
                del eng
                long_ir["plugin_call"] = pc
            except Exception as e:      # noqa: BLE001
                long_ir["plugin_call"] = {"error": str(e)[:200]}
            out["long_ir"] = long_ir
        except Exception as e:      # noqa: BLE001 -- never let an extra break the contract line
            out["long_ir"] = {"error": str(e)[:200]}
        # roofline.traffic measured now (two rocprofv3 --pmc child runs of this script); the committed summary
        # stays the fallback and is named as such
        if not args.no_live_pmc:
            live, src = measure_traffic_live(frames, args.streams)
            if live is not None and out is not None:
                rf = out["roofline"]
                if rf.get("kernel") in live:
                    rf["traffic"] = live[rf["kernel"]]
                    rf["traffic_source"] = src
                if "conv" in rf and rf["conv"].get("kernel") in live:
                    rf["conv"]["traffic"] = live[rf["conv"]["kernel"]]
            elif out is not None:
                out["roofline"]["traffic_live_error"] = src
    # ---- the node API (ohs_node_batch_*) next to the headline: a child run over the same devices ----
    want_leg = not args.no_node_batch_leg and not args.no_extras
    # (gloo rehearsal on a box with fewer GPUs than ranks: the child takes the devices there are)
    leg_n = world if backend != "gloo" else min(world, torch.cuda.device_count())
    if want_leg and world == 1 and rank == 0:
        torch.cuda.empty_cache()
        out["node_batch"] = node_batch_leg(1, args, frames, out["value"])
    elif want_leg and world > 1:
        # rank 0 runs the child over all `world` devices; the other ranks wait for it on the CPU (the rendezvous
        # store -- a barrier would park a spinning RCCL kernel on every GPU the child is about to time)
        try:
            del x, y
            torch.cuda.empty_cache()
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                try:
                    # (first contact of N communicators in one process happens on the driver's box, unattended: a tighter
                    # bound than at N = 1 -- the leg is an extra, the contract line must not wait minutes for it)
                    out["node_batch"] = node_batch_leg(leg_n, args, frames, out["value"], timeout_s=150.0)
                finally:
                    store.set("ohs_node_batch_leg_done", "1")
            else:
                import datetime
                store.wait(["ohs_node_batch_leg_done"], datetime.timedelta(seconds=200))
        except Exception as e:      # noqa: BLE001 -- an extra must never break the contract line
            if rank == 0 and out is not None:
                out.setdefault("node_batch", {"error": str(e)[:200]})
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
