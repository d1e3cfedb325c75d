"""Offline many-stream batch mode (BASELINE.json north_star) over the C ABI.

`n_streams` independent stereo streams share one HRIR set and one EQ table; all
per-stream state lives in HBM / registers of the GPU.  Audio is exchanged as
device tensors shaped [stream, channel(2), frame] -- torch is used only as the
owner of device memory and streams; every sample is computed by the HIP
kernels behind libohs_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._ffi import check, fp, lib
from .dsp import BLOCK_SIZE, BandConfig


class BatchProcessor:
    def __init__(self, n_streams: int, num_bands: int = 10, device: int = 0, library=None):
        """library: the CDLL the handle lives in (default: the product library; _ffi.experiments_lib() for plan
        overrides through _ffi.set_tuning)"""
        self.n_streams = int(n_streams)
        self.num_bands = int(num_bands)
        self.device = int(device)
        self._L = library
        h = C.c_void_p()
        self._check(self._lib.ohs_batch_create(self.device, self.n_streams, self.num_bands, C.byref(h)))
        self._h = h

    @property
    def _lib(self):
        return getattr(self, "_L", None) or lib()

    def _check(self, status: int) -> None:
        check(status, self._lib)

    # -- shared tables -----------------------------------------------------------
    def set_ir(self, path, ir_data) -> None:
        ir = np.ascontiguousarray(ir_data, dtype=np.float32).ravel()
        self._check(self._lib.ohs_batch_set_ir(self._h, int(path), ir.ctypes.data_as(fp), ir.size))

    def set_speakers(self, sofa, az_l: float = -30.0, el_l: float = 0.0, az_r: float = 30.0, el_r: float = 0.0,
                     radius_m: float = 1.0, fs: float = 0.0) -> int:
        """speaker angles (the plugin's: degrees, azimuth positive to the right) -> the four shared impulse responses;
        -> bit mask of the paths that were re-loaded (ohs_batch_set_speakers)"""
        m = C.c_uint()
        self._check(self._lib.ohs_batch_set_speakers(self._h, sofa._h, az_l, el_l, az_r, el_r, radius_m, fs, C.byref(m)))
        return int(m.value)

    def update_band_coeffs(self, band_idx: int, sample_rate: float, config: BandConfig) -> None:
        self._check(self._lib.ohs_batch_update_eq_band(self._h, int(band_idx), sample_rate,
                                             int(config.filter_type), config.center_freq, config.q,
                                             config.gain_db, int(bool(config.enabled))))

    def set_band_coeffs(self, band_idx: int, coeffs, enabled: bool) -> None:
        c = np.ascontiguousarray(coeffs, dtype=np.float32).ravel()
        if c.size != 5:
            raise ValueError("coeffs must be [b0, b1, b2, a1, a2]")
        self._check(self._lib.ohs_batch_set_eq_band_coeffs(self._h, int(band_idx), c.ctypes.data_as(fp),
                                                 int(bool(enabled))))

    def set_stream_band_coeffs(self, stream: int, band_idx: int, coeffs, enabled: bool) -> None:
        """stream `stream`'s own band (every plugin instance of the reference owns its bands, parametric_eq.rs:125-129)"""
        c = np.ascontiguousarray(coeffs, dtype=np.float32).ravel()
        if c.size != 5:
            raise ValueError("coeffs must be [b0, b1, b2, a1, a2]")
        self._check(self._lib.ohs_batch_set_stream_eq_band_coeffs(self._h, int(stream), int(band_idx), c.ctypes.data_as(fp),
                                                                  int(bool(enabled))))

    def update_stream_band_coeffs(self, stream: int, band_idx: int, sample_rate: float, config: BandConfig) -> None:
        self._check(self._lib.ohs_batch_update_stream_eq_band(self._h, int(stream), int(band_idx), sample_rate,
                                                              int(config.filter_type), config.center_freq, config.q,
                                                              config.gain_db, int(bool(config.enabled))))

    def share_eq_table(self) -> None:
        """back to the one shared EQ table (ohs_batch_share_eq_table)"""
        self._check(self._lib.ohs_batch_share_eq_table(self._h))

    def set_eq_enabled(self, eq_enable: bool) -> None:
        self._check(self._lib.ohs_batch_set_eq_enabled(self._h, int(bool(eq_enable))))

    def set_eq_exact_specials(self, enable: bool) -> None:
        self._check(self._lib.ohs_batch_set_eq_exact_specials(self._h, int(bool(enable))))

    def set_flush_denormals(self, mode: int) -> None:
        """0 = IEEE (default), 1 = FTZ, 2 = FTZ | DAZ for the EQ and the convolution (ohs_batch_set_flush_denormals)"""
        self._check(self._lib.ohs_batch_set_flush_denormals(self._h, int(mode)))

    def set_gain(self, gain: float) -> None:
        self._check(self._lib.ohs_batch_set_gain(self._h, float(gain)))

    def set_conv_plan(self, plan: int) -> None:
        """1 = block 512 / FFT 1024 (the reference's blocking; bit-stable for taps <= 512), 2 = the large-transform plans
        (taps <= 512: hop 1536 / FFT 2048; taps > 512: block 2048 / FFT 4096, or block 8192 / FFT 16384 for long out-of-place
        calls on taps <= 16384), 0 = the library's choice
        (ohs_batch_set_conv_plan)"""
        self._check(self._lib.ohs_batch_set_conv_plan(self._h, int(plan)))

    CONV_KERNELS = {0: "none", 1: "block512_p1", 2: "hop1536_p1", 3: "block512_tp", 4: "block2048", 5: "sequential", 6: "block8192"}

    def last_conv_plan(self):
        """(kernel family name, ranges per stream) of the most recent convolution launch (ohs_batch_last_conv_plan)"""
        k, r = C.c_int(), C.c_int()
        self._check(self._lib.ohs_batch_last_conv_plan(self._h, C.byref(k), C.byref(r)))
        return self.CONV_KERNELS.get(int(k.value), str(k.value)), int(r.value)

    def conv_plan_counts(self, reset: bool = False) -> dict:
        """convolution launch sequences per kernel family since creation / the last reset read (ohs_batch_conv_plan_counts)"""
        c = (C.c_uint64 * 8)()
        self._check(self._lib.ohs_batch_conv_plan_counts(self._h, c, int(bool(reset))))
        return {self.CONV_KERNELS[k]: int(c[k]) for k in range(1, 7) if c[k]}

    def reset(self) -> None:
        self._check(self._lib.ohs_batch_reset(self._h))

    # -- processing --------------------------------------------------------------
    def process_ptr(self, d_in: int, d_out: int, n_blocks: int, stream_stride: int,
                    channel_stride: int, hip_stream: int = 0, deferred: bool = False) -> None:
        fn = self._lib.ohs_batch_process_deferred if deferred else self._lib.ohs_batch_process
        self._check(fn(self._h, C.c_void_p(d_in), C.c_void_p(d_out), int(n_blocks), int(stream_stride),
                 int(channel_stride), C.c_void_p(hip_stream) if hip_stream else None))

    def join(self, hip_stream: int | None = None) -> None:
        """Make `hip_stream` (default: torch's current stream) wait for a pending deferred call."""
        if hip_stream is None:
            import torch
            hip_stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._lib.ohs_batch_join(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def process(self, x, out=None, hip_stream: int | None = None, deferred: bool = False):
        """x, out: torch.float32 CUDA tensors [n_streams, 2, frames], frames % 512 == 0.
        deferred=True: `out` is complete on the stream only after join() / sync() (ohs_batch_process_deferred)."""
        import torch
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()):
            raise TypeError("x must be a contiguous float32 CUDA tensor [streams, 2, frames]")
        S, ch, frames = x.shape
        if S != self.n_streams or ch != 2 or frames % BLOCK_SIZE:
            raise ValueError(f"expected [{self.n_streams}, 2, k*{BLOCK_SIZE}], got {tuple(x.shape)}")
        if x.device.index != self.device:
            raise ValueError("tensor is on a different device than the BatchProcessor")
        if out is None:
            out = torch.empty_like(x)
        elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous() or out.device != x.device:
            raise ValueError("out must match x")
        if hip_stream is None:
            hip_stream = torch.cuda.current_stream(x.device).cuda_stream
        self.process_ptr(x.data_ptr(), out.data_ptr(), frames // BLOCK_SIZE, 2 * frames, frames,
                         hip_stream, deferred)
        return out

    def process_host(self, x, out=None, chunk_blocks: int = 0):
        """x, out: HOST float32 tensors (torch, ideally pinned) or numpy arrays [n_streams, 2, frames]; blocking,
        copy-in / kernels / copy-out pipelined over time chunks (ohs_batch_process_host)."""
        is_np = isinstance(x, np.ndarray)
        if out is None:
            out = np.empty_like(x) if is_np else x.new_empty(x.shape).pin_memory()
        S, ch, frames = x.shape
        if S != self.n_streams or ch != 2 or frames % BLOCK_SIZE:
            raise ValueError(f"expected [{self.n_streams}, 2, k*{BLOCK_SIZE}], got {tuple(x.shape)}")
        if tuple(out.shape) != tuple(x.shape):
            raise ValueError("out must match x")
        if is_np:
            if x.dtype != np.float32 or not x.flags.c_contiguous or out.dtype != np.float32 or not out.flags.c_contiguous:
                raise TypeError("numpy buffers must be contiguous float32")
            pi, po = x.ctypes.data, out.ctypes.data
        else:
            import torch
            if x.is_cuda or out.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or not out.is_contiguous():
                raise TypeError("x / out must be contiguous float32 HOST tensors")
            pi, po = x.data_ptr(), out.data_ptr()
        self._check(self._lib.ohs_batch_process_host(self._h, C.c_void_p(pi), C.c_void_p(po), frames // BLOCK_SIZE,
                                           2 * frames, frames, int(chunk_blocks)))
        return out

    def sync(self, hip_stream: int = 0) -> None:
        self._check(self._lib.ohs_batch_sync(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_profiling(self, enable: bool) -> None:
        self._check(self._lib.ohs_batch_set_profiling(self._h, int(bool(enable))))

    def profile_read(self):
        """(eq_ms, conv_ms, n_calls, eq_launches, conv_launches) since the last read; waits for the events."""
        a, b, n, ne, nc = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.ohs_batch_profile_read(self._h, C.byref(a), C.byref(b), C.byref(n), C.byref(ne),
                                           C.byref(nc)))
        return a.value, b.value, int(n.value), int(ne.value), int(nc.value)

    def profile_eq_clock(self):
        """(shader clock in GHz, lifetime in us) of wave 0 of the most recent ring-form EQ launch (ohs_batch_profile_eq_clock)"""
        g, u = C.c_double(), C.c_double()
        self._check(self._lib.ohs_batch_profile_eq_clock(self._h, C.byref(g), C.byref(u)))
        return g.value, u.value

    def kernel_bytes(self, n_blocks: int):
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.ohs_batch_kernel_bytes(self._h, int(n_blocks), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def algorithmic_bytes(self, n_blocks: int) -> int:
        v = C.c_uint64()
        self._check(self._lib.ohs_batch_algorithmic_bytes(self._h, int(n_blocks), C.byref(v)))
        return int(v.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self, "_borrowed", None) is None:
            try:
                self._lib.ohs_batch_destroy(h)
            except Exception:
                pass


def device_pci_bus_id(device: int) -> str:
    """'domain:bus:device.function' of HIP device `device` (ohs_device_pci_bus_id)"""
    buf = C.create_string_buffer(64)
    check(lib().ohs_device_pci_bus_id(int(device), buf, 64))
    return buf.value.decode()


class NodeBatchProcessor:
    """The batch mode over the GPUs of one node in ONE process (ohs_node_batch_*, include/ohs_hip.h): contiguous
    stream-id shards, one per-device batch each, the shared tables carried from the first device to the others by one
    RCCL broadcast inside the library.  Host mirror of what a Rust host binds."""

    def __init__(self, streams_total: int, num_bands: int = 10, devices=None, n_devices: int | None = None, library=None):
        if devices is None:
            if n_devices is None:
                raise ValueError("give devices=[...] or n_devices=N")
            arr, n = None, int(n_devices)
        else:
            devices = [int(d) for d in devices]
            n = len(devices)
            arr = (C.c_int * n)(*devices)
        self.streams_total = int(streams_total)
        self.num_bands = int(num_bands)
        self.n_devices = n
        self._L = library
        h = C.c_void_p()
        self._check(self._lib.ohs_node_batch_create(n, arr, self.streams_total, self.num_bands, C.byref(h)))
        self._h = h

    @property
    def _lib(self):
        return getattr(self, "_L", None) or lib()

    def _check(self, status: int) -> None:
        check(status, self._lib)

    def shard(self, slot: int):
        """(HIP device index, first stream id, number of streams) of device slot `slot`"""
        d, f, c = C.c_int(), C.c_size_t(), C.c_size_t()
        self._check(self._lib.ohs_node_batch_shard(self._h, int(slot), C.byref(d), C.byref(f), C.byref(c)))
        return int(d.value), int(f.value), int(c.value)

    def set_tables(self, irs, eq_coeffs=None, eq_enabled=None) -> None:
        """irs: four impulse responses (an empty one mutes its path); eq_coeffs [num_bands, 5] + eq_enabled [num_bands]
        or both None.  One broadcast carries everything."""
        arrs = [np.ascontiguousarray(h, dtype=np.float32).ravel() for h in irs]
        if len(arrs) != 4:
            raise ValueError("need four impulse responses")
        ptrs = (fp * 4)(*[a.ctypes.data_as(fp) if a.size else None for a in arrs])
        lens = (C.c_size_t * 4)(*[a.size for a in arrs])
        if eq_coeffs is None:
            self._check(self._lib.ohs_node_batch_set_tables(self._h, ptrs, lens, None, None))
            return
        c = np.ascontiguousarray(eq_coeffs, dtype=np.float32).reshape(self.num_bands, 5)
        en = np.ascontiguousarray(np.asarray(eq_enabled).astype(np.int32)).reshape(self.num_bands)
        self._check(self._lib.ohs_node_batch_set_tables(self._h, ptrs, lens, c.ctypes.data_as(fp),
                                              en.ctypes.data_as(C.POINTER(C.c_int))))

    def set_ir(self, path, ir_data) -> None:
        ir = np.ascontiguousarray(ir_data, dtype=np.float32).ravel()
        self._check(self._lib.ohs_node_batch_set_ir(self._h, int(path), ir.ctypes.data_as(fp) if ir.size else None, ir.size))

    def set_speakers(self, sofa, az_l: float = -30.0, el_l: float = 0.0, az_r: float = 30.0, el_r: float = 0.0,
                     radius_m: float = 1.0, fs: float = 0.0) -> int:
        """speaker angles -> the four shared impulse responses, one broadcast per path that changed
        (ohs_node_batch_set_speakers); -> bit mask of the paths that were re-loaded"""
        m = C.c_uint()
        self._check(self._lib.ohs_node_batch_set_speakers(self._h, sofa._h, az_l, el_l, az_r, el_r, radius_m, fs, C.byref(m)))
        return int(m.value)

    def set_band_coeffs(self, band_idx: int, coeffs, enabled: bool) -> None:
        c = np.ascontiguousarray(coeffs, dtype=np.float32).ravel()
        if c.size != 5:
            raise ValueError("coeffs must be [b0, b1, b2, a1, a2]")
        self._check(self._lib.ohs_node_batch_set_eq_band_coeffs(self._h, int(band_idx), c.ctypes.data_as(fp), int(bool(enabled))))

    def set_stream_band_coeffs(self, stream: int, band_idx: int, coeffs, enabled: bool) -> None:
        """`stream` is the job's stream id (ohs_node_batch_set_stream_eq_band_coeffs)"""
        c = np.ascontiguousarray(coeffs, dtype=np.float32).ravel()
        if c.size != 5:
            raise ValueError("coeffs must be [b0, b1, b2, a1, a2]")
        self._check(self._lib.ohs_node_batch_set_stream_eq_band_coeffs(self._h, int(stream), int(band_idx),
                                                                       c.ctypes.data_as(fp), int(bool(enabled))))

    def share_eq_table(self) -> None:
        self._check(self._lib.ohs_node_batch_share_eq_table(self._h))

    def set_eq_enabled(self, eq_enable: bool) -> None:
        self._check(self._lib.ohs_node_batch_set_eq_enabled(self._h, int(bool(eq_enable))))

    def set_gain(self, gain: float) -> None:
        self._check(self._lib.ohs_node_batch_set_gain(self._h, float(gain)))

    def set_conv_plan(self, plan: int) -> None:
        self._check(self._lib.ohs_node_batch_set_conv_plan(self._h, int(plan)))

    def reset(self) -> None:
        self._check(self._lib.ohs_node_batch_reset(self._h))

    def rccl_info(self):
        """(size of the RCCL communicator, librccl.so loaded)"""
        n, ok = C.c_int(), C.c_int()
        self._check(self._lib.ohs_node_batch_rccl_info(self._h, C.byref(n), C.byref(ok)))
        return int(n.value), bool(ok.value)

    def device_batch(self, slot: int) -> "BatchProcessor":
        """the per-device batch of slot `slot` as a BatchProcessor VIEW (owned by the node batch: profiling, byte
        models and direct ohs_batch_process calls on that device)"""
        dev, _, cnt = self.shard(slot)
        h = C.c_void_p()
        self._check(self._lib.ohs_node_batch_device_batch(self._h, int(slot), C.byref(h)))
        v = BatchProcessor.__new__(BatchProcessor)
        v._L = self._L
        v.n_streams, v.num_bands, v.device = cnt, self.num_bands, dev
        v._h = h
        v._borrowed = self         # keeps the owner alive; __del__ must not destroy the handle
        return v

    # -- device-resident data path ---------------------------------------------------------
    def process_ptrs(self, d_in, d_out, n_blocks: int, stream_stride: int, channel_stride: int) -> None:
        """d_in / d_out: one device address per slot (ohs_node_batch_process).  Returns when every device has queued
        its work; sync() waits for it."""
        n = self.n_devices
        if len(d_in) != n or len(d_out) != n:
            raise ValueError("one pointer per device slot")
        ai = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in d_in])
        ao = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in d_out])
        self._check(self._lib.ohs_node_batch_process(self._h, ai, ao, int(n_blocks), int(stream_stride), int(channel_stride)))

    def process(self, xs, outs=None):
        """xs: one float32 CUDA tensor [n_streams(slot), 2, frames] per slot, each on its slot's device; outs likewise
        (None = in place).  Asynchronous: sync() before reading.  torch is only the owner of the device memory here --
        nothing is queued on torch's streams, so the tensors must be complete (torch.cuda.synchronize) beforehand."""
        if outs is None:
            outs = xs
        frames = None
        for slot, (x, y) in enumerate(zip(xs, outs)):
            dev, _, cnt = self.shard(slot)
            if not (x.is_cuda and x.dim() == 3 and x.is_contiguous() and x.device.index == dev and y.shape == x.shape
                    and y.is_contiguous() and y.device == x.device and str(x.dtype) == "torch.float32" and y.dtype == x.dtype):
                raise TypeError(f"slot {slot}: need contiguous float32 CUDA tensors on device {dev}")
            if x.shape[0] != cnt or x.shape[1] != 2 or x.shape[2] % BLOCK_SIZE:
                raise ValueError(f"slot {slot}: expected [{cnt}, 2, k*{BLOCK_SIZE}], got {tuple(x.shape)}")
            if frames is None:
                frames = x.shape[2]
            elif frames != x.shape[2]:
                raise ValueError("every slot must hold the same number of frames")
        self.process_ptrs([x.data_ptr() for x in xs], [y.data_ptr() for y in outs], frames // BLOCK_SIZE, 2 * frames, frames)
        return outs

    def sync(self) -> None:
        self._check(self._lib.ohs_node_batch_sync(self._h))

    def stream(self, slot: int) -> int:
        """the hipStream_t (as an integer) the slot's work is queued on"""
        p = C.c_void_p()
        self._check(self._lib.ohs_node_batch_stream(self._h, int(slot), C.byref(p)))
        return int(p.value or 0)

    def timer_begin(self) -> None:
        self._check(self._lib.ohs_node_batch_timer_begin(self._h))

    def timer_end(self):
        """-> device milliseconds per slot of what was queued since timer_begin (waits for it)"""
        ms = (C.c_float * self.n_devices)()
        self._check(self._lib.ohs_node_batch_timer_end(self._h, ms))
        return [float(v) for v in ms]

    def process_host(self, x, out=None, chunk_blocks: int = 0):
        """x, out: contiguous float32 HOST buffers [streams_total, 2, frames] (numpy, or torch tensors -- pinned ones
        must be portable across devices); every device runs its shard from a thread of its own.  Blocking."""
        is_np = isinstance(x, np.ndarray)
        if out is None:
            out = np.empty_like(x) if is_np else x.new_empty(x.shape)
        S, ch, frames = x.shape
        if S != self.streams_total or ch != 2 or frames % BLOCK_SIZE:
            raise ValueError(f"expected [{self.streams_total}, 2, k*{BLOCK_SIZE}], got {tuple(x.shape)}")
        if tuple(out.shape) != tuple(x.shape):
            raise ValueError("out must match x")
        if is_np:
            if x.dtype != np.float32 or not x.flags.c_contiguous or out.dtype != np.float32 or not out.flags.c_contiguous:
                raise TypeError("numpy buffers must be contiguous float32")
            pi, po = x.ctypes.data, out.ctypes.data
        else:
            import torch
            if x.is_cuda or out.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or not out.is_contiguous():
                raise TypeError("x / out must be contiguous float32 HOST tensors")
            pi, po = x.data_ptr(), out.data_ptr()
        self._check(self._lib.ohs_node_batch_process_host(self._h, C.c_void_p(pi), C.c_void_p(po), frames // BLOCK_SIZE,
                                                2 * frames, frames, int(chunk_blocks)))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.ohs_node_batch_destroy(h)
            except Exception:
                pass
