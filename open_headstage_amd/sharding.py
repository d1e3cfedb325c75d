"""Multi-GPU plan for the batch mode.

Streams are independent units, so the data path has NO collective: rank r owns a
contiguous range of stream ids.  The only exchange is one broadcast, at set-up,
of the shared tables (time-domain HRIR set + EQ coefficient table) from rank 0 --
`torch.distributed` over RCCL/xGMI on GPUs ("nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_streams: int, rank: int, world_size: int) -> range:
    """Contiguous, balanced ranges; the first (n_streams % world_size) ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    q, r = divmod(n_streams, world_size)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def pack_tables(irs, eq_coeffs, eq_enabled) -> np.ndarray:
    """Flatten [4 x taps] IRs + [nb x 5] coefficients + [nb] enables into one f32 vector
    (one broadcast instead of three; the payload is <= ~260 KiB, latency-bound)."""
    irs = [np.ascontiguousarray(h, np.float32).ravel() for h in irs]
    taps = irs[0].size
    if any(h.size != taps for h in irs) or len(irs) != 4:
        raise ValueError("need four IRs of equal length")
    c = np.ascontiguousarray(eq_coeffs, np.float32).reshape(-1, 5)
    en = np.ascontiguousarray(eq_enabled, np.float32).ravel()
    if en.size != c.shape[0]:
        raise ValueError("eq_enabled / eq_coeffs mismatch")
    head = np.array([taps, c.shape[0]], np.float32)
    return np.concatenate([head, *irs, c.ravel(), en])


def unpack_tables(buf: np.ndarray):
    buf = np.asarray(buf, np.float32)
    taps, nb = int(buf[0]), int(buf[1])
    o = 2
    irs = [buf[o + p * taps: o + (p + 1) * taps].copy() for p in range(4)]
    o += 4 * taps
    c = buf[o: o + 5 * nb].reshape(nb, 5).copy()
    o += 5 * nb
    en = buf[o: o + nb] != 0
    return irs, c, en


def table_numel(taps: int, nb: int) -> int:
    return 2 + 4 * taps + 6 * nb


def broadcast_tables(irs, eq_coeffs, eq_enabled, taps: int, nb: int, device, src: int = 0):
    """Rank `src` supplies the tables (others may pass None); everyone returns them.

    With the "nccl" backend and a CUDA `device` this is one RCCL broadcast over xGMI."""
    import torch
    import torch.distributed as dist

    n = table_numel(taps, nb)
    if dist.get_rank() == src:
        t = torch.from_numpy(pack_tables(irs, eq_coeffs, eq_enabled)).to(device)
        assert t.numel() == n
    else:
        t = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    return unpack_tables(t.cpu().numpy())
