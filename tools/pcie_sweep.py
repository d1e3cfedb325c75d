import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import open_headstage_amd as ohs
from open_headstage_amd import synth
S, frames = 256, 480256
bp = ohs.BatchProcessor(S, num_bands=10)
irs = synth.hrir_set(512)
for p in range(4): bp.set_ir(p, irs[p])
for i, b in enumerate(synth.eq_table()): bp.update_band_coeffs(i, synth.FS, b)
bp.set_eq_enabled(True)
hx = synth.white_noise_torch(0, S, frames, torch.device("cuda:0")).cpu().pin_memory()
hy = torch.empty_like(hx).pin_memory()
for cb in [int(v) for v in os.environ.get("OHS_SWEEP", "24,48,96,192,320,938").split(",")]:
    bp.process_host(hx, out=hy, chunk_blocks=cb)
    t0 = time.perf_counter()
    for _ in range(3): bp.process_host(hx, out=hy, chunk_blocks=cb)
    dt = (time.perf_counter() - t0) / 3
    print(f"chunk_blocks {cb}: {dt*1e3:.2f} ms  {S*frames*8/dt/1e9:.1f} GB/s each way", flush=True)
# plain copies for reference
d = torch.empty_like(hx, device="cuda")
torch.cuda.synchronize(); t0=time.perf_counter(); d.copy_(hx, non_blocking=True); torch.cuda.synchronize(); print("H2D alone GB/s", hx.numel()*4/(time.perf_counter()-t0)/1e9)
t0=time.perf_counter(); hy.copy_(d, non_blocking=True); torch.cuda.synchronize(); print("D2H alone GB/s", hx.numel()*4/(time.perf_counter()-t0)/1e9)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d2 = torch.empty_like(d)
torch.cuda.synchronize(); t0=time.perf_counter()
with torch.cuda.stream(s1): d.copy_(hx, non_blocking=True)
with torch.cuda.stream(s2): hy.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); print("H2D || D2H GB/s each", hx.numel()*4/(time.perf_counter()-t0)/1e9)
