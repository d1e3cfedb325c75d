#!/bin/bash
# Board power and clocks while the P = 1 convolution runs back to back (rocm-smi sampled in the background):
#   gpurun -- 'bash tools/power_probe.sh > gpurun_out/power_probe.txt 2>&1'
python3 - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import open_headstage_amd as ohs
from open_headstage_amd import synth
S, frames = 256, 480256
bp = ohs.BatchProcessor(S, num_bands=10)
for p, h in enumerate(synth.hrir_set(512)):
    bp.set_ir(p, h)
bp.set_eq_enabled(False)
x = synth.white_noise_torch(0, S, frames, torch.device("cuda:0")); y = torch.empty_like(x)
t_end = time.time() + 12
n = 0
while time.time() < t_end:
    for _ in range(50):
        bp.process(x, out=y)
    torch.cuda.synchronize(); n += 50
print("steps", n, flush=True)
PY
pid=$!
sleep 5
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|temp" | head -12; echo ---; sleep 1; done
wait $pid
rocm-smi --showmaxpower 2>&1 | grep -i power | head -3
