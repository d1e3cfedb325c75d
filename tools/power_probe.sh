#!/bin/bash
# Board power and clocks while the P = 1 convolution runs back to back (rocm-smi sampled in the background):
#   gpurun -- 'bash tools/power_probe.sh [plan] > gpurun_out/power_probe.txt 2>&1'     plan: 1 (default) or 2 (hop 1536)
#   OHS_PROBE_STREAMS / _FRAMES / _TAPS: another workload (config 4: 64 / 131072 / 16384)
export OHS_PROBE_PLAN=${1:-1}
echo "conv plan $OHS_PROBE_PLAN"
python3 - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import open_headstage_amd as ohs
from open_headstage_amd import synth
S, frames = int(os.environ.get('OHS_PROBE_STREAMS', '256')), int(os.environ.get('OHS_PROBE_FRAMES', '480256'))
bp = ohs.BatchProcessor(S, num_bands=10)
for p, h in enumerate(synth.hrir_set(int(os.environ.get('OHS_PROBE_TAPS', '512')))):
    bp.set_ir(p, h)
bp.set_eq_enabled(False)
bp.set_conv_plan(int(os.environ.get("OHS_PROBE_PLAN", "1")))
x = synth.white_noise_torch(0, S, frames, torch.device("cuda:0")); y = torch.empty_like(x)
t_end = time.time() + 12
n = 0
while time.time() < t_end:
    for _ in range(50):
        bp.process(x, out=y)
    torch.cuda.synchronize(); n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    bp.process(x, out=y)
e1.record(); torch.cuda.synchronize()
print("steps", n, "ms per step (hot)", round(e0.elapsed_time(e1) / 50, 4), flush=True)
PY
pid=$!
sleep 5
for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|temp" | head -12; echo ---; sleep 1; done
wait $pid
rocm-smi --showmaxpower 2>&1 | grep -i power | head -3
