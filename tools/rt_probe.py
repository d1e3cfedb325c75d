import sys, numpy as np
sys.path.insert(0, '/root/repo')
import open_headstage_amd as ohs
from open_headstage_amd import synth
irs = synth.hrir_set(128)
a, b = ohs.ConvolutionEngine.new(), ohs.ConvolutionEngine.new()
for p in range(4):
    a.set_ir(p, irs[p]); b.set_ir(p, irs[p])
a.set_realtime(True)
x = synth.white_noise([5], 4096)[0]
for i in range(4):
    l, r = x[0, i*1024:(i+1)*1024].copy(), x[1, i*1024:(i+1)*1024].copy()
    g = a.process_block(l, r); e = b.process_block(l, r)
    print(i, np.array_equal(g[0], e[0]), flush=True)
a.set_realtime(False)
print("probe ok", flush=True)
import os
os._exit(0)
