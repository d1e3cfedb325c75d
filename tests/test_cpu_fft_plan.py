"""CPU suite: the index plan of the wavefront FFT-1024 (tools/proto_wave_fft.py, a numpy model of csrc/wave_fft.hpp):
forward / inverse against numpy, the mirror-bin map, the mirror spectrum read back from the T2 image (round 2),
the paired bin layout of k_conv_p1 (round 3), and the LDS bank census under the gfx950 grouping rules (two 32-lane groups per ds_read_b64, four 16-lane groups
per ds_write_b64) -- everything conflict-free except the documented 2-way conflict of the inverse T2 write."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_fft_index_plan_and_bank_census():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "proto_wave_fft.py")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert out.strip().endswith("OK") and "mirror from the T2 image: OK" in out
    assert "T1 fwd write 1 read 1" in out and "T2 fwd write 1 read 1" in out and "T1 inv write 1 read 1" in out
    assert "T2 inv write 2 read 1" in out                      # the one documented 2-way conflict
    assert "T2 mirror reads 1 1" in out and "twiddle table rows / columns 1 1" in out
    assert "w64 table in the row padding, forward / inverse 1 1" in out and "slab 1084 entries: OK" in out
    # round 3: the paired bin layout of k_conv_p1 (coverage, mirror map and numerics are asserted by the tool itself)
    assert "paired layout: OK" in out and "T2 paired read 1 write 2" in out and "w64 rows, paired inverse 1" in out
