#!/usr/bin/env python3
"""Extracts the two-speaker HRIR set (azimuth +/-30 deg, elevation 0) of the reference's DATA asset
data/hrtf/subject_003.sofa (CIPIC subject 003; a data file, not source) into
tests/golden/cipic003_pm30.npz with this repo's SOFA reader, for the GPU tests (the GPU box has no
/root/reference).  Also stores the 48 kHz resampled set used as the realistic-HRIR input of
SURVEY section 8d.  Run from the repo root:  python tools/make_sofa_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_headstage_amd import sofa  # noqa: E402

SRC = "/root/reference/data/hrtf/subject_003.sofa"
s = sofa.MySofa(SRC)
il, ir = s.nearest(30.0, 0.0), s.nearest(-30.0, 0.0)
irs441 = sofa.speaker_hrirs(s)
irs48 = sofa.speaker_hrirs(s, target_samplerate=48000.0)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cipic003_pm30.npz"),
                    index_left=il, index_right=ir, pos_left=s.position(il), pos_right=s.position(ir),
                    fs=s.source_samplerate, irs_44100=np.stack(irs441), irs_48000=np.stack(irs48))
print("indices", il, ir, "taps", irs441[0].size, irs48[0].size)

# The reference's second data asset, data/hrtf/processed_hrir.wav (a data file: 16-bit PCM, 4 channels, 200 frames,
# 44.1 kHz): its channels are measurements 355 and 171 of the same SOFA file, receivers 0 and 1, quantised to int16
# (channels 0 / 2 clip).  Kept as a fixture: it pins ohs_sofa_get_ir against data the reference itself holds.
import struct  # noqa: E402

raw = open("/root/reference/data/hrtf/processed_hrir.wav", "rb").read()
assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
fmt_tag, channels, rate, _, _, bits = struct.unpack("<HHIIHH", raw[20:36])
assert (fmt_tag, channels, bits) == (1, 4, 16)
i = raw.index(b"data")
nbytes = struct.unpack("<I", raw[i + 4:i + 8])[0]
pcm = np.frombuffer(raw[i + 8:i + 8 + nbytes], dtype="<i2").reshape(-1, channels).copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "processed_hrir_wav.npz"), pcm=pcm, fs=np.float32(rate),
                    measurement=np.array([355, 355, 171, 171]), receiver=np.array([0, 1, 0, 1]))
print("processed_hrir.wav:", pcm.shape, rate)
