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
