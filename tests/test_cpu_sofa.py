"""CPU suite: SOFA HRIR source (src/sofa/loader.rs).  The reference's own loader
tests are re-created (non-existent file => FileOpen error, loader.rs:244-257; coordinate round trip,
loader.rs:324-343).  Decoding is checked against an independent decoder written here (zlib + numpy at
the dataset offsets SURVEY.md section 2 row 16 reports for the reference's data asset)."""
import os
import zlib

import numpy as np
import pytest

SOFA = "/root/reference/data/hrtf/subject_003.sofa"
need_asset = pytest.mark.skipif(not os.path.exists(SOFA), reason="reference data asset not present on this box")


def test_open_non_existent_file():
    from open_headstage_amd import OhsError
    from open_headstage_amd._ffi import OHS_ERR_IO
    from open_headstage_amd.sofa import MySofa
    with pytest.raises(OhsError) as ei:
        MySofa.open("/tmp/some_hopefully_non_existent_sofa_file_for_test.sofa", 48000.0)
    assert ei.value.status == OHS_ERR_IO          # SofaError::FileOpen


def test_open_rejects_non_hdf5(tmp_path):
    from open_headstage_amd import OhsError
    from open_headstage_amd._ffi import OHS_ERR_PARSE
    from open_headstage_amd.sofa import MySofa
    p = tmp_path / "not.sofa"
    p.write_bytes(b"RIFF" + bytes(100))
    with pytest.raises(OhsError) as ei:
        MySofa(str(p))
    assert ei.value.status == OHS_ERR_PARSE


def test_coordinate_conversion_roundtrip():
    from open_headstage_amd.sofa import MySofa
    spherical_in = np.array([30.0, 15.0, 1.5], np.float32)
    cart = MySofa.spherical_to_cartesian(spherical_in)
    out = MySofa.cartesian_to_spherical(cart)
    assert np.all(np.abs(out - spherical_in) < 1e-3)
    # AES69: x front, y left, z up
    assert np.allclose(MySofa.spherical_to_cartesian([90.0, 0.0, 1.0]), [0, 1, 0], atol=1e-6)
    assert np.allclose(MySofa.spherical_to_cartesian([0.0, 90.0, 2.0]), [0, 0, 2], atol=1e-6)


@need_asset
def test_cipic_subject_003_shapes_and_positions():
    from open_headstage_amd.sofa import MySofa
    s = MySofa(SOFA)
    assert (s.num_measurements, s.num_receivers, s.filter_length) == (1250, 2, 200)
    assert s.source_samplerate == 44100.0
    pos = np.stack([s.position(m) for m in range(s.num_measurements)])
    assert pos[:, 0].min() >= 0 and pos[:, 0].max() <= 360 and abs(pos[:, 1].min() + 50.6) < 0.5
    assert pos[:, 1].max() == 90.0 and np.all(pos[:, 2] == 1.0)
    assert np.allclose(s.position(308), [30, 0, 1]) and np.allclose(s.position(908), [330, 0, 1])


@need_asset
def test_data_ir_matches_independent_decoder():
    from open_headstage_amd.sofa import MySofa
    raw = open(SOFA, "rb").read()
    # Data.IR is one deflate+shuffle chunk of 4,000,000 B at file offset 60771
    dec = zlib.decompressobj().decompress(raw[60771:])
    assert len(dec) == 1250 * 2 * 200 * 8
    unshuffled = np.frombuffer(dec, np.uint8).reshape(8, -1).T.copy().view("<f8").reshape(1250, 2, 200)
    s = MySofa(SOFA)
    for m in (0, 1, 308, 908, 1249):
        for r in (0, 1):
            assert np.array_equal(s.ir(m, r), unshuffled[m, r].astype(np.float32)), (m, r)


@need_asset
def test_nearest_lookup_matches_brute_force():
    from open_headstage_amd.sofa import MySofa
    s = MySofa(SOFA)
    pos = np.stack([s.position(m) for m in range(s.num_measurements)]).astype(np.float64)
    az, el = np.radians(pos[:, 0]), np.radians(pos[:, 1])
    xyz = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
    rng = np.random.default_rng(0)
    for _ in range(200):
        qa, qe = rng.uniform(-180, 360), rng.uniform(-60, 90)
        q = np.array([np.cos(np.radians(qe)) * np.cos(np.radians(qa)), np.cos(np.radians(qe)) * np.sin(np.radians(qa)),
                      np.sin(np.radians(qe))])
        d = ((xyz - q) ** 2).sum(1)
        got = s.nearest(qa, qe, 1.0)
        assert d[got] <= d.min() + 1e-9
    assert s.nearest(30.0, 0.0) == 308 and s.nearest(-30.0, 0.0) == 908 and s.nearest(330.0, 0.0) == 908


@need_asset
def test_fixture_is_what_the_reader_returns_and_resampling_is_sane():
    from open_headstage_amd import sofa
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cipic003_pm30.npz"))
    s = sofa.MySofa(SOFA)
    irs = sofa.speaker_hrirs(s)
    assert np.array_equal(np.stack(irs), g["irs_44100"])
    # left speaker (az +30): the left ear is the near ear -- louder and earlier
    assert np.abs(irs[0]).max() > 2 * np.abs(irs[1]).max() and np.argmax(np.abs(irs[0])) < np.argmax(np.abs(irs[1]))
    # 44.1 -> 48 kHz: length ceil(200 * 160/147); the filter's frequency response below 15 kHz is preserved
    r = sofa.resample_ir(irs[0], 44100.0, 48000.0)
    assert r.size == 218
    f = np.linspace(200, 15000, 50)
    H0 = np.abs(np.exp(-2j * np.pi * np.outer(f, np.arange(200)) / 44100.0) @ irs[0].astype(np.float64))
    H1 = np.abs(np.exp(-2j * np.pi * np.outer(f, np.arange(218)) / 48000.0) @ r.astype(np.float64))
    assert np.max(np.abs(20 * np.log10(H1 / H0))) < 0.1


# ---- pin against the reference's own processed_hrir.wav (fixture: tools/make_sofa_fixture.py) ----------------------
@need_asset
def test_get_ir_matches_the_references_processed_hrir_wav():
    """data/hrtf/processed_hrir.wav holds measurements 355 and 171 of subject_003.sofa (receivers 0, 1) as
    16-bit PCM: every sample of ohs_sofa_get_ir, quantised the same way (clipping included), is within 1 LSB."""
    from open_headstage_amd.sofa import MySofa
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "processed_hrir_wav.npz"))
    s = MySofa(SOFA)
    assert float(g["fs"]) == s.source_samplerate and g["pcm"].shape == (s.filter_length, 4)
    for ch in range(4):
        ir = s.ir(int(g["measurement"][ch]), int(g["receiver"][ch]))
        q = np.clip(np.round(ir.astype(np.float64) * 32768.0), -32768, 32767)
        assert np.abs(q - g["pcm"][:, ch]).max() <= 1, ch
        # and it is THAT measurement, not a neighbour: the next-best measurement is far off
        other = s.ir(int(g["measurement"][ch]) + 1, int(g["receiver"][ch]))
        assert np.abs(np.clip(np.round(other * 32768.0), -32768, 32767) - g["pcm"][:, ch]).max() > 100


# ---- libmysofa-style conditioning (ohs_sofa_open_ex; parity unpinned, checked against independent numpy) -----------
@need_asset
def test_loudness_normalisation_as_mysofa_loudness():
    from open_headstage_amd import sofa
    raw = sofa.MySofa(SOFA)
    s = sofa.MySofa(SOFA, sofa.SOFA_LOUDNESS)
    pos = np.stack([raw.position(m) for m in range(raw.num_measurements)])
    key = pos[:, 0] + pos[:, 1]
    frontal = int(np.argmin(key))                       # libmysofa: smallest azimuth + elevation (first on ties)
    e = sum(float(np.sum(raw.ir(frontal, r).astype(np.float64) ** 2)) for r in (0, 1))
    factor = np.sqrt(2.0 / e)
    assert abs(s.loudness_factor() - factor) < 1e-5 * factor
    assert abs(sum(float(np.sum(s.ir(frontal, r).astype(np.float64) ** 2)) for r in (0, 1)) - 2.0) < 1e-4
    for m in (0, 308, 908, 1249):
        assert np.allclose(s.ir(m, 1), raw.ir(m, 1) * np.float32(s.loudness_factor()), rtol=1e-6, atol=0)
    assert raw.loudness_factor() == 1.0


@need_asset
def test_neighbours_and_interpolation_as_mysofa_getfilter():
    from open_headstage_amd import sofa
    s = sofa.MySofa(SOFA, sofa.SOFA_INTERPOLATE)
    raw = sofa.MySofa(SOFA)
    M = s.num_measurements
    pos = np.stack([s.position(m) for m in range(M)]).astype(np.float64)
    az, el = np.radians(pos[:, 0]), np.radians(pos[:, 1])
    xyz = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1) * pos[:, 2:3]

    def nearest(a_deg, e_deg, r=1.0):
        q = np.array([np.cos(np.radians(e_deg)) * np.cos(np.radians(a_deg)), np.cos(np.radians(e_deg)) * np.sin(np.radians(a_deg)),
                      np.sin(np.radians(e_deg))]) * r
        return int(np.argmin(((xyz - q) ** 2).sum(1))), q

    # neighbours: first different nearest measurement when stepping 0.5 degrees in azimuth / elevation
    for m in (308, 908, 400, 13):
        nb = s.neighbors(m)
        assert nb[4] == -1 and nb[5] == -1                # one radius in the file: no radial neighbours
        for k, (da, de) in enumerate(((0.5, 0), (-0.5, 0), (0, 0.5), (0, -0.5))):
            want = -1
            for step in range(1, 91):
                j, _ = nearest(pos[m, 0] + da * step, pos[m, 1] + de * step)
                if j != m:
                    want = j
                    break
            assert nb[k] == want, (m, k, nb, want)
    # a query ON a measurement returns it untouched; the plain handle agrees
    l, r = s.get_hrtf_irs(30.0, 0.0)
    assert np.array_equal(l, raw.ir(308, 0)) and np.array_equal(r, raw.ir(308, 1))
    # a query between measurements: inverse-distance weights over the nearest and the closer neighbour per axis
    qa, qe = 31.7, 1.2
    m0, q = nearest(qa, qe)
    nb = s.neighbors(m0)
    d0 = np.linalg.norm(q - xyz[m0])
    terms = [(m0, 1.0 / d0)]
    for a, b in ((0, 1), (2, 3)):
        if nb[a] >= 0 and nb[b] >= 0:
            da_, db_ = np.linalg.norm(q - xyz[nb[a]]), np.linalg.norm(q - xyz[nb[b]])
            if abs(da_ - db_) >= 1e-5:
                terms.append((nb[a], 1.0 / da_) if da_ < db_ else (nb[b], 1.0 / db_))
        elif nb[a] >= 0:
            terms.append((nb[a], 1.0 / np.linalg.norm(q - xyz[nb[a]])))
        elif nb[b] >= 0:
            terms.append((nb[b], 1.0 / np.linalg.norm(q - xyz[nb[b]])))
    wsum = sum(w for _, w in terms)
    for rcv, got in zip((0, 1), s.get_hrtf_irs(qa, qe)):
        want = sum(raw.ir(m, rcv).astype(np.float64) * w for m, w in terms) / wsum
        assert len(terms) >= 2 and np.max(np.abs(got - want)) < 2e-6 * max(1.0, np.abs(want).max())
    # a far-away query is projected onto the measured radius first
    l5, r5 = s.get_hrtf_irs(qa, qe, 5.0)
    l1, r1 = s.get_hrtf_irs(qa, qe, 1.0)
    assert np.allclose(l5, l1, atol=1e-7) and np.allclose(r5, r1, atol=1e-7)


@need_asset
def test_open_like_mysofa_resamples_everything_to_the_target_rate():
    from open_headstage_amd import sofa
    raw = sofa.MySofa(SOFA)
    s = sofa.MySofa.open_like_mysofa(SOFA, 48000.0)
    assert s.source_samplerate == 48000.0 and s.filter_length == 218          # ceil(200 * 48000 / 44100)
    f = np.float32(s.loudness_factor())
    want = sofa.resample_ir(raw.ir(308, 0), 44100.0, 48000.0)
    assert np.allclose(s.ir(308, 0), want * f, rtol=1e-6, atol=1e-9)
    irs = sofa.speaker_hrirs(s)                                              # no second resampling
    assert all(h.size == 218 for h in irs)
    with pytest.raises(Exception):
        sofa.MySofa(SOFA, 64)                                               # unknown flag


# ---- a SOFA file written from scratch (tests/util.py) + the speaker wiring's impulse responses (no device needed) -----
def test_minimal_sofa_written_from_scratch_reads_back(tmp_path):
    """The reader against an independent WRITER: superblock v2, version-2 object headers, compact links, contiguous f64
    datasets -- spherical and cartesian SourcePosition, Data.Delay as [1][R] and [M][R]."""
    from open_headstage_amd.sofa import MySofa
    from tests.util import write_minimal_sofa
    rng = np.random.default_rng(5)
    M, R, N = 14, 2, 37
    ir = rng.standard_normal((M, R, N))
    pos = np.stack([rng.uniform(0, 360, M), rng.uniform(-40, 80, M), rng.uniform(0.5, 2.0, M)], 1)
    for delay in (None, rng.uniform(0, 5, (1, R)), rng.uniform(0, 5, (M, R))):
        s = MySofa(write_minimal_sofa(str(tmp_path / "a.sofa"), ir, pos, 44100.0, delay=delay))
        assert (s.num_measurements, s.num_receivers, s.filter_length, s.source_samplerate) == (M, R, N, 44100.0)
        for m in range(M):
            assert np.array_equal(s.position(m), pos[m].astype(np.float32))
            for r in range(R):
                assert np.array_equal(s.ir(m, r), ir[m, r].astype(np.float32))
    a, e = np.radians(pos[:, 0]), np.radians(pos[:, 1])
    xyz = pos[:, 2:3] * np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], 1)
    s = MySofa(write_minimal_sofa(str(tmp_path / "c.sofa"), ir, xyz, 48000.0, position_type="cartesian"))
    for m in range(M):
        assert np.allclose(s.position(m), pos[m], atol=1e-3)


def _speaker_cases():
    return [(-30.0, 0.0, 30.0, 0.0, 48000.0), (-30.0, 0.0, 30.0, 0.0, 0.0), (-75.0, 20.0, 10.0, -15.0, 48000.0),
            (12.5, 44.0, 88.0, -44.0, 44100.0), (-90.0, 0.0, 90.0, 0.0, 96000.0)]


def _check_speaker_irs(s):
    from open_headstage_amd import sofa
    for az_l, el_l, az_r, el_r, fs in _speaker_cases():
        got = sofa.speaker_irs_plugin_angles(s, az_l, el_l, az_r, el_r, 1.0, fs)
        want = sofa.speaker_hrirs(s, -az_l, el_l, -az_r, el_r, 1.0, fs if fs > 0 else None)      # AES69: positive = left
        assert len(got) == 4
        for p in range(4):
            assert got[p].dtype == np.float32 and np.array_equal(got[p], want[p]), (az_l, az_r, fs, p)


def test_speaker_irs_of_the_c_abi_equal_the_python_wiring_on_a_synthetic_file(tmp_path):
    """ohs_sofa_speaker_irs (what ohs_*_set_speakers loads) = two get_hrtf_irs + resampling, left speaker -> (Lsl, Lsr),
    right -> (Rsl, Rsr); the plugin's azimuth is positive to the RIGHT (speaker_visualizer.rs:51-54), AES69's to the left."""
    from open_headstage_amd.sofa import MySofa
    from tests.util import write_minimal_sofa
    rng = np.random.default_rng(8)
    az, el = np.meshgrid(np.arange(0.0, 360.0, 15.0), np.array([-30.0, 0.0, 30.0]))
    pos = np.stack([az.ravel(), el.ravel(), np.ones(az.size)], 1)
    ir = rng.standard_normal((len(pos), 2, 64))
    _check_speaker_irs(MySofa(write_minimal_sofa(str(tmp_path / "g.sofa"), ir, pos, 44100.0)))


@need_asset
def test_speaker_irs_of_the_c_abi_equal_the_python_wiring_on_cipic_subject_003():
    from open_headstage_amd import sofa
    s = sofa.MySofa(SOFA)
    _check_speaker_irs(s)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cipic003_pm30.npz"))
    assert np.array_equal(np.stack(sofa.speaker_irs_plugin_angles(s, fs=48000.0)), g["irs_48000"])      # the plugin's default angles
    assert np.array_equal(np.stack(sofa.speaker_irs_plugin_angles(s, fs=0.0)), g["irs_44100"])
