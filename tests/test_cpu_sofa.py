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
