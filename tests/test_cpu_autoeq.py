"""CPU suite: AutoEQ CSV ingestion (src/autoeq_parser.rs:21-70) against an
independent restatement with Python's csv module, plus the error cases the Rust parser rejects."""
import csv
import io
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "autoeq_sample.csv")
TYPES = {"PK": 0, "LS": 1, "HS": 2}


def _restatement(text: str):
    rows = []
    rd = csv.DictReader(io.StringIO(text))
    for r in rd:
        rows.append((True, TYPES[r["Filter-Type"]], np.float32(r["Fc"]), np.float32(r["Q"]), np.float32(r["Gain"])))
    return rows


def test_parse_sample_profile_matches_restatement():
    from open_headstage_amd import parse_autoeq_csv
    bands = parse_autoeq_csv(GOLD)
    ref = _restatement(open(GOLD).read())
    assert len(bands) == len(ref) == 10
    for b, r in zip(bands, ref):
        assert (b.enabled, int(b.filter_type)) == (r[0], r[1])
        assert np.float32(b.frequency) == r[2] and np.float32(b.q) == r[3] and np.float32(b.gain) == r[4]
    assert [int(b.filter_type) for b in bands] == [1, 0, 0, 0, 0, 0, 0, 0, 0, 2]


def test_columns_by_name_extra_columns_quotes_crlf_blank_lines():
    from open_headstage_amd import parse_autoeq_csv_text
    text = 'Gain,Comment,Q,Fc,Filter-Type\r\n\r\n-3.5,"a, quoted ""note""",1.41,1000,PK\r\n4,x,0.7,105,LS\r\n'
    b = parse_autoeq_csv_text(text)
    assert len(b) == 2
    assert (int(b[0].filter_type), b[0].frequency, b[0].gain) == (0, 1000.0, -3.5)
    assert np.float32(b[0].q) == np.float32(1.41)
    assert (int(b[1].filter_type), b[1].frequency, b[1].q, b[1].gain) == (1, 105.0, np.float32(0.7), 4.0)
    assert parse_autoeq_csv_text("Filter-Type,Fc,Q,Gain\n") == []
    assert parse_autoeq_csv_text("") == []


@pytest.mark.parametrize("text,needle", [
    ("Filter-Type,Fc,Q,Gain\nLP,100,0.7,0\n", "Unsupported filter type: LP"),      # autoeq_parser.rs:48
    ("Filter-Type,Fc,Q,Gain\nPK,abc,0.7,0\n", "invalid float"),
    ("Filter-Type,Fc,Q,Gain\nPK, 100,0.7,0\n", "invalid float"),                   # Rust f32 parse: no whitespace
    ("Filter-Type,Fc,Q,Gain\nPK,100,0.7\n", "fields"),                             # csv: unequal lengths
    ("Type,Fc,Q,Gain\nPK,100,0.7,0\n", "missing field `Filter-Type`"),
])
def test_rejects_what_the_reference_rejects(text, needle):
    from open_headstage_amd import OhsError, parse_autoeq_csv_text
    from open_headstage_amd._ffi import OHS_ERR_PARSE
    with pytest.raises(OhsError) as ei:
        parse_autoeq_csv_text(text)
    assert ei.value.status == OHS_ERR_PARSE and needle in str(ei.value)


def test_missing_file_is_io_error():
    from open_headstage_amd import OhsError, parse_autoeq_csv
    from open_headstage_amd._ffi import OHS_ERR_IO
    with pytest.raises(OhsError) as ei:
        parse_autoeq_csv("/nonexistent/profile.csv")
    assert ei.value.status == OHS_ERR_IO
