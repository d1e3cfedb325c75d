"""Shared helpers for the parity tests."""
import numpy as np

RMS_TOL = 1e-6   # BASELINE.json north_star: <= 1e-6 RMS float error vs the reference CPU path


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def rms_err(test, ref):
    """(absolute RMS error, relative RMS error) over all frames / channels."""
    test = np.asarray(test, np.float64)
    ref = np.asarray(ref, np.float64)
    e = rms(test - ref)
    r = rms(ref)
    return e, (e / r if r > 0 else e)


def assert_parity(test, ref, what, tol=RMS_TOL):
    a, r = rms_err(test, ref)
    assert a <= tol and r <= tol, f"{what}: abs RMS {a:.3e}, rel RMS {r:.3e} (tol {tol:.0e})"
    return a, r


def setup_engines(oracle, gpu_engine, irs):
    eo = oracle.ConvolutionEngine()
    for p, h in enumerate(irs):
        eo.set_ir(p, h)
        gpu_engine.set_ir(p, h)
    return eo


def write_minimal_sofa(path, ir, positions, fs, delay=None, position_type=None):
    """A SOFA (HDF5) file written from scratch, independent of the reader under test: superblock version 2, version-2
    object headers, root-group links as compact Link messages, contiguous little-endian f64 datasets Data.IR [M][R][N],
    SourcePosition [M][3] (degrees / metres), Data.SamplingRate [1], optionally Data.Delay [1 or M][R].  Enough for
    ohs_sofa_open; not a complete AES69 file (no attributes beyond SourcePosition's optional Type)."""
    import struct
    ir = np.asarray(ir, np.float64)
    positions = np.asarray(positions, np.float64)
    sets = [("Data.IR", ir, None), ("SourcePosition", positions, position_type),
            ("Data.SamplingRate", np.asarray([fs], np.float64), None)]
    if delay is not None:
        sets.append(("Data.Delay", np.asarray(delay, np.float64), None))
    UNDEF = 0xFFFFFFFFFFFFFFFF

    def msg(mtype, body):
        return struct.pack("<BHB", mtype, len(body), 0) + body

    def ohdr(messages):
        body = b"".join(messages)
        return b"OHDR" + struct.pack("<BB", 2, 0x02) + struct.pack("<I", len(body)) + body + b"\0\0\0\0"    # (+ checksum, unchecked)

    def dataset_header(arr, data_addr, type_attr):
        space = struct.pack("<BBBB", 2, arr.ndim, 0, 1) + b"".join(struct.pack("<Q", d) for d in arr.shape)
        # datatype: class 1 (floating point) version 1; bit field: little-endian, IEEE sign / exponent / mantissa of f64
        dtype = struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        layout = struct.pack("<BB", 3, 1) + struct.pack("<QQ", data_addr, arr.size * 8)
        msgs = [msg(0x01, space), msg(0x03, dtype), msg(0x08, layout)]
        if type_attr:       # attribute message version 3: fixed-length string "Type"
            name, val = b"Type\0", type_attr.encode()
            atype = struct.pack("<BBBBI", 0x13, 0x00, 0x00, 0x00, len(val))        # class 3 (string), null-padded, ASCII
            aspace = struct.pack("<BBBB", 2, 0, 0, 0)                                # scalar
            msgs.append(msg(0x0C, struct.pack("<BBHHHB", 3, 0, len(name), len(atype), len(aspace), 0) + name + atype + aspace + val))
        return ohdr(msgs)

    # layout of the file: superblock | root header | dataset headers | raw data
    sizes = [len(dataset_header(a, 0, t)) for _, a, t in sets]
    links_size = sum(4 + 2 + 1 + len(n) + 8 for n, _, _ in sets)
    root_len = 4 + 2 + 4 + links_size + 4
    pos = 48 + root_len
    hdr_addr = []
    for sz in sizes:
        hdr_addr.append(pos)
        pos += sz
    data_addr = []
    for _, a, _ in sets:
        pos = (pos + 7) & ~7
        data_addr.append(pos)
        pos += a.size * 8
    eof = pos
    out = bytearray()
    out += b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBB", 2, 8, 8, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, 48) + b"\0\0\0\0"
    links = [msg(0x06, struct.pack("<BBB", 1, 0, len(n)) + n.encode() + struct.pack("<Q", hdr_addr[i]))
             for i, (n, _, _) in enumerate(sets)]
    root = ohdr(links)
    assert len(root) == root_len and len(out) == 48
    out += root
    for i, (_, a, t) in enumerate(sets):
        assert len(out) == hdr_addr[i]
        out += dataset_header(a, data_addr[i], t)
    for i, (_, a, _) in enumerate(sets):
        out += b"\0" * (data_addr[i] - len(out))
        out += np.ascontiguousarray(a, "<f8").tobytes()
    assert len(out) == eof
    with open(path, "wb") as f:
        f.write(bytes(out))
    return path
