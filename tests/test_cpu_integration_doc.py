"""INTEGRATION.md shows the Rust `extern "C"` declarations a maintainer of the reference adds.  No Rust toolchain exists
in this image, so the blocks cannot be compiled; this test keeps them from drifting: every function declared in an
`extern "C" { ... }` block of the document must exist in include/ohs_hip.h with the same number of parameters, the same
pointer / const shape and the matching scalar types, and the same kind of return value."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUST_SCALARS = {"c_int": "int", "i32": "int", "c_float": "float", "f32": "float", "usize": "size_t", "u32": "unsigned",
                "u64": "uint64_t", "c_char": "char", "c_void": "void", "f64": "double"}


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _rust_type(t):
    """-> (pointer depth, const flags outermost first, base C type name)"""
    t = t.strip()
    consts = []
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        assert m, t
        consts.append(m.group(1) == "const")
        t = t[m.end():]
    base = RUST_SCALARS.get(t, t)       # opaque structs keep their Rust name
    return len(consts), consts, base


def _c_type(t):
    t = re.sub(r"(\w+)\s*\[\d*\]$", r"* \1", t.strip())      # float out[5] -> float * out
    if not t.endswith("*"):
        t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", t)         # drop the parameter's name
    depth = t.count("*")
    t2 = t.replace("*", " ")
    toks = t2.split()
    const = bool(toks) and toks[0] == "const"       # const-ness of the innermost pointee: `float *const d[]` is not
    toks = [k for k in toks if k not in ("const", "struct")]
    return depth, const, " ".join(toks)


def rust_decls():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    decls = {}
    for block in re.findall(r"```rust\n(.*?)```", doc, flags=re.S):
        for ext in re.findall(r'extern "C" \{(.*?)\n?\}', block, flags=re.S):
            ext = re.sub(r"//[^\n]*", "", ext)
            for m in re.finditer(r"fn\s+(ohs_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
                name, args, ret = m.group(1), m.group(2), (m.group(3) or "").strip()
                params = []
                for a in _split_args(" ".join(args.split())):
                    pname, ptype = a.split(":", 1)
                    params.append(_rust_type(ptype))
                decls[name] = (params, ret)
    return decls


def c_decls():
    hdr = open(os.path.join(ROOT, "include", "ohs_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ohs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else [_c_type(a) for a in _split_args(args)]
        decls[name] = (params, ret)
    return decls


def test_every_rust_declaration_matches_the_header():
    rust, c = rust_decls(), c_decls()
    assert len(rust) >= 20, sorted(rust)            # the blocks were found and parsed
    opaque = {"OhsEngine": "ohs_engine", "OhsEq": "ohs_eq", "OhsBatch": "ohs_batch", "OhsSofa": "ohs_sofa",
              "OhsBandSetting": "ohs_band_setting", "OhsNodeBatch": "ohs_node_batch", "OhsBiquad": "ohs_biquad"}
    for name, (rparams, rret) in sorted(rust.items()):
        assert name in c, f"INTEGRATION.md declares {name}, include/ohs_hip.h does not"
        cparams, cret = c[name]
        assert len(rparams) == len(cparams), f"{name}: {len(rparams)} parameters in INTEGRATION.md, {len(cparams)} in the header"
        for i, ((rd, rconsts, rbase), (cd, cconst, cbase)) in enumerate(zip(rparams, cparams)):
            assert rd == cd, f"{name} parameter {i}: pointer depth {rd} vs {cd}"
            assert opaque.get(rbase, rbase) == cbase, f"{name} parameter {i}: {rbase} vs {cbase}"
            if rd >= 1:     # const-ness of the pointee (innermost level)
                assert rconsts[-1] == cconst, f"{name} parameter {i}: const {rconsts[-1]} vs {cconst}"
        # return: nothing <-> void, c_int <-> int, pointer <-> pointer
        if rret == "":
            assert cret == "void", (name, cret)
        elif rret.startswith("*"):
            assert "*" in cret, (name, cret)
        else:
            assert RUST_SCALARS.get(rret, rret) == cret, (name, rret, cret)


def test_the_eq_shim_takes_its_coefficients_from_the_biquad_crate():
    """Section 4's parity argument: the shim's update_band_coeffs CODE calls ohs_eq_set_band_coeffs with the crate's
    coefficients; ohs_eq_update_band appears in that method only inside comments."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"pub fn update_band_coeffs\(.*?\n    \}\n", doc, flags=re.S)
    assert m
    code = re.sub(r"//[^\n]*", "", m.group(0))
    assert "ohs_eq_set_band_coeffs" in code and "from_params" in code
    assert "ohs_eq_update_band" not in code


def test_the_document_agrees_with_itself_about_the_biquad_crate():
    """section 1 once said `Cargo.toml`: drop `rustfft`, `biquad` while section 4's shim calls the
    crate.  Every sentence of the document that speaks of dropping / removing a dependency must not name `biquad`, the
    Cargo.toml row of section 1 must say that it stays, and section 4 must still say so too."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    row = [l for l in doc.splitlines() if l.startswith("| `Cargo.toml`")]
    assert len(row) == 1, row
    row = row[0]
    assert "rustfft" in row and re.search(r"`biquad[^`]*`\s+stays", row), row
    for sentence in re.split(r"(?<=[.;:|])\s", doc):
        if re.search(r"\b(drop|drops|dropped|remove|removes|removed|delete)\b", sentence, flags=re.I) and "biquad" in sentence:
            m = re.search(r"\b(drop|drops|dropped|remove|removes|removed|delete)\b[^.;:|]*", sentence, flags=re.I)
            assert "biquad" not in m.group(0), f"INTEGRATION.md tells the maintainer to drop the biquad crate: {sentence!r}"
    assert re.search(r"`Cargo.toml`\s+keeps `biquad", doc), "section 4 no longer says that Cargo.toml keeps the crate"
    assert "biquad::Coefficients" in doc and "from_params" in doc
