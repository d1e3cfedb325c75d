// autoeq_parser.cpp -- AutoEQ profile ingestion: the C-ABI counterpart of
// parse_autoeq_csv (src/autoeq_parser.rs:21-70).  Host-only; no GPU work.
#include "../../include/ohs_hip.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

extern "C" void ohsint_set_error(const char *msg);     // api_core.hip: feeds ohs_last_error()

namespace {

int fail(int code, const std::string &m) { ohsint_set_error(m.c_str()); return code; }

}  // namespace

namespace {

// RFC-4180 record splitter with the csv crate's defaults: ',' delimiter, '"' quote with "" escape,
// \n / \r\n terminators, completely empty lines skipped.
bool csv_next_record(const char *t, size_t len, size_t &pos, std::vector<std::string> &fields)
{
    fields.clear();
    while (pos < len && (t[pos] == '\n' || t[pos] == '\r')) ++pos;     // empty lines
    if (pos >= len) return false;
    std::string cur;
    bool in_q = false, any = false;
    for (;;) {
        if (pos >= len) { fields.push_back(cur); return true; }
        const char ch = t[pos];
        if (in_q) {
            if (ch == '"') {
                if (pos + 1 < len && t[pos + 1] == '"') { cur.push_back('"'); pos += 2; }
                else { in_q = false; ++pos; }
            } else { cur.push_back(ch); ++pos; }
            continue;
        }
        if (ch == '"' && cur.empty() && !any) { in_q = true; any = true; ++pos; continue; }
        if (ch == ',') { fields.push_back(cur); cur.clear(); any = false; ++pos; continue; }
        if (ch == '\n' || ch == '\r') {
            fields.push_back(cur);
            if (ch == '\r' && pos + 1 < len && t[pos + 1] == '\n') ++pos;
            ++pos;
            return true;
        }
        cur.push_back(ch); any = true; ++pos;
    }
}

// Rust `str::parse::<f32>`: optional sign, decimal digits / '.', optional exponent, or inf / infinity /
// nan (any case); no surrounding whitespace, no hex.  strtof rounds correctly, as Rust does.
bool parse_f32_rust(const std::string &f, float *out)
{
    if (f.empty()) return false;
    const unsigned char c0 = (unsigned char)f[0];
    if (isspace(c0)) return false;
    for (size_t i = 0; i + 1 < f.size(); ++i)
        if (f[i] == '0' && (f[i + 1] == 'x' || f[i + 1] == 'X')) return false;
    char *end = nullptr;
    const float v = strtof(f.c_str(), &end);
    if (end == f.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}

int autoeq_parse(const char *text, size_t len, ohs_band_setting *out, size_t capacity, size_t *count)
{
    size_t pos = 0;
    if (len >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF)
        pos = 3;     // UTF-8 BOM is stripped by the csv crate
    std::vector<std::string> hdr, rec;
    size_t n = 0;
    if (!csv_next_record(text, len, pos, hdr)) { *count = 0; return OHS_OK; }    // empty file: no rows
    int c_type = -1, c_fc = -1, c_q = -1, c_gain = -1;
    for (size_t i = 0; i < hdr.size(); ++i) {
        if (hdr[i] == "Filter-Type") c_type = (int)i;
        else if (hdr[i] == "Fc") c_fc = (int)i;
        else if (hdr[i] == "Q") c_q = (int)i;
        else if (hdr[i] == "Gain") c_gain = (int)i;
    }
    size_t line = 1;
    while (csv_next_record(text, len, pos, rec)) {
        ++line;
        if (rec.size() != hdr.size())
            return fail(OHS_ERR_PARSE, "CSV error: record " + std::to_string(line) + " has " +
                                           std::to_string(rec.size()) + " fields, but the header has " +
                                           std::to_string(hdr.size()));
        if (c_type < 0) return fail(OHS_ERR_PARSE, "CSV deserialize error: missing field `Filter-Type`");
        if (c_fc < 0) return fail(OHS_ERR_PARSE, "CSV deserialize error: missing field `Fc`");
        if (c_q < 0) return fail(OHS_ERR_PARSE, "CSV deserialize error: missing field `Q`");
        if (c_gain < 0) return fail(OHS_ERR_PARSE, "CSV deserialize error: missing field `Gain`");
        ohs_band_setting b;
        b.enabled = 1;
        if (!parse_f32_rust(rec[c_fc], &b.frequency) || !parse_f32_rust(rec[c_q], &b.q) ||
            !parse_f32_rust(rec[c_gain], &b.gain))
            return fail(OHS_ERR_PARSE, "CSV deserialize error: record " + std::to_string(line) +
                                           ": invalid float literal");
        const std::string &ty = rec[c_type];
        if (ty == "PK") b.filter_type = OHS_FILTER_PEAK;
        else if (ty == "LS") b.filter_type = OHS_FILTER_LOWSHELF;
        else if (ty == "HS") b.filter_type = OHS_FILTER_HIGHSHELF;
        else return fail(OHS_ERR_PARSE, "Unsupported filter type: " + ty);
        if (n < capacity && out) out[n] = b;
        ++n;
    }
    *count = n;
    return OHS_OK;
}

}  // namespace

extern "C" {

int ohs_autoeq_parse_csv_text(const char *text, size_t len, ohs_band_setting *out, size_t capacity, size_t *count)
{
    if ((len && !text) || !count || (capacity && !out)) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    return autoeq_parse(text, len, out, capacity, count);
}

int ohs_autoeq_parse_csv(const char *path, ohs_band_setting *out, size_t capacity, size_t *count)
{
    if (!path || !count || (capacity && !out)) return fail(OHS_ERR_INVALID_ARG, "NULL argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(OHS_ERR_IO, std::string("cannot open ") + path);
    std::string buf;
    char tmp[4096];
    size_t r;
    while ((r = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, r);
    std::fclose(f);
    return autoeq_parse(buf.data(), buf.size(), out, capacity, count);
}

}  // extern "C"
