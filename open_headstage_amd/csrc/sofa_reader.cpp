// sofa_reader.cpp -- minimal SOFA (AES69 SimpleFreeFieldHRIR) reader: SURVEY.md section 8 row f1.
//
// Replaces what src/sofa/loader.rs:79-199 obtains from the system library libmysofa (not vendored
// under the reference, no source available here): open a .sofa file, expose M x R x N impulse
// responses + source positions, pick the measurement nearest to (azimuth, elevation, radius).
// Host code only -- file parsing is not on the data-parallel path.
//
// HDF5 subset (enough for netCDF-4 written SOFA files such as data/hrtf/subject_003.sofa):
//   superblock v2/v3; version-2 object headers with continuation chunks; root-group links stored
//   compactly (Link messages) or densely (fractal heap direct blocks, walked sequentially);
//   datasets: contiguous, compact, or chunked with a version-1 B-tree; filters shuffle + deflate;
//   element types: little-endian IEEE f64 / f32 and 1/2/4/8-byte integers.
// NOT libmysofa behaviours (parity unpinned, by design): no loudness normalisation, no
// interpolation between neighbours, no resampling inside open -- see ohs_sofa_* in ohs_hip.h.
#include "../../include/ohs_hip.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace ohs_sofa_detail {

struct Dataset {
    std::vector<uint64_t> dims;
    std::vector<double> values;         // converted to f64
    std::map<std::string, std::string> str_attrs;
};

struct Reader {
    std::vector<uint8_t> f;
    std::string err;
    int off_size = 8, len_size = 8;

    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
    bool ok(uint64_t pos, uint64_t n) const { return pos <= f.size() && n <= f.size() - pos; }
    // Every read of the (untrusted) file image goes through rd() / u8(): out of range reads return 0 and
    // latch `oob`, which the callers turn into OHS_ERR_PARSE -- nothing indexes `f` unchecked.
    mutable bool oob = false;
    uint64_t rd(uint64_t pos, int n) const
    {
        if (n < 0 || n > 8 || !ok(pos, (uint64_t)n)) { oob = true; return 0; }
        uint64_t v = 0;
        for (int i = n - 1; i >= 0; --i) v = (v << 8) | f[pos + i];
        return v;
    }
    int u8(uint64_t pos) const
    {
        if (!ok(pos, 1)) { oob = true; return 0; }
        return f[pos];
    }
    // limits against hostile files: B-tree nodes visited per dataset, and bytes a dataset may occupy
    // (zlib cannot expand by more than ~1032:1, so anything beyond that is not backed by file content)
    static constexpr size_t kMaxBtreeNodes = 1u << 16;
    static constexpr int kMaxBtreeDepth = 16;
    uint64_t max_dataset_bytes() const
    {
        const uint64_t lim = (uint64_t)f.size() * 1100ull + (1ull << 20);
        return lim < (1ull << 31) ? lim : (1ull << 31);
    }

    struct Msg { int type; uint64_t pos; uint32_t size; };

    // collect the messages of a version-2 object header (following continuation chunks)
    bool object_messages(uint64_t addr, std::vector<Msg> &out)
    {
        if (!ok(addr, 16) || std::memcmp(&f[addr], "OHDR", 4) != 0) return fail("not a version-2 object header");
        if (u8(addr + 4) != 2) return fail("object header version != 2");
        const int flags = u8(addr + 5);
        uint64_t p = addr + 6;
        if (flags & 0x20) p += 16;
        if (flags & 0x10) p += 4;
        const int szb = 1 << (flags & 3);
        if (!ok(p, szb)) return fail("truncated object header");
        const uint64_t size0 = rd(p, szb);
        p += szb;
        struct Chunk { uint64_t pos, end; };
        std::vector<Chunk> chunks;
        if (!ok(p, size0)) return fail("object header chunk out of range");
        chunks.push_back({p, p + size0});
        for (size_t ci = 0; ci < chunks.size() && ci < 64; ++ci) {
            uint64_t q = chunks[ci].pos;
            const uint64_t end = chunks[ci].end;
            if (end < q || !ok(q, end - q)) return fail("object header chunk out of range");
            const int hdr = 4 + ((flags & 4) ? 2 : 0);
            while (q + hdr <= end) {
                const int type = u8(q);
                const uint32_t sz = (uint32_t)rd(q + 1, 2);
                q += hdr;
                if (q + sz > end) break;
                if (type == 0x10) {   // continuation: offset, length -> "OCHK" ... checksum
                    if (sz < (uint32_t)(off_size + len_size)) break;
                    const uint64_t coff = rd(q, off_size), clen = rd(q + off_size, len_size);
                    if (ok(coff, clen) && clen >= 8 && std::memcmp(&f[coff], "OCHK", 4) == 0)
                        chunks.push_back({coff + 4, coff + clen - 4});
                } else if (type != 0) {
                    out.push_back({type, q, sz});
                }
                q += sz;
            }
        }
        return true;
    }

    // Link message body -> (name, object header address); false if not a hard link / malformed
    bool parse_link(uint64_t p, uint64_t limit, std::string &name, uint64_t &addr, uint64_t &consumed)
    {
        const uint64_t p0 = p;
        if (limit > f.size()) limit = f.size();
        if (p + 2 > limit || f[p] != 1) return false;
        const int flags = f[p + 1];
        p += 2;
        int ltype = 0;
        if (flags & 0x08) { if (p + 1 > limit) return false; ltype = f[p]; p += 1; }
        if (flags & 0x04) p += 8;
        if (flags & 0x10) p += 1;
        const int nlb = 1 << (flags & 3);
        if (p + nlb > limit) return false;
        const uint64_t nlen = rd(p, nlb);
        p += nlb;
        if (nlen == 0 || nlen > 1024 || p + nlen > limit) return false;
        name.assign((const char *)&f[p], (size_t)nlen);
        p += nlen;
        if (ltype == 0) {
            if (p + off_size > limit) return false;
            addr = rd(p, off_size);
            p += off_size;
        } else if (ltype == 1) {       // soft link: length + string
            if (p + 2 > limit) return false;
            p += 2 + rd(p, 2);
            if (p > limit) return false;
            addr = ~0ull;
        } else return false;
        consumed = p - p0;
        return true;
    }

    // walk the direct blocks of a fractal heap and parse the link messages stored back to back
    void heap_links(uint64_t heap_addr, std::map<std::string, uint64_t> &links)
    {
        if (!ok(heap_addr, 5 + 4 + 1 + 4 + 12 * 8 + 2 + 2 * 8 + 2 + 2 + 8 + 2) ||
            std::memcmp(&f[heap_addr], "FRHP", 4) != 0) return;
        uint64_t p = heap_addr + 5;
        p += 2 + 2;                        // heap id length, io filter length
        const int hflags = u8(p); p += 1;
        p += 4;                            // max size of managed objects
        p += len_size + off_size;          // next huge id, huge btree
        p += len_size + off_size;          // free space, free space manager
        p += 4 * len_size;                 // managed space, allocated, iterator offset, n managed
        p += 4 * len_size;                 // huge size/n, tiny size/n
        const uint64_t width = rd(p, 2); p += 2;
        const uint64_t start_block = rd(p, len_size); p += len_size;
        const uint64_t max_direct = rd(p, len_size); p += len_size;
        const uint64_t max_heap_bits = rd(p, 2); p += 2;
        p += 2;                            // starting rows in root indirect block
        const uint64_t root = rd(p, off_size); p += off_size;
        const uint64_t cur_rows = rd(p, 2);
        if (max_heap_bits > 64) return;   // corrupt heap header
        const int boff = (int)((max_heap_bits + 7) / 8);
        const bool cksum = (hflags & 2) != 0;

        auto scan_direct = [&](uint64_t baddr, uint64_t bsize) {
            if (bsize < 16 || !ok(baddr, bsize) || std::memcmp(&f[baddr], "FHDB", 4) != 0) return;
            uint64_t q = baddr + 5 + off_size + boff + (cksum ? 4 : 0);
            const uint64_t end = baddr + bsize;
            while (q + 4 < end && f[q] == 1) {
                std::string name; uint64_t a = 0, used = 0;
                if (!parse_link(q, end, name, a, used)) break;
                if (a != ~0ull) links[name] = a;
                q += used;
            }
        };
        if (cur_rows == 0) { scan_direct(root, start_block); return; }
        if (!ok(root, 16) || std::memcmp(&f[root], "FHIB", 4) != 0) return;
        uint64_t q = root + 5 + off_size + boff;
        // rows of direct blocks: row 0 and 1 have start_block size, then doubling up to max_direct
        if (start_block == 0 || start_block > max_direct) return;       // corrupt heap header
        uint64_t max_direct_rows = 2;
        for (uint64_t s = start_block; s < max_direct && max_direct_rows < 48; s <<= 1) ++max_direct_rows;
        for (uint64_t r = 0; r < cur_rows && r < max_direct_rows; ++r) {
            const uint64_t bsize = r < 2 ? start_block : start_block << (r - 1);
            for (uint64_t k = 0; k < width && k < 4096; ++k) {
                if (!ok(q, off_size)) return;
                const uint64_t a = rd(q, off_size);
                q += off_size;
                if (a != ~0ull && a != 0) scan_direct(a, bsize);
            }
        }
    }

    bool root_links(uint64_t root, std::map<std::string, uint64_t> &links)
    {
        std::vector<Msg> msgs;
        if (!object_messages(root, msgs)) return false;
        for (const Msg &m : msgs) {
            if (m.type == 0x06) {          // compact link
                std::string name; uint64_t a = 0, used = 0;
                if (parse_link(m.pos, m.pos + m.size, name, a, used) && a != ~0ull) links[name] = a;
            } else if (m.type == 0x02) {   // link info -> fractal heap
                uint64_t p = m.pos + 2;
                if (u8(m.pos + 1) & 1) p += 8;
                if (p + (uint64_t)off_size > m.pos + m.size) continue;
                const uint64_t heap = rd(p, off_size);
                if (heap != ~0ull) heap_links(heap, links);
            }
        }
        return true;
    }

    struct TypeInfo { int cls = -1; uint32_t size = 0; bool big_endian = false; bool is_signed = false; };

    static double conv(const uint8_t *p, const TypeInfo &t)
    {
        uint8_t b[8];
        for (uint32_t i = 0; i < t.size; ++i) b[i] = t.big_endian ? p[t.size - 1 - i] : p[i];
        if (t.cls == 1) {
            if (t.size == 8) { double d; std::memcpy(&d, b, 8); return d; }
            float fl; std::memcpy(&fl, b, 4); return fl;
        }
        uint64_t u = 0;
        for (int i = (int)t.size - 1; i >= 0; --i) u = (u << 8) | b[i];
        if (t.is_signed && t.size < 8 && (u >> (8 * t.size - 1))) u |= ~0ull << (8 * t.size);
        return t.is_signed ? (double)(int64_t)u : (double)u;
    }

    bool parse_type(uint64_t p, TypeInfo &t)
    {
        if (!ok(p, 8)) { oob = true; return false; }
        t.cls = f[p] & 0x0f;
        t.big_endian = (f[p + 1] & 1) != 0;
        t.is_signed = (f[p + 1] & 8) != 0;
        t.size = (uint32_t)rd(p + 4, 4);
        return true;
    }

    bool inflate_buf(const uint8_t *src, size_t n, std::vector<uint8_t> &dst, size_t expect)
    {
        dst.resize(expect);
        uLongf dl = (uLongf)expect;
        const int rc = uncompress(dst.data(), &dl, src, (uLong)n);
        if (rc != Z_OK) return fail("zlib inflate failed");
        dst.resize(dl);
        return true;
    }

    bool read_dataset(uint64_t addr, Dataset &ds)
    {
        std::vector<Msg> msgs;
        if (!object_messages(addr, msgs)) return false;
        TypeInfo ty;
        int layout_class = -1;
        uint64_t data_addr = ~0ull, data_size = 0, btree = ~0ull;
        std::vector<uint64_t> chunk_dims;
        std::vector<int> filters;
        uint64_t compact_pos = 0;
        for (const Msg &m : msgs) {
            const uint64_t p = m.pos;
            const uint64_t mend = m.pos + m.size;        // every cursor below stays inside [p, mend)
            if (!ok(p, m.size)) return fail("message out of range");
            if (m.type == 0x01) {
                if (m.size < 4) return fail("truncated dataspace message");
                const int ver = f[p], rank = f[p + 1];
                uint64_t q = p + (ver == 1 ? 8 : 4);
                if (rank > 8 || q + (uint64_t)rank * len_size > mend) return fail("bad dataspace message");
                ds.dims.clear();
                for (int i = 0; i < rank; ++i) { ds.dims.push_back(rd(q, len_size)); q += len_size; }
            } else if (m.type == 0x03) {
                if (m.size < 8) return fail("truncated datatype message");
                parse_type(p, ty);
            } else if (m.type == 0x08) {
                if (m.size < 4) return fail("truncated data layout message");
                const int ver = f[p];
                if (ver != 3) return fail("unsupported data layout message version " + std::to_string(ver));
                layout_class = f[p + 1];
                if (layout_class == 0) {
                    data_size = rd(p + 2, 2); compact_pos = p + 4;
                    if (compact_pos + data_size > mend) return fail("compact data exceeds its message");
                } else if (layout_class == 1) {
                    if (p + 2 + (uint64_t)off_size + len_size > mend) return fail("truncated data layout message");
                    data_addr = rd(p + 2, off_size); data_size = rd(p + 2 + off_size, len_size);
                } else if (layout_class == 2) {
                    const int nd = f[p + 2];
                    if (nd > 9 || p + 3 + (uint64_t)off_size + 4ull * nd > mend) return fail("bad chunked layout message");
                    btree = rd(p + 3, off_size);
                    uint64_t q = p + 3 + off_size;
                    chunk_dims.clear();
                    for (int i = 0; i < nd; ++i) { chunk_dims.push_back(rd(q, 4)); q += 4; }
                }
            } else if (m.type == 0x0B) {
                if (m.size < 2) return fail("truncated filter pipeline message");
                const int ver = f[p], nf = f[p + 1];
                uint64_t q = p + (ver == 1 ? 8 : 2);
                for (int i = 0; i < nf; ++i) {
                    if (q + 8 > mend) return fail("filter pipeline exceeds its message");
                    const int id = (int)rd(q, 2); q += 2;
                    uint64_t nl = 0;
                    if (ver == 1 || id >= 256) { nl = rd(q, 2); q += 2; }
                    q += 2;                                  // flags
                    const uint64_t ncd = rd(q, 2); q += 2;
                    if (ver == 1) nl = (nl + 7) & ~7ull;
                    q += nl + 4 * ncd;
                    if (ver == 1 && (ncd & 1)) q += 4;
                    if (q > mend) return fail("filter pipeline exceeds its message");
                    filters.push_back(id);
                    if (filters.size() > 32) return fail("filter pipeline too long");   // the chunk filter mask has 32 bits
                }
            } else if (m.type == 0x0C) {     // attribute: keep fixed-length string values
                if (m.size < 9) continue;
                const int ver = f[p];
                const uint64_t nsz = rd(p + 2, 2), tsz = rd(p + 4, 2), ssz = rd(p + 6, 2);
                uint64_t q = p + 8 + (ver == 3 ? 1 : 0);
                auto pad = [&](uint64_t v) { return ver == 1 ? (v + 7) & ~7ull : v; };
                if (q + pad(nsz) + pad(tsz) + pad(ssz) > mend || tsz < 8) continue;
                std::string an((const char *)&f[q], (size_t)nsz);
                while (!an.empty() && an.back() == '\0') an.pop_back();
                const uint64_t tq = q + pad(nsz);
                TypeInfo at;
                if (!parse_type(tq, at)) continue;
                const uint64_t dq = tq + pad(tsz) + pad(ssz);
                if (at.cls == 3 && dq + at.size <= mend) {
                    std::string v((const char *)&f[dq], at.size);
                    while (!v.empty() && (v.back() == '\0' || v.back() == ' ')) v.pop_back();
                    ds.str_attrs[an] = v;
                }
            }
        }
        if (oob) return fail("truncated or corrupt object header message");
        if (ty.cls != 0 && ty.cls != 1) return fail("unsupported datatype class");
        if (ty.size != 1 && ty.size != 2 && ty.size != 4 && ty.size != 8) return fail("unsupported element size");
        uint64_t count = 1;
        for (uint64_t d : ds.dims) {
            if (d != 0 && count > (1ull << 31) / d) return fail("dataset too large");
            count *= d;
        }
        if (count * ty.size > max_dataset_bytes()) return fail("dataset larger than the file can back");
        std::vector<uint8_t> raw(count * ty.size, 0);
        if (layout_class == 0) {
            if (data_size > raw.size()) data_size = raw.size();
            if (!ok(compact_pos, data_size)) return fail("compact data out of range");
            if (data_size) std::memcpy(raw.data(), &f[compact_pos], data_size);
        } else if (layout_class == 1) {
            if (data_addr != ~0ull) {
                if (!ok(data_addr, raw.size())) return fail("contiguous data out of range");
                std::memcpy(raw.data(), &f[data_addr], raw.size());
            }
        } else if (layout_class == 2) {
            const size_t rank = ds.dims.size();
            if (chunk_dims.size() != rank + 1) return fail("chunk dimensionality mismatch");
            uint64_t chunk_elems = 1;
            for (size_t i = 0; i < rank; ++i) {
                if (chunk_dims[i] == 0) return fail("zero chunk dimension");
                if (chunk_elems > (1ull << 31) / chunk_dims[i]) return fail("chunk too large");
                chunk_elems *= chunk_dims[i];
            }
            const size_t chunk_bytes = chunk_elems * ty.size;
            if (chunk_bytes > max_dataset_bytes()) return fail("chunk larger than the file can back");
            // iterative walk of the version-1 B-tree (node type 1); a node may be visited once, the tree
            // may not be deeper than kMaxBtreeDepth and may not hold more than kMaxBtreeNodes nodes
            struct Item { uint64_t node; int depth; };
            std::vector<Item> stack;
            std::set<uint64_t> seen;
            if (btree != ~0ull) stack.push_back({btree, 0});
            while (!stack.empty()) {
                const uint64_t node = stack.back().node;
                const int depth = stack.back().depth;
                stack.pop_back();
                if (depth > kMaxBtreeDepth) return fail("chunk B-tree too deep");
                if (seen.size() >= kMaxBtreeNodes) return fail("chunk B-tree has too many nodes");
                if (!seen.insert(node).second) return fail("chunk B-tree revisits a node (cycle)");
                if (!ok(node, 24) || std::memcmp(&f[node], "TREE", 4) != 0 || f[node + 4] != 1)
                    return fail("bad chunk B-tree node");
                const int level = f[node + 5];
                const uint64_t used = rd(node + 6, 2);
                uint64_t q = node + 8 + 2 * off_size;
                const uint64_t key_size = 8 + 8 * (rank + 1);
                for (uint64_t e = 0; e < used; ++e) {
                    if (!ok(q, key_size + off_size)) return fail("truncated B-tree node");
                    const uint64_t csize = rd(q, 4), fmask = rd(q + 4, 4);
                    std::vector<uint64_t> coff(rank);
                    for (size_t i = 0; i < rank; ++i) coff[i] = rd(q + 8 + 8 * i, 8);
                    const uint64_t child = rd(q + key_size, off_size);
                    q += key_size + off_size;
                    if (level > 0) { stack.push_back({child, depth + 1}); continue; }
                    if (csize > max_dataset_bytes()) return fail("chunk out of range");
                    if (!ok(child, csize)) return fail("chunk out of range");
                    std::vector<uint8_t> buf(&f[child], &f[child] + csize), tmp;
                    for (int fi = (int)filters.size() - 1; fi >= 0; --fi) {
                        if (fi < 32 && (fmask & (1u << fi))) continue;
                        if (filters[fi] == 1) {              // deflate
                            if (!inflate_buf(buf.data(), buf.size(), tmp, chunk_bytes)) return false;
                            buf.swap(tmp);
                        } else if (filters[fi] == 2) {       // shuffle
                            if (fi >= 32) return fail("filter pipeline too long");
                            const size_t n = buf.size() / ty.size;
                            tmp.resize(buf.size());
                            for (size_t i = 0; i < n; ++i)
                                for (uint32_t b = 0; b < ty.size; ++b) tmp[i * ty.size + b] = buf[b * n + i];
                            for (size_t i = n * ty.size; i < buf.size(); ++i) tmp[i] = buf[i];
                            buf.swap(tmp);
                        } else if (filters[fi] == 3) {       // fletcher32: drop the trailing checksum
                            if (buf.size() >= 4) buf.resize(buf.size() - 4);
                        } else return fail("unsupported filter id " + std::to_string(filters[fi]));
                    }
                    if (buf.size() < chunk_bytes) return fail("short chunk");
                    // scatter the chunk into the dataset (edge chunks may stick out)
                    std::vector<uint64_t> idx(rank, 0);
                    for (uint64_t ce = 0; ce < chunk_elems; ++ce) {
                        uint64_t lin = 0;
                        bool inside = true;
                        for (size_t i = 0; i < rank; ++i) {
                            const uint64_t g = coff[i] + idx[i];
                            if (g >= ds.dims[i]) { inside = false; break; }
                            lin = lin * ds.dims[i] + g;
                        }
                        if (inside) std::memcpy(&raw[lin * ty.size], &buf[ce * ty.size], ty.size);
                        for (int i = (int)rank - 1; i >= 0; --i) {
                            if (++idx[i] < chunk_dims[i]) break;
                            idx[i] = 0;
                        }
                    }
                }
            }
        } else return fail("dataset has no data layout");
        ds.values.resize(count);
        for (uint64_t i = 0; i < count; ++i) ds.values[i] = conv(&raw[i * ty.size], ty);
        return true;
    }
};

}  // namespace ohs_sofa_detail

#include "sofa_internal.h"

extern "C" void ohsint_set_error(const char *msg);     // api_core.hip: feeds ohs_last_error()
namespace {
int sfail(int code, const std::string &m) { ohsint_set_error(m.c_str()); return code; }
const double kPi = 3.14159265358979323846264338327950288;

// AES69 spherical (degrees, azimuth counter-clockwise from the front, elevation up) -> Cartesian
void s2c(double az, double el, double r, double out[3])
{
    const double a = az * kPi / 180.0, e = el * kPi / 180.0;
    out[0] = r * std::cos(e) * std::cos(a);
    out[1] = r * std::cos(e) * std::sin(a);
    out[2] = r * std::sin(e);
}
}  // namespace

extern "C" {

// MySofa::spherical_to_cartesian / cartesian_to_spherical  loader.rs:203-229 (degrees <-> AES69 xyz)
int ohs_sofa_spherical_to_cartesian(const float az_el_r[3], float xyz[3])
{
    if (!az_el_r || !xyz) return sfail(OHS_ERR_INVALID_ARG, "NULL argument");
    double o[3];
    s2c(az_el_r[0], az_el_r[1], az_el_r[2], o);
    for (int k = 0; k < 3; ++k) xyz[k] = (float)o[k];
    return OHS_OK;
}

int ohs_sofa_cartesian_to_spherical(const float xyz[3], float az_el_r[3])
{
    if (!az_el_r || !xyz) return sfail(OHS_ERR_INVALID_ARG, "NULL argument");
    const double x = xyz[0], y = xyz[1], z = xyz[2];
    const double r = std::sqrt(x * x + y * y + z * z);
    double az = std::atan2(y, x) * 180.0 / kPi;
    if (az < 0) az += 360.0;                        // libmysofa's c2s reports azimuth in [0, 360)
    az_el_r[0] = (float)az;
    az_el_r[1] = (float)(std::atan2(z, std::sqrt(x * x + y * y)) * 180.0 / kPi);
    az_el_r[2] = (float)r;
    return OHS_OK;
}

static int sofa_open_impl(const char *path, ohs_sofa **out)
{
    using namespace ohs_sofa_detail;
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return sfail(OHS_ERR_IO, std::string("cannot open ") + path);   // SofaError::FileOpen (loader.rs:92-99)
    Reader rd;
    {
        char tmp[1 << 16];
        size_t r;
        while ((r = std::fread(tmp, 1, sizeof(tmp), fp)) > 0) rd.f.insert(rd.f.end(), tmp, tmp + r);
        std::fclose(fp);
    }
    static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
    if (rd.f.size() < 48 || std::memcmp(rd.f.data(), sig, 8) != 0) return sfail(OHS_ERR_PARSE, "not an HDF5 file");
    const int sbv = rd.f[8];
    if (sbv != 2 && sbv != 3)
        return sfail(OHS_ERR_PARSE, "HDF5 superblock version " + std::to_string(sbv) + " not supported (need 2 or 3)");
    rd.off_size = rd.f[9];
    rd.len_size = rd.f[10];
    if (rd.off_size != 8 || rd.len_size != 8) return sfail(OHS_ERR_PARSE, "only 8-byte HDF5 offsets/lengths supported");
    const uint64_t root = rd.rd(12 + 3 * 8, 8);
    if (rd.oob) return sfail(OHS_ERR_PARSE, "truncated superblock");
    std::map<std::string, uint64_t> links;
    if (!rd.root_links(root, links)) return sfail(OHS_ERR_PARSE, "root group: " + rd.err);
    auto get = [&](const char *name, Dataset &ds) -> bool {
        auto it = links.find(name);
        if (it == links.end()) { rd.fail(std::string("dataset ") + name + " not found"); return false; }
        return rd.read_dataset(it->second, ds);
    };
    Dataset ir, sp, sr, dl;
    if (!get("Data.IR", ir) || !get("SourcePosition", sp) || !get("Data.SamplingRate", sr))
        return sfail(OHS_ERR_PARSE, rd.err);
    if (ir.dims.size() != 3 || sp.dims.size() != 2 || sp.dims[1] != 3 || sr.values.empty())
        return sfail(OHS_ERR_PARSE, "unexpected SOFA dataset shapes (need Data.IR [M][R][N], SourcePosition [M][3])");
    if (sp.dims[0] != ir.dims[0] && sp.dims[0] != 1) return sfail(OHS_ERR_PARSE, "SourcePosition rows != M");
    if (ir.values.size() != ir.dims[0] * ir.dims[1] * ir.dims[2] || sp.values.size() != sp.dims[0] * 3)
        return sfail(OHS_ERR_PARSE, "dataset size does not match its dimensions");
    std::unique_ptr<ohs_sofa> hold(new ohs_sofa());
    ohs_sofa *s = hold.get();
    s->M = (size_t)ir.dims[0]; s->R = (size_t)ir.dims[1]; s->N = (size_t)ir.dims[2];
    s->fs = (float)sr.values[0];
    s->ir.resize(ir.values.size());
    for (size_t i = 0; i < ir.values.size(); ++i) s->ir[i] = (float)ir.values[i];
    std::string ptype = "spherical";
    auto it = sp.str_attrs.find("Type");
    if (it != sp.str_attrs.end()) ptype = it->second;
    const bool cart = ptype.find("cartesian") != std::string::npos;
    s->pos_sph.resize(s->M * 3);
    s->pos_xyz.resize(s->M * 3);
    for (size_t m = 0; m < s->M; ++m) {
        const size_t row = sp.dims[0] == 1 ? 0 : m;
        const double a = sp.values[row * 3], b = sp.values[row * 3 + 1], c = sp.values[row * 3 + 2];
        double xyz[3], sph[3];
        if (cart) {
            xyz[0] = a; xyz[1] = b; xyz[2] = c;
            const double r = std::sqrt(a * a + b * b + c * c);
            sph[0] = std::atan2(b, a) * 180.0 / kPi;
            if (sph[0] < 0) sph[0] += 360.0;
            sph[1] = r > 0 ? std::asin(c / r) * 180.0 / kPi : 0.0;
            sph[2] = r;
        } else {
            sph[0] = a; sph[1] = b; sph[2] = c;
            s2c(a, b, c, xyz);
        }
        for (int k = 0; k < 3; ++k) { s->pos_sph[m * 3 + k] = (float)sph[k]; s->pos_xyz[m * 3 + k] = (float)xyz[k]; }
    }
    s->delay.assign(s->M * s->R, 0.f);
    if (links.count("Data.Delay")) {
        // [1][R] (one delay per receiver for every measurement) or [M][R]; any other shape is ignored
        // (delays stay 0) rather than indexed -- a dims[0] that is neither 1 nor M would read past the end
        Dataset d;
        if (rd.read_dataset(links["Data.Delay"], d) && d.dims.size() == 2 && d.dims[1] == s->R &&
            (d.dims[0] == 1 || d.dims[0] == s->M) && d.values.size() == d.dims[0] * s->R)
            for (size_t m = 0; m < s->M; ++m)
                for (size_t r = 0; r < s->R; ++r)
                    s->delay[m * s->R + r] = (float)d.values[(d.dims[0] == 1 ? 0 : m) * s->R + r];
    }
    *out = hold.release();
    return OHS_OK;
}

// nothing may unwind across the C ABI: allocation failures and anything else thrown while parsing an
// untrusted file become status codes
int ohs_sofa_open(const char *path, ohs_sofa **out)
{
    if (!path || !out) return sfail(OHS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    try {
        return sofa_open_impl(path, out);
    } catch (const std::bad_alloc &) {
        return sfail(OHS_ERR_ALLOC, "out of memory while reading the SOFA file");
    } catch (const std::exception &e) {
        return sfail(OHS_ERR_PARSE, std::string("malformed SOFA file: ") + e.what());
    } catch (...) {
        return sfail(OHS_ERR_PARSE, "malformed SOFA file");
    }
}

void ohs_sofa_close(ohs_sofa *s) { delete s; }

int ohs_sofa_info(const ohs_sofa *s, size_t *M, size_t *R, size_t *N, float *fs)
{
    if (!s) return sfail(OHS_ERR_INVALID_ARG, "sofa is NULL");
    if (M) *M = s->M;
    if (R) *R = s->R;
    if (N) *N = s->N;
    if (fs) *fs = s->fs;
    return OHS_OK;
}

int ohs_sofa_position(const ohs_sofa *s, size_t m, float az_el_r[3])
{
    if (!s || !az_el_r || m >= s->M) return sfail(OHS_ERR_INVALID_ARG, "bad argument");
    std::memcpy(az_el_r, &s->pos_sph[m * 3], 3 * sizeof(float));
    return OHS_OK;
}

int ohs_sofa_nearest(const ohs_sofa *s, float az_deg, float el_deg, float radius_m, size_t *index)
{
    if (!s || !index || s->M == 0) return sfail(OHS_ERR_INVALID_ARG, "bad argument");
    double q[3];
    s2c(az_deg, el_deg, radius_m, q);
    if (s->flags & OHS_SOFA_INTERPOLATE) {      // mysofa_lookup projects the query onto the measured radius range first
        const double r = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        double f = 1.0;
        if (r > s->radius_max && r > 0) f = s->radius_max / r;
        else if (r < s->radius_min && r > 0) f = s->radius_min / r;
        q[0] *= f; q[1] *= f; q[2] *= f;
    }
    double best = 1e300;
    size_t bi = 0;
    for (size_t m = 0; m < s->M; ++m) {       // nearest in Cartesian space, as a kd-tree lookup would return
        const double dx = s->pos_xyz[m * 3] - q[0], dy = s->pos_xyz[m * 3 + 1] - q[1], dz = s->pos_xyz[m * 3 + 2] - q[2];
        const double d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; bi = m; }
    }
    *index = bi;
    return OHS_OK;
}

int ohs_sofa_get_ir(const ohs_sofa *s, size_t m, size_t receiver, float *out, size_t capacity)
{
    if (!s || !out || m >= s->M || receiver >= s->R) return sfail(OHS_ERR_INVALID_ARG, "bad argument");
    const size_t n = capacity < s->N ? capacity : s->N;
    std::memcpy(out, &s->ir[(m * s->R + receiver) * s->N], n * sizeof(float));
    return OHS_OK;
}

// MySofa::get_hrtf_irs(az, el, r) -> (left, right)  loader.rs:136-199, nearest measurement, DEGREES
int ohs_sofa_get_hrtf_irs(const ohs_sofa *s, float az_deg, float el_deg, float radius_m, float *left,
                          float *right, size_t capacity, float *delay_left, float *delay_right)
{
    if (!s || !left || !right) return sfail(OHS_ERR_INVALID_ARG, "NULL argument");
    if (s->N == 0) return sfail(OHS_ERR_PARSE, "Filter length is zero.");      // loader.rs:147-149
    if (s->R < 2) return sfail(OHS_ERR_PARSE, "need two receivers (left, right ear)");
    if (s->flags & OHS_SOFA_INTERPOLATE)
        return ohsint_sofa_interpolated(s, az_deg, el_deg, radius_m, left, right, capacity, delay_left, delay_right);
    size_t m = 0;
    int rc = ohs_sofa_nearest(s, az_deg, el_deg, radius_m, &m);
    if (rc) return rc;
    ohs_sofa_get_ir(s, m, 0, left, capacity);
    ohs_sofa_get_ir(s, m, 1, right, capacity);
    if (delay_left) *delay_left = s->delay[m * s->R];
    if (delay_right) *delay_right = s->delay[m * s->R + 1];
    return OHS_OK;
}

// Kaiser-windowed-sinc resampling of one impulse response (this build's own; libmysofa resamples
// inside mysofa_open with its bundled resampler -- parity unpinned).  out_len = ceil(n * fs_out / fs_in).
int ohs_sofa_resample_ir(const float *in, size_t n, float fs_in, float fs_out, float *out, size_t capacity,
                         size_t *out_len)
{
    if (!in || !out_len || fs_in <= 0 || fs_out <= 0) return sfail(OHS_ERR_INVALID_ARG, "bad argument");
    const double ratio = (double)fs_out / (double)fs_in;
    const size_t m = (size_t)std::ceil((double)n * ratio);
    *out_len = m;
    if (!out) return OHS_OK;
    const double cutoff = ratio < 1.0 ? ratio : 1.0;     // relative to the input Nyquist
    const int half = 32;                                  // taps each side, in input samples / cutoff
    const double beta = 9.0;
    auto bessel0 = [](double x) { double s = 1, t = 1; for (int k = 1; k < 40; ++k) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; } return s; };
    const double inv_b0 = 1.0 / bessel0(beta);
    for (size_t j = 0; j < m && j < capacity; ++j) {
        const double t = (double)j / ratio;              // position in input samples
        const long c = (long)std::floor(t);
        const double span = half / cutoff;
        double acc = 0.0;
        for (long k = c - (long)span; k <= c + (long)span + 1; ++k) {
            if (k < 0 || k >= (long)n) continue;
            const double d = t - (double)k;
            const double u = d / span;
            if (std::fabs(u) >= 1.0) continue;
            const double w = bessel0(beta * std::sqrt(1.0 - u * u)) * inv_b0;
            const double xarg = kPi * d * cutoff;
            const double sinc = std::fabs(xarg) < 1e-12 ? 1.0 : std::sin(xarg) / xarg;
            acc += (double)in[k] * cutoff * sinc * w;
        }
        // an impulse response keeps its frequency response H(f) = sum h[n] e^{-j 2 pi f n / fs}
        // across the rate change only if the samples are scaled by fs_in / fs_out
        out[j] = (float)(acc / ratio);
    }
    return OHS_OK;
}

}  // extern "C"
