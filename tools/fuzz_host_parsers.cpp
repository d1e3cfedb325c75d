// Mutation fuzz of the host-side parsers (SOFA / HDF5 subset reader, AutoEQ CSV) under ASan + UBSan, CPU only:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude \\
//       -Iopen_headstage_amd/csrc -o /tmp/ohs_fuzz tools/fuzz_host_parsers.cpp \\
//       open_headstage_amd/csrc/sofa_reader.cpp open_headstage_amd/csrc/autoeq_parser.cpp -lz
//   /tmp/ohs_fuzz <some.sofa> [iterations]
// Round 2: bounds-checked reads, B-tree visit limits, size caps and try/catch at the ABI after the advisor's
// Data.Delay [2][R] over-read; `--open` mode runs single regression files (tests/test_cpu_parsers_asan.py).
// Round 1: found one shift-by-corrupt-count (fixed, with a guard on corrupt heap headers); then 6 000 mutated
// copies of the CIPIC subject-003 file and 20 000 mutated CSV texts without a report.
#include "ohs_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
static std::vector<unsigned char> slurp(const char *p) { std::vector<unsigned char> b; FILE *f = fopen(p, "rb"); if (!f) return b; unsigned char t[65536]; size_t r; while ((r = fread(t, 1, sizeof t, f)) > 0) b.insert(b.end(), t, t + r); fclose(f); return b; }
// the two symbols the parsers take from api_core.hip
static thread_local std::string g_err;
extern "C" void ohsint_set_error(const char *m) { g_err = m ? m : ""; }
extern "C" const char *ohs_last_error(void) { return g_err.c_str(); }

// open one file and touch every accessor (what a host does with a file it was handed)
static int exercise(const char *path, bool resample = false)
{
    ohs_sofa *s = nullptr;
    const int rc = ohs_sofa_open(path, &s);
    if (rc != OHS_OK) return rc;
    size_t M = 0, R = 0, N = 0; float fs = 0;
    ohs_sofa_info(s, &M, &R, &N, &fs);
    std::vector<float> l(N + 1), r(N + 1); float dl, dr; size_t idx;
    ohs_sofa_nearest(s, 30.f, 0.f, 1.f, &idx);
    ohs_sofa_get_hrtf_irs(s, -30.f, 10.f, 1.f, l.data(), r.data(), N, &dl, &dr);
    float p3[3]; if (M) ohs_sofa_position(s, M - 1, p3);
    ohs_sofa_close(s);
    // the same file through the libmysofa-style conditioning (resample + loudness + neighbour interpolation)
    ohs_sofa *c = nullptr;
    // (resampling every IR of every mutated file would dominate the fuzz; it runs on the single-file mode)
    if (ohs_sofa_open_ex(path, resample ? OHS_SOFA_MYSOFA_DEFAULTS : (OHS_SOFA_LOUDNESS | OHS_SOFA_INTERPOLATE), 48000.0f, &c) == OHS_OK) {
        size_t M2 = 0, R2 = 0, N2 = 0; float fs2 = 0;
        ohs_sofa_info(c, &M2, &R2, &N2, &fs2);
        std::vector<float> l2(N2 + 1), r2(N2 + 1); float a, b;
        ohs_sofa_get_hrtf_irs(c, 31.7f, 1.2f, 3.0f, l2.data(), r2.data(), N2, &a, &b);
        ohs_sofa_get_hrtf_irs(c, -120.f, -40.f, 0.2f, l2.data(), r2.data(), N2, &a, &b);
        int nb[6]; if (M2) ohs_sofa_neighbors(c, M2 / 2, nb);
        ohs_sofa_close(c);
    }
    return OHS_OK;
}

int main(int argc, char **argv)
{
    if (argc < 2) { printf("usage: %s <file.sofa> [iterations | --open]\n", argv[0]); return 2; }
    const char *src = argv[1];
    if (argc > 2 && !strcmp(argv[2], "--open")) {      // regression inputs: one file, as it is
        const int rc = exercise(src, true);
        printf("open: status %d (%s)\n", rc, rc ? ohs_last_error() : "ok");
        return 0;
    }
    const int iters = argc > 2 ? atoi(argv[2]) : 300;
    std::vector<unsigned char> good = slurp(src);
    if (good.empty()) { printf("cannot read %s\n", src); return 2; }
    std::mt19937_64 rng(12345);
    int ok = 0, bad = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<unsigned char> m = good;
        const int kind = it % 4;
        if (kind == 0) {                       // truncate
            m.resize(rng() % m.size());
        } else if (kind == 1) {                // flip bytes in the first 64 KiB (headers, B-trees, heaps)
            const int n = 1 + rng() % 8;
            for (int i = 0; i < n; ++i) m[rng() % std::min<size_t>(m.size(), 65536)] ^= (unsigned char)(1 + rng() % 255);
        } else if (kind == 2) {                // flip bytes anywhere (compressed payload too)
            const int n = 1 + rng() % 32;
            for (int i = 0; i < n; ++i) m[rng() % m.size()] = (unsigned char)rng();
        } else {                               // overwrite a 4/8-byte field with an extreme value
            const size_t o = rng() % (std::min<size_t>(m.size(), 65536) - 8);
            const unsigned long long v = (rng() & 1) ? ~0ull : (1ull << (rng() % 63));
            memcpy(&m[o], &v, (rng() & 1) ? 8 : 4);
        }
        if (it % 16 == 5) {                    // structure-aware: grow / zero a dimension or a count near a known tag
            static const char *tags[] = {"TREE", "OHDR", "FRHP", "FHDB", "OCHK"};
            const char *tag = tags[rng() % 5];
            std::vector<size_t> hits;
            for (size_t o = 0; o + 4 < std::min<size_t>(m.size(), 1u << 20); ++o)
                if (!memcmp(&m[o], tag, 4)) hits.push_back(o);
            if (!hits.empty()) {
                const size_t o = hits[rng() % hits.size()] + 4 + rng() % 40;
                if (o + 8 < m.size()) { const unsigned long long v = rng() & 3 ? rng() % 70000 : ~0ull; memcpy(&m[o], &v, 1 + rng() % 8); }
            }
        }
        FILE *f = fopen("/tmp/ohs_fuzz_mut.sofa", "wb"); fwrite(m.data(), 1, m.size(), f); fclose(f);
        if (exercise("/tmp/ohs_fuzz_mut.sofa") == OHS_OK) ++ok; else ++bad;
    }
    printf("sofa: %d mutated files opened, %d rejected, no crash\n", ok, bad);
    // AutoEQ CSV
    const char *seed = "Filter-Type,Fc,Q,Gain\nPK,105.0,0.7,4.0\nLS,31,1.41,-3\n\"HS\",10000,\"0.7\",-3.5\n";
    std::string g = seed;
    int cok = 0, cbad = 0;
    for (int it = 0; it < 20000; ++it) {
        std::string m = g;
        const int n = 1 + rng() % 6;
        for (int i = 0; i < n; ++i) {
            const int k = rng() % 3;
            const size_t o = m.empty() ? 0 : rng() % m.size();
            static const char alpha[] = "\",\n\r.-+eEinfNaPKLSH0123456789 \t\0x";
            const char c = alpha[rng() % (sizeof(alpha) - 1)];
            if (k == 0 && !m.empty()) m[o] = c; else if (k == 1) m.insert(m.begin() + o, c); else if (!m.empty()) m.erase(m.begin() + o);
        }
        ohs_band_setting out[4]; size_t cnt = 0;
        const int rc = ohs_autoeq_parse_csv_text(m.data(), m.size(), out, 4, &cnt);
        (rc == OHS_OK ? cok : cbad)++;
    }
    printf("autoeq: %d parsed, %d rejected, no crash\n", cok, cbad);
    return 0;
}
