// tuning.cpp -- see tuning.h.  Host-only.
#include "tuning.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace ohs {

namespace {
Tuning g_tuning;

#ifdef OHS_EXPERIMENTS
bool parse_ints(const std::string &v, int *out, int n)
{
    const char *p = v.c_str();
    for (int i = 0; i < n; ++i) {
        char *end = nullptr;
        const long x = strtol(p, &end, 10);
        if (end == p) return false;
        out[i] = (int)x;
        p = (*end == ',') ? end + 1 : end;
    }
    return true;
}

// the environment variables of rounds 1-3 keep their names: OHS_P1_TARGET_WAVES=4096 == tuning_set("p1_target_waves", "4096")
const struct { const char *env, *key; } kEnvKeys[] = {
    {"OHS_P1_TARGET_WAVES", "p1_target_waves"}, {"OHS_P1_ROUNDS", "p1_rounds"}, {"OHS_P1_OWN_TAILS", "p1_own_tails"},
    {"OHS_P1_STAGGER", "p1_stagger"}, {"OHS_P1_PRIO", "p1_prio"}, {"OHS_P1_WEIGHTS", "p1_weights"},
    {"OHS_P1_LAZY_STATE", "p1_lazy_state"}, {"OHS_P1_XCD", "p1_xcd"}, {"OHS_EQ_XCD", "eq_xcd"},
    {"OHS_EQ_CONVEYOR", "eq_conveyor"}, {"OHS_EQ_RING_V1", "eq_ring_v1"}, {"OHS_EQ_NO_PRIO", "eq_no_prio"}, {"OHS_EQ_FORM", "eq_form"},
    {"OHS_EQ_WG_WAVES", "eq_wg_waves"}, {"OHS_EQ_LDS", "eq_lds"}, {"OHS_NO_OVERLAP", "no_overlap"},
    {"OHS_FORCE_OVERLAP", "force_overlap"}, {"OHS_XCD_SPLIT", "xcd_split"}, {"OHS_OVERLAP_CUTS", "overlap_cuts"},
    {"OHS_CONV_CU_MASK", "conv_cu_mask"}, {"OHS_MAC_NO_XCD", "mac_no_xcd"}, {"OHS_HOST_PIPE_TRACE", "host_pipe_trace"},
    {"OHS_HOST_PIPE_FLAT_PRIORITIES", "host_pipe_flat_priorities"}, {"OHS_RT_MAX_RESIDENT", "rt_max_resident"},
    {"OHS_RT_IDLE_POLLS", "rt_idle_polls"}, {"OHS_RT_POLL_SLEEP", "rt_poll_sleep"},
    {"OHS_NODE_SHARED_DEVICE", "node_shared_device_rehearsal"},
};
std::once_flag g_env_once;
void load_env()
{
    for (const auto &e : kEnvKeys)
        if (const char *v = getenv(e.env))
            if (!tuning_set(e.key, v)) fprintf(stderr, "[ohs experiments] %s=%s ignored\n", e.env, v);
}
#endif
}  // namespace

const Tuning &tuning()
{
#ifdef OHS_EXPERIMENTS
    std::call_once(g_env_once, load_env);
#endif
    return g_tuning;
}

Tuning &tuning_rt()
{
#ifdef OHS_EXPERIMENTS
    std::call_once(g_env_once, load_env);
#endif
    return g_tuning;
}

#ifdef OHS_EXPERIMENTS
bool tuning_set(const std::string &key, const std::string &value)
{
    Tuning &t = g_tuning;
    int v[4] = {0, 0, 0, 0};
    auto one = [&](int *dst) { if (!parse_ints(value, v, 1)) return false; *dst = v[0]; return true; };
    if (key == "p1_target_waves") { t.p1_target_waves = atoll(value.c_str()); return true; }
    if (key == "p1_rounds") return one(&t.p1_rounds) && (t.p1_rounds = (t.p1_rounds >= 1 && t.p1_rounds <= 4) ? t.p1_rounds : 1);
    if (key == "p1_own_tails") return one(&t.p1_own_tails);
    if (key == "p1_stagger") return one(&t.p1_stagger);
    if (key == "p1_prio") return one(&t.p1_prio);
    if (key == "p1_lazy_state") return one(&t.p1_lazy_state);
    if (key == "p1_weights") {
        if (!parse_ints(value, v, 4)) return false;
        for (int i = 0; i < 4; ++i) if (v[i] <= 0 || v[i] >= 100000) return false;
        for (int i = 0; i < 4; ++i) t.p1_weights[i] = v[i];
        return true;
    }
    if (key == "p1_xcd" || key == "eq_xcd") {
        int *lo = key == "p1_xcd" ? &t.p1_xcd_lo : &t.eq_xcd_lo, *n = key == "p1_xcd" ? &t.p1_xcd_n : &t.eq_xcd_n;
        if (value.empty() || value == "off") { *lo = 0; *n = 0; return true; }
        if (!parse_ints(value, v, 2) || v[1] < 1 || v[0] < 0 || v[0] + v[1] > 8) return false;
        *lo = v[0]; *n = v[1];
        return true;
    }
    if (key == "eq_conveyor") return one(&t.eq_conveyor);
    if (key == "eq_ring_v1") return one(&t.eq_ring_v1);
    if (key == "eq_no_prio") return one(&t.eq_no_prio);
    if (key == "eq_form") return one(&t.eq_form);
    if (key == "eq_wg_waves") {
        if (!one(&t.eq_wg_waves)) return false;
        const int w = t.eq_wg_waves;
        if (!(w == 0 || w == 1 || w == 2 || w == 4 || w == 8 || w == 16)) { t.eq_wg_waves = 0; return false; }
        return true;
    }
    if (key == "eq_lds") { if (!one(&t.eq_lds)) return false; if (t.eq_lds < 0 || t.eq_lds > 65536) t.eq_lds = 0; return true; }
    if (key == "no_overlap") return one(&t.no_overlap);
    if (key == "force_overlap") return one(&t.force_overlap);
    if (key == "xcd_split") { if (!one(&t.xcd_split)) return false; if (t.xcd_split < 0 || t.xcd_split > 7) t.xcd_split = 0; return true; }
    if (key == "overlap_cuts") {
        std::vector<double> c = {0.0};
        const char *p = value.c_str();
        while (*p && c.size() < 16) {
            char *end = nullptr;
            const double x = strtod(p, &end);
            if (end == p) break;
            if (x > c.back() && x < 1.0) c.push_back(x);
            p = (*end == ',') ? end + 1 : end;
        }
        c.push_back(1.0);
        if (c.size() < 3) return false;
        t.overlap_cuts = c;
        return true;
    }
    if (key == "conv_cu_mask") {
        std::vector<unsigned> m;
        const char *p = value.c_str();
        while (*p) {
            char *end = nullptr;
            const unsigned long x = strtoul(p, &end, 16);
            if (end == p) break;
            m.push_back((unsigned)x);
            p = (*end == ',') ? end + 1 : end;
        }
        t.conv_cu_mask = m;
        return true;
    }
    if (key == "mac_no_xcd") return one(&t.mac_no_xcd);
    if (key == "lb_skip") return one(&t.lb_skip);
    if (key == "xb_min_p") return one(&t.xb_min_p);
    if (key == "xb_min_blocks") return one(&t.xb_min_blocks);
    if (key == "xb_min_streams_p2") return one(&t.xb_min_streams_p2);
    if (key == "xb_stagger") { if (!one(&t.xb_stagger)) return false; t.xb_stagger = std::max(0, std::min(t.xb_stagger, 64)); return true; }
    if (key == "xb_run") return one(&t.xb_run);
    if (key == "xb_append_launch") return one(&t.xb_append_launch);
    if (key == "xb_form") { if (!one(&t.xb_form)) return false; t.xb_form = t.xb_form == 2 ? 2 : 1; return true; }
    if (key == "xb_skip") return one(&t.xb_skip);
    if (key == "lb_seg_windows") { if (!one(&t.lb_seg_windows)) return false; if (t.lb_seg_windows < 64) t.lb_seg_windows = 64; return true; }
    if (key == "lb_min_p") { if (!one(&t.lb_min_p)) return false; if (t.lb_min_p < 2) t.lb_min_p = 2; return true; }
    if (key == "host_pipe_trace") return one(&t.host_pipe_trace);
    if (key == "host_pipe_flat_priorities") return one(&t.host_pipe_flat_priorities);
    if (key == "node_shared_device_rehearsal") return one(&t.node_shared_device_rehearsal);
    if (key == "node_inject_table_failure_slot") return one(&t.node_inject_table_failure_slot);
    if (key == "rt_max_resident") { int v = 0; if (!one(&v)) return false; rt_max_resident_store(v); return true; }
    if (key == "rt_idle_polls") { rt_idle_polls_store(strtoull(value.c_str(), nullptr, 10)); return true; }
    if (key == "rt_poll_sleep") { if (!one(v)) return false; t.rt_poll_sleep = (unsigned)v[0]; return true; }
    return false;
}
#endif

}  // namespace ohs
