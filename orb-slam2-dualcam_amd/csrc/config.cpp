// config.cpp -- the library's options (config.h) and the ONE place that reads the environment.
#include "config.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/dcs_abi.h"
#include "common.h"

namespace dcs {
namespace {

struct Def { const char* name; long long def; };
const Def kDefs[OPT_COUNT] = {
#define DCS_OPT_DEF(name, def) {"DCS_" #name, def},
    DCS_OPTIONS(DCS_OPT_DEF)
#undef DCS_OPT_DEF
};
std::atomic<long long> g_val[OPT_COUNT];
std::once_flag g_once;

void seed_from_environment()
{
    for (int i = 0; i < OPT_COUNT; ++i) {
        long long v = kDefs[i].def;
        if (const char* e = getenv(kDefs[i].name)) {          // the library's only getenv (+ the legacy spellings below)
            char* end = nullptr;
            const double d = strtod(e, &end);                 // "2.5e7" style values too
            if (end != e) v = (long long)d;
            else if (*e) v = 1;                               // set, not a number ("on", "yes"): on. An EMPTY value leaves the default (atoi gave 0 before round 5, "on" in round 5: neither was meant)
        }
        g_val[i].store(v, std::memory_order_relaxed);
    }
    // two historical spellings: DCS_BA_CUS=first:count, DCS_FAST_HW_PROBE=fail
    if (const char* e = getenv("DCS_FAST_HW_PROBE")) if (strcmp(e, "fail") == 0) g_val[OPT_FAST_HW_PROBE_FAIL] = 1;
    // switches of earlier rounds that no longer exist: the documented safety fallback keeps its meaning, the others are named once on stderr instead of
    // being dropped silently
    if (const char* e = getenv("DCS_FAST_D16Z")) if (*e && atoi(e) == 0) g_val[OPT_FAST_HW_PROBE_FAIL] = 1;     // "plain byte loads" = the probe-failed forms
    static const char* const kGone[] = {"DCS_BA_LDLT_VALU", "DCS_BA_SCHUR_WIDE", "DCS_BA_SCHUR_MID", "DCS_BA_FRONT", "DCS_BA_CUS" /* dcs_ba_set_cu_range */, "DCS_BA_STREAM_PRIORITY", "DCS_KNN2_VALU", "DCS_PROJ_SERIAL"};
    for (const char* name : kGone)
        if (getenv(name)) fprintf(stderr, "[dcs] %s is no longer an option of this library and is ignored (dcs_option_count / dcs_option_name list the current ones)\n", name);
}

}  // namespace

long long opt(Opt o) { std::call_once(g_once, seed_from_environment); return g_val[o].load(std::memory_order_relaxed); }
int opt_find(const char* name)
{
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i) if (!strcmp(name, kDefs[i].name) || !strcmp(name, kDefs[i].name + 4)) return i;
    return -1;
}
const char* opt_name(int i) { return i >= 0 && i < OPT_COUNT ? kDefs[i].name : nullptr; }
void opt_set(int i, long long v) { std::call_once(g_once, seed_from_environment); g_val[i].store(v, std::memory_order_relaxed); }
long long opt_default(int i) { return kDefs[i].def; }

}  // namespace dcs

using namespace dcs;

extern "C" {

int dcs_option_count(void) { return OPT_COUNT; }
const char* dcs_option_name(int index) { return opt_name(index); }
int dcs_option_get(const char* name, int64_t* value)
{
    const int i = opt_find(name);
    if (i < 0 || !value) { set_error("dcs_option_get: unknown option '%s'", name ? name : "(null)"); return DCS_ERR_INVALID; }
    *value = (int64_t)opt((Opt)i);
    return DCS_OK;
}
int dcs_option_set(const char* name, int64_t value)
{
    const int i = opt_find(name);
    if (i < 0) { set_error("dcs_option_set: unknown option '%s'", name ? name : "(null)"); return DCS_ERR_INVALID; }
    opt_set(i, (long long)value);
    return DCS_OK;
}

}  // extern "C"
