// The library's one process-wide record of launch-heuristic constants (amt_common.h: AmtTuning).
#include <stdlib.h>

#include <mutex>

#include "amt_common.h"

// (see amt_common.h: immutable after the first call; the environment is consulted by -DAMT_EXPERIMENT builds only)
#ifdef AMT_EXPERIMENT
static AmtTuning& tuning_record() { static AmtTuning t; return t; }
#endif

const AmtTuning& amt_tuning() {
#ifndef AMT_EXPERIMENT
    static const AmtTuning t;
#else
    AmtTuning& t = tuning_record();
    static std::once_flag once;
    std::call_once(once, [&t] {
        auto geti = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
        auto getl = [](const char* name, long& v) { if (const char* e = getenv(name)) v = atol(e); };
        geti("AMT_KV_PAD", t.kv_pad); geti("AMT_STEPS_PER_GRAPH", t.steps_per_graph); geti("AMT_NT", t.nt_mask);
        geti("AMT_WIDE_GROUPED", t.wide_grouped); geti("AMT_WIDE_NTW", t.wide_ntw); geti("AMT_WIDE_RB2", t.wide_rb2);
        getl("AMT_GEMM_SMALL_M", t.gemm_small_m); getl("AMT_GEMM_SMALL_MN", t.gemm_small_mn);
        geti("AMT_GEMM_T64_BELOW", t.gemm_t64_below); geti("AMT_GEMM_PF", t.gemm_pf);
        int dbg = 0; geti("AMT_DBG", dbg); t.prepacked = (dbg & 16) ? 1 : 0;
        if (t.steps_per_graph <= 0) t.steps_per_graph = 16;
    });
#endif
    return t;
}

#ifdef AMT_EXPERIMENT
#include <string.h>
// Experiment builds only (never in the release library: tests/test_host.py checks the export list): switch a field between two timed
// runs of ONE process, where box-to-box and process-to-process noise (+-1.5 %) would drown a 1 % effect.  Single-threaded use.
extern "C" int32_t amt_experiment_set(const char* name, int32_t value) {
    (void)amt_tuning();
    AmtTuning& t = tuning_record();
    if (!strcmp(name, "wide_rb2")) t.wide_rb2 = value;
    else if (!strcmp(name, "wide_ntw")) t.wide_ntw = value;
    else if (!strcmp(name, "wide_grouped")) t.wide_grouped = value;
    else if (!strcmp(name, "nt_mask")) t.nt_mask = value;
    else if (!strcmp(name, "steps_per_graph")) t.steps_per_graph = value;
    else if (!strcmp(name, "gemm_pf")) t.gemm_pf = value;
    else if (!strcmp(name, "exp_a")) t.exp_a = value;
    else if (!strcmp(name, "exp_b")) t.exp_b = value;
    else return -1;
    return 0;
}
#endif
