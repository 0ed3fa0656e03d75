// The library's one process-wide record of launch-heuristic constants (amt_common.h: AmtTuning).
#include <stdlib.h>

#include <mutex>

#include "amt_common.h"

// (see amt_common.h: immutable after the first call; the environment is consulted by -DAMT_EXPERIMENT builds only)
const AmtTuning& amt_tuning() {
    static AmtTuning t;
#ifdef AMT_EXPERIMENT
    static std::once_flag once;
    std::call_once(once, [] {
        auto geti = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
        auto getl = [](const char* name, long& v) { if (const char* e = getenv(name)) v = atol(e); };
        geti("AMT_KV_PAD", t.kv_pad); geti("AMT_STEPS_PER_GRAPH", t.steps_per_graph); geti("AMT_NT", t.nt_mask);
        geti("AMT_WIDE_GROUPED", t.wide_grouped); geti("AMT_WIDE_NTW", t.wide_ntw);
        getl("AMT_GEMM_SMALL_M", t.gemm_small_m); getl("AMT_GEMM_SMALL_MN", t.gemm_small_mn);
        geti("AMT_GEMM_T64_BELOW", t.gemm_t64_below); geti("AMT_GEMM_PF", t.gemm_pf);
        int dbg = 0; geti("AMT_DBG", dbg); t.prepacked = (dbg & 16) ? 1 : 0;
        if (t.steps_per_graph <= 0) t.steps_per_graph = 8;
    });
#endif
    return t;
}
