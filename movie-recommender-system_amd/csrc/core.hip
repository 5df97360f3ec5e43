// Library-wide entry points: version, thread-local error string, supported widths.
#include "common.h"

namespace mirec {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace mirec

// 101: cg_steps = MIREC_SOLVE_EXACT (and conf_max in mirec_als_half_sweep: a change of signature)
// 102: y_absmax in mirec_gram, mirec_gram_static and mirec_als_half_sweep (a change of signature); mirec_topk_small
extern "C" int mirec_version(void) { return 102; }

extern "C" const char* mirec_last_error(void) { return mirec::g_error; }

extern "C" int mirec_factor_width(int f, int for_scoring) {
    if (f < 1) return -1;
    if (for_scoring) {
        static const int w[] = {8, 16, 24, 32, 48, 64, 96, 128, 160, 192};
        for (int v : w)
            if (f <= v) return v;
        return -1;
    }
    static const int w[] = {32, 64, 96, 128, 160, 192};
    for (int v : w)
        if (f <= v) return v;
    return -1;
}
