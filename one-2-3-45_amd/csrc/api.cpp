// Error reporting + version for the o2345 C-ABI (see include/o2345.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

namespace o2345 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace o2345

extern "C" {
const char* o2345_last_error(void) { return o2345::g_err; }
int o2345_version(void) { return 150; }     // 1.5: o2345_list_sort_by_visibility (render work-list grouped by visibility); 1.4: color stats, bf16 entry removed; 1.2: O2345RenderIO.t_rand + o2345_ray_coarse_jitter (perturb > 0); 1.3: o2345_conv2d family
}
