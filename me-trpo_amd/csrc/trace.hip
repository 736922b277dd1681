#include <dlfcn.h>
#include <cstdlib>
#include <mutex>
#include "trace.h"

namespace {
int (*g_push)(const char*) = nullptr;
int (*g_pop)() = nullptr;
std::once_flag g_once;
void resolve() {
    const char* e = getenv("METRPO_ROCTX");
    if (e && e[0] == '0') return;
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    for (const char* n : names) {
        void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        g_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (g_push && g_pop) return;
        g_push = nullptr; g_pop = nullptr;
    }
}
}  // namespace

TraceRange::TraceRange(const char* name) : on(false) {
    std::call_once(g_once, resolve);
    if (g_push) { g_push(name); on = true; }
}
TraceRange::~TraceRange() { if (on && g_pop) g_pop(); }
