// roctx_ranges.hpp -- named ranges around the library's phases for `rocprofv3 --marker-trace` (the reference has
// no tracing; the build adds roctx ranges beside the rocprofv3 counter recipes).  The marker library is looked up at run time
// (no link dependency: the product library must load on a box without the profiler's packages) and ONLY when asked for:
// SKL_ROCTX=1, or a rocprofv3 / rocprofiler-sdk tool already in the process (ROCP_TOOL_LIBRARIES, ROCPROFILER_* set by the
// profiler's launcher).  A default run never loads or calls into profiler libraries; a range then costs one predictable branch.
#pragma once

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

namespace skl {

struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    static bool wanted()
    {
        const char *e = getenv("SKL_ROCTX");
        if (e && *e) return strcmp(e, "0") != 0;
        for (const char *v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_MARKER_API_TRACE", "ROCPROFILER_LIBRARY_CTOR"}) {
            const char *x = getenv(v);
            if (x && *x) return true;
        }
        return false;
    }
    RoctxApi()
    {
        if (!wanted()) return;
        for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
    static const RoctxApi &get()
    {
        static const RoctxApi api;
        return api;
    }
};

// RAII range on the calling host thread: what the kernels enqueued inside it belong to.
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(RoctxApi::get().push != nullptr)
    {
        if (on) (void)RoctxApi::get().push(name);
    }
    ~RoctxRange()
    {
        if (on) (void)RoctxApi::get().pop();
    }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};

}  // namespace skl
