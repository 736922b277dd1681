// roctx ranges around the phases of the inner loop (the reference accounts policy_time / env_time /
// process_time in samplers/vectorized_sampler.py:54-56,106 and policy_opt_time in model_based_rl.py:694 with time.time()).
// The ranges are emitted from the C entry points, so they show up in `rocprofv3 --marker-trace` whatever drives the library.
// librocprofiler-sdk-roctx (or the legacy libroctx64) is resolved at first use; METRPO_ROCTX=0 disables it; absent library = no-op.
#pragma once
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
    bool on;
};
