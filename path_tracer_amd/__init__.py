"""path_tracer_amd: MI355X-native path-tracing hot path (HIP/gfx950) behind a C ABI."""
