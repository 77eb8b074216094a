// Shared helper of the two drop-in classes: open the GPU context they upload
// their tables to.
#pragma once
struct mdc_ctx;
namespace mdc_host {
// Device = $MDC_DEVICE if set, else the calling thread's current HIP device.
// Returns 0 (and says so on stderr, naming `who`) when no GPU is usable; the
// classes then keep their host tables but every per-frame call fails loudly.
mdc_ctx* open_device_context(const char* who);
}  // namespace mdc_host
