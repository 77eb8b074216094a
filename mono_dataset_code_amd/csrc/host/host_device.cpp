#include "host_device.h"

#include <cstdio>
#include <cstdlib>

#include "mdc_hip.h"

namespace mdc_host {

mdc_ctx* open_device_context(const char* who) {
  int device = -1;  // current device
  if (const char* e = std::getenv("MDC_DEVICE")) device = std::atoi(e);
  mdc_ctx* ctx = 0;
  if (mdc_create(device, &ctx) != MDC_OK) {
    std::fprintf(stderr, "%s: no GPU context (%s); tables were built on the host but per-frame calls will fail -- there is no CPU fallback\n",
                 who, mdc_last_error(0));
    return 0;
  }
  return ctx;
}

}  // namespace mdc_host
