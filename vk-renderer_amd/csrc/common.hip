// common.hip — library-wide pieces of the C-ABI: version, last-error string, format table.
#include "vkr_host.hpp"
#include <cstdarg>

namespace vkr {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace vkr

extern "C" const char* vkr_version(void) { return "vkr_postfx 0.1 (gfx950)"; }
extern "C" const char* vkr_last_error(void) { return vkr::g_error; }

extern "C" uint32_t vkr_format_bytes(uint32_t format) {
  switch (format) {
    case VKR_FMT_D24_UNORM_S8: case VKR_FMT_RG16_UNORM: case VKR_FMT_RG16_SFLOAT:
    case VKR_FMT_RGBA8_SRGB: case VKR_FMT_RGBA8_UNORM: case VKR_FMT_R32_SFLOAT: return 4;
    case VKR_FMT_RGBA16_UNORM: case VKR_FMT_RGBA16_SFLOAT: return 8;
    case VKR_FMT_R16_SFLOAT: return 2;
    case VKR_FMT_R8_UNORM: return 1;
    default: return 0;
  }
}
