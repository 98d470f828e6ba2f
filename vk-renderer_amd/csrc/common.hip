// common.hip — library-wide pieces of the C-ABI: version, last-error string, format table.
#include "vkr_host.hpp"
#include <atomic>
#include <cstdarg>
#include <cstdlib>

namespace vkr {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace vkr

// Measurement switches: read from the environment ONCE (first launch that asks), changed afterwards only through vkr_set_switches().
namespace vkr {
static std::atomic<uint32_t> g_switches{0x80000000u};  // top bit: not initialised yet
uint32_t switches() {
  uint32_t v = g_switches.load(std::memory_order_relaxed);
  if (v & 0x80000000u) {
    v = 0u;
    if (getenv("VKR_BLUR_NO_SKIP")) v |= VKR_SWITCH_BLUR_NO_SKIP;
    if (getenv("VKR_FILTER_NO_SKIP")) v |= VKR_SWITCH_FILTER_NO_SKIP;
    if (getenv("VKR_TAA_GENERIC")) v |= VKR_SWITCH_TAA_GENERIC;
    if (getenv("VKR_SHADING_GENERIC")) v |= VKR_SWITCH_SHADING_GENERIC;
    if (getenv("VKR_BLUR_GENERIC")) v |= VKR_SWITCH_BLUR_GENERIC;
    if (getenv("VKR_TRACE_ONE_LAUNCH")) v |= VKR_SWITCH_TRACE_ONE_LAUNCH;
    if (getenv("VKR_BLUR_LANE_LOOPS")) v |= VKR_SWITCH_BLUR_LANE_LOOPS;
    uint32_t expected = 0x80000000u;
    if (!g_switches.compare_exchange_strong(expected, v)) v = expected;
  }
  return v;
}
}  // namespace vkr
extern "C" uint32_t vkr_get_switches(void) { return vkr::switches(); }
extern "C" void vkr_set_switches(uint32_t mask) { vkr::g_switches.store(mask & 0x7FFFFFFFu, std::memory_order_relaxed); }

extern "C" const char* vkr_version(void) { return "vkr_postfx 0.2 (gfx950)"; }
extern "C" uint32_t vkr_numeric_contract(void) { return VKR_CONTRACT; }
extern "C" const char* vkr_last_error(void) { return vkr::g_error; }

extern "C" uint32_t vkr_format_bytes(uint32_t format) {
  switch (format) {
    case VKR_FMT_D24_UNORM_S8: case VKR_FMT_RG16_UNORM: case VKR_FMT_RG16_SFLOAT:
    case VKR_FMT_RGBA8_SRGB: case VKR_FMT_RGBA8_UNORM: case VKR_FMT_R32_SFLOAT: return 4;
    case VKR_FMT_RGBA16_UNORM: case VKR_FMT_RGBA16_SFLOAT: return 8;
    case VKR_FMT_RGBA32_SFLOAT: return 16;
    case VKR_FMT_R16_SFLOAT: return 2;
    case VKR_FMT_R8_UNORM: return 1;
    default: return 0;
  }
}

// advanced_ssr.cpp:8-34 (float-floor division quirk kept) + host-evaluated cos/sin of 2*PI*y in zw
extern "C" void vkr_halton23_fill(float* out, uint32_t count) {
  auto halton_elem = [](uint32_t index, uint32_t base) {
    float f = 1.0f, r = 0.0f;
    uint32_t current = index;
    do {
      f = f / (float)base;
      r = r + f * (float)(current % base);
      current = (uint32_t)floorf((float)current / (float)base);
    } while (current > 0);
    return r;
  };
  const float PI = 3.1415926535897932384626433832795f;
  for (uint32_t i = 0; i < count; i++) {
    const float x = halton_elem(i + 1, 2), y = halton_elem(i + 1, 3);
    const float phi = (2.0f * PI) * y;
    out[4 * i + 0] = x;
    out[4 * i + 1] = y;
    out[4 * i + 2] = (float)cos((double)phi);
    out[4 * i + 3] = (float)sin((double)phi);
  }
}
