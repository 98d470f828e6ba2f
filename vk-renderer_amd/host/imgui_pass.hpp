// imgui_pass.hpp — the reference's pass sources include "imgui_pass.hpp" for their draw_ui() panels
// (gtao.cpp:528-536, advanced_ssr.cpp:556-567, defered_shading.cpp:120-126; src/imgui_pass.hpp:1-20).  This build is
// headless: the UI is out of scope, so the handful of ImGui names those panels spell are inert
// here — a widget never reports a change and leaves its value alone — and the reference sources compile unchanged
// against host/ (tests/test_reference_sources_compile.py).
#ifndef VKR_HOST_IMGUI_PASS_HPP_INCLUDED
#define VKR_HOST_IMGUI_PASS_HPP_INCLUDED
#include "rendergraph/rendergraph.hpp"

struct SDL_Window;

namespace ImGui {
inline bool Begin(const char*, bool* = nullptr, int = 0) { return true; }
inline void End() {}
inline bool Checkbox(const char*, bool*) { return false; }
inline bool SliderFloat(const char*, float*, float, float, const char* = "%.3f", int = 0) { return false; }
inline bool SliderInt(const char*, int*, int, int, const char* = "%d", int = 0) { return false; }
inline bool Button(const char*) { return false; }
inline void Text(const char*, ...) {}
}  // namespace ImGui

inline void imgui_init(SDL_Window*, VkRenderPass) {}
inline void imgui_draw(VkCommandBuffer) {}
inline void imgui_close() {}

#endif
