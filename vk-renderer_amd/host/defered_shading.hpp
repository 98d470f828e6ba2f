// defered_shading.hpp — kept so that `#include "defered_shading.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
#include "imgui_pass.hpp"  // defered_shading.hpp:6 includes it, and defered_shading.cpp relies on that
