// scene_renderer.hpp — kept so that `#include "scene_renderer.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
