// scene.hpp — kept so that `#include "scene.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
