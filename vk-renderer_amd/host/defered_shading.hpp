// defered_shading.hpp — kept so that `#include "defered_shading.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
