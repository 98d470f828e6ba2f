// ssr.hpp — kept so that `#include "ssr.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
