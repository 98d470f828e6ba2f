// advanced_ssr.hpp — kept so that `#include "advanced_ssr.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
