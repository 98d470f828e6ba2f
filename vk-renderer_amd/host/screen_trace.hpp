// screen_trace.hpp — kept so that `#include "screen_trace.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
