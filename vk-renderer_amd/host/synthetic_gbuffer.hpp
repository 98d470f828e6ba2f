// synthetic_gbuffer.hpp — kept so that `#include "synthetic_gbuffer.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
