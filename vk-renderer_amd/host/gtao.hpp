// gtao.hpp — kept so that `#include "gtao.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
