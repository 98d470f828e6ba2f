// image_readback.hpp — kept so that `#include "image_readback.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
