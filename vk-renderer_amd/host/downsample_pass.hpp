// downsample_pass.hpp — kept so that `#include "downsample_pass.hpp"` of the reference's sources resolves; the declarations live in passes.hpp.
#pragma once
#include "passes.hpp"
