// Kept so `#include "engine/engine_ell.hpp"` still works: EngineELL lives in engine/engines.hpp.
#pragma once
#include "engine/engines.hpp"
