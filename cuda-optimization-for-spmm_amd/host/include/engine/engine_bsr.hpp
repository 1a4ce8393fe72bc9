// Kept so `#include "engine/engine_bsr.hpp"` still works: EngineBSR lives in engine/engines.hpp.
#pragma once
#include "engine/engines.hpp"
