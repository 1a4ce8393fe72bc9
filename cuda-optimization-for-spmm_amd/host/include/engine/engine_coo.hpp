// Kept so `#include "engine/engine_coo.hpp"` still works: EngineCOO lives in engine/engines.hpp.
#pragma once
#include "engine/engines.hpp"
