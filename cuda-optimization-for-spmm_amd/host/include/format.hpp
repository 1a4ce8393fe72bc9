// Umbrella header of the format layer (dense + the four sparse containers).
#pragma once
#include "formats/sparse.hpp"
