// Kept so `#include "formats/sparse_bsr.hpp"` still works: the class lives in formats/sparse.hpp.
#pragma once
#include "formats/sparse.hpp"
