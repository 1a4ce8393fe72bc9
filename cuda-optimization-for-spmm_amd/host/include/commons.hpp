// Shared host-side definitions of the cuspmm engine (MI355X build).
// Mirrors the role of /root/reference/include/commons.hpp: index helpers and the type gate.
#pragma once

#include <cassert>
#include <chrono>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <type_traits>

#include "utils.hpp"  // <cstdint>, <iostream>, <string>, the record printer and the tolerances

// element (r, c) of a matrix with B columns stored row-major / with A rows stored column-major
#define RowMjIdx(r, c, B) ((size_t)(r) * (size_t)(B) + (size_t)(c))
#define ColMjIdx(r, c, A) ((size_t)(c) * (size_t)(A) + (size_t)(r))

#define assertTypes3(DT, ta, tb, tc)                                                             \
    static_assert(std::is_same_v<DT, ta> || std::is_same_v<DT, tb> || std::is_same_v<DT, tc>,   \
                  "Unsupported type")
