#pragma once

#include "formats/dense.hpp"
#include "formats/matrix.hpp"
#include "formats/sparse_bsr.hpp"
#include "formats/sparse_coo.hpp"
#include "formats/sparse_csr.hpp"
#include "formats/sparse_ell.hpp"
