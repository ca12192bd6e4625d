// The reference's algorithm headers include this path literally
// (graphblas/algorithm/bfs.hpp:8); everything lives in graphblas/graphblas.hpp here.
#ifndef GRAPHBLAST_AMD_BACKEND_UTIL_FORWARD_HPP_
#define GRAPHBLAST_AMD_BACKEND_UTIL_FORWARD_HPP_
#include "graphblas/graphblas.hpp"
#endif
