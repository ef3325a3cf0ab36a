// translation unit: kernels of the BLS_TU_CYCRUN4 section of kernels.cuh (four waves per SIMD: non-kernel functions get that budget too)
#define BLS_TU_CYCRUN4 1
#include "kernels.cuh"
