// translation unit: kernels of the BLS_TU_POINTS section of kernels.cuh
#define BLS_TU_POINTS 1
#include "kernels.cuh"
