// translation unit: kernels of the BLS_TU_MILLERS section of kernels.cuh
#define BLS_TU_MILLERS 1
#include "kernels.cuh"
