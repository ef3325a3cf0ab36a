// translation unit: kernels of the BLS_TU_SIGN2 section of kernels.cuh
#define BLS_TU_SIGN2 1
#include "kernels.cuh"
