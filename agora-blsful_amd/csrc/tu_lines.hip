// translation unit: kernels of the BLS_TU_LINES section of kernels.cuh
#define BLS_TU_LINES 1
#include "kernels.cuh"
