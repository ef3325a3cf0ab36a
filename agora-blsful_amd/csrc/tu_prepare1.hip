// translation unit: kernels of the BLS_TU_PREPARE1 section of kernels.cuh
#define BLS_TU_PREPARE1 1
#include "kernels.cuh"
