// translation unit: kernels of the BLS_TU_MSM1 section of kernels.cuh
#define BLS_TU_MSM1 1
#include "kernels.cuh"
