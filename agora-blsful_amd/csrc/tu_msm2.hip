// translation unit: kernels of the BLS_TU_MSM2 section of kernels.cuh
#define BLS_TU_MSM2 1
#include "kernels.cuh"
