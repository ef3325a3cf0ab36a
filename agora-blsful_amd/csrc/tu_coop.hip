// translation unit: kernels of the BLS_TU_COOP section of kernels.cuh
#define BLS_TU_COOP 1
#include "kernels.cuh"
