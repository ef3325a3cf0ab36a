// translation unit: kernels of the BLS_TU_MILLER section of kernels.cuh
#define BLS_TU_MILLER 1
#include "kernels.cuh"
