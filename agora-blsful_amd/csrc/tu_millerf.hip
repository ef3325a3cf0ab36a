// translation unit: kernels of the BLS_TU_MILLERF section of kernels.cuh
#define BLS_TU_MILLERF 1
#include "kernels.cuh"
