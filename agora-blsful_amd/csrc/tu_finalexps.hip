// translation unit: kernels of the BLS_TU_FINALEXPS section of kernels.cuh
#define BLS_TU_FINALEXPS 1
#include "kernels.cuh"
