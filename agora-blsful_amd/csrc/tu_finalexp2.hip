// translation unit: kernels of the BLS_TU_FINALEXP2 section of kernels.cuh
#define BLS_TU_FINALEXP2 1
#include "kernels.cuh"
