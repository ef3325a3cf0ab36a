// translation unit: kernels of the BLS_TU_FINALEXP section of kernels.cuh
#define BLS_TU_FINALEXP 1
#include "kernels.cuh"
