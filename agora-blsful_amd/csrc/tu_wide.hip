// translation unit: the row-wide Fp arithmetic of wide.cuh (single-verification latency path)
#define BLS_TU_WIDE 1
#include "kernels.cuh"
