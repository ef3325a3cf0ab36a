// translation unit: the row-wide Fp arithmetic of wide.cuh (single-verification latency path)
#define BLS_TU_WIDE 1
#define BLS_FP_INV_VAR 1     // every inversion of this unit is one item's on a lone lane (pair): the variable-time safegcd
#include "kernels.cuh"
