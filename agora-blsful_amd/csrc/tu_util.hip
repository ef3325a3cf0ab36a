// translation unit: the byte / index kernels of util_kernels.cuh (scan, key sort, coefficient hashes, duplicate rule)
#define BLS_TU_UTIL 1
#include "kernels.cuh"
