#pragma once
#include "common.h"
int quantize_launch(const float* x, int rows, int n, int ld, double* mnmx_ws, void* out, int ldo, int mode,
                    int q_levels, hipStream_t stream);
int mu2linear_launch(const int32_t* q, size_t n, float* out, hipStream_t stream);
