// Shared between the fp32 convolution translation units (not part of the C ABI).
#pragma once
#include "sf_common.h"

struct SfConvArgs {
  const float *feat;
  unsigned feat_bytes;   // size of the input feature table: the raw-buffer gathers are bounded by it
  const int *nbr;
  int nbr_stride, K, k_first;
  const float *wpk, *scale, *shift, *residual;
  int relu;
  float *out;
  const int *n_out_dev;
  int n_out_bound;
  hipStream_t st;
};

// spconv_f32_narrow.hip: 3x3x3 layers with cin_padded <= 16 and cout <= 32.  Returns SF_OK after a
// launch, 1 when the shape is not one of its own (caller falls through to the general kernels),
// or a negative SF_ERR_* code.
int sf_narrow_conv_f32(int cin_padded, int cout, const SfConvArgs &a);
