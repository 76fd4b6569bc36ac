// Internal launch helpers shared between translation units of libhortihip.
#pragma once
#include "hm_common.h"

namespace hm {

int launch_latent_bias(const hm_decoder_s* dec, const float* d_latent, int ld_latent, const int* d_active,
                       int B, float* d_c0, float* d_c4, hipStream_t stream);

int launch_decoder(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                   int n_stride, const float* d_c0, const float* d_c4, float* d_y, float* d_J, int ldJ,
                   int pose_dim, int mode, hipStream_t stream);

}  // namespace hm
