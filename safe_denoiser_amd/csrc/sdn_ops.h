// Internal (non-ABI) declarations shared between the translation units of libsdn.
#pragma once
#include "../../include/sdn.h"

int sdn_gemm_pick_nrep(int n_padded, int act);
// NREP actually launched for this shape (10 / 8 = the 256-row, 8-wave tile; 5 / 4 / 2 / 1 = the 128-row tile).
// epilogue_reads: the epilogue fetches a residual tile / row gates (keeps short k loops on the 2-blocks-per-CU tile).
int sdn_gemm_pick_tile(int M, int N, int K, int act, int epilogue_reads = 0);

// graph mode of the plan runner: the step's timestep lives in device memory so that a captured forward can be replayed
int sdn_temb_from_device(int dtype, const float* t_dev, int batch, int dim, void* out, void* stream);
int sdn_set_scalar(float* dst, float v, void* stream);

// k-loop slices the split-K form should use for this shape (1 = do not split)
int sdn_gemm_pick_split(int M, int N, int K, int act, int out_kind);

// fp32 precision mode (sdn_f32.hip): timestep features from a host scalar or from device memory (graph mode)
int sdn_temb_f32(float timestep, const float* t_dev, int batch, int dim, void* out, void* stream);

// Two linears with nothing but a residual between them, out = Wb (h + Wa f + ba) + bb, as one contraction over [f | h]:
// w_cat [C][K + C] = [Wb Wa | Wb] (16 bit, product formed in fp32 and rounded once), b_cat[C] = Wb ba + bb.
// wa [C][K], wb [C][C] in the plan's 16-bit dtype (0 = bf16, 1 = f16).  Prepare-time helper (sdn_unet_prepare).
int sdn_linear_pair_fold(int dtype, const void* wa, const void* wb, const float* ba, const float* bb, int C, int K, void* w_cat,
                         float* b_cat, void* stream);


// 3x3 convolutions the slab-ring kernel (sdn_conv.hip) takes over from the implicit-GEMM tile: shape part of the test
// (square side x side map, M = batch * side^2 output rows, N padded output channels).
int sdn_conv_slab_shape_ok(int M, int N, int Cin, int side, int stride, int upsample, int asym_pad, int out_kind, int n_valid);
