// row-chained dctr_embed_mlp_fwd kernel for sigmoid / tanh DNNs (reference layers/activation.py:75-85: activation_layer -> Activation;
// layers/core.py:189-208): the activation of an accumulator set on the transcendental units (chain_device.h: act_block EXPACT).  The
// throughput shape (256-row passes + in-kernel tail), DNN units[0] = 4 x 64, units[1] = 2 x 64 (other widths reach it zero-padded),
// every third-layer width, embedding_dim 16 / 32
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 1
#define DCTR_CHAIN_EXPACT 1
#include "chain_launch.inc"
