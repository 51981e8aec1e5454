// row-chained dctr_embed_mlp_fwd kernel for embedding_dim 8 and 4 (the reference's default is 4): several fields per 16-wide k-block
// (chain_device.h: FPB).  The throughput shape (256-row passes + in-kernel tail), DNN units[0] = 4 x 64, units[1] = 2 x 64 (other widths
// reach it zero-padded), every third-layer width
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 1
#define DCTR_CHAIN_SMALLE 1
#include "chain_launch.inc"
