// row-chained dctr_embed_mlp_fwd kernel for embedding_dim 64 (chain_device.h: EB = 4 — four 16-wide k-blocks per field, eight layer-0
// steps per field pair, the FM sums of four k-blocks in registers).  The throughput shape (256-row passes + in-kernel tail), DNN
// units[0] = 4 x 64, units[1] = 2 x 64 (other widths reach it zero-padded), every third-layer width, ReLU / linear
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 1
#define DCTR_CHAIN_WIDE 1
#include "chain_launch.inc"
