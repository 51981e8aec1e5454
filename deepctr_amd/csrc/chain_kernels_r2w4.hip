// row-chained dctr_embed_mlp_fwd kernel, 128 batch rows per pass: 4 waves x 32 rows; see chain_device.h
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 4
#include "chain_launch.inc"
