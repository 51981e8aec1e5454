// row-chained dctr_embed_mlp_fwd kernel, 64 batch rows per pass: 4 waves x 16 rows; see chain_device.h
#define DCTR_CHAIN_RT 1
#define DCTR_CHAIN_NW 4
#include "chain_launch.inc"
