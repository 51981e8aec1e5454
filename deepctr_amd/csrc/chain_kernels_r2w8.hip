// row-chained dctr_embed_mlp_fwd kernel, 256 batch rows per pass: 8 waves x 32 rows, two waves per SIMD (the throughput shape); see chain_device.h
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#include "chain_launch.inc"
