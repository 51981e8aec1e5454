// row-chained dctr_embed_mlp_fwd kernel, 256 batch rows per pass (8 waves x 32 rows, two waves per SIMD: the throughput shape) + the
// in-kernel tail phase, DNN units[0] = 4 x 64, units[1] = 2 x 64, every third-layer width; see chain_device.h
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 1
#include "chain_launch.inc"
