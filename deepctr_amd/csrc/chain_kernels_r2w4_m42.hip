// row-chained dctr_embed_mlp_fwd kernel, 128 batch rows per pass (4 waves x 32 rows, one wave per SIMD): the forced shape
// tile_rows = 128 (a diagnostic shape: bit-identical results from another wave / pass membership), units 256-128-64 only
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 4
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 0
#include "chain_launch.inc"
