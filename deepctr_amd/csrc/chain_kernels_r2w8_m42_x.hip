// row-chained dctr_embed_mlp_fwd kernel WITH the folded vector CrossNet (DCN's cross branch: L + 1 dot products of the gathered row
// taken beside the layer-0 MFMAs, dctr_mlp_args_t.cross_*): the throughput shape (256-row passes + in-kernel tail), DNN units[0] =
// 4 x 64, units[1] = 2 x 64 (other widths reach it zero-padded), every third-layer width; see chain_device.h, mlp_device.h
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 1
#define DCTR_CHAIN_CROSS 1
#include "chain_launch.inc"
