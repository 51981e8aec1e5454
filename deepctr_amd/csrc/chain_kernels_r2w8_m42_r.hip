// row-chained dctr_embed_mlp_fwd kernel for RECORD-form tables (dctr_field_t.row_pitch = 32, embedding_dim 16: a row's first-order weight lies
// behind it in one 128-B record, chain_device.h: REC).  The throughput shape (256-row passes + in-kernel tail), DNN units[0] = 4 x 64,
// units[1] = 2 x 64 (other widths reach it zero-padded), every third-layer width, ReLU / linear, int32 / int64 ids
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 1
#define DCTR_CHAIN_REC 1
#include "chain_launch.inc"
