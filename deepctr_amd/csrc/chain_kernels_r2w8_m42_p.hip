// row-chained dctr_embed_mlp_fwd kernel WITH in-pass sequence pooling (VarLenSparseFeat with combiner sum / mean: reference
// deepctr/inputs.py:120-158, layers/sequence.py:76-106): the rows of a sample's sequences are requested two positions per layer-0 step
// beside the MFMAs, the pooled vector becomes the B operand of the sequence's k-block — no dctr_embed_pool pre-pass, no pooled buffer in
// HBM.  The throughput shape (256-row passes + in-kernel tail), embedding_dim 16, DNN units[0] = 4 x 64, units[1] = 2 x 64 (other widths
// reach it zero-padded), every third-layer width; see chain_device.h (pool_piece)
#define DCTR_CHAIN_RT 2
#define DCTR_CHAIN_NW 8
#define DCTR_CHAIN_M0 4
#define DCTR_CHAIN_M1 2
#define DCTR_CHAIN_M2SET 1
#define DCTR_CHAIN_POOL 1
#include "chain_launch.inc"
