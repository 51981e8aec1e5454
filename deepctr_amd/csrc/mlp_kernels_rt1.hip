// dctr_mlp_fwd / dctr_embed_mlp_fwd kernel for 16 batch rows per workgroup (RT = 1 row tiles); see mlp_device.h
#define DCTR_MLP_RT 1
#include "mlp_launch.inc"
