// dctr_mlp_fwd / dctr_embed_mlp_fwd kernel for 32 batch rows per workgroup (RT = 2 row tiles); see mlp_device.h
#define DCTR_MLP_RT 2
#include "mlp_launch.inc"
