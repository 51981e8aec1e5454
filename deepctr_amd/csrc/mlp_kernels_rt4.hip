// dctr_mlp_fwd / dctr_embed_mlp_fwd kernel for 64 batch rows per workgroup (RT = 4 row tiles); see mlp_device.h
#define DCTR_MLP_RT 4
#include "mlp_launch.inc"
