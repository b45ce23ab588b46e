// Gradient-only twin (vc_set_loss_every) of the per-lane U-only kernel, float32 count storage: noise=nb, H in 1..3, NB in 0..4.
#include "vc_main_kernel.h"
VC_DEFINE_TABLE_CS(vc_tab_vu_nb_pwl_nl, VC_KIND_VU, VC_NOISE_NB, 6)
