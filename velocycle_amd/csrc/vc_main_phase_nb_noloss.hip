// Gradient-only instantiations of the likelihood kernel (vc_set_loss_every), float32 count storage: kind=phase, noise=nb, H in 1..3, NB in 0..4.
#include "vc_main_kernel.h"
VC_DEFINE_TABLE_CS(vc_tab_phase_nb_nl, VC_KIND_PHASE, VC_NOISE_NB, 2)
