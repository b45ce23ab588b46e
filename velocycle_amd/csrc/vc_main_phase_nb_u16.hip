// Instantiations of the likelihood kernel on uint16 count storage: kind=phase, noise=nb, H in 1..3, NB in 0..4.
#include "vc_main_kernel.h"
VC_DEFINE_TABLE_U16(vc_tab_phase_nb_u16, VC_KIND_PHASE, VC_NOISE_NB)
