// Instantiations of the likelihood kernel: kind=phase, noise=poisson, H in 1..3, NB in 0..4.
#include "vc_main_kernel.h"
VC_DEFINE_TABLE(vc_tab_phase_poisson, VC_KIND_PHASE, VC_NOISE_POISSON)
