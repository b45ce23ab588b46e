// Instantiations of the likelihood kernel: kind=vfull, noise=nb, H in 1..3, NB in 0..4.
#include "vc_main_kernel.h"
VC_DEFINE_TABLE(vc_tab_vfull_nb, VC_KIND_VFULL, VC_NOISE_NB)
