// U-only likelihood kernel with the nu_omega partials kept per lane (round 6: one condition, D == 1), uint16 count storage: noise=nb, H in 1..3, NB in 0..4.
#include "vc_main_kernel.h"
VC_DEFINE_TABLE_CS(vc_tab_vu_nb_u16_pwl, VC_KIND_VU, VC_NOISE_NB, 5)
