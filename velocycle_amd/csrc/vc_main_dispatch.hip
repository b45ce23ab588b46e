// Lookup of the compiled likelihood-kernel instantiation for (H, NB, kind, noise, genes per lane, count storage | 2 gradient-only).
#include "vc_main_kernel.h"
extern const VcMainEntry vc_tab_phase_nb[30];
extern const VcMainEntry vc_tab_phase_poisson[30];
extern const VcMainEntry vc_tab_phase_lognormal[30];
extern const VcMainEntry vc_tab_vfull_nb[30];
extern const VcMainEntry vc_tab_vfull_poisson[30];
extern const VcMainEntry vc_tab_vfull_lognormal[30];
extern const VcMainEntry vc_tab_vu_nb[30];
extern const VcMainEntry vc_tab_vu_poisson[30];
extern const VcMainEntry vc_tab_vu_lognormal[30];
extern const VcMainEntry vc_tab_phase_nb_u16[30];
extern const VcMainEntry vc_tab_phase_poisson_u16[30];
extern const VcMainEntry vc_tab_vfull_nb_u16[30];
extern const VcMainEntry vc_tab_vfull_poisson_u16[30];
extern const VcMainEntry vc_tab_vu_nb_u16[30];
extern const VcMainEntry vc_tab_vu_poisson_u16[30];
extern const VcMainEntry vc_tab_vu_nb_nl[30];          // gradient-only (c16 | 2)
extern const VcMainEntry vc_tab_vu_nb_u16_nl[30];
extern const VcMainEntry vc_tab_vfull_nb_nl[30];
extern const VcMainEntry vc_tab_vfull_nb_u16_nl[30];
extern const VcMainEntry vc_tab_phase_nb_nl[30];
extern const VcMainEntry vc_tab_phase_nb_u16_nl[30];
extern const VcMainEntry vc_tab_vu_nb_pwl[30];         // nu_omega partials per lane (c16 | 4), + gradient-only (| 2)
extern const VcMainEntry vc_tab_vu_nb_u16_pwl[30];
extern const VcMainEntry vc_tab_vu_nb_pwl_nl[30];
extern const VcMainEntry vc_tab_vu_nb_u16_pwl_nl[30];

vc_main_launch_fn vc_find_main_kernel(int H, int NB, int kind, int noise, int gpl, int c16, const char** name,
                                      const void** kernel) {
  static const VcMainEntry* tabs[] = {vc_tab_phase_nb, vc_tab_phase_poisson, vc_tab_phase_lognormal, vc_tab_vfull_nb,
                                      vc_tab_vfull_poisson, vc_tab_vfull_lognormal, vc_tab_vu_nb, vc_tab_vu_poisson,
                                      vc_tab_vu_lognormal, vc_tab_phase_nb_u16, vc_tab_phase_poisson_u16,
                                      vc_tab_vfull_nb_u16, vc_tab_vfull_poisson_u16, vc_tab_vu_nb_u16, vc_tab_vu_poisson_u16,
                                      vc_tab_vu_nb_nl, vc_tab_vu_nb_u16_nl, vc_tab_vfull_nb_nl, vc_tab_vfull_nb_u16_nl,
                                      vc_tab_phase_nb_nl, vc_tab_phase_nb_u16_nl, vc_tab_vu_nb_pwl, vc_tab_vu_nb_u16_pwl,
                                      vc_tab_vu_nb_pwl_nl, vc_tab_vu_nb_u16_pwl_nl};
  static const char* tab_names[] = {"phase_nb", "phase_poisson", "phase_lognormal", "vfull_nb", "vfull_poisson",
                                    "vfull_lognormal", "vu_nb", "vu_poisson", "vu_lognormal", "phase_nb", "phase_poisson",
                                    "vfull_nb", "vfull_poisson", "vu_nb", "vu_poisson", "vu_nb_gradonly", "vu_nb_gradonly",
                                    "vfull_nb_gradonly", "vfull_nb_gradonly", "phase_nb_gradonly", "phase_nb_gradonly",
                                    "vu_nb", "vu_nb", "vu_nb_gradonly", "vu_nb_gradonly"};
  for (unsigned t = 0; t < sizeof(tabs) / sizeof(tabs[0]); ++t)
    for (int i = 0; i < 30; ++i) {
      const VcMainEntry& e = tabs[t][i];
      if (e.H == H && e.NB == NB && e.kind == kind && e.noise == noise && e.gpl == gpl && e.c16 == c16) {
        if (name) *name = tab_names[t];
        if (kernel) *kernel = e.kernel;
        return e.fn;
      }
    }
  return nullptr;
}
