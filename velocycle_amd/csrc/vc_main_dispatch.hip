// Lookup of the compiled likelihood-kernel instantiation for (H, NB, kind, noise).
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

vc_main_launch_fn vc_find_main_kernel(int H, int NB, int kind, int noise, int gpl, const char** name,
                                      const void** kernel) {
  static const VcMainEntry* tabs[] = {vc_tab_phase_nb, vc_tab_phase_poisson, vc_tab_phase_lognormal, vc_tab_vfull_nb, vc_tab_vfull_poisson, vc_tab_vfull_lognormal, vc_tab_vu_nb, vc_tab_vu_poisson, vc_tab_vu_lognormal};
  static const char* tab_names[] = {"phase_nb", "phase_poisson", "phase_lognormal", "vfull_nb", "vfull_poisson", "vfull_lognormal", "vu_nb", "vu_poisson", "vu_lognormal"};
  for (unsigned t = 0; t < sizeof(tabs) / sizeof(tabs[0]); ++t)
    for (int i = 0; i < 30; ++i) {
      const VcMainEntry& e = tabs[t][i];
      if (e.H == H && e.NB == NB && e.kind == kind && e.noise == noise && e.gpl == gpl) {
        if (name) *name = tab_names[t];
        if (kernel) *kernel = e.kernel;
        return e.fn;
      }
    }
  return nullptr;
}
