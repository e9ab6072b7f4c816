/* octo_oracle.h — TEST INFRASTRUCTURE (see octo_oracle.c). PARITY UNPINNED. */
#ifndef OCTO_ORACLE_H
#define OCTO_ORACLE_H
#include <stdint.h>
#include "../include/octofitter_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Same buffer conventions as octo_eval (host SoA, walker fastest). active_mask
 * [n_planets*OCTO_N_EL + n_obs*OCTO_N_NUIS] selects which inputs carry a partial
 * (NULL = all); n_threads < 1 = all cores (OpenMP over walkers). */
int32_t octo_oracle_eval(const octo_consts* c,
                         const octo_obs_desc* obs, int32_t n_obs,
                         const octo_planet_desc* planets, int32_t n_planets,
                         const double* elems, const double* nuis, int64_t ld, int64_t W,
                         double* ll_out, double* g_elems, double* g_nuis,
                         const uint8_t* active_mask, int32_t n_threads);
double octo_oracle_kepler_markley(double MA, double e);
int32_t octo_oracle_orbitsolve(const octo_consts* c, int32_t orbit_kind, const double* el9, double t, double* out10);
int32_t octo_oracle_consts_default(octo_consts* out);
int32_t octo_oracle_ofti(const octo_consts* c, const double* epochs, const double* ra, const double* dec, const double* s_ra,
                         const double* s_dec, const double* cor, int64_t N, double sigma_abfg, const double* nl5, double* out5);
int32_t octo_oracle_max_partials(void);
int32_t octo_oracle_model_logpost(const octo_consts* c, const octo_obs_desc* obs, int32_t n_obs,
                                  const octo_planet_desc* planets, int32_t n_planets,
                                  const octo_prior* priors, int32_t D, const octo_source* elem_src, const octo_source* nuis_src,
                                  const double* theta_t, int64_t ld, int64_t W, double* lp_out, double* grad_out, int32_t n_threads);
#ifdef __cplusplus
}
#endif
#endif
