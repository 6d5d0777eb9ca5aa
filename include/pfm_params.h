/* pfm_params.h — resolved scalar inputs of the assembly hot path.
 *
 * Shared by the product library (include/pfm_assemble.h) and by the CPU oracle
 * (oracle/oracle.cpp).  Every field is a member of the reference's
 * FracturePhaseFieldProblem<dim> that assemble_system() reads
 * (/root/reference cracks.cc:2129-2498); the harness / glue code resolves them
 * exactly where the reference does and passes the plain numbers.
 */
#ifndef PFM_PARAMS_H
#define PFM_PARAMS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfm_params
{
  double lambda;                  /* lame_coefficient_lambda  cracks.cc:1509,1516 (per-cell override: 2207-2216) */
  double mu;                      /* lame_coefficient_mu      cracks.cc:1507,1515 */
  double G_c;                     /* cracks.cc:1493 */
  double alpha_eps;               /* eps, cracks.cc:3881 */
  double constant_k;              /* kappa, cracks.cc:3879 */
  double pressure;                /* current_pressure = func_pressure(time), cracks.cc:2145 */
  double alpha_biot;              /* cracks.cc:1497 (hard-wired 0 in the reference) */
  double gamma_penal;             /* cracks.cc:1484-1487; forced to 0 for this call when
                                     outer_solver==simple_monolithic && timestep_number<1 (2141-2144) */
  double timestep;                /* cracks.cc:4297-4299 */
  double time;
  double old_timestep;
  double old_old_timestep;
  double decompose_stress_rhs;    /* cracks.cc:1568 */
  double decompose_stress_matrix; /* cracks.cc:1569 */
  int timestep_number;            /* gates the stress split, cracks.cc:2294,2338 */
  int outer_solver;               /* 0 = active_set, 1 = simple_monolithic (cracks.cc:1425-1428) */
  int use_old_timestep_pf;        /* cracks.cc:2276-2277 */
  int reserved;
} pfm_params;

enum
{
  PFM_SOLVER_ACTIVE_SET = 0,
  PFM_SOLVER_SIMPLE_MONOLITHIC = 1
};

#ifdef __cplusplus
}
#endif
#endif
