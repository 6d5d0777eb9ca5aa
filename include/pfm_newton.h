/* pfm_newton.h — C ABI of the other per-iteration sweeps of the reference's Newton / active-set loop
 * (SURVEY.md §8(f) N2, N3).  They share the context (mesh tables, node state, constraint flags) of
 * pfm_assemble.h, so that between two assemblies residual_total, diag_mass and the solution stay on the
 * device.  Same conventions: plain pointers and sizes, int status (pfm_status), pfm_last_error() for text.
 *
 * Reference interfaces replaced (all private members of FracturePhaseFieldProblem<dim>, cracks.cc):
 *   assemble_diag_mass_matrix()                       cracks.cc:2514-2562   -> pfm_diag_mass_device
 *   active-set block of newton_active_set()           cracks.cc:2826-2909   -> pfm_active_set_device
 *   compute_energy(), compute_tcv()                   cracks.cc:3615-3701, 3553-3611 -> pfm_functionals
 *   constraints_update.set_zero(residual); residual.l2_norm() / linfty_norm()
 *                                                     cracks.cc:2791-2794, 2918-2919, 2947-2949 -> pfm_residual_norms
 */
#ifndef PFM_NEWTON_H
#define PFM_NEWTON_H

#include <stdint.h>

#include "pfm_assemble.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diag_mass (cracks.cc:2514-2562): lumped (QGaussLobatto(2)) phase-field mass of every owned node,
 * d_mass[n_owned_nodes] (device pointer), asynchronous on the context's stream.  Displacement dofs have no
 * entry (the reference leaves them 0 and never reads them). */
int pfm_diag_mass_device(pfm_ctx *ctx, double *d_mass);

/* Active-set update (cracks.cc:2837-2886) + cycle counter (cracks.cc:2903-2909) + re-distribution of the
 * hanging nodes (cracks.cc:2888-2890), for the owned nodes of this rank:
 *   a phase-field dof that is not hanging becomes ACTIVE unless
 *        residual_total/diag_mass + c (phi - phi_old) <= 0  and  cycle_counter < 5;
 *   an active dof gets phi := phi_old and a homogeneous constraint line (bit `dim` of the node's flag byte in
 *   the context, i.e. exactly what pfm_set_constraints would have been given); a dof that leaves the set
 *   increments its cycle counter.
 * d_residual_total, d_solution, d_old_solution: device vectors over the owned dofs in the context's layout
 * (d_solution is modified); d_mass from pfm_diag_mass_device; d_cycle_counter[n_owned_nodes] int32, zeroed by
 * the caller at the start of newton_active_set (cracks.cc:2811).
 * counts[0] = owned active dofs, counts[1] = cycling dofs among them, counts[2] = 1 if the set changed.
 * Synchronous (the counts are returned on the host, as the reference prints them).  On a partitioned mesh the
 * flags of ghost nodes are the caller's to exchange (pfm_get_constraints / pfm_set_constraints). */
int pfm_active_set_device(pfm_ctx *ctx, const double *d_residual_total, const double *d_mass, double c,
                          double *d_solution, const double *d_old_solution, int32_t *d_cycle_counter,
                          int64_t counts[3]);

/* What the line search reads after every assemble_nl_residual() (cracks.cc:2946-2949; 10-50 times per Newton step) and the
 * Newton loop after every assembly (cracks.cc:2791-2794, 2918-2919): the norms of a residual vector with the constrained
 * lines zeroed -- constraints_update.set_zero(system_pde_residual); system_pde_residual.l2_norm() -- without the vector
 * ever leaving the device (24 bytes instead of 2 x 327 MB at 216^3).
 *   d_residual: device vector over the owned dofs in the context's layout (residual_pde of pfm_assemble_device /
 *   pfm_assemble_nl_residual_device, or residual_total); zeroed lines = the flag bits last given to pfm_set_constraints
 *   (Dirichlet lines, active set) and every component of a hanging node.
 *   out[0] = l2 norm, out[1] = l-infinity norm, out[2] = sum of squares -- of THIS rank's owned dofs; a partitioned caller
 *   adds out[2] over the ranks and takes the root (Utilities::MPI::sum inside l2_norm), and the maximum of out[1].
 * Deterministic two-stage reduction (the grid depends on n_owned only); ordered behind the assembly on the context's
 * stream; synchronous, out is a host pointer. */
int pfm_residual_norms(pfm_ctx *ctx, const double *d_residual, double out[3]);

/* read back the flag byte of every local node (bit c: dof (node,c) has a homogeneous constraint line) */
int pfm_get_constraints(pfm_ctx *ctx, uint8_t *node_flags);

/* compute_energy + compute_tcv on the node state last given to pfm_state_set (after the ghost import):
 *   out[0] = bulk energy   int ((1+k) pf^2 + k) psi(E)                          cracks.cc:3677
 *   out[1] = crack energy  G_c/2 int ((pf-1)^2/eps + eps |grad pf|^2)            cracks.cc:3679-3680
 *   out[2] = TCV           int u . grad pf                                       cracks.cc:3587
 * over the cells with cell_owned[cell] != 0 (host array [n_cells]; NULL = every local cell), this rank's
 * part of the sums (the caller adds the ranks, Utilities::MPI::sum, cracks.cc:3590, 3685-3686).
 * Deterministic two-stage reduction; synchronous, out is a host pointer. */
int pfm_functionals(pfm_ctx *ctx, const uint8_t *cell_owned, double out[3]);
/* The same with the Lame coefficients of the energy given per cell (host arrays [n_cells]; both NULL = the context's,
 * i.e. pfm_functionals).  Needed for the reference's heterogeneous test case: assemble_system adds 1.0 to the
 * Young's modulus read from the bitmap (cracks.cc:2209-2210) but compute_energy does not (cracks.cc:3649-3657), so
 * the energy is NOT evaluated with the coefficients the assembly uses; a caller that wants the reference's
 * statistics passes the un-shifted ones here. */
int pfm_functionals_material(pfm_ctx *ctx, const uint8_t *cell_owned, const double *cell_lambda, const double *cell_mu,
                             double out[3]);

#ifdef __cplusplus
}
#endif
#endif
