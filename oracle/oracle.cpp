// oracle.cpp — CPU ORACLE for the assembly hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is a loop-for-loop C++ restatement of the reference's
//   FracturePhaseFieldProblem<dim>::assemble_system(bool residual_only)
//   (/root/reference cracks.cc:2129-2475)
// together with the helpers it calls
//   eigen_vectors_and_values   cracks.cc:1691-1737
//   decompose_stress           cracks.cc:1923-2120
//   Tensors::get_divergence_u  cracks.cc:331-347
//   Tensors::get_Identity      cracks.cc:290-301
// and of the Newton-side sweeps that share its data
//   assemble_diag_mass_matrix  cracks.cc:2514-2562
//   active-set update          cracks.cc:2837-2886, 2903-2909
//   compute_energy, compute_tcv cracks.cc:3615-3701, 3553-3611
// and a textbook restatement of the deal.II pieces the reference leans on and
// that are NOT under /root/reference (deal.II >= 9.5, CMakeLists.txt:14):
//   FE_Q<dim>(1) x (dim+1) FESystem, local dof i <-> (vertex i/(dim+1), comp i%(dim+1))
//   QGauss<dim>(3) (cracks.cc:2156, fe.degree+2), x fastest
//   MappingQ1 FEValues: JxW, J^{-T} grad N        (cracks.cc:2158-2160, 2203)
//   get_function_values / get_function_gradients  (cracks.cc:2222-2232)
//   cell->diameter()                               (cracks.cc:2370, 2419)
//   AffineConstraints::distribute_local_to_global  (cracks.cc:2442-2463)
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library, and only as the checker / the reported CPU baseline.  The
// product path (cracks_amd/) never links, imports or calls anything in oracle/.
//
// Parity pin: the oracle is checked against the reference's golden Newton
// tables (tests/golden/kat.json) to all 7 printed digits and
// against the six Catch eigen-decomposition cases (cracks.cc:1740-1919); the energy functionals are
// pinned by the bulk/crack energies of the reference's *.statistics files (tests/test_newton_goldens.py,
// tests/test_newton_sweeps.py).  compute_tcv has no golden in the reference's tests: parity unpinned for TCV.
// The reference binary itself cannot be built here (deal.II, Trilinos and p4est
// are absent), so there is no oracle/_ref.
//
// Build: see oracle/Makefile  (g++ -O3 -march=native -std=c++17 -shared -fPIC)

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../include/pfm_params.h"

namespace
{
  enum
  {
    ORACLE_OK = 0,
    ORACLE_BAD_ARG = 1,
    ORACLE_NOT_ORTHOGONAL = 3, // reference abort(), cracks.cc:1732-1736
    ORACLE_PATTERN_MISS = 7    // a non-zero contribution has no slot in the CSR pattern
  };

  // ---------------------------------------------------------------- tensors
  template <int dim>
  struct T2
  {
    double v[dim][dim];
    T2() { clear(); }
    void clear()
    {
      for (int i = 0; i < dim; ++i)
        for (int j = 0; j < dim; ++j)
          v[i][j] = 0.0;
    }
    double *operator[](int i) { return v[i]; }
    const double *operator[](int i) const { return v[i]; }
  };

  template <int dim>
  struct T1
  {
    double v[dim];
    T1()
    {
      for (int i = 0; i < dim; ++i)
        v[i] = 0.0;
    }
    double &operator[](int i) { return v[i]; }
    const double &operator[](int i) const { return v[i]; }
  };

  template <int dim>
  T2<dim> operator+(const T2<dim> &a, const T2<dim> &b)
  {
    T2<dim> r;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j)
        r[i][j] = a[i][j] + b[i][j];
    return r;
  }
  template <int dim>
  T2<dim> operator-(const T2<dim> &a, const T2<dim> &b)
  {
    T2<dim> r;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j)
        r[i][j] = a[i][j] - b[i][j];
    return r;
  }
  template <int dim>
  T2<dim> operator*(const double s, const T2<dim> &a)
  {
    T2<dim> r;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j)
        r[i][j] = s * a[i][j];
    return r;
  }
  // Tensor<2,dim> * Tensor<2,dim> = contraction over the inner index
  template <int dim>
  T2<dim> operator*(const T2<dim> &a, const T2<dim> &b)
  {
    T2<dim> r;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j)
        {
          double s = 0.0;
          for (int k = 0; k < dim; ++k)
            s += a[i][k] * b[k][j];
          r[i][j] = s;
        }
    return r;
  }
  template <int dim>
  T2<dim> transpose(const T2<dim> &a)
  {
    T2<dim> r;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j)
        r[i][j] = a[j][i];
    return r;
  }
  template <int dim>
  double trace(const T2<dim> &a)
  {
    double t = 0.0;
    for (int i = 0; i < dim; ++i)
      t += a[i][i];
    return t;
  }
  template <int dim>
  double scalar_product(const T2<dim> &a, const T2<dim> &b)
  {
    double s = 0.0;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j)
        s += a[i][j] * b[i][j];
    return s;
  }
  template <int dim>
  double dot(const T1<dim> &a, const T1<dim> &b)
  {
    double s = 0.0;
    for (int i = 0; i < dim; ++i)
      s += a[i] * b[i];
    return s;
  }

  // Tensors::get_Identity, cracks.cc:290-301
  template <int dim>
  T2<dim> get_Identity()
  {
    T2<dim> id;
    for (int i = 0; i < dim; ++i)
      id[i][i] = 1.0;
    return id;
  }
  // Tensors::get_divergence_u, cracks.cc:331-347
  template <int dim>
  double get_divergence_u(const T2<dim> &grad_u)
  {
    double tmp = 0.0;
    for (int i = 0; i < dim; ++i)
      tmp += grad_u[i][i];
    return tmp;
  }

  // ------------------------------------------------ stress spectral split
  // eigen_vectors_and_values, cracks.cc:1691-1737.  Only entries [0..1][0..1]
  // are touched, exactly as in the reference (which is why it is 2-D only).
  template <int dim>
  int eigen_vectors_and_values(double &E_eigenvalue_1, double &E_eigenvalue_2,
                               T2<dim> &ev_matrix, const T2<dim> &matrix)
  {
    double E_eigenvector_1[2] = {0, 0};
    double E_eigenvector_2[2] = {0, 0};
    if (std::abs(matrix[0][1]) < 1e-10 * std::abs(matrix[0][0]) ||
        std::abs(matrix[0][1]) < 1e-10 * std::abs(matrix[1][1]))
      {
        // E is close to diagonal
        E_eigenvalue_1 = matrix[0][0];
        E_eigenvector_1[0] = 1;
        E_eigenvector_1[1] = 0;
        E_eigenvalue_2 = matrix[1][1];
        E_eigenvector_2[0] = 0;
        E_eigenvector_2[1] = 1;
      }
    else
      {
        double sq = std::sqrt((matrix[0][0] - matrix[1][1]) * (matrix[0][0] - matrix[1][1]) +
                              4.0 * matrix[0][1] * matrix[1][0]);
        E_eigenvalue_1 = 0.5 * ((matrix[0][0] + matrix[1][1]) + sq);
        E_eigenvalue_2 = 0.5 * ((matrix[0][0] + matrix[1][1]) - sq);

        E_eigenvector_1[0] =
          1.0 / (std::sqrt(1 + (E_eigenvalue_1 - matrix[0][0]) / matrix[0][1] *
                                 (E_eigenvalue_1 - matrix[0][0]) / matrix[0][1]));
        E_eigenvector_1[1] =
          (E_eigenvalue_1 - matrix[0][0]) /
          (matrix[0][1] * (std::sqrt(1 + (E_eigenvalue_1 - matrix[0][0]) / matrix[0][1] *
                                           (E_eigenvalue_1 - matrix[0][0]) / matrix[0][1])));
        E_eigenvector_2[0] =
          1.0 / (std::sqrt(1 + (E_eigenvalue_2 - matrix[0][0]) / matrix[0][1] *
                                 (E_eigenvalue_2 - matrix[0][0]) / matrix[0][1]));
        E_eigenvector_2[1] =
          (E_eigenvalue_2 - matrix[0][0]) /
          (matrix[0][1] * (std::sqrt(1 + (E_eigenvalue_2 - matrix[0][0]) / matrix[0][1] *
                                           (E_eigenvalue_2 - matrix[0][0]) / matrix[0][1])));
      }

    ev_matrix[0][0] = E_eigenvector_1[0];
    ev_matrix[0][1] = E_eigenvector_2[0];
    ev_matrix[1][0] = E_eigenvector_1[1];
    ev_matrix[1][1] = E_eigenvector_2[1];

    // Sanity check if orthogonal (reference: abort(); here: error code)
    double scalar_prod =
      E_eigenvector_1[0] * E_eigenvector_2[0] + E_eigenvector_1[1] * E_eigenvector_2[1];
    if (scalar_prod > 1.0e-6)
      return ORACLE_NOT_ORTHOGONAL;
    return ORACLE_OK;
  }

  // decompose_stress, cracks.cc:1923-2120
  template <int dim>
  int decompose_stress(T2<dim> &stress_term_plus, T2<dim> &stress_term_minus, const T2<dim> &E,
                       const double tr_E, const T2<dim> &E_LinU, const double tr_E_LinU,
                       const double lame_coefficient_lambda, const double lame_coefficient_mu,
                       const bool derivative)
  {
    const T2<dim> Identity = get_Identity<dim>();

    double E_eigenvalue_1, E_eigenvalue_2;
    T2<dim> P_matrix;
    int err = eigen_vectors_and_values(E_eigenvalue_1, E_eigenvalue_2, P_matrix, E);
    if (err)
      return err;

    double E_eigenvalue_1_plus = std::max(0.0, E_eigenvalue_1);
    double E_eigenvalue_2_plus = std::max(0.0, E_eigenvalue_2);

    T2<dim> Lambda_plus;
    Lambda_plus[0][0] = E_eigenvalue_1_plus;
    Lambda_plus[0][1] = 0.0;
    Lambda_plus[1][0] = 0.0;
    Lambda_plus[1][1] = E_eigenvalue_2_plus;

    if (!derivative)
      {
        T2<dim> E_plus = P_matrix * Lambda_plus * transpose(P_matrix);

        double tr_E_positive = std::max(0.0, tr_E);

        stress_term_plus =
          (lame_coefficient_lambda * tr_E_positive) * Identity + (2 * lame_coefficient_mu) * E_plus;

        stress_term_minus = (lame_coefficient_lambda * (tr_E - tr_E_positive)) * Identity +
                            (2 * lame_coefficient_mu) * (E - E_plus);
      }
    else
      {
        double E_eigenvalue_1_LinU, E_eigenvalue_2_LinU;
        double E_eigenvector_1_LinU[2];
        double E_eigenvector_2_LinU[2];
        T2<dim> P_matrix_LinU;

        // linearized eigenvalues
        double diskriminante =
          std::sqrt(E[0][1] * E[1][0] + (E[0][0] - E[1][1]) * (E[0][0] - E[1][1]) / 4.0);

        E_eigenvalue_1_LinU =
          0.5 * tr_E_LinU + 1.0 / (2.0 * diskriminante) *
                              (E_LinU[0][1] * E[1][0] + E[0][1] * E_LinU[1][0] +
                               (E[0][0] - E[1][1]) * (E_LinU[0][0] - E_LinU[1][1]) / 2.0);

        E_eigenvalue_2_LinU =
          0.5 * tr_E_LinU - 1.0 / (2.0 * diskriminante) *
                              (E_LinU[0][1] * E[1][0] + E[0][1] * E_LinU[1][0] +
                               (E[0][0] - E[1][1]) * (E_LinU[0][0] - E_LinU[1][1]) / 2.0);

        // normalized eigenvectors and P
        double normalization_1 =
          1.0 / (std::sqrt(1 + (E_eigenvalue_1 - E[0][0]) / E[0][1] * (E_eigenvalue_1 - E[0][0]) /
                                 E[0][1]));
        double normalization_2 =
          1.0 / (std::sqrt(1 + (E_eigenvalue_2 - E[0][0]) / E[0][1] * (E_eigenvalue_2 - E[0][0]) /
                                 E[0][1]));

        double normalization_1_LinU = 0.0;
        double normalization_2_LinU = 0.0;

        normalization_1_LinU =
          -1.0 *
          (1.0 / (1.0 + (E_eigenvalue_1 - E[0][0]) / E[0][1] * (E_eigenvalue_1 - E[0][0]) / E[0][1]) *
           1.0 /
           (2.0 * std::sqrt(1.0 + (E_eigenvalue_1 - E[0][0]) / E[0][1] * (E_eigenvalue_1 - E[0][0]) /
                                    E[0][1])) *
           (2.0 * (E_eigenvalue_1 - E[0][0]) / E[0][1]) *
           ((E_eigenvalue_1_LinU - E_LinU[0][0]) * E[0][1] -
            (E_eigenvalue_1 - E[0][0]) * E_LinU[0][1]) /
           (E[0][1] * E[0][1]));

        normalization_2_LinU =
          -1.0 *
          (1.0 / (1.0 + (E_eigenvalue_2 - E[0][0]) / E[0][1] * (E_eigenvalue_2 - E[0][0]) / E[0][1]) *
           1.0 /
           (2.0 * std::sqrt(1.0 + (E_eigenvalue_2 - E[0][0]) / E[0][1] * (E_eigenvalue_2 - E[0][0]) /
                                    E[0][1])) *
           (2.0 * (E_eigenvalue_2 - E[0][0]) / E[0][1]) *
           ((E_eigenvalue_2_LinU - E_LinU[0][0]) * E[0][1] -
            (E_eigenvalue_2 - E[0][0]) * E_LinU[0][1]) /
           (E[0][1] * E[0][1]));

        E_eigenvector_1_LinU[0] = normalization_1 * 1.0;
        E_eigenvector_1_LinU[1] = normalization_1 * (E_eigenvalue_1 - E[0][0]) / E[0][1];

        E_eigenvector_2_LinU[0] = normalization_2 * 1.0;
        E_eigenvector_2_LinU[1] = normalization_2 * (E_eigenvalue_2 - E[0][0]) / E[0][1];

        // product rule on normalization and vector entries
        double EV_1_part_1_comp_1 = normalization_1 * 0.0;
        double EV_1_part_1_comp_2 =
          normalization_1 *
          ((E_eigenvalue_1_LinU - E_LinU[0][0]) * E[0][1] -
           (E_eigenvalue_1 - E[0][0]) * E_LinU[0][1]) /
          (E[0][1] * E[0][1]);

        double EV_1_part_2_comp_1 = normalization_1_LinU * 1.0;
        double EV_1_part_2_comp_2 = normalization_1_LinU * (E_eigenvalue_1 - E[0][0]) / E[0][1];

        double EV_2_part_1_comp_1 = normalization_2 * 0.0;
        double EV_2_part_1_comp_2 =
          normalization_2 *
          ((E_eigenvalue_2_LinU - E_LinU[0][0]) * E[0][1] -
           (E_eigenvalue_2 - E[0][0]) * E_LinU[0][1]) /
          (E[0][1] * E[0][1]);

        double EV_2_part_2_comp_1 = normalization_2_LinU * 1.0;
        double EV_2_part_2_comp_2 = normalization_2_LinU * (E_eigenvalue_2 - E[0][0]) / E[0][1];

        E_eigenvector_1_LinU[0] = EV_1_part_1_comp_1 + EV_1_part_2_comp_1;
        E_eigenvector_1_LinU[1] = EV_1_part_1_comp_2 + EV_1_part_2_comp_2;

        E_eigenvector_2_LinU[0] = EV_2_part_1_comp_1 + EV_2_part_2_comp_1;
        E_eigenvector_2_LinU[1] = EV_2_part_1_comp_2 + EV_2_part_2_comp_2;

        P_matrix_LinU[0][0] = E_eigenvector_1_LinU[0];
        P_matrix_LinU[0][1] = E_eigenvector_2_LinU[0];
        P_matrix_LinU[1][0] = E_eigenvector_1_LinU[1];
        P_matrix_LinU[1][1] = E_eigenvector_2_LinU[1];

        double E_eigenvalue_1_plus_LinU = 0.0;
        double E_eigenvalue_2_plus_LinU = 0.0;

        // zero where the corresponding rhs value is zeroed (cracks.cc:2065-2081)
        if (E_eigenvalue_1 < 0.0)
          E_eigenvalue_1_plus_LinU = 0.0;
        else
          E_eigenvalue_1_plus_LinU = E_eigenvalue_1_LinU;

        if (E_eigenvalue_2 < 0.0)
          E_eigenvalue_2_plus_LinU = 0.0;
        else
          E_eigenvalue_2_plus_LinU = E_eigenvalue_2_LinU;

        T2<dim> Lambda_plus_LinU;
        Lambda_plus_LinU[0][0] = E_eigenvalue_1_plus_LinU;
        Lambda_plus_LinU[0][1] = 0.0;
        Lambda_plus_LinU[1][0] = 0.0;
        Lambda_plus_LinU[1][1] = E_eigenvalue_2_plus_LinU;

        T2<dim> E_plus_LinU = P_matrix_LinU * Lambda_plus * transpose(P_matrix) +
                              P_matrix * Lambda_plus_LinU * transpose(P_matrix) +
                              P_matrix * Lambda_plus * transpose(P_matrix_LinU);

        double tr_E_positive_LinU = 0.0;
        if (tr_E < 0.0)
          tr_E_positive_LinU = 0.0;
        else
          tr_E_positive_LinU = tr_E_LinU;

        stress_term_plus = (lame_coefficient_lambda * tr_E_positive_LinU) * Identity +
                           (2 * lame_coefficient_mu) * E_plus_LinU;

        stress_term_minus = (lame_coefficient_lambda * (tr_E_LinU - tr_E_positive_LinU)) * Identity +
                            (2 * lame_coefficient_mu) * (E_LinU - E_plus_LinU);
      }
    return ORACLE_OK;
  }

  // ------------------------------------------------------ FEValues (deal.II)
  // QGauss(3) on [0,1]
  const double gauss_x[3] = {0.5 - 0.5 * 0.7745966692414834, 0.5, 0.5 + 0.5 * 0.7745966692414834};
  const double gauss_w[3] = {5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0};

  template <int dim>
  struct FEValues
  {
    static constexpr int nv = 1 << dim;
    static constexpr int nq = (dim == 2 ? 9 : 27);
    // reference data
    double Nref[nq][nv];
    double dNref[nq][nv][dim];
    double wref[nq];
    // per-cell data
    double N[nq][nv]; // == Nref
    double dN[nq][nv][dim];
    double JxW[nq];
    double diameter;

    FEValues()
    {
      for (int q = 0; q < nq; ++q)
        {
          int qi[3] = {q % 3, (q / 3) % 3, q / 9};
          double w = 1.0;
          for (int d = 0; d < dim; ++d)
            w *= gauss_w[qi[d]];
          wref[q] = w;
          for (int v = 0; v < nv; ++v)
            {
              double val = 1.0;
              for (int d = 0; d < dim; ++d)
                {
                  const double x = gauss_x[qi[d]];
                  val *= ((v >> d) & 1) ? x : (1.0 - x);
                }
              Nref[q][v] = val;
              N[q][v] = val;
              for (int e = 0; e < dim; ++e)
                {
                  double g = 1.0;
                  for (int d = 0; d < dim; ++d)
                    {
                      const double x = gauss_x[qi[d]];
                      if (d == e)
                        g *= ((v >> d) & 1) ? 1.0 : -1.0;
                      else
                        g *= ((v >> d) & 1) ? x : (1.0 - x);
                    }
                  dNref[q][v][e] = g;
                }
            }
        }
    }

    // fe_values.reinit(cell), cracks.cc:2203 (MappingQ1)
    void reinit(const double (*xv)[dim])
    {
      for (int q = 0; q < nq; ++q)
        {
          double J[dim][dim];
          for (int i = 0; i < dim; ++i)
            for (int j = 0; j < dim; ++j)
              {
                double s = 0.0;
                for (int v = 0; v < nv; ++v)
                  s += xv[v][i] * dNref[q][v][j];
                J[i][j] = s;
              }
          double inv[dim][dim];
          double det;
          if constexpr (dim == 2)
            {
              det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
              const double id = 1.0 / det;
              inv[0][0] = J[1][1] * id;
              inv[0][1] = -J[0][1] * id;
              inv[1][0] = -J[1][0] * id;
              inv[1][1] = J[0][0] * id;
            }
          else
            {
              const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
              const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
              const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
              det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
              const double id = 1.0 / det;
              inv[0][0] = c00 * id;
              inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
              inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
              inv[1][0] = c01 * id;
              inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
              inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
              inv[2][0] = c02 * id;
              inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
              inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
            }
          JxW[q] = det * wref[q];
          for (int v = 0; v < nv; ++v)
            for (int d = 0; d < dim; ++d)
              {
                double s = 0.0;
                for (int e = 0; e < dim; ++e)
                  s += inv[e][d] * dNref[q][v][e];
                dN[q][v][d] = s;
              }
        }
      // cell->diameter(): longest vertex-to-opposite-vertex diagonal
      double dmax = 0.0;
      for (int v = 0; v < nv / 2; ++v)
        {
          const int o = (nv - 1) - v;
          double s = 0.0;
          for (int d = 0; d < dim; ++d)
            s += (xv[v][d] - xv[o][d]) * (xv[v][d] - xv[o][d]);
          dmax = std::max(dmax, std::sqrt(s));
        }
      diameter = dmax;
    }
  };

  // --------------------------------------------------- constraints (deal.II)
  struct Constraints
  {
    const uint8_t *flag; // is_constrained(dof)
    const int64_t *ptr;  // entries of constrained dof d: [ptr[d], ptr[d+1])
    const int32_t *col;
    const double *w;
    bool is_constrained(int32_t d) const { return flag && flag[d]; }
  };

  struct Csr
  {
    const int64_t *rowptr;
    const int32_t *colind;
    double *values;
    // returns false when (r,c) is not in the pattern
    bool add(int32_t r, int32_t c, double v)
    {
      const int32_t *b = colind + rowptr[r];
      const int32_t *e = colind + rowptr[r + 1];
      const int32_t *p = std::lower_bound(b, e, c);
      if (p == e || *p != c)
        return false;
      values[p - colind] += v;
      return true;
    }
  };

  // AffineConstraints::distribute_local_to_global(local_vector, indices, global_vector)
  // (deal.II): unconstrained rows are added; a constrained row's value is
  // redistributed to its constraint entries with their weights; inhomogeneities
  // are not used by this overload.
  void distribute_vector(const Constraints &c, const double *local, const int32_t *dofs, int n,
                         double *global)
  {
    for (int i = 0; i < n; ++i)
      {
        const int32_t g = dofs[i];
        if (!c.is_constrained(g))
          global[g] += local[i];
        else
          for (int64_t k = c.ptr[g]; k < c.ptr[g + 1]; ++k)
            global[c.col[k]] += local[i] * c.w[k];
      }
  }

  // AffineConstraints::distribute_local_to_global(local_matrix, local_vector,
  // indices, global_matrix, global_vector) (deal.II), all inhomogeneities zero
  // (Newton-update constraints, cracks.cc:2711-2714, 2878-2879):
  //   global(R,C) += w_R * w_C * local(i,j) over the constraint expansion of i, j
  //   global_vector(R) += w_R * local_vector(i)
  //   constrained rows get a diagonal entry |local(i,i)| (or the mean |diag| of the
  //   local matrix when that is zero) so the matrix stays regular.
  // deal.II-knowledge (affine_constraints.templates.h); not pinned by any
  // reference golden (SURVEY.md §7 "Constraint semantics").
  int distribute_matrix(const Constraints &c, const double *lm, const double *lv,
                        const int32_t *dofs, int n, Csr &A, double *gv)
  {
    bool any_constrained = false;
    for (int i = 0; i < n; ++i)
      {
        const int32_t gi = dofs[i];
        const bool ci = c.is_constrained(gi);
        any_constrained |= ci;
        const int64_t ib = ci ? c.ptr[gi] : 0, ie = ci ? c.ptr[gi + 1] : 1;
        for (int64_t ki = ib; ki < ie; ++ki)
          {
            const int32_t R = ci ? c.col[ki] : gi;
            const double wR = ci ? c.w[ki] : 1.0;
            gv[R] += wR * lv[i];
            for (int j = 0; j < n; ++j)
              {
                const int32_t gj = dofs[j];
                const bool cj = c.is_constrained(gj);
                const int64_t jb = cj ? c.ptr[gj] : 0, je = cj ? c.ptr[gj + 1] : 1;
                for (int64_t kj = jb; kj < je; ++kj)
                  {
                    const int32_t C = cj ? c.col[kj] : gj;
                    const double wC = cj ? c.w[kj] : 1.0;
                    const double val = wR * wC * lm[i * n + j];
                    if (!A.add(R, C, val) && val != 0.0)
                      return ORACLE_PATTERN_MISS;
                  }
              }
          }
      }
    if (any_constrained)
      {
        double average_diagonal = 0.0;
        for (int i = 0; i < n; ++i)
          average_diagonal += std::abs(lm[i * n + i]);
        average_diagonal /= static_cast<double>(n);
        for (int i = 0; i < n; ++i)
          if (c.is_constrained(dofs[i]))
            {
              const double d = std::abs(lm[i * n + i]);
              const double new_diagonal = (d != 0.0 ? d : average_diagonal);
              if (!A.add(dofs[i], dofs[i], new_diagonal) && new_diagonal != 0.0)
                return ORACLE_PATTERN_MISS;
            }
      }
    return ORACLE_OK;
  }

  // ---------------------------------------------------------- cell kernel
  // local_matrix / local_rhs of one cell: cracks.cc:2218-2437
  template <int dim>
  int cell_local(FEValues<dim> &fe_values, const pfm_params &P, double lame_coefficient_lambda,
                 double lame_coefficient_mu, const double *U /*dpc*/, const double *Uold_pf /*nv*/,
                 const double *Uoldold_pf /*nv*/, bool residual_only, double *local_matrix /*dpc*dpc*/,
                 double *local_rhs /*dpc*/)
  {
    constexpr int nv = 1 << dim;
    constexpr int nc = dim + 1;
    constexpr int dofs_per_cell = nv * nc;
    constexpr int n_q_points = FEValues<dim>::nq;

    double gamma_penal = P.gamma_penal;
    if (P.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC && P.timestep_number < 1)
      gamma_penal = 0.0; // cracks.cc:2141-2144
    const double current_pressure = P.pressure;
    const double constant_k = P.constant_k;
    const double alpha_eps = P.alpha_eps;
    const double G_c = P.G_c;
    const double alpha_biot = P.alpha_biot;
    const double timestep = P.timestep, time = P.time, old_timestep = P.old_timestep,
                 old_old_timestep = P.old_old_timestep;
    const double decompose_stress_rhs = P.decompose_stress_rhs;
    const double decompose_stress_matrix = P.decompose_stress_matrix;
    const int timestep_number = P.timestep_number;
    const double cell_diameter = fe_values.diameter;

    for (int i = 0; i < dofs_per_cell * dofs_per_cell; ++i)
      local_matrix[i] = 0.0;
    for (int i = 0; i < dofs_per_cell; ++i)
      local_rhs[i] = 0.0;

    T2<dim> zero_matrix;

    // test functions
    T2<dim> phi_i_grads_u[dofs_per_cell];
    double phi_i_pf[dofs_per_cell];
    T1<dim> phi_i_grads_pf[dofs_per_cell];

    for (int q = 0; q < n_q_points; ++q)
      {
        // get_function_values / get_function_gradients, cracks.cc:2222-2232
        T2<dim> old_displacement_grads;
        T1<dim> old_phase_field_grads;
        double old_phase_field_values = 0.0;
        double old_timestep_phase_field_values = 0.0;
        double old_old_timestep_phase_field_values = 0.0;
        for (int k = 0; k < dofs_per_cell; ++k)
          {
            const int v = k / nc, comp = k % nc;
            if (comp < dim)
              for (int d = 0; d < dim; ++d)
                old_displacement_grads[comp][d] += U[k] * fe_values.dN[q][v][d];
            else
              {
                old_phase_field_values += U[k] * fe_values.N[q][v];
                for (int d = 0; d < dim; ++d)
                  old_phase_field_grads[d] += U[k] * fe_values.dN[q][v][d];
                old_timestep_phase_field_values += Uold_pf[v] * fe_values.N[q][v];
                old_old_timestep_phase_field_values += Uoldold_pf[v] * fe_values.N[q][v];
              }
          }

        // cracks.cc:2237-2244
        for (int k = 0; k < dofs_per_cell; ++k)
          {
            const int v = k / nc, comp = k % nc;
            phi_i_grads_u[k].clear();
            phi_i_pf[k] = 0.0;
            phi_i_grads_pf[k] = T1<dim>();
            if (comp < dim)
              for (int d = 0; d < dim; ++d)
                phi_i_grads_u[k][comp][d] = fe_values.dN[q][v][d];
            else
              {
                phi_i_pf[k] = fe_values.N[q][v];
                for (int d = 0; d < dim; ++d)
                  phi_i_grads_pf[k][d] = fe_values.dN[q][v][d];
              }
          }

        // cracks.cc:2248-2277
        double pf = old_phase_field_values;
        double old_timestep_pf = old_timestep_phase_field_values;
        double old_old_timestep_pf = old_old_timestep_phase_field_values;
        if (P.outer_solver == PFM_SOLVER_SIMPLE_MONOLITHIC)
          {
            pf = std::max(0.0, old_phase_field_values);
            old_timestep_pf = std::max(0.0, old_timestep_phase_field_values);
            old_old_timestep_pf = std::max(0.0, old_old_timestep_phase_field_values);
          }

        double pf_minus_old_timestep_pf_plus = std::max(0.0, pf - old_timestep_pf);

        double pf_extra = pf;
        pf_extra = old_old_timestep_pf +
                   (time - (time - old_timestep - old_old_timestep)) /
                     (time - old_timestep - (time - old_timestep - old_old_timestep)) *
                     (old_timestep_pf - old_old_timestep_pf);
        if (pf_extra <= 0.0)
          pf_extra = 0.0;
        if (pf_extra >= 1.0)
          pf_extra = 1.0;

        if (P.use_old_timestep_pf)
          pf_extra = old_timestep_pf;

        // cracks.cc:2280-2306
        const T2<dim> grad_u = old_displacement_grads;
        const T1<dim> grad_pf = old_phase_field_grads;
        const double divergence_u = get_divergence_u<dim>(grad_u);
        const T2<dim> Identity = get_Identity<dim>();
        const T2<dim> E = 0.5 * (grad_u + transpose(grad_u));
        const double tr_E = trace(E);

        T2<dim> stress_term_plus;
        T2<dim> stress_term_minus;
        if (decompose_stress_matrix > 0 && timestep_number > 0)
          {
            int err = decompose_stress(stress_term_plus, stress_term_minus, E, tr_E, zero_matrix,
                                       0.0, lame_coefficient_lambda, lame_coefficient_mu, false);
            if (err)
              return err;
          }
        else
          {
            stress_term_plus =
              (lame_coefficient_lambda * tr_E) * Identity + (2 * lame_coefficient_mu) * E;
            stress_term_minus.clear();
          }

        const double JxW = fe_values.JxW[q];

        // cracks.cc:2308-2389
        if (!residual_only)
          for (int i = 0; i < dofs_per_cell; ++i)
            {
              double pf_minus_old_timestep_pf_plus = 0.0; // shadows (cracks.cc:2311)
              if ((pf - old_timestep_pf) < 0.0)
                pf_minus_old_timestep_pf_plus = 0.0;
              else
                pf_minus_old_timestep_pf_plus = phi_i_pf[i];

              const T2<dim> E_LinU = 0.5 * (phi_i_grads_u[i] + transpose(phi_i_grads_u[i]));
              const double tr_E_LinU = trace(E_LinU);

              const double divergence_u_LinU = get_divergence_u<dim>(phi_i_grads_u[i]);

              T2<dim> stress_term_plus_LinU;
              T2<dim> stress_term_minus_LinU;

              const int comp_i = i % nc;
              if (comp_i == dim)
                {
                  stress_term_plus_LinU.clear();
                  stress_term_minus_LinU.clear();
                }
              else if (decompose_stress_matrix > 0.0 && timestep_number > 0)
                {
                  int err = decompose_stress(stress_term_plus_LinU, stress_term_minus_LinU, E, tr_E,
                                             E_LinU, tr_E_LinU, lame_coefficient_lambda,
                                             lame_coefficient_mu, true);
                  if (err)
                    return err;
                }
              else
                {
                  stress_term_plus_LinU = (lame_coefficient_lambda * tr_E_LinU) * Identity +
                                          (2 * lame_coefficient_mu) * E_LinU;
                  stress_term_minus.clear(); // cracks.cc:2350 (side effect kept)
                }

              for (int j = 0; j < dofs_per_cell; ++j)
                {
                  const int comp_j = j % nc;
                  if (comp_j < dim)
                    {
                      // Solid
                      local_matrix[j * dofs_per_cell + i] +=
                        1.0 *
                        (scalar_product(((1 - constant_k) * pf_extra * pf_extra + constant_k) *
                                          stress_term_plus_LinU,
                                        phi_i_grads_u[j])
                         // stress term minus
                         + decompose_stress_matrix *
                             scalar_product(stress_term_minus_LinU, phi_i_grads_u[j])) *
                        JxW;
                    }
                  else
                    {
                      // Simple penalization for simple monolithic
                      local_matrix[j * dofs_per_cell + i] +=
                        gamma_penal / timestep * 1.0 / (cell_diameter * cell_diameter) *
                        pf_minus_old_timestep_pf_plus * phi_i_pf[j] * JxW;

                      // Phase-field
                      local_matrix[j * dofs_per_cell + i] +=
                        ((1 - constant_k) *
                           (scalar_product(stress_term_plus_LinU, E) +
                            scalar_product(stress_term_plus, E_LinU)) *
                           pf * phi_i_pf[j] +
                         (1 - constant_k) * scalar_product(stress_term_plus, E) * phi_i_pf[i] *
                           phi_i_pf[j] +
                         G_c / alpha_eps * phi_i_pf[i] * phi_i_pf[j] +
                         G_c * alpha_eps * dot(phi_i_grads_pf[i], phi_i_grads_pf[j])
                         // Pressure terms
                         - 2.0 * (alpha_biot - 1.0) * current_pressure *
                             (pf * divergence_u_LinU + phi_i_pf[i] * divergence_u) * phi_i_pf[j]) *
                        JxW;
                    }
                }
            }

        // RHS: cracks.cc:2393-2432
        for (int i = 0; i < dofs_per_cell; ++i)
          {
            const int comp_i = i % nc;
            if (comp_i < dim)
              {
                const T2<dim> &phi_i_grads_u_i = phi_i_grads_u[i];
                const double divergence_u_LinU = get_divergence_u<dim>(phi_i_grads_u_i);

                // Solid
                local_rhs[i] -=
                  (scalar_product(((1.0 - constant_k) * pf_extra * pf_extra + constant_k) *
                                    stress_term_plus,
                                  phi_i_grads_u_i) +
                   decompose_stress_rhs * scalar_product(stress_term_minus, phi_i_grads_u_i)
                   // Pressure terms
                   - (alpha_biot - 1.0) * current_pressure * pf_extra * pf_extra *
                       divergence_u_LinU) *
                  JxW;
              }
            else
              {
                const double phi_i_pf_i = phi_i_pf[i];
                const T1<dim> &phi_i_grads_pf_i = phi_i_grads_pf[i];

                // Simple penalization
                local_rhs[i] -= gamma_penal / timestep * 1.0 / (cell_diameter * cell_diameter) *
                                pf_minus_old_timestep_pf_plus * phi_i_pf_i * JxW;

                // Phase field
                local_rhs[i] -=
                  ((1.0 - constant_k) * scalar_product(stress_term_plus, E) * pf * phi_i_pf_i -
                   G_c / alpha_eps * (1.0 - pf) * phi_i_pf_i +
                   G_c * alpha_eps * dot(grad_pf, phi_i_grads_pf_i)
                   // Pressure terms
                   - 2.0 * (alpha_biot - 1.0) * current_pressure * pf * divergence_u * phi_i_pf_i) *
                  JxW;
              }
          }
      }
    return ORACLE_OK;
  }

  template <int dim>
  int assemble(int64_t n_cells, int32_t n_dofs, const int32_t *cell_nodes, const double *coords,
               const int32_t *cell_dofs, const double *cell_lambda, const double *cell_mu,
               const pfm_params &P, const double *sol, const double *old, const double *oldold,
               const Constraints &cu, const Constraints &ch, int residual_only, Csr A,
               double *residual_pde, double *residual_total, int64_t cell_begin, int64_t cell_end,
               int zero_outputs)
  {
    constexpr int nv = 1 << dim;
    constexpr int nc = dim + 1;
    constexpr int dpc = nv * nc;

    // cracks.cc:2133-2137
    if (zero_outputs)
      {
        if (residual_only)
          std::fill(residual_total, residual_total + n_dofs, 0.0);
        else
          std::fill(A.values, A.values + A.rowptr[n_dofs], 0.0);
        std::fill(residual_pde, residual_pde + n_dofs, 0.0);
      }

    FEValues<dim> fe_values;
    std::vector<double> local_matrix(dpc * dpc);
    double local_rhs[dpc];
    double U[dpc], Uo[nv], Uoo[nv];
    double xv[nv][dim];

    (void)n_cells;
    for (int64_t cell = cell_begin; cell < cell_end; ++cell)
      {
        for (int v = 0; v < nv; ++v)
          for (int d = 0; d < dim; ++d)
            xv[v][d] = coords[(int64_t)cell_nodes[cell * nv + v] * dim + d];
        fe_values.reinit(xv);

        double lambda = P.lambda, mu = P.mu;
        if (cell_lambda)
          { // cracks.cc:2207-2216 (values resolved by the harness)
            lambda = cell_lambda[cell];
            mu = cell_mu[cell];
          }

        const int32_t *dofs = cell_dofs + cell * dpc;
        for (int i = 0; i < dpc; ++i)
          U[i] = sol[dofs[i]];
        for (int v = 0; v < nv; ++v)
          {
            Uo[v] = old[dofs[v * nc + dim]];
            Uoo[v] = oldold[dofs[v * nc + dim]];
          }

        int err = cell_local<dim>(fe_values, P, lambda, mu, U, Uo, Uoo, residual_only != 0,
                                  local_matrix.data(), local_rhs);
        if (err)
          return err;

        // cracks.cc:2439-2464
        if (residual_only)
          {
            distribute_vector(cu, local_rhs, dofs, dpc, residual_pde);
            if (P.outer_solver == PFM_SOLVER_ACTIVE_SET)
              distribute_vector(ch, local_rhs, dofs, dpc, residual_total);
            else
              distribute_vector(cu, local_rhs, dofs, dpc, residual_total);
          }
        else
          {
            err = distribute_matrix(cu, local_matrix.data(), local_rhs, dofs, dpc, A, residual_pde);
            if (err)
              return err;
          }
      }
    return ORACLE_OK;
  }
} // namespace


  // ------------------------------------------------------------------------------------
  // Newton-side sweeps over the same mesh data
  // ------------------------------------------------------------------------------------

  // assemble_diag_mass_matrix, cracks.cc:2514-2562.  QGaussLobatto<dim>(2) (deal.II): the 2^dim
  // vertices of the reference cell with weight 2^-dim each, so shape_value(i,q) = delta(vertex(i), q)
  // and the phase-field dof of vertex a receives JxW(vertex a) = det J(a) 2^-dim.
  template <int dim>
  void diag_mass(int64_t n_cells, int32_t n_dofs, const int32_t *cell_nodes, const double *coords,
                 const int32_t *cell_dofs, double *diag)
  {
    constexpr int nv = 1 << dim, dpc = nv * (dim + 1);
    for (int32_t i = 0; i < n_dofs; ++i)
      diag[i] = 0.0; // diag_mass = 0, cracks.cc:2518
    for (int64_t cell = 0; cell < n_cells; ++cell)
      {
        double xv[nv][dim];
        for (int v = 0; v < nv; ++v)
          for (int d = 0; d < dim; ++d)
            xv[v][d] = coords[(int64_t)cell_nodes[cell * nv + v] * dim + d];
        double local_rhs[dpc] = {};
        for (int q = 0; q < nv; ++q) // Lobatto point q = vertex q
          {
            double J[dim][dim];
            for (int i = 0; i < dim; ++i)
              for (int j = 0; j < dim; ++j)
                {
                  double s = 0.0;
                  for (int v = 0; v < nv; ++v)
                    {
                      double g = 1.0;
                      for (int d = 0; d < dim; ++d)
                        {
                          const double x = (q >> d) & 1;
                          if (d == j)
                            g *= ((v >> d) & 1) ? 1.0 : -1.0;
                          else
                            g *= ((v >> d) & 1) ? x : (1.0 - x);
                        }
                      s += xv[v][i] * g;
                    }
                  J[i][j] = s;
                }
            double det;
            if constexpr (dim == 2)
              det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
            else
              det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
                    J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
            const double JxW = det / (double)nv;
            for (int i = 0; i < dpc; ++i)
              {
                if (i % (dim + 1) != dim)
                  continue; // only look at phase field, cracks.cc:2548
                const double N = (i / (dim + 1) == q) ? 1.0 : 0.0;
                local_rhs[i] += N * N * JxW;
              }
          }
        for (int i = 0; i < dpc; ++i)
          diag[cell_dofs[cell * dpc + i]] += local_rhs[i]; // cracks.cc:2555-2556
      }
  }

  // compute_energy (cracks.cc:3615-3701) and compute_tcv (cracks.cc:3553-3611) over the owned cells
  template <int dim>
  void functionals(int64_t n_cells, const int32_t *cell_nodes, const double *coords, const int32_t *cell_dofs,
                   const double *cell_lambda, const double *cell_mu, const pfm_params &prm, const double *sol,
                   const uint8_t *cell_owned, double *out)
  {
    constexpr int nv = 1 << dim, dpc = nv * (dim + 1);
    FEValues<dim> fe_values;
    double local_bulk_energy = 0.0, local_crack_energy = 0.0, local_integral = 0.0;
    for (int64_t cell = 0; cell < n_cells; ++cell)
      {
        if (cell_owned && !cell_owned[cell])
          continue; // cell->is_locally_owned()
        double xv[nv][dim];
        for (int v = 0; v < nv; ++v)
          for (int d = 0; d < dim; ++d)
            xv[v][d] = coords[(int64_t)cell_nodes[cell * nv + v] * dim + d];
        fe_values.reinit(xv);
        const double lame_coefficient_lambda = cell_lambda ? cell_lambda[cell] : prm.lambda;
        const double lame_coefficient_mu = cell_mu ? cell_mu[cell] : prm.mu;
        for (int q = 0; q < FEValues<dim>::nq; ++q)
          {
            T2<dim> grad_u;
            double grad_pf[dim] = {}, u[dim] = {}, pf = 0.0;
            for (int i = 0; i < dpc; ++i)
              {
                const int v = i / (dim + 1), c = i % (dim + 1);
                const double val = sol[cell_dofs[cell * dpc + i]];
                if (c < dim)
                  {
                    u[c] += val * fe_values.N[q][v];
                    for (int d = 0; d < dim; ++d)
                      grad_u[c][d] += val * fe_values.dN[q][v][d];
                  }
                else
                  {
                    pf += val * fe_values.N[q][v];
                    for (int d = 0; d < dim; ++d)
                      grad_pf[d] += val * fe_values.dN[q][v][d];
                  }
              }
            T2<dim> E;
            for (int a = 0; a < dim; ++a)
              for (int b = 0; b < dim; ++b)
                E[a][b] = 0.5 * (grad_u[a][b] + grad_u[b][a]);
            const double tr_E = trace(E);
            double tr_e_2 = 0.0; // trace(E*E)
            for (int a = 0; a < dim; ++a)
              for (int b = 0; b < dim; ++b)
                tr_e_2 += E[a][b] * E[b][a];
            const double psi_e = 0.5 * lame_coefficient_lambda * tr_E * tr_E + lame_coefficient_mu * tr_e_2;
            double gg = 0.0, ug = 0.0;
            for (int d = 0; d < dim; ++d)
              {
                gg += grad_pf[d] * grad_pf[d];
                ug += u[d] * grad_pf[d];
              }
            local_bulk_energy += ((1 + prm.constant_k) * pf * pf + prm.constant_k) * psi_e * fe_values.JxW[q];
            local_crack_energy += prm.G_c / 2.0 * ((pf - 1) * (pf - 1) / prm.alpha_eps + prm.alpha_eps * gg) * fe_values.JxW[q];
            local_integral += ug * fe_values.JxW[q]; // cracks.cc:3587
          }
      }
    out[0] = local_bulk_energy;
    out[1] = local_crack_energy;
    out[2] = local_integral;
  }

extern "C"
{
  // Full assembly (cracks.cc:2133-2475 on one rank: compress() is a no-op then).
  // All index spaces are the reference's: global dof indices as delivered by
  // cell->get_dof_indices (cracks.cc:2439), one global CSR over all dofs.
  int oracle_assemble(int dim, int64_t n_cells, int32_t n_dofs, const int32_t *cell_nodes,
                      const double *coords, const int32_t *cell_dofs, const double *cell_lambda,
                      const double *cell_mu, const pfm_params *prm, const double *sol,
                      const double *old, const double *oldold, const uint8_t *cu_flag,
                      const int64_t *cu_ptr, const int32_t *cu_col, const double *cu_w,
                      const uint8_t *ch_flag, const int64_t *ch_ptr, const int32_t *ch_col,
                      const double *ch_w, int residual_only, const int64_t *rowptr,
                      const int32_t *colind, double *values, double *residual_pde,
                      double *residual_total)
  {
    if (!prm || (dim != 2 && dim != 3))
      return ORACLE_BAD_ARG;
    Constraints cu{cu_flag, cu_ptr, cu_col, cu_w};
    Constraints ch{ch_flag, ch_ptr, ch_col, ch_w};
    Csr A{rowptr, colind, values};
    if (dim == 2)
      return assemble<2>(n_cells, n_dofs, cell_nodes, coords, cell_dofs, cell_lambda, cell_mu, *prm,
                         sol, old, oldold, cu, ch, residual_only, A, residual_pde, residual_total, 0,
                         n_cells, 1);
    return assemble<3>(n_cells, n_dofs, cell_nodes, coords, cell_dofs, cell_lambda, cell_mu, *prm,
                       sol, old, oldold, cu, ch, residual_only, A, residual_pde, residual_total, 0,
                       n_cells, 1);
  }

  // Same, restricted to cells [cell_begin, cell_end) and without zeroing: used by
  // the threaded CPU-baseline driver (one colour at a time) in bench.py.
  int oracle_assemble_range(int dim, int64_t n_cells, int32_t n_dofs, const int32_t *cell_nodes,
                            const double *coords, const int32_t *cell_dofs,
                            const double *cell_lambda, const double *cell_mu,
                            const pfm_params *prm, const double *sol, const double *old,
                            const double *oldold, const uint8_t *cu_flag, const int64_t *cu_ptr,
                            const int32_t *cu_col, const double *cu_w, const uint8_t *ch_flag,
                            const int64_t *ch_ptr, const int32_t *ch_col, const double *ch_w,
                            int residual_only, const int64_t *rowptr, const int32_t *colind,
                            double *values, double *residual_pde, double *residual_total,
                            int64_t cell_begin, int64_t cell_end)
  {
    if (!prm || (dim != 2 && dim != 3))
      return ORACLE_BAD_ARG;
    Constraints cu{cu_flag, cu_ptr, cu_col, cu_w};
    Constraints ch{ch_flag, ch_ptr, ch_col, ch_w};
    Csr A{rowptr, colind, values};
    if (dim == 2)
      return assemble<2>(n_cells, n_dofs, cell_nodes, coords, cell_dofs, cell_lambda, cell_mu, *prm,
                         sol, old, oldold, cu, ch, residual_only, A, residual_pde, residual_total,
                         cell_begin, cell_end, 0);
    return assemble<3>(n_cells, n_dofs, cell_nodes, coords, cell_dofs, cell_lambda, cell_mu, *prm,
                       sol, old, oldold, cu, ch, residual_only, A, residual_pde, residual_total,
                       cell_begin, cell_end, 0);
  }

  // One cell's local_matrix (row-major [dpc][dpc], entry (j,i) as the reference
  // stores it) and local_rhs, for element-level parity tests.
  int oracle_cell_local(int dim, const double *vertex_coords /*nv*dim*/, const pfm_params *prm,
                        double lambda, double mu, const double *U, const double *Uold_pf,
                        const double *Uoldold_pf, int residual_only, double *local_matrix,
                        double *local_rhs)
  {
    if (dim == 2)
      {
        FEValues<2> fe;
        double xv[4][2];
        std::memcpy(xv, vertex_coords, sizeof(xv));
        fe.reinit(xv);
        return cell_local<2>(fe, *prm, lambda, mu, U, Uold_pf, Uoldold_pf, residual_only != 0,
                             local_matrix, local_rhs);
      }
    if (dim == 3)
      {
        FEValues<3> fe;
        double xv[8][3];
        std::memcpy(xv, vertex_coords, sizeof(xv));
        fe.reinit(xv);
        return cell_local<3>(fe, *prm, lambda, mu, U, Uold_pf, Uoldold_pf, residual_only != 0,
                             local_matrix, local_rhs);
      }
    return ORACLE_BAD_ARG;
  }

  // eigen_vectors_and_values on a 2x2 matrix (row-major m[4]); evecs[4] is the
  // ev_matrix (columns = eigenvectors) — the six Catch cases, cracks.cc:1740-1919.
  int oracle_eigen_2x2(const double *m, double *eval1, double *eval2, double *evecs)
  {
    T2<2> M, P;
    M[0][0] = m[0];
    M[0][1] = m[1];
    M[1][0] = m[2];
    M[1][1] = m[3];
    int err = eigen_vectors_and_values<2>(*eval1, *eval2, P, M);
    evecs[0] = P[0][0];
    evecs[1] = P[0][1];
    evecs[2] = P[1][0];
    evecs[3] = P[1][1];
    return err;
  }

  // decompose_stress on 2x2 tensors (row-major), both branches.
  int oracle_decompose_stress_2d(const double *E, const double *E_LinU, double lambda, double mu,
                                 int derivative, double *stress_plus, double *stress_minus)
  {
    T2<2> e, el, sp, sm;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        {
          e[i][j] = E[2 * i + j];
          el[i][j] = E_LinU[2 * i + j];
        }
    int err = decompose_stress<2>(sp, sm, e, trace(e), el, trace(el), lambda, mu, derivative != 0);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        {
          stress_plus[2 * i + j] = sp[i][j];
          stress_minus[2 * i + j] = sm[i][j];
        }
    return err;
  }

  // assemble_diag_mass_matrix (cracks.cc:2514-2562); diag has one entry per dof (zero for displacement dofs).
  int oracle_diag_mass(int dim, int64_t n_cells, int32_t n_dofs, const int32_t *cell_nodes, const double *coords,
                       const int32_t *cell_dofs, double *diag)
  {
    if (dim == 2)
      diag_mass<2>(n_cells, n_dofs, cell_nodes, coords, cell_dofs, diag);
    else if (dim == 3)
      diag_mass<3>(n_cells, n_dofs, cell_nodes, coords, cell_dofs, diag);
    else
      return ORACLE_BAD_ARG;
    return ORACLE_OK;
  }

  // out[0] = bulk energy, out[1] = crack energy (cracks.cc:3615-3701), out[2] = TCV (cracks.cc:3553-3611)
  int oracle_functionals(int dim, int64_t n_cells, const int32_t *cell_nodes, const double *coords,
                         const int32_t *cell_dofs, const double *cell_lambda, const double *cell_mu,
                         const pfm_params *prm, const double *sol, const uint8_t *cell_owned, double *out)
  {
    if (!prm || !out)
      return ORACLE_BAD_ARG;
    if (dim == 2)
      functionals<2>(n_cells, cell_nodes, coords, cell_dofs, cell_lambda, cell_mu, *prm, sol, cell_owned, out);
    else if (dim == 3)
      functionals<3>(n_cells, cell_nodes, coords, cell_dofs, cell_lambda, cell_mu, *prm, sol, cell_owned, out);
    else
      return ORACLE_BAD_ARG;
    return ORACLE_OK;
  }

  // Active-set update of newton_active_set (cracks.cc:2837-2886) and the cycle counter (cracks.cc:2903-2909),
  // written over dofs (the reference's cell loop visits every phase-field dof once: "already processed").
  //   is_phi / hanging: per dof flags;  active: in = active_set_old, out = active_set
  //   counts[0] = active dofs, counts[1] = cycling dofs, counts[2] = 1 if the set changed
  int oracle_active_set(int32_t n_dofs, const uint8_t *is_phi, const uint8_t *hanging, const double *residual_relevant,
                        const double *diag_mass_relevant, double c, double *solution, const double *old_solution_relevant,
                        int32_t *cycle_counter, uint8_t *active, int64_t *counts)
  {
    const unsigned int n_cycling_threshold = 5; // cracks.cc:2866
    int64_t n_active = 0, n_cycling_dofs = 0, changed = 0;
    for (int32_t idx = 0; idx < n_dofs; ++idx)
      {
        const uint8_t was = active[idx];
        uint8_t now = 0;
        if (is_phi[idx] && !hanging[idx])
          {
            const double old_value = old_solution_relevant[idx];
            const double new_value = solution[idx];
            const double massm = diag_mass_relevant[idx];
            const double gap = new_value - old_value;
            const double active_set_tolarance = 0.0;
            if (!(residual_relevant[idx] / massm + c * (gap) <= active_set_tolarance &&
                  ((unsigned int)cycle_counter[idx] < n_cycling_threshold)))
              {
                if ((unsigned int)cycle_counter[idx] >= n_cycling_threshold)
                  ++n_cycling_dofs;
                now = 1; // constraints_update.add_line(idx); inhomogeneity 0
                solution[idx] = old_value;
                ++n_active;
              }
          }
        if (was && !now)
          ++cycle_counter[idx]; // cracks.cc:2905-2908
        if (was != now)
          changed = 1;
        active[idx] = now;
      }
    counts[0] = n_active;
    counts[1] = n_cycling_dofs;
    counts[2] = changed;
    return ORACLE_OK;
  }
}
