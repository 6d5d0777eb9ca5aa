// pfm_split.h — 2-D stress split on the device (decompose_stress + eigen_vectors_and_values, cracks.cc:1691-1737,
// 1923-2120), shared by the general cell kernel (pfm_kernels.hip) and the 2-D row-owner kernel (pfm_cart2d.hip).
// The closed forms are the reference's; its divisions are taken as products with shared reciprocals (see below).  The IEEE
// corner cases (diagonal / zero strain: 0/0 and x/0 in the derivative branch) produce the reference's NaN / Inf pattern
// (tests/test_gpu_split_corners.py).
#pragma once
#include <hip/hip_runtime.h>

namespace pfm
{
  namespace
  {
    // ------------------------------------------------------------ stress split (2-D)
    // eigen_vectors_and_values (cracks.cc:1691-1737) on a symmetric 2x2 tensor, together with the quantities of E alone that
    // the derivative branch of decompose_stress (cracks.cc:1976-2109) shares with it.  P = [v1 v2] (columns).
    //
    // Every quotient of the two reference functions has one of four divisors -- E_01 (and its square), the discriminant root,
    // q_i = 1 + ((l_i - E_00) / E_01)^2 and sqrt(q_i) -- and none of them depends on the direction E_LinU.  They are formed
    // once per q-point as reciprocals (the three calls of a q-point are identical expressions of E: the compiler shares
    // them) and the reference's divisions become products: 6 instead of 34 FP64 divisions per q-point in the Jacobian
    // kernel.  x / y and x * (1 / y) agree to an ulp and give the same NaN / Inf for y = 0 (x / 0 = x * inf, 0 / 0 = 0 * inf):
    // tests/test_gpu_split_corners.py pins that pattern and the 1e-12 parity against the oracle (which divides, statement
    // by statement as the reference does) and keeps the reference's abort() condition reachable.
    struct SplitCommon
    {
      double l1, l2, P[2][2];
      double d1, d2, r01, t1, t2, n1, n2; // l_i - E_00, 1 / E_01, d_i / E_01, 1 / sqrt(q_i)
      bool ok;                            // false when the orthogonality check fails
    };
    __device__ __forceinline__ void split_common(double m00, double m01, double m10, double m11, SplitCommon &C)
    {
      const bool diag = fabs(m01) < 1e-10 * fabs(m00) || fabs(m01) < 1e-10 * fabs(m11);
      const double sq = sqrt((m00 - m11) * (m00 - m11) + 4.0 * m01 * m10);
      C.l1 = diag ? m00 : 0.5 * ((m00 + m11) + sq);
      C.l2 = diag ? m11 : 0.5 * ((m00 + m11) - sq);
      C.r01 = 1.0 / m01;
      C.d1 = C.l1 - m00;
      C.d2 = C.l2 - m00;
      C.t1 = C.d1 * C.r01;
      C.t2 = C.d2 * C.r01;
      const double q1 = 1.0 + C.t1 * C.d1 * C.r01, q2 = 1.0 + C.t2 * C.d2 * C.r01;
      C.n1 = 1.0 / sqrt(q1);
      C.n2 = 1.0 / sqrt(q2);
      const double v1x = diag ? 1.0 : C.n1, v1y = diag ? 0.0 : C.t1 * C.n1;
      const double v2x = diag ? 0.0 : C.n2, v2y = diag ? 1.0 : C.t2 * C.n2;
      C.P[0][0] = v1x;
      C.P[0][1] = v2x;
      C.P[1][0] = v1y;
      C.P[1][1] = v2y;
      C.ok = !(v1x * v2x + v1y * v2y > 1.0e-6);
    }

    // decompose_stress(..., derivative=false), cracks.cc:1959-1970
    __device__ __forceinline__ bool split_stress(const double E[2][2], double trE, double lam, double mu,
                                                 double sp[2][2], double sm[2][2])
    {
      SplitCommon C;
      split_common(E[0][0], E[0][1], E[1][0], E[1][1], C);
      const double l1p = fmax(0.0, C.l1), l2p = fmax(0.0, C.l2);
      const double trp = fmax(0.0, trE);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          {
            // (P Lambda+ P^T)_ij
            const double Ep = (C.P[i][0] * l1p) * C.P[j][0] + (C.P[i][1] * l2p) * C.P[j][1];
            const double id = (i == j) ? 1.0 : 0.0;
            sp[i][j] = lam * trp * id + 2 * mu * Ep;
            sm[i][j] = lam * (trE - trp) * id + 2 * mu * (E[i][j] - Ep);
          }
      return C.ok;
    }

    // decompose_stress(..., derivative=true), cracks.cc:1976-2109
    __device__ __forceinline__ bool split_stress_lin(const double E[2][2], double trE, const double EL[2][2],
                                                     double trEL, double lam, double mu, double sp[2][2],
                                                     double sm[2][2])
    {
      SplitCommon C;
      split_common(E[0][0], E[0][1], E[1][0], E[1][1], C);
      const double l1 = C.l1, l2 = C.l2;
      const double l1p = fmax(0.0, l1), l2p = fmax(0.0, l2);
      const double E00 = E[0][0], E01 = E[0][1], E10 = E[1][0], E11 = E[1][1];

      const double disk = sqrt(E01 * E10 + (E00 - E11) * (E00 - E11) / 4.0);
      const double r_disk2 = 1.0 / (2.0 * disk);
      const double mix = EL[0][1] * E10 + E01 * EL[1][0] + (E00 - E11) * (EL[0][0] - EL[1][1]) / 2.0;
      const double l1L = 0.5 * trEL + r_disk2 * mix;
      const double l2L = 0.5 * trEL - r_disk2 * mix;

      const double r01sq = C.r01 * C.r01;
      const double n1 = C.n1, n2 = C.n2;
      // d/dU of (l - E00)/E01
      const double dt1 = ((l1L - EL[0][0]) * E01 - C.d1 * EL[0][1]) * r01sq;
      const double dt2 = ((l2L - EL[0][0]) * E01 - C.d2 * EL[0][1]) * r01sq;
      // -1/q 1/(2 sqrt q) 2 t dt = -n^3 t dt
      const double n1L = -1.0 * ((n1 * n1) * (0.5 * n1) * (2.0 * C.t1) * dt1);
      const double n2L = -1.0 * ((n2 * n2) * (0.5 * n2) * (2.0 * C.t2) * dt2);

      double PL[2][2];
      PL[0][0] = n1 * 0.0 + n1L * 1.0;
      PL[1][0] = n1 * dt1 + n1L * C.t1;
      PL[0][1] = n2 * 0.0 + n2L * 1.0;
      PL[1][1] = n2 * dt2 + n2L * C.t2;

      const double l1pL = (l1 < 0.0) ? 0.0 : l1L;
      const double l2pL = (l2 < 0.0) ? 0.0 : l2L;
      const double trpL = (trE < 0.0) ? 0.0 : trEL;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          {
            const double a = (PL[i][0] * l1p) * C.P[j][0] + (PL[i][1] * l2p) * C.P[j][1];
            const double b = (C.P[i][0] * l1pL) * C.P[j][0] + (C.P[i][1] * l2pL) * C.P[j][1];
            const double c = (C.P[i][0] * l1p) * PL[j][0] + (C.P[i][1] * l2p) * PL[j][1];
            const double EpL = a + b + c;
            const double id = (i == j) ? 1.0 : 0.0;
            sp[i][j] = lam * trpL * id + 2 * mu * EpL;
            sm[i][j] = lam * (trEL - trpL) * id + 2 * mu * (EL[i][j] - EpL);
          }
      return C.ok;
    }

  } // namespace
} // namespace pfm
