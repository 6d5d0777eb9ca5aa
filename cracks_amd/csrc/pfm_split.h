// pfm_split.h — 2-D stress split on the device (decompose_stress + eigen_vectors_and_values, cracks.cc:1691-1737,
// 1923-2120), shared by the general cell kernel (pfm_kernels.hip) and the 2-D row-owner kernel (pfm_cart2d.hip).
// Operation order follows the reference statement by statement: the IEEE corner cases (diagonal / zero strain: 0/0 and
// x/0 in the derivative branch) must produce the reference's NaN / Inf pattern (tests/test_gpu_split_corners.py).
#pragma once
#include <hip/hip_runtime.h>

namespace pfm
{
  namespace
  {
    // ------------------------------------------------------------ stress split (2-D)
    // eigen_vectors_and_values (cracks.cc:1691-1737) on a symmetric 2x2 tensor.
    // P = [v1 v2] (columns).  Returns false when the orthogonality check fails.
    __device__ __forceinline__ bool eigen2(double m00, double m01, double m10, double m11, double &l1,
                                           double &l2, double P[2][2])
    {
      double v1x, v1y, v2x, v2y;
      if (fabs(m01) < 1e-10 * fabs(m00) || fabs(m01) < 1e-10 * fabs(m11))
        {
          l1 = m00;
          v1x = 1;
          v1y = 0;
          l2 = m11;
          v2x = 0;
          v2y = 1;
        }
      else
        {
          const double sq = sqrt((m00 - m11) * (m00 - m11) + 4.0 * m01 * m10);
          l1 = 0.5 * ((m00 + m11) + sq);
          l2 = 0.5 * ((m00 + m11) - sq);
          const double t1 = (l1 - m00) / m01, t2 = (l2 - m00) / m01;
          const double s1 = sqrt(1 + t1 * (l1 - m00) / m01), s2 = sqrt(1 + t2 * (l2 - m00) / m01);
          v1x = 1.0 / s1;
          v1y = (l1 - m00) / (m01 * s1);
          v2x = 1.0 / s2;
          v2y = (l2 - m00) / (m01 * s2);
        }
      P[0][0] = v1x;
      P[0][1] = v2x;
      P[1][0] = v1y;
      P[1][1] = v2y;
      return !(v1x * v2x + v1y * v2y > 1.0e-6);
    }

    // decompose_stress(..., derivative=false), cracks.cc:1959-1970
    __device__ __forceinline__ bool split_stress(const double E[2][2], double trE, double lam, double mu,
                                                 double sp[2][2], double sm[2][2])
    {
      double l1, l2, P[2][2];
      const bool ok = eigen2(E[0][0], E[0][1], E[1][0], E[1][1], l1, l2, P);
      const double l1p = fmax(0.0, l1), l2p = fmax(0.0, l2);
      const double trp = fmax(0.0, trE);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          {
            // (P Lambda+ P^T)_ij
            const double Ep = (P[i][0] * l1p) * P[j][0] + (P[i][1] * l2p) * P[j][1];
            const double id = (i == j) ? 1.0 : 0.0;
            sp[i][j] = lam * trp * id + 2 * mu * Ep;
            sm[i][j] = lam * (trE - trp) * id + 2 * mu * (E[i][j] - Ep);
          }
      return ok;
    }

    // decompose_stress(..., derivative=true), cracks.cc:1976-2109
    __device__ __forceinline__ bool split_stress_lin(const double E[2][2], double trE, const double EL[2][2],
                                                     double trEL, double lam, double mu, double sp[2][2],
                                                     double sm[2][2])
    {
      double l1, l2, P[2][2];
      const bool ok = eigen2(E[0][0], E[0][1], E[1][0], E[1][1], l1, l2, P);
      const double l1p = fmax(0.0, l1), l2p = fmax(0.0, l2);
      const double E00 = E[0][0], E01 = E[0][1], E10 = E[1][0], E11 = E[1][1];

      const double disk = sqrt(E01 * E10 + (E00 - E11) * (E00 - E11) / 4.0);
      const double mix = EL[0][1] * E10 + E01 * EL[1][0] + (E00 - E11) * (EL[0][0] - EL[1][1]) / 2.0;
      const double l1L = 0.5 * trEL + 1.0 / (2.0 * disk) * mix;
      const double l2L = 0.5 * trEL - 1.0 / (2.0 * disk) * mix;

      const double t1 = (l1 - E00) / E01, t2 = (l2 - E00) / E01;
      const double q1 = 1.0 + t1 * (l1 - E00) / E01, q2 = 1.0 + t2 * (l2 - E00) / E01;
      const double n1 = 1.0 / sqrt(q1), n2 = 1.0 / sqrt(q2);
      // d/dU of (l - E00)/E01
      const double dt1 = ((l1L - EL[0][0]) * E01 - (l1 - E00) * EL[0][1]) / (E01 * E01);
      const double dt2 = ((l2L - EL[0][0]) * E01 - (l2 - E00) * EL[0][1]) / (E01 * E01);
      const double n1L = -1.0 * (1.0 / q1 * 1.0 / (2.0 * sqrt(q1)) * (2.0 * t1) * dt1);
      const double n2L = -1.0 * (1.0 / q2 * 1.0 / (2.0 * sqrt(q2)) * (2.0 * t2) * dt2);

      double PL[2][2];
      PL[0][0] = n1 * 0.0 + n1L * 1.0;
      PL[1][0] = n1 * dt1 + n1L * (l1 - E00) / E01;
      PL[0][1] = n2 * 0.0 + n2L * 1.0;
      PL[1][1] = n2 * dt2 + n2L * (l2 - E00) / E01;

      const double l1pL = (l1 < 0.0) ? 0.0 : l1L;
      const double l2pL = (l2 < 0.0) ? 0.0 : l2L;
      const double trpL = (trE < 0.0) ? 0.0 : trEL;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          {
            const double a = (PL[i][0] * l1p) * P[j][0] + (PL[i][1] * l2p) * P[j][1];
            const double b = (P[i][0] * l1pL) * P[j][0] + (P[i][1] * l2pL) * P[j][1];
            const double c = (P[i][0] * l1p) * PL[j][0] + (P[i][1] * l2p) * PL[j][1];
            const double EpL = a + b + c;
            const double id = (i == j) ? 1.0 : 0.0;
            sp[i][j] = lam * trpL * id + 2 * mu * EpL;
            sm[i][j] = lam * (trEL - trpL) * id + 2 * mu * (EL[i][j] - EpL);
          }
      return ok;
    }

  } // namespace
} // namespace pfm
