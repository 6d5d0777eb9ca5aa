// pfm_split.h — 2-D stress split on the device (decompose_stress + eigen_vectors_and_values, cracks.cc:1691-1737,
// 1923-2120), used by the general cell kernel and its patch form (pfm_kernels.hip).
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
    //
    // Round 4: square roots and reciprocals by the hardware estimate (v_rsq_f64 / v_rcp_f64, ~2^-26) and two coupled
    // Newton steps -- sqrt(x) and 1 / sqrt(x) come out of the same iteration -- instead of the IEEE-rounded library
    // sequences (15 instructions each): results within 2 ulp, zero and infinity as IEEE gives them (sqrt(0) = 0, 1 / 0 = inf,
    // 1 / sqrt(inf) = 0), NaN for negative arguments.  The discriminant root of eigen_vectors_and_values
    // (sqrt((E00 - E11)^2 + 4 E01 E10)) and `diskriminante` of decompose_stress (sqrt(E01 E10 + (E00 - E11)^2 / 4)) are the
    // same number up to the factor 2: one evaluation serves both and the quotient 1 / (2 diskriminante).
    struct RootPair
    {
      double root, rroot; // sqrt(x), 1 / sqrt(x)
    };
    __device__ __forceinline__ RootPair sqrt_rsqrt(double x)
    {
      const double y = __builtin_amdgcn_rsq(x);
      double g = x * y, h = 0.5 * y;
      double r = fma(-h, g, 0.5);
      g = fma(g, r, g);
      h = fma(h, r, h);
      r = fma(-h, g, 0.5);
      g = fma(g, r, g);
      h = fma(h, r, h);
      const bool edge = __builtin_isinf(x) || x == 0.0; // 0 * inf above: take the IEEE results
      return RootPair{edge ? x : g, edge ? y : 2.0 * h};
    }
    __device__ __forceinline__ double recip(double x)
    {
      const double y0 = __builtin_amdgcn_rcp(x);
      double y = fma(fma(-x, y0, 1.0), y0, y0);
      y = fma(fma(-x, y, 1.0), y, y);
      // x = 0, x = inf AND arguments whose estimate over- or underflows (subnormal or near-maximal |x|): the refinement
      // would turn inf / 0 into NaN there (fma(-x, inf, 1) * inf); the raw estimate is what IEEE division gives up to the
      // last binade (1 / subnormal = inf or 1.8e308)
      return (__builtin_isinf(y0) || y0 == 0.0) ? y0 : y;
    }

    struct SplitCommon
    {
      double l1, l2, P[2][2];
      double d1, d2, r01, t1, t2, n1, n2; // l_i - E_00, 1 / E_01, d_i / E_01, 1 / sqrt(q_i)
      double r_disc;                      // 1 / sqrt((E00 - E11)^2 + 4 E01 E10) = 1 / (2 diskriminante)
      bool ok;                            // false when the orthogonality check fails
    };
    __device__ __forceinline__ void split_common(double m00, double m01, double m10, double m11, SplitCommon &C)
    {
      const bool diag = fabs(m01) < 1e-10 * fabs(m00) || fabs(m01) < 1e-10 * fabs(m11);
      const RootPair disc = sqrt_rsqrt((m00 - m11) * (m00 - m11) + 4.0 * m01 * m10);
      const double sq = disc.root;
      C.r_disc = disc.rroot;
      C.l1 = diag ? m00 : 0.5 * ((m00 + m11) + sq);
      C.l2 = diag ? m11 : 0.5 * ((m00 + m11) - sq);
      C.r01 = recip(m01);
      C.d1 = C.l1 - m00;
      C.d2 = C.l2 - m00;
      // (l_i - E_00) / E_01 as the reference divides it (cracks.cc:1716-1721): with a subnormal E_01 the reciprocal alone
      // overflows where the quotient of two equally tiny numbers is an ordinary number -- numerator and denominator are
      // scaled by 2^200 there (exact), so that d_i / E_01 stays what IEEE division gives
      const bool tiny = fabs(m01) < 0x1p-900;
      const double rs = tiny ? recip(m01 * 0x1p200) : C.r01, up = tiny ? 0x1p200 : 1.0;
      C.t1 = (C.d1 * up) * rs;
      C.t2 = (C.d2 * up) * rs;
      const double q1 = 1.0 + (tiny ? C.t1 * C.t1 : C.t1 * C.d1 * C.r01), q2 = 1.0 + (tiny ? C.t2 * C.t2 : C.t2 * C.d2 * C.r01);
      C.n1 = sqrt_rsqrt(q1).rroot;
      C.n2 = sqrt_rsqrt(q2).rroot;
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

      const double r_disk2 = C.r_disc; // 1 / (2 diskriminante), see split_common
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

    // decompose_stress(..., derivative=true) (cracks.cc:1976-2109) for the three unit directions e0 e0, e1 e1, sym(e0 e1)
    // at once: the function is linear in E_LinU for a fixed strain, the Jacobian rows of ALL trial dofs of a cell are
    // combinations of these three results (pfm_kernels.hip).  Same statements as split_stress_lin with E_LinU = (1,0;0,0),
    // (0,0;0,1), (0,.5;.5,0) and tr = 1, 1, 0 put in; products with an exact zero of a FINITE factor are left out (x * 0 of a
    // NaN / Inf x cannot occur there: d_i, E_10 and n_i * 0 + n_iL keep their NaN through the other term), everything that
    // does not depend on the direction is formed once.  Output in Voigt order r = 00, 11, 01.
    __device__ __forceinline__ bool split_tangent(const double E[2][2], double trE, double lam, double mu,
                                                  double spL[3][3] /* [k][r] */, double smL[3][3])
    {
      SplitCommon C;
      split_common(E[0][0], E[0][1], E[1][0], E[1][1], C);
      const double l1 = C.l1, l2 = C.l2;
      const double l1p = fmax(0.0, l1), l2p = fmax(0.0, l2);
      const double E00 = E[0][0], E01 = E[0][1], E10 = E[1][0], E11 = E[1][1];
      const double r_disk2 = C.r_disc; // 1 / (2 diskriminante), see split_common
      const double r01sq = C.r01 * C.r01;
      const double n1 = C.n1, n2 = C.n2;
      const double w1 = (n1 * n1) * (0.5 * n1) * (2.0 * C.t1), w2 = (n2 * n2) * (0.5 * n2) * (2.0 * C.t2);
      // P Lambda+ and the dyads of the eigenvectors
      const double Pl[2][2] = {{C.P[0][0] * l1p, C.P[0][1] * l2p}, {C.P[1][0] * l1p, C.P[1][1] * l2p}};
      const double B1[3] = {C.P[0][0] * C.P[0][0], C.P[1][0] * C.P[1][0], C.P[0][0] * C.P[1][0]};
      const double B2[3] = {C.P[0][1] * C.P[0][1], C.P[1][1] * C.P[1][1], C.P[0][1] * C.P[1][1]};
      const double h = (E00 - E11) / 2.0;
      const double mixk[3] = {h, -h, 0.5 * E10 + E01 * 0.5};
#pragma unroll
      for (int k = 0; k < 3; ++k)
        {
          const double trEL = k < 2 ? 1.0 : 0.0;
          const double l1L = 0.5 * trEL + r_disk2 * mixk[k];
          const double l2L = 0.5 * trEL - r_disk2 * mixk[k];
          double dt1, dt2;
          if (k == 0)
            {
              dt1 = ((l1L - 1.0) * E01) * r01sq;
              dt2 = ((l2L - 1.0) * E01) * r01sq;
            }
          else if (k == 1)
            {
              dt1 = (l1L * E01) * r01sq;
              dt2 = (l2L * E01) * r01sq;
            }
          else
            {
              dt1 = (l1L * E01 - C.d1 * 0.5) * r01sq;
              dt2 = (l2L * E01 - C.d2 * 0.5) * r01sq;
            }
          const double n1L = -1.0 * (w1 * dt1), n2L = -1.0 * (w2 * dt2);
          const double PL[2][2] = {{n1L, n2L}, {n1 * dt1 + n1L * C.t1, n2 * dt2 + n2L * C.t2}};
          const double l1pL = (l1 < 0.0) ? 0.0 : l1L;
          const double l2pL = (l2 < 0.0) ? 0.0 : l2L;
          const double trpL = (trE < 0.0) ? 0.0 : trEL;
          // (P_L Lambda+ P^T)_ij = a_ij, (P Lambda+_L P^T)_ij = b_ij, (P Lambda+ P_L^T)_ij = a_ji
          const double a00 = PL[0][0] * Pl[0][0] + PL[0][1] * Pl[0][1], a11 = PL[1][0] * Pl[1][0] + PL[1][1] * Pl[1][1];
          const double a01 = PL[0][0] * Pl[1][0] + PL[0][1] * Pl[1][1], a10 = PL[1][0] * Pl[0][0] + PL[1][1] * Pl[0][1];
          const double EpL[3] = {(a00 + (l1pL * B1[0] + l2pL * B2[0])) + a00, (a11 + (l1pL * B1[1] + l2pL * B2[1])) + a11,
                                 (a01 + (l1pL * B1[2] + l2pL * B2[2])) + a10};
          const double ELr[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 0.5 : 0.0};
#pragma unroll
          for (int r = 0; r < 3; ++r)
            {
              const double id = r < 2 ? 1.0 : 0.0;
              spL[k][r] = lam * trpL * id + 2 * mu * EpL[r];
              smL[k][r] = lam * (trEL - trpL) * id + 2 * mu * (ELr[r] - EpL[r]);
            }
        }
      return C.ok;
    }

  } // namespace
} // namespace pfm
