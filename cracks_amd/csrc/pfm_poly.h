// pfm_poly.h — polynomial-coefficient helpers of the cartesian kernels (round 4).
//
// On a box cell every nodal field is trilinear in the reference coordinates (t, s, r) in [0,1]^3: 8 monomial coefficients
// (index a + 2 b + 4 c <-> t^a s^b r^c), gradients are multilinear with known zero coefficients, products of two such
// polynomials have 27 coefficients (power index i_x + 3 i_y + 9 i_z, powers 0..2).  The 27-point Gauss sums of the
// reference (cracks.cc:2222-2432) over such integrands are contractions of the coefficients with a handful of constants
// (k_cart_phi4: pu_role_poly / pp_role_poly; k_cart_residual3: residual_cell_poly) -- the 3-point rule is exact for every
// degree that occurs, and where a factor is NOT a polynomial (the clamped phase-field extrapolation) its discrete moments
// sum_q w f(q) x_q^m take the place of the integrals: the identity is algebraic, not an approximation.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace pfm
{
  namespace
  {
    template <int N, class F>
    __device__ __forceinline__ __attribute__((always_inline)) void poly_for(F &&f)
    {
      if constexpr (N > 0)
        {
          poly_for<N - 1>(f);
          f(std::integral_constant<int, N - 1>{});
        }
    }

    // in: vertex values, index x + 2 y + 4 z; out: coefficient of t^a s^b r^c at a + 2 b + 4 c
    __device__ __forceinline__ void monomials(double (&v)[8])
    {
#pragma unroll
      for (int i = 0; i < 8; i += 2)
        v[i + 1] -= v[i];
      v[2] -= v[0], v[3] -= v[1], v[6] -= v[4], v[7] -= v[5];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i + 4] -= v[i];
    }
    // power index i_x + 3 i_y + 9 i_z of the product of two multilinear monomials
    constexpr int pow_of(int ia, int ip)
    {
      return ((ia & 1) + (ip & 1)) + 3 * (((ia >> 1) & 1) + ((ip >> 1) & 1)) + 9 * (((ia >> 2) & 1) + ((ip >> 2) & 1));
    }
    // Q += w p^2 for a multilinear polynomial p whose coefficients outside MASK (bit idx) vanish structurally
    template <int MASK>
    __device__ __forceinline__ void add_square(const double (&p)[8], double w, double (&Q)[27])
    {
      poly_for<8>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value;
        if constexpr ((MASK >> i) & 1)
          {
            const double a = w * p[i], a2 = a + a; // w p_i, 2 w p_i: formed per row (all eight rows at once cost 32 registers)
            poly_for<8>([&](auto Jc) __attribute__((always_inline)) {
              constexpr int j = decltype(Jc)::value;
              if constexpr (j >= i && ((MASK >> j) & 1))
                Q[pow_of(i, j)] = fma(j == i ? a : a2, p[j], Q[pow_of(i, j)]);
            });
          }
      });
    }

    // 1-D Gauss(3) constants on [0,1] as literals
    constexpr double gq_t(int q) { return q == 0 ? 0.5 - 0.5 * 0.7745966692414834 : (q == 1 ? 0.5 : 0.5 + 0.5 * 0.7745966692414834); }
    constexpr double gq_w(int q) { return q == 1 ? 8.0 / 18.0 : 5.0 / 18.0; }
    constexpr double gq_wt(int q, int p) { return gq_w(q) * (p == 0 ? 1.0 : (p == 1 ? gq_t(q) : gq_t(q) * gq_t(q))); } // w t^p
    // sum_q w t_q^n = integral of t^n over [0,1] for n <= 5
    constexpr double gq_mom(int n)
    {
      double s = 0.0;
      for (int q = 0; q < 3; ++q)
        {
          double tn = 1.0;
          for (int i = 0; i < n; ++i)
            tn *= gq_t(q);
          s += gq_w(q) * tn;
        }
      return s;
    }
    template <int N>
    struct GqMom
    {
      static constexpr double v = gq_mom(N);
    };
    template <int Q, int P>
    struct GqWt
    {
      static constexpr double v = gq_wt(Q, P);
    };
    template <int Q>
    struct GqT
    {
      static constexpr double v = gq_t(Q);
    };
    constexpr int NOX = 0x55, NOY = 0x33, NOZ = 0x0f; // monomial masks: without t, without s, without r

    // ---- the same in 2-D: bilinear fields, 4 monomial coefficients (index a + 2 b), products with 9 (power index i_x + 3 i_y)
    __device__ __forceinline__ void monomials2(double (&v)[4])
    {
      v[1] -= v[0], v[3] -= v[2];
      v[2] -= v[0], v[3] -= v[1];
    }
    constexpr int pow2_of(int ia, int ip) { return ((ia & 1) + (ip & 1)) + 3 * (((ia >> 1) & 1) + ((ip >> 1) & 1)); }
    template <int MASK>
    __device__ __forceinline__ void add_square2(const double (&p)[4], double w, double (&Q)[9])
    {
      poly_for<4>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int i = decltype(Ic)::value;
        if constexpr ((MASK >> i) & 1)
          {
            const double a = w * p[i], a2 = a + a;
            poly_for<4>([&](auto Jc) __attribute__((always_inline)) {
              constexpr int j = decltype(Jc)::value;
              if constexpr (j >= i && ((MASK >> j) & 1))
                Q[pow2_of(i, j)] = fma(j == i ? a : a2, p[j], Q[pow2_of(i, j)]);
            });
          }
      });
    }
  } // namespace
} // namespace pfm
