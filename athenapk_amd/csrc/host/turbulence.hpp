// turbulence.hpp -- host half of the few-modes turbulence driver (BASELINE config 4).
// The spectral state is 3 x num_modes complex numbers; like the reference we keep it, its RNG
// (std::mt19937 + std::uniform_real_distribution, few_modes_ft.hpp:39-40) and the
// Ornstein-Uhlenbeck update on the host so that runs are reproducible on any device, and only
// ship the 3*M coefficients to the GPU each cycle (apk_fmft_inverse).
//   FewModesFT ctor checks   src/utils/few_modes_ft.cpp:30-99
//   SetPhases                src/utils/few_modes_ft.cpp:101-195
//   Generate (spectral part) src/utils/few_modes_ft.cpp:197-320
#pragma once

#include <cmath>
#include <complex>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <vector>

namespace apk {

class FewModesFT {
 public:
  using Complex = std::complex<double>;

  FewModesFT(int num_modes, std::vector<double> k_vec /*[3][M]*/, double k_peak, double sol_weight, double t_corr,
             uint32_t rseed, const int gnx[3])
      : M_(num_modes), k_(std::move(k_vec)), k_peak_(k_peak), sol_weight_(sol_weight), t_corr_(t_corr),
        hat_(3 * (size_t)num_modes), hat_new_(3 * (size_t)num_modes), dist_(-1.0, 1.0) {
    static const char *axis[3] = {"x1", "x2", "x3"};
    for (int d = 0; d < 3; ++d)
      for (int m = 0; m < M_; ++m)
        if (std::abs(k(d, m)) > gnx[d] / 2) throw std::runtime_error(std::string("k_vec ") + axis[d] + " mode too large");
    if (!((sol_weight == -1.0) || (sol_weight >= 0.0 && sol_weight <= 1.0)))
      throw std::runtime_error("sol_weight for projection in few modes fft module needs to be between 0.0 and 1.0 "
                               "or set to -1.0 (to disable projection).");
    rng_.seed(rseed);
  }

  int num_modes() const { return M_; }
  double k(int d, int m) const { return k_[(size_t)d * M_ + m]; }
  const std::vector<Complex> &var_hat() const { return hat_; }  // [3][M]
  std::vector<Complex> &var_hat() { return hat_; }
  std::mt19937 &rng() { return rng_; }
  std::uniform_real_distribution<> &dist() { return dist_; }

  // phase table of one axis for n interior cells starting at global index g0 on a gn-cell axis:
  // out(re|im, m, idx) = exp(i * 2 pi k_axis(m) / gn * ((idx + g0) mod gn)), the k_x = 0 modes
  // halved along x1 (the complex-to-real transform counts them twice)
  void Phases(int axis, int n, int g0, int gn, double *out /*[2][M][n]*/) const {
    for (int idx = 0; idx < n; ++idx) {
      const double g = static_cast<double>((idx + g0) % gn);
      for (int m = 0; m < M_; ++m) {
        const double w = k(axis, m) * 2. * M_PI / static_cast<double>(gn);
        const double arg = w * g;
        double re = std::cos(arg), im = std::sin(arg);
        if (axis == 0 && k(0, m) == 0.0) {
          re = 0.5 * re;
          im = 0.5 * im;
        }
        out[((size_t)0 * M_ + m) * n + idx] = re;
        out[((size_t)1 * M_ + m) * n + idx] = im;
      }
    }
  }

  // advance the spectral field by dt: new realisation of the injection spectrum, conjugate
  // symmetry on the k_x = 0 plane, Helmholtz projection, OU blend with the previous state
  void Evolve(double dt) {
    std::vector<double> rnd(3 * (size_t)M_ * 2);
    for (int n = 0; n < 3; ++n)
      for (int m = 0; m < M_; ++m) {
        double v1, v2, v_sqr;
        do {  // rejection step of the polar Box-Muller method
          v1 = dist_(rng_);
          v2 = dist_(rng_);
          v_sqr = v1 * v1 + v2 * v2;
        } while (v_sqr >= 1.0 || v_sqr == 0.0);
        rnd[((size_t)n * M_ + m) * 2] = v1;
        rnd[((size_t)n * M_ + m) * 2 + 1] = v2;
      }
    for (int n = 0; n < 3; ++n)
      for (int m = 0; m < M_; ++m) {
        const double kmag = std::sqrt(k(0, m) * k(0, m) + k(1, m) * k(1, m) + k(2, m) * k(2, m));
        double spec = std::pow(kmag / k_peak_, 2.) * (2. - std::pow(kmag / k_peak_, 2.));
        if (spec < 0.) spec = 0.;
        const double r0 = rnd[((size_t)n * M_ + m) * 2], r1 = rnd[((size_t)n * M_ + m) * 2 + 1];
        const double v_sqr = r0 * r0 + r1 * r1;
        const double norm = std::sqrt(-2.0 * std::log(v_sqr) / v_sqr);
        hat_new_[(size_t)n * M_ + m] = Complex(spec * norm * r0, spec * norm * r1);
      }
    for (int n = 0; n < 3; ++n)
      for (int m = 0; m < M_; ++m)
        if (k(0, m) == 0.)
          for (int m2 = 0; m2 < m; ++m2)
            if (k(1, m) == -k(1, m2) && k(2, m) == -k(2, m2)) {
              const Complex o = hat_new_[(size_t)n * M_ + m2];
              hat_new_[(size_t)n * M_ + m] = Complex(o.real(), -o.imag());
            }
    if (sol_weight_ >= 0.0) {
      const double sw = sol_weight_;
      for (int m = 0; m < M_; ++m) {
        double kh[3] = {k(0, m), k(1, m), k(2, m)};
        double kmag = std::sqrt(kh[0] * kh[0] + kh[1] * kh[1] + kh[2] * kh[2]);
        if (kmag == 0.) kmag = 1.;
        for (double &c : kh) c /= kmag;
        const Complex a0 = hat_new_[m], a1 = hat_new_[(size_t)M_ + m], a2 = hat_new_[2 * (size_t)M_ + m];
        const double dot_r = a0.real() * kh[0] + a1.real() * kh[1] + a2.real() * kh[2];
        const double dot_i = a0.imag() * kh[0] + a1.imag() * kh[1] + a2.imag() * kh[2];
        for (int n = 0; n < 3; ++n) {
          const Complex a = hat_new_[(size_t)n * M_ + m];
          hat_new_[(size_t)n * M_ + m] = Complex(a.real() * sw + (1. - 2. * sw) * dot_r * kh[n],
                                                 a.imag() * sw + (1. - 2. * sw) * dot_i * kh[n]);
        }
      }
    }
    const double c_drift = std::exp(-dt / t_corr_);
    const double c_diff = std::sqrt(1.0 - c_drift * c_drift);
    for (size_t q = 0; q < hat_.size(); ++q)
      hat_[q] = Complex(hat_[q].real() * c_drift + hat_new_[q].real() * c_diff,
                        hat_[q].imag() * c_drift + hat_new_[q].imag() * c_diff);
  }

 private:
  int M_;
  std::vector<double> k_;
  double k_peak_, sol_weight_, t_corr_;
  std::vector<Complex> hat_, hat_new_;
  std::mt19937 rng_;
  std::uniform_real_distribution<> dist_;
};

}  // namespace apk
