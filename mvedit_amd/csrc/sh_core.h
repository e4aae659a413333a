// Real spherical-harmonics encoding of directions and its Jacobian (sh.hip), host/device.  Replaces the reference's only other native
// op, lib/ops/shencoder (src/shencoder.cu:28-337 kernel_sh, :359-384 kernel_sh_backward; Python lib/ops/shencoder/sphere_harmonics.py:
// SHEncoder(degree <= 8), the direction encoder of the tri-plane decoders, lib/models/decoders/triplane_ingp_decoder.py:202).
//
// The reference hard-codes every basis polynomial; here they come out of the textbook recurrences, evaluated in the same un-normalised
// form (polynomials in x, y, z that coincide with Y_lm on the unit sphere, Condon-Shortley phase, index l^2 + l + m):
//   A_m + i B_m = (x + i y)^m
//   Q_m^m = (-1)^m (2m-1)!!,  Q_{m+1}^m = (2m+1) z Q_m^m,  Q_l^m = ((2l-1) z Q_{l-1}^m - (l+m-1) Q_{l-2}^m) / (l-m)       (P_l^m / sin^m)
//   Y_l^0 = K_l^0 Q_l^0,   Y_l^{+m} = sqrt(2) K_l^m A_m Q_l^m,   Y_l^{-m} = sqrt(2) K_l^m B_m Q_l^m,   K_l^m = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!)
// and the Jacobian differentiates exactly these polynomials (d/dz acts on Q only, d/dx and d/dy on A, B only), as the reference's dy_dx does.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MVE_SHE_FN __device__ __forceinline__
#else
#define MVE_SHE_FN static inline
#endif

#define MVE_SH_MAX_DEGREE 8

// normalisation constants K_l^m (times sqrt 2 for m > 0), computed once on the host in double: k[l * 8 + m]
static inline void she_constants(float* k) {
    for (int l = 0; l < MVE_SH_MAX_DEGREE; ++l)
        for (int m = 0; m <= l; ++m) {
            double ratio = 1.0;                                        // (l-m)! / (l+m)!
            for (int t = l - m + 1; t <= l + m; ++t) ratio /= (double)t;
            const double v = sqrt((2.0 * l + 1.0) / (4.0 * 3.14159265358979323846) * ratio);
            k[l * MVE_SH_MAX_DEGREE + m] = (float)(m ? 1.4142135623730950488 * v : v);
        }
}

// out[C*C]; jac (optional) [3][C*C] = d out / d x, d y, d z (the reference's dy_dx layout)
MVE_SHE_FN void she_eval(const float* k, float x, float y, float z, int C, float* out, float* jac) {
    const int C2 = C * C;
    float A = 1.f, B = 0.f;                  // (x + i y)^m
    float dAx = 0.f, dBx = 0.f;              // d/dx of A, B  (d/dy: dA/dy = -dB/dx, dB/dy = dA/dx by Cauchy-Riemann)
    float qmm = 1.f;                         // Q_m^m
    for (int m = 0; m < C; ++m) {
        if (m > 0) {
            // derivatives first: d/dx (x+iy)^m = m (x+iy)^(m-1)
            dAx = (float)m * A; dBx = (float)m * B;
            const float An = A * x - B * y, Bn = A * y + B * x;
            A = An; B = Bn;
            qmm *= -(float)(2 * m - 1);
        }
        float q2 = 0.f, q1 = 0.f, dq2 = 0.f, dq1 = 0.f;      // Q_{l-2}^m, Q_{l-1}^m and their z-derivatives
        for (int l = m; l < C; ++l) {
            float q, dq;
            if (l == m) { q = qmm; dq = 0.f; }
            else if (l == m + 1) { q = (float)(2 * m + 1) * z * q1; dq = (float)(2 * m + 1) * q1; }
            else {
                const float a = (float)(2 * l - 1), b = (float)(l + m - 1), inv = 1.0f / (float)(l - m);
                q = (a * z * q1 - b * q2) * inv;
                dq = (a * (q1 + z * dq1) - b * dq2) * inv;
            }
            const float kk = k[l * MVE_SH_MAX_DEGREE + m];
            const int ip = l * l + l + m, im = l * l + l - m;
            out[ip] = kk * A * q;
            if (m) out[im] = kk * B * q;
            if (jac) {
                jac[ip] = kk * dAx * q; jac[C2 + ip] = -kk * dBx * q; jac[2 * C2 + ip] = kk * A * dq;
                if (m) { jac[im] = kk * dBx * q; jac[C2 + im] = kk * dAx * q; jac[2 * C2 + im] = kk * B * dq; }
            }
            q2 = q1; dq2 = dq1; q1 = q; dq1 = dq;
        }
    }
}
