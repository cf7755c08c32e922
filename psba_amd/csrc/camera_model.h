// camera_model.h -- projection, residual and analytic Jacobian of one observation (fp64).
//
// Model (reference: CL_files/compute_exQT.cl:36-69, compute_jacobiQT.cl:113-140; restated
// from the mathematics in SURVEY.md Appendix B, not from the Maple expression list):
//   q_l = (sqrt(1-|v|^2), v),  q = q_l (x) q0  (Hamilton product, local rotation on the left)
//   P   = R'(q) M + t,  R'(q) = 2 u u^T + (s^2 - |u|^2) I + 2 s [u]x   (q = (s,u); the
//         quaternion sandwich q (0,M) q*, equal to the rotation matrix for unit q)
//   x   = (fu Px + sk Py + u0 Pz)/Pz,  y = (fu ar Py + v0 Pz)/Pz,  K = (fu,u0,v0,ar,sk)
//   e   = measured - (x,y);  A = d(x,y)/d(v,t) (2x6 row-major),  B = d(x,y)/dM (2x3 row-major)
#pragma once
#include <hip/hip_runtime.h>

namespace psba {

#define PSBA_HD __host__ __device__ __forceinline__

struct Quat {
  double s, u0, u1, u2;
};

PSBA_HD Quat compose_quat(const double *q0, double v0, double v1, double v2, double &sl) {
  sl = sqrt(1.0 - v0 * v0 - v1 * v1 - v2 * v2);
  const double s0 = q0[0], a0 = q0[1], a1 = q0[2], a2 = q0[3];
  Quat q;
  q.s = sl * s0 - (a0 * v0 + a1 * v1 + a2 * v2);
  q.u0 = s0 * v0 + sl * a0 + a2 * v1 - a1 * v2;
  q.u1 = s0 * v1 + sl * a1 + a0 * v2 - a2 * v0;
  q.u2 = s0 * v2 + sl * a2 + a1 * v0 - a0 * v1;
  return q;
}

// R'(q), row-major
PSBA_HD void quat_matrix(const Quat &q, double *R) {
  const double ss = q.s * q.s, x = q.u0, y = q.u1, z = q.u2;
  const double xx = x * x, yy = y * y, zz = z * z;
  R[0] = ss + xx - yy - zz;
  R[4] = ss - xx + yy - zz;
  R[8] = ss - xx - yy + zz;
  const double xy = x * y, xz = x * z, yz = y * z, sx = q.s * x, sy = q.s * y, sz = q.s * z;
  R[1] = 2.0 * (xy - sz);
  R[2] = 2.0 * (xz + sy);
  R[3] = 2.0 * (xy + sz);
  R[5] = 2.0 * (yz - sx);
  R[6] = 2.0 * (xz - sy);
  R[7] = 2.0 * (yz + sx);
}

// residual only.  cam = (v0,v1,v2,t0,t1,t2)
PSBA_HD void residual_obs(const double *K, const double *q0, const double *cam, const double *M,
                          double mx, double my, double &e0, double &e1) {
  double sl, R[9];
  const Quat q = compose_quat(q0, cam[0], cam[1], cam[2], sl);
  quat_matrix(q, R);
  const double Px = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + cam[3];
  const double Py = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + cam[4];
  const double Pz = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + cam[5];
  const double inv = 1.0 / Pz;
  e0 = mx - (K[0] * Px + K[4] * Py + K[1] * Pz) * inv;
  e1 = my - (K[0] * K[3] * Py + K[2] * Pz) * inv;
}

// residual + Jacobian blocks
// xn: (optional) the normalised image coordinates (Px / Pz, Py / Pz): what d(x, y) / dK needs
PSBA_HD void linearize_obs(const double *K, const double *q0, const double *cam, const double *M,
                           double mx, double my, double *e, double *A, double *B, double *xn = nullptr) {
  double sl, R[9];
  const Quat q = compose_quat(q0, cam[0], cam[1], cam[2], sl);
  quat_matrix(q, R);
  const double Px = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + cam[3];
  const double Py = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + cam[4];
  const double Pz = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + cam[5];
  const double inv = 1.0 / Pz;
  const double x = (K[0] * Px + K[4] * Py + K[1] * Pz) * inv;
  const double y = (K[0] * K[3] * Py + K[2] * Pz) * inv;
  e[0] = mx - x;
  e[1] = my - y;
  if (xn) {
    xn[0] = Px * inv;
    xn[1] = Py * inv;
  }
  // D = d(x,y)/dP
  const double d00 = K[0] * inv, d01 = K[4] * inv, d02 = (K[1] - x) * inv;
  const double d11 = K[0] * K[3] * inv, d12 = (K[2] - y) * inv;
  // translation columns
  A[3] = d00;
  A[4] = d01;
  A[5] = d02;
  A[9] = 0.0;
  A[10] = d11;
  A[11] = d12;
  // B = D R'
  B[0] = d00 * R[0] + d01 * R[3] + d02 * R[6];
  B[1] = d00 * R[1] + d01 * R[4] + d02 * R[7];
  B[2] = d00 * R[2] + d01 * R[5] + d02 * R[8];
  B[3] = d11 * R[3] + d12 * R[6];
  B[4] = d11 * R[4] + d12 * R[7];
  B[5] = d11 * R[5] + d12 * R[8];
  // rotation columns: dq_l/dv_k = (-v_k/s_l, e_k), dq = dq_l (x) q0 = (ds, du);
  // dP = 2 du (u.M) + 2 u (du.M) + 2 (s ds - u.du) M + 2 ds (u x M) + 2 s (du x M)
  const double s0 = q0[0], a0 = q0[1], a1 = q0[2], a2 = q0[3];
  const double isl = 1.0 / sl;
  const double udM = q.u0 * M[0] + q.u1 * M[1] + q.u2 * M[2];
  const double c0 = q.u1 * M[2] - q.u2 * M[1];  // u x M
  const double c1 = q.u2 * M[0] - q.u0 * M[2];
  const double c2 = q.u0 * M[1] - q.u1 * M[0];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double dsl = -cam[k] * isl;
    const double ak = (k == 0) ? a0 : (k == 1 ? a1 : a2);
    const double ds = dsl * s0 - ak;
    // e_k x a
    const double x0 = (k == 0) ? 0.0 : (k == 1 ? a2 : -a1);
    const double x1 = (k == 0) ? -a2 : (k == 1 ? 0.0 : a0);
    const double x2 = (k == 0) ? a1 : (k == 1 ? -a0 : 0.0);
    const double du0 = ((k == 0) ? s0 : 0.0) + dsl * a0 + x0;
    const double du1 = ((k == 1) ? s0 : 0.0) + dsl * a1 + x1;
    const double du2 = ((k == 2) ? s0 : 0.0) + dsl * a2 + x2;
    const double dudM = du0 * M[0] + du1 * M[1] + du2 * M[2];
    const double udu = q.u0 * du0 + q.u1 * du1 + q.u2 * du2;
    const double g = q.s * ds - udu;
    const double m0 = du1 * M[2] - du2 * M[1];  // du x M
    const double m1 = du2 * M[0] - du0 * M[2];
    const double m2 = du0 * M[1] - du1 * M[0];
    const double dP0 = 2.0 * (du0 * udM + q.u0 * dudM + g * M[0] + ds * c0 + q.s * m0);
    const double dP1 = 2.0 * (du1 * udM + q.u1 * dudM + g * M[1] + ds * c1 + q.s * m1);
    const double dP2 = 2.0 * (du2 * udM + q.u2 * dudM + g * M[2] + ds * c2 + q.s * m2);
    A[k] = d00 * dP0 + d01 * dP1 + d02 * dP2;
    A[6 + k] = d11 * dP1 + d12 * dP2;
  }
}

// Free intrinsics (the reference reads 11 parameters per camera, PSBA/main.cpp:73,140-149, and never
// optimises the first five, CL_files/PSBA.cl:5-7): camera block p = (fu, u0, v0, ar, s | v0, v1, v2 | t0, t1, t2).
// x = fu xn + s yn + u0, y = fu ar yn + v0 with (xn, yn) = (Px, Py) / Pz, so
//   d(x, y) / d(fu, u0, v0, ar, s) = [ xn, 1, 0, 0, yn ;  ar yn, 0, 1, fu yn, 0 ]
// and the other six columns are those of the fixed-K Jacobian.  A is 2 x 11 row-major.
constexpr int FK_CNP = 11;
PSBA_HD void linearize_obs_freek(const double *p, const double *q0, const double *M, double mx, double my, double *e,
                                 double *A, double *B) {
  double A6[12], xn[2];
  linearize_obs(p, q0, p + 5, M, mx, my, e, A6, B, xn);
  A[0] = xn[0];
  A[1] = 1.0;
  A[2] = 0.0;
  A[3] = 0.0;
  A[4] = xn[1];
  A[FK_CNP + 0] = p[3] * xn[1];
  A[FK_CNP + 1] = 0.0;
  A[FK_CNP + 2] = 1.0;
  A[FK_CNP + 3] = p[0] * xn[1];
  A[FK_CNP + 4] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    A[5 + k] = A6[k];
    A[FK_CNP + 5 + k] = A6[6 + k];
  }
}

// symmetric 3x3 inverse by the closed form the reference uses (T = -det,
// CL_files/compute_Vinv.cl:29,76-86).  v = (v00,v01,v02,v11,v12,v22) -> same packing.
// Returns true when |T| < 1e-16 (the reference's singular flag).
PSBA_HD bool sym3_inverse(const double *v, double *o) {
  const double a11 = v[0], a12 = v[1], a13 = v[2], a22 = v[3], a23 = v[4], a33 = v[5];
  const double T = (a33 * a12 * a12 - 2.0 * a12 * a13 * a23 + a22 * a13 * a13 + a11 * a23 * a23 -
                    a11 * a22 * a33);
  const double iT = -1.0 / T;
  o[0] = (a22 * a33 - a23 * a23) * iT;
  o[1] = (a13 * a23 - a12 * a33) * iT;
  o[2] = (a12 * a23 - a13 * a22) * iT;
  o[3] = (a11 * a33 - a13 * a13) * iT;
  o[4] = (a12 * a13 - a11 * a23) * iT;
  o[5] = (a11 * a22 - a12 * a12) * iT;
  return fabs(T) < 1e-16;
}

}  // namespace psba
