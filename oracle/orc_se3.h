/* oracle/orc_se3.h — SE3Quat / quaternion helpers shared by the pose-optimisation and bundle-adjustment oracles.
 * TEST INFRASTRUCTURE ONLY.  Restates g2o::SE3Quat (src/sg-slam/Thirdparty/g2o/g2o/types/se3quat.h:41-296) and the
 * Eigen quaternion operations it calls (Eigen is external to the reference tree: parity unpinned, see poseopt_oracle.c). */
#ifndef ORC_SE3_H
#define ORC_SE3_H
#include <math.h>
#include <string.h>
typedef struct { double q[4]; /* x,y,z,w */ double t[3]; } se3q;

static inline void quat_from_R(const double R[3][3], double q[4])
{   /* Eigen Quaternion = Matrix3 */
    double t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[2][1] - R[1][2]) * t; q[1] = (R[0][2] - R[2][0]) * t; q[2] = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0; if (R[1][1] > R[0][0]) i = 1; if (R[2][2] > R[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[k][j] - R[j][k]) * t; q[j] = (R[j][i] + R[i][j]) * t; q[k] = (R[k][i] + R[i][k]) * t;
    }
}
static inline void quat_normalize_rot(double q[4])
{   /* SE3Quat::normalizeRotation se3quat.h:280-285 */
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] *= -1;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
static inline void quat_rotate(const double q[4], const double v[3], double o[3])
{   /* Eigen QuaternionBase::_transformVector */
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static inline void quat_mul(const double a[4], const double b[4], double o[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static inline void quat_to_R(const double q[4], double R[3][3])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}
static inline void se3_map(const se3q *T, const double x[3], double o[3]) { quat_rotate(T->q, x, o); o[0] += T->t[0]; o[1] += T->t[1]; o[2] += T->t[2]; }

static inline void mat3_mul(const double A[3][3], const double B[3][3], double C[3][3])
{ for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[i][k] * B[k][j]; C[i][j] = s; } }

static inline void se3_exp(const double u[6], se3q *out)
{   /* SE3Quat::exp se3quat.h:223-257 (omega = u[0..2], upsilon = u[3..5]) */
    const double w[3] = { u[0], u[1], u[2] }, up[3] = { u[3], u[4], u[5] };
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double O[3][3] = { { 0, -w[2], w[1] }, { w[2], 0, -w[0] }, { -w[1], w[0], 0 } };
    double O2[3][3]; mat3_mul(O, O, O2);
    double R[3][3], V[3][3];
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = (i == j) + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            R[i][j] = (i == j) + a * O[i][j] + b * O2[i][j];
            V[i][j] = (i == j) + b * O[i][j] + c * O2[i][j];
        }
    }
    quat_from_R(R, out->q);
    for (int i = 0; i < 3; i++) out->t[i] = V[i][0] * up[0] + V[i][1] * up[1] + V[i][2] * up[2];
    quat_normalize_rot(out->q);
}
static inline void se3_mul(const se3q *a, const se3q *b, se3q *o)
{   /* SE3Quat::operator* se3quat.h:104-110 */
    double rt[3]; quat_rotate(a->q, b->t, rt);
    se3q r; r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
    quat_mul(a->q, b->q, r.q); quat_normalize_rot(r.q);
    *o = r;
}
static inline void se3_from_cv(const float *T, se3q *o)
{   /* Converter::toSE3Quat */
    double R[3][3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[i][j] = T[4 * i + j]; o->t[i] = T[4 * i + 3]; }
    quat_from_R(R, o->q); quat_normalize_rot(o->q);
}
static inline void se3_to_cv(const se3q *s, float *T)
{   /* Converter::toCvMat(SE3Quat) */
    double R[3][3]; quat_to_R(s->q, R);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = (float)R[i][j]; T[4 * i + 3] = (float)s->t[i]; }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}


#endif
