// kabsch.hpp -- the closed-form 3x3 rotation of the Kabsch step, shared by the ICP kernels (icp.hip, icp_fp32.hip).
#pragma once
#include "common.hpp"

#ifndef ICPFLOW_STAMP
#define ICPFLOW_STAMP(k) do { } while (0)
#endif

namespace icpflow {

// ---------------------------------------------------------------------------------
// 3x3 Kabsch rotation, row-vector convention y = x R:  R = U diag(1,1,det(U V^T)) V^T for
// H = U S V^T (utils_icp_pytorch3d.py:339-362), computed WITHOUT an SVD.  That R is the proper
// rotation maximising sum_ij R_ij H_ij, i.e. Horn's closed-form absolute orientation: the unit
// quaternion q that is the eigenvector of the largest eigenvalue of the symmetric 4x4 matrix N(H)
// below (reflection case included).  lambda_max comes from Newton's method on the characteristic
// quartic started at the upper bound (|Xc|^2 + |Yc|^2) / 2W (monotone from above: every root is
// real), the eigenvector from the adjugate of N - lambda I, whose sixteen 3x3 minors are evaluated
// side by side on sixteen lanes.  A dozen dependent fp64 operations per Newton step replace the
// div / sqrt / rsqrt chains of a Jacobi SVD (three rotations per sweep) in the serial tail of an
// ICP iteration.  horn_rotation returns false when the maximiser is numerically not unique
// (rank(H) <= 1: fewer than three non-collinear correspondences; the reference's answer is then an
// accident of its SVD backend) -- the caller then takes rank1_rotation.
// ---------------------------------------------------------------------------------
static __device__ __forceinline__ double det3(double a, double b, double c, double d, double e, double f, double g,
                                       double h, double i)
{
    // (explicit fused multiply-adds: half the dependent operations of the mul / sub form in the serial tail)
    const double m0 = fma(e, i, -(f * h)), m1 = fma(d, i, -(f * g)), m2 = fma(d, h, -(e * g));
    return fma(c, m2, fma(a, m0, -(b * m1)));
}

static __device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Lane l < 16 returns cofactor (l / 4, l % 4) of the 4x4 matrix M (row-major, LDS): the sixteen 3x3
// minors are evaluated side by side on sixteen lanes instead of one after the other, which also
// keeps the register footprint of the solve at nine doubles.
// The nine element indices of the minor depend on the lane only: computed once per solve (an opaque copy of the
// lane index keeps the compiler from hoisting them out of the ICP loop, where they would be spilled and come back
// through dependent scratch loads in every iteration) and shared by the two cofactor evaluations of a solve.
struct MinorIdx {
    int e[9];
    bool neg;
};

static __device__ __forceinline__ MinorIdx minor_indices(int lane)
{
    asm volatile("" : "+v"(lane));
    const int i = (lane >> 2) & 3, j = lane & 3;
    const int r0 = (0 >= i) ? 4 : 0, r1 = (1 >= i) ? 8 : 4, r2 = (2 >= i) ? 12 : 8;
    const int c0 = (0 >= j) ? 1 : 0, c1 = (1 >= j) ? 2 : 1, c2 = (2 >= j) ? 3 : 2;
    MinorIdx m;
    m.e[0] = r0 + c0; m.e[1] = r0 + c1; m.e[2] = r0 + c2;
    m.e[3] = r1 + c0; m.e[4] = r1 + c1; m.e[5] = r1 + c2;
    m.e[6] = r2 + c0; m.e[7] = r2 + c1; m.e[8] = r2 + c2;
    m.neg = ((i + j) & 1) != 0;
    return m;
}

static __device__ __forceinline__ double cofactor16(const double *M, const MinorIdx &m)
{
    const double d = det3(M[m.e[0]], M[m.e[1]], M[m.e[2]], M[m.e[3]], M[m.e[4]], M[m.e[5]], M[m.e[6]], M[m.e[7]], M[m.e[8]]);
    return m.neg ? -d : d;
}

// Called by all 64 lanes of ONE wave with wave-uniform arguments; Nsh: 16 doubles of LDS scratch.
// *lamOut: the largest eigenvalue = s1 + s2 + s3 (s3 signed by det H) = trace(E S) of the reference's SVD form.
static __device__ bool horn_rotation(const double *S, double gsum, double *Nsh, int lane, double (&R)[9],
                                     double *lamOut = nullptr)
{
    double frob2 = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) frob2 = fma(S[k], S[k], frob2);
    if (!(frob2 > 0.0)) return false;
    // characteristic polynomial  l^4 + c2 l^2 + c1 l + c0  (N is traceless).  With the singular values s1, s2, s3 of H
    // (s3 carrying the sign of det H) the eigenvalues of N are s1+s2+s3, s1-s2-s3, -s1+s2-s3, -s1-s2+s3, hence
    //   c2 = -2 |H|_F^2,   c1 = -8 det H,   c0 = det N = |H|_F^4 - 4 |cof H|_F^2
    // (the squared singular values of the cofactor matrix are the pairwise products): nine 2x2 minors in registers
    // instead of a 4x4 determinant through LDS, and det H comes out of three of them.
    const double m00 = fma(S[4], S[8], -(S[5] * S[7])), m01 = fma(S[3], S[8], -(S[5] * S[6])), m02 = fma(S[3], S[7], -(S[4] * S[6]));
    const double m10 = fma(S[1], S[8], -(S[2] * S[7])), m11 = fma(S[0], S[8], -(S[2] * S[6])), m12 = fma(S[0], S[7], -(S[1] * S[6]));
    const double m20 = fma(S[1], S[5], -(S[2] * S[4])), m21 = fma(S[0], S[5], -(S[2] * S[3])), m22 = fma(S[0], S[4], -(S[1] * S[3]));
    const double cof2 = fma(m22, m22, fma(m21, m21, fma(m20, m20, fma(m12, m12, fma(m11, m11,
                        fma(m10, m10, fma(m02, m02, fma(m01, m01, m00 * m00))))))));
    const double detH = fma(S[2], m02, fma(S[0], m00, -(S[1] * m01)));
    const double c2 = -2.0 * frob2;
    const double c1 = -8.0 * detH;
    const double c0 = fma(frob2, frob2, -4.0 * cof2);
    {
        // N (symmetric 4x4) goes to LDS for the adjugate below; every lane stores the same values
        const double n01 = S[5] - S[7], n02 = S[6] - S[2], n03 = S[1] - S[3];
        const double n12 = S[1] + S[3], n13 = S[6] + S[2], n23 = S[5] + S[7];
        Nsh[1] = n01; Nsh[2] = n02; Nsh[3] = n03;
        Nsh[4] = n01; Nsh[6] = n12; Nsh[7] = n13;
        Nsh[8] = n02; Nsh[9] = n12; Nsh[11] = n23;
        Nsh[12] = n03; Nsh[13] = n13; Nsh[14] = n23;
    }
    const double n00 = S[0] + S[4] + S[8], n11 = S[0] - S[4] - S[8], n22 = -S[0] + S[4] - S[8], n33 = -S[0] - S[4] + S[8];
    ICPFLOW_STAMP(13);
    double lam = 0.5 * gsum, prevStep = 1e300;
    for (int it = 0; it < 40; ++it) {
        const double x2 = lam * lam;
        const double b = (x2 + c2) * lam;
        const double a = b + c1;
        const double den = fma(2.0 * x2, lam, b + a);
        if (den == 0.0) break;
        // quotient by a refined reciprocal (the step need not be correctly rounded: the iteration corrects itself
        // and stops on the size of the step), a third of the dependent operations of an IEEE division
        double rc = __builtin_amdgcn_rcp(den);
        rc = fma(fma(-den, rc, 1.0), rc, rc);
        rc = fma(fma(-den, rc, 1.0), rc, rc);
        const double step = fma(a, lam, c0) * rc;
        lam -= step;
        const double as = fabs(step);
        // steps shrink monotonically above the largest root (real-rooted quartic): the first one
        // that does not is rounding noise.  (Round 4) Newton converges quadratically on a simple root: after a step of
        // relative size 1e-9 the iterate is exact to ~1e-18, the steps that used to follow it (one or two, until the
        // step fell below 1e-16 or stopped shrinking) only moved lam by its own rounding -- a quarter of the
        // iteration's dependent chain in the serial tail of every ICP iteration.
        // The bound holds for a root that is well separated: the error left after a step e is ~ e^2 |f''| / (2 |f'|), and
        // near a (nearly) double top root f' -> 0 while f'' does not (near-collinear correspondences, sigma2 ~ sigma3 with
        // a reflection), so the early exit is taken only where that estimate is below the rounding of lam; otherwise the
        // iteration goes on to the old criterion (step below 1e-16 relative, or not shrinking any more).
        if (as >= prevStep || as <= 1e-16 * fabs(lam)) break;
        if (as <= 1e-9 * fabs(lam) && as * fabs(fma(6.0, x2, c2)) <= 1e-7 * fabs(den)) break;
        prevStep = as;
    }
    ICPFLOW_STAMP(14);
    if (lamOut != nullptr) *lamOut = lam;
    // adjugate of A = N - lam I (symmetric, rank 3): adj = c q q^T, entry (i, j) on lane 4 i + j
    Nsh[0] = n00 - lam; Nsh[5] = n11 - lam; Nsh[10] = n22 - lam; Nsh[15] = n33 - lam;
    const MinorIdx mi = minor_indices(lane);
    const double C = cofactor16(Nsh, mi);
    // column of the largest diagonal entry (c q_k^2): the best conditioned one
    int k = 0;
    double big = fabs(readlane_f64(C, 0));
#pragma unroll
    for (int d = 1; d < 4; ++d) {
        const double v = fabs(readlane_f64(C, 5 * d));
        if (v > big) { big = v; k = d; }
    }
    if (!(big * big > 1e-18 * frob2 * frob2 * frob2)) return false;   // (nearly) double top eigenvalue
    k = __builtin_amdgcn_readfirstlane(k);
    double q0 = readlane_f64(C, 4 * k + 0), q1 = readlane_f64(C, 4 * k + 1), q2 = readlane_f64(C, 4 * k + 2),
           q3 = readlane_f64(C, 4 * k + 3);
    // column-convention rotation Rc (y = Rc x) of the quaternion q / |q|; the row convention wants Rc^T.  The products of
    // the UNNORMALISED q are formed beside the reciprocal of |q|^2 (a refined v_rcp_f64: five dependent operations
    // where rsqrt + four scalings took a dozen) and scaled once.
    const double n2 = fma(q3, q3, fma(q2, q2, fma(q1, q1, q0 * q0)));
    double inv = __builtin_amdgcn_rcp(n2);
    inv = fma(fma(-n2, inv, 1.0), inv, inv);
    inv = fma(fma(-n2, inv, 1.0), inv, inv);
    const double ww = q0 * q0, xx = q1 * q1, yy = q2 * q2, zz = q3 * q3;
    const double xy = q1 * q2, xz = q1 * q3, yz = q2 * q3, wx = q0 * q1, wy = q0 * q2, wz = q0 * q3;
    const double i2 = 2.0 * inv;
    R[0] = (ww + xx - yy - zz) * inv; R[3] = (xy - wz) * i2;            R[6] = (xz + wy) * i2;
    R[1] = (xy + wz) * i2;            R[4] = (ww - xx + yy - zz) * inv; R[7] = (yz - wx) * i2;
    R[2] = (xz - wy) * i2;            R[5] = (yz + wx) * i2;            R[8] = (ww - xx - yy + zz) * inv;
    return true;
}

// rank(H) <= 1:  H = sigma u v^T (or 0).  Every rotation with u R = v maximises sum R_ij H_ij; take the
// smallest one (Rodrigues from u to v).  H = 0 (no gated correspondence): R = I, like torch.svd(0).
static __device__ void rank1_rotation(const double *S, double (&R)[9])
{
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    // v = direction of the largest row of H;  u = H v / |H v|  (signs consistent by construction)
    int r = 0;
    double best = -1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double n2 = S[i * 3] * S[i * 3] + S[i * 3 + 1] * S[i * 3 + 1] + S[i * 3 + 2] * S[i * 3 + 2];
        if (n2 > best) { best = n2; r = i; }
    }
    if (!(best > 0.0)) return;
    const double iv = rsqrt(best);
    const double v[3] = {S[r * 3] * iv, S[r * 3 + 1] * iv, S[r * 3 + 2] * iv};
    double u[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) u[i] = S[i * 3] * v[0] + S[i * 3 + 1] * v[1] + S[i * 3 + 2] * v[2];
    const double un2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    if (!(un2 > 0.0)) return;
    const double iu = rsqrt(un2);
#pragma unroll
    for (int i = 0; i < 3; ++i) u[i] *= iu;
    // column-convention Rc with Rc u = v:  Rc = c I + [w]x + w w^T / (1 + c),  w = u x v, c = u . v
    const double c = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
    double w[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    double Rc[9];
    if (c > -1.0 + 1e-12) {
        const double k = 1.0 / (1.0 + c);
        Rc[0] = c + k * w[0] * w[0];    Rc[1] = k * w[0] * w[1] - w[2]; Rc[2] = k * w[0] * w[2] + w[1];
        Rc[3] = k * w[1] * w[0] + w[2]; Rc[4] = c + k * w[1] * w[1];    Rc[5] = k * w[1] * w[2] - w[0];
        Rc[6] = k * w[2] * w[0] - w[1]; Rc[7] = k * w[2] * w[1] + w[0]; Rc[8] = c + k * w[2] * w[2];
    } else {
        // v = -u: half turn about any axis a perpendicular to u,  Rc = 2 a a^T - I
        const int m = (fabs(u[0]) <= fabs(u[1]) && fabs(u[0]) <= fabs(u[2])) ? 0 : (fabs(u[1]) <= fabs(u[2]) ? 1 : 2);
        double e[3] = {0.0, 0.0, 0.0};
        e[m] = 1.0;
        double a[3] = {u[1] * e[2] - u[2] * e[1], u[2] * e[0] - u[0] * e[2], u[0] * e[1] - u[1] * e[0]};
        const double ia = rsqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] *= ia;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Rc[i * 3 + j] = 2.0 * a[i] * a[j] - (i == j ? 1.0 : 0.0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = Rc[j * 3 + i];   // row convention
}

}  // namespace icpflow
