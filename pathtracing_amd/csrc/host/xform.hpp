// Matrix set-up of a `transform` node, computed once on the host in the
// reference's f32 arithmetic: Mat4::inverse (math/src/lib.rs:586-648, cofactors
// through get_minor :564-584 and Mat3::det :426-430) and Mat4::normal_matrix
// (:554-562 over Mat3::transponsed_inverse :432-450).
#pragma once
#include "../kernels/pt_vec.h"

namespace pt {

inline float det3(const float r0[3], const float r1[3], const float r2[3]) {
    return r0[0] * (r1[1] * r2[2] - r2[1] * r1[2]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0]) +
           r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

inline float minor4(const ptk::M4& a, int row, int col) {
    float rows[3][3];
    int n = 0;
    for (int r = 0; r < 4; r++) {
        if (r == row) continue;
        int k = 0;
        for (int c = 0; c < 4; c++) {
            if (c == col) continue;
            rows[n][k++] = a.m[r][c];
        }
        n++;
    }
    return det3(rows[0], rows[1], rows[2]);
}

inline ptk::M4 inverse4(const ptk::M4& a) {
    ptk::M4 cof;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float m = minor4(a, r, c);
            cof.m[r][c] = ((r + c) & 1) ? -m : m;
        }
    float det = a.m[0][0] * cof.m[0][0] + a.m[0][1] * cof.m[0][1] + a.m[0][2] * cof.m[0][2] + a.m[0][3] * cof.m[0][3];
    float f = 1.0f / det;
    ptk::M4 inv;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) inv.m[c][r] = cof.m[r][c] * f; // scale, then transpose
    return inv;
}

// Transposed inverse of the upper-left 3x3.
inline ptk::M3 normal_matrix4(const ptk::M4& a) {
    const float(*s)[4] = a.m;
    float c[3][3];
    c[0][0] = s[1][1] * s[2][2] - s[2][1] * s[1][2];
    c[0][1] = s[1][2] * s[2][0] - s[1][0] * s[2][2];
    c[0][2] = s[1][0] * s[2][1] - s[2][0] * s[1][1];
    c[1][0] = s[0][2] * s[2][1] - s[0][1] * s[2][2];
    c[1][1] = s[0][0] * s[2][2] - s[0][2] * s[2][0];
    c[1][2] = s[2][0] * s[0][1] - s[0][0] * s[2][1];
    c[2][0] = s[0][1] * s[1][2] - s[0][2] * s[1][1];
    c[2][1] = s[1][0] * s[0][2] - s[0][0] * s[1][2];
    c[2][2] = s[0][0] * s[1][1] - s[1][0] * s[0][1];
    float r0[3] = {s[0][0], s[0][1], s[0][2]}, r1[3] = {s[1][0], s[1][1], s[1][2]}, r2[3] = {s[2][0], s[2][1], s[2][2]};
    float f = 1.0f / det3(r0, r1, r2);
    ptk::M3 o;
    o.row0 = ptk::mk3(c[0][0] * f, c[0][1] * f, c[0][2] * f);
    o.row1 = ptk::mk3(c[1][0] * f, c[1][1] * f, c[1][2] * f);
    o.row2 = ptk::mk3(c[2][0] * f, c[2][1] * f, c[2][2] * f);
    return o;
}

} // namespace pt
