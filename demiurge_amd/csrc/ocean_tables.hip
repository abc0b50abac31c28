// Per-row coefficient tables.  The GLSL evaluates these per fragment (pixelsize() has a cos(), the Jacobi
// shader two pow()s, OceanCurrents.cpp:478-479); they depend on the row only, so they are computed once on
// the host, in fp32, statement by statement as the shader does, and the kernels read 3 floats per row.
#include "ocean_internal.h"
#include <algorithm>

int ocean_upload_row_tables(dmg_ocean *o)
{
    const int L = o->L;
    std::vector<float> t((size_t)ROWTAB_FIELDS * L);
    const float scale = o->prm.scale;
    for (int l = 0; l < L; l++) {
        int y = o->y0 - o->gs + l;
        float dx, dy;
        dmg_row_pixelsize(o->grid, y, &dx, &dy);
        float wx = dx / scale, wy = dy / scale;                 // pixelsize(st)/420
        float cx = wx * wx, cy = wy * wy;                       // pow(.,vec2(2,2))
        t[(size_t)RT_JCX * L + l] = cx;
        t[(size_t)RT_JCY * L + l] = cy;
        float beta = 2.0f * (1.0f / cx + 1.0f / cy);
        t[(size_t)RT_JBETA * L + l] = beta;
        t[(size_t)RT_JRX * L + l] = 1.0f / cx;               // IEEE division on the host: correctly rounded
        t[(size_t)RT_JRY * L + l] = 1.0f / cy;
        t[(size_t)RT_JRB * L + l] = 1.0f / beta;
        t[(size_t)RT_DX * L + l] = dx;
        t[(size_t)RT_DY * L + l] = dy;
        // 1*1/pow(pixelsize,2)*420*420 and 2*(pw2.x+pw2.y) as Mesa's algebraic pass leaves them (probe-pinned, see
        // oracle/dmg_oracle.c orc_ocean_diffusion): the two constants merge, and the common factor is pulled out of the sum
        float ix = 1.0f / (dx * dx), iy = 1.0f / (dy * dy);
        float px = ix * (scale * scale), py = iy * (scale * scale);
        float s = (ix + iy) * (2.0f * (scale * scale));
        t[(size_t)RT_DIFPX * L + l] = px;
        t[(size_t)RT_DIFPY * L + l] = py;
        const float dbeta = s * (1.0f + 1.0f / s);
        t[(size_t)RT_DIFBETA * L + l] = dbeta;
        t[(size_t)RT_DIFRB * L + l] = 1.0f / dbeta;          // correctly rounded reciprocal (blocked diffusion)
        t[(size_t)RT_LAT * L + l] = dmg_row_lat(o->grid, y);
    }
    DMG_HIP(hipMemcpyAsync(o->rowtab, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, o->ctx->stream));
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    o->jac_cx = o->rt(RT_JCX);
    o->jac_cy = o->rt(RT_JCY);
    o->jac_beta = o->rt(RT_JBETA);
    o->jac_rx = o->rt(RT_JRX);
    o->jac_ry = o->rt(RT_JRY);
    o->jac_rb = o->rt(RT_JRB);
    std::vector<float> jc((size_t)L * 8, 0.0f);
    for (int l = 0; l < L; l++) {
        static const int f[6] = { RT_JCX, RT_JRX, RT_JBETA, RT_JRB, RT_JCY, RT_JRY };     // first four: one 16-byte load per row
        for (int k = 0; k < 6; k++) jc[(size_t)l * 8 + k] = t[(size_t)f[k] * L + l];
    }
    {   // the same coefficients, 4 per row, for local rows -16 .. L+15: a row beyond the grid is the mirrored row (as the
        // streaming kernels map it), a row beyond the stored band the band's last row
        const int PAD = 16, H = o->grid.H, y_first = o->y0 - o->gs;
        std::vector<float> jp((size_t)(L + 2 * PAD) * 4);
        for (int e = -PAD; e < L + PAD; e++) {
            int y = y_first + e;
            if (y < 0) y = -1 - y; else if (y >= H) y = 2 * H - 1 - y;
            const int l = std::min(std::max(y - y_first, 0), L - 1);
            for (int k = 0; k < 4; k++) jp[(size_t)(e + PAD) * 4 + k] = jc[(size_t)l * 8 + k];
        }
        DMG_HIP(hipMemcpyAsync(o->jcoefp, jp.data(), jp.size() * sizeof(float), hipMemcpyHostToDevice, o->ctx->stream));
        DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    }
    DMG_HIP(hipMemcpyAsync(o->jcoef, jc.data(), jc.size() * sizeof(float), hipMemcpyHostToDevice, o->ctx->stream));
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    return DMG_OK;
}
