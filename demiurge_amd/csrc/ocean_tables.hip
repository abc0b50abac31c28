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
    {   // the same divisors as exact two-operation pairs (div2.hip): [L][8] = (h, l) of cx, of Beta, of cy.  A row whose cx or
        // Beta has no exact pair keeps (c, 1/c) for both and is flagged in rowbad; without a pair for cy (one constant for
        // the whole grid) the streaming kernel keeps the three-operation form everywhere.
        std::vector<float> c((size_t)2 * L + 1), h(c.size()), lo(c.size());
        std::vector<unsigned char> ok(c.size());
        for (int l = 0; l < L; l++) { c[l] = jc[(size_t)l * 8 + 0]; c[(size_t)L + l] = jc[(size_t)l * 8 + 2]; }
        c[(size_t)2 * L] = jc[4];
        int rc = dmg_div2_splits(o->ctx, c.data(), (int)c.size(), h.data(), lo.data(), ok.data());
        if (rc) return rc;
        o->div2_ok = ok[(size_t)2 * L] != 0;
        o->div2_nbad = 0;
        std::vector<float> j2((size_t)L * 8, 0.0f);
        std::vector<unsigned char> bad((size_t)L, 0);
        // tests only: DMG_DIV2_FORCE_BAD=<k> treats every k-th grid row as if it had no pair (rows without one are one in a
        // thousand: small grids may have none, and the path that handles them has to be exercised)
        static const int force_bad = [] { const char *e = dmg_hook("DMG_DIV2_FORCE_BAD"); return e ? atoi(e) : 0; }();
        for (int l = 0; l < L; l++) {
            float *e = &j2[(size_t)l * 8];
            const bool forced = force_bad > 0 && ((o->y0 - o->gs + l) % force_bad) == 0;
            if (ok[l] && ok[(size_t)L + l] && !forced) { e[0] = h[l]; e[1] = lo[l]; e[2] = h[(size_t)L + l]; e[3] = lo[(size_t)L + l]; }
            else { for (int k = 0; k < 4; k++) e[k] = jc[(size_t)l * 8 + k]; bad[l] = 1; o->div2_nbad++; }
            e[4] = h[(size_t)2 * L]; e[5] = lo[(size_t)2 * L];
        }
        // (the planner prices the rows without a pair: ocean_jacobi_stream_plan.hip; the tolerance rendering has no divisions, so none)
        if (o->div2_nbad && !(o->prm.quirks & DMG_QUIRK_JACOBI_TOLERANCE)) o->rowbad_host = bad; else o->rowbad_host.clear();
        DMG_HIP(hipMemcpyAsync(o->jcoef2, j2.data(), j2.size() * sizeof(float), hipMemcpyHostToDevice, o->ctx->stream));
        DMG_HIP(hipMemcpyAsync(o->rowbad, bad.data(), bad.size(), hipMemcpyHostToDevice, o->ctx->stream));
        DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    }
    {   // the tolerance rendering's coefficients (DMG_QUIRK_JACOBI_TOLERANCE; ocean_stream_dev.h, st_value2s<3>): the statement
        // ((pW + pE)/pw2.x + (pS + pN)/pw2.y - bC)/Beta is a (pW + pE) + b (pS + pN) - c bC with a = 1/(pw2.x Beta), b = 1/(pw2.y Beta),
        // c = 1/Beta, from the fp32 row constants above (the reference's own) in fp64.  [L][8] = ah, bh, c, e: ah + bh = 0.5 exactly
        // (the larger of a, b rounded, the other 0.5 minus it: exact), e = RN((a - ah) + (b - bh))
        std::vector<float> j3((size_t)L * 8, 0.0f);
        for (int l = 0; l < L; l++) {
            const double cx = jc[(size_t)l * 8 + 0], be = jc[(size_t)l * 8 + 2], cy = jc[(size_t)l * 8 + 4];
            const double a = 1.0 / (cx * be), b = 1.0 / (cy * be);
            float ah, bh;
            if (a >= b) { ah = (float)a; bh = 0.5f - ah; } else { bh = (float)b; ah = 0.5f - bh; }
            float *e = &j3[(size_t)l * 8];
            e[0] = ah; e[1] = bh; e[2] = (float)(1.0 / be); e[3] = (float)((a - (double)ah) + (b - (double)bh));
        }
        DMG_HIP(hipMemcpyAsync(o->jcoef3, j3.data(), j3.size() * sizeof(float), hipMemcpyHostToDevice, o->ctx->stream));
        DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    }
    return DMG_OK;
}
