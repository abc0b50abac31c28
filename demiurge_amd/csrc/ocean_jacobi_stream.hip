// The streaming Jacobi kernels (device code shared with the deep form: ocean_stream_dev.h) and their launcher; the cut of a launch
// into units is host code of its own (ocean_jacobi_stream_plan.hip).
#include "ocean_stream_dev.h"
#include "ocean_stream_plan.h"
#include <algorithm>
#include <chrono>


// one wave per unit; two waves per SIMD leave 256 registers each: all issue slots of a SIMD belong to two fat waves
// Units that stream SOUTHWARDS (round 4).  The kernel is bound by its bytes (above), and a unit of n rows reads n + 16: the
// eight rows beyond each end are read again by the unit above and the unit below -- 9 % of a 180-row unit's reads, 36 % of a
// 28-row unit's (a 1024-row band of an 8-GPU split).  A unit that streams northwards reads its lower halo FIRST and its upper
// halo LAST; its neighbour above, streaming northwards too, reads those same rows first, a whole unit's time earlier or
// later: by then they have left the L2.  So every other segment of the cut streams the other way: a northward unit and the
// southward unit above it END on the same sixteen rows, that southward unit and the northward unit above IT START on the same
// sixteen rows -- at the same time when they were dispatched together (a band's single round), behind the same L2 (units of
// a strip share an XCD: st_units' XCD-aware order) -- and one of the two finds them in the cache.
// A southward unit is the same program on the mirrored band: the kernel takes a second view of the band (`a_dn`: pointers to
// the last row, a negative row stride, the row tables reversed, y_first and the poles mirrored, a unit's rows counted from
// the top) and a unit's flag picks the view.  The formula is symmetric in north and south -- (pS + pN) and the land-neighbour
// counts commute -- and the code bits are swapped on arrival like a mirrored row's: the same bits, by construction and by
// tests/stream_worker.py (every case runs with and without DMG_ST_FLIP).
// Measured (profiles/r04_southward_units.txt, one build, DMG_ST_FLIP=0/1, two repetitions each): 16384x8192 survey 3.931 / 3.938 ->
// 3.959 / 3.970 M (+0.8 %), all-ocean 2.933 / 2.943 -> 2.997 / 2.998 M (+2.0 %), 16384x2048 +0.5 / +1.5 %, a 16384x1024 band
// +0.2 % / within its noise (the middle band of an 8-way split with its ghost rows: 59.8 -> 59.3 us per launch).  Small, never a
// loss: the default.  What it says about the small band: its 36 % of re-read halo rows are not what it waits for.
template <int DV>
__global__ __launch_bounds__(64, 2) void k_jacobi_stream(st_args a_up, st_args a_dn)
{
    __shared__ v4f bring[ST_RING][64];
    __shared__ unsigned cring[ST_RING][64];          // coast rows: land-neighbour counts, eight fp4 values per row and lane (st_row_any)
    __shared__ v4f ktab[ST_MAXROWS + 2 * TB_T + 6];   // coefficients {cx, 1/cx, Beta, 1/Beta} of every window row, filled once
    const int lane = threadIdx.x;
    const int4 u = a_up.units[blockIdx.x];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const st_args &a = (__builtin_amdgcn_readfirstlane(u.w) & 2) ? a_dn : a_up;                // bit 1: a unit that streams southwards (its rows are the mirrored view's)
    const int W = a.W;
    // (DV = 3: with the open-unit program inlined this kernel crashes the register allocator of ROCm 7.2's clang under the
    // iterative-ILP scheduling strategy (VirtRegAuxInfo::isRematerializable); its open units run the generic program)
    if (ST_OPEN_UNITS && DV != 3 && !ST_DBG(a, 1) && (__builtin_amdgcn_readfirstlane(u.w) & 1)) {       // an open-water unit (st_units flags it)
        if (a.prof && lane == 0) {
            a.prof[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
            a.prof[3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        }
#ifdef ST_EXPERIMENTS
        if (a.clk && lane == 0) a.clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime();
#endif
        st_open_unit<DV>(a, bring, ktab, lane, tx, r0, r1);
        if (a.prof && lane == 0) a.prof[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#ifdef ST_EXPERIMENTS
        if (a.clk && lane == 0) a.clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
#endif
        return;
    }
#ifdef ST_EXPERIMENTS
    if (a.clk && lane == 0) a.clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime();
#endif
    if (a.prof && lane == 0) {
        a.prof[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        // where the unit ran: HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID
        a.prof[3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
    // two window rows per step, and an even number of steps: the loop body is two steps with no branch in between, so that
    // every wait for a staged load is a counted vmcnt (a conditional second step makes the compiler wait for vmcnt(0))
    const int nsteps = (((r1 - r0) + 2 * TB_T + 3) / 4) * 2;
    const int e0 = r0 - TB_T;                       // unclamped local row of the first window row
    int col = tx * TB_OW - TB_T + 4 * lane;         // this lane's 4 columns (wrapped), plain and pole-shifted
    const int xcol = col;                           // unwrapped, for the store
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);

    // (Measured and dropped, round 3: requesting the window's next 8 .. 32 rows up front -- LDS-DMA loads into a scrap of
    // LDS -- so that the pipeline's own requests hit the L2 while it fills.  Bit-identical, and slower: a 16384 x 1024 band's
    // launch went from 67 to 71 / 74 / 80 us.  A unit's fixed cost is not memory latency: with every load pointed at cached
    // rows and the stores dropped the same launch still takes 58 us; see DESIGN.md 4.1b, "what a unit's first steps cost".)
    // the first two steps' rows are requested before anything else: their round trip covers the tables' below
    st_state z;
    {
        const int e_last = e0 + 2 * nsteps + 5;             // the last row a step requests (st_step: e_t + 5)
        z.plain = e0 >= 0 && e_last < a.L && a.y_first + e0 >= 0 && a.y_first + e_last < a.H;
        z.nx_p = a.pin + (long)e0 * a.rs; z.nx_b = a.b + (long)e0 * a.rs; z.nx_c = a.code + (long)e0 * a.rs;
    }
    st_fetch(a, z, 0, 0, e0, col, colp);
    st_fetch(a, z, 0, 1, e0 + 1, col, colp);
    st_fetch(a, z, 1, 0, e0 + 2, col, colp);          // (rows past the window are clamped into the band: harmless)
    st_fetch(a, z, 1, 1, e0 + 3, col, colp);

    unsigned cls_lo, cls_hi = 0u;                            // (cls_lo: four classes per lane, st_class4)
    st_unit_tables<DV>(a, lane, tx, e0, 2 * nsteps + 2, cls_lo, ktab);

#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
    z.landbits = 0xFFFFFFFFu;                                // rows before the window: "land", so that a level idles until its rows exist
    z.mixbits = 0; z.slot = 0; z.badbits = 0;
    z.fc[0][0] = st_class4(cls_lo, 0); z.fc[0][1] = st_class4(cls_lo, 1); z.fc[1][0] = st_class4(cls_lo, 2); z.fc[1][1] = st_class4(cls_lo, 3);
    {
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    // (the two waves of a SIMD taking turns at priority, as in k_jacobi_stream_twin: measured here, 16384 x 8192, and dropped -- 277.5
    // against 272.4 us, 364.7 against 359.3 without land: in a launch of several rounds the older wave ending early is what
    // staggers the rounds)
    for (int t = 0; t < nsteps; t += 2) {
#ifdef ST_EXPERIMENTS
        if (a.stepprof && lane == 0 && t < 44) a.stepprof[(size_t)blockIdx.x * 64 + t / 2] = __builtin_amdgcn_s_memrealtime();
#endif
        st_step<0, DV>(a, z, bring, cring, ktab, lane, t, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol);
        st_step<1, DV>(a, z, bring, cring, ktab, lane, t + 1, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol);
    }
    if (a.prof && lane == 0) a.prof[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#ifdef ST_EXPERIMENTS
    if (a.clk && lane == 0) a.clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
    if (a.stepprof && lane == 0) a.stepprof[(size_t)blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Twin units (st_step's TW form, described there): a workgroup of two waves per pair -- wave 0 the unit above the seam (the
// plain view), wave 1 the unit below it (the mirrored view); a_up.units holds two entries per workgroup.  The window of a twin
// unit starts ON its first row (no halo towards the seam); ST_TWIN_MAXROWS keeps the two waves' tables within a quarter of a
// CU's LDS: four workgroups, eight waves, two per SIMD as before.
template <int DV>
__global__ __launch_bounds__(128, 2) void k_jacobi_stream_twin(st_args a_up, st_args a_dn)
{
    __shared__ v4f bring2[2][ST_RING][64];
    __shared__ unsigned cring2[2][ST_RING][64];
    __shared__ v4f ktab2[2][ST_TWIN_MAXROWS + TB_T + 8];
    // the rows the two waves hand each other travel through ring slots 10 and 11 of the SENDER's right-hand-side ring: a unit's own
    // rows reach those slots with the arrivals of step 5, the last hand-over is read after step 3, and one more barrier at the top of
    // step 4 (st_step) keeps the two apart whatever the waves' pace -- no LDS of their own, which is what lets the tables hold 232 rows
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    v4f (*bring)[64] = bring2[wave];
    unsigned (*cring)[64] = cring2[wave];
    v4f *ktab = ktab2[wave];
#ifdef ST_EXPERIMENTS
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int4 u = a_up.units[2 * blockIdx.x + wave];
    const int tx = __builtin_amdgcn_readfirstlane(u.x), r0 = __builtin_amdgcn_readfirstlane(u.y), r1 = __builtin_amdgcn_readfirstlane(u.z);
    const int uflags = __builtin_amdgcn_readfirstlane(u.w);
    if (uflags & 8) return;                            // a workgroup that only holds its place in the list (st_place_twins): both its entries say so
    const st_args &a = (uflags & 2) ? a_dn : a_up;     // (either unit of a pair can be the workgroup's first wave)
    const int W = a.W;
#ifdef ST_EXPERIMENTS
    if (a.clk && lane == 0) a.clk[2 * (2 * blockIdx.x + wave)] = __builtin_amdgcn_s_memtime();
#endif
    if (a.prof && lane == 0) {
        a.prof[3 * (2 * blockIdx.x + wave)] = __builtin_amdgcn_s_memrealtime();
        a.prof[3 * (2 * blockIdx.x + wave) + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
    // (at least four steps, whatever the unit's rows: the partner's pipeline fills through this wave's first four)
    const int nsteps = (((r1 - r0) + TB_T + 3) / 4) * 2;
    const int e0 = r0;                                // the window starts on the seam
    int col = tx * TB_OW - TB_T + 4 * lane;
    const int xcol = col;
    col %= W; if (col < 0) col += W;
    int colp = col + W / 2; if (colp >= W) colp -= W;
    const bool lane_ok = (lane >= TB_T / 4) && (lane < 64 - TB_T / 4) && (xcol < W);
    st_state z;
    {
        const int e_last = e0 + 2 * nsteps + 5;
        z.plain = e0 >= 0 && e_last < a.L && a.y_first + e0 >= 0 && a.y_first + e_last < a.H;
        z.nx_p = a.pin + (long)e0 * a.rs; z.nx_b = a.b + (long)e0 * a.rs; z.nx_c = a.code + (long)e0 * a.rs;
    }
    st_fetch(a, z, 0, 0, e0, col, colp);
    st_fetch(a, z, 0, 1, e0 + 1, col, colp);
    st_fetch(a, z, 1, 0, e0 + 2, col, colp);
    st_fetch(a, z, 1, 1, e0 + 3, col, colp);
    unsigned cls_lo, cls_hi = 0u;
    st_unit_tables<DV>(a, lane, tx, e0, 2 * nsteps + 2, cls_lo, ktab);
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int k = 0; k < 2; k++) { z.w[s][k][0] = (v4f)(0.0f); z.w[s][k][1] = (v4f)(0.0f); }
    z.landbits = 0xFFFFFFFFu;
    z.mixbits = 0; z.slot = 0; z.badbits = 0;
    z.fc[0][0] = st_class4(cls_lo, 0); z.fc[0][1] = st_class4(cls_lo, 1); z.fc[1][0] = st_class4(cls_lo, 2); z.fc[1][1] = st_class4(cls_lo, 3);
    {
        const st_kptr cf = (st_kptr)(unsigned long long)a.jcoef;
        z.cy = cf[4]; z.ry = cf[5];
    }
    v4f (*xme)[64] = &bring2[wave][ST_RING - 2];
    const v4f (*xpeer)[64] = &bring2[wave ^ 1][ST_RING - 2];
#if ST_TWIN_FAIR
    // The SIMD's arbiter serves the older of its two waves first: of two equal units the older ends after 37 us, the younger after
    // 44, and runs its last stretch alone at two thirds of the SIMD's throughput.  Here the two take turns at a raised priority
    // (s_setprio), iteration by iteration, so that they end together: the 8-way band 48.2 -> 47.7 us per launch, the 4-way band
    // 80.5 -> 78.3 (same-box A/B of two builds; turns every step: the same; every other iteration: nothing)
    const int wslot = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11))) & 1;
#endif
    // (Round 6, measured and dropped: starting the wave in a SIMD's second slot 1000 .. 6400 clocks late, so that the two are out of
    // phase -- one takes its rows in while the other runs its levels: the 8-way bands 51.5 -> 51.8 / 53.3 / 54.1 us; the delay is just lost.)
    for (int t = 0; t < nsteps; t += 2) {
#if ST_TWIN_FAIR == 4
        if (wslot) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#elif ST_TWIN_FAIR
        if (uflags & 16) __builtin_amdgcn_s_setprio(3);          // an edge unit: ahead of whatever shares its SIMD, all the way
        else if (((t >> ST_TWIN_FAIR) ^ wslot) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
#ifdef ST_EXPERIMENTS
        if (a.stepprof && lane == 0 && t < 44) a.stepprof[(size_t)(2 * blockIdx.x + wave) * 64 + t / 2] = __builtin_amdgcn_s_memrealtime();
#endif
        st_step<0, DV, 1>(a, z, bring, cring, ktab, lane, t, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol, 0, xme, xpeer);
#if ST_TWIN_FAIR == 4
        if (wslot) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
#endif
        st_step<1, DV, 1>(a, z, bring, cring, ktab, lane, t + 1, e0, nsteps, r1, cls_lo, cls_hi, col, colp, lane_ok, xcol, 0, xme, xpeer);
    }
    if (a.prof && lane == 0) a.prof[3 * (2 * blockIdx.x + wave) + 1] = __builtin_amdgcn_s_memrealtime();
#ifdef ST_EXPERIMENTS
    if (a.clk && lane == 0) a.clk[2 * (2 * blockIdx.x + wave) + 1] = __builtin_amdgcn_s_memtime();
    if (a.stepprof && lane == 0) { a.stepprof[(size_t)(2 * blockIdx.x + wave) * 64 + 63] = __builtin_amdgcn_s_memrealtime(); a.stepprof[(size_t)(2 * blockIdx.x + wave) * 64 + 62] = t_entry; }
#endif
}

// (Two more forms of the unit lived here through round 4 and were retired in round 5, when no launch of a BASELINE size had ever
// chosen them: persistent waves with work stealing (k_jacobi_stream_p: 2.79 M against 3.11 M at 16384x8192, profiles/r02_stream_p_*)
// and narrow strips of 2 cells per lane at four waves per SIMD (k_jacobi_stream_n: 2.51 M against 2.52 M, 1.83 against 1.97 M
// without land).  What they measured is in profiles/NOTEBOOK.md; the code is in the history, commit a116cc2.)

// ---- host ---------------------------------------------------------------------------------------------------------
bool ocean_stream_enabled()
{
    static const bool on = [] { const char *e = dmg_hook("DMG_JACOBI_STREAM"); return !e || atoi(e) != 0; }();
    return on;
}

// the row tables of the mirrored view (k_jacobi_stream's southward units): local row l' of that view is row L - 1 - l'
static int ocean_stream_flipped_tables(dmg_ocean *o)
{
    if (o->jcoef_f) return DMG_OK;
    // all four tables into locals first: they are published together or not at all (a half-built set would hand null mirrored
    // tables to every later twin or southward launch)
    const int L = o->L;
    const size_t nj = (size_t)L * 8;
    float *jf_d = nullptr, *j2f_d = nullptr, *j3f_d = nullptr;
    unsigned char *bad_d = nullptr, *cls_d = nullptr;
    auto drop = [&](int rc) { for (void *q : { (void *)jf_d, (void *)j2f_d, (void *)j3f_d, (void *)bad_d, (void *)cls_d }) if (q) (void)hipFree(q); return rc; };
    std::vector<float> j(nj), jf(nj);
    auto flip_j = [&](const float *src, float **dst) -> int {
        DMG_HIP(hipMemcpy(j.data(), src, nj * sizeof(float), hipMemcpyDeviceToHost));
        for (int l = 0; l < L; l++) memcpy(&jf[(size_t)l * 8], &j[(size_t)(L - 1 - l) * 8], 8 * sizeof(float));
        DMG_HIP(hipMalloc((void **)dst, nj * sizeof(float)));
        DMG_HIP(hipMemcpy(*dst, jf.data(), nj * sizeof(float), hipMemcpyHostToDevice));
        return DMG_OK;
    };
    DMG_HIP(hipStreamSynchronize(o->ctx->stream));
    int rc;
    if ((rc = flip_j(o->jcoef, &jf_d)) || (rc = flip_j(o->jcoef2, &j2f_d)) || (rc = flip_j(o->jcoef3, &j3f_d))) return drop(rc);
    {
        std::vector<unsigned char> b((size_t)L, 0), bf((size_t)L);
        if (hipMemcpy(b.data(), o->rowbad, (size_t)L, hipMemcpyDeviceToHost) != hipSuccess) return drop(dmg_fail(DMG_EHIP, "mirrored row tables: download failed"));
        for (int l = 0; l < L; l++) bf[l] = b[L - 1 - l];
        if (hipMalloc((void **)&bad_d, (size_t)L) != hipSuccess || hipMemcpy(bad_d, bf.data(), (size_t)L, hipMemcpyHostToDevice) != hipSuccess)
            return drop(dmg_fail(DMG_ENOMEM, "mirrored row tables: allocation failed"));
    }
    if (!o->rowclass_host.empty()) {
        const size_t n = o->rowclass_host.size(), tiles_x = n / (size_t)L;
        std::vector<unsigned char> cf(n);
        for (size_t tx = 0; tx < tiles_x; tx++)
            for (int l = 0; l < L; l++) cf[tx * L + l] = o->rowclass_host[tx * L + (L - 1 - l)];
        if (hipMalloc((void **)&cls_d, n) != hipSuccess || hipMemcpy(cls_d, cf.data(), n, hipMemcpyHostToDevice) != hipSuccess)
            return drop(dmg_fail(DMG_ENOMEM, "mirrored row tables: allocation failed"));
    }
    o->jcoef2_f = j2f_d; o->jcoef3_f = j3f_d; o->rowbad_f = bad_d; o->rowclass_f = cls_d;
    o->jcoef_f = jf_d;                                   // (the guard above reads this one: last)
    return DMG_OK;
}

// 8 sweeps over local rows [lo, hi) from p[from] into p[from ^ 1]; seg = rows per unit (0: the default)
int ocean_launch_stream(dmg_ocean *o, int lo, int hi, int part, int zone_lo, int zone_hi, int from, hipStream_t st, int seg, int variant)
{
    if (seg <= 0) seg = st_unit_rows();                    // 0: chosen per geometry
    variant = 0;                                         // (1 and 2 were the narrow and the persistent form: retired, round 5)
    if (seg > ST_MAXROWS) seg = ST_MAXROWS;
    const dmg_ocean::tile_list *tl = st_units(o, variant, lo, hi, part, zone_lo, zone_hi, seg);
    if (!tl) {
        if (dmg_hook("DMG_ST_DEBUG")) fprintf(stderr, "[stream] no unit list for rows [%d,%d) part %d: %s\n", lo, hi, part, st_units_failed() ? dmg_last_error() : "allocation failed");
        return st_units_failed() ? DMG_ESTATE : dmg_fail(DMG_ENOMEM, "unit list allocation failed");     // (st_units has said what failed)
    }
    if (tl->n == 0) return DMG_OK;
    st_args a;
    a.pin = o->p[from];
    a.pout = o->p[from ^ 1];
    a.b = o->divw;
    a.code = o->code;
    // the division form: two operations per quotient where every row constant of this band has an exact pair (div2.hip)
    static const bool div2_env = [] { const char *e = dmg_hook("DMG_JACOBI_DIV2"); return !e || atoi(e) != 0; }();
    // the tolerance rendering (DMG_QUIRK_JACOBI_TOLERANCE): fused row coefficients, within 1e-5 of the exact form (st_value2s<3>)
    const bool tol = (o->prm.quirks & DMG_QUIRK_JACOBI_TOLERANCE) != 0;
    const bool div2 = !tol && div2_env && o->div2_ok && o->jcoef2;
    a.jcoef = tol ? o->jcoef3 : div2 ? o->jcoef2 : o->jcoef;
    a.rowbad = div2 && o->div2_nbad ? o->rowbad : nullptr;
    a.jcoef_cy = o->jcoef + 4;                           // row 0's {cy, 1/cy}: the same for every row
    a.W = o->grid.W; a.L = o->L;
    a.rs = o->grid.W; a.flipped = 0;
    a.y_first = o->y0 - o->gs; a.H = o->grid.H;
    a.spole = o->topo.spole ? 1 : 0; a.npole = o->topo.npole ? 1 : 0;
    static const bool use_cls = dmg_hook("DMG_TB_NOCLASS") == nullptr;
    a.rowclass = use_cls ? o->rowclass : nullptr;
    a.units = (const int4 *)tl->d;
    a.nseg0 = 0;
    o->last_kernel = "k_jacobi_stream";
    static const int dbg = [] { const char *e = dmg_hook("DMG_ST_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    a.prof = nullptr;
    a.clk = nullptr;
    a.stepprof = nullptr;
#ifdef ST_EXPERIMENTS
    static const char *stepprof_path = dmg_hook("DMG_ST_STEPPROF");
    static int stepprof_count = 0;
    const bool stepprof_now = stepprof_path && !variant && tl->n_units > 1024 && ++stepprof_count == 40;    // the 40th full-size launch
    if (stepprof_now && (hipMalloc((void **)&a.stepprof, tl->n_units * 64 * 8) != hipSuccess || hipMemset(a.stepprof, 0, tl->n_units * 64 * 8) != hipSuccess)) a.stepprof = nullptr;
#endif
    static const char *prof_path = dmg_hook("DMG_ST_PROF");  // experiments: the first full-size launch writes one line per unit to this file
    static bool prof_done = false;
    static int prof_skip = [] { const char *e = dmg_hook("DMG_ST_PROF_SKIP"); return e ? atoi(e) : 0; }();      // ... the (n+1)-th such launch: a warm one
    const bool prof_now = prof_path && !prof_done && !variant && tl->n_units > 1024 && prof_skip-- <= 0;
    if (prof_now && hipMalloc((void **)&a.prof, tl->n_units * 24) != hipSuccess) a.prof = nullptr;
#ifdef ST_EXPERIMENTS
    if (prof_now && a.prof && (hipMalloc((void **)&a.clk, tl->n_units * 16) != hipSuccess || hipMemset(a.clk, 0, tl->n_units * 16) != hipSuccess)) a.clk = nullptr;
#endif
    static const int extra_lds = [] { const char *e = dmg_hook("DMG_ST_EXTRA_LDS"); return e ? atoi(e) : 0; }();   // experiments: caps the waves per CU
    {
        // the mirrored view of the band for the units that stream southwards: pointers to the last stored row, a negative row
        // stride, the row tables reversed (built once per ocean and terrain), grid rows and poles mirrored
        st_args ad = a;
        if (tl->n_flip || tl->twin) {
            int rc = ocean_stream_flipped_tables(o);
            if (rc) return rc;
            const long last = (long)(o->L - 1) * o->grid.W;
            ad.pin = a.pin + last; ad.pout = a.pout + last; ad.b = a.b + last; ad.code = a.code + last;
            ad.rs = -(long)o->grid.W; ad.flipped = 1;
            ad.jcoef = tol ? o->jcoef3_f : div2 ? o->jcoef2_f : o->jcoef_f;
            ad.rowbad = a.rowbad ? o->rowbad_f : nullptr;
            ad.rowclass = a.rowclass ? o->rowclass_f : nullptr;
            ad.y_first = o->grid.H - a.y_first - o->L;
            ad.spole = a.npole; ad.npole = a.spole;
        }
        if (tl->twin) {
            o->last_kernel = "k_jacobi_stream_twin";
            if (tol) hipLaunchKernelGGL(k_jacobi_stream_twin<3>, dim3((unsigned)tl->n), dim3(128), 0, st, a, ad);
            else if (div2) hipLaunchKernelGGL(k_jacobi_stream_twin<1>, dim3((unsigned)tl->n), dim3(128), 0, st, a, ad);
            else hipLaunchKernelGGL(k_jacobi_stream_twin<0>, dim3((unsigned)tl->n), dim3(128), 0, st, a, ad);
        }
        else if (tol) hipLaunchKernelGGL(k_jacobi_stream<3>, dim3((unsigned)tl->n), dim3(64), (size_t)extra_lds, st, a, ad);
        else if (div2) hipLaunchKernelGGL(k_jacobi_stream<1>, dim3((unsigned)tl->n), dim3(64), (size_t)extra_lds, st, a, ad);
        else hipLaunchKernelGGL(k_jacobi_stream<0>, dim3((unsigned)tl->n), dim3(64), (size_t)extra_lds, st, a, ad);
    }
    DMG_HIP(hipGetLastError());
#ifdef ST_EXPERIMENTS
    if (a.stepprof) {
        std::vector<unsigned long long> tt(tl->n_units * 64);
        DMG_HIP(hipStreamSynchronize(st));
        DMG_HIP(hipMemcpy(tt.data(), a.stepprof, tt.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(a.stepprof);
        if (FILE *f = fopen(stepprof_path, "w")) {
            for (int i = 0; i < (int)tl->n_units; i++) { for (int k = 0; k < 64; k++) fprintf(f, "%llu%c", tt[(size_t)i * 64 + k], k == 63 ? '\n' : ','); }
            fclose(f);
        }
    }
#endif
    if (prof_now && a.prof) {
        prof_done = true;
        const int n_units = (int)tl->n_units;
        std::vector<unsigned long long> tt((size_t)n_units * 3), cc((size_t)n_units * 2, 0ull);
        std::vector<int> uu((size_t)n_units * 4);
        DMG_HIP(hipStreamSynchronize(st));
        DMG_HIP(hipMemcpy(tt.data(), a.prof, tt.size() * 8, hipMemcpyDeviceToHost));
        if (a.clk) { DMG_HIP(hipMemcpy(cc.data(), a.clk, cc.size() * 8, hipMemcpyDeviceToHost)); (void)hipFree(a.clk); }
        DMG_HIP(hipMemcpy(uu.data(), tl->d, uu.size() * 4, hipMemcpyDeviceToHost));
        (void)hipFree(a.prof);
        if (FILE *f = fopen(prof_path, "w")) {
            // (r0, r1: the unit's rows in the band's own numbering; open / mixed / land: the classes of its window's rows; then what its
            // steps are made of, st_unit_features)
            fprintf(f, "tx,r0,r1,open,mixed,land,t0,t1,hwid,xcc,flip,twin,fast,coast,tested,t_open,t_mixed,t_land,f_open,f_mixed,f_land,bad,c0,c1\n");
            for (int i = 0; i < n_units; i++) {
                int cnt[3] = { 0, 0, 0 };
                const unsigned char *cls = o->rowclass_host.empty() ? nullptr : o->rowclass_host.data() + (size_t)uu[4 * i] * o->L;
                const bool flip = (uu[4 * i + 3] & 2) != 0, twin = (uu[4 * i + 3] & 4) != 0;
                const int v0 = uu[4 * i + 1], v1 = uu[4 * i + 2];                   // the unit's rows in its view
                const int b0 = flip ? o->L - v1 : v0, b1 = flip ? o->L - v0 : v1;
                const int w0 = (twin && !flip) ? b0 : b0 - TB_T, w1 = (twin && flip) ? b1 : b1 + TB_T;
                for (int r = w0; r < w1; r++) cnt[(cls && r >= 0 && r < o->L) ? cls[r] : 2]++;
                const st_features ft = st_unit_features(cls, o->L, o->y0 - o->gs, o->grid.H, v0, v1, flip, twin, o->rowbad_host.empty() ? nullptr : o->rowbad_host.data());
                fprintf(f, "%d,%d,%d,%d,%d,%d,%llu,%llu,%llu,%llu,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%llu,%llu\n", uu[4 * i], b0, b1, cnt[0], cnt[2], cnt[1], tt[3 * i], tt[3 * i + 1],
                        tt[3 * i + 2] & 0xFFFFFFFFull, tt[3 * i + 2] >> 32, (int)flip, (int)twin, ft.fast, ft.coast, ft.tested, ft.t_open, ft.t_mixed, ft.t_land,
                        ft.f_open, ft.f_mixed, ft.f_land, ft.bad, cc[2 * i], cc[2 * i + 1]);
            }
            fclose(f);
        }
    }
    return DMG_OK;
}
