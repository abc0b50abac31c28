/* demiurge_hip.h -- C ABI of the MI355X-native compute backend for Demiurge's
 * spherical-grid simulation filters.
 *
 * This is the drop-in boundary.  The reference has no
 * FFI; its simulation tools are C++ classes behind Filter/SubFilter
 * (src/filter/Filter.h:58-130) that reach the GPU through
 *     Project::apply(ShaderProgram*, Texture* target, ...)   src/Project.cpp:284-292
 *     Texture::uploadData / downloadDataRAW / swap           src/Texture.cpp:38-93
 *     Project::getCoords / circumference / getWidth/Height   src/Project.cpp:419-421, src/Project.h:88
 * Each entry point below names the reference interface it replaces.  A
 * maintainer-side adapter (HipOceanCurrents : AsyncSubFilter) is shown in
 * INTEGRATION.md and implemented against a GL-free Project in demiurge_amd/host/.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no exceptions cross the boundary.
 *  - every function returns 0 (DMG_OK) or a negative DMG_E* code;
 *    dmg_last_error() returns the message of the calling thread's last failure.
 *  - the caller owns all host buffers; the library owns device memory.
 *  - a handle may be used by one thread at a time (the reference runs each
 *    filter on one worker thread, src/filter/Filter.cpp:125-143).
 *  - field layout = Texture::downloadDataRAW (src/Texture.cpp:56-67): W*H fp32,
 *    row-major, x (longitude) fastest, row 0 = latitude coords[0].  Velocity is
 *    handed over as two planes vx, vy (the reference exports it that way:
 *    scratch1 := v.x, scratch2 := v.y, src/filter/OceanCurrents.cpp:175-205).
 *  - work is enqueued on the context's HIP stream; calls that return data to the
 *    host synchronise that stream, the others do not (dmg_sync does).
 *
 * Latitude bands (multi-GPU).  A dmg_ocean created with band (rank r of n) owns
 * grid rows [dmg_band_row0(H,n,r), dmg_band_row0(H,n,r+1)) plus `ghost` rows
 * towards each neighbouring band; dmg_ocean_halo() hands out the slabs to
 * exchange.  dmg_multi (below) is the library's own driver of such bands: one
 * process with a stream and a thread per GPU, or one process per GPU, RCCL
 * send/recv over xGMI.  A host that brings its own transport can still drive
 * dmg_ocean bands itself (demiurge_amd/bands.py does, over torch.distributed).
 */
#ifndef DEMIURGE_HIP_H
#define DEMIURGE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMG_VERSION 101            /* 0.1.1: round 6 -- quirks: set bits mean tolerance (DMG_QUIRK_EXACT_LLVMPIPE ignored) */

enum {
    DMG_OK = 0,
    DMG_EINVAL = -1,               /* bad argument (null, size, unsupported geometry) */
    DMG_ENOMEM = -2,               /* host or device allocation failed */
    DMG_EHIP = -3,                 /* a HIP runtime call or kernel launch failed */
    DMG_ENODEVICE = -4,            /* no usable gfx950 device: the backend never falls back to the CPU */
    DMG_ESTATE = -5                /* call order (e.g. step before set_terrain) */
};

typedef struct dmg_ctx dmg_ctx;        /* one per GPU / process */
typedef struct dmg_ocean dmg_ocean;    /* OceanCurrents instance (src/filter/OceanCurrents.h:24-54) */

/* Canvas geometry = Project::getWidth/getHeight/getCoords/circumference. */
typedef struct {
    int W, H;                      /* texels: longitude x latitude */
    float coords[4];               /* lat0, lat1, lon0, lon1 in radians; default {-pi/2, pi/2, -pi, pi} (src/Project.h:97) */
    float circumference;           /* km; default 42000 (src/Project.h:88) */
} dmg_grid;

/* Every literal the reference hard-codes in OceanCurrents.cpp, as a field. */
typedef struct {
    float pressurefactor;          /* 100   OceanCurrentsMenu::pressure (OceanCurrents.h:21), uniform :430,:618 */
    int jacobi_iters;              /* 5000  :50 */
    int diffusion_iters;           /* 50    :701 */
    float timestep_h;              /* 24    :294 */
    float scale;                   /* 420   :419,:478,:574,:689 */
    float wind;                    /* 10    :342 */
    int quirks;                    /* DMG_QUIRK_* bit set; default = reference behaviour */
    int jacobi_kernel;             /* DMG_JACOBI_AUTO (default) | _SINGLE (one sweep per launch) | _BLOCKED (8 sweeps
                                      per launch, temporally blocked on chip); results are bit-identical.  The 50
                                      velocity-diffusion sweeps follow it: blocked where the blocked Jacobi kernel
                                      applies, one sweep per launch with _SINGLE */
} dmg_ocean_params;
enum { DMG_JACOBI_AUTO = 0, DMG_JACOBI_SINGLE = 1, DMG_JACOBI_BLOCKED = 2 };

/* the reference's `float bC = texture(v,st).r` adds v.x to BOTH components in velocity diffusion (:687,:693) */
#define DMG_QUIRK_DIFFUSION_SCALAR_RHS 1
/* opt-in, Jacobi sweeps of a single band only: evaluate the shader per fragment the way Mesa llvmpipe does, including its
 * rasteriser's `st` (an ulp of 1.0 off (x+0.5)/W on grids that are not powers of two) and float-weight GL_LINEAR taps --
 * the software GL path's pixels on e.g. the reference's default 1000x500 canvas.  Identical to the default kernels on
 * power-of-two grids; slow (not a performance path). */
#define DMG_QUIRK_LLVMPIPE_RASTER 2
/* The default (no tolerance bit set) reproduces the reference's pixels -- as Mesa llvmpipe renders its GLSL -- bit for bit in every
 * pass.  north_star asks for 1e-5 relative; the bits cost the trig-heavy passes and the pressure solve most of their time.  The
 * tolerance renderings are opt-in, one bit each (round 6, ADVICE r5: SET means tolerance, so every quirks value an older caller
 * holds keeps its meaning):
 * DMG_QUIRK_FAST_TRANSCENDENTALS: advect and divergence keep the reference's expressions -- Mesa's arcsine and arctangent
 *   polynomials included, which are approximations of their own, and gallivm's sine / cosine (the hardware's are good to 1e-6
 *   absolute: 2.6e-3 cells of longitude at 16384 wide) -- but take log2, exp2, reciprocals and square roots from the hardware
 *   (1 ulp): within 1e-5 of the exact rendering after a whole outer step (tests/test_gpu_ocean_passes.py).  Diffusion and subDiv
 *   (whose coastline snapping turns on comparisons of angles: no tolerance there) are the same in both renderings.
 * DMG_QUIRK_JACOBI_TOLERANCE: the pressure sweep ((pW + pE)/pw2.x + (pS + pN)/pw2.y - bC)/Beta (OceanCurrents.cpp:480) as
 *   a (pW + pE) + b (pS + pN) - c bC with a = 1/(pw2.x Beta), b = 1/(pw2.y Beta), c = 1/Beta formed in fp64 from the reference's own
 *   fp32 row constants; a = ah + al, b = bh + bl with ah + bh = 0.5 exactly and the residues folded into one term, so that the gain
 *   of a smooth field carries no systematic error through the 5000 sweeps: three fused multiply-adds per cell instead of three
 *   divisions.  2.5e-6 of the field's maximum from the exact rendering after 5000 sweeps (1.8e-6 from the llvmpipe golden at
 *   256x128; tools/jacobi_tolerance_eval.py), bit-equal to the oracle's form 3.  Applies to the launches of 8 sweeps (the
 *   streaming kernel); the last 1..7 sweeps of a count that is no multiple of 8 run in the exact arithmetic.
 * DMG_QUIRK_EXACT_LLVMPIPE (4) was round 5's "exact" bit, set by default then; it is accepted and ignored. */
#define DMG_QUIRK_EXACT_LLVMPIPE 4
#define DMG_QUIRK_FAST_TRANSCENDENTALS 8
#define DMG_QUIRK_JACOBI_TOLERANCE 16
#define DMG_QUIRK_TOLERANCE (DMG_QUIRK_FAST_TRANSCENDENTALS | DMG_QUIRK_JACOBI_TOLERANCE)

/* field ids for dmg_ocean_halo */
enum { DMG_F_PRESSURE = 0, DMG_F_DIVW = 1, DMG_F_TERRAIN = 2, DMG_F_V = 3 /* interleaved (vx,vy), 8 B per cell */,
       DMG_F_VADVECT = 4 /* the deeper copy of v that dmg_ocean_advect_prepare sets up for one advect */ };
enum { DMG_SIDE_SOUTH = 0, DMG_SIDE_NORTH = 1 };   /* south = towards row 0 */

/* ---- library / context ------------------------------------------------- */
int dmg_version(void);
const char *dmg_last_error(void);
/* Optional, and only useful BEFORE the process's first HIP call (in a host that links or loads the library early): asks the HIP
 * runtime for eight hardware queues (setenv GPU_MAX_HW_QUEUES=8 unless the host has set the variable), which dmg_multi's
 * overlapped exchange order needs (two streams per band; with HIP's default of four they serialise).  A host that initialises HIP
 * first sets the variable itself.  The library never edits the environment on its own.  Returns DMG_OK, or DMG_ESTATE when the
 * variable is already set to less than 8. */
int dmg_runtime_init(void);
/* The library's behaviour does not depend on its host's environment, with two documented exceptions: DMG_MULTI_TIMEOUT_MS (how long
 * a band waits without progress before it gives its neighbours up, dmg_multi_destroy below) and DMG_ST_TWIN=0 (plain instead of twin
 * units for the band launches of the pressure solve: the switch a first run between two devices can fall back on).  Everything a
 * test or an experiment needs to force -- failure and delay injection (DMG_MULTI_FAIL, DMG_ST_TWIN_FAIL, DMG_MULTI_FAKE_LINK_US ...),
 * a kernel form (DMG_TB_SHAPE, DMG_TEMPERATURE_LITERAL ...), the unit cut's constants, traces -- goes through this call: a
 * process-wide table of name -> value that the library consults where it used to read the environment.  Names start with DMG_;
 * value NULL removes a name.  Most hooks are read once, when the code that looks at them first runs: set them before the first
 * handle is created.  The Python test binding (demiurge_amd/capi.py) forwards the DMG_* variables of ITS environment here when it
 * loads the library; the C++ adapters never call it. */
int dmg_debug_set(const char *name, const char *value);

/* Binds HIP device `device_id`.  Work is enqueued on `stream`: an existing hipStream_t of the host program
 * (NULL = the device's default stream, which is what e.g. torch.cuda.current_stream() is unless changed), or,
 * with DMG_PRIVATE_STREAM in `flags`, a new non-blocking stream owned by the context (`stream` ignored). */
#define DMG_PRIVATE_STREAM 1
int dmg_create(int device_id, void *stream, int flags, dmg_ctx **out);
/* Handles created on a context (dmg_ocean, dmg_tectonics, dmg_climate, dmg_erosion, dmg_multi) keep it alive:
 * dmg_destroy while such handles exist only marks the context, and the last handle's destroy releases it.  Destroying
 * in either order is therefore safe; a caller-provided stream must outlive the last handle. */
void dmg_destroy(dmg_ctx *ctx);
int dmg_sync(dmg_ctx *ctx);
/* Temporaries of finished calls stay with the context for reuse (a 1000-round erosion run would otherwise spend a
 * sixth of its time in hipMalloc/hipFree); dmg_trim returns them to the device, dmg_destroy does so too. */
int dmg_trim(dmg_ctx *ctx);                                   /* hipStreamSynchronize */
int dmg_device_name(dmg_ctx *ctx, char *buf, size_t len);

/* hipMemcpyAsync(device -> device) on the context's stream: the single-process halo transport used when
 * several bands live on one GPU (tests, SURVEY.md section 4 "fake cluster"); multi-GPU runs use RCCL. */
int dmg_copy_device(dmg_ctx *ctx, void *dst, const void *src, size_t bytes);

/* hipEvent timing on the context's stream (the stream the kernels run on). */
int dmg_timer_start(dmg_ctx *ctx);
int dmg_timer_stop(dmg_ctx *ctx, float *elapsed_ms);          /* records, synchronises, returns ms since start */

/* ---- latitude bands ------------------------------------------------------ */
int dmg_band_row0(int H, int band_count, int band_rank);      /* first row of a band; band_rank==band_count gives H */

/* ---- ocean currents  (replaces OceanCurrents, src/filter/OceanCurrents.cpp) ---- */
void dmg_ocean_params_default(dmg_ocean_params *p);

/* OceanCurrents::OceanCurrents (:23-103): allocates v, pressure, scratch, divw for this band; v := 0 (:92,:102).
 * ghost = rows of halo kept towards each neighbouring band (>= 1; also the largest sweep block
 * dmg_ocean_jacobi_sweeps accepts between halo refreshes).  band_count == 1: whole grid on this GPU. */
int dmg_ocean_create(dmg_ctx *ctx, const dmg_grid *grid, const dmg_ocean_params *params,
                     int band_rank, int band_count, int ghost, dmg_ocean **out);
void dmg_ocean_destroy(dmg_ocean *o);

/* Texture::downloadDataRAW of the terrain layer -> device (only the sign matters: img > 0 is land,
 * :284,:467,:536).  `img` is the FULL W*H field; the band (plus ghosts) is sliced out here. */
int dmg_ocean_set_terrain(dmg_ocean *o, const float *img);
/* optional: start from a given velocity instead of 0; full-grid planes */
int dmg_ocean_set_velocity(dmg_ocean *o, const float *vx, const float *vy);
/* right-hand side of the pressure solve (what OceanCurrents::divergence(p, divw) leaves in `divw`); full grid */
int dmg_ocean_set_divw(dmg_ocean *o, const float *divw);
/* starting pressure (the reference always starts from 0, :495-497); full grid */
int dmg_ocean_set_pressure(dmg_ocean *o, const float *pressure);

/* OceanCurrents::jacobi (:437-509) on one GPU (band_count == 1): pressure := 0 (:495-497) when
 * zero_first != 0, then `iters` sweeps of the Jacobi shader (:450-481) with the ping-pong of :500-508.
 * Asynchronous on the context's stream.
 * Arithmetic: the shader's three divisions by row constants are evaluated through the constant's correctly rounded
 * reciprocal with one FMA correction, which IS the IEEE quotient for 0 and for 2^-100 <= |numerator| <= 2^100.
 * Outside that window (|x| < 7.9e-31, where llvmpipe itself flushes to zero below 1.2e-38, or a diverged solve) the
 * last bit may differ, and an infinite iterate becomes NaN one sweep earlier than with a true division.
 * Zeros: land holds +0.0 and the streaming kernel evaluates "a land neighbour reads the centre value" (:470-473) as
 * (pW + pE) + n * pC with n land neighbours among the two.  That is the shader's sum bit for bit, except that a pair of
 * open neighbours that both hold -0.0, in a 256-column strip within ten rows of a coast, sums to +0.0 instead of -0.0 --
 * a value no solve produces from the +0.0 start inside the window above (a starting pressure given through
 * dmg_ocean_set_pressure may hold it).
 * Since round 3 the streaming kernel forms these quotients in two operations where that is exact: for a divisor c with
 * 1/c = h + l it computes fma(x, h, x * l), after every one of the 2^23 significands of x has been checked on the device to
 * give the correctly rounded x / c for that (c, h, l) (dmg_div2_split; rows whose divisors have no such pair keep the
 * three-operation form).  Same window, same results: the IEEE quotient. */
int dmg_ocean_jacobi(dmg_ocean *o, int iters, int zero_first);
/* The pair (h, l) the streaming Jacobi kernel uses to divide by the row constant c (pixelwidth2 and Beta of
 * OceanCurrents.cpp:478-480): x / c == fma(x, h, x * l) for every finite x whose products neither underflow nor
 * overflow, verified exhaustively on the device.  DMG_EINVAL if c has no such pair (about one constant in a thousand). */
int dmg_div2_split(dmg_ctx *ctx, float c, float *h, float *l);
/* Band form: `n` (<= ghost) sweeps assuming the pressure ghost rows are current; afterwards the owned
 * rows are n sweeps ahead and the ghost rows are stale until the host refreshes them via dmg_ocean_halo. */
int dmg_ocean_jacobi_sweeps(dmg_ocean *o, int n);
int dmg_ocean_zero_pressure(dmg_ocean *o);
/* The same block of n sweeps in two calls, so that the host can exchange the NEXT block's ghost rows while this
 * block's last launch is still running: _begin runs all but the interior tiles of the last launch and makes the new
 * iterate current (dmg_ocean_halo then returns its slabs: start the sends/receives), _end sweeps the interior tiles,
 * which touch neither the rows being sent nor the ghost rows being received.  Bit-identical to
 * dmg_ocean_jacobi_sweeps(o, n); with one band or without the blocked kernel _begin does everything.  Between the two
 * calls only dmg_ocean_halo (and velocity / terrain accessors) may be used on this ocean. */
int dmg_ocean_jacobi_sweeps_begin(dmg_ocean *o, int n);
int dmg_ocean_jacobi_sweeps_end(dmg_ocean *o);
/* Pieces for a host that schedules the launches of a block of sweeps itself, e.g. to run the tile rows next to the
 * band edges on a second stream ahead of the rest so that the halo exchange hides behind the interior sweeps
 * (demiurge_amd/bands.py, jacobi_solve_pipelined).  All results stay bit-identical to dmg_ocean_jacobi_sweeps.
 *   _geometry         rows [lo, hi) of the launch that ends k sweeps after the last halo refresh; output rows per tile
 *                     row of the tile shape chosen for it (0: still being tuned, or no blocked kernel -- do not split)
 *   _band_rows        stored rows and ghost rows of this band
 *   _jacobi_launch    n (1..8) sweeps over [lo, hi) from pressure buffer src (0/1) into the other one on `stream`
 *                     (hipStream_t, NULL = the context's); part 0 all tile rows, 1 those producing rows < zone_lo or
 *                     >= zone_hi, 2 the others; touches neither the current-buffer index nor the counters
 *   _pressure_buffer / _set_pressure_buffer   which buffer is current; make one current after a self-scheduled block
 *   _pressure_halo    dmg_ocean_halo's pressure slabs of a given buffer */
int dmg_ocean_jacobi_geometry(dmg_ocean *o, int k, int *lo, int *hi, int *tile_rows);
int dmg_ocean_band_rows(dmg_ocean *o, int *stored_rows, int *ghost_south, int *ghost_north);
int dmg_ocean_jacobi_launch(dmg_ocean *o, int lo, int hi, int n, int part, int zone_lo, int zone_hi, int src, void *stream);
int dmg_ocean_pressure_buffer(dmg_ocean *o, int *current);
int dmg_ocean_set_pressure_buffer(dmg_ocean *o, int idx, int launches, int sweeps);
int dmg_ocean_pressure_halo(dmg_ocean *o, int buffer, int side, int rows, void **send, void **recv, size_t *bytes);
/* cumulative counters since creation: Jacobi kernel launches and sweeps performed (a temporally blocked launch
 * advances up to 8 sweeps) -- what bench.py divides its HIP-event time by */
int dmg_ocean_counters(dmg_ocean *o, long long *jacobi_launches, long long *jacobi_sweeps);
/* name of the kernel the most recent Jacobi launch ran (k_jacobi_stream, k_jacobi_tb8, k_jacobi_sweep_v4 ...): the row to
 * look for in a rocprofv3 --kernel-trace --stats table */
int dmg_ocean_jacobi_kernel_name(dmg_ocean *o, char *buf, size_t len);

/* One exchange slab of a field: `send` = this band's `rows` rows next to `side`, `recv` = the ghost rows
 * beyond that side; both contiguous device memory of `bytes` bytes.  bytes == 0 when the side is a grid edge. */
int dmg_ocean_halo(dmg_ocean *o, int field, int side, int rows, void **send, void **recv, size_t *bytes);

/* OceanCurrents::run body (:155-209) n_outer times: advect -> diffusion -> divergence -> jacobi -> subDiv
 * (the WIP `resize(600,200)` :207 and `while(true)` :212 are not replicated).  band_count == 1 only; with bands the
 * host interleaves the passes below with halo refreshes (demiurge_amd/bands.py ocean_step shows the order). */
int dmg_ocean_step(dmg_ocean *o, int n_outer);
/* The individual passes, each leaving its result in the field the reference leaves it in.  With bands a pass
 * computes every local row whose inputs are valid, assuming the host refreshed the input field's ghost rows:
 * advect/subDiv -> owned rows; diffusion -> iters <= ghost sweeps with the region shrinking a row per sweep;
 * divergence -> all local rows but the outermost ghost row (so divw is valid ghost-1 rows deep for the Jacobi). */
int dmg_ocean_advect(dmg_ocean *o);                            /* :235-360  v <- advect(v) */
int dmg_ocean_diffusion(dmg_ocean *o, int iters);              /* :625-708  v <- diffuse(v) */
int dmg_ocean_divergence(dmg_ocean *o);                        /* :363-435  divw <- div(v) */
int dmg_ocean_subdiv(dmg_ocean *o);                            /* :511-622  v <- v - grad p, coastline snapping */
/* rows of velocity halo this band's next advect needs (back-trace distance |v|*timestep, :298-299) */
int dmg_ocean_advect_reach(dmg_ocean *o, int *rows);
/* When that exceeds `ghost` (at H = 8192 the reference's own wind forcing, :341-348, drives |v| to ~10: 96 rows): the next
 * dmg_ocean_advect gathers from a copy of v `reach` rows deep.  _prepare allocates it (grown on demand, in steps of 32 rows)
 * and copies the owned rows in; the host then refreshes its ghost rows through dmg_ocean_halo(o, DMG_F_VADVECT, side, reach,
 * ...) and calls dmg_ocean_advect.  reach may not exceed the rows a band owns.  dmg_multi_step does all of this itself. */
int dmg_ocean_advect_prepare(dmg_ocean *o, int reach);

/* Results back to the host (Texture::uploadData side of the adapter): owned rows only are written,
 * into FULL-grid arrays at their grid position (so ranks can fill one array); NULL skips a field. */
int dmg_ocean_get(dmg_ocean *o, float *vx, float *vy, float *pressure, float *divw);

/* Checkpoint / resume (the reference has none; its solve runs 10000 outer steps, :155): the band's owned rows of
 * v, pressure and divw as NumPy files <prefix>.gN.v.npy (rows,W,2), <prefix>.gN.pressure.npy, <prefix>.gN.divw.npy of a
 * generation N, and a sidecar <prefix>.meta that names the generation a restore reads.  A checkpoint writes generation
 * N + 1 beside N, replaces the sidecar with one rename and then removes N: a crash at any point leaves a complete set.
 * A restore that fails (missing or truncated file, another canvas) leaves the handle unchanged. */
int dmg_ocean_checkpoint(dmg_ocean *o, const char *prefix);
int dmg_ocean_restore(dmg_ocean *o, const char *prefix);      /* after dmg_ocean_set_terrain, same grid and band */

/* ---- one ocean over several GPUs: latitude bands driven by the library -----------------------------
 * Replaces the same interface as dmg_ocean (OceanCurrents behind AsyncSubFilter::run, src/filter/Filter.cpp:125-166): the
 * bands, one stream and one host thread per band, every halo exchange (RCCL ncclSend/ncclRecv of the slabs that
 * dmg_ocean_halo hands out, or peer copies) and the all-reduce of the advect reach live below this boundary.  Results
 * are bit-identical to a single dmg_ocean for every split and transport.  Full-grid host arrays in, full-grid arrays out. */
typedef struct dmg_multi dmg_multi;
/* flags of dmg_multi_create / dmg_multi_create_rank: one transport, optionally the exchange order */
#define DMG_MULTI_AUTO 0            /* RCCL when every band has its own GPU, else peer copies */
#define DMG_MULTI_RCCL 1            /* grouped ncclSend/ncclRecv on each band's stream (ncclCommInitAll / ncclCommInitRank) */
#define DMG_MULTI_PEER 2            /* hipMemcpyPeerAsync + events: one process; also several bands on one GPU */
#define DMG_MULTI_RCCL_LOOPBACK 3   /* tests on a one-GPU box: all bands on one device and stream, every slab sent through a
                                       communicator of size 1 to the process's own rank -- the RCCL calls run for real */
#define DMG_MULTI_STORE 4           /* direct stores: the sender's copy kernel writes its edge rows straight into a landing slot in the
                                       neighbour's memory (peer-mapped pointer in one process, hipIpc mapping between processes) and
                                       then a sequence number into the neighbour's flag word; the receiver's stream waits for the number
                                       and moves the slot into its ghost rows.  No RCCL kernel, host barrier or event per exchange; RCCL
                                       is kept for set-up (the handles' all-gather) and the advect reach all-reduce */
#define DMG_MULTI_TRANSPORT_MASK 15
#define DMG_MULTI_EXCHANGE_THEN_SWEEP 0     /* order of a Jacobi block: refresh the ghost rows, then sweep (default) */
#define DMG_MULTI_OVERLAP_LAST_LAUNCH 16    /* the exchange for the next block runs while the interior of this block's last
                                               launch is swept: that launch's edge units (the rows the neighbours need, cut
                                               into short units of their own) go first, alone, on the band's second,
                                               high-priority stream, the exchange behind them; the interior units start on
                                               the band's own stream when the edge units are done, beside the exchange, and
                                               the own stream waits for the exchange in front of the next block only.
                                               Pays ~28 us per block for the edge units and hides ~57 us of exchange on a
                                               1024-row band: ahead of exchange-then-sweep from ~35 us of link time on.
                                               Needs GPU_MAX_HW_QUEUES >= 8 in the environment before the process's first HIP
                                               call (dmg_runtime_init); a handle that does not find it falls back to
                                               exchange-then-sweep and reports DMG_MULTI_WARN_HWQ_FALLBACK in its stats */
#define DMG_MULTI_ID_BYTES 128
#define DMG_MULTI_STORE_BLOB_BYTES 192
#define DMG_MULTI_WARN_HWQ_FALLBACK 1   /* dmg_multi_stats.warnings: the overlapped order was asked for with fewer than 8 hardware queues */
#define DMG_MULTI_WARN_SHARED_GPU_FALLBACK 2   /* ... or with more than four bands of one process on one GPU (a rehearsal): 9+ streams
                                                  of one device that wait on each other are time-sliced by its scheduler */
/* One process, n_bands bands; device_ids[i] = HIP device of band i (ids may repeat with DMG_MULTI_PEER / _LOOPBACK).
 * ghost = halo rows per band side: the Jacobi solve exchanges every `ghost` sweeps (rounded down to whole 8-sweep
 * launches).  An advect that reaches beyond `ghost` rows gathers from a deeper copy of v that dmg_multi_step exchanges for that
 * step (dmg_ocean_advect_prepare). */
int dmg_multi_create(int n_bands, const int *device_ids, const dmg_grid *grid, const dmg_ocean_params *params, int ghost, int flags,
                     dmg_multi **out);
/* One process per GPU: rank 0 calls dmg_multi_unique_id, the host hands the 128 bytes to every rank (torch.distributed,
 * MPI, a file), each rank creates its band.  Collective: every rank must call it. */
int dmg_multi_unique_id(void *id128);
int dmg_multi_create_rank(int rank, int n_ranks, const void *id128, int device_id, const dmg_grid *grid, const dmg_ocean_params *params,
                          int ghost, int flags, dmg_multi **out);
/* DMG_MULTI_STORE with one process per GPU: with an id the ranks exchange their memory handles over RCCL inside
 * dmg_multi_create_rank and nothing more is needed.  A host with a bootstrap of its own passes id128 == NULL (no RCCL
 * communicator: dmg_multi_jacobi works, dmg_multi_step -- which all-reduces the advect reach -- does not), then has every
 * rank call _blob, gathers the DMG_MULTI_STORE_BLOB_BYTES bytes of all ranks in rank order, and hands them to _connect. */
int dmg_multi_store_blob(dmg_multi *m, void *blob);
int dmg_multi_store_connect(dmg_multi *m, const void *blobs, int n_ranks);
/* A band that fails with a runtime error marks the handle, which then refuses everything but dmg_multi_destroy (DMG_ESTATE);
 * once the band threads have returned the local RCCL communicators are aborted (queued receives end instead of waiting for a
 * peer that will never send).  Host-side waits of the RCCL and store transports give up, with DMG_EHIP, when no exchange of
 * their band has completed for DMG_MULTI_TIMEOUT_MS (environment, default 120000): a long queue that is progressing is not a
 * failure.  The store transport's landing kernels give up after the same time by themselves. */
void dmg_multi_destroy(dmg_multi *m);
int dmg_multi_set_order(dmg_multi *m, int order);          /* DMG_MULTI_EXCHANGE_THEN_SWEEP | DMG_MULTI_OVERLAP_LAST_LAUNCH, per call site */
/* the dmg_ocean_* calls of the same names, on every local band (full-grid arrays; each band takes / fills its rows) */
int dmg_multi_set_terrain(dmg_multi *m, const float *img);
int dmg_multi_generate_terrain(dmg_multi *m, const struct dmg_noise_params *params);
int dmg_multi_set_divw(dmg_multi *m, const float *divw);
int dmg_multi_set_pressure(dmg_multi *m, const float *pressure);
int dmg_multi_set_velocity(dmg_multi *m, const float *vx, const float *vy);
int dmg_multi_get(dmg_multi *m, float *vx, float *vy, float *pressure, float *divw);
int dmg_multi_sync(dmg_multi *m);
/* OceanCurrents::jacobi (:437-509) over the bands: pressure := 0 when zero_first, then `iters` sweeps, ghost rows refreshed
 * every block.  Asynchronous on the bands' streams (dmg_multi_sync / dmg_multi_get wait) -- up to a point: a band's host thread
 * runs at most 128 halo exchanges ahead of its GPU (the progress ring the failure detection reads), so a call that queues more
 * than that -- this one from ~4000 sweeps at ghost 32, dmg_multi_climate_steps from 128 x ghost steps, dmg_multi_step always --
 * returns when all but its last 128 exchanges have completed.  A host that overlaps other work with these calls gives them
 * a thread. */
int dmg_multi_jacobi(dmg_multi *m, int iters, int zero_first);
/* OceanCurrents::run body (:155-209) n_outer times over the bands.  Before each advect the bands all-reduce the longest
 * back-trace (|v| * timestep, :298-299): if it exceeds `ghost` every rank returns DMG_EINVAL with its state untouched. */
int dmg_multi_step(dmg_multi *m, int n_outer);
int dmg_multi_exchange(dmg_multi *m, int field, int rows);  /* one halo refresh of DMG_F_* on every band */
int dmg_multi_band_count(const dmg_multi *m);               /* bands in this process */
int dmg_multi_band(dmg_multi *m, int local_index, dmg_ocean **out);   /* borrow a band (checkpoints, dmg_ocean_speed_shade ...) */
typedef struct {
    int bands_total, bands_local, transport, order, ghost;
    int warnings;                    /* DMG_MULTI_WARN_* */
    long long exchanges;             /* halo exchanges since creation */
    double exchange_ms;              /* their time on band 0's stream (HIP events around each exchange) */
    long long exchange_bytes;        /* bytes band 0 sent + received */
    long long jacobi_launches, jacobi_sweeps;     /* of band 0 */
} dmg_multi_stats;
int dmg_multi_get_stats(dmg_multi *m, dmg_multi_stats *s);  /* synchronises */
/* *ranks = the size RCCL reports for the communicator the halo slabs travel on (ncclCommCount; the band count where the
 * transport has no communicator), *devices (may be NULL) = distinct HIP devices among this process's bands: what a
 * measurement states beside its "n_gpus" (the loop it spans is OceanCurrents.cpp:500-508, one rank per latitude band) */
int dmg_multi_comm_ranks(dmg_multi *m, int *ranks, int *devices);
/* The climate year cycle (Temperature::run, src/filter/Temperature.cpp:47-162: 500 000 steps) over latitude bands on
 * several GPUs: the bands are dmg_climate_create_band handles, everything else -- one thread and two streams per band,
 * the RCCL / peer / loopback transports, failure handling -- is dmg_multi's.  Flags and `ghost` as for dmg_multi_create;
 * the ghost rows are refreshed every `ghost` steps.  T and terrain are FULL-grid arrays; results are bit-identical to
 * dmg_climate_steps on one band. */
typedef struct dmg_multi_climate dmg_multi_climate;
struct dmg_climate_params;                       /* below, with dmg_temperature_steps */
int dmg_multi_climate_create(int n_bands, const int *device_ids, const dmg_grid *grid, const float *T, const float *terrain, int ghost, int flags,
                             dmg_multi_climate **out);
int dmg_multi_climate_create_rank(int rank, int n_ranks, const void *id128, int device_id, const dmg_grid *grid, const float *T,
                                  const float *terrain, int ghost, int flags, dmg_multi_climate **out);
void dmg_multi_climate_destroy(dmg_multi_climate *h);
int dmg_multi_climate_steps(dmg_multi_climate *h, int first_step, int n_steps);   /* asynchronous on the bands' streams */
int dmg_multi_climate_set_terrain(dmg_multi_climate *h, const float *terrain);    /* Temperature.cpp:131-132: only the sign matters */
int dmg_multi_climate_set_params(dmg_multi_climate *h, const struct dmg_climate_params *params);   /* dmg_climate_set_params on every local band */
int dmg_multi_climate_get(dmg_multi_climate *h, float *T);                        /* owned rows of this process's bands; synchronises */
int dmg_multi_climate_sync(dmg_multi_climate *h);
int dmg_multi_climate_store_blob(dmg_multi_climate *h, void *blob);                          /* as dmg_multi_store_blob / _connect */
int dmg_multi_climate_store_connect(dmg_multi_climate *h, const void *blobs, int n_ranks);
int dmg_multi_timer_start(dmg_multi *m);                    /* HIP events on band 0's stream; _stop waits for every band */
int dmg_multi_timer_stop(dmg_multi *m, float *elapsed_ms);

/* ---- generic stencil filters on a whole grid ---------------------------- */
/* Blur::Blur + Blur::step until finished (src/filter/BlurMenu.cpp:24-117), in place on host data */
int dmg_blur(dmg_ctx *ctx, const dmg_grid *grid, float *field_inout, float radius);
/* Every literal the reference hard-codes in Temperature.cpp, as a field.  The defaults reproduce the
 * shader bit for bit; the arithmetic keeps the shader's form whatever the values.  Why a host wants them: with the reference's
 * values the explicit scheme is unstable on fine grids -- the diffusion number diffusivity * (year_seconds / steps_per_year) /
 * (heat_capacity * dy^2) must stay below 1/4, and at 16384x8192 (dy = 2.56 km) it is 11.8 over land: fp32 overflows within 38
 * steps.  More steps per year, or a smaller diffusivity, give BASELINE config 5 a year it can finish. */
typedef struct dmg_climate_params {
    int steps_per_year;            /* 15000    :133 (the time step, 3.154e7/15000 s) and :146 (the orbital phase, M = 2 pi i / 15000) */
    float year_seconds;            /* 3.154e7  :133 (written 3.154*1e7) */
    float diffusivity;             /* 0.55e6   :128 (written 0.55*1e6) */
    float albedo;                  /* 0.30     :102 (the terrain-dependent 0.15 / 0.06 of :101 is overwritten there) */
    float olr_coeff;               /* 210      :125  OLR = olr_coeff * ((T + celsius_offset) / olr_ref_kelvin)^4 * emissivity */
    float olr_ref_kelvin;          /* 273.4    :125 */
    float emissivity;              /* 0.93     :125 */
    float celsius_offset;          /* 273.15   :125 */
    float heat_capacity_land;      /* 1.5e7    :131-132  atmosphere + atmosphere*0.5, atmosphere = 1e7 */
    float heat_capacity_ocean;     /* 7e7      :131-132  atmosphere + 4*1.5*atmosphere */
    float eccentricity;            /* 0.017    :59 */
    float obliquity_deg;           /* 23.44    :60 (gamma) */
    float perihelion_deg;          /* 77.05    :62 (omega2) */
    float solar_constant;          /* 1365     :65 (S0) */
    float initial_temperature;     /* 0        Temperature::Temperature clears the layer (:26-44): the field a handle starts from when it is
                                               created with T == NULL */
    int quirks;                    /* DMG_CLIMATE_QUIRK_* bit set; default = reference behaviour */
} dmg_climate_params;
/* h0 (:85-89) takes acos(-tan(phi)*tan(delta)) whenever |phi| <= pi/2 - |delta|; on the row where the two sides are within an ulp
 * the product rounds to 1 + 2^-23, GLSL's acos is undefined there and llvmpipe returns NaN: once or twice a year a whole row of the
 * field turns NaN and stays NaN (the Laplacian's isnan guard keeps it from spreading; its neighbours lose their diffusion term
 * instead).  Set: the same (default).  Cleared: the argument is clamped to [-1, 1] first and the field stays finite. */
#define DMG_CLIMATE_QUIRK_ACOS_NAN 1
void dmg_climate_params_default(dmg_climate_params *p);
/* Temperature::run inner pass (src/filter/Temperature.cpp:140-150) n_steps times from step index first_step;
 * params NULL = the reference's values */
int dmg_temperature_steps(dmg_ctx *ctx, const dmg_grid *grid, float *T_inout, const float *terrain,
                          const dmg_climate_params *params, int first_step, int n_steps);
/* The same with the field resident on the device between calls: Temperature::run (:47-162) steps the field ten times
 * per UI frame (:140) for as long as the tool runs, so a frame costs its kernels, not two PCIe crossings.
 * T and terrain are copied at creation. */
typedef struct dmg_climate dmg_climate;
int dmg_climate_create(dmg_ctx *ctx, const dmg_grid *grid, const float *T, const float *terrain, dmg_climate **out);
void dmg_climate_destroy(dmg_climate *k);
/* New values for the steps that follow (a handle starts with the reference's).  The divisors among them get their exact
 * two-operation pairs here (one short device check per new constant); DMG_EINVAL for values the scheme cannot use (steps_per_year
 * < 1, a heat capacity or olr_ref_kelvin that is not a positive finite number).  T == NULL at creation is allowed since this
 * exists: the field then starts at the default initial_temperature (0), or at the one set here before the first step. */
int dmg_climate_set_params(dmg_climate *k, const dmg_climate_params *params);
int dmg_climate_steps(dmg_climate *k, int first_step, int n_steps);   /* asynchronous on the context's stream */
int dmg_climate_get(dmg_climate *k, float *T_out);                    /* synchronises */
/* The same field as one latitude band of a multi-GPU run (the laplacian reads 1 row): rows of band
 * `band_rank` of `band_count` plus `ghost` rows per side; T and terrain are FULL-grid arrays, the band is sliced out.
 * dmg_climate_steps then takes n_steps <= ghost between halo refreshes (the computed region shrinks a row per step) and
 * dmg_climate_get writes the owned rows into a FULL-grid array; dmg_climate_halo = dmg_ocean_halo for the temperature. */
int dmg_climate_create_band(dmg_ctx *ctx, const dmg_grid *grid, const float *T, const float *terrain, int band_rank, int band_count,
                            int ghost, dmg_climate **out);
int dmg_climate_halo(dmg_climate *k, int side, int rows, void **send, void **recv, size_t *bytes);
int dmg_climate_set_terrain(dmg_climate *k, const float *terrain);   /* new heights (FULL-grid array) under the same field */
/* erosion/uplift pass of cpufilter::run (src/filter/cpufilter.cpp:106-214) */
int dmg_erosion_pass(dmg_ctx *ctx, const dmg_grid *grid, float *h_inout, const float *flow, const float *updrift,
                     float factor, float slope_exponent);

/* FlowFilter::run (src/filter/FlowFilter.cpp:719-786): pre-blur, stochastic-D8 directions, catchments, lake graph,
 * upstream area.  flow_out[i] = pow(area, exponent), -1 where no river mouth is reached (what the reference uploads
 * into the terrain texture, :714-716).  sel = selection mask or NULL (all ones, src/Project.cpp:102-104).
 * lakes = bit set: DMG_FLOW_LAKES = FlowFilter's lakeflag (:660-699), cells at or below their lake's water level are
 * set to 0; DMG_FLOW_FIX_MOUTH_FILTER = do NOT replicate `if (Nthbit(c.from,10)) continue;` (:552), which tests bit 10
 * of the neighbouring lake's cell index and so drops about half of a river-mouth lake's first candidates.
 * Host arithmetic follows the reference's overload resolution: area = double cos, flow = double pow, rounded once. */
#define DMG_FLOW_LAKES 1
#define DMG_FLOW_FIX_MOUTH_FILTER 2
int dmg_flow_accumulate(dmg_ctx *ctx, const dmg_grid *grid, const float *h, const float *sel_or_null, float preblur,
                        float exponent, int lakes, float *flow_out);
/* The two GLSL passes of FlowFilter::findMagicNumbers (:104-320) alone, on h pre-blurred by `preblur` (0 = none):
 * codes_out = flow direction per cell in keypad numbering (1..9, 5 = sink, 0 = sea or unselected, :149-262),
 * mask_out = 0.5 + bitmask of the neighbours that drain into the cell, +16 sink, +512 river mouth, -1 off the land
 * (:268-315); both as floats, like the reference's textures. */
int dmg_flow_passes(dmg_ctx *ctx, const dmg_grid *grid, const float *h, const float *sel_or_null, float preblur,
                    float *codes_out, float *mask_out);
/* cpufilter::run (src/filter/cpufilter.cpp:41-222): uplift = max(h,0)/N, heights scaled by 1/N on land, then
 * `iters` (reference: N*3 = 150) rounds of { flow accumulation (preblur 0.5, `exponent`) ; erosion/uplift pass }.
 * h_inout receives the scaled heights the reference leaves in the terrain texture. */
int dmg_erosion_run(dmg_ctx *ctx, const dmg_grid *grid, float *h_inout, int N, int iters, float exponent,
                    float slope_exponent, float factor);
/* The same round by round, heights resident on the device (the editor shows the terrain after every round and reports
 * progress, cpufilter.cpp:93-104).  `lakes` = the menu's toggle handed to FlowFilter (:95, dolakes > 0).
 * dmg_erosion_get returns the scaled heights of the current round. */
typedef struct dmg_erosion dmg_erosion;
int dmg_erosion_create(dmg_ctx *ctx, const dmg_grid *grid, const float *h, int N, float exponent, float slope_exponent,
                       float factor, int lakes, dmg_erosion **out);
void dmg_erosion_destroy(dmg_erosion *e);
int dmg_erosion_rounds(dmg_erosion *e, int n_rounds);
int dmg_erosion_get(dmg_erosion *e, float *h_out);

/* ---- latitude-band forms of the generic filters ---------------------------------------------------------
 * dmg_field = one scalar field (a Texture, src/Texture.h) of one band, resident on the device with `ghost` rows towards each
 * neighbouring band: the layout and the halo slabs of a dmg_ocean band.  A pass computes the band's OWNED rows from fields
 * whose ghost rows are current as deep as the pass reads; the host refreshes them in between (dmg_field_halo slabs over RCCL,
 * or dmg_fields_exchange for bands that live in one process).  Bit-identical to the whole-grid calls above. */
typedef struct dmg_field dmg_field;
int dmg_field_create(dmg_ctx *ctx, const dmg_grid *grid, int band_rank, int band_count, int ghost, dmg_field **out);
void dmg_field_destroy(dmg_field *f);
int dmg_field_set(dmg_field *f, const float *full);           /* band + ghost rows out of a FULL-grid array */
int dmg_field_get(dmg_field *f, float *full);                 /* owned rows into a FULL-grid array */
int dmg_field_halo(dmg_field *f, int side, int rows, void **send, void **recv, size_t *bytes);
int dmg_fields_exchange(dmg_field *const *bands, int n, int rows);   /* bands[i] = band i of n, all in this process: peer copies */
/* Blur (src/filter/BlurMenu.cpp:24-117): dmg_blur_schedule returns Blur::Blur's radius list (count, or < 0); Blur::step is,
 * per radius r in order, dmg_field_blur_pass(f, r, 1) then (f, r, 0); a pass reads dmg_field_blur_halo(r, vertical) ghost rows */
int dmg_blur_schedule(float radius, float *r_out, int cap);
int dmg_field_blur_halo(float r, int vertical);
int dmg_field_blur_pass(dmg_field *f, float r, int vertical);
/* erosion/uplift pass (cpufilter.cpp:106-214): h reads 1 ghost row; flow / updrift at the cell */
int dmg_field_erosion_pass(dmg_field *h, const dmg_field *flow, const dmg_field *updrift, float factor, float slope_exponent);
/* one of ThermalErosion::step's ten rounds (ThermalErosion.cpp:25-142): h reads 2 ghost rows; `loss` = scratch of the same band */
int dmg_field_thermal_round(dmg_field *h, dmg_field *loss);

/* ---- SURVEY.md section 8(f) "next" rows: adjacent Filter-menu tools on the same stencil skeleton ---- */
/* ThermalErosion::step (src/filter/ThermalErosion.cpp:25-142): 10 rounds of talus loss + gain, in place */
int dmg_thermal_erosion(dmg_ctx *ctx, const dmg_grid *grid, float *h_inout);
/* Morphological (src/filter/Morphological.cpp:28-85): 64-tap ring erode (op 0) / dilate (op 1), radius schedule :31-43 */
int dmg_morphological(dmg_ctx *ctx, const dmg_grid *grid, float *field_inout, float radius, int op);
/* MorphologicalGradient (:87-131): dilate((radius+1)/2) - erode(radius/2) */
int dmg_morphological_gradient(dmg_ctx *ctx, const dmg_grid *grid, float *field_inout, float radius);

/* ---- field export, SURVEY.md 8(f-4) ---- */
/* Project::file_write (src/Project.cpp:56-67): 8-bit greyscale PNG, pixel = (unsigned char)(int)(255*field[i]), rows in
 * memory order.  clamp != 0 limits the field to [0,1] first (the reference lets the byte wrap). */
int dmg_field_write_png(const char *path, int W, int H, const float *field, int clamp);
/* the ocean-current display shading (src/appearance/VectorField.cpp:117-118): 1 - 8*|v| on ocean cells, land_value on
 * land; owned rows written into a FULL-grid array like dmg_ocean_get */
int dmg_ocean_speed_shade(dmg_ocean *o, float land_value, float *out_full);

/* ---- gradient-noise terrain generator: GradientNoiseFilter (src/filter/GradientNoise.cpp), SURVEY.md 8(f-3) ---- */
enum { DMG_NOISE_DEFAULT = 0, DMG_NOISE_RIDGED, DMG_NOISE_BILLOWY, DMG_NOISE_IQ, DMG_NOISE_SWISS, DMG_NOISE_JORDAN,
       DMG_NOISE_PLATEAUS };                       /* enum noise_mode, GradientNoise.h:10 */
enum { DMG_BLEND_REPLACE = 0, DMG_BLEND_ADD, DMG_BLEND_SUBTRACT, DMG_BLEND_MULTIPLY, DMG_BLEND_DIVIDE, DMG_BLEND_MAX,
       DMG_BLEND_MIN };                            /* filter::blendMode(), src/filter/Filter.cpp:170-236 */
struct dmg_noise_params;
typedef struct dmg_noise_params {                  /* struct NoiseParams, GradientNoise.h:15-25, + the blend mode */
    float scale, lacunarity, persistence, min, max;
    int octaves, seed;
    float warp;
    int mode, blend;
} dmg_noise_params;
/* GradientNoiseMenu::params (GradientNoise.h:39): {1.0, 2, 0.5, -10, 10, 8, 0, 0.0, DEFAULT}, blend Replace */
void dmg_noise_params_default(dmg_noise_params *p);
/* GradientNoise.cpp:489-491: std::mt19937{seed} + uniform_real_distribution<float>(0, 10000), three draws */
int dmg_noise_seed_offset(int seed, float out3[3]);
/* GradientNoiseFilter::step (:61-500): field := blend_mode(field, noise(st), sel); sel NULL = all ones */
int dmg_gradient_noise(dmg_ctx *ctx, const dmg_grid *grid, float *field_inout, const float *sel_or_null,
                       const dmg_noise_params *params);
/* the same generator writing straight into an ocean handle's terrain on the device (each band generates its own
 * rows; nothing crosses PCIe), followed by what dmg_ocean_set_terrain does after its upload.  blend must be Replace. */
int dmg_ocean_generate_terrain(dmg_ocean *o, const dmg_noise_params *params);

/* ---- plate tectonics: Tectonics + Plate (src/filter/tectonics/Tectonics.cpp, Plate.cpp), SURVEY.md 8(a-11) ----
 * State: one RGBA texture (height, age, 0, 0) per plate in the plate's own rotating frame, and a W x W global map
 * (plate index, height, spreading flag, overlap marker).  The reference builds two plates from the terrain, split at
 * st.x = 0.5, turning about the x axis at +-0.01 rad/step (Tectonics.cpp:15-58). */
typedef struct dmg_tectonics dmg_tectonics;
/* Tectonics::Tectonics (:15-154): terrain = W*H heights (Texture::downloadDataRAW of Project::get_terrain) */
int dmg_tectonics_create(dmg_ctx *ctx, const dmg_grid *grid, const float *terrain, dmg_tectonics **out);
void dmg_tectonics_destroy(dmg_tectonics *t);
/* Plate::updateRotationBy (Plate.cpp:46-48): angularVelocity += theta * axis */
int dmg_tectonics_update_rotation(dmg_tectonics *t, int plate, float theta, const float axis[3]);
/* Tectonics::run (:156-275), n_steps iterations (reference: 70) = { Plate::rotate ; fold ; oceanSpreading ;
 * back-to-plates }, and the render pass (:186-210) after the last one.  Asynchronous on the context's stream. */
int dmg_tectonics_step(dmg_tectonics *t, int n_steps);
/* the passes of one iteration, for callers that interleave their own work (and for the parity tests):
 *   fold    Plate::rotate (Plate.cpp:26-28) for every plate, then Tectonics::fold (:278-293)
 *   spread  Tectonics::oceanSpreading (:295-397)
 *   render  :186-210 -- what collision() (:399-618) leaves for it is plateIndices.x (:590)
 *   unfold  back to plates (:216-268), including the c <-> plate Texture::swap that moves plate storage between
 *           W x H and W x W (period 3) */
int dmg_tectonics_fold(dmg_tectonics *t);
int dmg_tectonics_spread(dmg_tectonics *t);
int dmg_tectonics_render(dmg_tectonics *t);
int dmg_tectonics_unfold(dmg_tectonics *t);
/* results: terrain (W*H floats, what the render pass wrote into Project::get_terrain), the global map (W*W RGBA),
 * plate textures (w*h RGBA with the size dmg_tectonics_plate_size reports).  These synchronise the stream. */
int dmg_tectonics_get_terrain(dmg_tectonics *t, float *terrain);
int dmg_tectonics_get_map(dmg_tectonics *t, float *rgba);
int dmg_tectonics_plate_count(const dmg_tectonics *t);
int dmg_tectonics_plate_size(const dmg_tectonics *t, int plate, int *w, int *h);
int dmg_tectonics_get_plate(dmg_tectonics *t, int plate, float *rgba);

/* ---- BASELINE.json config 5: tectonic uplift + stream-power erosion + year-cycle climate on one terrain ----------------
 * The reference applies its tools one after the other to the shared terrain layer; this handle is that sequence behind the
 * C ABI.  One dmg_world_step = Tectonics::run steps (tectonics/Tectonics.cpp:156-275) -> cpufilter::run rounds on the
 * rendered terrain (cpufilter.cpp:41-222) -> Temperature::run steps over the eroded land/sea mask, the orbital phase
 * advancing through the year from step to step (Temperature.cpp:140-150).  One-way coupling, as in the editor. */
typedef struct dmg_world dmg_world;
typedef struct {
    int tectonics_steps, erosion_rounds, climate_steps;      /* per dmg_world_step; 0 skips a stage */
    int erosion_N;                                           /* cpufilter's N = 50 (:42) */
    float exponent, slope_exponent, factor;                  /* the erosion menu's 0.5, 1, 1 (cpufilter.h:18-21) */
    int lakes;
    dmg_climate_params climate;                              /* the climate stage's literals (defaults: the reference's) */
} dmg_world_params;
void dmg_world_params_default(dmg_world_params *p);
int dmg_world_create(dmg_ctx *ctx, const dmg_grid *grid, const float *terrain, const float *T0, dmg_world **out);
void dmg_world_destroy(dmg_world *w);
int dmg_world_step(dmg_world *w, const dmg_world_params *params_or_null);
/* terrain = what the tectonics render pass left; eroded = after the erosion rounds (scaled by 1/N like the reference's
 * terrain texture); T = the climate field.  NULL skips. */
int dmg_world_get(dmg_world *w, float *terrain, float *eroded, float *T);
/* The same on N GPUs (BASELINE config 5): the stage that shards -- the climate year cycle, Temperature.cpp:140-150,
 * 500 000 steps -- runs in latitude bands through dmg_multi_climate (flags and ghost as for dmg_multi_climate_create); the global
 * stages (Tectonics' rotated gathers, FlowFilter's lake graph: "replicas only") run once on the first band's GPU, every band
 * pulling its rows of the eroded terrain from there with one peer copy per iteration (_create: one process), or as a replica on
 * every rank's own GPU (_create_rank: one process per GPU; nothing to broadcast).  Bit-identical to dmg_world.  _step is
 * asynchronous on the bands' streams; _get synchronises (T: the owned rows of this process's bands, into the full-grid array). */
typedef struct dmg_multi_world dmg_multi_world;
int dmg_multi_world_create(int n_bands, const int *device_ids, const dmg_grid *grid, const float *terrain, const float *T0, int ghost, int flags,
                           dmg_multi_world **out);
int dmg_multi_world_create_rank(int rank, int n_ranks, const void *id128, int device_id, const dmg_grid *grid, const float *terrain,
                                const float *T0, int ghost, int flags, dmg_multi_world **out);
void dmg_multi_world_destroy(dmg_multi_world *w);
int dmg_multi_world_step(dmg_multi_world *w, const dmg_world_params *params_or_null);
int dmg_multi_world_get(dmg_multi_world *w, float *terrain, float *eroded, float *T);

/* Test hook: the unit list the streaming pressure-solve kernel (ocean_jacobi_stream.hip) would be launched with for local rows
 * [lo, hi) of a band (grid rows [y0, y1) with gs / gn ghost rows), computed on the host alone from a row-class table
 * (rowclass[strip][local row]: 0 open water, 1 land everywhere, 2 mixed; strips of 240 columns) for a device of num_cu compute
 * units.  part 0: the whole launch; 1 / 2: the edge / interior half of a split launch with send zones [lo, zone_lo) and
 * [zone_hi, hi).  out: 4 ints per entry {strip, first row, end row, flags} as the kernel reads them (flags: 1 open-water unit,
 * 2 rows are the mirrored view's, 4 twin unit, 8 place holder, 16 an edge unit of a split launch: it keeps a raised priority); *twin: two entries per workgroup.  No device is touched: the
 * cut, the halves of a split launch and the placement are host code, covered by tests/test_stream_cut.py on the CPU. */
int dmg_stream_units_probe(int W, int H, int y0, int y1, int gs, int gn, const unsigned char *rowclass, int num_cu,
                           int lo, int hi, int part, int zone_lo, int zone_hi, int *out, size_t cap, size_t *n_entries, int *twin);

#ifdef __cplusplus
}
#endif
#endif /* DEMIURGE_HIP_H */
