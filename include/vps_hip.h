/* vps_hip.h -- C ABI of libvps_hip.so: the MI355X (gfx950) hot path
 *   particles -> grid (NGP deposit | exact-NN resample) -> 3-D R2C FFT -> |f(k)|^2
 *   -> spherical-shell binning
 * that replaces the CPU path of YujieH3/large-velocity-power-spectrum behind the
 * reference's own Python function surface.
 *
 * The reference has NO FFI of its own for this path (it is pure numpy calling
 * pyFFTW / pyann / Annoy): every entry point below cites the
 * reference Python function (file:line under /root/reference) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns 0 on success or a
 *     negative vps_status; nothing throws.  vps_last_error() gives the message.
 *   - "dev" pointers are device (HBM) addresses owned by the CALLER (e.g.
 *     torch.Tensor.data_ptr() or vps_malloc); "host" pointers are ordinary memory.
 *   - All work is enqueued on the context's stream (vps_set_stream; default: the
 *     null stream); only vps_sync, vps_timing_get, vps_memcpy_d2h block.
 *   - Grids are float32, C order.  A multi-channel grid is channel-major (SoA):
 *     grid[c][x][y][z].  Slab decomposition is along x: a rank owns
 *     x in [x0, x0+nx).  N is the global cells-per-axis.
 *   - Calls on one context are not re-entrant.
 */
#ifndef VPS_HIP_H
#define VPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vps_ctx vps_ctx;

enum vps_status {
  VPS_OK = 0,
  VPS_ERR_ARG = -1,        /* bad argument (message says which)                 */
  VPS_ERR_HIP = -2,        /* a HIP runtime call failed                          */
  VPS_ERR_UNSUPPORTED = -3,/* e.g. N not a supported FFT length                  */
  VPS_ERR_NOMEM = -4
};

/* quantities of BoxField.spctrm (interp.py:573-583) */
enum vps_quantity { VPS_VELOCITY = 0, VPS_MOMENTUM = 1, VPS_ENERGY = 2,
                    VPS_VM = 3 /* BoxField form: v in channels 0..2, mass in channel 3 */ };
/* flags for vps_field_algebra */
#define VPS_FLAG_REFERENCE_MOMENTUM_BUG 1 /* py=pz=vx*mass as interp.py:523-525 */
#define VPS_FLAG_INPUT_IS_VM 2             /* channels already hold vx,vy,vz,mass (a BoxField) */
#define VPS_FLAG_REUSE_SORT 4              /* vps_deposit_fft_zy: work_dev still holds the bucketed records of the
                                              previous call with the SAME particles, N, Lbox, x0, nx -- skip the sort
                                              (several quantities of one snapshot) */
#define VPS_FLAG_SHARE_ENERGY 8            /* vps_deposit_fft_zy / vps_deposit_fft_z[_slab], momentum and kinetic energy of ONE snapshot:
                                              on the VPS_MOMENTUM call (all three components, no VPS_FLAG_REFERENCE_MOMENTUM_BUG) the
                                              launch ALSO leaves the z image of the energy field E = mass |v|^2 -- made of the same cell
                                              totals of rho v_c -- as a FOURTH component (workspace sized by
                                              vps_deposit_fft_zy_workspace_bytes_shared; a zimg_dev of four components); on the following
                                              VPS_ENERGY call (with VPS_FLAG_REUSE_SORT: same particles, N, Lbox, x0, nx, same
                                              workspace / zimg_dev) nothing is deposited: vps_deposit_fft_zy runs the y pass of that
                                              component, vps_deposit_fft_z[_slab] returns at once (the caller reads component 3).
                                              Same numbers as the energy launch of its own up to the order of float32 additions. */

#define VPS_FLAG_COMPONENTS(mask) (((mask) & 7) << 4) /* vps_deposit_fft_zy / vps_deposit_fft_z[_slab], velocity or momentum: produce
                                              ONLY the components in `mask` (bit c = component c; 0 = all three), in ascending
                                              order at the start of the output -- the scalar images / half spectra of the
                                              three-component call (up to the order of the LDS float adds).  For hosts that deal
                                              the scalar fields of a step out over several GPUs (whole grids, no exchange). */
#define VPS_FLAG_COMPONENT(c) VPS_FLAG_COMPONENTS(1 << (c))
#define VPS_FLAG_COMPONENT_MASK 0x70

/* ---- lifecycle ---------------------------------------------------------- */
int vps_create(vps_ctx** out, int device_id);
int vps_destroy(vps_ctx* ctx);
const char* vps_last_error(const vps_ctx* ctx);   /* ctx may be NULL: global slot */
int vps_set_stream(vps_ctx* ctx, void* hip_stream);
int vps_sync(vps_ctx* ctx);
#define VPS_ABI_VERSION 6
int vps_version(void);                            /* ABI version (VPS_ABI_VERSION)  */
/* Tuning / test switches, process-wide.  The library never reads the environment: a stray variable in a user's job cannot
 * change a code path; the host sets what it wants explicitly (vpower/_ffi.py maps VPS_OPT_<NAME> variables once, at load,
 * and lists them in _ffi.OPTIONS).  Names: no_fast_binning, no_pair_binning, nn_query_centric, nn_column, nn_build_atomic, nn_kappa, nn_stats,
 * sort_groups, sort_staged, sort_atomic (all result-preserving), nn_ablate (timing-only builds; ignored by the product
 * build).  A NaN value restores the default.  Unknown names: VPS_ERR_ARG.
 * Further names: no_int_binning (1: float64 shells even where integer shells are exact), x_wg_per_cu (persistent x-pass workgroups
 * per CU, tuning), comm_fail_send (error-path tests: the n-th ncclSend fails without being issued).
 * Options that shape a WORKSPACE LAYOUT -- nn_build_atomic, sort_atomic, sort_staged, sort_groups -- must not change between a
 * *_workspace_bytes call and the run that uses the workspace sized with it (nor before a VPS_FLAG_REUSE_SORT call): the run
 * entry points take no size and recompute the layout under the current values. */
int vps_set_option(const char* name, double value);
double vps_get_option(const char* name, double dflt);
/* device facts for the host side: out[0]=CUs, out[1]=LDS bytes/CU, out[2]=wave size,
 * out[3]=HBM bytes total (MiB) */
int vps_device_info(vps_ctx* ctx, int64_t out[4]);
/* how the binning x pass will decide shells for the tables of the last vps_set_binning, under the current options:
 * 0 = general monotone shell walk; 1 = mirrored kx, float64 k^2 sums against float64 thresholds; 2 = mirrored kx, INTEGER
 * ix^2 + iy^2 + iz^2 against integer thresholds (exactly the same shells: chosen only where vps_set_binning has checked
 * that no threshold lies on an integer multiple of k2[1]); negative: no tables set */
int vps_binning_mode(vps_ctx* ctx);

/* ---- memory helpers (so the library is usable without torch) ------------ */
int vps_malloc(vps_ctx* ctx, void** dev, size_t bytes);
int vps_free(vps_ctx* ctx, void* dev);
int vps_memset(vps_ctx* ctx, void* dev, int value, size_t bytes);
int vps_memcpy_h2d(vps_ctx* ctx, void* dev, const void* host, size_t bytes);
int vps_memcpy_d2h(vps_ctx* ctx, void* host, const void* dev, size_t bytes); /* blocks */

/* ---- per-kernel timing (HIP events on the context's stream) -------------- */
/* When enabled every kernel launched by the library is bracketed by events.
 * vps_timing_get synchronises and returns, for kernel family `kind`, the launch
 * count and the summed duration in milliseconds since vps_timing_reset. */
enum vps_kernel_kind {
  VPS_K_DEPOSIT = 0, VPS_K_ALGEBRA = 1, VPS_K_FFT_Z = 2, VPS_K_FFT_Y = 3,
  VPS_K_FFT_X = 4, VPS_K_NN_BUILD = 5, VPS_K_NN_QUERY = 6, VPS_K_MISC = 7,
  /* exchange inside the library (vps_spectrum_zimages): one interval per kz chunk each --
   * EXCHANGE: the grouped ncclSend / ncclRecv of the chunk, on the communication stream;
   * EXCHANGE_WAIT: how long the context's stream stood still for it ahead of the chunk's x pass (the EXPOSED part) */
  VPS_K_EXCHANGE = 8, VPS_K_EXCHANGE_WAIT = 9,
  VPS_K_COUNT = 10
};
int vps_timing_enable(vps_ctx* ctx, int on);
int vps_timing_reset(vps_ctx* ctx);
int vps_timing_get(vps_ctx* ctx, int kind, int64_t* launches, double* total_ms);
/* per-launch durations (ms) of kind `kind` in launch order; *n_out = how many exist */
int vps_timing_list(vps_ctx* ctx, int kind, double* ms_out, int64_t cap, int64_t* n_out);

/* ---- stage A0: particle preprocessing ----------------------------------- */
/* In place on the device: coords[:,a] -= min(coords[:,a]) (scripts/parallel_optimized.py:280-282,
 * vpower/interp.py:169-175; exact in the dtype of pos) and v[:,a] -= sum(m v[:,a])/sum(m)
 * (script:285-289, interp.py:178-182; mean accumulated in float64, subtracted as float32).
 * min_out_host[3] / bulk_out_host[3] (may be NULL) receive what was subtracted.  Blocks. */
int vps_preprocess(vps_ctx* ctx, void* pos_dev, int pos_is_f64, float* vel_dev,
                   const float* mass_dev, int64_t np, int shift_to_origin,
                   int remove_bulk_velocity, double* min_out_host, double* bulk_out_host);

/* ---- conservation diagnostics -------------------------------------------- */
/* out_host[5] = sum m, sum m vx, sum m vy, sum m vz, sum m (vx^2+vy^2+vz^2), accumulated in float64:
 * the totals behind total_mass / total_momentum / total_kinetic_energy (x 0.5 on the host) of
 * GasParticles (vpower/interp.py:424-450) and BoxField (interp.py:639-666), i.e. check_conservation
 * (interp.py:1269-1319).  Component c of element i is v_dev[i*v_elem_stride + c*v_comp_stride]:
 * particles (vel [np][3]): strides 3, 1; a gridded field (chans [4][ncell] = vx,vy,vz,mass): strides 1, ncell
 * with mass_dev = chans + 3*ncell.  Blocks. */
int vps_totals(vps_ctx* ctx, const float* v_dev, int64_t v_elem_stride, int64_t v_comp_stride,
               const float* mass_dev, int64_t n, double out_host[5]);

/* ---- stage A1: nearest-grid-point deposition ---------------------------- */
/* Replaces deposit_to_grid (vpower/interp.py:996-1015).
 * Cell index per axis = int((pos // Lcell) % N) evaluated with numpy's
 * floor_divide/remainder algorithm in the dtype of pos (bit exact).
 * pos_dev: [np][3] float32 (pos_is_f64=0) or float64 (1).                        */
int vps_cell_index(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, int64_t np,
                   int N, double Lbox, int32_t* cell_dev /* [np][3] */);
/* Two-level bucket deposition (count -> scan -> scatter into per-brick buckets -> one
 * workgroup per brick accumulates in LDS and streams the tile out).
 * payload_dev: [np][C] float32, C in {1,3,4}.  grid_dev: [C][nx][N][N] float32 is
 * OVERWRITTEN: every cell of the slab is written exactly once (no memset needed);
 * particles whose x cell is outside [x0,x0+nx) are skipped.
 * work_dev: vps_deposit_workspace_bytes(np, C, N, nx) bytes of scratch.              */
size_t vps_deposit_workspace_bytes(int64_t np, int C, int N, int nx);
int vps_deposit_ngp(vps_ctx* ctx, const void* pos_dev, int pos_is_f64,
                    const float* payload_dev, int64_t np, int C,
                    int N, double Lbox, int x0, int nx, float* grid_dev, void* work_dev);
/* The fused hot-path form: deposits density_velocity_vector (interp.py:199-213) built on
 * the fly from vel_dev [np][3] and rho_dev [np], and applies the field algebra of
 * vps_field_algebra in the brick epilogue.  fields_dev: [ncomp][nx][N][N] float32,
 * ncomp = 3 (VPS_VELOCITY, VPS_MOMENTUM), 1 (VPS_ENERGY) or 4 (VPS_VM: vx,vy,vz,mass).
 * work_dev: vps_deposit_workspace_bytes(np, 4, N, nx).                              */
int vps_deposit_field(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                      const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx,
                      int quantity, int flags, float* fields_dev, void* work_dev);

/* The shortest form of stages A1 + A3 + the z and y passes of B (VPS_VELOCITY, VPS_MOMENTUM,
 * VPS_ENERGY; N in [64, 4096]): particle records are bucketed by "pencil" (one z-pass tile:
 * x, 16 y-lines, all z), each pencil is accumulated in LDS, turned into v = rho v / rho (or p, or
 * E = m |v|^2) and transformed along z without the real-space grid ever touching HBM; then the
 * y pass.  Outputs, per component c (3 for velocity / momentum, 1 for energy), the arrays
 * vps_fft_zy produces:
 *   spec_dev [ncomp][N/2][N][nx] complex64,  nyq_dev [ncomp][N][nx] complex64.
 * work_dev: vps_deposit_fft_zy_workspace_bytes(np, N, nx).                          */
int vps_deposit_fft_zy_supported(vps_ctx* ctx, int N, int quantity);
size_t vps_deposit_fft_zy_workspace_bytes_shared(int64_t np, int N, int nx);   /* ... with room for the fourth z image of VPS_FLAG_SHARE_ENERGY */
size_t vps_deposit_fft_zy_workspace_bytes(int64_t np, int N, int nx);
int vps_deposit_fft_zy(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                       const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx,
                       int quantity, int flags, void* spec_dev, void* nyq_dev, void* work_dev);

/* Higher-order assignment by expansion (extension): every particle becomes S = order^3 weighted
 * sub-particles at the centres of the cells it touches (order 2: cloud-in-cell, 3: triangular-shaped cloud; periodic):
 * pos_out_dev [np*S][3] float32, payload_out_dev [np*S][C] = payload * weight (C <= 4).  Depositing them with
 * vps_deposit_ngp gives the CIC / TSC grid; with replicated particles every slab gets its contributions without a halo. */
int vps_assign_expand(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* payload_dev, int64_t np,
                      int C, int N, double Lbox, int order, float* pos_out_dev, float* payload_out_dev);

/* out_dev[np][4] = [vx*rho, vy*rho, vz*rho, rho]: GasParticles.density_velocity_vector
 * (vpower/interp.py:199-213).  vel_dev [np][3], rho_dev [np], float32.             */
int vps_density_velocity_vector(vps_ctx* ctx, const float* vel_dev, const float* rho_dev,
                                int64_t np, float* out_dev);

/* ---- stage A2: exact nearest-neighbour resampling ------------------------ */
/* Replaces ann_interpolate (vpower/interp.py:1018-1049; pyann k=1, eps=0) and the
 * per-cell Annoy query loop (scripts/parallel_optimized.py:337-358), as EXACT 1-NN:
 * squared distance ((qx-px)^2+(qy-py)^2)+(qz-pz)^2 in float64, lowest particle
 * index on exact ties.  Query lattice = qx_host[i] x qy_host[j] x qz_host[k]
 * (float64 host arrays, so both reference lattices are expressible: interp.py:1063
 * and parallel_optimized.py:343-345).  Only x rows [x0,x0+nx) of the lattice are
 * produced.  out_dev: [C][nx][nqy][nqz] float32 = payload of the NN;
 * nn_idx_dev (may be NULL): [nx][nqy][nqz] int32.
 * work_dev: vps_nn_workspace_bytes(np, pos_is_f64, nx*nqy*nqz) bytes of scratch (cell list, sorted
 * records, and the list of lattice points the scatter pass leaves to the exact fallback search).
 * Uniformly spaced axes (both reference lattices) take the particle-centric scatter search; any other
 * axes the query-centric ring search -- same results.                                             */
size_t vps_nn_workspace_bytes(int64_t np, int pos_is_f64, int64_t nq_slab);
int vps_nn_resample(vps_ctx* ctx, const void* pos_dev, int pos_is_f64,
                    const float* payload_dev, int64_t np, int C,
                    const double* qx_host, int nqx, const double* qy_host, int nqy,
                    const double* qz_host, int nqz, int x0, int nx,
                    float* out_dev, int32_t* nn_idx_dev, void* work_dev);

/* The same search with the algebra of GasParticles.ann_interp_to_field (vpower/interp.py:272-273) in its epilogue:
 * rhov_dev [np][4] = density_velocity_vector; out_dev [4][nx][nqy][nqz] = vx, vy, vz (= rho v / rho of the nearest
 * particle) and mass (= rho * Lcell^3): the BoxField form, without a separate pass over the grid. */
int vps_nn_resample_field(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* rhov_dev,
                          int64_t np, const double* qx_host, int nqx, const double* qy_host, int nqy,
                          const double* qz_host, int nqz, int x0, int nx, double Lcell,
                          float* out_dev, int32_t* nn_idx_dev, void* work_dev);

/* ... and with the algebra of BoxField.spctrm's quantity as well (interp.py:501-557): out_dev [ncomp][nx][nqy][nqz] holds what
 * the spectrum of `quantity` transforms -- VPS_VELOCITY: v (3 channels); VPS_MOMENTUM: p = v * mass (3; with
 * VPS_FLAG_REFERENCE_MOMENTUM_BUG py = pz = px, interp.py:523-525); VPS_ENERGY: E = mass |v|^2 (1); VPS_VM: v and mass (4, =
 * vps_nn_resample_field) -- v = rho v / rho and mass = rho Lcell^3 of the nearest particle, rounded as the reference rounds them.
 * For `ann_interp_to_field(N).spctrm(q)`: no fourth channel is written and the z pass reads one array per component. */
int vps_nn_resample_quantity(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* rhov_dev,
                             int64_t np, const double* qx_host, int nqx, const double* qy_host, int nqy,
                             const double* qz_host, int nqz, int x0, int nx, double Lcell, int quantity, int flags,
                             float* out_dev, int32_t* nn_idx_dev, void* work_dev);

/* ---- stage A3: field algebra ------------------------------------------- */
/* chans_dev: [4][ncell] = rho*vx, rho*vy, rho*vz, rho (density_velocity_vector,
 * interp.py:199-213, after deposit/resample).  In place:
 *   VPS_VELOCITY: chans[0..2] = rho v / rho, 0 where rho==0 (interp.py:272, NaN->0
 *                 as interp.py:329-331);
 *   VPS_MOMENTUM: chans[0..2] = v * (rho*Lcell^3)            (interp.py:273,523-525);
 *   VPS_ENERGY  : chans[0]    = (rho*Lcell^3)*(vx^2+vy^2+vz^2) (interp.py:546);
 *   VPS_VM      : chans[0..2] = v, chans[3] = rho*Lcell^3   (BoxField, interp.py:272-275).
 * With VPS_FLAG_INPUT_IS_VM the channels are taken as vx,vy,vz,mass instead.        */
int vps_field_algebra(vps_ctx* ctx, int quantity, int flags, double Lcell,
                      float* chans_dev, int64_t ncell);
/* Out-of-place form: chans_dev is only read, the result's channels (3: velocity / momentum, 1: energy,
 * 4: VM) go to out_dev[c][ncell]; out_dev == NULL is the in-place call above. */
int vps_field_algebra_out(vps_ctx* ctx, int quantity, int flags, double Lcell, const float* chans_dev,
                          int64_t ncell, float* out_dev);

/* ---- stage B+C: 3-D R2C FFT, |f|^2, shell binning ------------------------ */
/* Supported N: powers of two, 16 <= N <= 4096; 96, 192, 384, 768, 1536 (radix 3 / 6 / 12 / 24);
 * 250, 500, 1000, 2000 (radix 5 / 10 / 20).                                          */
int vps_fft_supported(int N);
/* Binning tables (host pointers, copied):
 *   k2_axis[N]   = fl(k*k) for k = 2*pi*fftfreq(N, Lcell)   (interp.py:1449,1460)
 *   thr[nbins+1] : thr[i] = smallest double t with fl(sqrt(t)) >= edge[i] (i<nbins),
 *                  thr[nbins] = smallest t with fl(sqrt(t)) > edge[nbins]; so that
 *                  thr[i] <= s < thr[i+1]  <=>  edge[i] <= sqrt(s) < edge[i+1] with the
 *                  last bin right-closed, i.e. numpy.histogram's rule
 *                  (interp.py:1474-1477, parallel_optimized.py:181-182).
 *   edge0, inv_spacing: a first guess b=(sqrt(s)-edge0)*inv_spacing, corrected with thr. */
int vps_set_binning(vps_ctx* ctx, int N, const double* k2_axis_host,
                    const double* thr_host, int nbins, double edge0, double inv_spacing);

/* Binning-only consumers: while `on`, the y passes (vps_fft_zy, vps_fft_zy_weighted, vps_deposit_fft_zy, vps_fft_y) do
 * not store rows (ky, kz) all of whose modes lie beyond the last shell edge of the current vps_set_binning tables --
 * fl(ky^2 + kz^2) >= thr[nbins]; with the default k range a fifth of the half spectrum -- and leave that memory untouched;
 * the binning x passes (vps_fft_x modes 0 / 3, vps_fft_x_bin) never read them.  Switch it off (the default) before any
 * call whose output is read in full (vps_fft_x modes 1, 2; vps_rfft3 and vps_power_grid do so themselves). */
int vps_set_bin_only(vps_ctx* ctx, int on);

/* Deconvolution of a mass-assignment window in the binning x pass (extension; the reference has no
 * higher-order assignment): inv_w2_axis_host[N] (float32, indexed like k2_axis, even in k) holds 1 / W(k)^2 of ONE axis,
 * W(k) = sinc(pi k / (2 k_Nyquist))^p with p = 1 (NGP), 2 (CIC), 3 (TSC); every |F(k)|^2 that vps_fft_x (modes 0, 3) and
 * vps_fft_x_bin accumulate is multiplied by the product of the three axis factors.  NULL switches it off (default). */
int vps_set_window(vps_ctx* ctx, int N, const float* inv_w2_axis_host);

/* Local part of the transform on an x-slab: z pass (R2C) and y pass.
 * field_dev: [nx][N][N] float32 real input (NOT modified).
 * spec_dev : [N/2][N][nx] complex64 = F_zy[kz][ky][x]   (kz < N/2)
 * nyq_dev  : [N][nx]      complex64 = F_zy[kz=N/2][ky][x]
 * work_dev : vps_fft_workspace_bytes(N,nx) bytes.
 * nx must be a multiple of 16 (or equal to N when N < 16... see vps_fft_supported). */
size_t vps_fft_workspace_bytes(int N, int nx);
int vps_fft_zy(vps_ctx* ctx, int N, int nx, const float* field_dev,
               void* spec_dev, void* nyq_dev, void* work_dev);
/* The same for the product field_dev * weight_dev (cell by cell; weight_dev NULL = vps_fft_zy): momentum
 * components v_c * mass of a gridded field (interp.py:523-525) without a separate algebra pass. */
int vps_fft_zy_weighted(vps_ctx* ctx, int N, int nx, const float* field_dev, const float* weight_dev,
                        void* spec_dev, void* nyq_dev, void* work_dev);

/* The same two passes as separate calls, for the chunked slab exchange of several ranks (G = number of x-slabs):
 *   vps_fft_z          z pass of field_dev (* weight_dev if not NULL) into a z image zimg_dev
 *                      (vps_fft_zimage_bytes(N, nx) bytes: B[x][kz][y], kz < N/2, followed by the Nyquist plane BN[x][y]);
 *   vps_deposit_fft_z  the fused deposit + field algebra + z pass of vps_deposit_fft_zy, stopping there: ncomp z images
 *                      (3: velocity / momentum, 1: energy) in zimg_dev; work_dev: vps_deposit_fft_z_workspace_bytes;
 *   vps_fft_y          y pass of chunk `chunk` of `nchunks` of a z image.  The kz < N/2 planes are cut into nchunks BANDS
 *                      of G*nkc planes (nkc = N/2/G/nchunks) and dealt out round-robin inside a band: slot j of rank h is
 *                      plane chunk*G*nkc + j*G + h (every rank gets planes of every |kz|, and the j-th planes of all ranks
 *                      are neighbours).  They are transformed and written as ONE send buffer for an equal-split all-to-all,
 *                      out_dev = [h][ slot 0 rows | slot 1 rows | ... | -- last chunk only -- Nyquist rows
 *                      F_zy[N/2][ky in rank h's N/G rows][x] (N/G*nx) ], each row nx complex64.  A slot holds all N rows of
 *                      its plane (row = ky), or -- inside a binning-only scope (vps_set_bin_only), vps_fft_y_packed() = 1
 *                      -- only the 2 kc + 1 rows |ky| <= kc that the slot's planes can still contribute to a shell (row ky
 *                      at position ky, row N - i at position 2 kc + 1 - i), so the rows nobody bins do not cross the node
 *                      either.  One destination's block is vps_fft_y_chunk_block(ctx, ..., packed) elements,
 *                      vps_fft_y_chunk_elems(...) (all rows, all G blocks) an upper bound of the whole buffer.  After the
 *                      exchange the G blocks a rank received (x running over the senders' slabs) go to
 *                      vps_fft_x_bin_chunk with the same `packed`.
 * This replaces the four allgathers per buffer flush of scripts/parallel_optimized.py:365-368 with one message per
 * scalar field (or several chunks of it, so that the exchange of one chunk overlaps the passes of its neighbours). */
size_t vps_fft_zimage_bytes(int N, int nx);
int vps_fft_z(vps_ctx* ctx, int N, int nx, const float* field_dev, const float* weight_dev, void* zimg_dev);
size_t vps_deposit_fft_z_workspace_bytes(int64_t np, int N, int nx);
int vps_deposit_fft_z(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                      const float* rho_dev, int64_t np, int N, double Lbox, int x0, int nx,
                      int quantity, int flags, void* zimg_dev, void* work_dev);
/* Ranks that hold a REPLICATED particle set but deposit one x-slab of it (scripts/parallel_optimized.py:272-276 loads the whole
 * snapshot on every rank): the sort workspace sized for the particles INSIDE the slab instead of all np --
 *   vps_count_in_slab                       that number, with the deposit's own bit-exact cell rule (one pass over the
 *                                           positions, blocks; < 0: error);
 *   vps_deposit_fft_z_workspace_bytes_slab  workspace for np particles of which at most np_slab lie in the slab: no per-input
 *                                           key array, compact {key, payload} and record arrays of np_slab entries (C5 on 8
 *                                           ranks: 52 GB -> 9 GB); the slab's particles are filtered into them by one pass
 *                                           over the positions, the sort runs on the compact arrays only;
 *   vps_deposit_fft_z_slab                  vps_deposit_fft_z on such a workspace; VPS_ERR_ARG if the slab holds more particles
 *                                           than the workspace has room for (np_slab + the filter pass's block slack; checked
 *                                           before anything is written past it). */
int64_t vps_count_in_slab(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, int64_t np, int N, double Lbox, int x0, int nx);
size_t vps_deposit_fft_z_workspace_bytes_slab(int64_t np, int64_t np_slab, int N, int nx);
int vps_deposit_fft_z_slab(vps_ctx* ctx, const void* pos_dev, int pos_is_f64, const float* vel_dev,
                           const float* rho_dev, int64_t np, int64_t np_slab, int N, double Lbox, int x0, int nx,
                           int quantity, int flags, void* zimg_dev, void* work_dev);
int64_t vps_fft_y_chunk_elems(int N, int nx, int G, int nchunks, int chunk);   /* -1: G, nchunks do not divide */
int vps_fft_y_packed(vps_ctx* ctx, int N);   /* 1: vps_fft_y packs rows now (binning-only scope with a row cut for N) */
int64_t vps_fft_y_chunk_block(vps_ctx* ctx, int N, int nx, int G, int nchunks, int chunk, int packed);   /* pure size query (host tables only). -1: bad arguments, -2: G x nchunks does not divide N/2, -3: packed without a row cut (vps_last_error says which) */
int vps_fft_y(vps_ctx* ctx, int N, int nx, const void* zimg_dev, int G, int nchunks, int chunk, void* out_dev);

/* x pass over `nlines` lines of length N.  Line i is made of nseg segments of
 * N/nseg contiguous complex64: element x of line i lives at
 *   in_dev[(x / seglen) * seg_stride + i * seglen + (x % seglen)],  seglen = N/nseg
 * (nseg=1: plain contiguous lines; nseg=G: the layout an all-to-all of G x-slabs
 * leaves behind).  Line i has global mode indices ky = (line0+i) % N and
 * kz = kz0 + (line0+i)/N.
 * mode 0: accumulate w*|F|^2 (w = 1 for kz in {0,N/2}, else 2) into
 *         psum_dev[nbins] (float64) and w into nsample_dev[nbins] (uint64) using
 *         the tables of vps_set_binning;  replaces _pair_power+_hist_sample
 *         (interp.py:1440-1482) / pair_power+hist_sample (parallel_optimized.py:145-190).
 * mode 3: as mode 0 but shell sums only (nsample_dev untouched): the counts depend on the
 *         mode lattice alone, so a vector spectrum counts once, on its first component.
 * mode 4: transform and discard (a timing aid: the pass without its epilogue).
 * mode 1: write the transformed lines to out_dev[i][kx] (complex64, contiguous).
 * mode 2: out_dev[i][kx] (float32) += |F|^2, no binning (component sums of
 *         _vector_power, interp.py:1386).                                           */
int vps_fft_x(vps_ctx* ctx, int N, int64_t nlines, int64_t line0, int kz0,
              const void* in_dev, int nseg, int64_t seg_stride, int mode,
              double* psum_dev, unsigned long long* nsample_dev, void* out_dev);

/* Binning x pass of a VECTOR field: the ncomp (1..3) component spectra in_devs[c] (same layout and
 * arguments as vps_fft_x) are transformed line by line, their |F|^2 SUMMED, and the sum binned once
 * (mode 0 when count != 0, else mode 3) -- the component sum of _vector_power (interp.py:1386,
 * parallel_optimized.py:131-137) fused with _pair_power/_hist_sample.  in_devs is a HOST array of
 * device pointers. */
int vps_fft_x_bin(vps_ctx* ctx, int N, int64_t nlines, int64_t line0, int kz0,
                  const void* const* in_devs, int ncomp, int nseg, int64_t seg_stride, int count,
                  double* psum_dev, unsigned long long* nsample_dev);
/* Binning x pass of ONE received chunk of the slab exchange (layout: vps_fft_y): in_devs[c] = the G blocks this rank
 * received for component c, `packed` what the senders' vps_fft_y_packed() returned; count as in vps_fft_x_bin.  The last
 * chunk's Nyquist-plane rows are binned by the same call. */
int vps_fft_x_bin_chunk(vps_ctx* ctx, int N, int nx, int G, int nchunks, int chunk, int rank, int packed,
                        const void* const* in_devs, int ncomp, int count, double* psum_dev,
                        unsigned long long* nsample_dev);

/* ---- slab exchange inside the library: RCCL over xGMI (one process per GPU) -------------------------------------------
 * For hosts without a collective of their own (the Python host drives the same chunk pipeline through torch.distributed,
 * vpower/device.py).  Replaces the four comm.allgather per buffer flush and the two comm.Reduce of
 * scripts/parallel_optimized.py:365-368, 455-456.  RCCL is loaded at run time (dlopen): VPS_ERR_UNSUPPORTED without it.
 *   vps_comm_unique_id    rank 0: 128 bytes for the host to hand to every rank (ncclGetUniqueId)
 *   vps_comm_create       every rank, collectively: ncclCommInitRank on the context's device + a communication stream
 *   vps_spectrum_zimages  the x-side of the transform of ncomp (1..3) z images (vps_fft_z / vps_deposit_fft_z[_slab]) of this
 *                         rank's slab, nx = N / world: per kz chunk the y passes into the send buffers, ONE grouped
 *                         ncclSend / ncclRecv exchange per chunk on the communication stream, the binning x pass of the
 *                         received blocks (component |F|^2 summed before the shell search) -- all y passes are enqueued
 *                         first, so chunk c travels while c + 1 is transformed and is binned while c + 1 travels.  Needs the
 *                         tables of vps_set_binning; blocks carry only the rows a shell can reach.  Accumulates into
 *                         psum_dev / (count != 0) nsample_dev; xwork_dev: vps_spectrum_zimages_workspace_bytes.
 *   vps_allreduce_shells  sum of the accumulators over the ranks (ncclAllReduce, float64 + uint64), on the context's stream */
int vps_comm_unique_id(char* id128);
int vps_comm_create(vps_ctx* ctx, int rank, int world, const char* id128);
int vps_comm_destroy(vps_ctx* ctx);
int vps_comm_info(vps_ctx* ctx, int* rank, int* world);
size_t vps_spectrum_zimages_workspace_bytes(int N, int nx, int G, int nchunks, int ncomp);
int vps_spectrum_zimages(vps_ctx* ctx, int N, int nx, const void* const* zimg_devs, int ncomp, int nchunks, void* xwork_dev,
                         int count, double* psum_dev, unsigned long long* nsample_dev);
int vps_allreduce_shells(vps_ctx* ctx, double* psum_dev, unsigned long long* nsample_dev, int nbins);

/* Single-GPU convenience: full |F(k)|^2 binning of one real field.
 * field_dev [N][N][N] float32 is preserved; work_dev: vps_power_workspace_bytes(N). */
size_t vps_power_workspace_bytes(int N);
int vps_power_bin(vps_ctx* ctx, int N, const float* field_dev, void* work_dev,
                  double* psum_dev, unsigned long long* nsample_dev);
/* Single-GPU half spectrum for the un-binned API (_vector_power/_scalar_power,
 * interp.py:1372-1421): out_dev [N/2+1][N][N] complex64 = F[kz][ky][kx].          */
int vps_rfft3(vps_ctx* ctx, int N, const float* field_dev, void* work_dev, void* out_dev);
/* Same transform, but power_dev [N/2+1][N][N] float32 += |F[kz][ky][kx]|^2 (caller zeroes
 * it; several components accumulate): the un-binned P grid of _vector_power.       */
int vps_power_grid(vps_ctx* ctx, int N, const float* field_dev, void* work_dev, float* power_dev);

/* ---- un-fused binning API (not on the hot path) --------------------------- */
/* out_dev[N^3] float64 = sqrt(kx*kx + ky*ky + kz*kz) over the meshgrid of the host axes,
 * C order 'ij': column 0 of _pair_power (interp.py:1449-1462); the three axes are
 * separate so that the per-axis shift of interp.py:1453-1458 can be applied.        */
int vps_pair_k(vps_ctx* ctx, int N, const double* kx_host, const double* ky_host,
               const double* kz_host, double* out_dev);
/* numpy.histogram(k, bins=edges, weights=w) and the unweighted counts in one pass:
 * edges[i] <= k < edges[i+1], last bin right-closed (interp.py:1474-1477).
 * k_dev, w_dev: float64[n] (w_dev may be NULL: weights 1); edges_host[nbins+1];
 * psum_dev float64[nbins] and nsample_dev uint64[nbins] are accumulated into.  Blocks. */
int vps_hist_pairs(vps_ctx* ctx, const double* k_dev, const double* w_dev, int64_t n,
                   const double* edges_host, int nbins, double* psum_dev,
                   unsigned long long* nsample_dev);

#ifdef __cplusplus
}
#endif
#endif /* VPS_HIP_H */
