/*
 * momcore.h -- C ABI of libmomcore.so, the MI355X (gfx950) Matrix-Operator RT core.
 *
 * Drop-in boundary for ONE path of vSmartMOM.jl (RadiativeTransfer/RadiativeTransfer.jl):
 * the CoreRT layer-adding loop  elemental! -> doubling! -> interaction!  inside rt_run,
 * the Lambertian surface + post-processing that close it, and the Absorption Voigt
 * line-shape kernel.  Every entry point names the reference interface it replaces
 * (file:line relative to the reference root).  A Julia maintainer binds these with
 * `ccall((:mom_xxx, libmomcore), Cint, (...), ...)` -- see INTEGRATION.md.
 *
 * Conventions
 *   - all functions return int status: MOM_OK (0) or a negative MOM_E* code; the text of
 *     the last error is available from mom_last_error().
 *   - arrays use the reference's memory order: Julia column-major [i, j, n] with the
 *     spectral index n slowest (batch stride N*N); sources are [i, 1, n]; optical depth
 *     tables are [nSpec, Nz] (model_from_parameters.jl:48); outputs are
 *     [nVza, nStokes, nSpec] (rt_run.jl:89-90).
 *   - indices passed as `*_1based` are Julia indices.
 *   - host pointers are borrowed for the duration of the call only.  All device memory is
 *     owned by the handle.  One handle <-> one GPU <-> one HIP stream; a handle is not
 *     thread-safe, distinct handles are independent.
 *   - `dtype` of mom_create: 0 = Float64 (the reference's default float_type), 1 = Float32 (float_type = Float32,
 *     parameters_from_yaml.jl:160): operators, sources and products in f32 on the GPU.  The ABI keeps Float64 host
 *     arrays for both (inputs are rounded on upload, outputs widened on download).  A Float32 handle supports the
 *     scene-level path -- mom_set_streams, mom_scene_set, mom_scene_set_optics with the mom_absorption_* / mom_voigt_tau_abs*
 *     entry points that feed it (the layer optics are assembled in Float64 on the device and rounded to Float32 there),
 *     mom_scene_set_surface, mom_set_option, mom_rt_run, mom_get_RT, mom_get_hdr, mom_timers, mom_sync, mom_check: the lane- and wave-per-point kernels (incl. the packed
 *     points at N = 5 ... 8), the strip-chained images (N = 36 ... 60, 4-wave builds: two workgroups per CU), the m = 0
 *     (I,Q) reduction and the padding to the strip sizes --, the operator-level API -- mom_elemental, mom_doubling,
 *     mom_interaction, mom_copy_added_to_composite, mom_surface_lambertian, mom_upload, mom_download (the operator-level
 *     composite layer is kept apart from the scene-level state on such a handle) -- and the two batched operators
 *     mom_batch_inv / mom_batched_mul (gpu_batched.jl:45-58); every other entry point returns MOM_EINVAL on it.
 */
#ifndef MOMCORE_H
#define MOMCORE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mom_handle mom_t;

enum {
  MOM_OK = 0,
  MOM_EINVAL = -1,   /* bad argument */
  MOM_EHIP = -2,     /* HIP runtime error (no device, launch failure, OOM ...) */
  MOM_ESTATE = -3,   /* call sequence error (e.g. streams not set) */
  MOM_ESINGULAR = -4, /* an (I - R r) operator was numerically singular (zero pivot) */
  MOM_EUNSUPPORTED = -5 /* the reference itself raises on this path (RRS strict position), or outside the built scope */
};

/* which-codes for mom_upload / mom_download (AddedLayer / CompositeLayer fields,
 * types.jl:105-142).  Matrices are N*N*S doubles, sources N*S doubles.
 * The surface layer (rt_run.jl:111-112 `added_layer_surface`) is a second AddedLayer. */
enum {
  MOM_ADDED_R_PM = 0, MOM_ADDED_R_MP = 1, MOM_ADDED_T_MM = 2, MOM_ADDED_T_PP = 3,
  MOM_ADDED_J0P = 4, MOM_ADDED_J0M = 5,
  MOM_COMP_R_MP = 6, MOM_COMP_R_PM = 7, MOM_COMP_T_PP = 8, MOM_COMP_T_MM = 9,
  MOM_COMP_J0P = 10, MOM_COMP_J0M = 11,
  MOM_SURF_R_PM = 12, MOM_SURF_R_MP = 13, MOM_SURF_T_MM = 14, MOM_SURF_T_PP = 15,
  MOM_SURF_J0P = 16, MOM_SURF_J0M = 17
};

/* ---- lifetime ------------------------------------------------------------------------ */

/* Allocates the layer state of rt_run.jl:109-115 (make_added_layer x2, make_composite_layer;
 * rt_helper_functions.jl:105-121,152-159) on GPU `device` for operators of edge
 * N = nStokes*Nquad and S spectral points, for up to `max_m` Fourier moments processed
 * as one batch (the moments of rt_run.jl:125 are independent).  */
int mom_create(mom_t **out, int device, int N, int nStokes, int S, int max_m, int dtype);
int mom_destroy(mom_t *h);
const char *mom_last_error(const mom_t *h);
/* library-level error text for failures that happen before a handle exists */
const char *mom_last_global_error(void);
/* blocks until everything queued on the handle's stream has finished */
int mom_sync(mom_t *h);
/* mom_sync + the deferred error report of the asynchronous calls (MOM_ESINGULAR if a kernel met a zero pivot) */
int mom_check(mom_t *h);

/* QuadPoints (types.jl:456-473) + polarization type (Scattering/types.jl:82-123).
 * qp_muN/wt_muN: N values (already repeated per Stokes component); imu0_1based = iμ₀;
 * mu0 = quad_points.μ₀; I0, D: nStokes values.  strict_reference_indexing != 0 keeps the
 * reference's 1-based `mod(i, nStokes)` Stokes-component rule of elemental.jl:259-269,
 * doubling.jl:95-117; 0 uses the zero-based component. */
int mom_set_streams(mom_t *h, const double *qp_muN, const double *wt_muN, int N, int imu0_1based, double mu0,
                    const double *I0, const double *D, int strict_reference_indexing);

/* ---- operator-level API: one call per reference operator (used for parity tests and for
 *      a Julia shim that overloads the operators one by one) ------------------------------ */

/* elemental!(pol_type, SFI, τ_sum, dτ, computed_layer_properties, m, ndoubl, scatter, quad_points,
 *            added_layer, architecture)  -- elemental.jl:109-162 (+ kernels :164-285).
 * tau_sum, dtau, varpi: S values; Zpp/Zmp: N*N*z_batch with z_batch = 1 or S
 * (expandOpticalProperties, compEffectiveLayerProperties.jl:124-135).  Fills the added layer. */
int mom_elemental(mom_t *h, int m, int ndoubl, const double *tau_sum, const double *dtau, const double *varpi,
                  const double *Zpp, const double *Zmp, int z_batch);

/* doubling!(pol_type, SFI, expk, ndoubl, added_layer, I_static, architecture) -- doubling.jl:81-91
 * (helper :13-79).  expk: S values, updated in place like the reference's `expk .= expk.^2`. */
int mom_doubling(mom_t *h, int ndoubl, double *expk);

/* interaction!(RS_type::noRS, scattering_interface, SFI, composite_layer, added_layer, I_static)
 * -- interaction_inelastic.jl:474-484 -> interaction.jl:8-117.
 * iface: 0 = ScatteringInterface_00, 1 = _01, 2 = _10, 3 = _11.
 * with_surface_layer != 0 uses added_layer_surface instead of added_layer (rt_run.jl:179-185). */
int mom_interaction(mom_t *h, int iface, int with_surface_layer);

/* rt_kernel.jl:227-230: composite <- added (first layer, iz == 1) */
int mom_copy_added_to_composite(mom_t *h);

/* create_surface_layer!(::LambertianSurfaceScalar, added_layer_surface, SFI, m, pol_type, quad_points,
 *                       τ_sum, architecture) -- lambertian_surface.jl:20-75.  tau_tot: S values. */
int mom_surface_lambertian(mom_t *h, int m, double albedo, const double *tau_tot);

/* elemental_inelastic!(RS_type::RRS, pol_type, SFI, τ_sum, dτ_λ, ϖ_λ, Z⁺⁺_λ₁λ₀, Z⁻⁺_λ₁λ₀, m, ndoubl, scatter, quad_points,
 *                      added_layer, I_static, architecture) -- CoreKernel/elemental_inelastic.jl:23-91: the elemental layer of
 * the rotational-Raman source operators (get_elem_rt_RRS! :93-160, get_elem_rt_SFI_RRS! :320-382, apply_D_elemental_RRS!
 * :384-402).  i_l1l0 [nRaman]: grid offsets n₀ - n₁ of the Raman lines (RS_type.i_λ₁λ₀), varpi_l1l0 [nRaman], fscattRayl,
 * tau_sum, dtau, varpi [nSpec], Z*_l1l0 [N,N].  Outputs (host): ier⁻⁺, iet⁺⁺, ier⁺⁻, iet⁻⁻ [N,N,nSpec,nRaman], ieJ₀⁺, ieJ₀⁻
 * [N,nSpec,nRaman] (Julia [N,1,nSpec,nRaman]).  For ndoubl >= 1 the reference leaves ier⁺⁻ / iet⁻⁻ untouched (they are
 * filled after doubling); zeros are returned.  Stateless form (fresh arrays in, host arrays out); the stateful RRS path
 * below (mom_rrs_*) keeps the reference's persistent AddedLayerRS instead. */
int mom_elemental_inelastic_rrs(mom_t *h, int m, int ndoubl, int nRaman, const int *i_l1l0, const double *varpi_l1l0,
                                const double *fscattRayl, const double *tau_sum, const double *dtau, const double *varpi,
                                const double *Zpp_l1l0, const double *Zmp_l1l0, double *ier_mp, double *iet_pp,
                                double *ier_pm, double *iet_mm, double *ieJ0p, double *ieJ0m);

/* ---- rotational-Raman scattering (BASELINE config 5): rt_run(RS_type::RRS, model, iBand) ---------------------------
 *
 * The reference's inelastic branch keeps, next to the elastic layers, the 4-D operators ier-+, ier+-, iet++, iet--
 * [N,N,nSpec,nRaman] and sources ieJ0+-[N,1,nSpec,nRaman] of the added layer and ieR-+, ieR+-, ieT++, ieT--, ieJ0+- of the
 * composite layer (AddedLayerRS / CompositeLayerRS, src/CoreRT/types.jl:145-205): element [.., n1, dn] takes radiation
 * from spectral index n0 = n1 + i_l1l0[dn] to n1.  These layers persist over layers and Fourier moments exactly like the
 * reference's (rt_run.jl:108-116), in the reference's memory order.
 *
 *   mom_rrs_set        the fields of the RRS struct (src/Inelastic/types.jl:13-33) the path reads: i_l1l0 [nRaman] (grid
 *                      offsets n0 - n1, |offset| < nSpec), varpi_l1l0 [nRaman]; allocates the layers (zeros, like
 *                      make_added_layer(::RRS, ...) rt_helper_functions.jl:127-141).  N <= 64 (N <= 16: one MFMA tile per operator and wavefront,
 *                      <= 32: 2 x 2 tiles in registers, <= 48 / <= 64: 3 x 3 / 4 x 4 tiles with the operators in scratch).
 *                      rrs_strict_reference: the reference's RRS text has defects (DESIGN.md "RRS", D1..D5); != 0 executes
 *                      it AS WRITTEN with the semantics of a single-threaded Julia run -- and returns MOM_EUNSUPPORTED
 *                      where the reference raises (interaction_helper! for interfaces 00/01/10, a MethodError) --,
 *                      0 applies the five documented corrections and nothing else.
 *   mom_rrs_elemental  rt_kernel!(::RRS) rt_kernel.jl:277-304: elemental_inelastic!(RS_type, pol_type, SFI, tau_sum, dtau, varpi,
 *                      Z++_l1l0, Z-+_l1l0, m, ndoubl, scatter, quad_points, added_layer, I_static, architecture)
 *                      (CoreKernel/elemental_inelastic.jl:23-91) followed by elemental!(...) (elemental.jl:109-162) on the
 *                      persistent added layer.  tau_sum, dtau, varpi, fscattRayl [nSpec]; Z* [N,N] (one phase matrix).
 *   mom_rrs_doubling   doubling_inelastic!(RS_type, pol_type, SFI, expk, ndoubl, added_layer, I_static, architecture)
 *                      (CoreKernel/doubling_inelastic.jl:13-134, 263-280); expk [nSpec] updated in place.
 *   mom_rrs_interaction  interaction!(RS_type::RRS, scattering_interface, SFI, composite_layer, added_layer, I_static)
 *                      (CoreKernel/interaction_inelastic.jl:464-472 -> :8-340); with_surface_layer != 0 uses added_layer_surface,
 *                      whose ie* arrays the reference never writes (zeros).
 *   mom_rrs_copy_added_to_composite   rt_kernel.jl:326-333 (iz == 1), all twelve arrays.
 *   mom_rrs_surface_lambertian        create_surface_layer!(::LambertianSurfaceScalar) into added_layer_surface.
 *   mom_rrs_upload / mom_rrs_download  test access; which = MOM_ADDED_* / MOM_COMP_* / MOM_SURF_* for the elastic fields of the RRS
 *                      layers, MOM_IE_ADDED_* / MOM_IE_COMP_* for the 4-D fields ([N,N,nSpec,nRaman] / [N,nSpec,nRaman]).
 *   mom_scene_set_rrs  scene-level inputs on top of mom_scene_set: fscattRayl [nSpec, Nz] (fScattRayleigh of
 *                      constructCoreOpticalProperties, compEffectiveLayerProperties.jl:58, expanded per band), Z*_l1l0 [N,N,M]
 *                      (computeRamanZlambda!, src/Inelastic/inelastic_helper.jl:457-464, per Fourier moment).
 *   mom_rt_run_rrs     rt_run.jl:125-215 with RS_type::RRS for the resident scene (every surface kind of
 *                      mom_scene_set_surface); asynchronous.
 *   mom_rrs_set_shard  spectral sharding of the RRS path (the Raman operators couple spectral points at the
 *                      offsets i_l1l0, inelastic_helper.jl:13-21, so a shard carries a halo): the handle's nSpec points are the
 *                      window [n_glob0, n_glob0 + nSpec) of a global axis of nSpec_global points, of which this rank OWNS the
 *                      local indices [n1_lo, n1_hi).  The elastic layers are computed for the whole window (they are the
 *                      operands at n0 = n1 + i_l1l0), the inelastic pairs (n1, dn) and the spectra only for the owned
 *                      points; on an interior edge the halo must be at least max |i_l1l0| long (MOM_EINVAL otherwise).
 *                      ndoubl / interface codes must come from the GLOBAL axis (rt_kernel.jl:241-242), as in the elastic
 *                      sharding.  Call after mom_rrs_set; the default is the whole window, n_glob0 = 0.
 *   mom_get_RT_rrs     R_SFI, T_SFI, ieR_SFI, ieT_SFI [nVza, nStokes, nSpec] (postprocessing_vza!(::RRS),
 *                      tools/postprocessing_vza.jl:95-147); any pointer may be NULL; gpu_ms (optional) = GPU time of the run.
 *   mom_get_hdr_rrs    the elastic RAMI extras of the same return tuple (rt_run.jl:187-213, 226): hdr [nVza, nStokes, nSpec],
 *                      bhr_uw, bhr_dw [nStokes, nSpec] (interaction_hdrf! + postprocessing_vza_hdrf!).
 *   mom_rrs_timers     HIP-event times of the last mom_rt_run_rrs, summed per kernel: ms[0] / launches[0] the doubling pair
 *                      kernel, [1] the interaction pair kernel, [2] the inelastic elemental kernel, [3] the whole run (n >= 4).
 *   mom_get_spectra_rrs_device  the SEVEN spectra of the return tuple (rt_run.jl:226) restricted to the owned points
 *                      [n1_lo, n1_hi) of mom_rrs_set_shard, packed into the caller's DEVICE buffer as
 *                      [R | T | ieR | ieT | hdr][nVza, nStokes, per] then [bhr_uw | bhr_dw][nStokes, per]; per >= owned count,
 *                      tails zero (ragged last shard); mom_rrs_spectra_count(h, per) doubles; asynchronous.
 *   mom_allgather_rrs_device    the ONE collective of a sharded RRS run: that block of every rank into d_global
 *                      [nranks][mom_rrs_spectra_count(h, per)] (RCCL on the handle's stream; the reference has no
 *                      multi-GPU path -- SURVEY section 8e / 8f-3; needs mom_comm_init). */
enum {
  MOM_IE_ADDED_R_PM = 18, MOM_IE_ADDED_R_MP = 19, MOM_IE_ADDED_T_MM = 20, MOM_IE_ADDED_T_PP = 21,
  MOM_IE_ADDED_J0P = 22, MOM_IE_ADDED_J0M = 23,
  MOM_IE_COMP_R_MP = 24, MOM_IE_COMP_R_PM = 25, MOM_IE_COMP_T_PP = 26, MOM_IE_COMP_T_MM = 27,
  MOM_IE_COMP_J0P = 28, MOM_IE_COMP_J0M = 29
};
int mom_rrs_set(mom_t *h, int nRaman, const int *i_l1l0, const double *varpi_l1l0, int rrs_strict_reference);
int mom_rrs_set_shard(mom_t *h, int nSpec_global, int n_glob0, int n1_lo, int n1_hi);
int mom_rrs_elemental(mom_t *h, int m, int ndoubl, const double *tau_sum, const double *dtau, const double *varpi,
                      const double *Zpp, const double *Zmp, const double *fscattRayl, const double *Zpp_l1l0,
                      const double *Zmp_l1l0);
int mom_rrs_doubling(mom_t *h, int ndoubl, double *expk);
int mom_rrs_interaction(mom_t *h, int iface, int with_surface_layer);
int mom_rrs_copy_added_to_composite(mom_t *h);
int mom_rrs_surface_lambertian(mom_t *h, int m, double albedo, const double *tau_tot);
int mom_rrs_upload(mom_t *h, int which, const double *src);
int mom_rrs_download(mom_t *h, int which, double *dst);
int mom_scene_set_rrs(mom_t *h, const double *fscattRayl, const double *Zpp_l1l0, const double *Zmp_l1l0);
int mom_rt_run_rrs(mom_t *h);
int mom_get_RT_rrs(mom_t *h, double *R_SFI, double *T_SFI, double *ieR_SFI, double *ieT_SFI, double *gpu_ms);
int mom_get_hdr_rrs(mom_t *h, double *hdr, double *bhr_uw, double *bhr_dw);
int mom_rrs_timers(mom_t *h, double *ms, int *launches, int n);
/* test access: number of nonzero entries in the zero padding of the handle's RRS layer arrays (device blocks are padded to the
 * MFMA tiling and the kernels store whole tiles: the padding must stay zero by value); 0 = intact */
int mom_rrs_check_padding(mom_t *h, unsigned long long *violations);
size_t mom_rrs_spectra_count(mom_t *h, int per);
int mom_get_spectra_rrs_device(mom_t *h, int per, void *d_local);
int mom_allgather_rrs_device(mom_t *h, int per, void *d_global);

/* batch_inv!(X, A) -- gpu_batched.jl:36-87;  X, A: n*n*batch doubles (host). */
int mom_batch_inv(mom_t *h, int n, int batch, const double *A, double *X);
/* A ⊠ B = batched_mul(A, B) -- gpu_batched.jl:90-97. */
int mom_batched_mul(mom_t *h, int n, int batch, const double *A, const double *B, double *C);
/* The ForwardDiff.Dual methods of the same two operators (gpu_batched.jl:100-150), values and partials as separate
 * arrays: values [n,n,batch], partials [n,n,batch,P] (one [n,n,batch] block per partial i = ForwardDiff.partials.(A, i)).
 *   batched_mul:  C = A ⊠ B,   dC_i = A ⊠ dB_i + dA_i ⊠ B            (gpu_batched.jl:100-110)
 *   batch_inv!:   X = A⁻¹,     dX_i = -A⁻¹ ⊠ dA_i ⊠ A⁻¹              (gpu_batched.jl:129-150)
 * P = 0 reduces to the plain operators (dA/dB/dC may be NULL then). */
int mom_batched_mul_dual(mom_t *h, int n, int batch, int P, const double *A, const double *dA, const double *B,
                         const double *dB, double *C, double *dC);
int mom_batch_inv_dual(mom_t *h, int n, int batch, int P, const double *A, const double *dA, double *X, double *dX);

/* Array(composite_layer.J₀⁻) etc. (postprocessing_vza.jl:17-20) / test access.
 * Operator-level state lives in moment slot 0 of the handle; the added and surface layers are allocated on the
 * first operator-level call.  After a scene-level mom_rt_run the composite codes return Fourier moment 0 of that
 * run in the same [N,N,nSpec] / [N,nSpec] layout (the internal row pitch is removed), or MOM_ESTATE when moment 0
 * ran on the (I,Q) sub-problem (MOM_OPT_M0_REDUCTION); mom_interaction / mom_postprocess refuse to continue from
 * scene-level state (MOM_ESTATE) until mom_copy_added_to_composite or a composite mom_upload restarts the
 * operator-level sequence. */
int mom_upload(mom_t *h, int which, const double *src);
int mom_download(mom_t *h, int which, double *dst);

/* postprocessing_vza!(RS_type::noRS, iμ₀, pol_type, composite_layer, vza, qp_μ, m, vaz, μ₀, weight, nSpec, SFI, R, R_SFI,
 * T, T_SFI, ieR_SFI, ieT_SFI) -- postprocessing_vza.jl:9-60, SFI branch, for ONE Fourier moment m of the
 * operator-level composite layer: R_SFI[i,:,s] += bigCS * J₀⁻[rows(i),1,s], T_SFI likewise with J₀⁺ (:48-49);
 * node_1based[i] = nearest_point(qp_μ, cosd(vza[i])) (:28), vaz_deg in degrees, weight = 0.5 for m = 0 else 1
 * (rt_run.jl:128).  R_SFI, T_SFI: [nVza, nStokes, nSpec] host arrays, accumulated in place like the reference. */
int mom_postprocess(mom_t *h, int m, int nVza, const int *node_1based, const double *vaz_deg, double weight,
                    double *R_SFI, double *T_SFI);

/* ---- scene-level API: inputs resident in HBM, fused per-layer kernels -------------------
 *
 * mom_scene_set uploads everything rt_run's loops (rt_run.jl:125-215) consume, prepared by
 * the host exactly as the reference's host code does (constructCoreOpticalProperties /
 * extractEffectiveProps / get_dtau_ndoubl):
 *   tau, varpi  [S, Nz]         layer optical depth and single-scattering albedo
 *   zw          [K, S, Nz]      weights of the K phase-matrix bases (Rayleigh + aerosols):
 *                               Z[:,:,n] = sum_k zw[k,n,z] * Zbasis_k  (types.jl:656-661)
 *   Zpp, Zmp    [N, N, K, M]    compute_Z_moments per basis and Fourier moment
 *   ndoubl      [Nz]            doubling_number per layer (GLOBAL over the spectral axis,
 *                               rt_kernel.jl:238-246 -- compute before sharding)
 *   iface       [Nz]            scattering-interface codes (rt_helper_functions.jl:8-27)
 *   tau_sum     [S, Nz+1]       cumulative optical depth (compEffectiveLayerProperties.jl:108)
 *   albedo                      LambertianSurfaceScalar
 *   node_1based [nVza]          nearest stream per view (postprocessing_vza.jl:28)
 *   cos_mphi, sin_mphi [nVza, M] cosd(m*vaz), sind(m*vaz) (postprocessing_vza.jl:32)
 *
 * m = 0 reduction.  For Fourier moment 0 the generalized spherical function T_l^0 vanishes, so every
 * phase-matrix basis from compute_Z_moments couples (I,Q) only with (I,Q) and (U,V) only with (U,V)
 * (compute_Z_matrices.jl:36-57 with legendre_functions.jl:38-58).  With an unpolarised source
 * (I0 = [1,0,0(,0)]) and a Lambertian surface the (U,V) block then has no source: it never reaches
 * R_SFI/T_SFI/hdr (its azimuthal weight sin(0) is 0 as well), and the (I,Q) block evolves exactly as in
 * the full problem (the cross products are exact zeros).  mom_scene_set verifies these conditions on the
 * uploaded arrays (bitwise zeros) and, if they hold, runs moment 0 on operators of edge 2*Nquad; the
 * outputs are unchanged.  MOM_OPT_M0_REDUCTION = 0 disables it.
 */
int mom_scene_set(mom_t *h, int Nz, int K, int M, const double *tau, const double *varpi, const double *zw,
                  const double *Zpp, const double *Zmp, const int *ndoubl, const int *iface, const double *tau_sum,
                  double albedo, int nVza, const int *node_1based, const double *cos_mphi, const double *sin_mphi);

/* ---- device-side layer optics: tau_abs never visits the host -----------------------
 *
 * mom_absorption_begin   τ_abs = zeros(nSpec, Nz) resident in HBM (model_from_parameters.jl:48) + the spectral grid
 *                        [nSpec] the line shapes are evaluated on (NULL keeps a previously uploaded grid)
 * mom_voigt_tau_abs      compute_absorption_profile! for ONE layer (atmo_prof.jl:427-449):
 *                          τ_abs[:, iz] += absorption_cross_section(model, grid, p[iz], T[iz]) * vcd_dry[iz] * vmr
 *                        with the Voigt/HW32SD line shape; the host passes the per-line prefactors like for
 *                        mom_voigt_xsec and factor = vcd_dry[iz] * vmr; σ is accumulated straight into the resident
 *                        table by the line-shape kernel (no σ round trip, no allocation in steady state)
 * mom_absorption_set     uploads a host τ_abs [nSpec, Nz] instead (ABSCO/LUT or synthetic tables)
 * mom_absorption_get     test access
 * mom_scene_set_optics   mom_scene_set with the layer optics assembled ON THE DEVICE from their ingredients:
 *                        constructCoreOpticalProperties (compEffectiveLayerProperties.jl:1-78) = Rayleigh
 *                        (tau_rayl [nSpec, Nz], ϖ = varpi_rayl = ϖ_Cabannes) `+` each aerosol type (createAero :80-85:
 *                        tau_aer [nAer, Nz], omega_aer, ft_aer [nAer]; types.jl:632-661) `+` the resident gas
 *                        absorption (:672-678), τ_sum (:108), then ndoubl (rt_kernel.jl:238-246) and the
 *                        interface codes (rt_helper_functions.jl:8-27) from per-layer maxima of τϖ -- reduced over all
 *                        ranks first when a communicator is initialised (mom_comm_init), so a sharded run uses the
 *                        GLOBAL doubling numbers.  Zpp/Zmp: [N, N, 1 + nAer, M].  Results are bitwise those of the
 *                        host route (mom_scene_set fed by the reference's host algebra).
 * mom_scene_get_layers   ndoubl / iface [Nz] and (optionally, NULL to skip) τ, ϖ [nSpec,Nz], zw [K,nSpec,Nz],
 *                        τ_sum [nSpec,Nz+1] of the resident scene */
int mom_absorption_begin(mom_t *h, int Nz, const double *grid);
/* The same accumulation with the per-line prefactors formed ON THE DEVICE (last clause):
 *   mom_absorption_set_lines   ONE resident table per absorber: the HITRAN columns (read_hitran.jl:14-68) of the lines inside
 *                              the padded grid -- selected once by the host, compute_absorption_cross_section.jl:54-72 --,
 *                              sqrt_mol_weight = Float64(sqrt(mol_weight(mol, iso)::Float32)) per line, iso_index = the line's
 *                              row in the TIPS tables below, and for each of the nIso isotopologues in use the TIPS-2017
 *                              table as qoft! (:197-214) interpolates it: nT[k] knots tips_T, values tips_Q and the second
 *                              derivatives tips_z of DataInterpolations.CubicSpline ([nTmax, nIso] column-major, i.e. one
 *                              row of nTmax entries per isotopologue; computed once by the host in the tables' Float32).
 *   mom_voigt_tau_abs_layer    compute_absorption_profile! for layer iz (atmo_prof.jl:427-449): pressure shift, Lorentz and
 *                              Doppler widths, y, TIPS ratio, Boltzmann and stimulated-emission factors, grid windows
 *                              (compute_absorption_cross_section.jl:79-107) by k_line_prefactors, then the Voigt kernel adds
 *                              sigma * factor into tau_abs[:, iz].  Only (p, T, vmr, wing_cutoff, factor) cross the bus.
 *                              Agrees with the host route (mom_voigt_tau_abs fed by the host's prefactors) to rounding of
 *                              exp/pow (device vs host libm), not bitwise.
 *   mom_absorption_get_prefactors   the prefactors of the last layer call (test access). */
int mom_absorption_set_lines(mom_t *h, int nLines, const double *nu0, const double *S0, const double *gamma_air,
                             const double *gamma_self, const double *E_lower, const double *n_air, const double *delta_air,
                             const double *sqrt_mol_weight, const int *iso_index, int nIso, int nTmax, const int *nT,
                             const double *tips_T, const double *tips_Q, const double *tips_z);
int mom_voigt_tau_abs_layer(mom_t *h, int iz_1based, double pressure, double temperature, double vmr, double wing_cutoff,
                            double factor);
/* compute_absorption_profile! (atmo_prof.jl:427-449) for layers 1..Nz of the profile in TWO launches (prefactors of every
 * (layer, line) pair, line shapes of every (layer, grid point) pair) instead of the reference's host loop over layers with
 * one launch per line: tau_abs[:, z] += sigma(nu; pressure[z], temperature[z]) * factor[z], factor[z] = vcd_dry[z] * vmr[z].
 * Same arithmetic per layer as mom_voigt_tau_abs_layer (bitwise).  gpu_ms (may be NULL): HIP-event time of the two kernels. */
int mom_voigt_tau_abs_profile(mom_t *h, int Nz, const double *pressure, const double *temperature, double vmr, double wing_cutoff,
                              const double *factor, double *gpu_ms);
int mom_absorption_get_prefactors(mom_t *h, int n, double *nu, double *gamma_d, double *y, double *S, int *ind_start_1based,
                                  int *ind_stop_1based);
int mom_voigt_tau_abs(mom_t *h, int iz_1based, int nLines, const double *nu, const double *gamma_d, const double *y,
                      const double *S, const int *ind_start_1based, const int *ind_stop_1based, double factor);
int mom_absorption_set(mom_t *h, int Nz, const double *tau_abs);
int mom_absorption_get(mom_t *h, double *tau_abs);
int mom_scene_set_optics(mom_t *h, int Nz, int nAer, int M, const double *tau_rayl, double varpi_rayl,
                         const double *tau_aer, const double *omega_aer, const double *ft_aer, const double *Zpp,
                         const double *Zmp, double albedo, int nVza, const int *node_1based, const double *cos_mphi,
                         const double *sin_mphi);
int mom_scene_get_layers(mom_t *h, int *ndoubl, int *iface, double *tau, double *varpi, double *zw, double *tau_sum);

/* Surface type of the resident scene (call after mom_scene_set / mom_scene_set_optics, which select
 * LambertianSurfaceScalar(albedo)):
 *   kind 0  LambertianSurfaceScalar (lambertian_surface.jl:20-75), the `albedo` of mom_scene_set
 *   kind 1  any BRDF surface handled by create_surface_layer!(brdf::AbstractSurfaceType, ...) (rpv_surface.jl:20-66:
 *           rpvSurfaceScalar, RossLiSurfaceScalar, ...): Rsurf [N, N, M] = the matrices the host obtains from
 *           reflectance(brdf, pol_type, qp_μ, m) (rpv_surface.jl:104-134), times 2 for m = 0 (:39-43).  The library
 *           forms j₀⁺, j₀⁻ = μ₀ (R_surf I₀) e^{-τ/μ₀}, r⁻⁺ = R_surf Diagonal(qp_μN .* wt_μN) (:48-62) and interacts
 *           the surface with the composite layer for EVERY Fourier moment; hdr accumulates over all moments.
 *   kind 2  LambertianSurfaceLegendre (lambertian_surface.jl:77-138): albedo_spec [nSpec] = P * legendre_coeff
 *           evaluated by the host (:90-96); reproduces that method's j₀⁺ = 0 (:112) and its t = 0 for m > 0 (:131-132).
 * Unused pointers may be NULL. */
int mom_scene_set_surface(mom_t *h, int kind, int M, const double *Rsurf, const double *albedo_spec);

/* The whole of rt_run.jl:125-215 for the resident scene: all Fourier moments, all layers,
 * surface, post-processing.  Asynchronous on the handle's stream; results stay on the GPU. */
int mom_rt_run(mom_t *h);

/* ---- rt_run on ForwardDiff.Dual numbers: Jacobians out of the hot path --------------------------------------------
 * The reference differentiates rt_run by running it on Dual element types: rt_run.jl:89-96 allocates R, T, R_SFI, T_SFI
 * in the Dual type of the optical properties, every AddedLayer / CompositeLayer array is a Dual array, and the two
 * batched operators have Dual methods (gpu_batched.jl:100-150).  This is that run for the resident scene: values and P
 * partials of every layer operator go through elemental! / doubling! / interaction! / create_surface_layer! /
 * postprocessing_vza! together (csrc/mom_dual.hip: the Dual product rule per product, dX = -X dA X per inverse, the
 * analytic derivative of every elemental expression; integer decisions -- ndoubl, interface codes -- are the value run's,
 * as ForwardDiff takes them on the values).
 *
 * mom_scene_set_partials   the partials of what mom_scene_set / mom_scene_set_surface uploaded, i.e. of the Dual inputs the
 *                          reference's host code (model_from_parameters on Dual parameters) hands to rt_run.  Layouts are
 *                          those of the value arrays with the partial index as the SLOWEST axis; NULL = no dependence:
 *                            dtau, dvarpi [S, Nz, P];  dzw [K, S, Nz, P];  dZpp, dZmp [N, N, K, M, P] (both or neither);
 *                            dalbedo [P] (LambertianSurfaceScalar);  dRsurf [N, N, M, P] (surface kind 1);
 *                            dalbedo_spec [S, P] (kind 2).
 *                          Call after mom_scene_set (and mom_scene_set_surface); P = 0 clears them (values only).
 * mom_rt_run_dual          the run; R_SFI / T_SFI are then read with mom_get_RT (they agree with mom_rt_run's to rounding:
 *                          the same statements in a different association), the partials with mom_get_RT_partials.
 *                          hdr / bhr_uw / bhr_dw (interaction_hdrf!, postprocessing_vza_hdrf!) come out as well: values through
 *                          mom_get_hdr, partials through mom_get_hdr_partials.  Operators larger than 128 x 128 return
 *                          MOM_EUNSUPPORTED.
 * mom_get_RT_partials      dR_SFI, dT_SFI [nVza, nStokes, nSpec, P] (host).
 * mom_get_hdr_partials     dhdr [nVza, nStokes, nSpec, P], dbhr_uw, dbhr_dw [nStokes, nSpec, P] (host). */
int mom_scene_set_partials(mom_t *h, int P, const double *dtau, const double *dvarpi, const double *dzw, const double *dZpp,
                           const double *dZmp, const double *dalbedo, const double *dRsurf, const double *dalbedo_spec);
int mom_rt_run_dual(mom_t *h);
int mom_get_RT_partials(mom_t *h, double *dR_SFI, double *dT_SFI);
int mom_get_hdr_partials(mom_t *h, double *dhdr, double *dbhr_uw, double *dbhr_dw);

/* rt_run_test_ms(RS_type::noRS, sensor_levels, model, iBand) (src/CoreRT/rt_run_multisensor.jl:14-191) for the resident
 * scene: sensors inside the atmosphere.  sensor_levels[ims] = 0 is the TOA/BOA pair (uwJ = R_SFI, dwJ = T_SFI of
 * mom_rt_run); L in 1..Nz-1 puts the sensor below layer L counted from the top: rt_kernel_multisensor!
 * (rt_kernel_multisensor.jl:2-113) builds the composite of layers 1..L ("top") and of layers L+1..Nz plus the surface
 * ("bot"), interlayer_flux_helper! (CoreKernel/interlayer_flux.jl:7-24) solves for the fields at the interface,
 *   dwJ = (I - topR+- botR-+)^-1 (topJ0+ + topR+- botJ0-),  uwJ = (I - botR-+ topR+-)^-1 (botJ0- + botR-+ topJ0+),
 * and postprocessing_vza_ms! (tools/postprocessing_vza_ms.jl:9-77) weights them azimuthally.
 * uwJ, dwJ: host [nVza, nStokes, nSpec, nSensors] (the reference's vector-of-arrays, sensor-major).  Synchronous.
 * Float64, noRS; the hdr / bhr outputs of mom_get_hdr are not defined after this call. */
int mom_rt_run_multisensor(mom_t *h, int nSensors, const int *sensor_levels, double *uwJ, double *dwJ);

/* R_SFI, T_SFI [nVza, nStokes, S] to host (synchronises). */
int mom_get_RT(mom_t *h, double *R_SFI, double *T_SFI);
/* The RAMI extras of rt_run's return tuple (rt_run.jl:226): hdr [nVza, nStokes, S] from
 * interaction_hdrf! (CoreKernel/interaction_hdrf.jl:9-45) + postprocessing_vza_hdrf!
 * (postprocessing_vza.jl:63-93), and the up-/down-welling flux sums bhr_uw, bhr_dw [nStokes, S]
 * (the reference returns their first rows). */
int mom_get_hdr(mom_t *h, double *hdr, double *bhr_uw, double *bhr_dw);
/* Same, into caller-owned DEVICE buffers (asynchronous on the handle's stream; errors surface at the next
 * synchronising call). */
int mom_get_RT_device(mom_t *h, void *dR_SFI, void *dT_SFI);

/* ---- multi-GPU: one process per GPU, spectral shards, ONE RCCL all-gather ----------
 * The reference has no multi-GPU path; a sharded host (Julia Distributed/MPI.jl ranks, or this repository's
 * bench.py) gives every rank a contiguous slice of the spectral axis with the GLOBAL ndoubl / interface codes
 * (rt_kernel.jl:241-242 takes maxima over the whole axis) and gathers the spectra once at the end.
 *   mom_comm_unique_id   rank 0 obtains the RCCL id (MOM_COMM_ID_BYTES bytes) and hands it to the other ranks
 *                        by whatever means the host has (MPI_Bcast, a TCP store, a file)
 *   mom_comm_init        ncclCommInitRank on the handle's GPU; librccl.so.1 is loaded on first use
 *   mom_allgather        count doubles per rank, device pointers, asynchronous on the handle's stream
 *   mom_allgather_RT_device  the handle's R_SFI || T_SFI block (2*nVza*nStokes*S_loc doubles) of every rank into
 *                        d_global [nranks][2][nVza*nStokes*S_loc]; asynchronous
 *   mom_allgather_RT     the same into host arrays R_SFI, T_SFI [nVza, nStokes, nranks*S_loc] (rank-major
 *                        spectral axis: every rank must run the same S_loc, pad the tail); synchronises */
enum { MOM_COMM_ID_BYTES = 128 };
int mom_comm_unique_id(void *id_out, size_t bytes);
int mom_comm_init(mom_t *h, int rank, int nranks, const void *nccl_id);
int mom_comm_destroy(mom_t *h);
int mom_allgather(mom_t *h, const void *d_local, void *d_global, size_t count);
int mom_allgather_RT_device(mom_t *h, void *d_global);
int mom_allgather_RT(mom_t *h, double *R_SFI_global, double *T_SFI_global);

/* Per-stage GPU time of the last mom_rt_run in milliseconds (hipEvent based):
 * ms[0] = layer kernels, ms[1] = surface, ms[2] = post-processing, ms[3] = total;
 * with n >= 8 also ms[4] = sum over the full-problem layer launches (k_layer of namespace mom, or the
 * only layer kernel when the m = 0 reduction is off), ms[5] = sum over the reduced m = 0 launches,
 * ms[6], ms[7] = their launch counts.  kernel_launches = number of layer-kernel launches.  Synchronises. */
int mom_timers(mom_t *h, double *ms, int n, int *kernel_launches);

/* Tuning / test knobs (call before mom_scene_set):
 *   MOM_OPT_INVERSE       0 = automatic (default), 1 = force the pivoted Gauss-Jordan inverse, 2 = automatic
 *                         without the strip-chained kernels' register-resident chains (A/B testing)
 *   MOM_OPT_FORCE_GENERIC 1 = use the generic (global-memory) kernels even when the LDS-resident ones apply
 *   MOM_OPT_M0_REDUCTION  1 (default) = run Fourier moment 0 on the (I,Q) sub-problem when the scene allows
 *                         it (see mom_scene_set), 0 = always the full nStokes problem
 *   MOM_OPT_SMALL_WG      1 (default) = operators small enough for two LDS images per CU run in 4-wave
 *                         workgroups, two per CU; 0 = always 8-wave workgroups
 *   MOM_OPT_STAGGER       1 (default) = the persistent one-per-CU workgroups of the strip-chained kernels start
 *                         with offsets spread over one unit time, so that the CUs' composite loads/stores do
 *                         not all fall into the same microseconds; 0 = all start together
 *   MOM_OPT_SMALL_N       1 (default) = operators of edge N <= 4 (with at most 4 view angles and 4 phase-matrix bases)
 *                         run the lane-per-spectral-point sweep kernel: the whole of mom_rt_run in one launch, all
 *                         operators in registers (csrc/mom_small.hip); edges 4 < N <= 32 (scattering in every layer after the
 *                         first, at most 256 view x Stokes outputs) the wave-per-spectral-point sweep kernel
 *                         (csrc/mom_wave.hip), with 3 (N = 5) or 2 (N = 6..8) spectral points per wavefront as diagonal
 *                         blocks of one MFMA tile; 2 = the same kernels with one point per wavefront throughout;
 *                         0 = the general workgroup-per-point kernels
 *   MOM_OPT_LAYER_SWEEP   1 (default) = one launch per problem size walks ALL layers of a (spectral point, moment)
 *                         unit before the next unit: the composite blocks stay in the storing CU's L2 between layers,
 *                         one tail per sweep instead of one per layer (needs one interface code for all layers >= 2,
 *                         else falls back); 0 = one launch per layer (per-layer timing for profiling)
 *   MOM_OPT_STRIP_PAD     1 (default) = an operator edge N (or the m = 0 sub-problem's) without a strip-chained kernel
 *                         image is padded with up to 4 decoupled dummy stream entries (mu = 1, weight 0, zero
 *                         phase-matrix rows and columns) when that reaches a size with one (36, 40, 44, 52, 56, 60);
 *                         results for the real streams are unchanged; 0 = run the edge as given
 *   MOM_OPT_LEAN          operators of edge 36 / 40 (Float64, layer-sweep mode, ScatteringInterface_11 after the first layer; the
 *                         m = 0 (I,Q) sub-problem of a 20-stream IQU scene is one): which image runs them before the full image
 *                         finishes whatever it left (series beyond 12 terms) -- TWO launches per sweep, both counted in mom_timers'
 *                         kernel_launches.  3 (default, r6) = the QUAD-BLOCK image: one wavefront per unit, products on
 *                         v_mfma_f64_4x4x4 (no padding at N = 40, no idle SIMD), four units per CU (csrc/mom_q4.hpp); needs two or
 *                         more Stokes components per stream, else falls to 1; 1 = the four-wave LEAN strip image (three operator
 *                         buffers, 168 registers: three workgroups per CU; bitwise the full image's strip path); 2 = the six-wave lean
 *                         image (doubling chains on half-strips: an experiment, measured slower); 0 = the full image only.
 *   MOM_OPT_OVERLAP       1 (default) = when Fourier moment 0 runs on the (I,Q) sub-problem (MOM_OPT_M0_REDUCTION) in layer-sweep
 *                         mode AND the full problem runs on a one-workgroup-per-CU strip image (operator edge 52, 56, 60: the two
 *                         kernels then cannot share a CU; where they can, overlapping them measured slower), its launches and its surface interaction go to a second, high-priority stream of the handle and
 *                         overlap the launch of moments 1..M-1: the partial last round of the persistent workgroups of either
 *                         launch is filled by the other; results are unchanged (the launches are independent); mom_timers then
 *                         reports overlapping intervals for the two.  0 = everything on the handle's one stream.
 *   MOM_OPT_RRS_KERNELS   kernel forms of the rotational-Raman path, a mask (default 15; every form executes the same products in
 *                         the same order, results agree to 1e-13): 1 = one workgroup per (n1, dn) pair above N = 16 (else one
 *                         wavefront per pair), 2 = ... also for 16 < N <= 32, 4 = one workgroup per spectral point above N = 16 for
 *                         the per-point operands (Gauss-Jordan inverse over all waves), 8 = the inelastic elemental layer as a tile
 *                         kernel, 16 / 32 = form it inside the first doubling step always / never (neither: above N = 48 only).
 *                         (r5 read these from environment variables, once per process.)
 *   MOM_OPT_DUAL_WORKSPACE_MB   operator workspace of mom_rt_run_dual in MiB (0, default: 60 % of the free HBM when the run
 *                         starts); a scene that needs more is processed in chunks of spectral points.
 */
int mom_set_option(mom_t *h, int option, int value);
enum { MOM_OPT_INVERSE = 0, MOM_OPT_FORCE_GENERIC = 1, MOM_OPT_M0_REDUCTION = 2, MOM_OPT_SMALL_WG = 3, MOM_OPT_STAGGER = 4,
       MOM_OPT_SMALL_N = 5, MOM_OPT_LAYER_SWEEP = 6, MOM_OPT_STRIP_PAD = 7, MOM_OPT_LEAN = 8, MOM_OPT_OVERLAP = 9,
       MOM_OPT_RRS_KERNELS = 10, MOM_OPT_DUAL_WORKSPACE_MB = 11 };

/* ---- Voigt line-by-line cross section --------------------------------------------------
 * compute_absorption_cross_section(model::HitranModel, grid, p, T)
 * (compute_absorption_cross_section.jl:19-130) with Voigt broadening and the
 * HumlicekWeidemann32SDErrorFunction (complex_error_functions.jl:226-234).  The host passes
 * the per-line prefactors of :79-107: nu = pressure-shifted centre, gamma_d, y, S =
 * temperature-corrected strength, and the 1-based inclusive grid window of each line.
 * sigma[nGrid] (host) receives sum over lines in line order.  `device` as in mom_create.
 * Error text: mom_last_global_error(). */
int mom_voigt_xsec(int device, int nLines, const double *nu, const double *gamma_d, const double *y,
                   const double *S, const int *ind_start_1based, const int *ind_stop_1based, int nGrid,
                   const double *grid, double *sigma);

/* GPU time (HIP events around the kernel, ms) of the last mom_voigt_xsec call of the calling thread. */
double mom_voigt_last_kernel_ms(void);

#ifdef __cplusplus
}
#endif
#endif /* MOMCORE_H */
