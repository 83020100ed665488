// mom_rrs.hpp -- host interface of the rotational-Raman (RRS) kernels (mom_rrs.hip) towards momcore.hip.
//
// State of the inelastic path of rt_run(::RRS) (src/CoreRT/rt_run.jl:41-230): the reference's AddedLayerRS / CompositeLayerRS
// (types.jl:145-205) as persistent HBM arrays in the reference's own memory order -- elastic operators [N,N,S], sources
// [N,S], inelastic operators [N,N,S,nRaman], inelastic sources [N,S,nRaman] (Julia column-major) -- plus per-point scratch
// operands of the current doubling step / interaction.  All device memory is owned by State.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

namespace momr {

struct Streams {
  const double *mu, *wt;  // [N] device
  double I0[4], D[4];
  int N, nS, imu0, strict_idx;
  double mu0;
};

// indices into State::added / surf / comp
enum { R_PM = 0, R_MP = 1, T_MM = 2, T_PP = 3, J0P = 4, J0M = 5 };        // AddedLayer field order of momcore.h
enum { C_R_MP = 0, C_R_PM = 1, C_T_PP = 2, C_T_MM = 3, C_J0P = 4, C_J0M = 5 };  // CompositeLayer field order

// kernel-form switches (State::kopt; MOM_OPT_RRS_KERNELS): every form computes the same products in the same order
enum { KOPT_WG = 1, KOPT_WG2 = 2, KOPT_WG_POINT = 4, KOPT_EL_TILE = 8, KOPT_EL_FUSE_ON = 16, KOPT_EL_FUSE_OFF = 32,
       KOPT_DEFAULT = KOPT_WG | KOPT_WG2 | KOPT_WG_POINT | KOPT_EL_TILE };

struct State {
  int N = 0, nS = 0, S = 0, nR = 0;
  int kopt = KOPT_DEFAULT;
  int P = 16;              // row pitch of every device block: 16 (N <= 16) or 32; matrices P x P, vectors P, zero padding --
                           // the tile loads of the kernels are then unmasked, 128-byte aligned and at immediate offsets
  int strict_rrs = 1;      // rrs_strict_reference (DESIGN.md "RRS": D1..D5)
  int n_glob0 = 0;         // global 0-based spectral index of local index 0 (shards; strict D2/D3 use absolute indices)
  int n1_lo = 0, n1_hi = 0;  // spectral points this rank owns: pairs (n1, dn) are processed for n1 in [n1_lo, n1_hi)
  hipStream_t stream = nullptr;
  int *d_off = nullptr;         // [nR] i_l1l0
  double *d_varpiR = nullptr;   // [nR]
  int max_off = 0;
  // elastic added layer, double-buffered over the doubling steps (cur = buffer holding the current state)
  double *added[2][6] = {};     // R_PM / T_MM exist once (added[0]) and are written after the doubling
  double *expk[2] = {};
  int cur = 0;
  double *surf[6] = {};
  double *comp[2][6] = {};
  int ccur = 0;
  // per-point scratch operands: up to 10 matrices [N,N,S], 6 vectors [N,S], the j0+ sequence of the strict position
  double *smat[10] = {};
  double *svec[6] = {};
  double *jpseq = nullptr;      // [N,S,nR]
  // inelastic layers
  double *ie_added[6] = {};     // ier+-, ier-+, iet--, iet++ [N,N,S,nR]; ieJ0+, ieJ0- [N,S,nR]   (R_PM.. order)
  double *ie_comp[6] = {};      // ieR-+, ieR+-, ieT++, ieT-- ; ieJ0+, ieJ0-                     (C_R_MP.. order)
  double *d_out = nullptr;      // R_SFI | T_SFI | ieR_SFI | ieT_SFI | hdr [5][nVza,nS,S], then bhr_uw | bhr_dw [2][nS,S]
  int out_nVza = 0;
  int *d_info = nullptr;
  double *d_stage = nullptr;    // ABI-order staging of upload / download
  size_t stage_cap = 0;
  // fast scene-level mode (mom_rt_run_rrs): the inelastic elemental of a layer with ndoubl >= 1 is deferred into the first
  // doubling step (el_pending + its inputs), and in the corrected position ier+- / iet-- are not stored by the doubling but
  // derived where they are read (pm_derivable); pm_valid says whether the arrays themselves hold the current values
  bool fast = false, el_pending = false, pm_valid = true, pm_derivable = false;
  // true once an operator-level entry may have written the layers (uploads, mom_rrs_elemental ...) or a strict-position run
  // has ended: the next scene-level run zeroes the layers first (the reference allocates them zeroed per rt_run)
  bool dirty = false;
  struct { int m, nd, sh; const double *tau_sum, *tau, *varpi, *fscatt, *Zr_pp, *Zr_mp; } el{};
  // HIP-event pairs around the launches of the heavy kernels of the last run (timing_reset .. timing_read)
  std::vector<std::pair<double *, size_t>> guarded;  // layer arrays allocated between guard bands (mom_rrs.hip dmg): pointer, doubles
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_kind;  // kind of the launch bracketed by ev_pool[2k], ev_pool[2k+1]
  bool timing = false;
  std::string err;
};
enum { TK_DBL_PAIR = 0, TK_INT_PAIR = 1, TK_IE_ELEMENTAL = 2, TK_COUNT = 3 };
void timing_reset(State *s, bool on);
// ms[k], launches[k] for k < TK_COUNT (synchronises the stream)
hipError_t timing_read(State *s, double *ms, int *launches);

hipError_t create(State **out, hipStream_t st, int N, int nS, int S, int nR, const int *off_host, const double *varpi_host,
                  int strict_rrs);
void destroy(State *s);
// one layer array between the ABI's memory order (host, [N,N,nblk] or [N,nblk]) and the padded device blocks (synchronous)
hipError_t upload(State *s, double *dev, const double *host, bool matrix, size_t nblk);
hipError_t download(State *s, double *host, const double *dev, bool matrix, size_t nblk);

// elemental!(...) elastic part + elemental_inelastic!(::RRS) on the persistent added layer.  Zpp/Zmp: device [N,N] x nTerms with
// per-point weights zw [nTerms,S] (zw == nullptr: one term of weight 1); Zr*: device [N,N] Raman phase matrices; all
// spectral vectors device [S]; the elemental optical thickness is tau / 2^shift.
hipError_t elemental(State *s, const Streams &q, int m, int nd, int shift, const double *tau_sum, const double *tau, const double *varpi,
                     const double *Zpp, const double *Zmp, int nTerms, const double *zw, const double *fscatt,
                     const double *Zr_pp, const double *Zr_mp, bool elastic, bool inelastic);
// doubling_helper!(::RRS): nd steps on the persistent added layer (expk in s->expk[s->cur]); applies the D kernels at the end
hipError_t doubling(State *s, const Streams &q, int nd);
// rt_kernel.jl:326-333
hipError_t copy_added_to_composite(State *s, const Streams &q);
// writes ier+- / iet-- out if they are only derivable at the moment (before downloads, copies and operator-level reads)
hipError_t ensure_pm(State *s, const Streams &q);
// an upload replaced layer arrays: whatever is resident is what the arrays hold
inline void mark_uploaded(State *s) { s->pm_valid = true; s->pm_derivable = false; s->el_pending = false; }
// interaction_helper!(::RRS, iface): returns hipErrorInvalidValue with s->err set where the reference raises (strict position)
hipError_t interaction(State *s, const Streams &q, int iface, bool with_surface);
// create_surface_layer! into s->surf (ie* of the surface layer are zeros: never allocated).  kind 0 LambertianSurfaceScalar
// (albedo), 1 BRDF Fourier matrix Rsurf_m [N,N] of moment m (device), 2 LambertianSurfaceLegendre (albedo_spec [S], device)
hipError_t surface(State *s, const Streams &q, int m, int kind, double albedo, const double *tau_tot, const double *Rsurf_m,
                   const double *albedo_spec);
// number of nonzero (or NaN) entries in the zero padding of all layer arrays (the invariant the whole-tile stores rest on)
hipError_t count_padding(State *s, unsigned long long *out);
// start of a scene-level run: zeroed layers where needed (State::dirty, strict position), double-buffer / pm flags reset,
// output block zeroed (postprocessing_vza!(::RRS) accumulates into s->d_out)
hipError_t begin_run(State *s, int nVza);
hipError_t postprocess(State *s, const Streams &q, int m, int nVza, const int *d_node, const double *d_cos, const double *d_sin,
                       int M, double weight);

}  // namespace momr
