# integration/MomCoreRT.jl -- the hand-written Julia host layer above integration/MomCore.jl (the generated `ccall`
# wrappers): the `MI355X <: AbstractArchitecture` methods a vSmartMOM.jl maintainer adds so that
# parameters_from_yaml() / model_from_parameters() / rt_run() keep their surface and the CoreRT layer loop runs in
# libmomcore.so.  Included from src/CoreRT/CoreRT.jl after `include("MomCore.jl")`.
#
# NOT executed in the build image (Julia is absent there).  What IS machine-checked (tests/test_julia_bindings.py):
# every `mom_*` call below names a wrapper that integration/MomCore.jl generates from include/momcore.h, with exactly
# the number of arguments the header declares; the Python twin of each function (radiativetransfer.jl_amd/corert.py,
# same call sequence through ctypes) is what the GPU parity tests execute.
#
# Seams in the reference: src/Architectures.jl:20-55 (architecture types, array_type, devi),
# src/CoreRT/rt_run.jl:19-21 (rt_run(model)), :41-230 (rt_run(RS_type, model, iBand)), CoreKernel/rt_kernel.jl:173-183,
# CoreKernel/elemental.jl:109-118, CoreKernel/doubling.jl:81-86, CoreKernel/interaction_inelastic.jl:474-477,
# rt_run_multisensor.jl:14-191, tools/atmo_prof.jl:427-449.

using .MomCore
using .MomCore: momcheck, MomError

# >>> architecture
# src/Architectures.jl (add): one more architecture; CPU() / GPU() behaviour is untouched because Julia picks the most
# specific method.  Host arrays stay `Array`: device memory is owned by the mom_t handle.
struct MI355X <: AbstractArchitecture
    device::Cint
end
MI355X() = MI355X(0)
array_type(::MI355X) = Array
devi(::MI355X)       = KernelAbstractions.CPU()   # never used on this path
# src/CoreRT/tools/parameters_from_yaml.jl:97-98 -- add "Architectures.MI355X()" to the whitelist of `architecture:`
# <<< architecture

# >>> handle
"""One handle <-> one GPU <-> one HIP stream (include/momcore.h); destroyed by `close` or the finalizer."""
mutable struct MomHandle
    ptr::Ptr{Cvoid}
    N::Int; nStokes::Int; nSpec::Int; max_m::Int
    function MomHandle(arch::MI355X, N, nStokes, nSpec, max_m; float_type = Float64)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        momcheck(MomCore.mom_create(out, arch.device, N, nStokes, nSpec, max_m, float_type === Float32 ? 1 : 0))
        h = new(out[], N, nStokes, nSpec, max_m)
        finalizer(close, h)
        return h
    end
end
function Base.close(h::MomHandle)
    h.ptr == C_NULL && return
    MomCore.mom_destroy(h.ptr)
    h.ptr = C_NULL
    return
end
# <<< handle

iface_code(::ScatteringInterface_00) = Cint(0)
iface_code(::ScatteringInterface_01) = Cint(1)
iface_code(::ScatteringInterface_10) = Cint(2)
iface_code(::ScatteringInterface_11) = Cint(3)

# >>> scene
"""
Host preparation of rt_run.jl:43-138 (unchanged Julia, runs on `Array`) and the upload of the scene: streams,
layer optics in the library's native input form (K phase-matrix bases + per-point weights instead of the N×N×nSpec
array of expandOpticalProperties, compEffectiveLayerProperties.jl:124-135), doubling numbers, interface codes,
surface.  Returns the handle with everything resident in HBM.
"""
function momcore_scene(RS_type, model::vSmartMOM_Model, iBand, arch::MI355X; strict_reference_indexing::Bool = true, split = nothing)
    @unpack qp_μ, qp_μN, wt_μN, iμ₀, μ₀ = model.quad_points
    pol   = model.params.polarization_type
    max_m = model.params.max_m
    N     = length(qp_μN)
    nSpec = sum(size(model.τ_abs[iB], 1) for iB in iBand)
    props = constructCoreOpticalProperties(RS_type, iBand, 0, model)[1]                    # τ, ϖ do not depend on m
    ifaces, τ_sum = extractEffectiveProps(props, model.quad_points)
    nd    = Cint[get_dtau_ndoubl(l, model.quad_points)[2] for l in props]                  # GLOBAL maxima (rt_kernel.jl:241)
    τ     = reduce(hcat, [l.τ for l in props]);  ϖ = reduce(hcat, [l.ϖ for l in props])   # [nSpec, Nz]
    Zpp, Zmp, zw = z_bases_and_weights(RS_type, iBand, model)                              # [N,N,K,M] ×2, [K,nSpec,Nz]
    node  = Cint[nearest_point(qp_μ, cosd(v)) for v in model.obs_geom.vza]
    cosm  = [cosd(m * a) for a in model.obs_geom.vaz, m in 0:max_m-1]
    sinm  = [sind(m * a) for a in model.obs_geom.vaz, m in 0:max_m-1]
    brdf  = model.params.brdf[iBand[1]]

    # Dual models (rt_run_dual below): `split = (val, part)` takes the values for the uploads here and returns the partials
    val, part = split === nothing ? (identity, A -> nothing) : split
    h = MomHandle(arch, N, pol.n, nSpec, max_m; float_type = model.params.float_type)
    MomCore.mom_set_streams!(h.ptr, qp_μN, wt_μN, N, iμ₀, μ₀, pol.I₀, pol.D, strict_reference_indexing ? 1 : 0)
    albedo = brdf isa LambertianSurfaceScalar ? brdf.albedo : zero(eltype(τ))
    MomCore.mom_scene_set!(h.ptr, length(nd), size(zw, 1), max_m, val(τ), val(ϖ), val(zw), val(Zpp), val(Zmp), nd,
                           iface_code.(ifaces), val(τ_sum), Float64(val([albedo])[1]), length(node), node, cosm, sinm)
    dRsurf = dalb = nothing
    if brdf isa LambertianSurfaceLegendre                       # lambertian_surface.jl:90-96: spectral albedo
        alb = legendre_albedo(brdf, model, iBand)
        MomCore.mom_scene_set_surface!(h.ptr, 2, max_m, C_NULL, val(alb));  dalb = part(alb)
    elseif !(brdf isa LambertianSurfaceScalar)                  # rpvSurfaceScalar, RossLiSurfaceScalar: BRDF Fourier moments
        Rsurf = cat([(m == 0 ? 2 : 1) * reflectance(brdf, pol, Array(qp_μ), m) for m in 0:max_m-1]...; dims = 3)
        MomCore.mom_scene_set_surface!(h.ptr, 1, max_m, val(Rsurf), C_NULL);  dRsurf = part(Rsurf)
    end
    split === nothing && return h
    return h, (dτ = part(τ), dϖ = part(ϖ), dzw = part(zw), dZpp = part(Zpp), dZmp = part(Zmp), dalbedo = part([albedo]),
               dRsurf = dRsurf, dalbedo_spec = dalb)
end
# <<< scene

# >>> rt_run_noRS
"""rt_run(RS_type::noRS, model, iBand) on the MI355X: the layer loop of rt_run.jl:125-215 is ONE call."""
function rt_run(RS_type::noRS, model::vSmartMOM_Model, iBand, arch::MI355X)
    h = momcore_scene(RS_type, model, iBand, arch)
    try
        pol, nV, nSpec = model.params.polarization_type, length(model.obs_geom.vza), h.nSpec
        MomCore.mom_rt_run!(h.ptr)                                    # elemental! -> doubling! -> interaction!, surface, post-processing
        R_SFI = zeros(nV, pol.n, nSpec);  T_SFI = similar(R_SFI)      # rt_run.jl:89-90 layout
        MomCore.mom_get_RT!(h.ptr, R_SFI, T_SFI)
        hdr = similar(R_SFI);  bhr_uw = zeros(pol.n, nSpec);  bhr_dw = similar(bhr_uw)    # rt_run.jl:91-94
        MomCore.mom_get_hdr!(h.ptr, hdr, bhr_uw, bhr_dw)
        return R_SFI, T_SFI, zero(R_SFI), zero(R_SFI), hdr, bhr_uw[1, :], bhr_dw[1, :]    # the 7-tuple of rt_run.jl:226
    finally
        close(h)
    end
end

# >>> rt_run_dual
"""
rt_run on a model whose optical properties are ForwardDiff.Dual (the reference's Jacobian route: rt_run.jl:89-96 allocates
R, T, R_SFI, T_SFI in the Dual type and the whole layer loop runs on Dual arrays, gpu_batched.jl:100-150).  The host
preparation above runs unchanged on the Dual arrays; values and partials are separated at the boundary
(`ForwardDiff.value` / `ForwardDiff.partials`), uploaded with mom_scene_set / mom_scene_set_partials, and the results are
re-assembled into Dual arrays of the caller's tag: every downstream `ForwardDiff.jacobian` sees exactly what the CPU run returns.
"""
function rt_run_dual(RS_type::noRS, model::vSmartMOM_Model, iBand, arch::MI355X, ::Type{D}) where {T, V, P, D <: ForwardDiff.Dual{T, V, P}}
    val(A)     = ForwardDiff.value.(A)
    part(A)    = cat((ForwardDiff.partials.(A, i) for i in 1:P)...; dims = ndims(A) + 1)   # partial index = slowest axis
    h, dual_in = momcore_scene(RS_type, model, iBand, arch; split = (val, part))            # τ, ϖ, zw, Z bases, surface as Duals
    try
        pol, nV, nSpec = model.params.polarization_type, length(model.obs_geom.vza), h.nSpec
        nn(x) = x === nothing ? C_NULL : x                            # an input without partials: NULL
        MomCore.mom_scene_set_partials!(h.ptr, P, dual_in.dτ, dual_in.dϖ, dual_in.dzw, dual_in.dZpp, dual_in.dZmp,
                                        dual_in.dalbedo, nn(dual_in.dRsurf), nn(dual_in.dalbedo_spec))
        MomCore.mom_rt_run_dual!(h.ptr)
        R = zeros(nV, pol.n, nSpec);  Tr = similar(R);  dR = zeros(nV, pol.n, nSpec, P);  dT = similar(dR)
        MomCore.mom_get_RT!(h.ptr, R, Tr)
        MomCore.mom_get_RT_partials!(h.ptr, dR, dT)
        hdr = similar(R);  uw = zeros(pol.n, nSpec);  dw = similar(uw)                     # the RAMI extras of rt_run.jl:91-94
        dhdr = similar(dR);  duw = zeros(pol.n, nSpec, P);  ddw = similar(duw)
        MomCore.mom_get_hdr!(h.ptr, hdr, uw, dw)
        MomCore.mom_get_hdr_partials!(h.ptr, dhdr, duw, ddw)
        mk(X, dX) = [D(X[i], ForwardDiff.Partials(ntuple(p -> dX[i, p], P))) for i in CartesianIndices(X)]
        Rd, Td = mk(R, dR), mk(Tr, dT)
        return Rd, Td, zero(Rd), zero(Rd), mk(hdr, dhdr), mk(uw, duw)[1, :], mk(dw, ddw)[1, :]   # the 7-tuple of rt_run.jl:226
    finally
        close(h)
    end
end
# <<< rt_run_dual

# The architecture is a FIELD of model.params (vSmartMOM_Parameters), not a type parameter of vSmartMOM_Model
# (src/CoreRT/types.jl:483), so the two entry points of rt_run.jl:19-21,41-42 gain a run-time branch and the reference's
# own body moves, unchanged, into `rt_run_reference`:
function rt_run(model::vSmartMOM_Model; i_band::Integer = 1)
    rt_run(noRS(), model, i_band)
end
function rt_run(RS_type::AbstractRamanType, model::vSmartMOM_Model, iBand)
    arch = model.params.architecture
    return arch isa MI355X ? rt_run(RS_type, model, iBand, arch) : rt_run_reference(RS_type, model, iBand)
end
# <<< rt_run_noRS

# >>> rt_run_RRS
"""
rt_run(RS_type::RRS, model, iBand) on the MI355X (src/Inelastic/types.jl:13-33): the host side is the reference's own
(getRamanSSProp!, computeRamanZλ!, constructCoreOpticalProperties stay Julia), only the layer loop moves.
`rrs_strict_reference = true` executes the reference's RRS text as written (MOM_EUNSUPPORTED where the reference itself
raises); `false` applies the corrections D1..D5 of DESIGN.md section 7.
"""
function rt_run(RS_type::RRS, model::vSmartMOM_Model, iBand, arch::MI355X; rrs_strict_reference::Bool = true)
    pol, max_m = model.params.polarization_type, model.params.max_m
    qp_μ = model.quad_points.qp_μ
    h = momcore_scene_rrs(RS_type, model, iBand, arch, rrs_strict_reference)
    try
        nV, nSpec = length(model.obs_geom.vza), h.nSpec
        MomCore.mom_rt_run_rrs!(h.ptr)
        R, T, ieR, ieT = (zeros(nV, pol.n, nSpec) for _ in 1:4)
        MomCore.mom_get_RT_rrs!(h.ptr, R, T, ieR, ieT, C_NULL)
        hdr = zeros(nV, pol.n, nSpec);  bhr_uw = zeros(pol.n, nSpec);  bhr_dw = similar(bhr_uw)
        MomCore.mom_get_hdr_rrs!(h.ptr, hdr, bhr_uw, bhr_dw)
        return R, T, ieR, ieT, hdr, bhr_uw[1, :], bhr_dw[1, :]
    finally
        close(h)
    end
end

function momcore_scene_rrs(RS_type::RRS, model, iBand, arch::MI355X, strict::Bool; shard = nothing)
    pol, max_m = model.params.polarization_type, model.params.max_m
    qp_μ = model.quad_points.qp_μ
    # the elastic scene exactly as for noRS (the Rayleigh ϖ is RS_type.ϖ_Cabannes, compEffectiveLayerProperties.jl:27);
    # the Raman layers are allocated for the unpadded operator edge
    h = momcore_scene(RS_type, model, iBand, arch)
    MomCore.mom_set_option!(h.ptr, MomCore.MOM_OPT_STRIP_PAD, 0)
    MomCore.mom_rrs_set!(h.ptr, length(RS_type.i_λ₁λ₀), Cint.(RS_type.i_λ₁λ₀), RS_type.ϖ_λ₁λ₀, strict ? 1 : 0)
    if shard !== nothing            # (nSpec_global, wlo, lo, hi), 0-based owned slice [lo, hi) of the window starting at wlo
        nSpec_global, wlo, lo, hi = shard
        MomCore.mom_rrs_set_shard!(h.ptr, nSpec_global, wlo, lo - wlo, hi - wlo)
    end
    # fScattRayleigh [nSpec, Nz] (compEffectiveLayerProperties.jl:58, expandBandScalars :150-158) and the Raman phase
    # matrices of every Fourier moment (computeRamanZλ!, inelastic_helper.jl:457-464)
    fScattRayleigh = constructCoreOpticalProperties(RS_type, iBand, 0, model)[2]
    fsc  = reduce(hcat, [expandBandScalars(RS_type, f) for f in fScattRayleigh])
    Zr   = [Scattering.compute_Z_moments(pol, Array(qp_μ), RS_type.greek_raman, m) for m in 0:max_m-1]
    Zrpp = cat((z[1] for z in Zr)...; dims = 3);  Zrmp = cat((z[2] for z in Zr)...; dims = 3)
    MomCore.mom_scene_set_rrs!(h.ptr, fsc, Zrpp, Zrmp)
    return h
end
# <<< rt_run_RRS

# >>> rt_run_ms
"""rt_run_test_ms(RS_type::noRS, sensor_levels, model, iBand) (rt_run_multisensor.jl:14-191): ONE call after the scene is resident."""
function rt_run_test_ms(RS_type::noRS, sensor_levels::Vector{Int64}, model::vSmartMOM_Model, iBand, arch::MI355X)
    h = momcore_scene(RS_type, model, iBand, arch)
    try
        nV, n, nSpec, ns = length(model.obs_geom.vza), model.params.polarization_type.n, h.nSpec, length(sensor_levels)
        uw = zeros(Float64, nV, n, nSpec, ns);  dw = similar(uw)
        MomCore.mom_rt_run_multisensor!(h.ptr, ns, Cint.(sensor_levels), uw, dw)
        uwJ = [uw[:, :, :, i] for i in 1:ns];  dwJ = [dw[:, :, :, i] for i in 1:ns]
        return uwJ, dwJ, zero.(uwJ), zero.(dwJ)          # (uwJ, dwJ, uwieJ, dwieJ), rt_run_multisensor.jl:190
    finally
        close(h)
    end
end
# <<< rt_run_ms

# >>> operators
# Alternative A: keep rt_run's loops in Julia and overload the operators one by one.  The layer state lives in the
# handle; AddedLayer / CompositeLayer become tags that carry it.
struct MomAddedLayer;     h::MomHandle; surface::Bool; end
struct MomCompositeLayer; h::MomHandle; end

# elemental!(pol_type, SFI, τ_sum, dτ, computed_layer_properties, m, ndoubl, scatter, quad_points, added_layer, architecture)
# -- elemental.jl:109-118
function elemental!(pol_type, SFI, τ_sum, dτ, computed_layer_properties, m, ndoubl, scatter, quad_points,
                    added_layer::MomAddedLayer, architecture::MI355X)
    @unpack ϖ, Z⁺⁺, Z⁻⁺ = computed_layer_properties
    MomCore.mom_elemental!(added_layer.h.ptr, m, ndoubl, τ_sum, dτ, ϖ, Z⁺⁺, Z⁻⁺, size(Z⁺⁺, 3))
end
# doubling!(pol_type, SFI, expk, ndoubl, added_layer, I_static, architecture) -- doubling.jl:81-86 (expk updated in place)
doubling!(pol_type, SFI, expk, ndoubl, added_layer::MomAddedLayer, I_static, architecture::MI355X) =
    MomCore.mom_doubling!(added_layer.h.ptr, ndoubl, expk)
# interaction!(RS_type::noRS, scattering_interface, SFI, composite_layer, added_layer, I_static) -- interaction_inelastic.jl:474-477
interaction!(RS_type::noRS, scattering_interface, SFI, composite_layer::MomCompositeLayer, added_layer::MomAddedLayer, I_static) =
    MomCore.mom_interaction!(composite_layer.h.ptr, iface_code(scattering_interface), added_layer.surface ? 1 : 0)
# rt_kernel.jl:227-230: composite_layer.X[:] = added_layer.x for the first layer
copy_added_to_composite!(composite_layer::MomCompositeLayer) = MomCore.mom_copy_added_to_composite!(composite_layer.h.ptr)
# create_surface_layer!(brdf::LambertianSurfaceScalar, added_layer, SFI, m, pol_type, quad_points, τ_sum, architecture)
# -- lambertian_surface.jl:20-27
create_surface_layer!(brdf::LambertianSurfaceScalar, added_layer::MomAddedLayer, SFI, m, pol_type, quad_points, τ_sum,
                      architecture::MI355X) =
    MomCore.mom_surface_lambertian!(added_layer.h.ptr, m, Float64(brdf.albedo), τ_sum)
# postprocessing_vza!(RS_type, iμ₀, pol_type, composite_layer, vza, qp_μ, m, vaz, μ₀, weight, nSpec, SFI, R, R_SFI, T, T_SFI,
#                     ieR_SFI, ieT_SFI) -- postprocessing_vza.jl:9-11 (accumulates in place)
function postprocessing_vza!(RS_type::noRS, iμ₀, pol_type, composite_layer::MomCompositeLayer, vza, qp_μ, m, vaz, μ₀, weight,
                             nSpec, SFI, R, R_SFI, T, T_SFI, ieR_SFI, ieT_SFI)
    node = Cint[nearest_point(qp_μ, cosd(v)) for v in vza]
    MomCore.mom_postprocess!(composite_layer.h.ptr, m, length(vza), node, Float64.(vaz), weight, R_SFI, T_SFI)
end
# batch_inv!(X, A) / A ⊠ B -- gpu_batched.jl:60,85,90-97
batch_inv!(h::MomHandle, X::Array{Float64,3}, A::Array{Float64,3}) = MomCore.mom_batch_inv!(h.ptr, size(A, 1), size(A, 3), A, X)
batched_mul!(h::MomHandle, C::Array{Float64,3}, A::Array{Float64,3}, B::Array{Float64,3}) =
    MomCore.mom_batched_mul!(h.ptr, size(A, 1), size(A, 3), A, B, C)
# Array(composite_layer.J₀⁻) (postprocessing_vza.jl:17-20) and friends
download!(dst::Array{Float64}, h::MomHandle, which::Integer) = MomCore.mom_download!(h.ptr, which, dst)
upload!(h::MomHandle, which::Integer, src::Array{Float64}) = MomCore.mom_upload!(h.ptr, which, src)
# <<< operators

# >>> multigpu
"""
One process per GPU (Julia Distributed / MPI.jl).  Rank r of G runs the binding on its slice of τ, ϖ, zw, τ_sum while
nd / ifaces stay the GLOBAL ones (get_dtau_ndoubl and extractEffectiveProps take maxima over the whole spectral axis),
then ONE RCCL all-gather of the spectra -- behind the C ABI, on the library's own stream.
"""
function momcore_comm_init!(h::MomHandle, r::Integer, G::Integer, bcast!)
    id = Vector{UInt8}(undef, MomCore.MOM_COMM_ID_BYTES)
    r == 0 && momcheck(MomCore.mom_comm_unique_id(id, length(id)))
    bcast!(id)                                                   # any host-side broadcast will do (MPI.Bcast!(id, 0, comm))
    MomCore.mom_comm_init!(h.ptr, r, G, id)
end
function momcore_allgather_RT(h::MomHandle, nV, nStokes, S_loc, G)
    R_SFI = zeros(nV, nStokes, G * S_loc);  T_SFI = similar(R_SFI)
    MomCore.mom_allgather_RT!(h.ptr, R_SFI, T_SFI)
    return R_SFI, T_SFI
end
# <<< multigpu

# >>> absorption
"""
compute_absorption_profile! (src/CoreRT/tools/atmo_prof.jl:427-449) for a HitranModel with Voigt broadening: one
resident line table per absorber, then all layers of the profile in two launches (the reference: a loop over layers
with one kernel launch per line, compute_absorption_cross_section.jl:118-124).
"""
function compute_absorption_profile!(h::MomHandle, grid, hitran, tips, p_full, T, vmr, vcd_dry, wing_cutoff)
    Nz = length(p_full)
    MomCore.mom_absorption_begin!(h.ptr, Nz, collect(Float64, grid))
    keep = (minimum(grid) - wing_cutoff) .< hitran.νᵢ .< (maximum(grid) + wing_cutoff)      # compute_absorption_cross_section.jl:54-72
    MomCore.mom_absorption_set_lines!(h.ptr, count(keep), hitran.νᵢ[keep], hitran.Sᵢ[keep], hitran.γ_air[keep], hitran.γ_self[keep],
                                      hitran.E″[keep], hitran.n_air[keep], hitran.δ_air[keep], tips.sqrt_mol_weight[keep],
                                      tips.iso_index[keep], tips.nIso, tips.nTmax, tips.nT, tips.T, tips.Q, tips.z)
    ms = Ref{Cdouble}(0)
    MomCore.mom_voigt_tau_abs_profile!(h.ptr, Nz, p_full, T, Float64(vmr), Float64(wing_cutoff), vcd_dry .* vmr, ms)
    return ms[]      # τ_abs stays resident; mom_scene_set_optics assembles the layer optics from it
end

"""compute_absorption_cross_section(model::HitranModel, grid, p, T) with architecture isa MI355X: ONE call for all lines."""
function voigt_xsec(arch::MI355X, ν, γ_d, y, S, ind_start, ind_stop, grid)
    result = zeros(Float64, length(grid))
    momcheck(MomCore.mom_voigt_xsec(arch.device, length(ν), ν, γ_d, y, S, Cint.(ind_start), Cint.(ind_stop), length(grid),
                                    collect(Float64, grid), result))
    return result
end
# <<< absorption
