# NextGPHIP.jl -- reference-side binding of libnextgp_hip.so (include/nextgp_hip.h) for NextGP.jl.
#
# NEVER EXECUTED: the build container has no Julia toolchain (SURVEY.md section 8c), so this file has not been run once.  It is
# the stub a NextGP.jl maintainer adds next to src/samplers.jl; every ccall signature below is checked by hand against
# include/nextgp_hip.h, and the same entry points are exercised through Python ctypes (nextgp.jl_amd/_lib.py).  Two seams, both defined by the reference:
#
#   coarse:  replace `samplers.runSampler!(...)` (src/samplers.jl:23, called at src/MCMC.jl:39)
#            by `NextGPHIP.runSampler!(...)`: the whole chain runs on the GPU, the same *Out files
#            are written (src/samplers.jl:56-103).
#   fine:    replace the stored callback `M[set][:funct]` (src/mme.jl:326,333,355, invoked at
#            src/samplers.jl:52) by `NextGPHIP.sweep!`: fixed / random effects stay in Julia,
#            only the per-SNP sweep of one marker set runs on the GPU.
#
# Ownership: Julia arrays are passed for the duration of the ccall only (GC roots them); the
# library copies what it keeps.  All status codes != 0 become `error(msg)`, like src/mme.jl:77.
module NextGPHIP

using DelimitedFiles

const LIB = get(ENV, "NEXTGP_HIP_LIB", "libnextgp_hip")

mutable struct Handle
    ptr::Ptr{Cvoid}
end

function check(h::Handle, rc::Integer)
    rc == 0 && return
    msg = unsafe_string(ccall((:ngp_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr))
    error("libnextgp_hip ($rc): $msg")
end

function Handle(; device::Integer=0, seed::Integer=1, chain::Integer=0)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:ngp_create, LIB), Int32, (Int32, UInt64, UInt32, Ref{Ptr{Cvoid}}), device, seed, chain, out)
    rc == 0 || error("ngp_create ($rc): " * unsafe_string(ccall((:ngp_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    h = Handle(out[])
    finalizer(x -> ccall((:ngp_destroy, LIB), Int32, (Ptr{Cvoid},), x.ptr), h)
    return h
end

# M[set][:data] is the centred Float64 N x P matrix of src/prepMatVec.jl:129-131 (centre = 0)
set_panel!(h::Handle, data::Matrix{Float64}; centre::Bool=false) =
    check(h, ccall((:ngp_set_panel_f64, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int32),
                   h.ptr, data, size(data, 1), size(data, 2), stride(data, 2), centre))

# The same panel one marker set after another (M[s].data are separate matrices, src/mme.jl:296-311): no concatenated host copy.
# begin_panel!(h, N, Ptot); panel_columns!(h, col0, M[s].data) for every set (col0 0-based); end_panel!(h) builds mpm and the Gram window.
begin_panel!(h::Handle, N::Integer, P::Integer) = check(h, ccall((:ngp_begin_panel, LIB), Int32, (Ptr{Cvoid}, Int64, Int64), h.ptr, N, P))
panel_columns!(h::Handle, col0::Integer, data::Matrix{Float64}; centre::Bool=false) =
    check(h, ccall((:ngp_panel_columns_f64, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Int64, Int32),
                   h.ptr, col0, data, size(data, 2), stride(data, 2), centre))
panel_columns!(h::Handle, col0::Integer, data::Matrix{Float32}; centre::Bool=false) =
    check(h, ccall((:ngp_panel_columns_f32, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Float32}, Int64, Int64, Int32),
                   h.ptr, col0, data, size(data, 2), stride(data, 2), centre))
panel_columns!(h::Handle, col0::Integer, data::Matrix{UInt8}; centre::Bool=true) =      # genotype codes (the compact storage takes only these)
    check(h, ccall((:ngp_panel_columns_u8, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{UInt8}, Int64, Int64, Int32),
                   h.ptr, col0, data, size(data, 2), stride(data, 2), centre))
end_panel!(h::Handle) = check(h, ccall((:ngp_end_panel, LIB), Int32, (Ptr{Cvoid},), h.ptr))

# one byte per genotype (raw allele counts, e.g. read from a binary file instead of src/prepMatVec.jl:116): centred on the device
set_panel!(h::Handle, data::Matrix{UInt8}; centre::Bool=true) =
    check(h, ccall((:ngp_set_panel_u8, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64, Int64, Int64, Int32),
                   h.ptr, data, size(data, 1), size(data, 2), stride(data, 2), centre))

# Storage of the panel on the device, before it is set: :f32 (centred fp32 tiles) or :u8 (the codes themselves, one byte per
# genotype, centred analytically with the Float64 column means -- include/nextgp_hip.h "panel storage").  :u8 takes codes only:
# set_panel!(h, ::Matrix{UInt8}) or load_panel_file!; the centred Float64 M[set].data of the seam cannot be turned back into codes.
set_storage!(h::Handle, storage::Symbol) =
    check(h, ccall((:ngp_set_storage, LIB), Int32, (Ptr{Cvoid}, Int32), h.ptr, storage === :u8 ? 1 : 0))
# at most n streamer workgroups (taller shards, CUs left free for a second chain on the same device); 0 = automatic
set_max_shards!(h::Handle, n::Integer) = check(h, ccall((:ngp_set_max_shards, LIB), Int32, (Ptr{Cvoid}, Int32), h.ptr, n))
function shards_for_chains(h::Handle, chains::Integer)   # the largest max_shards with which `chains` chains share the device side by side
    v = Ref{Int32}(0)
    check(h, ccall((:ngp_shards_for_chains, LIB), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}), h.ptr, chains, v))
    return Int(v[])
end

# Binary panel file in place of the text genotype file (src/prepMatVec.jl:116-131): header + codes, 8 or 2 bits per genotype
function write_panel_file(path::AbstractString, G::Matrix{UInt8}; bits::Integer=8)
    rc = ccall((:ngp_write_panel_file, LIB), Int32, (Cstring, Ptr{UInt8}, Int64, Int64, Int64, Int32),
               path, G, size(G, 1), size(G, 2), stride(G, 2), bits)
    rc == 0 || error("ngp_write_panel_file ($rc): path not writable, or codes above 2 with bits = 2")
end
load_panel_file!(h::Handle, path::AbstractString; centre::Bool=true) =
    check(h, ccall((:ngp_load_panel_file, LIB), Int32, (Ptr{Cvoid}, Cstring, Int32), h.ptr, path, centre))

# regionArray::Vector{UnitRange{Int}} (1-based, src/mme.jl:335-358) -> 0-based [start, stop)
function add_marker_set!(h::Handle, col0::Integer, ncol::Integer, method::Integer, df::Float64, scale::Float64,
                         regionArray, varBeta0::Vector{Float64}; pi0::Float64=0.0, estPi::Bool=false,
                         lhs0=C_NULL, rhs0=C_NULL)
    rs = Int64[first(r) - 1 for r in regionArray]
    re = Int64[last(r) for r in regionArray]
    id = Ref{Int32}(0)
    check(h, ccall((:ngp_add_marker_set, LIB), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Int32, Float64, Float64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Float64}, Float64, Int32,
                    Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                   h.ptr, col0, ncol, method, df, scale, rs, re, length(rs), varBeta0, pi0, estPi, lhs0, rhs0, id))
    return id[]
end

# BayesR set (src/mme.jl:374-383): ONE variance, M[s].vClass multipliers, M[s].piHat class probabilities
function add_marker_set_r!(h::Handle, col0::Integer, ncol::Integer, df::Float64, scale::Float64, varBeta0::Float64,
                           vClass::Vector{Float64}, pi::Vector{Float64}; estPi::Bool=false, lhs0=C_NULL, rhs0=C_NULL)
    id = Ref{Int32}(0)
    check(h, ccall((:ngp_add_marker_set_r, LIB), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Float64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                   h.ptr, col0, ncol, df, scale, varBeta0, vClass, pi, length(vClass), estPi, lhs0, rhs0, id))
    return id[]
end

# Correlated marker sets -- sampleBayesPR!(::Tuple), src/functions.jl:140-154, set-up src/mme.jl:448-489.  M[pSet].data is a Vector of
# N x k matrices X_l (src/mme.jl:456-457); on the device the k columns of a locus sit side by side, floor(64 / k) loci per 64-column
# block from a block boundary on (include/nextgp_hip.h, ngp_add_marker_set_tuple).  tuple_panel builds that block of the panel,
# tuple_columns says where component m of locus l went (1-based panel columns), add_marker_set_tuple! declares the set:
# df = M[pSet].df (3 + k), scale = M[pSet].scale (k x k), regionArray in loci, v = varBeta[pSet][1] (k x k).
function tuple_columns(col0::Integer, nloc::Integer, k::Integer)      # col0: 0-based first panel column of the set (a multiple of 64)
    Lb = 64 ÷ k
    return [col0 + 64 * ((l - 1) ÷ Lb) + k * ((l - 1) % Lb) + m for l in 1:nloc, m in 1:k]   # 1-based panel column of (locus l, component m)
end
function tuple_panel(data::Vector{Matrix{Float64}})                   # data[l] = X_l, N x k
    nloc, (N, k) = length(data), size(data[1])
    Lb = 64 ÷ k; nblk = cld(nloc, Lb)
    out = zeros(Float64, N, 64 * nblk)                                # the set owns its blocks to the end of the last one
    cols = tuple_columns(0, nloc, k)
    for l in 1:nloc, m in 1:k
        out[:, cols[l, m]] .= data[l][:, m]
    end
    return out
end
function add_marker_set_tuple!(h::Handle, col0::Integer, nloc::Integer, k::Integer, df::Float64, scale::Matrix{Float64}, regionArray,
                               v::Matrix{Float64})
    rs = Int64[first(r) - 1 for r in regionArray]
    re = Int64[last(r) for r in regionArray]
    id = Ref{Int32}(0)
    sc = Matrix{Float64}(permutedims(scale)); vb = Matrix{Float64}(permutedims(v))   # row-major for the C side (both are symmetric)
    check(h, ccall((:ngp_add_marker_set_tuple, LIB), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Int32, Float64, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Float64}, Ref{Int32}),
                   h.ptr, col0, nloc, k, df, sc, rs, re, length(rs), vb, id))
    # a model with correlated sets takes its block chains in the inverse form (dlt = T e0: Tuple blocks 3.8 -> 3.0 us per block)
    check(h, ccall((:ngp_set_chain_form, LIB), Int32, (Ptr{Cvoid}, Int32), h.ptr, 1))
    return id[]
end

# K chains per pass over the panel: h takes owner's panel by reference (no copy); run_many! then gives all of them ONE sweep
# launch per iteration.  shards_for_pass: the max_shards the first handle needs (before its panel is set) so that K chains fit.
share_panel!(h::Handle, owner::Handle) = check(h, ccall((:ngp_share_panel, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), h.ptr, owner.ptr))
function shards_for_pass(h::Handle, chains::Integer)
    v = Ref{Int32}(0)
    check(h, ccall((:ngp_shards_for_pass, LIB), Int32, (Ptr{Cvoid}, Int32, Ref{Int32}), h.ptr, chains, v))
    return Int(v[])
end

# Kept samples to a binary file while the chain runs (instead of a text row per kept iteration, src/samplers.jl:56-104): call before
# run!, close with `nothing`; nextgp.jl_amd/api.py (samples_to_out_files) or the reader below turn the file into the *Out tables.
set_sample_file!(h::Handle, path::Union{AbstractString,Nothing}) =
    check(h, path === nothing ? ccall((:ngp_set_sample_file, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}), h.ptr, C_NULL) :
                                ccall((:ngp_set_sample_file, LIB), Int32, (Ptr{Cvoid}, Cstring), h.ptr, path))
# f(sets, sample) for every record of the file, one record in memory at a time (a record is 9 bytes per locus)
function foreach_sample(f, path::AbstractString)
    open(path, "r") do io
        String(read(io, 8)) == "NGPSMP01" || error("not a sample file: $path")
        P, nvb, nsets, nfix, ncls, rec = ntuple(_ -> read(io, Int64), 6)
        sets = [ntuple(_ -> read(io, Int64), 6) for _ in 1:nsets]     # (method, K, col0, ncol, variance entries, tuple k)
        nd = 3 + nfix + P + nvb + 2 * nsets + ncls
        raw = Vector{UInt8}(undef, rec)
        while !eof(io)
            readbytes!(io, raw, rec) == rec || break
            d = reinterpret(Float64, view(raw, 1:8 * nd))
            o = 3
            f(sets, (iter = reinterpret(Int64, view(raw, 1:8))[1], varE = d[2], b = d[3], b_fixed = d[o + 1:o + nfix],
                     beta = d[o + nfix + 1:o + nfix + P], varBeta = d[o + nfix + P + 1:o + nfix + P + nvb],
                     piHat = d[o + nfix + P + nvb + 1:o + nfix + P + nvb + 2 * nsets],
                     class_pi = d[o + nfix + P + nvb + 2 * nsets + 1:nd], delta = raw[8 * nd + 1:8 * nd + P]))
        end
    end
end
function read_sample_file(path::AbstractString)
    samples = NamedTuple[]; sets = nothing
    foreach_sample(path) do st, smp
        sets = st; push!(samples, smp)
    end
    return (sets = sets, samples = samples)
end

function class_state(h::Handle, set_id::Integer)
    pi = zeros(16); sp = zeros(16); K = Ref{Int64}(0)
    check(h, ccall((:ngp_get_class_state, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ref{Int64}), h.ptr, set_id, pi, sp, K))
    return pi[1:K[]], sp[1:K[]]
end

# resume: chain state + posterior sums + stream identity in one file (the role of the append-only *Out files, src/outFiles.jl:17-21)
save_snapshot(h::Handle, path::AbstractString) = check(h, ccall((:ngp_save_snapshot, LIB), Int32, (Ptr{Cvoid}, Cstring), h.ptr, path))
load_snapshot!(h::Handle, path::AbstractString) = check(h, ccall((:ngp_load_snapshot, LIB), Int32, (Ptr{Cvoid}, Cstring), h.ptr, path))

# pooled posterior sums of several chains (one Handle per chain / device): ONE RCCL all-reduce inside the library
# niter iterations of every chain at once, one host thread per handle inside the library (chains sharing a device run side by side
# when their grids fit it together -- set_max_shards! -- and in turns otherwise)
function run_many!(hs::Vector{Handle}, niter::Integer)
    ptrs = [h.ptr for h in hs]
    rc = ccall((:ngp_run_many, LIB), Int32, (Ptr{Ptr{Cvoid}}, Int32, Int64), ptrs, length(ptrs), niter)
    rc == 0 || error("ngp_run_many ($rc): " * join((unsafe_string(ccall((:ngp_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr)) for h in hs), " | "))
end

function allreduce_posterior!(hs::Vector{Handle})
    ptrs = Ptr{Cvoid}[x.ptr for x in hs]
    check(hs[1], ccall((:ngp_allreduce_posterior, LIB), Int32, (Ptr{Ptr{Cvoid}}, Int32), ptrs, length(ptrs)))
end

set_y!(h::Handle, y::Vector{Float64}) = check(h, ccall((:ngp_set_y, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64), h.ptr, y, length(y)))
set_residual_prior!(h::Handle, df, scale) = check(h, ccall((:ngp_set_residual_prior, LIB), Int32, (Ptr{Cvoid}, Float64, Float64), h.ptr, df, scale))
set_schedule!(h::Handle, n, burn, thin) = check(h, ccall((:ngp_set_schedule, LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64), h.ptr, n, burn, thin))
run!(h::Handle, niter) = check(h, ccall((:ngp_run, LIB), Int32, (Ptr{Cvoid}, Int64), h.ptr, niter))

"""
    sweep!(h, set_id, mSet, M, beta, delta, ycorr, varE, varBeta)

Fine seam: same argument list as the reference's `sampleBayesPR!/sampleBayesB!/sampleBayesC!(mSet, M, beta, delta, ycorr,
varE, varBeta)` (src/functions.jl:118,157,197) plus the handle and the set id.  Mutates `beta[M[mSet].pos]`, `delta[M[mSet].pos]`, `ycorr`,
`varBeta[mSet]` and, for BayesB / BayesC, `M[mSet].piHat` / `M[mSet].logPi` in place.
"""
function sweep!(h::Handle, set_id::Integer, mSet, M, beta, delta, ycorr::Vector{Float64}, varE::Float64, varBeta)
    b = vec(beta[M[mSet].pos])                 # 1 x P Matrix{Float64}: vec() shares the memory
    d = vec(delta[M[mSet].pos])                # 1 x P Matrix{Int64}
    vb = varBeta[mSet] isa Vector{Float64} ? varBeta[mSet] : Float64.(varBeta[mSet])
    isR = M[mSet].method == "BayesR"          # K class probabilities (1 x K piHat, src/mme.jl:374-383): read through ngp_get_class_state
    pih = (haskey(M[mSet], :piHat) && !isR) ? vec(M[mSet].piHat) : Float64[0.0, 0.0]
    check(h, ccall((:ngp_sweep_set, LIB), Int32,
                   (Ptr{Cvoid}, Int32, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
                   h.ptr, set_id, varE, ycorr, b, d, vb, pih))
    varBeta[mSet] isa Vector{Float64} || (varBeta[mSet] .= vb)
    if haskey(M[mSet], :piHat)
        if isR
            M[mSet].piHat .= reshape(class_state(h, set_id)[1], size(M[mSet].piHat))   # src/functions.jl:284-288
        else
            M[mSet].piHat .= reshape(pih, size(M[mSet].piHat))
        end
        M[mSet].logPi .= log.(M[mSet].piHat)  # src/functions.jl:193, :289
    end
    return nothing
end

"""
    sweep_dev!(h, set_id, varE, d_ycorr, d_beta, d_delta, d_varBeta, d_piHat = C_NULL)

The fine seam for a host that keeps its state on the GPU (`ROCArray`s of AMDGPU.jl: pass `pointer(a)` converted to `Ptr{Cvoid}`):
`ycorr` (N), the set's `beta` (ncol) and `varBeta` (regions) as Float64, `delta` (ncol) as Int64 or `C_NULL`, `piHat` (2, BayesB /
BayesC) -- all in device memory of the handle's device, updated in place by device-to-device copies (`ngp_sweep_set_dev`).
"""
function sweep_dev!(h::Handle, set_id::Integer, varE::Float64, d_ycorr::Ptr{Cvoid}, d_beta::Ptr{Cvoid}, d_delta::Ptr{Cvoid},
                    d_varBeta::Ptr{Cvoid}, d_piHat::Ptr{Cvoid} = C_NULL)
    check(h, ccall((:ngp_sweep_set_dev, LIB), Int32,
                   (Ptr{Cvoid}, Int32, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                   h.ptr, set_id, varE, d_ycorr, d_beta, d_delta, d_varBeta, d_piHat))
    return nothing
end

"""
    runSampler!(ycorr, nData, E, X, b, Z, u, varU, M, beta, varBeta, delta, chainLength, burnIn, outputFreq, outPut; seed=1)

Coarse seam: drop-in for `samplers.runSampler!` (src/samplers.jl:23) for models made of fixed effects (intercept, covariates,
factors, blocked groups) and Symbol marker sets with BayesPR / BayesB / BayesC / BayesR priors.  Anything else falls back to the reference sampler.
"""
function runSampler!(ycorr, nData, E, X, b, Z, u, varU, M, beta, varBeta, delta, chainLength, burnIn, outputFreq, outPut;
                     seed::Integer=1, device::Integer=0)
    isempty(Z) || error("random effects present: use the fine seam (NextGPHIP.sweep!) instead")
    E.str == "I" || error("weighted residuals: use the reference sampler")
    h = Handle(device=device, seed=seed)
    sets = collect(keys(M))                       # Dict order, as src/samplers.jl:50
    # consecutive column ranges of ONE panel on the device, handed over set by set (no hcat of the M[s].data on the host)
    begin_panel!(h, size(M[sets[1]].data, 1), sum(M[s].dims[2] for s in sets))
    col0 = 0
    for s in sets
        panel_columns!(h, col0, M[s].data)       # already centred (src/prepMatVec.jl:129)
        col0 += M[s].dims[2]
    end
    end_panel!(h)
    col0 = 0
    ids = Dict{Any,Int32}()
    for s in sets
        P = M[s].dims[2]
        method = M[s].method == "BayesB" ? 1 : (M[s].method == "BayesC" ? 2 : (M[s].method == "BayesR" ? 3 : 0))
        if method == 3   # ONE variance, class multipliers and class probabilities (src/mme.jl:374-383)
            ids[s] = add_marker_set_r!(h, col0, P, Float64(M[s].df), Float64(M[s].scale), Float64(varBeta[s][1]), Float64.(vec(M[s].vClass)),
                                       Float64.(vec(M[s].piHat)); estPi = M[s].estPi, lhs0 = Float64.(M[s].lhs), rhs0 = Float64.(M[s].rhs))
        else
            # BayesC loops over one-locus ranges but has ONE variance (nVarCov = 1, src/mme.jl:370): a single region for the library
            regions = method == 2 ? [1:P] : M[s].regionArray
            ids[s] = add_marker_set!(h, col0, P, method, Float64(M[s].df), Float64(M[s].scale), regions,
                                     Float64.(varBeta[s]); pi0 = method >= 1 ? M[s].piHat[2] : 0.0,
                                     estPi = method >= 1 ? M[s].estPi : false, lhs0 = Float64.(M[s].lhs), rhs0 = Float64.(M[s].rhs))
        end
        col0 += P
    end
    # fixed effects: EVERY set of X (the intercept's column of ones included) becomes a fixed-effect set of the library, in the
    # order of keys(X) -- exactly the order src/samplers.jl:39-41 samples them in; the library's own intercept is switched off
    xsets = collect(keys(X))
    for x in xsets
        Xd = Matrix{Float64}(reshape(X[x].data, :, X[x].nCol))
        check(h, ccall((:ngp_add_fixed_set, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
                       h.ptr, Xd, size(Xd, 1), size(Xd, 2), size(Xd, 1), Float64.(X[x].lhs), Float64.(X[x].rhs), Ref{Int32}(0)))
    end
    nfix = isempty(xsets) ? 0 : sum(X[x].nCol for x in xsets)
    set_y!(h, Vector{Float64}(ycorr))            # ycorr == y at this point (src/mme.jl:57)
    set_residual_prior!(h, E.df, E.scale)
    check(h, ccall((:ngp_set_intercept, LIB), Int32, (Ptr{Cvoid}, Int32), h.ptr, 0))
    set_schedule!(h, chainLength, burnIn, outputFreq)
    # ONE call for the whole chain: the kept samples (these2Keep, src/samplers.jl:26) go to a binary file through a copy stream and
    # a writer thread of the library while the chain runs -- the device never stops for a sample
    smpfile = joinpath(outPut, "samples.ngp")
    set_sample_file!(h, smpfile)
    run!(h, chainLength)
    set_sample_file!(h, nothing)                 # flushes and closes
    # ... and become the rows of the reference's *Out files afterwards, one record in memory at a time
    foreach_sample(smpfile) do sinfo, smp
        open(io -> writedlm(io, smp.b_fixed'), outPut * "/bOut", "a")       # src/samplers.jl:57
        open(io -> writedlm(io, smp.varE), outPut * "/varEOut", "a")        # src/samplers.jl:58
        c0 = 0; v0 = 0; k0 = 0
        for (k, s) in enumerate(sets)
            P = M[s].dims[2]
            open(io -> writedlm(io, smp.beta[c0+1:c0+P]'), outPut * "/beta$(s)Out", "a")          # :80
            open(io -> writedlm(io, Int.(smp.delta[c0+1:c0+P])'), outPut * "/delta$(s)Out", "a")  # :81
            M[s].method in ("BayesB", "BayesC") && open(io -> writedlm(io, smp.piHat[2k-1:2k]'), outPut * "/pi$(s)Out", "a")   # :80-82
            if M[s].method == "BayesR"                                                              # one column per class
                K = Int(sinfo[k][2])
                open(io -> writedlm(io, smp.class_pi[k0+1:k0+K]'), outPut * "/pi$(s)Out", "a")
                k0 += K
            end
            nr = length(varBeta[s])
            open(io -> writedlm(io, smp.varBeta[v0+1:v0+nr]'), outPut * "/var$(s)Out", "a")       # :101-103
            c0 += P; v0 += nr
        end
    end
    # the caller's arrays as the reference's sampler leaves them: the state after the last iteration
    Ptot = col0
    bet = Vector{Float64}(undef, Ptot); del = Vector{Int64}(undef, Ptot)
    nvb = sum(length(varBeta[s]) for s in sets); vb = Vector{Float64}(undef, nvb); pih = Vector{Float64}(undef, 2 * length(sets))
    ve = Ref{Float64}(0.0); bb = Ref{Float64}(0.0); it = Ref{Int64}(0)
    bfix = Vector{Float64}(undef, max(nfix, 1)); sbfix = similar(bfix); nfx = Ref{Int64}(0)
    check(h, ccall((:ngp_get_state, LIB), Int32,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Float64}, Ref{Int64}),
                   h.ptr, ycorr, bet, del, vb, pih, ve, bb, it))
    check(h, ccall((:ngp_get_fixed, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Int64}), h.ptr, bfix, sbfix, nfx))
    b[1:nfix] .= bfix[1:nfix]                    # positions follow keys(X), like X[xSet].pos (src/mme.jl:112-117)
    c0 = 0; v0 = 0
    for s in sets
        P = M[s].dims[2]; nr = length(varBeta[s])
        vec(beta[M[s].pos]) .= bet[c0+1:c0+P]; vec(delta[M[s].pos]) .= del[c0+1:c0+P]
        varBeta[s] isa Vector{Float64} && (varBeta[s] .= vb[v0+1:v0+nr])
        c0 += P; v0 += nr
    end
    return h
end

end # module
