# cross_val_hip.jl -- sac/cross_validation/cross_val.jl ("cv:NNN") on the GPU engine.
#
#   julia cross_val_hip.jl <free|peak|edge> <rep>        (the reference's argument order, cv:514-516)
#
# As in the reference (cv:518-529) the chosen sampler's file is included, and for halves "a" and "b" of replica `rep`
# its run is repeated up to the main anneal with tol = 1e-6 (cv:50, 200, 415) while [round(θ, 8), cov * Gbar...] is
# appended to GSAC.csv after every Θ stage (cv:105-108, 260-263, 494-497).  Gbar is the mean over the handle's chains,
# fetched in ONE call (sac_get_gbar_mean / sac_edge_get_gbar_mean).  With SAC_CV_FUSED=1 the validation χ² against the
# OTHER half's t.in is taken on the device as well (sac_validation_chi2: calc_cv.jl:29-58 without the file round trip)
# and written to validation.csv.
# Written against the ABI; not executed by this repository's test-suite (no Julia in the build image).
const PARAM = length(ARGS) >= 1 ? ARGS[1] : "free"
const REP = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 1

if PARAM == "free"
    include(joinpath(@__DIR__, "sac_free_hip.jl"))
elseif PARAM == "peak"
    include(joinpath(@__DIR__, "sac_peak_hip.jl"))
elseif PARAM == "edge"
    include(joinpath(@__DIR__, "sac_edge_hip.jl"))
else
    error("usage: julia cross_val_hip.jl <free|peak|edge> <rep>")
end

using Printf

other_half(ab::String) = ab == "a" ? "b" : "a"
tin_name(folder::String, rep::Int, ab::String) = joinpath(folder, @sprintf("t_%03i%s.in", rep, ab))

function append_gsac(file::String, θ::Float64, cov::Matrix{Float64}, gbar::Vector{Float64})
    open(file, "a") do f                                    # cv:105-108
        println(f, join(map(SacHip.fmt, vcat(round(θ, digits=8), cov * gbar)), ","))
    end
end

# the fused scorer: χ²_v of the chain-mean Gbar against the other half's data, straight from the device
function validator(sac, other_tin::String, edge::Bool)
    if get(ENV, "SAC_CV_FUSED", "0") != "1" || !isfile(other_tin)
        return nothing
    end
    o = SacHip.read_tin(other_tin)
    vfile = joinpath(sac.output_folder, "validation.csv")
    open(vfile, "w") do f
        println(f, "theta,chi2_validation")
    end
    return function (θ::Float64)
        v = edge ? SacHip.edge_validation_chi2(sac.handle, sac.td.cov, o.cov, o.G, o.sigma_inv) :
                   SacHip.validation_chi2(sac.handle, sac.td.cov, o.cov, o.G, o.sigma_inv)[1]
        open(vfile, "a") do f
            println(f, SacHip.fmt(round(θ, digits=8)), ",", SacHip.fmt(v / sac.td.n_tau))      # calc_cv.jl:51
        end
    end
end

# cv:6-144
function free_cv(rep::Int, ab::String; in_file::String="in_free.in")
    sac = SACs.load_input(in_file)
    G_folder = sac.G_file                                   # row 5 holds the FOLDER of the t_###{a,b}.in files here (cv:14)
    sac.G_file = tin_name(G_folder, rep, ab)                # cv:17
    symm_tag = (sac.symm == 1 && sac.kernel_type != "bosonic") ? "_symm" : ""
    base = symm_tag == "" ? sac.output_folder : sac.output_folder[1:(lastindex(sac.output_folder) - 5)]
    sac.output_folder = joinpath(base, @sprintf("%03i%s", rep, ab), "free" * symm_tag)          # cv:18-26
    mkpath(sac.output_folder)
    cp(in_file, joinpath(sac.output_folder, "in_free.in"); force=true)
    cp(sac.G_file, joinpath(sac.output_folder, "t.in"); force=true)
    sac.tol = 1e-6                                          # cv:50
    gsac = joinpath(sac.output_folder, "GSAC.csv")
    open(gsac, "w") do f
    end
    SACs.init_outfiles(sac)
    SACs.write_log(sac, "Beginning Initialization.", "w")
    SACs.read_G!(sac)
    SACs.initialize!(sac)
    SACs.reset_windows!(sac)
    SACs.write_log(sac, "Initialization Finished.")
    SACs.write_log(sac, "Beginning Equillibration.")
    SACs.fast_anneal(sac)                                   # cv:85
    SACs.write_log(sac, "Equillibration Finished.")
    SACs.write_log(sac, "Beginning Main Anneal.")
    score = validator(sac, tin_name(G_folder, rep, other_half(ab)), false)
    SACs.main_anneal(sac, false; on_stage=function (i, θ)   # cv:98-140
        append_gsac(gsac, θ, sac.td.cov, SacHip.get_gbar_mean(sac.handle, sac.td.n_tau))
        if score !== nothing
            score(θ)
        end
    end)
    SACs.write_log(sac, "Main Anneal Finished.")
    SacHip.destroy(sac.handle)
    return sac.output_folder
end

# cv:147-312
function peak_cv(rep::Int, ab::String; in_file::String="in_peak.in", A_0=nothing, N_p=nothing)
    sac = SACs.load_input(A_0, N_p; in_file=in_file)
    G_folder = sac.G_file
    sac.G_file = tin_name(G_folder, rep, ab)
    tail = joinpath(@sprintf("Np_%02i", sac.N_p), @sprintf("A0_%.3f", sac.A_0))                 # cv:158-177
    symm_tag = (sac.symm == 1 && sac.kernel_type != "bosonic") ? "_symm" : ""
    root = sac.output_folder[1:(lastindex(sac.output_folder) - length(tail) - 1 - length(symm_tag))]
    sac.output_folder = joinpath(root, @sprintf("%03i%s", rep, ab), "peak" * symm_tag, tail)
    mkpath(sac.output_folder)
    sac.tol = 1e-6                                          # cv:200
    gsac = joinpath(sac.output_folder, "GSAC.csv")
    open(gsac, "w") do f
    end
    SACs.init_outfiles(sac)
    SACs.write_log(sac, "Beginning Initialization.", "w")
    SACs.initialize!(sac)
    SACs.write_log(sac, "Initialization Finished.")
    SACs.intial_sampling(sac, sac.θ_0)
    if sac.χ2_min_all > 1000 * sac.td.n_tau                 # peak:1522-1533
        sac.indiv_update = true
        SACs.initialize!(sac)
        SACs.intial_sampling(sac, sac.θ_0)
    end
    SACs.repair_chains(sac)
    score = validator(sac, tin_name(G_folder, rep, other_half(ab)), false)
    SACs.run_anneal(sac, sac.θ_0, false; on_stage=function (i, θ)      # cv:250-300
        append_gsac(gsac, θ, sac.td.cov, SacHip.get_gbar_mean(sac.handle, sac.td.n_tau))
        if score !== nothing
            score(θ)
        end
    end)
    SacHip.destroy(sac.handle)
    return sac.output_folder
end

# cv:315-509
function edge_cv(rep::Int, ab::String; in_file::String="in_edge.in", A_c=nothing, A_r=nothing, p=nothing)
    sac = SACs.load_input(A_c, A_r, p; in_file=in_file)
    G_folder = sac.G_file
    sac.G_file = tin_name(G_folder, rep, ab)
    parts = splitpath(sac.output_folder)                    # <out>_<mode>/<settings>/Ac_.../p_...[/Ar_...]
    sac.output_folder = joinpath(parts[1], @sprintf("%03i%s", rep, ab), "edge", parts[2:lastindex(parts)]...)   # cv:330-352
    mkpath(sac.output_folder)
    sac.tol = 1e-6                                          # cv:415
    gsac = joinpath(sac.output_folder, "GSAC.csv")
    open(gsac, "w") do f
    end
    SACs.write_log(sac, "Beginning Initialization.", "w")
    SACs.init_outfiles(sac)
    SACs.initialize!(sac)
    SACs.run_bins(sac, 5 * sac.anneal_steps, sac.bins, sac.θ_0)        # cv:473
    SACs.write_res(sac, 0, sac.θ_0)
    score = validator(sac, tin_name(G_folder, rep, other_half(ab)), true)
    θ = sac.θ_0
    i_trans = sac.N_anneal * 0.2
    SACs.write_log(sac, "Beginning Main Anneal.")
    for i in 1:sac.N_anneal                                 # cv:480-505: anneal() with the GSAC dump inside
        steps = i < i_trans ? ceil(Int, sac.anneal_steps * (1 - ((5 / 6) * i / i_trans))) : ceil(Int, sac.anneal_steps / 6)
        SACs.run_bins(sac, steps, sac.bins, θ)
        SACs.write_res(sac, i, θ)
        append_gsac(gsac, θ, sac.td.cov, SacHip.edge_get_gbar_mean(sac.handle, sac.td.n_tau))   # cv:494-497
        if score !== nothing
            score(θ)
        end
        χ2_avg = sac.res.chi2_res[2]
        if (χ2_avg - sac.χ2_min) < (sac.tol * sac.td.n_tau)
            break
        end
        θ /= sac.f_anneal
        sac.χ2_anneal[i] = χ2_avg
    end
    SACs.write_log(sac, "Main Anneal Finished.")
    SacHip.edge_destroy(sac.handle)
    return sac.output_folder
end

if abspath(PROGRAM_FILE) == @__FILE__
    cv = PARAM == "free" ? free_cv : (PARAM == "peak" ? peak_cv : edge_cv)
    cv(REP, "a")                                            # cv:518-529
    cv(REP, "b")
end
