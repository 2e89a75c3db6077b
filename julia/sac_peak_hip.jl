# sac_peak_hip.jl -- the `peak` program of gabes135/sac (sac/peak/sac_peak.jl, "peak:NNN") with its fixed-Theta stages
# running on an MI355X through libsac_hip.so (SAC_KIND_PEAK).
#
#   julia sac_peak_hip.jl [A_0 N_p]      reads in_peak.in, as `julia sac_peak.jl [A_0 N_p]` does (peak:1572-1575)
#
# Same inputs, same schedule (intial_sampling / run_anneal / final_anneal), same output layout (anneal.csv,
# accept_rate.csv, sample.csv, a_vals.csv, sw000.csv, log.txt).  Host side: read_G!'s window and +/- ratio guess, the
# four delta groups and amplitudes of initialize!, the half-offset tables Kp / Kn (or SAC_DEVICE_TABLES=1).
# Stage calls adjust_windows / sample(...; bin) are `ccall`s.  Environment as in sac_free_hip.jl.
# Written against the ABI; not executed by this repository's test-suite (no Julia in the build image).
include(joinpath(@__DIR__, "SacHip.jl"))

module SACs

using Printf
using ..SacHip
import ..SacHip: TimeData, read_tin, kernel_value, fmt

mutable struct SAC
    N_ω::Int
    N_p::Int
    A_0::Float64
    ω_m::Float64
    δω::Float64
    δω_h::Float64
    θ_0::Float64
    f_anneal::Float64
    a_criterion::Float64
    N_anneal::Int
    anneal_steps::Int
    sample_steps::Int
    G_file::String
    output_folder::String
    ω_0::Float64
    fix_edge::Int
    symm::Int
    kernel_type::String
    tol::Float64
    indiv_update::Bool
    td::Union{TimeData,Nothing}
    ωi_0::Int
    ωi_m::Int
    ω_window::Float64
    sample_neg::Bool
    sampled_χ2::Float64
    χ2_min::Float64
    χ2_min_all::Float64
    accept_rates::Vector{Float64}
    update_windows::Vector{Float64}
    edge_res::Matrix{Float64}          # [3, 2] as the reference holds it (peak:74)
    bounds::Vector{Float64}            # ωi_pp, ωi_pc, ωi_np, ωi_nc
    weights::Vector{Float64}           # Σ A over peak+, cont+, peak-, cont-
    χ2_anneal::Vector{Float64}
    spec_divisor::Float64              # > 0 once the binned final sampling has run (peak:1380-1381)
    handle::SacHip.Handle
    n_chains::Int
    seed::UInt64
    device::Int
    χ2_min_mode::String
end

anneal_file(self::SAC) = joinpath(self.output_folder, "anneal.csv")
accept_rate_file(self::SAC) = joinpath(self.output_folder, "accept_rate.csv")
sample_file(self::SAC) = joinpath(self.output_folder, "sample.csv")
write_log(self::SAC, msg::String, mode::String="a") = SacHip.logline(self.output_folder, msg; mode=mode)

# peak:97-131: t.in, the window from tau <= beta ÷ 4 and the guess of the +/- weight ratio (peak:117-127)
function read_G!(self::SAC)
    td = read_tin(self.G_file)
    self.td = td
    last = 0
    for i in 1:td.n_tau
        if td.tau[i] <= fld(td.beta, 4)
            last = i
        end
    end
    self.ω_window = log(1 / td.G_raw[last]) / td.tau[last]
    if self.symm == 1 || self.kernel_type == "bosonic" || maximum(td.tau) <= fld(td.beta, 2)
        return Inf
    end
    return td.G_raw[1] / td.G_raw[td.n_tau]
end

# peak:149-285: groups peak+, cont+, peak-, cont- (peak:166-169), their amplitudes, the tables; then the handle
function initialize!(self::SAC)
    if self.handle != C_NULL
        SacHip.destroy(self.handle)
        self.handle = C_NULL
    end
    ratio = read_G!(self)
    td = self.td
    N, P = self.N_ω, self.N_p
    self.ωi_0 = floor(Int, self.ω_0 / self.δω)
    self.ωi_m = ceil(Int, self.ω_m / self.δω)
    window = self.ω_window / self.δω
    self.sample_neg = !(self.symm == 1 || self.kernel_type == "bosonic")
    n_grid = self.ωi_m + 1
    ωi = zeros(Int64, 2 * (N + P))
    A = zeros(2 * (N + P))
    inv_r = isinf(ratio) ? 0.0 : 1 / ratio
    A0_p = self.A_0 / (1 + inv_r)
    A0_n = isinf(ratio) ? 0.0 : self.A_0 / (1 + ratio)
    Ac_p = (1 - self.A_0) / (1 + inv_r)
    Ac_n = isinf(ratio) ? 0.0 : (1 - self.A_0) / (1 + ratio)
    start = self.fix_edge == 1 ? self.ωi_0 : floor(Int, window)
    tri = N * (N + 1) / 2
    for i in 1:P
        ωi[i] = start
        ωi[P + N + i] = start
        A[i] = A0_p / P
        A[P + N + i] = A0_n / P
    end
    for i in 1:N
        spread = floor(Int, (window / N) * i)
        ωi[P + i] = start + spread
        ωi[2 * P + N + i] = start + spread
        A[P + i] = (i / tri) * Ac_p
        A[2 * P + N + i] = (i / tri) * Ac_n
    end
    if !self.sample_neg                                     # peak:214-227
        for i in 1:P
            A[i] = self.A_0 / P
            A[P + N + i] = 0.0
            ωi[P + N + i] = 0
        end
        for i in 1:N
            A[P + i] = (1 - self.A_0) / N
            A[2 * P + N + i] = 0.0
            ωi[2 * P + N + i] = 0
        end
        if self.symm == 1
            A ./= 2 * sum(A)
        end
    else
        A ./= sum(A)                                        # peak:229
    end

    cfg = SacHip.SacConfig(SacHip.KIND_PEAK, 0, self.symm, td.n_tau, N, self.n_chains, n_grid, self.ωi_0, self.ωi_m,
                           self.seed, 0, self.device, SacHip.HIST_REFERENCE, 0, 0, P, self.fix_edge,
                           self.kernel_type == "bosonic" ? 1 : 0)
    self.handle = SacHip.create(cfg)
    kt = SacHip.KERNEL_CODE[self.kernel_type]
    if get(ENV, "SAC_DEVICE_TABLES", "0") == "1"
        SacHip.build_table(self.handle, 0, SacHip.SacTableSpec(kt, self.symm, 0, 0, 0, 0.5, self.δω, td.beta), td.tau, td.cov)
        if self.sample_neg
            SacHip.build_table(self.handle, 1, SacHip.SacTableSpec(kt, 0, 1, 0, 0, 0.5, self.δω, td.beta), td.tau, td.cov)
        end
        SacHip.upload_data(self.handle, td.G, td.sigma_inv)
    else
        Kp = zeros(td.n_tau, n_grid)                        # peak:248-276: half-offset grid ω = (i + 1/2) δω
        Kn = self.sample_neg ? zeros(td.n_tau, n_grid) : nothing
        colp = zeros(td.n_tau)
        coln = zeros(td.n_tau)
        for c in 1:n_grid
            ω = (c - 1 + 0.5) * self.δω
            for j in 1:td.n_tau
                colp[j] = kernel_value(self.kernel_type, ω, td.tau[j], td.beta)
                coln[j] = kernel_value(self.kernel_type, -ω, td.tau[j], td.beta)
            end
            Kp[:, c] = td.cov' * (self.symm == 1 ? colp .+ coln : colp)
            if self.sample_neg
                Kn[:, c] = td.cov' * coln
            end
        end
        SacHip.upload_problem_peak(self.handle, Kp, Kn, td.G, td.sigma_inv)
    end
    SacHip.set_state(self.handle; wi=ωi, A=A, windows=fill(window / 10, 11))       # peak:1507
    SacHip.init_gbar(self.handle)
    SacHip.set_indiv_update(self.handle, self.indiv_update)
    merge_stats!(self)
end

function merge_stats!(self::SAC)
    st = SacHip.get_peak_stats(self.handle)
    self.sampled_χ2 = st.sampled_chi2
    self.χ2_min_all = st.chi2_min
    self.χ2_min = self.χ2_min_mode == "min" ? st.chi2_min : st.chi2_min_mean
    self.accept_rates .= st.accept_rates
    self.update_windows .= st.windows
    for q in 1:3
        self.edge_res[q, 1] = st.edge_res[2 * q - 1]
        self.edge_res[q, 2] = st.edge_res[2 * q]
    end
    self.bounds .= st.bounds
    self.weights .= st.weights
    return st
end

# n_chains > 1 only: a chain that fails the reference's own bad-start test (χ2_min > 1000 N_τ, peak:1522) takes over a
# healthy chain's configuration and continues on its own random stream
function repair_chains(self::SAC)
    if self.n_chains == 1
        return 0
    end
    _, c2min = SacHip.get_chain_chi2(self.handle, self.n_chains)
    limit = 1000 * self.td.n_tau
    good = sort(findall(x -> x <= limit, c2min), by=c -> c2min[c])
    bad = findall(x -> x > limit, c2min)
    if isempty(good) || isempty(bad)
        return 0
    end
    for (j, c) in enumerate(bad)
        SacHip.copy_chain(self.handle, good[mod1(j, length(good))] - 1, c - 1)
    end
    return length(bad)
end

adjust_windows(self::SAC, steps::Int, θ::Float64) = SacHip.adjust_windows(self.handle, steps, θ)     # peak:1063

function sample(self::SAC, steps::Int, θ::Float64, bin::Bool=false)                                   # peak:1100
    if bin
        SacHip.sample_bin(self.handle, steps, θ)
    else
        SacHip.sample(self.handle, steps, θ)
    end
    merge_stats!(self)
end

# edge_p, edge_n, A0_p, A0_n, Ac_p, Ac_n as peak:1301-1312 prints them
function edges_and_weights(self::SAC)
    edge_p = self.bounds[1] * self.δω
    if !self.sample_neg
        return edge_p, -edge_p, self.weights[1], self.weights[1], self.weights[2], self.weights[2]
    end
    return edge_p, -self.bounds[3] * self.δω, self.weights[1], self.weights[3], self.weights[2], self.weights[4]
end

function init_outfiles(self::SAC)                           # peak:310-343
    mkpath(self.output_folder)
    open(anneal_file(self), "w") do f
        println(f, "i,theta,chi2_min,chi2_avg,edge_p,edge_n,Ap_p,Ap_n,Ac_p,Ac_n")
    end
    open(sample_file(self), "w") do f
        println(f, "i,a,theta,chi2_min,chi2_avg,edge_p,edge_n,sigma_edge_p,sigma_edge_n,Ap_p,Ap_n,Ac_p,Ac_n")
    end
    open(accept_rate_file(self), "w") do f
        println(f, "i,ar_pcont1,ar_pcont2,ar_pcont3,ar_ncont1,ar_ncont2,ar_ncont3,∆w_pcont1,∆w_pcont2,∆w_ncont1,∆w_ncont2,",
                "ar_ppeak,ar_npeak,∆w_ppeak,∆w_npeak,ar_wAc,ar_wAp,ar_wApc,∆w_Ac,∆w_Ap,∆w_Apc")
    end
    open(joinpath(self.output_folder, "log.txt"), "w") do f
    end
end

# peak:1322-1333 / 1410-1421 (the final row rounds the windows to 4 digits and leaves window 11 unscaled, as the reference does)
function accept_row(self::SAC, i::Int, final::Bool)
    ar, w, δ = self.accept_rates, self.update_windows, self.δω
    wd = final ? 4 : 8
    vals = Float64[]
    for k in (1, 2, 3, 8, 9, 10)
        push!(vals, round(ar[k], digits=4))
    end
    for k in (1, 2, 8, 9)
        push!(vals, round(w[k] * δ, digits=wd))
    end
    push!(vals, round(ar[4], digits=4), round(ar[5], digits=4), round(w[4] * δ, digits=wd), round(w[5] * δ, digits=wd))
    push!(vals, round(ar[6], digits=4), round(ar[7], digits=4), round(ar[11], digits=4))
    push!(vals, round(w[6] * δ, digits=8), round(w[7] * δ, digits=8), round(final ? w[11] : w[11] * δ, digits=8))
    open(accept_rate_file(self), "a") do f
        println(f, fmt(Float64(i)), ",", join(map(fmt, vals), ","))
    end
end

function intial_sampling(self::SAC, θ_0::Float64)           # peak:1272-1276
    adjust_windows(self, self.anneal_steps, 10 * θ_0)
    adjust_windows(self, self.anneal_steps, 5 * θ_0)
    adjust_windows(self, self.anneal_steps, 2 * θ_0)
    merge_stats!(self)
end

function run_anneal(self::SAC, θ_0::Float64, wr::Bool=false; on_stage=nothing)     # peak:1289-1350
    θ = θ_0
    N_τ = self.td.n_tau
    for i in 1:self.N_anneal
        adjust_windows(self, self.anneal_steps, θ)
        sample(self, self.anneal_steps, θ)
        if on_stage !== nothing
            on_stage(i, θ)
        end
        vals = (θ, self.χ2_min / N_τ, self.sampled_χ2 / N_τ, edges_and_weights(self)...)
        open(anneal_file(self), "a") do f                   # peak:1315-1319
            println(f, fmt(Float64(i)), ",", join(map(v -> fmt(round(v, digits=8)), vals), ","))
        end
        accept_row(self, i, false)
        if wr
            write_spec(self, i)
        end
        self.χ2_anneal[i] = self.sampled_χ2                 # peak:1341
        if (self.sampled_χ2 - self.χ2_min) < (self.tol * N_τ)          # peak:1343
            return
        end
        θ /= self.f_anneal
    end
end

function final_anneal(self::SAC, θ_opt::Float64)            # peak:1354-1427
    N_τ = self.td.n_tau
    write_log(self, "Beginning Fast Anneal to θ_opt.")
    for i in 1:10
        θ = θ_opt * (11 - i)
        adjust_windows(self, self.anneal_steps * i, θ)
        sample(self, self.anneal_steps * i, θ)
    end
    write_log(self, "Fast Anneal to θ_opt Finished.")
    write_log(self, "Beginning Final Sampling at θ_opt.")
    N_bins = 10
    SacHip.zero_accumulators(self.handle)                   # peak:1370-1372
    for b in 1:N_bins
        sample(self, self.sample_steps, θ_opt, true)
    end
    self.spec_divisor = N_bins * self.sample_steps          # peak:1380-1381
    e = self.edge_res ./ N_bins                             # peak:1382-1384
    for pn in 1:2
        e[3, pn] = sqrt(abs(e[3, pn] - e[2, pn]^2)) / sqrt(N_bins - 1)
    end
    e .*= self.δω
    a = (self.sampled_χ2 - self.χ2_min) / sqrt(2 * self.χ2_min)
    _, _, A0_p, A0_n, Ac_p, Ac_n = edges_and_weights(self)
    vals = (0.0, round(a, digits=4), round(θ_opt, digits=8), round(self.χ2_min / N_τ, digits=4),
            round(self.sampled_χ2 / N_τ, digits=4), round(e[2, 1], digits=4), round(-e[2, 2], digits=4),
            round(e[3, 1], digits=4), round(e[3, 2], digits=4), round(A0_p, digits=4), round(A0_n, digits=4),
            round(Ac_p, digits=4), round(Ac_n, digits=4))
    open(sample_file(self), "a") do f                       # peak:1400-1408
        println(f, join(map(fmt, vals), ","))
    end
    accept_row(self, 0, true)
    write_spec(self, 0)
    write_log(self, "Final Sampling at θ_opt Finished.")
end

# peak:1170-1265
function write_spec(self::SAC, n::Int)
    td = self.td
    n_grid = self.ωi_m + 1
    div_by = self.spec_divisor > 0 ? self.spec_divisor : 1.0
    pspec = zeros(n_grid, 2)
    cspec = zeros(n_grid, 2)
    pspec[:, 1] = SacHip.get_histogram_peak(self.handle, 0, n_grid) ./ div_by
    cspec[:, 1] = SacHip.get_histogram_peak(self.handle, 1, n_grid) ./ div_by
    pspec[:, 2] = SacHip.get_histogram_peak(self.handle, 2, n_grid) ./ div_by
    cspec[:, 2] = SacHip.get_histogram_peak(self.handle, 3, n_grid) ./ div_by
    for c in 1:n_grid                                       # peak:1176-1183
        f = self.kernel_type == "bosonic" ? 1.0 + exp(-td.beta * (self.δω * (c - 1))) : 1.0
        s = (td.norm * pi) / f
        pspec[c, 1] *= s
        pspec[c, 2] *= s
        cspec[c, 1] *= s
        cspec[c, 2] *= s
    end
    conv = round(Int, self.δω_h / self.δω)                  # peak:1185-1186
    N_h = Int(div(self.ωi_m, conv))
    P = zeros(N_h, 2)
    C = zeros(N_h, 2)
    for k in 1:N_h                                          # peak:1189-1197
        for pn in 1:2
            sp = 0.0
            sc = 0.0
            for c in ((k - 1) * conv + 1):(k * conv)
                sp += pspec[c, pn]
                sc += cspec[c, pn]
            end
            P[k, pn] = sp / self.δω_h
            C[k, pn] = sc / self.δω_h
        end
    end
    function last_bin(pn)                                   # peak:1202-1216
        j = N_h
        for k in N_h:-1:1                                   # 1-based k: the count of coarse bins BELOW the last filled one
            j = k - 1
            if C[k, pn] > 1e-10
                break
            end
        end
        return j
    end
    jf = last_bin(1)
    kf = last_bin(2)
    open(joinpath(self.output_folder, @sprintf("sw%03d.csv", n)), "w") do fh
        println(fh, "omega,S,S_p,S_c")
        if self.kernel_type == "bosonic"                    # peak:1226-1233
            for k in jf:-1:1
                ω = self.δω_h * (k - 1)
                x = exp(-td.beta * ω)
                println(fh, fmt(-ω), ",", fmt((P[k, 1] + C[k, 1]) * x), ",", fmt(P[k, 1] * x), ",", fmt(C[k, 1] * x))
            end
        elseif self.symm == 1                               # peak:1235-1242
            for k in jf:-1:1
                ω = self.δω_h * (k - 1)
                println(fh, fmt(-ω), ",", fmt(P[k, 1] + C[k, 1]), ",", fmt(P[k, 1]), ",", fmt(C[k, 1]))
            end
        else                                                # peak:1245-1251
            for k in kf:-1:1
                ω = self.δω_h * (k - 1)
                println(fh, fmt(-ω), ",", fmt(P[k, 2] + C[k, 2]), ",", fmt(P[k, 2]), ",", fmt(C[k, 2]))
            end
        end
        for k in 1:jf                                       # peak:1255-1261
            ω = self.δω_h * (k - 1)
            println(fh, fmt(ω), ",", fmt(P[k, 1] + C[k, 1]), ",", fmt(P[k, 1]), ",", fmt(C[k, 1]))
        end
    end
end

function optimal_θ(self::SAC)                               # peak:1545-1548
    a_vals = (self.χ2_anneal .- self.χ2_min) ./ sqrt(2 * self.χ2_min)
    θ_vals = self.θ_0 ./ (self.f_anneal .^ (0:(self.N_anneal - 1)))
    return a_vals, θ_vals, θ_vals[argmin(abs.(a_vals .- self.a_criterion))]
end

# peak:1441-1473 + the SAC(...) constructor call: in_peak.in -> struct (no files touched)
function load_input(A_0_in=nothing, N_p_in=nothing; in_file::String="in_peak.in")
    rows = map(split, readlines(in_file))
    N_ω, N_p = Int(parse(Float64, rows[1][1])), Int(parse(Float64, rows[1][2]))
    A_0, ω_m, δω, δω_h = parse.(Float64, rows[2][1:4])
    θ_0, f_anneal, a_criterion = parse.(Float64, rows[3][1:3])
    N_anneal, anneal_steps, sample_steps = Int.(parse.(Float64, rows[4][1:3]))
    G_file, output_folder = String(rows[5][1]), String(rows[5][2])
    fix_edge, symm, kernel_type = parse(Float64, rows[6][1]), Int(parse(Float64, rows[6][2])), String(rows[6][3])
    if A_0_in !== nothing
        A_0 = A_0_in
    end
    if N_p_in !== nothing
        N_p = N_p_in
    end
    if kernel_type == "bosonic"                             # peak:1460-1461
        symm = 0
    elseif symm == 1
        output_folder *= "_symm"
    end
    output_folder = joinpath(output_folder, @sprintf("Np_%02i", N_p), @sprintf("A0_%.3f", A_0))
    ω_0 = fix_edge != 0 ? fix_edge : 0.0                    # peak:1466-1471
    return SAC(N_ω, N_p, A_0, ω_m, δω, δω_h, θ_0, f_anneal, a_criterion, N_anneal, anneal_steps, sample_steps, G_file,
               output_folder, ω_0, fix_edge != 0 ? 1 : 0, symm, kernel_type, 1e-3, false, nothing, 0, 0, 0.0, true,
               0.0, 0.0, 0.0, zeros(11), zeros(11), zeros(3, 2), zeros(4), zeros(4), zeros(N_anneal), 0.0, C_NULL,
               parse(Int, get(ENV, "SAC_N_CHAINS", "1")), parse(UInt64, get(ENV, "SAC_SEED", "0")),
               parse(Int, get(ENV, "SAC_DEVICE", "0")), get(ENV, "SAC_CHI2_MIN_MODE", "chain_mean"))
end

# peak:1439-1569
function run(A_0_in=nothing, N_p_in=nothing; in_file::String="in_peak.in")
    sac = load_input(A_0_in, N_p_in; in_file=in_file)
    output_folder, θ_0 = sac.output_folder, sac.θ_0
    mkpath(output_folder)
    cp(in_file, joinpath(output_folder, "in_peak.in"); force=true)
    cp(sac.G_file, joinpath(output_folder, "t.in"); force=true)
    init_outfiles(sac)
    write_log(sac, "Beginning Initialization.", "w")
    initialize!(sac)
    write_log(sac, "Initialization Finished.")
    write_log(sac, "Beginning Initial Sampling.")
    intial_sampling(sac, θ_0)
    if sac.χ2_min_all > 1000 * sac.td.n_tau                 # peak:1522-1533
        sac.indiv_update = true
        initialize!(sac)
        intial_sampling(sac, θ_0)
    end
    n_fixed = repair_chains(sac)
    write_log(sac, "Initial Sampling Finished ($n_fixed of $(sac.n_chains) chains restarted from a healthy chain).")
    write_log(sac, "Beginning Anneal.")
    run_anneal(sac, θ_0, false)
    write_log(sac, "Anneal Finished.")
    a_vals, θ_vals, θ_opt = optimal_θ(sac)
    open(joinpath(output_folder, "a_vals.csv"), "w") do f   # peak:1551-1554
        println(f, "a,theta")
        for i in 1:length(a_vals)
            println(f, fmt(a_vals[i]), ",", fmt(θ_vals[i]))
        end
    end
    SacHip.set_state(sac.handle; windows=fill(sac.ω_window / sac.δω / 10, 11))     # peak:1557
    write_log(sac, "Beginning Final Anneal.")
    final_anneal(sac, θ_opt)
    write_log(sac, "Final Anneal Finished.")
    SacHip.destroy(sac.handle)
    return θ_opt
end

end # module

if abspath(PROGRAM_FILE) == @__FILE__
    if length(ARGS) >= 2
        SACs.run(parse(Float64, ARGS[1]), parse(Int, ARGS[2]))
    else
        SACs.run()
    end
end
