# sac_free_hip.jl -- the `free` program of gabes135/sac (sac/free/sac_free.jl, "free:NNN") with its fixed-Theta stages
# running on an MI355X through libsac_hip.so.
#
#   julia sac_free_hip.jl            reads in_free.in in the working directory, exactly as `julia sac_free.jl` does
#
# Same input files (in_free.in, t.in), same annealing schedule (fast_anneal / main_anneal / final_anneal), same output
# folder layout (anneal.csv, accept_rate.csv, sample.csv, sw%03d.csv, log.txt).  What the reference builds once on the
# host stays on the host (read_G!, the frequency grid, the initial ladder; the K table too unless SAC_DEVICE_TABLES=1);
# adjust_windows(self, steps, theta) and sample(self, steps, theta) are `ccall`s (SacHip.jl).
# No counterpart in the reference: SAC_N_CHAINS independent chains (default 1), SAC_SEED, SAC_DEVICE,
# SAC_CHI2_MIN_MODE = chain_mean | min (how many chains form the chi2_min of the Theta criteria; one chain: the same).
# Written against the ABI; not executed by this repository's test-suite (no Julia in the build image).
include(joinpath(@__DIR__, "SacHip.jl"))

module SACs

using Printf
using ..SacHip
import ..SacHip: TimeData, read_tin, kernel_value, fmt

mutable struct SAC
    # parameters of in_free.in (free:5-43)
    par::Int
    N_ω::Int
    ω_0::Float64
    ω_m::Float64
    δω::Float64
    δω_h::Float64
    θ_0::Float64
    f_anneal::Float64
    f_final::Float64
    a1::Float64
    a2::Float64
    N_anneal::Int
    anneal_steps::Int
    sample_steps::Int
    G_file::String
    output_folder::String
    symm::Int
    kernel_type::String
    tol::Float64
    # what read_G! / initialize! leave behind
    td::Union{TimeData,Nothing}
    ωi_0::Int
    ωi_m::Int
    N_ωi::Int
    ω_window::Float64
    ωi_array::Vector{Int64}
    A_array::Vector{Float64}
    # stage results (filled from the library after every stage)
    sampled_χ2::Float64
    χ2_min::Float64
    χ2_min_all::Float64
    accept_rates::Vector{Float64}
    update_windows::Vector{Float64}
    χ2_anneal::Vector{Float64}
    # the GPU side
    handle::SacHip.Handle
    n_chains::Int
    seed::UInt64
    device::Int
    χ2_min_mode::String
end

function SAC(par, N_ω, ω_0, ω_m, δω, δω_h, θ_0, f_anneal, f_final, a1, a2, N_anneal, anneal_steps, sample_steps,
             G_file, output_folder, symm, kernel_type)
    return SAC(par, N_ω, ω_0, ω_m, δω, δω_h, θ_0, f_anneal, f_final, a1, a2, N_anneal, anneal_steps, sample_steps,
               G_file, output_folder, symm, kernel_type, 1e-3, nothing, 0, 0, 0, 0.0, Int64[], Float64[],
               0.0, 0.0, 0.0, zeros(6), zeros(4), zeros(N_anneal), C_NULL,
               parse(Int, get(ENV, "SAC_N_CHAINS", "1")), parse(UInt64, get(ENV, "SAC_SEED", "0")),
               parse(Int, get(ENV, "SAC_DEVICE", "0")), get(ENV, "SAC_CHI2_MIN_MODE", "chain_mean"))
end

anneal_file(self::SAC) = joinpath(self.output_folder, "anneal.csv")
accept_rate_file(self::SAC) = joinpath(self.output_folder, "accept_rate.csv")
sample_file(self::SAC) = joinpath(self.output_folder, "sample.csv")
write_log(self::SAC, msg::String, mode::String="a") = SacHip.logline(self.output_folder, msg; mode=mode)

# free:99-131 (t.in) and the window guess free:125-130: log(1 / G(tau <= beta ÷ 2)[end]) / tau
function read_G!(self::SAC)
    td = read_tin(self.G_file)
    self.td = td
    last = 0
    for i in 1:td.n_tau
        if td.tau[i] <= fld(td.beta, 2)
            last = i
        end
    end
    self.ω_window = log(1 / td.G_raw[last]) / td.tau[last]
    return td
end

# free:134-218: grid, initial ladder of deltas, amplitudes; the K table in the covariance eigenbasis
function initialize!(self::SAC)
    td = self.td
    if self.symm == 1
        self.ω_0 = 0.0
    end
    self.ωi_0 = floor(Int, self.ω_0 / self.δω)
    self.ωi_m = ceil(Int, self.ω_m / self.δω)
    first_i = self.ωi_0 < 0 ? self.ωi_0 : 0
    self.N_ωi = self.ωi_0 < 0 ? self.ωi_m - self.ωi_0 + 1 : self.ωi_m + 1
    step = div(self.ωi_m - self.ωi_0, self.N_ω)
    self.ωi_array = zeros(Int64, self.N_ω)
    self.A_array = zeros(self.N_ω)
    for i in 1:self.N_ω
        self.ωi_array[i] = first_i + step * i
        self.A_array[i] = (self.par == 1 || self.par == 2) ? 1.0 : Float64(i)
    end
    total = sum(self.A_array)
    self.A_array ./= (self.symm == 1 ? 2 * total : total)
    self.update_windows .= self.ω_window / self.δω

    cfg = SacHip.SacConfig(SacHip.KIND_FREE, self.par, self.symm, td.n_tau, self.N_ω, self.n_chains, self.N_ωi,
                           self.ωi_0, self.ωi_m, self.seed, 0, self.device, SacHip.HIST_REFERENCE, 0, 0, 0, 0, 0)
    self.handle = SacHip.create(cfg)
    if get(ENV, "SAC_DEVICE_TABLES", "0") == "1"
        spec = SacHip.SacTableSpec(SacHip.KERNEL_CODE[self.kernel_type], self.symm, 0, 0, self.ωi_0, 0.0, self.δω, td.beta)
        SacHip.build_table(self.handle, 0, spec, td.tau, td.cov)
        SacHip.upload_data(self.handle, td.G, td.sigma_inv)
    else
        K = zeros(td.n_tau, self.N_ωi)                      # free:199-213, tau contiguous per grid point
        col = zeros(td.n_tau)
        for c in 1:self.N_ωi
            ω = (self.ωi_0 + c - 1) * self.δω
            for j in 1:td.n_tau
                col[j] = kernel_value(self.kernel_type, ω, td.tau[j], td.beta)
                if self.symm == 1
                    col[j] += kernel_value(self.kernel_type, -ω, td.tau[j], td.beta)
                end
            end
            K[:, c] = td.cov' * col                         # free:212
        end
        SacHip.upload_problem(self.handle, K, td.G, td.sigma_inv)
    end
    SacHip.set_state(self.handle; wi=self.ωi_array, A=self.A_array, windows=self.update_windows)
    SacHip.init_gbar(self.handle)                           # Gbar, χ2, χ2_min = χ2 (free:1213-1216)
    merge_stats!(self)
end

# self.sampled_χ2, χ2_min, accept_rates, update_windows after a stage (what free:1064-1076 prints)
function merge_stats!(self::SAC)
    st = SacHip.get_stats(self.handle)
    self.sampled_χ2 = st.sampled_chi2
    self.χ2_min_all = st.chi2_min
    self.χ2_min = self.χ2_min_mode == "min" ? st.chi2_min : st.chi2_min_mean
    self.accept_rates .= st.accept_rates
    self.update_windows .= st.windows
    return st
end

# free:866
adjust_windows(self::SAC, steps::Int, θ::Float64) = SacHip.adjust_windows(self.handle, steps, θ)

# free:903
function sample(self::SAC, steps::Int, θ::Float64)
    SacHip.sample(self.handle, steps, θ)
    merge_stats!(self)
end

function reset_windows!(self::SAC)                          # free:1218, 1254
    SacHip.set_state(self.handle; windows=fill(self.ω_window / self.δω, 4))
end

function init_outfiles(self::SAC)                           # free:931-955
    mkpath(self.output_folder)
    open(anneal_file(self), "w") do f
        println(f, "i,theta,chi2_min,chi2_avg")
    end
    open(accept_rate_file(self), "w") do f
        println(f, "i,ar_1,ar_2,ar_3,ar_4,ar_5,ar_6,omega_range_1,omega_range_2,omega_range_4")
    end
    open(sample_file(self), "w") do f
        println(f, "i,a,theta,chi2_min,chi2_avg")
    end
    open(joinpath(self.output_folder, "log.txt"), "w") do f
    end
end

# free:959-1019: normalisation back in, rebin to δω_h (inclusive window: neighbouring coarse bins share one fine bin),
# trim the empty tail, mirror for the bosonic kernel
function write_spec(self::SAC, n::Int)
    td = self.td
    spec = SacHip.get_histogram(self.handle, self.N_ωi)
    conv = round(Int, self.δω_h / self.δω)
    N_h = Int(div(self.ωi_m - self.ωi_0, conv))
    for c in 1:(self.ωi_m - self.ωi_0 + 1)
        ω = self.δω * (self.ωi_0 + c - 1)
        f = self.kernel_type == "bosonic" ? 1.0 + exp(-td.beta * ω) : 1.0
        spec[c] = spec[c] * (td.norm * pi) / f
    end
    for k in 1:N_h
        s = 0.0
        for c in ((k - 1) * conv + 1):(k * conv + 1)
            s += spec[c]
        end
        spec[k] = s / self.δω_h
    end
    j0 = 1
    jf = N_h
    for k in (N_h - 1):-1:0                                 # free:987-992 (the walk starts one bin below the top)
        jf = k
        if k >= 1 && spec[k] > 1e-10
            break
        end
    end
    base = self.ωi_0 * self.δω
    open(joinpath(self.output_folder, @sprintf("sw%03d.csv", n)), "w") do fh
        println(fh, "omega,S")
        if self.kernel_type == "bosonic"
            for k in jf:-1:j0
                ω = self.δω_h * (k - 1) + base
                println(fh, fmt(-ω), ",", fmt(spec[k] * exp(-td.beta * ω)))
            end
        end
        for k in j0:jf
            ω = self.δω_h * (k - 1) + base
            println(fh, fmt(ω), ",", fmt(spec[k]))
        end
    end
end

function fast_anneal(self::SAC)                             # free:1036-1048
    for i in 1:10
        θ = self.θ_0 * (11 - i)^2
        adjust_windows(self, div(self.anneal_steps, 2), θ)
        sample(self, div(self.anneal_steps, 2), θ)
    end
end

function main_anneal(self::SAC, wr::Bool=false; on_stage=nothing)      # free:1051-1094
    θ = self.θ_0
    N_τ = self.td.n_tau
    for i in 1:self.N_anneal
        adjust_windows(self, self.anneal_steps, θ)
        sample(self, self.anneal_steps, θ)
        if on_stage !== nothing
            on_stage(i, θ)                                  # cross_val.jl:105-108 hooks in here
        end
        open(anneal_file(self), "a") do f
            println(f, fmt(Float64(i)), ",", fmt(round(θ, digits=8)), ",", fmt(round(self.χ2_min / N_τ, digits=4)), ",",
                    fmt(round(self.sampled_χ2 / N_τ, digits=4)))
        end
        open(accept_rate_file(self), "a") do f
            print(f, fmt(Float64(i)))
            for v in self.accept_rates
                print(f, ",", fmt(round(v, digits=8)))
            end
            for k in (1, 2, 4)
                print(f, ",", fmt(round(self.update_windows[k] * self.δω, digits=8)))
            end
            println(f)
        end
        self.χ2_anneal[i] = self.sampled_χ2
        if wr
            write_spec(self, i)
        end
        if (self.sampled_χ2 - self.χ2_min) < (self.tol * N_τ)           # free:1087
            return
        end
        θ /= self.f_anneal
    end
end

function optimal_θ(self::SAC)                               # free:1249-1252
    best = 1
    best_d = Inf
    for i in 1:self.N_anneal
        a = (self.χ2_anneal[i] - self.χ2_min) / sqrt(2 * self.χ2_min)
        if abs(a - self.a2) < best_d
            best_d = abs(a - self.a2)
            best = i
        end
    end
    return self.θ_0 / self.f_anneal^(best - 1)
end

function final_anneal(self::SAC, θ_opt::Float64)            # free:1102-1145
    N_τ = self.td.n_tau
    for i in 1:10
        θ = θ_opt * (11 - i)
        adjust_windows(self, self.anneal_steps * i, θ)
        sample(self, self.anneal_steps * i, θ)
    end
    n_final = self.a1 == self.a2 ? 1 : 20
    θ = θ_opt
    for n in 1:n_final
        sample(self, self.sample_steps, θ)
        a = (self.sampled_χ2 - self.χ2_min) / sqrt(2 * self.χ2_min)    # free:1128
        open(sample_file(self), "a") do f
            println(f, fmt(Float64(n - 1)), ",", fmt(a), ",", fmt(θ), ",", fmt(self.χ2_min / N_τ), ",", fmt(self.sampled_χ2 / N_τ))
        end
        write_spec(self, n - 1)
        if a < self.a1
            return
        end
        θ /= self.f_final
    end
end

# free:1163-1179 + the SAC(...) constructor call: in_free.in -> struct (no files touched)
function load_input(in_file::String="in_free.in")
    rows = map(split, readlines(in_file))
    par = Int(parse(Float64, rows[1][1]))
    N_ω = Int(parse(Float64, rows[2][1]))
    ω_0, ω_m, δω, δω_h = parse.(Float64, rows[2][2:5])
    θ_0, f_anneal, f_final, a1, a2 = parse.(Float64, rows[3][1:5])
    N_anneal, anneal_steps, sample_steps = Int.(parse.(Float64, rows[4][1:3]))
    G_file, output_folder = String(rows[5][1]), String(rows[5][2])
    symm = Int(parse(Float64, rows[6][1]))
    kernel_type = String(rows[6][2])
    if kernel_type == "bosonic"                             # free:1172-1174
        ω_0 = 0.0
        symm = 0
    elseif symm == 1                                        # free:1176-1179
        ω_0 = 0.0
        output_folder *= "_symm"
    end
    return SAC(par, N_ω, ω_0, ω_m, δω, δω_h, θ_0, f_anneal, f_final, a1, a2, N_anneal, anneal_steps, sample_steps,
               G_file, output_folder, symm, kernel_type)
end

# free:1160-1261
function run(in_file::String="in_free.in")
    sac = load_input(in_file)
    mkpath(sac.output_folder)
    cp(in_file, joinpath(sac.output_folder, "in_free.in"); force=true)
    cp(sac.G_file, joinpath(sac.output_folder, "t.in"); force=true)
    init_outfiles(sac)
    write_log(sac, "Beginning Initialization.")
    read_G!(sac)
    initialize!(sac)
    reset_windows!(sac)                                     # free:1218
    write_log(sac, "Initialization Finished.")
    write_log(sac, "Beginning Equillibration.")
    fast_anneal(sac)
    write_log(sac, "Equillibration Finished.")
    write_log(sac, "Beginning Main Anneal.")
    main_anneal(sac, false)
    write_log(sac, "Main Anneal Finished.")
    write_log(sac, "Beginning Final Anneal.")
    θ_opt = optimal_θ(sac)
    reset_windows!(sac)                                     # free:1254
    final_anneal(sac, θ_opt)
    write_log(sac, "Final Anneal Finished.")
    SacHip.destroy(sac.handle)
    return θ_opt
end

end # module

if abspath(PROGRAM_FILE) == @__FILE__
    SACs.run()
end
