# sac_edge_hip.jl -- the `edge` program of gabes135/sac (sac/edge/sac_edge.jl, "edge:NNN") with its fixed-Theta stage
# run_bins(self, steps, bins, θ) running on an MI355X through libsac_hip.so (sac_edge_*).
#
#   julia sac_edge_hip.jl [A_c A_r p]     reads in_edge.in, as `julia sac_edge.jl [A_c A_r p]` does (edge:2068-2071)
#
# Same inputs, same schedule (anneal / anneal_and_sample / anneal_and_scan), same output layout (anneal.csv, sample.csv,
# accept_rate_{1,2}.csv, sw%03d_R.dat, dw%03d_R.dat, log.txt).  Host side, as in the reference: read_G!'s edge guess, the
# K / dK / d2K tables of init_kernal!, set_amplitudes!, the deterministic 50 x 50 scan init_config_dual!.
# Environment as in sac_free_hip.jl (SAC_N_CHAINS, SAC_SEED, SAC_DEVICE, SAC_CHI2_MIN_MODE).
# Written against the ABI; not executed by this repository's test-suite (no Julia in the build image).
include(joinpath(@__DIR__, "SacHip.jl"))

module SACs

using Printf
using ..SacHip
import ..SacHip: TimeData, read_tin, kernel_value, fmt

mutable struct SAC
    N_e::Int
    N_c::Int
    N_ω::Int
    ω_0::Float64
    ω_m::Float64
    δω_h::Float64
    δω::Float64
    p::Float64
    c::Float64
    A_c::Float64
    A_r::Float64
    θ_0::Float64
    f_anneal::Float64
    N_anneal::Int
    a_criterion::Float64
    anneal_steps::Int
    sample_steps::Int
    bins::Int
    G_file::String
    output_folder::String
    fix_edge_at::Float64
    fix_edge::Int
    kernel_type::String
    mode::Symbol
    tol::Float64
    td::Union{TimeData,Nothing}
    ωi_0::Int
    ωi_m::Int
    N_ωi::Int
    n_rl::Int
    ω_floor::Vector{Float64}
    edge_guess::Float64
    K::Array{Float64,3}                # [N_τ, N_ωi, n_rl]
    dK::Array{Float64,3}
    d2K::Array{Float64,3}
    ω_array::Matrix{Float64}           # [N_ω + 1, 2], row 1 = the reference's unused leading slot
    d_array::Matrix{Float64}
    A_array::Matrix{Float64}
    ∆ω_array::Matrix{Float64}
    n_multiδ::Matrix{Int64}
    n0::Vector{Float64}
    ε_0::Vector{Float64}
    ∆n0::Vector{Float64}
    ∆ε_0::Vector{Float64}
    j_opt::Int
    k_opt::Int
    res::Union{SacHip.SacEdgeResults,Nothing}
    χ2_min::Float64
    χ2_anneal::Vector{Float64}
    saved::Vector{Any}
    handle::SacHip.Handle
    n_chains::Int
    seed::UInt64
    device::Int
    χ2_min_mode::String
end

path(self::SAC, name::String) = joinpath(self.output_folder, name)
write_log(self::SAC, msg::String, mode::String="a") = SacHip.logline(self.output_folder, msg; mode=mode)
r3(t, q, rl) = t[2 * (q - 1) + rl]     # element [q, rl] of a C double[3][2] carried as an NTuple{6}

# edge:130-160: t.in and the guess of the edge from the log-slope of G at the last two tau <= beta ÷ 4 (edge:150-153)
function read_G!(self::SAC)
    td = read_tin(self.G_file)
    self.td = td
    last = 0
    for i in 1:td.n_tau
        if td.tau[i] <= fld(td.beta, 4)
            last = i
        end
    end
    self.edge_guess = -(log(td.G_raw[last]) - log(td.G_raw[last - 1])) / (td.tau[last] - td.tau[last - 1])
end

# edge:167-242: K, central-difference dK and d2K (the Taylor 1/2 folded into d2K), one-sided dK at column 1 and at the
# column the reference addresses as ωi_m + 1 (edge:235)
function init_kernal!(self::SAC)
    td = self.td
    self.ωi_0 = floor(Int, self.ω_0 / self.δω)
    self.ωi_m = ceil(Int, self.ω_m / self.δω)
    self.N_ωi = self.ωi_m - self.ωi_0 + 1
    self.n_rl = (self.mode == :double_edge_in || self.mode == :double_edge_out) ? 2 : 1
    n, nt = self.N_ωi, td.n_tau
    self.K = zeros(nt, n, self.n_rl)
    self.dK = zeros(nt, n, self.n_rl)
    self.d2K = zeros(nt, n, self.n_rl)
    col = zeros(nt)
    for rl in 1:self.n_rl
        for cidx in 1:n
            ω = (self.ωi_0 + cidx - 1) * self.δω            # edge:195
            for j in 1:nt
                if self.mode == :double_edge_symm           # edge:223
                    col[j] = kernel_value(self.kernel_type, ω, td.tau[j], td.beta) + kernel_value(self.kernel_type, -ω, td.tau[j], td.beta)
                else
                    col[j] = kernel_value(self.kernel_type, rl == 1 ? ω : -ω, td.tau[j], td.beta)
                end
            end
            self.K[:, cidx, rl] = td.cov' * col
        end
        for cidx in 2:(n - 1)                               # edge:230-233
            for j in 1:nt
                self.dK[j, cidx, rl] = (self.K[j, cidx + 1, rl] - self.K[j, cidx - 1, rl]) / (2 * self.δω)
                self.d2K[j, cidx, rl] = (self.K[j, cidx + 1, rl] - (2 * self.K[j, cidx, rl]) + self.K[j, cidx - 1, rl]) / (2 * self.δω^2)
            end
        end
        e = self.ωi_m + 1
        for j in 1:nt
            self.dK[j, 1, rl] = (self.K[j, 2, rl] - self.K[j, 1, rl]) / self.δω                  # edge:234
            self.dK[j, e, rl] = (self.K[j, e, rl] - self.K[j, e - 1, rl]) / self.δω              # edge:235
        end
    end
end

# edge:354-390, one side
function set_amplitudes!(self::SAC, rl::Int)
    NE, N = self.N_e, self.N_ω
    ε1 = (log(2) * self.c)^2 * self.ε_0[rl]
    tot = 0.0
    for i in 1:NE
        x = log(i / self.n0[rl])
        root = sqrt((self.c * x)^2 + ε1)
        lnA = self.c < 0 ? self.c * x + root : self.c * x - root
        self.A_array[i + 1, rl] = exp(lnA / 2)
        tot += self.A_array[i + 1, rl]
    end
    f1 = rl == 1 ? self.A_r / (1 + self.A_c) : (1 - self.A_r) / (1 + self.A_c)
    f2 = rl == 1 ? self.A_c / (1 + self.A_c) : 0.0
    for i in 1:NE
        self.A_array[i + 1, rl] *= f1 / tot
    end
    for i in (NE + 1):N
        self.A_array[i + 1, rl] = f2 / (N - NE)
    end
end

# Taylor lookup of calc_Gbar! (edge:595-606, 626-633)
function lookup(self::SAC, rl::Int, ω::Float64)
    i = floor(Int, ω / self.δω)
    d = ω - i * self.δω
    cidx = clamp(i - self.ωi_0 + 1, 1, self.N_ωi)
    return (self.K[:, cidx, rl] .+ self.dK[:, cidx, rl] .* d) .+ self.d2K[:, cidx, rl] .* d^2
end

function host_χ2(self::SAC)
    td = self.td
    g = zeros(td.n_tau)
    for rl in 1:self.n_rl
        for k in 1:(self.N_ω + 1)
            g .+= lookup(self, rl, self.ω_array[k, rl]) .* self.A_array[k, rl]
        end
    end
    return sum(((g .- td.G) .* td.sigma_inv) .^ 2)
end

# edge:399-481
function init_config_jk!(self::SAC, j::Int, k::Int)
    ω, d, NE, N, ωm = self.ω_array, self.d_array, self.N_e, self.N_ω, self.ω_m
    g = sign(self.edge_guess)
    k0 = 55
    dd = 0.0
    if self.fix_edge == 1
        ω[2, 1] = ω[1, 1]
        ω[2, 2] = ω[1, 2]
        dd = abs((ωm * (k + 5) / k0) - ω[2, 1])
    elseif self.mode == :single_edge
        ω[2, 1] = ω[1, 1] + (g * ((j - 1) * ωm / 500))
        dd = abs((ωm * (k + 5) / k0) - ω[2, 1])
    elseif self.mode == :double_edge_in
        ω[2, 1] = ω[1, 1] + (g * (j * ωm / 500))
        ω[2, 2] = ω[2, 1] - ((k + 5) * ωm / k0)
        dd = abs(-ω[2, 2] - ω[2, 1]) * 0.5
    elseif self.mode == :double_edge_out
        ω[2, 1] = ω[1, 1] + (j * ωm / 500)
        ω[2, 2] = ω[1, 2] + (k * ωm / 500)
        dd = abs((ωm * 0.5) - ω[2, 1])
    else
        ω[2, 1] = ω[1, 1] + (j * ωm / 500)
        dd = abs((ωm * (k + 5) / k0) - ω[2, 1])
    end
    tot = 0.0
    for i in 1:NE
        d[i + 1, 1] = sqrt(i)
        d[i + 1, 2] = sqrt(i)
        if i >= 2
            tot += 2 * sqrt(i)                              # both columns enter the normalisation (edge:448)
        end
    end
    for i in 2:NE
        d[i + 1, 1] *= dd / tot
        d[i + 1, 2] *= dd / tot
    end
    for i in 2:NE
        for rl in 1:self.n_rl
            ω[i + 1, rl] = ω[i, rl] + d[i + 1, rl]
        end
    end
    if NE < N                                               # edge:458-473
        for i in (NE + 1):N
            ω[i + 1, 2] = ω[3, 2]
        end
        ω[NE + 2, 1] = ω[3, 1]
        dc = self.mode == :double_edge_in ? abs(-ω[2, 2] - ω[NE + 2, 1]) * 0.5 : abs(ω[NE + 1, 1] - ω[NE + 2, 1]) * 0.5
        for lin in (NE + 2):length(d)                       # d_array[N_e+2:end], linear index over both columns
            d[lin] = dc / (N - NE)
        end
        for i in (NE + 2):N
            ω[i + 1, 1] = ω[i, 1] + dc / (N - NE)
        end
    end
end

# edge:487-583: the deterministic scan over (j, k) for the best initial edge
function init_config_dual!(self::SAC)
    NE, N = self.N_e, self.N_ω
    j_max, k_max = 50, 50
    if self.fix_edge == 1
        self.ω_array[1, 1] = self.ω_floor[1]
        self.ω_array[1, 2] = self.ω_floor[2]
        j_max = 1
    else
        self.ω_array[1, 1] = 0.0
        self.ω_array[1, 2] = 0.0
    end
    best, k_opt, j_opt = 1e20, 1, 1
    for k in 1:k_max
        for j in 1:j_max
            init_config_jk!(self, j, k)
            if self.mode == :double_edge_in
                if self.ω_array[2, 1] > -self.ω_array[NE + 1, 2] || maximum(self.ω_array[:, 1]) > -self.ω_array[2, 2]
                    continue
                end
            end
            χ2 = host_χ2(self)
            if χ2 < best
                k_opt, j_opt, best = k, j, χ2
            end
        end
    end
    init_config_jk!(self, j_opt, k_opt)
    self.j_opt, self.k_opt = j_opt, k_opt
    for rl in 1:self.n_rl
        for i in 2:(NE - 1)
            self.n_multiδ[i, rl] = min(1 + div(N, 20), NE - i - 1)      # edge:568-571
        end
        self.∆ω_array[2, rl] = self.d_array[3, rl]          # edge:575-578
        for i in 3:(N + 1)
            self.∆ω_array[i, rl] = self.d_array[i, rl]
        end
    end
end

function init_outfiles(self::SAC)                           # edge:305-334
    head = "j,theta,chi2_min,chi2_avg,chi2_sigma,edge_R,wn0_R,n0_R,eps_R,edge_R_sigma,wn0_R_sigma,n0_R_sigma," *
           "edge_L,wn0_L,n0_L,eps_L,edge_L_sigma,wn0_L_sigma,n0_L_sigma,wc0,w_cf"
    for name in ("anneal.csv", "sample.csv")
        open(path(self, name), "w") do f
            println(f, head)
        end
    end
    for name in ("accept_rate_1.csv", "accept_rate_2.csv")
        open(path(self, name), "w") do f
            println(f, "j,ar_cont,∆w_cont,ar_cont_amp,ar_edge_s,∆w_edge,ar_edge_m,n_clust,ar_edge_amp")
        end
    end
end

# edge:2038-2049 + the hand-over to the GPU
function initialize!(self::SAC)
    write_log(self, "Initializing Kernel.")
    read_G!(self)
    init_kernal!(self)
    N1 = self.N_ω + 1
    self.ω_array = zeros(N1, 2); self.d_array = zeros(N1, 2); self.A_array = zeros(N1, 2)
    self.∆ω_array = zeros(N1, 2); self.n_multiδ = zeros(Int64, N1, 2)
    self.n0 = fill(0.75 * self.N_e, 2); self.ε_0 = fill(0.5, 2); self.∆n0 = fill(2.0, 2); self.∆ε_0 = fill(0.05, 2)
    for rl in 1:self.n_rl
        set_amplitudes!(self, rl)
    end
    write_log(self, "Finding Best Initial Configuration.")
    init_config_dual!(self)
    write_log(self, "Rightward Edge: [$(fmt(self.ω_array[2, 1])), $(fmt(self.ω_array[self.N_e + 1, 1]))] ($(self.j_opt), $(self.k_opt))")
    if self.n_rl == 2                                       # edge:555-557
        write_log(self, "Leftward Edge: [$(fmt(-self.ω_array[2, 2])), $(fmt(-self.ω_array[self.N_e + 1, 2]))]")
    end
    if self.handle != C_NULL
        SacHip.edge_destroy(self.handle)
    end
    cfg = SacHip.SacEdgeConfig(self.td.n_tau, self.N_e, self.N_ω, self.n_rl, self.mode == :double_edge_in ? 1 : 0, self.fix_edge,
                               self.n_chains, self.device, self.ωi_0, self.N_ωi, self.A_c, self.A_r, 0.0,
                               (self.ω_floor[1], self.ω_floor[2]), self.ω_m, self.δω, (self.c, self.c), self.seed, 0, 0, 0, 0)
    self.handle = SacHip.edge_create(cfg)
    SacHip.edge_upload_problem(self.handle, self.K, self.dK, self.d2K, self.td.G, self.td.sigma_inv)
    SacHip.edge_set_state(self.handle; w=self.ω_array, A=self.A_array, dw=self.∆ω_array, n_multi=max.(self.n_multiδ, 0),
                          scal8=vcat(self.n0, self.ε_0, self.∆n0, self.∆ε_0))
    SacHip.edge_init_gbar(self.handle)                      # calc_Gbar!(self, 1); χ2; χ2_min = χ2
    _, _, self.χ2_min = SacHip.edge_get_gbar(self.handle, 0, self.td.n_tau)
    write_log(self, "Initialization Finished.")
end

# edge:1357: the one stage function the annealing routines call
function run_bins(self::SAC, steps::Int, bins::Int, θ::Float64)
    SacHip.edge_run_bins(self.handle, steps, bins, θ)
    r = SacHip.edge_get_results(self.handle, -1)
    self.res = r
    self.χ2_min = self.χ2_min_mode == "min" ? r.chi2_min : r.chi2_min_mean
    if r.violations > 0                                     # the reference exit()s here (check_ω, edge:668)
        open(path(self, "error.txt"), "a") do f
            println(f, fmt(θ), " ordering / bound violations in ", r.violations, " chain-bins")
        end
        error("check_ω: ordering or bound violation (see error.txt)")
    end
end

function result_row(self::SAC, θ::Float64)
    r, N_τ = self.res, self.td.n_tau
    return Float64[θ, self.χ2_min / N_τ, r.chi2_res[2] / N_τ, r.chi2_res[3] / N_τ,
                   r3(r.edge_res, 2, 1), r3(r.wn0_res, 2, 1), r3(r.n0_res, 2, 1), r.eps0[1],
                   r3(r.edge_res, 3, 1), r3(r.wn0_res, 3, 1), r3(r.n0_res, 3, 1),
                   -r3(r.edge_res, 2, 2), -r3(r.wn0_res, 2, 2), r3(r.n0_res, 2, 2), r.eps0[2],
                   r3(r.edge_res, 3, 2), r3(r.wn0_res, 3, 2), r3(r.n0_res, 3, 2)]
end

# edge:1452-1510
function write_res(self::SAC, j::Int, θ::Float64)
    NE, N = self.N_e, self.N_ω
    dwm, nmm, ext = SacHip.edge_get_move_means(self.handle, N)
    vals = result_row(self, θ)
    push!(vals, self.A_c > 0 ? ext[1] : 0.0, self.A_c > 0 ? ext[2] : 0.0)
    open(path(self, "anneal.csv"), "a") do f
        println(f, j, ",", join(map(v -> fmt(round(v, digits=8)), vals), ","))
    end
    for rl in 1:self.n_rl
        ar1 = SacHip.edge_get_accept_rates(self.handle, -1, 0, rl - 1, N)
        ar2 = SacHip.edge_get_accept_rates(self.handle, -1, 1, rl - 1, N)
        nc = N - NE
        a1 = sum(ar1[(NE + 2):(N + 1)]) / nc                # 0 / 0 = NaN without continuum, as in the reference
        a2 = sum(dwm[(NE + 2):(N + 1), rl]) / nc
        a3 = r3(self.res.amp_accept_rates, 2, rl)
        a4 = sum(ar1[2:(NE + 1)]) / NE
        a5 = sum(dwm[2:(NE + 1), rl]) / NE
        a6 = sum(ar2[3:(NE - 1)]) / (NE - 3)
        a7 = sum(nmm[3:(NE - 1), rl]) / (NE - 3)
        a8 = r3(self.res.amp_accept_rates, 1, rl)
        open(path(self, "accept_rate_$(rl).csv"), "a") do f
            println(f, j, ",", join(map(v -> fmt(round(v, digits=8)), (a1, a2, a3, a4, a5, a6, a7, a8)), ","))
        end
    end
end

# edge:1514-1637: the histogram on the δω_h grid (sw) and the density on the self-generated grid of the averaged deltas (dw)
function write_spec(self::SAC, n::Int, total_steps::Int, rl::Int)
    td, NE = self.td, self.N_e
    ωi_0, ωi_m, δω, δω_h = self.ωi_0, self.ωi_m, self.δω, self.δω_h
    ngrid = ωi_m + 1 - ωi_0
    spec_bin = zeros(ngrid, 2)
    spec_bin[:, 1] = SacHip.edge_get_histogram(self.handle, 0, rl - 1, ngrid) ./ total_steps .* (td.norm * pi)
    spec_bin[:, 2] = SacHip.edge_get_histogram(self.handle, 1, rl - 1, ngrid) ./ total_steps .* (td.norm * pi)
    ω_avg, A_avg = SacHip.edge_get_avg(self.handle, -1, self.N_ω)
    hist = zeros(ngrid + 2, 2)
    conv = ceil(Int, δω_h / δω)
    N_h = floor(Int, (ωi_m - ωi_0) / conv)
    for i in 1:N_h                                          # edge:1536-1537
        for q in 1:2
            s = 0.0
            for cidx in (2 + (i - 1) * conv):(1 + i * conv)
                s += spec_bin[cidx, q]
            end
            hist[i, q] = s / δω_h
        end
    end
    j1 = 1
    for i in 0:N_h                                          # edge:1541-1548
        j1 = i + 1
        if max(hist[i + 1, 1], hist[i + 1, 2]) > 1e-10
            j1 -= 1
            break
        end
    end
    j1 = max(1, j1)
    j2 = N_h + 1
    for i in N_h:-1:0                                       # edge:1550-1555
        j2 = i + 1
        if max(hist[i + 1, 1], hist[i + 1, 2]) > 1e-10
            break
        end
    end
    open(path(self, @sprintf("sw%03d_%d.dat", n, rl)), "w") do f
        println(f, "omega,S,S_edge,S_cont")
        println(f, fmt(δω_h * (j1 - 1 - 0.5) + self.ω_0), ",0,0,0")
        for i in j1:j2
            ω = δω_h * (i - 0.5) + self.ω_0
            println(f, fmt(ω), ",", fmt(hist[i, 1] + hist[i, 2]), ",", fmt(hist[i, 1]), ",", fmt(hist[i, 2]))
        end
        println(f, fmt(δω_h * (j2 + 1 - 0.5) + self.ω_0), ",0,0,0")
    end
    ωb = ω_avg[:, rl] ./ total_steps                        # edge:1573-1635; index k = the reference's k
    Ab = A_avg[:, rl] ./ total_steps
    jj = NE - 1
    i2 = 0
    for i in 1:jj
        gap = ωb[i + 2] - ωb[i + 1]
        hist[i + 1, 1] = ((td.norm * pi) / gap) * 0.5 * (Ab[i + 1] + Ab[i + 2])
        i1 = ceil(Int, ωb[i + 1] / δω) - ωi_0
        i2 = ceil(Int, ωb[i + 2] / δω) - 1 - ωi_0
        s = 0.0
        for cidx in max(i1, 1):max(i2, 0)
            s += spec_bin[cidx, 2]
        end
        hist[i + 1, 2] = s / gap
    end
    δω_h2 = ωb[jj + 2] - ωb[jj + 1]
    conv2 = floor(Int, (δω_h2 / δω) + 0.5)
    kk = jj
    while true                                              # edge:1601-1616
        i1 = i2 + 1
        i2 = i1 + conv2
        if i2 > (ωi_m - ωi_0)
            break
        end
        kk += 1
        hist[kk, 1] = 0.0
        s = 0.0
        for cidx in i1:i2
            s += spec_bin[cidx, 2]
        end
        hist[kk, 2] = s / δω_h2
        if hist[kk, 2] < 1.0e-10
            kk -= 1
            break
        end
    end
    open(path(self, @sprintf("dw%03d_%d.dat", n, rl)), "w") do f
        println(f, "omega,S,S_edge,S_cont")
        println(f, fmt(ωb[2]), ",0,0,0")
        ω = 0.0
        for i in 1:jj
            ω = 0.5 * (ωb[i + 1] + ωb[i + 2])
            println(f, fmt(ω), ",", fmt(hist[i + 1, 1] + hist[i + 1, 2]), ",", fmt(hist[i + 1, 1]), ",", fmt(hist[i + 1, 2]))
        end
        for i in (jj + 1):kk
            ω += δω_h2
            println(f, fmt(ω), ",", fmt(hist[i + 1, 1] + hist[i + 1, 2]), ",", fmt(hist[i + 1, 1]), ",", fmt(hist[i + 1, 2]))
        end
        println(f, fmt(ω + δω_h2 / 2.0), ",0,0,0")
    end
end

# edge:1646-1693.  first_passage (many chains; no counterpart in the one-chain reference): the target rule is taken chain by
# chain -- each chain has its own target (edge_set_targets_from_chi2_min) and the anneal stops when half of all chains have
# had their stage average below it: the MEDIAN of the chains' first passages.  With one chain it is the reference's test.
function anneal(self::SAC, χ2_target::Float64, bins::Int, spec::Int; first_passage::Bool=false)
    θ = self.θ_0
    i_trans = self.N_anneal * 0.2
    N_τ = self.td.n_tau
    for i in 1:self.N_anneal
        steps = i < i_trans ? ceil(Int, self.anneal_steps * (1 - ((5 / 6) * i / i_trans))) : ceil(Int, self.anneal_steps / 6)
        run_bins(self, steps, bins, θ)
        write_res(self, i, θ)
        if spec == 1
            for rl in 1:self.n_rl
                write_spec(self, i, bins * steps, rl)
            end
        end
        if i == 5                                           # edge:1671-1675: every chain's (ω, ∆ω, n_multiδ)
            empty!(self.saved)
            for chain in 0:(self.n_chains - 1)
                push!(self.saved, SacHip.edge_get_state(self.handle, chain, self.N_ω))
            end
        end
        χ2_avg = self.res.chi2_res[2]
        reached = first_passage ? 2 * self.res.chains_below_target >= self.res.n_chains_total : χ2_avg < χ2_target
        if reached
            return θ * self.f_anneal
        elseif (χ2_avg - self.χ2_min) < (self.tol * N_τ)
            return θ
        else
            θ /= self.f_anneal
            self.χ2_anneal[i] = χ2_avg
        end
    end
    return θ
end

function restore_saved!(self::SAC)                          # edge:1724-1730
    for (k, st) in enumerate(self.saved)
        w, _, dw, nm, _ = st
        SacHip.edge_set_state(self.handle; chain=k - 1, w=w, dw=dw, n_multi=nm)
    end
    SacHip.edge_init_gbar(self.handle)
end

function sample_row(self::SAC, j::Int, θ::Float64, with_cont::Bool)
    vals = result_row(self, θ)
    if with_cont
        _, _, ext = SacHip.edge_get_move_means(self.handle, self.N_ω)
        push!(vals, self.A_c > 0 ? ext[1] : 0.0, self.A_c > 0 ? ext[2] : 0.0)
    end
    open(path(self, "sample.csv"), "a") do f
        println(f, j, ",", join(map(v -> fmt(round(v, digits=8)), vals), ","))
    end
end

# edge:1705-1774
function anneal_and_sample(self::SAC)
    empty!(self.saved)
    write_log(self, "Beginning Main Anneal.")
    anneal(self, 0.0, self.bins, 0)
    write_log(self, "Main Anneal Finished.")
    χ2_target = self.χ2_min + self.a_criterion * sqrt(2 * self.χ2_min)             # edge:1718
    first_passage = self.χ2_min_mode != "min"
    if first_passage                                        # per-chain targets from per-chain χ2_min, before the restore resets it
        SacHip.edge_set_targets_from_chi2_min(self.handle, self.a_criterion)
    end
    if !isempty(self.saved)
        restore_saved!(self)
    end
    write_log(self, "Beginning Final Anneal.")
    θ_opt = anneal(self, χ2_target, self.bins, 0; first_passage=first_passage)
    if first_passage
        SacHip.edge_set_chain_targets(self.handle, nothing)
    end
    write_log(self, "Final Anneal Finished.")
    write_log(self, "Beginning Final Sampling.")
    run_bins(self, self.sample_steps, 10, θ_opt)
    for rl in 1:self.n_rl
        write_spec(self, 0, 10 * self.sample_steps, rl)
    end
    sample_row(self, 0, θ_opt, true)
    write_log(self, "Final Sampling Finished.")
    return θ_opt
end

# edge:1783-1843
function anneal_and_scan(self::SAC, θ_1::Float64, θ_2::Float64)
    self.N_anneal = ceil(Int, log(self.θ_0 / θ_1) / log(self.f_anneal))
    self.χ2_anneal = zeros(self.N_anneal)
    write_log(self, "Beginning First Anneal.")
    empty!(self.saved)
    write_log(self, "Beginning Main Anneal.")
    anneal(self, 0.0, div(self.bins, 2), 0)
    write_log(self, "Main Anneal Finished.")
    if self.χ2_min > 2 * self.td.n_tau                      # edge:1800-1803
        return anneal_and_scan(self, θ_1, θ_2)
    end
    write_log(self, "Beginning Final Sampling.")
    f_scan = 1.15
    N_scan = ceil(Int, log(θ_1 / θ_2) / log(f_scan))
    θ = θ_1
    for i in 1:N_scan
        bins = ceil(Int, self.bins * (1 + i / N_scan))
        run_bins(self, self.sample_steps, bins, θ)
        sample_row(self, i, θ, false)
        θ /= f_scan
        for rl in 1:self.n_rl
            write_spec(self, i, bins * self.sample_steps, rl)
        end
    end
    write_log(self, "Final Sampling Finished.")
end

# edge:1884-2035 + the SAC(...) constructor call: in_edge.in -> struct (no files touched)
function load_input(A_c_in=nothing, A_r_in=nothing, p_in=nothing; in_file::String="in_edge.in", scan::Bool=false)
    rows = map(split, readlines(in_file))
    N_e, N_c = Int(parse(Float64, rows[1][1])), Int(parse(Float64, rows[1][2]))
    ω_0, ω_m, δω_h, δω = parse.(Float64, rows[2][1:4])      # edge:1888: ω_0, ω_m, δω_h, δω
    p, A_c, A_r = parse.(Float64, rows[3][1:3])
    θ_0, f_anneal = parse(Float64, rows[4][1]), parse(Float64, rows[4][2])
    N_anneal, a_criterion = Int(parse(Float64, rows[4][3])), parse(Float64, rows[4][4])
    anneal_steps, sample_steps, bins = Int.(parse.(Float64, rows[5][1:3]))
    G_file, out = String(rows[6][1]), String(rows[6][2])
    fix_edge_at, kernel_type = parse(Float64, rows[7][1]), String(rows[7][2])
    mode = Symbol(rows[8][1])
    A_c = A_c_in === nothing ? A_c : A_c_in
    A_r = A_r_in === nothing ? A_r : A_r_in
    p = p_in === nothing ? p : p_in
    if mode == :single_edge                                 # edge:1909-1921
        out *= "_single/"
        A_r = 1.0
    elseif mode == :double_edge_in
        out *= "_double_in/"
    elseif mode == :double_edge_out
        out *= "_double_out/"
        ω_0 = 0.0
    elseif mode == :double_edge_symm
        out *= "_double_symm/"
        ω_0 = 0.0
        A_r = 0.5
    else
        error("mode $mode")
    end
    settings = String[]
    if fix_edge_at != 0
        push!(settings, "fixed")
    end
    if A_c > 0.0                                            # edge:1931-1942
        if N_c == 0
            N_c = ceil(Int, 0.5 * N_e)
        end
        @assert A_c < 1
    end
    if A_c == 0.0
        N_c = 0
    end
    push!(settings, "Nw$(N_e)")
    if scan
        push!(settings, "scan")
    end
    out *= join(settings, "_") * @sprintf("/Ac_%.3f", A_c) * @sprintf("/p_%.3f", p)
    if mode == :double_edge_in || mode == :double_edge_out
        out *= @sprintf("/Ar_%.3f", A_r)
    end
    if kernel_type == "bosonic"                             # edge:1964-1969
        mode == :single_edge || error("Only mode that can be run with the bosonic kernel is single_edge. Exiting.")
        ω_0 = 0.0
    end
    fix_edge = fix_edge_at != 0 ? 1 : 0
    ω_floor = mode == :double_edge_in ? [ω_0, -ω_m] : (fix_edge == 1 ? [fix_edge_at, 1.5 * fix_edge_at] : [ω_0, ω_0])   # edge:1999-2015
    N_ω = N_e + N_c
    return SAC(N_e, N_c, N_ω, ω_0, ω_m, δω_h, δω, p, 1 - 2 * p, A_c, A_r, θ_0, f_anneal, N_anneal, a_criterion, anneal_steps,
               sample_steps, bins, G_file, out, fix_edge_at, fix_edge, kernel_type, mode, 1e-3, nothing, 0, 0, 0, 1, ω_floor, 0.0,
               zeros(0, 0, 0), zeros(0, 0, 0), zeros(0, 0, 0), zeros(0, 0), zeros(0, 0), zeros(0, 0), zeros(0, 0), zeros(Int64, 0, 0),
               zeros(2), zeros(2), zeros(2), zeros(2), 0, 0, nothing, 0.0, zeros(N_anneal), Any[], C_NULL,
               parse(Int, get(ENV, "SAC_N_CHAINS", "1")), parse(UInt64, get(ENV, "SAC_SEED", "0")),
               parse(Int, get(ENV, "SAC_DEVICE", "0")), get(ENV, "SAC_CHI2_MIN_MODE", "chain_mean"))
end

# edge:1882-2065
function run(A_c_in=nothing, A_r_in=nothing, p_in=nothing; in_file::String="in_edge.in", θ_1=nothing, θ_2=nothing)
    sac = load_input(A_c_in, A_r_in, p_in; in_file=in_file, scan=θ_1 !== nothing)
    out, mode, N_ω, N_e, A_r, A_c, p = sac.output_folder, sac.mode, sac.N_ω, sac.N_e, sac.A_r, sac.A_c, sac.p
    anneal_steps, bins, θ_0 = sac.anneal_steps, sac.bins, sac.θ_0
    mkpath(out)
    cp(in_file, path(sac, "in_edge.in"); force=true)
    cp(sac.G_file, path(sac, "t.in"); force=true)
    write_log(sac, "N_ω, N_e, A_r, A_c, p = $(N_ω), $(N_e), $(fmt(A_r)), $(fmt(A_c)),$(fmt(p))", "w")
    write_log(sac, "Beginning Initialization.")
    if mode == :single_edge || mode == :double_edge_symm
        write_log(sac, "Only using rightward decaying spectrum.")
    else
        write_log(sac, "Using rightward and leftward decaying spectrum (double edge).")
    end
    write_log(sac, "Reading G_file.")
    init_outfiles(sac)
    initialize!(sac)
    write_log(sac, "Running Equillibration Sweeps.")
    run_bins(sac, 5 * anneal_steps, bins, θ_0)              # edge:2052-2053
    write_res(sac, 0, θ_0)
    result = θ_1 === nothing ? anneal_and_sample(sac) : anneal_and_scan(sac, θ_1, θ_2)
    SacHip.edge_destroy(sac.handle)
    return result
end

end # module

if abspath(PROGRAM_FILE) == @__FILE__
    if length(ARGS) >= 3
        SACs.run(parse(Float64, ARGS[1]), parse(Float64, ARGS[2]), parse(Float64, ARGS[3]))
    else
        SACs.run()
    end
end
