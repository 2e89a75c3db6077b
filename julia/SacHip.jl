# SacHip.jl -- Julia binding of libsac_hip.so (include/sac_hip.h), the MI355X sampling engine.
#
# One `ccall` wrapper per entry point of the C ABI; the structs mirror the C structs field for field
# (tests/test_julia_layout.py compares them with the header through a compiled sizeof/offsetof probe).
# The drivers next to this file (sac_free_hip.jl, sac_peak_hip.jl, sac_edge_hip.jl, cross_val_hip.jl) keep the
# entry points, input files and output layout of gabes135/sac's sac_free.jl / sac_peak.jl / sac_edge.jl /
# cross_val.jl and call the fixed-Theta stages through this module.
#
# Julia is not part of the build image of this repository: these files are written against the ABI and
# checked structurally, not executed by the test-suite (the Python mirror sac_amd/ is what the GPU tests run).
module SacHip

const LIBSAC = get(ENV, "SAC_HIP_LIB", joinpath(@__DIR__, "..", "sac_amd", "libsac_hip.so"))

const KIND_FREE = Int32(0)
const KIND_PEAK = Int32(1)
const HIST_REFERENCE = Int32(0)
const HIST_ACCUMULATE = Int32(1)
const KERNEL_CODE = Dict("finiteT" => Int32(0), "zeroT" => Int32(1), "bosonic" => Int32(2))
const COMM_ID_BYTES = 128

struct SacConfig                       # sac_config_t
    kind::Int32
    par::Int32
    symm::Int32
    n_tau::Int32
    n_omega::Int32
    n_chains::Int32
    n_grid::Int64
    wi_0::Int64
    wi_m::Int64
    seed::UInt64
    chain_offset::UInt32
    device::Int32
    hist_mode::Int32
    hist_copies::Int32
    sweeps_per_launch::Int32
    n_peak::Int32
    fix_edge::Int32
    bosonic::Int32
end

struct SacStats                        # sac_stats_t
    sampled_chi2::Float64
    chi2_min::Float64
    accept_rates::NTuple{6,Float64}
    windows::NTuple{4,Float64}
    proposals::UInt64
    evaluated::UInt64
    accepted::UInt64
    accepted_moves::UInt64
    sweeps::UInt64
    columns_gathered::UInt64
    last_stage_ms::Float64
    last_kernel_ms::Float64
    last_kernel_launches::Int64
    chi2_min_mean::Float64
    n_chains_total::Int64
end

struct SacPeakStats                    # sac_peak_stats_t
    sampled_chi2::Float64
    chi2_min::Float64
    accept_rates::NTuple{11,Float64}
    windows::NTuple{11,Float64}
    edge_res::NTuple{6,Float64}        # C: double[3][2], row-major: (mean+, mean-, sum+, sum-, sumsq+, sumsq-)
    bounds::NTuple{4,Float64}
    weights::NTuple{4,Float64}
    proposals::UInt64
    evaluated::UInt64
    accepted::UInt64
    accepted_moves::UInt64
    sweeps::UInt64
    columns_gathered::UInt64
    last_stage_ms::Float64
    last_kernel_ms::Float64
    last_kernel_launches::Int64
    chi2_min_mean::Float64
    n_chains_total::Int64
end

struct SacMoveRec                      # sac_move_rec_t
    chi2_new::Float64
    flags::Int32
    move::Int32
end

struct SacTableSpec                    # sac_table_spec_t
    kernel_type::Int32
    mirror::Int32
    negate::Int32
    reserved::Int32
    i_first::Int64
    offset::Float64
    d_omega::Float64
    beta::Float64
end

struct SacProposal                     # sac_proposal_t
    move::Int32
    k::NTuple{3,Int32}
    w_new::NTuple{3,Int64}
    u::Float64
end

struct SacPeakProposal                 # sac_peak_proposal_t
    n::Int32
    transfer::Int32
    k::NTuple{4,Int32}
    w_new::NTuple{4,Int64}
    A_new::NTuple{4,Float64}
    u::Float64
end

struct SacEdgeConfig                   # sac_edge_config_t
    n_tau::Int32
    n_e::Int32
    n_omega::Int32
    n_rl::Int32
    double_in::Int32
    fix_edge::Int32
    n_chains::Int32
    device::Int32
    wi_0::Int64
    n_grid::Int64
    A_c::Float64
    A_r::Float64
    d0::Float64
    w_floor::NTuple{2,Float64}
    w_m::Float64
    d_omega::Float64
    c::NTuple{2,Float64}
    seed::UInt64
    chain_offset::UInt32
    sweeps_per_launch::Int32
    hist_copies::Int32
    reserved::Int32
end

struct SacEdgeResults                  # sac_edge_results_t (the [3][2] arrays row-major: index 2 q + rl + 1, q = 0 last, 1 mean, 2 sigma)
    chi2_res::NTuple{3,Float64}
    edge_res::NTuple{6,Float64}
    n0_res::NTuple{6,Float64}
    wn0_res::NTuple{6,Float64}
    amp_accept_rates::NTuple{6,Float64}
    n0::NTuple{2,Float64}
    eps0::NTuple{2,Float64}
    dn0::NTuple{2,Float64}
    deps0::NTuple{2,Float64}
    chi2::Float64
    chi2_min::Float64
    sweeps::UInt64
    proposals::UInt64
    evaluated::UInt64
    accepted::UInt64
    lookups::UInt64
    violations::UInt64
    last_stage_ms::Float64
    chi2_min_mean::Float64
    n_chains_total::Int64
    chains_below_target::Int64
end

struct SacEdgeProposal                 # sac_edge_proposal_t
    side::Int32
    k::Int32
    n::Int32
    reserved::Int32
    w_offset::Int64
    u::Float64
end

const Handle = Ptr{Cvoid}

last_error() = unsafe_string(ccall((:sac_last_error, LIBSAC), Cstring, ()))
build_id() = unsafe_string(ccall((:sac_build_id, LIBSAC), Cstring, ()))

# every entry point returns 0 or a negative SAC_E_* code and never exits (replaces throw("Invalid Kernel type."),
# free:195, and exit(), edge:668)
function check(rc::Integer)
    rc == 0 || error("libsac_hip error $rc: " * last_error())
    return nothing
end

# ---- free / peak handle ------------------------------------------------------------------------------------------

function create(cfg::SacConfig)
    h = Ref{Handle}(C_NULL)
    check(ccall((:sac_create, LIBSAC), Cint, (Ref{SacConfig}, Ref{Handle}), Ref(cfg), h))
    return h[]
end
destroy(h::Handle) = check(ccall((:sac_destroy, LIBSAC), Cint, (Handle,), h))

# K[N_tau, N_wi] column-major (Julia's K_array) = tau contiguous: the layout the library wants
upload_problem(h, K::Matrix{Float64}, G::Vector{Float64}, sinv::Vector{Float64}) =
    check(ccall((:sac_upload_problem, LIBSAC), Cint, (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, K, G, sinv))
upload_problem_peak(h, Kp::Matrix{Float64}, Kn, G::Vector{Float64}, sinv::Vector{Float64}) =
    check(ccall((:sac_upload_problem_peak, LIBSAC), Cint, (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                h, Kp, Kn === nothing ? C_NULL : Kn, G, sinv))
upload_data(h, G::Vector{Float64}, sinv::Vector{Float64}) =
    check(ccall((:sac_upload_data, LIBSAC), Cint, (Handle, Ptr{Float64}, Ptr{Float64}), h, G, sinv))

# cov (eigenvectors in columns, column-major in Julia) crosses the ABI row-major: permutedims
function build_table(h, which::Integer, spec::SacTableSpec, tau::Vector{Float64}, cov::Matrix{Float64}; noalloc::Bool=false)
    U = permutedims(cov)
    if noalloc
        check(ccall((:sac_build_table_noalloc, LIBSAC), Cint, (Handle, Int32, Ref{SacTableSpec}, Ptr{Float64}, Ptr{Float64}),
                    h, which, Ref(spec), tau, U))
    else
        check(ccall((:sac_build_table, LIBSAC), Cint, (Handle, Int32, Ref{SacTableSpec}, Ptr{Float64}, Ptr{Float64}),
                    h, which, Ref(spec), tau, U))
    end
end
function get_table(h, which::Integer, n_tau::Integer, n_grid::Integer)
    out = Matrix{Float64}(undef, n_tau, n_grid)
    check(ccall((:sac_get_table, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}), h, which, out))
    return out
end

ptr_or_null(a) = a === nothing ? C_NULL : pointer(a)

# chain = -1: all chains; `nothing` keeps what the handle has
function set_state(h; chain::Integer=-1, wi=nothing, A=nothing, windows=nothing)
    GC.@preserve wi A windows check(ccall((:sac_set_state, LIBSAC), Cint, (Handle, Int32, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
                                          h, chain, ptr_or_null(wi), ptr_or_null(A), ptr_or_null(windows)))
end
function get_state(h, chain::Integer, n_state::Integer, n_windows::Integer)
    wi = Vector{Int64}(undef, n_state); A = Vector{Float64}(undef, n_state); w = Vector{Float64}(undef, n_windows)
    check(ccall((:sac_get_state, LIBSAC), Cint, (Handle, Int32, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}), h, chain, wi, A, w))
    return wi, A, w
end

init_gbar(h) = check(ccall((:sac_init_gbar, LIBSAC), Cint, (Handle,), h))
function get_gbar(h, chain::Integer, n_tau::Integer)
    g = Vector{Float64}(undef, n_tau); c = Ref{Float64}(0.0); m = Ref{Float64}(0.0)
    check(ccall((:sac_get_gbar, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}, Ref{Float64}, Ref{Float64}), h, chain, g, c, m))
    return g, c[], m[]
end
function get_gbar_mean(h, n_tau::Integer)
    g = Vector{Float64}(undef, n_tau)
    check(ccall((:sac_get_gbar_mean, LIBSAC), Cint, (Handle, Ptr{Float64}), h, g))
    return g
end
set_gbar(h, chain::Integer, g::Vector{Float64}, chi2::Float64, chi2_min::Float64) =
    check(ccall((:sac_set_gbar, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}, Float64, Float64), h, chain, g, chi2, chi2_min))

# the two stage functions of the reference (free:866, 903; peak:1063, 1100)
adjust_windows(h, steps::Integer, theta::Float64) =
    check(ccall((:sac_adjust_windows, LIBSAC), Cint, (Handle, Int64, Float64), h, steps, theta))
sample(h, steps::Integer, theta::Float64) =
    check(ccall((:sac_sample, LIBSAC), Cint, (Handle, Int64, Float64), h, steps, theta))
sample_bin(h, steps::Integer, theta::Float64) =
    check(ccall((:sac_sample_bin, LIBSAC), Cint, (Handle, Int64, Float64), h, steps, theta))
set_indiv_update(h, on::Bool) = check(ccall((:sac_set_indiv_update, LIBSAC), Cint, (Handle, Int32), h, on ? 1 : 0))
zero_accumulators(h) = check(ccall((:sac_zero_accumulators, LIBSAC), Cint, (Handle,), h))

function get_stats(h)
    st = Ref{SacStats}()
    check(ccall((:sac_get_stats, LIBSAC), Cint, (Handle, Ref{SacStats}), h, st))
    return st[]
end
function get_peak_stats(h)
    st = Ref{SacPeakStats}()
    check(ccall((:sac_get_peak_stats, LIBSAC), Cint, (Handle, Ref{SacPeakStats}), h, st))
    return st[]
end
function get_chain_stats(h, chain::Integer, n_rates::Integer)
    o = Vector{Float64}(undef, 6); ar = Vector{Float64}(undef, n_rates)
    check(ccall((:sac_get_chain_stats, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}, Ptr{Float64}), h, chain, o, ar))
    return o, ar
end
function get_chain_chi2(h, n_chains::Integer)
    a = Vector{Float64}(undef, n_chains); b = Vector{Float64}(undef, n_chains)
    check(ccall((:sac_get_chain_chi2, LIBSAC), Cint, (Handle, Ptr{Float64}, Ptr{Float64}), h, a, b))
    return a, b
end
copy_chain(h, src::Integer, dst::Integer) = check(ccall((:sac_copy_chain, LIBSAC), Cint, (Handle, Int32, Int32), h, src, dst))
function get_bounds(h, chain::Integer)
    b = Vector{Float64}(undef, 4)
    check(ccall((:sac_get_bounds, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}), h, chain, b))
    return b
end

function get_histogram(h, n_grid::Integer)
    out = Vector{Float64}(undef, n_grid)
    check(ccall((:sac_get_histogram, LIBSAC), Cint, (Handle, Ptr{Float64}), h, out))
    return out
end
function get_histogram_peak(h, which::Integer, n_grid::Integer)
    out = Vector{Float64}(undef, n_grid)
    check(ccall((:sac_get_histogram_peak, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}), h, which, out))
    return out
end
# the un-normalised sum over this handle's chains into DEVICE memory (e.g. an AMDGPU.jl array); returns steps * chains
function get_histogram_sum_device(h, dst_device::Ptr{Cvoid})
    d = Ref{Float64}(0.0)
    check(ccall((:sac_get_histogram_sum_device, LIBSAC), Cint, (Handle, Ptr{Cvoid}, Ref{Float64}), h, dst_device, d))
    return d[]
end

set_chains_per_wave(h, n::Integer) = check(ccall((:sac_set_chains_per_wave, LIBSAC), Cint, (Handle, Int32), h, n))
function get_chains_per_wave(h)
    o = Ref{Int32}(0); check(ccall((:sac_get_chains_per_wave, LIBSAC), Cint, (Handle, Ref{Int32}), h, o)); return o[]
end
set_proposals_per_step(h, n::Integer) = check(ccall((:sac_set_proposals_per_step, LIBSAC), Cint, (Handle, Int32), h, n))
function get_proposals_per_step(h)
    o = Ref{Int32}(0); check(ccall((:sac_get_proposals_per_step, LIBSAC), Cint, (Handle, Ref{Int32}), h, o)); return o[]
end

# calc_cv.jl:29-58 on the device (cov matrices cross row-major); gbar_mean = nothing: the mean over this handle's chains
function validation_chi2(h, cov_this::Matrix{Float64}, cov_other::Matrix{Float64}, G_other::Vector{Float64},
                         sinv_other::Vector{Float64}; gbar_mean=nothing, n_chains::Integer=0)
    out = Ref{Float64}(0.0)
    per = n_chains > 0 ? Vector{Float64}(undef, n_chains) : nothing
    Ut, Uo = permutedims(cov_this), permutedims(cov_other)
    GC.@preserve gbar_mean per check(ccall((:sac_validation_chi2, LIBSAC), Cint,
        (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}),
        h, Ut, Uo, G_other, sinv_other, ptr_or_null(gbar_mean), out, ptr_or_null(per)))
    return out[], per
end

# parity entries: an explicit proposal stream (SURVEY 8b-7)
function replay(h, chain::Integer, theta::Float64, props::Vector{SacProposal})
    n = length(props); c2 = Vector{Float64}(undef, n); acc = Vector{Int32}(undef, n)
    check(ccall((:sac_replay, LIBSAC), Cint, (Handle, Int32, Float64, Ptr{SacProposal}, Int64, Ptr{Float64}, Ptr{Int32}),
                h, chain, theta, props, n, c2, acc))
    return c2, acc
end
function replay_peak(h, chain::Integer, theta::Float64, props::Vector{SacPeakProposal})
    n = length(props); c2 = Vector{Float64}(undef, n); acc = Vector{Int32}(undef, n)
    check(ccall((:sac_replay_peak, LIBSAC), Cint, (Handle, Int32, Float64, Ptr{SacPeakProposal}, Int64, Ptr{Float64}, Ptr{Int32}),
                h, chain, theta, props, n, c2, acc))
    return c2, acc
end
set_move_log(h, chain::Integer, capacity::Integer) = check(ccall((:sac_set_move_log, LIBSAC), Cint, (Handle, Int32, Int64), h, chain, capacity))
function get_move_log(h, capacity::Integer)
    out = Vector{SacMoveRec}(undef, capacity); n = Ref{Int64}(0)
    check(ccall((:sac_get_move_log, LIBSAC), Cint, (Handle, Ptr{SacMoveRec}, Int64, Ref{Int64}), h, out, capacity, n))
    return out[1:min(n[], capacity)], n[]
end

# ---- multi-GPU merge behind the boundary -------------------------------------------------------------------------
# One Julia process per GPU (e.g. under mpiexec or Distributed): rank 0 draws the id, hands the 128 bytes to the others
# (MPI.Bcast!, a file, ...), every rank attaches; from then on adjust_windows / sample / run_bins end with ONE fused
# all-reduce(sum) + ONE all-reduce(min) on the device and get_stats / get_histogram / edge_get_results(-1) answer for
# all ranks' chains, so every rank takes the same schedule decisions.
function comm_unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    check(ccall((:sac_comm_unique_id, LIBSAC), Cint, (Ptr{UInt8},), id))
    return id
end
comm_init(h, id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(ccall((:sac_comm_init, LIBSAC), Cint, (Handle, Ptr{UInt8}, Int32, Int32), h, id, rank, world))
function comm_info(h)
    r = Ref{Int32}(0); w = Ref{Int32}(0); m = Ref{Int64}(0); a = Ref{Int64}(0)
    check(ccall((:sac_comm_info, LIBSAC), Cint, (Handle, Ref{Int32}, Ref{Int32}, Ref{Int64}, Ref{Int64}), h, r, w, m, a))
    return (rank=r[], world=w[], merges=m[], allreduce_calls=a[])
end

# ---- edge handle -------------------------------------------------------------------------------------------------

function edge_create(cfg::SacEdgeConfig)
    h = Ref{Handle}(C_NULL)
    check(ccall((:sac_edge_create, LIBSAC), Cint, (Ref{SacEdgeConfig}, Ref{Handle}), Ref(cfg), h))
    return h[]
end
edge_destroy(h) = check(ccall((:sac_edge_destroy, LIBSAC), Cint, (Handle,), h))

# K_array[N_tau, N_wi, R_L] etc.: one contiguous [N_tau, N_wi] slice per side
function edge_upload_problem(h, K::Array{Float64,3}, dK::Array{Float64,3}, d2K::Array{Float64,3}, G::Vector{Float64}, sinv::Vector{Float64})
    n = size(K, 1) * size(K, 2)
    sides(a) = Ptr{Float64}[r <= size(a, 3) ? pointer(a, 1 + (r - 1) * n) : Ptr{Float64}(C_NULL) for r in 1:2]
    pk, pd, p2 = sides(K), sides(dK), sides(d2K)
    GC.@preserve K dK d2K check(ccall((:sac_edge_upload_problem, LIBSAC), Cint,
        (Handle, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Float64}, Ptr{Float64}), h, pk, pd, p2, G, sinv))
end
edge_build_tables(h, side::Integer, spec::SacTableSpec, tau::Vector{Float64}, cov::Matrix{Float64}) =
    check(ccall((:sac_edge_build_tables, LIBSAC), Cint, (Handle, Int32, Ref{SacTableSpec}, Ptr{Float64}, Ptr{Float64}),
                h, side, Ref(spec), tau, permutedims(cov)))
edge_upload_data(h, G::Vector{Float64}, sinv::Vector{Float64}) =
    check(ccall((:sac_edge_upload_data, LIBSAC), Cint, (Handle, Ptr{Float64}, Ptr{Float64}), h, G, sinv))
function edge_get_tables(h, side::Integer, n_tau::Integer, n_grid::Integer)
    K = Matrix{Float64}(undef, n_tau, n_grid); dK = similar(K); d2K = similar(K)
    check(ccall((:sac_edge_get_tables, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, side, K, dK, d2K))
    return K, dK, d2K
end

# state arrays [N_w+1, 2] column-major, as the reference holds them; scal8 = vcat(n0, eps_0, Dn0, Deps_0)
function edge_set_state(h; chain::Integer=-1, w=nothing, A=nothing, dw=nothing, n_multi=nothing, scal8=nothing)
    GC.@preserve w A dw n_multi scal8 check(ccall((:sac_edge_set_state, LIBSAC), Cint,
        (Handle, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}),
        h, chain, ptr_or_null(w), ptr_or_null(A), ptr_or_null(dw), ptr_or_null(n_multi), ptr_or_null(scal8)))
end
function edge_get_state(h, chain::Integer, n_omega::Integer)
    n1 = n_omega + 1
    w = Matrix{Float64}(undef, n1, 2); A = similar(w); dw = similar(w); nm = Matrix{Int64}(undef, n1, 2); s = Vector{Float64}(undef, 8)
    check(ccall((:sac_edge_get_state, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}),
                h, chain, w, A, dw, nm, s))
    return w, A, dw, nm, s
end
edge_init_gbar(h) = check(ccall((:sac_edge_init_gbar, LIBSAC), Cint, (Handle,), h))
# run_bins(self, steps, bins, theta)  edge:1357
edge_run_bins(h, steps::Integer, bins::Integer, theta::Float64) =
    check(ccall((:sac_edge_run_bins, LIBSAC), Cint, (Handle, Int64, Int64, Float64), h, steps, bins, theta))
function edge_get_results(h, chain::Integer=-1)
    r = Ref{SacEdgeResults}()
    check(ccall((:sac_edge_get_results, LIBSAC), Cint, (Handle, Int32, Ref{SacEdgeResults}), h, chain, r))
    return r[]
end
function edge_get_accept_rates(h, chain::Integer, m::Integer, side::Integer, n_omega::Integer)
    a = Vector{Float64}(undef, n_omega + 1)
    check(ccall((:sac_edge_get_accept_rates, LIBSAC), Cint, (Handle, Int32, Int32, Int32, Ptr{Float64}), h, chain, m, side, a))
    return a
end
function edge_get_histogram(h, which::Integer, side::Integer, n_grid::Integer)
    o = Vector{Float64}(undef, n_grid)
    check(ccall((:sac_edge_get_histogram, LIBSAC), Cint, (Handle, Int32, Int32, Ptr{Float64}), h, which, side, o))
    return o
end
function edge_get_avg(h, chain::Integer, n_omega::Integer)
    w = Matrix{Float64}(undef, n_omega + 1, 2); A = similar(w)
    check(ccall((:sac_edge_get_avg, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}, Ptr{Float64}), h, chain, w, A))
    return w, A
end
function edge_get_move_means(h, n_omega::Integer)
    dw = Matrix{Float64}(undef, n_omega + 1, 2); nm = similar(dw); ext = Vector{Float64}(undef, 2)
    check(ccall((:sac_edge_get_move_means, LIBSAC), Cint, (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h, dw, nm, ext))
    return dw, nm, ext
end
# the many-chain form of anneal(chi2_target) (edge:1680, 1718): per-chain targets, first passages counted by the library
edge_set_targets_from_chi2_min(h, a::Float64) =
    check(ccall((:sac_edge_set_targets_from_chi2_min, LIBSAC), Cint, (Handle, Float64), h, a))
edge_set_chain_targets(h, targets) =
    check(ccall((:sac_edge_set_chain_targets, LIBSAC), Cint, (Handle, Ptr{Float64}), h, targets === nothing ? C_NULL : targets))
function edge_get_first_passages(h, n_chains::Integer)
    out = Vector{Int32}(undef, n_chains)
    check(ccall((:sac_edge_get_first_passages, LIBSAC), Cint, (Handle, Ptr{Int32}), h, out))
    return out
end
function edge_get_gbar(h, chain::Integer, n_tau::Integer)
    g = Vector{Float64}(undef, n_tau); c = Ref{Float64}(0.0); m = Ref{Float64}(0.0)
    check(ccall((:sac_edge_get_gbar, LIBSAC), Cint, (Handle, Int32, Ptr{Float64}, Ref{Float64}, Ref{Float64}), h, chain, g, c, m))
    return g, c[], m[]
end
function edge_get_gbar_mean(h, n_tau::Integer)
    g = Vector{Float64}(undef, n_tau)
    check(ccall((:sac_edge_get_gbar_mean, LIBSAC), Cint, (Handle, Ptr{Float64}), h, g))
    return g
end
function edge_validation_chi2(h, cov_this::Matrix{Float64}, cov_other::Matrix{Float64}, G_other::Vector{Float64},
                              sinv_other::Vector{Float64}; gbar_mean=nothing)
    out = Ref{Float64}(0.0)
    Ut, Uo = permutedims(cov_this), permutedims(cov_other)
    GC.@preserve gbar_mean check(ccall((:sac_edge_validation_chi2, LIBSAC), Cint,
        (Handle, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}),
        h, Ut, Uo, G_other, sinv_other, ptr_or_null(gbar_mean), out, C_NULL))
    return out[]
end
function edge_replay(h, chain::Integer, theta::Float64, props::Vector{SacEdgeProposal}, w_new::Vector{Float64})
    n = length(props); c2 = Vector{Float64}(undef, n); acc = Vector{Int32}(undef, n)
    check(ccall((:sac_edge_replay, LIBSAC), Cint,
                (Handle, Int32, Float64, Ptr{SacEdgeProposal}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Int32}),
                h, chain, theta, props, n, w_new, length(w_new), c2, acc))
    return c2, acc
end
edge_set_move_log(h, chain::Integer, capacity::Integer) =
    check(ccall((:sac_edge_set_move_log, LIBSAC), Cint, (Handle, Int32, Int64), h, chain, capacity))
function edge_get_move_log(h, capacity::Integer)
    out = Vector{SacMoveRec}(undef, capacity); n = Ref{Int64}(0)
    check(ccall((:sac_edge_get_move_log, LIBSAC), Cint, (Handle, Ptr{SacMoveRec}, Int64, Ref{Int64}), h, out, capacity, n))
    return out[1:min(n[], capacity)], n[]
end
edge_comm_init(h, id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(ccall((:sac_edge_comm_init, LIBSAC), Cint, (Handle, Ptr{UInt8}, Int32, Int32), h, id, rank, world))
function edge_comm_info(h)
    r = Ref{Int32}(0); w = Ref{Int32}(0); m = Ref{Int64}(0); a = Ref{Int64}(0)
    check(ccall((:sac_edge_comm_info, LIBSAC), Cint, (Handle, Ref{Int32}, Ref{Int32}, Ref{Int64}, Ref{Int64}), h, r, w, m, a))
    return (rank=r[], world=w[], merges=m[], allreduce_calls=a[])
end

# ---- shared host helpers of the drivers --------------------------------------------------------------------------

# t.in (written by make_tin.jl:143-153; read as free:99-131): beta, N_tau, norm; tau, G, sigma_tilde; eigenvector blocks
struct TimeData
    beta::Float64
    n_tau::Int
    norm::Float64
    tau::Vector{Float64}
    G_raw::Vector{Float64}         # G as the file has it (tau basis)
    G::Vector{Float64}             # cov' * G (eigenbasis)
    sigma_inv::Vector{Float64}
    cov::Matrix{Float64}           # eigenvectors in columns
end

function read_tin(path::AbstractString)
    lines = [split(l) for l in eachline(path) if !isempty(strip(l))]
    beta = parse(Float64, lines[1][1]); nt = Int(parse(Float64, lines[1][2])); norm = parse(Float64, lines[1][4])
    tau = zeros(nt); G = zeros(nt); sinv = zeros(nt)
    for i in 1:nt
        f = lines[1 + i]
        tau[i] = parse(Float64, f[1]); G[i] = parse(Float64, f[2]); sinv[i] = 1 / parse(Float64, f[4])
    end
    cov = zeros(nt, nt)
    row = 1 + nt
    for j in 1:nt
        row += 1
        @assert Int(parse(Float64, lines[row][1])) == j      # block index line (free:115)
        for i in 1:nt
            row += 1
            cov[i, j] = parse(Float64, lines[row][1])
        end
    end
    return TimeData(beta, nt, norm, tau, G, cov' * G, sinv, cov)
end

# the kernels (identical copies free:220-239, peak:289-307, edge:246-265)
function kernel_value(kt::AbstractString, w::Float64, tau::Float64, beta::Float64)
    if kt == "finiteT"
        a, b = -w * tau, -beta * w
        m = max(a, b, 0.0)
        return exp(a - m) / (exp(-m) + exp(b - m))
    elseif kt == "zeroT"
        return exp(-w * tau)
    elseif kt == "bosonic"
        return (exp(-w * tau) + exp(-w * (beta - tau))) / (1 + exp(-beta * w))
    end
    error("Invalid Kernel type.")
end

# Float64 the way the reference's writers print it (print / writedlm: shortest round-trip digits)
fmt(x) = string(Float64(x))

logline(folder::AbstractString, msg::AbstractString; mode::AbstractString="a") =
    open(joinpath(folder, "log.txt"), mode) do f
        println(f, Dates_now(), " - ", msg)
    end
import Dates
Dates_now() = Dates.now()

end # module
