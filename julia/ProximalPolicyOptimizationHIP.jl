# ProximalPolicyOptimizationHIP.jl -- reference-side binding of libppo_hip.so (include/ppo_hip.h).
#
# UNTESTED AT RUN TIME: the build container has no `julia` binary (SURVEY.md 8(c)), so this file has never been
# executed.  What IS checked, on every test run, is its C boundary: tests/test_abi.py parses every `ccall` below and
# compares symbol, arity and argument types with include/ppo_hip.h.  It is the binding a maintainer of
# ProximalPolicyOptimization.jl would add: methods of the package's own generic functions
# (src/ProximalPolicyOptimization.jl:16-30) for GPU-resident types, each a thin `ccall`; the Python mirror
# (proximalpolicyoptimization.jl_amd/__init__.py) is the executed twin of this file.
module ProximalPolicyOptimizationHIP

using ProximalPolicyOptimization
import Flux
using Printf
const PPO = ProximalPolicyOptimization
const LIB = get(ENV, "PPO_HIP_LIB", "libppo_hip.so")

function check(status::Int32)
    status == 0 && return
    buf = Vector{UInt8}(undef, 1024)
    ccall((:ppo_last_error, LIB), Int32, (Ptr{UInt8}, Int64), buf, 1024)
    msg = unsafe_string(pointer(buf))
    startswith(msg, "AssertionError") ? throw(AssertionError(msg)) : error(msg)
end

# Seed of the device-side minibatch permutations (the stand-in for randperm, src/train.jl:93).  The reference never
# seeds its RNG; here runs are reproducible by default and `set_seed!` changes the stream.  The optimiser handle
# counts the epochs it has trained, so successive ppo_train! calls draw different permutations from one seed.
const SEED = Ref{UInt64}(0)
set_seed!(s::Integer) = (SEED[] = UInt64(s))

# ---------------------------------------------------------------- handles
mutable struct HipVecEnv
    h::Ptr{Cvoid}; N::Int; Q::Int; H::Int; F::Int; A::Int; max_actions::Int
    function HipVecEnv(num_envs; Q = 8, max_actions = 128, no_action_reward = -4f0, seed = 1234, global_offset = 0)
        r = Ref{Ptr{Cvoid}}()
        check(ccall((:ppo_env_create, LIB), Int32, (Int32, Int64, Int64, Int32, Int32, Float32, UInt64, Ref{Ptr{Cvoid}}),
                    0, num_envs, global_offset, Q, max_actions, no_action_reward, seed, r))
        n, hh, ff, aa = Ref{Int64}(), Ref{Int32}(), Ref{Int32}(), Ref{Int32}()          # shapes as the engine reports them
        check(ccall((:ppo_env_dims, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int32}, Ref{Int32}, Ref{Int32}), r[], n, hh, ff, aa))
        e = new(r[], n[], Q, hh[], ff[], aa[], max_actions)
        finalizer(x -> ccall((:ppo_env_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), e)
    end
end

mutable struct HipPolicy
    h::Ptr{Cvoid}; nparams::Int
    function HipPolicy(in_channels, hidden_channels, num_hidden_layers, num_output)   # test/policy.jl:9
        r = Ref{Ptr{Cvoid}}()
        check(ccall((:ppo_policy_create, LIB), Int32, (Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
                    in_channels, hidden_channels, num_hidden_layers, num_output, r))
        n = Ref{Int64}()
        check(ccall((:ppo_policy_num_params, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}), r[], n))
        p = new(r[], n[])
        finalizer(x -> ccall((:ppo_policy_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), p)
    end
end

# arithmetic of the MLP's Dense products: :f32 (Flux's Float32, default) or :bf16 (bf16 MFMA, fp32 accumulation)
set_dtype!(p::HipPolicy, dtype::Symbol) =
    check(ccall((:ppo_policy_set_dtype, LIB), Int32, (Ptr{Cvoid}, Int32), p.h, dtype === :bf16 ? 1 : 0))

# arithmetic of the fp32 TRAINING pass of Policy(72, h, 2, 4): true (default) = its Dense products as split-fp32 products on the
# bf16 matrix pipe (three exact bfloat16 pieces per operand, six piece products, fp32 accumulation: the distance to a Float64
# gradient is that of the fp32 kernels), false = fp32 MFMA.  Rollouts are not affected (their actions are pinned bit for bit).
set_training_split_bf16!(on::Bool) = check(ccall((:ppo_set_bwd_split_bf16, LIB), Int32, (Int32,), on ? 1 : 0))

# Flux.params(policy) round trip: flat vector in Flux order (W1,b1, the num_hidden_layers-1 hidden (W,b) pairs, W_out,b_out),
# W [out,in] column-major; any num_hidden_layers in 1..4 (test/policy.jl:9-19)
set_params!(p::HipPolicy, flat::Vector{Float32}) =
    check(ccall((:ppo_policy_set_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}), p.h, flat))
function get_params(p::HipPolicy)
    flat = Vector{Float32}(undef, p.nparams)
    check(ccall((:ppo_policy_get_params, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}), p.h, flat)); flat
end
load_flux_chain!(p::HipPolicy, chain) = set_params!(p, vcat([vec(Float32.(x)) for x in Flux.params(chain)]...))

struct StateData                      # test/quad_game_utilities.jl:17-20 (int8 rows + active-quad bits)
    vertex_score::Array{Int8}; action_mask
end

mutable struct HipRollouts
    h::Ptr{Cvoid}; env::Union{Nothing,HipVecEnv}
    HipRollouts() = new(C_NULL, nothing)              # PPO.BufferRollouts()
end
function ensure!(r::HipRollouts, env::HipVecEnv, T)
    if r.h == C_NULL
        ref = Ref{Ptr{Cvoid}}()
        check(ccall((:ppo_rollouts_create, LIB), Int32, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}), env.h, T, ref))
        r.h = ref[]; r.env = env
        finalizer(x -> ccall((:ppo_rollouts_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), r)
    end
    r.h
end

# BufferRollouts for a user env whose state(env) rows are not the built-in env's (host-supplied columns: N columns of
# [F, H] Int8 rows, any F the policy was created for -- e.g. the 216-feature level-4 template); filled with set_columns!
function HipRollouts(N::Integer, H::Integer, F::Integer, T::Integer)
    r = HipRollouts()
    ref = Ref{Ptr{Cvoid}}()
    check(ccall((:ppo_rollouts_create_shape, LIB), Int32, (Int64, Int32, Int32, Int64, Ref{Ptr{Cvoid}}), N, H, F, T, ref))
    r.h = ref[]
    finalizer(x -> ccall((:ppo_rollouts_destroy, LIB), Int32, (Ptr{Cvoid},), x.h), r)
    r
end
function set_columns!(r::HipRollouts, states::Array{Int8}, active::Array{UInt32}, actions1, p_sel::Array{Float32}, returns::Array{Float32})
    T = size(active, 2)                                # columns are [N, T] column-major == [T][N]
    check(ccall((:ppo_rollouts_set, LIB), Int32,
                (Ptr{Cvoid}, Int64, Ptr{Int8}, Ptr{UInt32}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{UInt8}),
                r.h, T, states, active, Int32.(actions1 .- 1), p_sel, returns, C_NULL))
end

struct HipAdam; h::Ptr{Cvoid}; eta::Float64; end      # member of a Flux.Optimiser-like iterable
function HipAdam(p::HipPolicy, eta = 1e-3, beta = (0.9, 0.999), eps = 1e-8)
    r = Ref{Ptr{Cvoid}}()
    check(ccall((:ppo_adam_create, LIB), Int32, (Ptr{Cvoid}, Float64, Float64, Float64, Float64, Ref{Ptr{Cvoid}}),
                p.h, eta, beta[1], beta[2], eps, r))
    HipAdam(r[], eta)
end

# ---------------------------------------------------------------- env plugin methods  (:16-20)
function PPO.state(env::HipVecEnv)
    obs = Array{Int8}(undef, env.F, env.H, env.N); act = Vector{UInt32}(undef, env.N)   # column-major [F,H,N]
    check(ccall((:ppo_env_get_state, LIB), Int32, (Ptr{Cvoid}, Ptr{Int8}, Ptr{UInt32}), env.h, obs, act))
    StateData(obs, act)
end
function PPO.reward(env::HipVecEnv)
    r = Vector{Float32}(undef, env.N)
    check(ccall((:ppo_env_get_reward, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}), env.h, r)); r
end
function PPO.is_terminal(env::HipVecEnv)
    d = Vector{UInt8}(undef, env.N)
    check(ccall((:ppo_env_get_terminal, LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}), env.h, d)); d .!= 0
end
PPO.reset!(env::HipVecEnv) = check(ccall((:ppo_env_reset, LIB), Int32, (Ptr{Cvoid},), env.h))
PPO.step!(env::HipVecEnv, actions::AbstractVector{<:Integer}) =                      # 1-based -> 0-based
    check(ccall((:ppo_env_step, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}), env.h, Int32.(actions .- 1)))

# ---------------------------------------------------------------- policy / batching plugin methods  (:23-29)
function PPO.batch_action_probabilities(p::HipPolicy, s::StateData)      # -> [A,B]
    H, B = size(s.vertex_score, 2), size(s.vertex_score, 3)
    probs = Matrix{Float32}(undef, 4H, B)
    check(ccall((:ppo_policy_forward, LIB), Int32, (Ptr{Cvoid}, Ptr{Int8}, Ptr{UInt32}, Int64, Int32, Ptr{Float32}),
                p.h, s.vertex_score, UInt32.(s.action_mask), B, H, probs))
    probs
end
PPO.action_probabilities(p::HipPolicy, s::StateData) = vec(PPO.batch_action_probabilities(p, s))
PPO.number_of_actions_per_state(s::StateData) = 4 * size(s.vertex_score, 2)
PPO.batch_advantage(s::StateData, returns) = returns                    # raw returns (no method exists upstream)

# ---------------------------------------------------------------- path entry points
# compute_returns(rewards, terminal, discount) (src/collect_rollouts.jl:26-42) on the device.  Its own name on purpose:
# the reference's method is untyped, so a `PPO.compute_returns(::Vector{Float32}, ::Vector{Bool}, ::Any)` method here
# would silently re-route every caller's CPU buffers; a user who wants exactly that adds the one-line method
#     PPO.compute_returns(r::Vector{Float32}, t::Vector{Bool}, d) = ProximalPolicyOptimizationHIP.compute_returns_hip(r, t, d)
function compute_returns_hip(rewards::Vector{Float32}, terminal::AbstractVector{Bool}, discount)
    out = similar(rewards)
    check(ccall((:ppo_compute_returns, LIB), Int32, (Ptr{Float32}, Ptr{UInt8}, Int64, Float64, Int32, Ptr{Float32}),
                rewards, UInt8.(terminal), length(rewards), Float64(discount), discount isa Float32, out)); out
end

# collect_rollouts!(rollouts, env, policy, num_episodes, discount) (src/rollout_buffer.jl:66-79): exactly num_episodes
# whole episodes, played in parallel on the resident envs (episode e on env e mod N)
function PPO.collect_rollouts!(r::HipRollouts, env::HipVecEnv, p::HipPolicy, num_episodes, discount)
    h = ensure!(r, env, cld(num_episodes, env.N) * env.max_actions)
    check(ccall((:ppo_collect_rollouts_episodes, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Float64, Int32),
                h, env.h, p.h, num_episodes, Float64(discount), discount isa Float32))
end
function Base.length(r::HipRollouts)
    n = Ref{Int64}(); check(ccall((:ppo_rollouts_len, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}), r.h, n)); n[]
end
PPO.construct_dataset(r::HipRollouts) = r              # dataset == non-owning view of the same handle

# average_returns(policy, env, num_trajectories) (src/evaluate.jl:18-25) -> (mean, std)
function PPO.average_returns(p::HipPolicy, env::HipVecEnv, num_trajectories)
    scratch = HipRollouts()
    h = ensure!(scratch, env, cld(num_trajectories, env.N) * env.max_actions)
    m, s = Ref{Float64}(), Ref{Float64}()
    check(ccall((:ppo_average_returns, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Float64}),
                p.h, env.h, h, num_trajectories, m, s))
    m[], s[]
end

# evaluator variants of test/quad_game_utilities.jl:280-307,369-387 (argument order as there: env first)
function average_best_returns(env::HipVecEnv, p::HipPolicy, num_trajectories)
    scratch = HipRollouts()
    h = ensure!(scratch, env, 1)
    m, s = Ref{Float64}(), Ref{Float64}()
    check(ccall((:ppo_average_best_returns, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Float64}),
                p.h, env.h, h, num_trajectories, m, s))
    m[], s[]
end
function average_normalized_returns(env::HipVecEnv, p::HipPolicy, num_trajectories)
    scratch = HipRollouts()
    h = ensure!(scratch, env, 1)
    m, s = Ref{Float64}(), Ref{Float64}()
    check(ccall((:ppo_average_normalized_returns, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Float64}),
                p.h, env.h, h, num_trajectories, m, s))
    m[], s[]
end
# per-trajectory values: kind 1 = single_trajectory_return, 2 = best_single_trajectory_return, 3 = single_trajectory_normalized_return
function evaluate_trajectories(env::HipVecEnv, p::HipPolicy, num_trajectories, kind)
    scratch = HipRollouts()
    h = ensure!(scratch, env, 1)
    v = zeros(Float64, num_trajectories)
    check(ccall((:ppo_evaluate_trajectories, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Ptr{Float64}),
                p.h, env.h, h, num_trajectories, Int32(kind), v))
    v
end

# batch_advantage plugin mode handed to the engine: 0 = returns (PPO.batch_advantage above), 1 = normalised returns,
# 2 / 3 = GAE(gamma, lambda) / normalised GAE over values supplied through compute_gae!
const ADV_MODE = Ref{Int32}(0)
function compute_gae!(r::HipRollouts, values::Matrix{Float32}, gamma, lambda)           # values: [N, T+1] column-major == [T+1][N]
    check(ccall((:ppo_rollouts_compute_gae, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}, Float64, Float64, Ptr{Float32}, Ptr{Float32}),
                r.h, values, Float64(gamma), Float64(lambda), C_NULL, C_NULL))
end

# ppo_train!(policy, optimizer, dataset, epsilon, batch_size, num_epochs, entropy_weight) (src/train.jl:130-153).
# rank / world / hook: data-parallel runs (one process per GPU; INTEGRATION.md section 5); single process: 0 / 1 / C_NULL
function PPO.ppo_train!(p::HipPolicy, optimizer, r::HipRollouts, epsilon, batch_size, num_epochs, entropy_weight;
                        rank = 0, world = 1, hook = C_NULL)
    adam = first(optimizer)::HipAdam
    ph, eh, lh = zeros(num_epochs), zeros(num_epochs), zeros(num_epochs)
    check(ccall((:ppo_train, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, Int32, Float64, Int32, Ptr{Int64}, UInt64, Int32, Int32,
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                p.h, adam.h, r.h, epsilon, batch_size, num_epochs, entropy_weight, ADV_MODE[], C_NULL, SEED[], rank, world,
                hook, C_NULL, ph, eh, lh))
    for e in 1:num_epochs
        @printf "EPOCH : %d \t PPO LOSS : %1.4f\t ENTROPY LOSS : %1.4f \t LR : %1.1e\n" e ph[e] eh[e] lh[e]
    end
    ph, eh, lh
end
# ppo_iterate!(policy, env, optimizer, ...) (src/train.jl:210-249) then works unchanged once
# `BufferRollouts()` on its line 230 is replaced by `HipRollouts()` (or dispatched on the env type).

end # module
