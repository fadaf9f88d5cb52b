# bench/julia_reference.jl -- the REFERENCE's own Julia path, timed on the host (SURVEY.md 8(d), BASELINE.md 3.1).
#
#   julia bench/julia_reference.jl <path-to-ProximalPolicyOptimization.jl checkout> [num_envs] [T] [epochs] [batch]
#
# Drives the reference's `collect_rollouts!` + `ppo_train!` (src/rollout_buffer.jl:66-79, src/train.jl:130-153) -- its code,
# unmodified, included from the checkout -- against a Julia implementation of the SAME synthetic rand-poly-shaped env and
# the same Policy(72, 256, 2, 4) the GPU run uses (DESIGN.md "Synthetic env"; plugin methods as in
# test/quad_game_utilities.jl:35-79, test/policy.jl:9-31).  Prints ONE JSON line that bench.py embeds as
# `cpu_baseline_julia`.  bench.py runs this only when `julia` is on PATH and PPO_JULIA_REFERENCE names a checkout;
# neither holds in the build image or on the GPU box (no julia, no network), so this file is checked statically only
# (tests/test_host_logic.py) and the line then simply has no `cpu_baseline_julia` key.
#
# Same inputs as the GPU run (BASELINE.md 3.4): Q = 8 quad slots -> H = 32 half-edges -> A = 128 actions, F = 72 integer
# features per half-edge, 6 / 8 quads active at reset, max_actions = T, no_action_reward = -4; Glorot-uniform fp32
# weights; gamma = 1, epsilon = 0.05, entropy weight 0.01, Adam 1e-4.  The reference is single-task: 1 core.

const REF = length(ARGS) >= 1 ? ARGS[1] : get(ENV, "PPO_JULIA_REFERENCE", "")
isdir(REF) || error("usage: julia bench/julia_reference.jl <reference checkout> [num_envs] [T] [epochs] [batch]")
include(joinpath(REF, "src", "ProximalPolicyOptimization.jl"))
const PPO = ProximalPolicyOptimization
using Flux
using Random

const NUM_ENVS = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 64
const T_STEPS = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 128
const EPOCHS = length(ARGS) >= 4 ? parse(Int, ARGS[4]) : 4
const BATCH = length(ARGS) >= 5 ? parse(Int, ARGS[5]) : 4096

# ---------------------------------------------------------------------------------------------- Philox4x32-10
function philox4x32_10(c0::UInt32, c1::UInt32, c2::UInt32, c3::UInt32, k0::UInt32, k1::UInt32)
    for _ in 1:10
        p0 = UInt64(0xD2511F53) * UInt64(c0)
        p1 = UInt64(0xCD9E8D57) * UInt64(c2)
        n0 = (UInt32(p1 >> 32) ⊻ c1) ⊻ k0
        n1 = UInt32(p1 & 0xFFFFFFFF)
        n2 = (UInt32(p0 >> 32) ⊻ c3) ⊻ k1
        n3 = UInt32(p0 & 0xFFFFFFFF)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 += 0x9E3779B9
        k1 += 0xBB67AE85
    end
    return (c0, c1, c2, c3)
end

# ---------------------------------------------------------------------------------------------- the synthetic env
# one env instance, 0-based vertex / quad arithmetic kept in the comments' terms, arrays 1-based
mutable struct SynthEnv
    Q::Int
    max_actions::Int
    no_action_reward::Float32
    id::UInt32                    # global env id (Philox counter word)
    seed::UInt64
    score::Vector{Int8}           # [4Q]
    degree::Vector{Int8}          # [4Q]
    active::UInt32                # bit q = quad q active
    steps::Int
    reward::Float32
    done::Bool
    episode::UInt32
end

SynthEnv(id; Q = 8, max_actions = 128, no_action_reward = -4.0f0, seed = UInt64(1234)) =
    SynthEnv(Q, max_actions, no_action_reward, UInt32(id), seed, zeros(Int8, 4Q), zeros(Int8, 4Q), 0x00000000, 0, 0.0f0, false,
             0x00000000)

# template vertex of half-edge h, template row t (0-based); -1 = missing
function env_template(Q, h, t)
    V = 4Q
    q, ed = h >> 2, h & 3
    t < 4 && return 4q + ((ed + t) & 3)
    c = (h * 5 + t * 7 + 3) % (V + 6)
    return c >= V ? -1 : c
end

deg_ok(d) = 2 <= d <= 7
quad_on(env, q) = (env.active >> q) & 0x1 == 0x1
function total_abs(env)
    s = 0
    for q in 0:env.Q-1
        quad_on(env, q) || continue
        for i in 0:3
            s += abs(Int(env.score[4q+i+1]))
        end
    end
    return s
end
function total_sum(env)
    s = 0
    for q in 0:env.Q-1
        quad_on(env, q) || continue
        for i in 0:3
            s += Int(env.score[4q+i+1])
        end
    end
    return s
end

function PPO.reset!(env::SynthEnv)
    Q = env.Q
    nact = (3Q) ÷ 4
    k0, k1 = UInt32(env.seed & 0xFFFFFFFF), UInt32(env.seed >> 32)
    for q in 0:Q-1
        w = philox4x32_10(env.id, env.episode, 0x00000001, UInt32(q), k0, k1)
        for i in 0:3
            v = 4q + i + 1
            if q < nact
                s = Int(w[i+1] % 0x5) - 2
                desired = 3 + Int((w[i+1] >> 8) & 0x1)
                env.score[v] = Int8(s)
                env.degree[v] = Int8(desired - s)
            else
                env.score[v] = 0
                env.degree[v] = 0
            end
        end
    end
    env.active = nact >= 32 ? 0xFFFFFFFF : (UInt32(1) << nact) - UInt32(1)
    env.steps = 0
    env.reward = 0.0f0
    env.done = false
    env.episode += 0x1
    return
end

# step!(env, a): a is the reference's 1-based action index (quad, edge, type) = test/quad_game_utilities.jl:95-105
function PPO.step!(env::SynthEnv, a1)
    Q = env.Q
    A = 16Q
    @assert !env.done
    @assert 1 <= a1 <= A
    a = a1 - 1
    q, ed, typ = a ÷ 16, (a % 16) ÷ 4, a % 4
    sc, dg = env.score, env.degree
    old_total = total_abs(env)
    valid = false
    @assert quad_on(env, q)
    v0, v1, v2, v3 = 4q + ed, 4q + ((ed + 1) & 3), 4q + ((ed + 2) & 3), 4q + ((ed + 3) & 3)
    nq = (q + 1 + ed) % Q
    w0, w1 = 4nq + ed, 4nq + ((ed + 1) & 3)
    nq_ok = nq != q && quad_on(env, nq)
    D(v) = Int(dg[v+1])
    function bump!(v, dd)            # degree += dd, score -= dd
        dg[v+1] += Int8(dd)
        sc[v+1] -= Int8(dd)
    end
    if typ == 0 || typ == 1
        p, rr = typ == 0 ? (v3, w0) : (v2, w1)
        if nq_ok && deg_ok(D(v0) - 1) && deg_ok(D(v1) - 1) && deg_ok(D(p) + 1) && deg_ok(D(rr) + 1)
            bump!(v0, -1); bump!(v1, -1); bump!(p, 1); bump!(rr, 1)
            valid = true
        end
    elseif typ == 2
        f = -1
        for s in 0:Q-1
            if !quad_on(env, s)
                f = s
                break
            end
        end
        if f >= 0 && deg_ok(D(v0) + 1) && deg_ok(D(v2) + 1)
            bump!(v0, 1); bump!(v2, 1)
            for i in 0:3
                sc[4f+i+1] = 0
                dg[4f+i+1] = 4
            end
            env.active |= UInt32(1) << f
            valid = true
        end
    else
        if nq_ok && count_ones(env.active) > Q ÷ 2 && deg_ok(D(w0) - 1) && deg_ok(D(w1) - 1)
            bump!(w0, -1); bump!(w1, -1)
            for i in 0:3
                sc[4q+i+1] = 0
                dg[4q+i+1] = 0
            end
            env.active &= ~(UInt32(1) << q)
            valid = true
        end
    end
    new_total = total_abs(env)
    env.reward = valid ? Float32(old_total - new_total) : env.no_action_reward
    env.steps += 1
    env.done = (new_total == abs(total_sum(env))) || (env.steps >= env.max_actions)
    return
end

PPO.reward(env::SynthEnv) = env.reward
PPO.is_terminal(env::SynthEnv) = env.done

struct StateData
    vertex_score::Any             # [F, H] Int matrix (test/quad_game_utilities.jl:46-59)
    action_mask::Any              # [A] 0 / -Inf32 (:39-44)
end

function PPO.state(env::SynthEnv)
    Q = env.Q
    H = 4Q
    m = zeros(Int, 72, H)
    for h in 0:H-1
        own = quad_on(env, h >> 2)
        for t in 0:35
            v = env_template(Q, h, t)
            ok = own && v >= 0 && quad_on(env, v >> 2)
            m[t+1, h+1] = ok ? Int(env.score[v+1]) : 0
            m[36+t+1, h+1] = ok ? Int(env.degree[v+1]) : 0
        end
    end
    mask = zeros(Float32, 16Q)
    for q in 0:Q-1
        quad_on(env, q) || (mask[16q+1:16q+16] .= -Inf32)
    end
    return StateData(m, mask)
end

# ---------------------------------------------------------------------------------------------- policy plugin
struct Policy                     # test/policy.jl:9-19
    model
end
Flux.@functor Policy
function Policy(in_channels, hidden, num_hidden_layers, num_output)
    layers = Any[Dense(in_channels, hidden, leakyrelu)]
    for _ in 1:num_hidden_layers-1
        push!(layers, Dense(hidden, hidden, leakyrelu))
    end
    push!(layers, Dense(hidden, num_output))
    Policy(Chain(layers...))
end
(p::Policy)(x) = p.model(x)

function PPO.action_probabilities(policy::Policy, s::StateData)           # test/quad_game_utilities.jl:65-71
    logits = vec(policy(Float32.(s.vertex_score))) + s.action_mask
    return softmax(logits)
end
function PPO.batch_action_probabilities(policy::Policy, s::StateData)     # :73-79
    nf, nq, nb = size(s.vertex_score)
    logits = reshape(policy(s.vertex_score), :, nb) + s.action_mask
    return softmax(logits, dims = 1)
end
function PPO.batch_state(states)                                          # pad-free: every state has the same shape here
    vs = Float32.(cat([s.vertex_score for s in states]..., dims = 3))
    am = cat([s.action_mask for s in states]..., dims = 2)
    return StateData(vs, am)
end
PPO.number_of_actions_per_state(s::StateData) = size(s.action_mask, 1)
PPO.batch_advantage(state, returns) = returns                              # what the reference's scripts do

# N resident envs behind the reference's single-env interface: collect_rollouts!(., env, ., num_episodes, .) plays
# `num_episodes` episodes one after the other (src/rollout_buffer.jl:73-77); episode e runs on env e mod N
mutable struct EnvBank
    envs::Vector{SynthEnv}
    cur::Int
end
PPO.reset!(b::EnvBank) = (b.cur = b.cur % length(b.envs) + 1; PPO.reset!(b.envs[b.cur]))
PPO.state(b::EnvBank) = PPO.state(b.envs[b.cur])
PPO.step!(b::EnvBank, a) = PPO.step!(b.envs[b.cur], a)
PPO.reward(b::EnvBank) = PPO.reward(b.envs[b.cur])
PPO.is_terminal(b::EnvBank) = PPO.is_terminal(b.envs[b.cur])

function main()
    Random.seed!(0)
    policy = Policy(72, 256, 2, 4)
    optimizer = Flux.Optimiser(Flux.Adam(1f-4))
    bank = EnvBank([SynthEnv(i - 1; max_actions = T_STEPS) for i in 1:NUM_ENVS], 0)
    function iteration()
        rollouts = PPO.BufferRollouts()
        PPO.collect_rollouts!(rollouts, bank, policy, NUM_ENVS, 1.0)       # one episode per resident env
        dataset = PPO.construct_dataset(rollouts)
        n = length(dataset)
        redirect_stdout(devnull) do
            PPO.ppo_train!(policy, optimizer, dataset, 0.05f0, min(BATCH, n), EPOCHS, 0.01f0)
        end
        return n
    end
    iteration()                                                            # compile
    t0 = time()
    n = iteration()
    dt = time() - t0
    println("{\"value\": $(n / dt), \"unit\": \"env-steps/s\", \"cores\": 1, \"kind\": \"reference\", \"sample\": ",
            "\"ProximalPolicyOptimization.jl collect_rollouts! + ppo_train! (Julia $(VERSION), Flux, 1 task): $(NUM_ENVS) ",
            "episodes of <= $(T_STEPS) steps = $(n) env-steps, $(EPOCHS) epochs, minibatch $(min(BATCH, n)), ",
            "2x256 MLP, $(round(dt, digits = 1)) s\"}")
end

main()
