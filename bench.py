#!/usr/bin/env python3
"""bench.py -- end-to-end PPO throughput of the MI355X engine (BASELINE.json metric).

One "step" = one PPO iteration over synthetic input: rollout of N parallel envs for T steps (batched env
step + policy forward + masked softmax + categorical sample + record), the discounted-return scan, and E
epochs of minibatch updates (forward + loss + backward + Adam) over all N*T transitions.
Workload (BASELINE configs[1], survey-chosen sizes SURVEY.md 8(d)): 4096 synthetic rand-poly-shaped envs
(Q=8 -> H=32 half-edges, A=128 actions, F=72 int8 features), policy Dense(72,256)->Dense(256,256)->Dense(256,4)
fp32, T=128, E=4 epochs, minibatch 4096 per GPU, gamma=1.0, eps=0.05, entropy weight 0.01, Adam 1e-4.
Multi-GPU (one process per GPU): `python bench.py --gpus N` starts its N rank processes ITSELF (before anything touches
the GPU) unless a launcher (torchrun) already did.  Headline = weak scaling (BASELINE config 3): 4096 envs and a
4096-sample minibatch PER GPU, one RCCL all-reduce of the flat gradient per optimiser step (issued by the library
itself, ppo_rccl_*), no other exchange.  For N > 1 the same JSON line also carries a "strong" object: 4096 envs and a
4096-sample global minibatch split over the N GPUs, timed the same way.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_ENVS, T_STEPS, HID, F, EPOCHS, MINIBATCH = 4096, 128, 256, 72, 4, 4096
QUADS = 8                             # quad slots per env: H = 4*QUADS half-edge rows, A = 16*QUADS actions
GAMMA, EPS, ENT_W, LR = 1.0, 0.05, 0.01, 1e-4
PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense" (--dtype bf16 runs only)
PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"   # tools/pmc_traffic2.py: measured HBM bytes per launch, keyed by launch shape
LAYERS = 2                            # hidden layers of the policy (test/policy.jl:9-19); 2 = the headline 2x256 MLP


def flops_per_state(kind):
    """Algorithmic flops of one 32-row state tile on the matrix pipe (DESIGN.md 'Kernels')."""
    rows = 4 * QUADS                     # half-edge rows of one state (32 for the headline workload)
    fwd = 2 * rows * (F * HID + (LAYERS - 1) * HID * HID + HID * 4)
    if kind == "fwd":
        return fwd
    if LAYERS != 2:                      # deep policies: dH and dW of every hidden->hidden layer, dW1, dW3 / dH_top
        return 2 * rows * (2 * (LAYERS - 1) * HID * HID + HID * F + 2 * HID * 4)
    # backward: dH1 = W2^T dZ2 (HID*HID), dW2 (HID*HID), dW1 (HID*F), dW3/dH2 (2*HID*4)
    if kind == "bwd_no_dw1":             # bf16 mode at HID = 256: dW1 = dZ1 X^T runs in k_policy_dw1_bf16, not in k_policy_bwd_bf16
        return 2 * rows * (2 * HID * HID + 2 * HID * 4)
    if kind == "dw1":
        return 2 * rows * HID * F
    return 2 * rows * (2 * HID * HID + HID * F + 2 * HID * 4)


def split_backward_on(dtype):
    """True when the fused backward of this run is the split-fp32 (bf16x6) kernel (csrc/ppo_policy_bwd_x6.hip): fp32
    policies with two hidden layers; PPO_BWD_SPLIT_BF16=0 selects the pure fp32-MFMA kernel."""
    return dtype == "f32" and LAYERS == 2 and os.environ.get("PPO_BWD_SPLIT_BF16", "1") != "0"


def split_backward_mfma_flops_per_state():
    """bf16 MFMA flops the split-fp32 backward EXECUTES per 32-row tile: six piece products per fp32 product for
    dH1 = dZ2 W2 and dW2 += dZ2^T H1, three for dW1 += dZ1^T X (X is exact in bf16; 72 inputs + the ones column padded
    to 96).  The identity-MFMA transposes and the two small fp32 MFMAs are not counted."""
    rows = 4 * QUADS
    return 2 * rows * (6 * HID * HID + 6 * HID * HID + 3 * HID * 96)


def cpu_baseline():
    """The CPU restatement of the Julia path (oracle/, 1 core) timed on a bounded sample of the same workload:
    serial per-step forward/sample/step, serial scan, per-minibatch forward+backward+Adam."""
    from oracle import oracle as orc
    n_env, T, B = 6, 64, 96              # 384 env-steps: ~25 s of scalar C on one core (round 2 timed 192 in 12 s)
    params = orc.glorot_params(F, HID, 2, seed=0)
    env = orc.Env(Q=8, max_actions=T_STEPS, N=n_env, seed=1234)
    env.reset()
    t0 = time.perf_counter()
    ro = orc.collect_rollouts_tn(env, params, HID, T, mode_dev=False)
    ret = orc.compute_returns_tn(ro["rewards"], ro["done"], GAMMA)
    M = n_env * T
    st = ro["states"].reshape(M, 32, F)
    act = ro["active"].reshape(M)
    a0 = ro["actions"].reshape(M)
    po = ro["p_sel"].reshape(M)
    adv = ret.reshape(M)
    m, v, bp = np.zeros_like(params), np.zeros_like(params), np.array([0.9, 0.999])
    rng = np.random.default_rng(0)
    for _ in range(EPOCHS):
        perm = rng.permutation(M)
        for s in range(0, M, B):
            sel = perm[s:s + B]
            g, _, _ = orc.step_batch_grad_f64(params, F, HID, st[sel], act[sel], a0[sel], po[sel], adv[sel], EPS, ENT_W)
            orc.adam_step(params, g.astype(np.float32), m, v, bp, LR)
    dt = time.perf_counter() - t0
    return {"value": M / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "CPU restatement of the Julia path (Julia unavailable): %d envs x %d steps, %d epochs, "
                      "minibatch %d, 2x256 MLP, %.1f s on 1 core" % (n_env, T, EPOCHS, B, dt)}


def cpu_baseline_blas(threads=None):
    """Second CPU figure, closer to what the Julia reference does on a host: the same path with the Dense products and
    their gradients on the host BLAS / autograd (torch CPU, Float32, all threads) instead of the scalar C restatement --
    serial per-env forward + sample + step! in the rollout (src/collect_rollouts.jl:1-24), batched minibatch step
    (src/train.jl:54-84).  Env dynamics and sampling come from the oracle.  Context only: never the headline."""
    import torch
    from oracle import oracle as orc
    from oracle import np_oracle as npo
    torch.set_num_threads(threads or min(16, os.cpu_count() or 1))   # a one-GPU box's CPU share is 16 cores
    # ~10 s of CPU work each (round 2 timed 2 s / 0.7 s samples): 16,384 env-steps with the headline minibatch on 16 threads,
    # 12,288 on one core
    n_env, T, B = (128, 128, 4096) if threads != 1 else (96, 128, 2048)
    params = orc.glorot_params(F, HID, 2, seed=0)
    layers = [(torch.tensor(W, dtype=torch.float32, requires_grad=True), torch.tensor(b, dtype=torch.float32, requires_grad=True))
              for (W, b) in npo.unpack_params(params, F, HID, 2)]
    flat = [t for wb in layers for t in wb]
    opt = torch.optim.Adam(flat, lr=LR)

    def probs_of(x, mask):                                     # x [B,32,F] float32, mask [B,128] (0 / -inf)
        a = x
        for (W, b) in layers[:-1]:
            a = torch.nn.functional.leaky_relu(a @ W.T + b, 0.01)
        W, b = layers[-1]
        return torch.softmax((a @ W.T + b).reshape(x.shape[0], -1) + mask, dim=1)

    env = orc.Env(Q=8, max_actions=T_STEPS, N=n_env, seed=1234)
    env.reset()
    rng = np.random.default_rng(0)
    M = n_env * T
    st = np.zeros((M, 32, F), np.int8); msk = np.zeros((M, 128), np.float32)
    a0 = np.zeros(M, np.int64); po = np.zeros(M, np.float32); rew = np.zeros((T, n_env), np.float32); dn = np.zeros((T, n_env), np.uint8)
    t0 = time.perf_counter()
    with torch.no_grad():
        for t in range(T):
            obs = env.observe(); act = np.array(env.active)
            acts = np.zeros(n_env, np.int32)
            for n in range(n_env):                             # the reference walks one env at a time
                mask = np.repeat(np.where((act[n] >> np.arange(8)) & 1, 0.0, -np.inf).astype(np.float32), 16)
                p = probs_of(torch.from_numpy(obs[n:n + 1].astype(np.float32)), torch.from_numpy(mask[None]))[0].numpy()
                a = min(int(np.searchsorted(np.cumsum(p), rng.random())), 127)
                if not p[a] > 0:
                    a = int(np.flatnonzero(p > 0)[-1])
                k = t * n_env + n
                st[k] = obs[n]; msk[k] = mask; a0[k] = a; po[k] = p[a]; acts[n] = a
            env.step(acts)
            rew[t] = env.reward; dn[t] = env.done
            for n in np.flatnonzero(dn[t]):
                env.reset_one(int(n))
    ret = orc.compute_returns_tn(rew, dn, GAMMA).reshape(M)
    X = torch.from_numpy(st.astype(np.float32)); Mk = torch.from_numpy(msk)
    A0 = torch.from_numpy(a0); PO = torch.from_numpy(po); ADV = torch.from_numpy(ret.astype(np.float32))
    for _ in range(EPOCHS):
        perm = torch.from_numpy(rng.permutation(M))
        for s0 in range(0, M, B):
            sel = perm[s0:s0 + B]
            p = probs_of(X[sel], Mk[sel])
            adv = ADV[sel]
            gain = p[torch.arange(len(sel)), A0[sel]] / PO[sel] * adv
            clip = torch.where(adv >= 0, (1.0 + EPS) * adv, (1.0 - EPS) * adv)
            sp = p + 1e-8 / 128
            loss = -torch.mean(torch.minimum(gain, clip)) + ENT_W * torch.mean((sp * torch.log(sp)).sum(dim=1))
            opt.zero_grad(); loss.backward(); opt.step()
    dt = time.perf_counter() - t0
    return {"value": M / dt, "unit": "env-steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "same path with the MLP on host BLAS/autograd (torch CPU fp32): %d envs x %d steps, %d epochs, "
                      "minibatch %d, %.1f s on %d threads" % (n_env, T, EPOCHS, B, dt, torch.get_num_threads())}


def cpu_baseline_julia():
    """the REFERENCE's own Julia path (its collect_rollouts! + ppo_train!, included unmodified from a
    checkout) against a Julia implementation of the same synthetic env and MLP: bench/julia_reference.jl.  Runs only when
    `julia` is on PATH and PPO_JULIA_REFERENCE names a ProximalPolicyOptimization.jl checkout (with Flux, Distributions,
    BSON, CSV, DataFrames, Tables loadable); returns None otherwise -- the expected case on the build image and the GPU
    box (no julia, no network) -- and the line then carries no `cpu_baseline_julia` key."""
    import shutil
    jl, ref = shutil.which("julia"), os.environ.get("PPO_JULIA_REFERENCE", "")
    if not jl or not os.path.isdir(os.path.join(ref, "src")):
        return None
    r = subprocess.run([jl, "--threads=1", os.path.join(ROOT, "bench", "julia_reference.jl"), ref, "64", str(T_STEPS), str(EPOCHS), "4096"],
                       capture_output=True, text=True, timeout=float(os.environ.get("PPO_JULIA_TIMEOUT", "1500")))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"value": None, "error": (r.stderr or r.stdout)[-400:]}
    return json.loads(lines[-1])


def cpu_baseline_all_cores():
    """the restatement on all host cores -- one replica of the 1-core sample per thread, each on
    its own env shard (the C restatement releases the GIL), no gradient exchange between them: an upper bound on what
    partitioning the env batch over `cores` threads can give."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    cores = min(os.cpu_count() or 1, 16)                      # a one-GPU box's CPU share is 16 cores
    n_env, T, B = 2, 32, 32

    def replica(k):
        params = orc.glorot_params(F, HID, 2, seed=0)
        env = orc.Env(Q=8, max_actions=T_STEPS, N=n_env, seed=1234, global_offset=k * n_env)
        env.reset()
        ro = orc.collect_rollouts_tn(env, params, HID, T, mode_dev=False)
        ret = orc.compute_returns_tn(ro["rewards"], ro["done"], GAMMA)
        M = n_env * T
        st, act, a0 = ro["states"].reshape(M, 32, F), ro["active"].reshape(M), ro["actions"].reshape(M)
        po, adv = ro["p_sel"].reshape(M), ret.reshape(M)
        m, v, bp = np.zeros_like(params), np.zeros_like(params), np.array([0.9, 0.999])
        rng = np.random.default_rng(k)
        for _ in range(EPOCHS):
            perm = rng.permutation(M)
            for s0 in range(0, M, B):
                sel = perm[s0:s0 + B]
                g, _, _ = orc.step_batch_grad_f64(params, F, HID, st[sel], act[sel], a0[sel], po[sel], adv[sel], EPS, ENT_W)
                orc.adam_step(params, g.astype(np.float32), m, v, bp, LR)
        return M

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(replica, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "CPU restatement, %d independent replicas (one thread each, own env shard, no gradient exchange) of "
                      "%d envs x %d steps, %d epochs, minibatch %d, %.1f s" % (cores, n_env, T, EPOCHS, B, dt)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here, one per GPU.  This parent never
    touches the GPU (no HIP call, no torch.cuda call), so nothing is exec'ed from a GPU-initialised process.  Rank 0's
    stdout is forwarded (the one JSON line); any rank failing ends the others and the run exits non-zero."""
    import tempfile
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else sys.stderr.fileno()))
        deadline = time.time() + float(os.environ.get("PPO_BENCH_LAUNCH_TIMEOUT", "1500"))
        rc = 0
        while any(p.poll() is None for p in procs):
            failed = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if failed or time.time() > deadline:
                rc = failed[0] if failed else 124
                for p in procs:                               # end exactly the processes started above
                    if p.poll() is None:
                        p.kill()
                break
            time.sleep(0.1)
        for p in procs:
            p.wait()
            rc = rc or p.returncode
        out0.seek(0)
        sys.stdout.write(out0.read().decode("utf-8", "replace"))
        sys.stdout.flush()
    if rc:
        sys.stderr.write("bench.py: a rank process failed (exit %d); no multi-GPU figure was produced\n" % rc)
    return rc


def main():
    global T_STEPS, EPOCHS, N_ENVS, MINIBATCH, QUADS, HID, LAYERS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--t-steps", type=int, default=T_STEPS, help="profiling only: shorter rollout (flagged in the output)")
    ap.add_argument("--epochs", type=int, default=EPOCHS, help="profiling only: fewer epochs (flagged in the output)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (the headline line); bf16 = the config-5 arithmetic on the same workload "
                         "(flagged in the output, not the headline)")
    ap.add_argument("--envs", type=int, default=N_ENVS, help="envs per GPU (default 4096 = the headline workload)")
    ap.add_argument("--stream", metavar="DIR", default=None,
                    help="stream every rollout to DIR/rollout.bin while it is collected (DiskRollouts: device -> pinned host -> "
                         "file; BASELINE config 5 'rollouts streamed to disk'); training reads the resident copy.  Not the headline")
    ap.add_argument("--hid", type=int, choices=[128, 256], default=HID,
                    help="hidden width: 256 = the headline 2x256 MLP, 128 = the reference's own Policy(72,128,2,4) (not the headline)")
    ap.add_argument("--layers", type=int, choices=[1, 2, 3, 4], default=LAYERS,
                    help="hidden layers of the policy: 2 = the headline MLP; 1, 3, 4 run the layer-looped kernels (not the headline)")
    ap.add_argument("--quads", type=int, choices=[8, 32], default=QUADS,
                    help="quad slots per env: 8 = the headline rand-poly shape (A=128), 32 = the square-mesh-sized action "
                         "space of BASELINE config 4 (A=512, variable-length masked episodes; flagged in the output)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, BASELINE config 3): 4096 envs and a 4096-sample minibatch PER GPU; strong: the 4096 envs "
                         "and the 4096-sample global minibatch are split over the GPUs (flagged in the output)")
    args = ap.parse_args()
    reduced = (args.t_steps != T_STEPS) or (args.epochs != EPOCHS)
    T_STEPS, EPOCHS = args.t_steps, args.epochs
    nonheadline = (args.dtype != "f32") or (args.envs != N_ENVS) or (args.scaling != "weak") or (args.quads != QUADS) or (args.hid != HID) or bool(args.stream) or (args.layers != LAYERS)
    HID = args.hid
    LAYERS = args.layers
    QUADS = args.quads
    N_ENVS = MINIBATCH = args.envs

    # ---- ranks: a launcher's (torchrun) environment wins; otherwise --gpus N > 1 starts the N ranks itself
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if N_ENVS % world:
        raise SystemExit("the env count must divide by the number of GPUs (strong-scaling leg)")
    # rehearsal switches (tests): PPO_BENCH_BACKEND=gloo runs the exchange over gloo (host round trip of the gradient
    # buffer), PPO_BENCH_SHARE_GPU=1 puts every rank on device 0 (a one-GPU box), PPO_BENCH_DRYRUN=1 exercises the rank
    # launch, the rendezvous and the report only (no GPU at all: the CPU test of the launcher)
    backend = os.environ.get("PPO_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("PPO_BENCH_SHARE_GPU") == "1"
    dry = os.environ.get("PPO_BENCH_DRYRUN") == "1"
    dev_index = 0 if share_gpu else local_rank

    import ctypes as C
    import ppo_amd as PPO
    dist = None
    torch = None
    # PPO_BENCH_FORCE_DIST=1 rehearses the whole distributed path (RCCL group, stream hand-over, all-reduce hook)
    # with a single rank on a one-GPU box
    use_dist = world > 1 or os.environ.get("PPO_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch
        import torch.distributed as dist
        kw = {}
        if not dry:
            torch.cuda.set_device(dev_index)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", dev_index)
        if "MASTER_ADDR" in os.environ:
            dist.init_process_group(backend, **kw)
        else:
            dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, **kw)
        if dist.get_world_size() != world:
            raise SystemExit("process group size %d != WORLD_SIZE %d" % (dist.get_world_size(), world))
    if dry:
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "env-steps/sec end-to-end PPO (rollout+GAE+update), 4096 envs, 1/2/4/8 MI355X",
                              "value": None, "unit": "env-steps/s", "n_gpus": dist.get_world_size(), "dry_run": True,
                              "steps": args.steps, "warmup": args.warmup}))
        dist.destroy_process_group()
        return 0
    PPO._lib.call("ppo_device_init", dev_index)
    if use_dist:
        PPO._lib.call("ppo_set_stream", C.c_void_p(torch.cuda.current_stream().cuda_stream))

    if args.stream:
        PPO.set_disk_async(os.environ.get("PPO_DISK_ASYNC", "1") != "0")     # deferred finish of the streamed file
    dp = PPO.DataParallel(rank, world, force_hook=use_dist)
    pol = PPO.HipPolicy(F, HID, LAYERS, 4, seed=0, dtype=args.dtype)
    opt = PPO.Optimiser(PPO.Adam(LR))

    def sync():
        PPO.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(n_envs, minibatch, seed_base):
        """W warm-up + K timed PPO iterations on `n_envs` envs per rank; returns (max-over-ranks seconds, iteration fn)."""
        env = PPO.HipVecEnv(num_envs=n_envs, Q=QUADS, max_actions=T_STEPS, seed=1234, global_offset=rank * n_envs)
        ro = PPO.BufferRollouts()
        stream_pending = []

        def iteration(i):
            if args.stream:                              # a fresh DiskRollouts per iteration, as src/train.jl:185 does
                dro = PPO.DiskRollouts(os.path.join(args.stream, "rank%d" % rank))
                PPO.collect_rollouts_steps_(dro, env, pol, T_STEPS, GAMMA)
                ds = PPO.construct_dataset(dro._device)
                stream_pending.append(dro)
            else:
                PPO.collect_rollouts_steps_(ro, env, pol, T_STEPS, GAMMA)
                ds = PPO.construct_dataset(ro)
            PPO.ppo_train_(pol, opt, ds, EPS, minibatch, EPOCHS, ENT_W, seed=seed_base + i, parallel=dp, verbose=False)
            while stream_pending:                        # deferred finish (PPO_DISK_ASYNC=1): the file of THIS iteration is complete
                PPO.disk_sync(stream_pending.pop())      # before the iteration counts as done (inside the timed region)

        for i in range(args.warmup):
            iteration(i)
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            iteration(args.warmup + i)
        sync()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, iteration

    strong_requested = args.scaling == "strong"
    if strong_requested:                                 # the whole line is the strong-scaling figure
        N_ENVS = MINIBATCH = N_ENVS // world
    dt, iteration = timed_run(N_ENVS, MINIBATCH, 1000)

    # ---- roofline leg: per-kernel HIP-event durations of one more iteration (outside the timed region)
    roof, kernels = None, {}
    peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    if rank == 0:
        PPO.profile_enable(True)
        iteration(args.warmup + args.steps)
        PPO.synchronize()
        # k_rollout_persistent: the whole rollout (T steps of every env: observe + MLP + sample + step!) in one launch
        # (the default for Q = 8 in fp32 mode); k_policy_fwd_rollout is the per-step form (bf16 mode, Q = 32, disk
        # streaming, PPO_ROLLOUT_PERSISTENT=0)
        # bf16 mode at HID = 256 runs the backward as a PAIR of launches (k_policy_bwd_bf16 + k_policy_dw1_bf16): the
        # per-kernel rows carry each kernel's own flops, the roofline line carries the pair (sum of times, full flops)
        dw1_ms, dw1_n = PPO.profile_get("k_policy_dw1")
        split_dw1 = dw1_n > 0
        for name, kind, per in [("k_policy_bwd", "bwd_no_dw1" if split_dw1 else "bwd", MINIBATCH), ("k_policy_dw1", "dw1", MINIBATCH),
                                ("k_policy_fwd_train", "fwd", MINIBATCH),
                                ("k_policy_fwd_rollout", "fwd", N_ENVS), ("k_rollout_persistent", "fwd", N_ENVS * T_STEPS)]:
            ms, n = PPO.profile_get(name)
            if n:
                avg = ms / n
                if name == "k_rollout_persistent":
                    per = per / n                        # streamed to disk the rollout is a chain of launches, T / n steps each
                tf = flops_per_state(kind) * per / (avg * 1e-3) / 1e12
                kernels[name] = {"avg_ms": round(avg, 4), "launches": n, "tflops": round(tf, 2),
                                 "frac": round(tf / peak, 4)}
        bd_ms, bd_n = PPO.profile_get("k_policy_bwd_data")
        wg_ms, wg_n = PPO.profile_get("k_policy_wgrad")
        if bd_n and wg_n and "k_policy_bwd" not in kernels:      # three-product backward (deep policies, small minibatches): the pair
            pair_ms = bd_ms / bd_n + wg_ms / wg_n
            tf = flops_per_state("bwd") * MINIBATCH / (pair_ms * 1e-3) / 1e12
            kernels["k_policy_bwd_data+k_policy_wgrad"] = {"avg_ms": round(pair_ms, 4), "launches": bd_n, "tflops": round(tf, 2), "frac": round(tf / peak, 4)}
        if split_dw1 and "k_policy_bwd" in kernels:
            pair_ms = kernels["k_policy_bwd"]["avg_ms"] + kernels["k_policy_dw1"]["avg_ms"]
            tf = flops_per_state("bwd") * MINIBATCH / (pair_ms * 1e-3) / 1e12
            kernels["k_policy_bwd+k_policy_dw1"] = {"avg_ms": round(pair_ms, 4), "launches": kernels["k_policy_bwd"]["launches"],
                                                    "tflops": round(tf, 2), "frac": round(tf / peak, 4)}
        for name in ("k_policy_train_tile", "k_policy_bwd_data", "k_policy_wgrad", "k_returns_tn", "k_env_step", "k_env_observe", "k_grad_reduce", "k_adam",
                     "k_reduce_adam", "allreduce"):
            ms, n = PPO.profile_get(name)
            if n:
                kernels[name] = {"avg_ms": round(ms / n, 4), "launches": n}
        PPO.profile_enable(False)
        # K6 return scan at the config-5 size (65536 envs x 128 steps = 8.4 M transitions): HBM-bound, 9 B/transition
        ms = PPO.profile_returns(128, 65536, GAMMA, 20)
        gbs = 9.0 * 128 * 65536 / (ms * 1e-3) / 1e9
        kernels["k_returns_tn@65536x128"] = {"avg_ms": round(ms, 4), "GB/s": round(gbs, 1),
                                             "frac_of_hbm_8TBs": round(gbs / 8000.0, 4)}
        # the same scan on 4x the columns: at the config-5 size a 20 us launch is half ramp-up (75 MB over 256 CUs)
        ms = PPO.profile_returns(128, 262144, GAMMA, 10)
        gbs = 9.0 * 128 * 262144 / (ms * 1e-3) / 1e9
        kernels["k_returns_tn@262144x128"] = {"avg_ms": round(ms, 4), "GB/s": round(gbs, 1),
                                              "frac_of_hbm_8TBs": round(gbs / 8000.0, 4)}
        # GAE(gamma, lambda) scan at the same size: 17 B/transition (r f32 + done u8 + V f32 in, adv f32 + lambda-return f32 out)
        for cols, it in ((65536, 20), (262144, 10)):
            ms = PPO.profile_gae(128, cols, 0.99, 0.95, it)
            gbs = 17.0 * 128 * cols / (ms * 1e-3) / 1e9
            kernels["k_gae_tn@%dx128" % cols] = {"avg_ms": round(ms, 4), "GB/s": round(gbs, 1),
                                                  "frac_of_hbm_8TBs": round(gbs / 8000.0, 4)}
        # the split-fp32 backward: the same iteration once more through the pure fp32-MFMA kernel (outside the timed region),
        # so the line carries both forms of the dominant kernel measured in the same process
        fp32_form = None
        split_on = split_backward_on(args.dtype) and "k_policy_bwd" in kernels
        if split_on and not use_dist:
            PPO.set_bwd_split_bf16(False)
            PPO.profile_enable(True)
            iteration(args.warmup + args.steps + 1)
            PPO.synchronize()
            ms, n = PPO.profile_get("k_policy_bwd")
            fms, fn = PPO.profile_get("k_policy_fwd_train")
            PPO.profile_enable(False)
            PPO.set_bwd_split_bf16(None)
            if n:
                tf = flops_per_state("bwd") * MINIBATCH / (ms / n * 1e-3) / 1e12
                fp32_form = {"kernel": "k_policy_bwd<F,HID> (v_mfma_f32_32x32x2_f32: ppo_set_bwd_split_bf16(0))", "avg_ms": round(ms / n, 4),
                             "launches": n, "tflops": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}
                if fn:
                    ftf = flops_per_state("fwd") * MINIBATCH / (fms / fn * 1e-3) / 1e12
                    fp32_form["train_forward"] = {"kernel": "k_policy_fwd<F,HID,MODE 2> (fp32 MFMA, one wave per state)", "avg_ms": round(fms / fn, 4),
                                                  "tflops": round(ftf, 2), "frac": round(ftf / PEAK_FP32_MFMA_TFLOPS, 4)}
        if split_on:
            rows = 4 * QUADS
            # bf16 MFMA flops the split kernels EXECUTE per algorithmic flop: backward see split_backward_mfma_flops_per_state;
            # train forward: layer 1 three piece products (X exact in bf16, 72 inputs padded to 80), layer 2 six, layer 3 on the VALU
            for name, ratio in (("k_policy_bwd", split_backward_mfma_flops_per_state() / flops_per_state("bwd")),
                                ("k_policy_fwd_train", 2 * rows * (3 * HID * 80 + 6 * HID * HID) / flops_per_state("fwd"))):
                kk = kernels.get(name)
                if not kk or (name == "k_policy_fwd_train" and ((QUADS == 32 and HID != 256) or os.environ.get("PPO_FWD_SPLIT_MAX_TILES") == "0")):
                    continue
                kk["executed_bf16_mfma_tflops"] = round(kk["tflops"] * ratio, 1)
                kk["frac"] = round(kk["tflops"] * ratio / PEAK_BF16_MFMA_TFLOPS, 4)       # of the pipe that bounds it
                kk["algorithmic_over_fp32_mfma_peak"] = round(kk["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
                kk["form"] = "split-fp32 (bf16x6) on the bf16 MFMA pipe"
        k = kernels.get("k_policy_bwd+k_policy_dw1") or kernels.get("k_policy_bwd") or kernels.get("k_policy_bwd_data+k_policy_wgrad")
        if k:
            # HBM bytes per launch of the dominant kernel come from the committed PMC passes (rocprofv3 cannot run
            # inside this process); null when no pass was made for this exact launch shape
            traffic, tsrc = None, None
            try:
                # PPO_PMC_TRAFFIC_FILE: the measurement bundle points this at the PMC passes it has just made, so every line
                # of a bundle carries its own bundle's figure (nothing is filled in afterwards)
                pfile = os.environ.get("PPO_PMC_TRAFFIC_FILE") or os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
                pm = json.load(open(pfile))
                key = "%s|envs=%d|quads=%d|hid=%d" % (args.dtype, MINIBATCH, QUADS, HID) + ("" if LAYERS == 2 else "|layers=%d" % LAYERS)
                ent = pm["launch_shapes"].get(key)
                if ent:
                    traffic = ent["k_policy_bwd_hbm_bytes"]
                    if split_dw1 and ent.get("k_policy_dw1_hbm_bytes"):
                        traffic += ent["k_policy_dw1_hbm_bytes"]          # the roofline line is the backward PAIR
                    tsrc = "%s [%s] (%s)" % (os.path.relpath(pfile, ROOT), key, pm["source"])
            except Exception:
                pass
            tiles = MINIBATCH * (QUADS // 8)
            roof = {"bound": "mfma", "kernel": "k_policy_bwd+k_policy_dw1 (the backward is two launches in this mode)" if split_dw1 else
                    ("k_policy_bwd" if "k_policy_bwd" in kernels else "k_policy_bwd_data+k_policy_wgrad (three-product backward)"),
                    "achieved": k["tflops"],
                    # split-fp32 backward: `achieved` stays the ALGORITHMIC fp32 flops per second; the pipe that bounds the
                    # kernel is the bf16 MFMA, which executes `mfma_flops_per_algorithmic_flop` flops per algorithmic flop,
                    # so the peak of this algorithm in algorithmic flops is the bf16 dense peak divided by that ratio
                    "peak": (round(PEAK_BF16_MFMA_TFLOPS * flops_per_state("bwd") / split_backward_mfma_flops_per_state(), 1) if split_on else peak),
                    "unit": "TFLOP/s", "frac": k["frac"], "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                    "traffic_source": tsrc,
                    "form": (None if not split_on else
                             "split-fp32 (bf16x6) on v_mfma_f32_32x32x16_bf16: every fp32 operand is the exact sum of three bf16 pieces, "
                             "six piece products with fp32 accumulation per fp32 product (three where X is exact in bf16); peak = %.0f TFLOP/s "
                             "bf16 dense / %.2f executed MFMA flops per algorithmic flop; the same kernel's algorithmic rate is %.2fx the "
                             "fp32-MFMA peak of %.1f TFLOP/s" % (PEAK_BF16_MFMA_TFLOPS, split_backward_mfma_flops_per_state() / flops_per_state("bwd"),
                                                               k["tflops"] / PEAK_FP32_MFMA_TFLOPS, PEAK_FP32_MFMA_TFLOPS)),
                    "fp32_mfma_form": fp32_form,
                    # context for `frac` (profiles/history/r01_mfma_f32_valu_overlap.txt): on gfx950 the fp32 MFMA
                    # runs on the packed-fp32 vector ALU and vector instructions do not hide under it
                    "ceiling_note": None if (args.dtype != "f32" or split_on) else
                    "pure v_mfma_f32_32x32x2_f32 chain measured 140-143 TFLOP/s on this chip (clock under load); every "
                    "vector instruction between MFMAs adds ~2 ns per SIMD: instruction-mix ceiling of this kernel "
                    "(2432 MFMA + ~5250 vector instr per tile) ~125-130 TFLOP/s",
                    "algorithmic_flop_per_launch": flops_per_state("bwd") * MINIBATCH,
                    # TRUE minimum of the backward per launch: per 32-row tile the two saved activation tiles, the state
                    # rows and dY in (fp32: 4 B, bf16: 2 B per activation), ONE gradient out.  The per-workgroup gradient
                    # slabs (written here, re-read by the reduction) are design traffic and listed separately.
                    # (bf16 mode: ONE saved tile -- the kernel recomputes H1 from the state rows; what it hands to the dW1
                    # kernel at HID = 256, dZ1^T as bf16 fragments, is design traffic too)
                    "algorithmic_hbm_bytes_per_launch": tiles * ((2 * HID * 32 * 4 if args.dtype == "f32" else HID * 32 * 2) + 32 * F + 32 * 16)
                    + 4 * (HID * HID + HID * F + HID * 6 + 4),
                    "design_hbm_bytes_per_launch": dict(
                        {"gradient_slabs_written": min(256 if (HID == 256 or args.dtype == "bf16") else 512, tiles) * 4 * (HID * HID + HID * 96 + HID * 6 + 4)},
                        **({"dz1_fragments_written": tiles * HID * 32 * 2} if args.dtype == "bf16" and HID > 128 else {}))}
    elif use_dist:
        iteration(args.warmup + args.steps)          # keep the collectives of the extra iteration matched
        PPO.synchronize()

    # ---- strong-scaling leg (N > 1): the 4096 envs and the 4096-sample global minibatch split over the ranks
    strong = None
    if world > 1 and not strong_requested:
        sdt, _ = timed_run(N_ENVS // world, MINIBATCH // world, 5000)
        strong = {"value": N_ENVS * T_STEPS * args.steps / sdt, "unit": "env-steps/s", "ms_per_step": sdt / args.steps * 1e3,
                  "envs_total": N_ENVS, "envs_per_gpu": N_ENVS // world, "minibatch_global": MINIBATCH,
                  "minibatch_per_gpu": MINIBATCH // world,
                  "note": "same workload as n_gpus=1 (4096 envs, global minibatch 4096) split over the ranks; "
                          "strong-scaling efficiency = this value / (n_gpus x the n_gpus=1 value)"}

    if rank == 0:
        comm = PPO.rccl_comm_info() if use_dist else None
        value = world * N_ENVS * T_STEPS * args.steps / dt
        out = {
            "metric": "env-steps/sec end-to-end PPO (rollout+GAE+update), 4096 envs, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s",
            # read back from the communicator, not from the launcher's environment
            "n_gpus": (comm[1] if comm else (dist.get_world_size() if use_dist else 1)),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%d parallel synthetic rand-poly-shaped envs per GPU (Q=%d,H=%d,A=%d,F=72 int8), "
                                   "%dx%d MLP policy %s, T=%d steps/iteration, %d epochs, minibatch %d/GPU, "
                                   "gamma=1.0 eps=0.05 entropy_w=0.01 Adam 1e-4; returns mode (lambda=1,V=0)"
                                   % (N_ENVS, QUADS, 4 * QUADS, 16 * QUADS, LAYERS, HID, "fp32" if args.dtype == "f32" else "bf16 MFMA / fp32 accumulate (config 5 arithmetic)",
                                      T_STEPS, EPOCHS, MINIBATCH),
                       "envs_per_gpu": N_ENVS, "T": T_STEPS, "epochs": EPOCHS, "minibatch_per_gpu": MINIBATCH,
                       "parallelism": "dp%d" % world,
                       "rollouts_streamed_to_disk": bool(args.stream),
                       # fp32 data, fp32 accumulation everywhere; the fused backward's three big products run as split-fp32
                       # (three exact bf16 pieces per operand, six piece products) on the bf16 matrix pipe unless switched off
                       "training_pass_products": ("split-fp32 (bf16x6) on the bf16 MFMA pipe, fp32 accumulate (train forward + backward; rollouts: fp32 MFMA)" if split_backward_on(args.dtype)
                                             else ("fp32 MFMA" if args.dtype == "f32" else "bf16 MFMA"))},
            "allreduce": (dp.hook_kind if use_dist else None), "rccl_ranks": (comm[1] if comm else None),
            "strong": strong,
            "roofline": roof, "kernels": kernels, "reduced_profiling_run": reduced, "headline_config": not nonheadline,
            "target_frac_of_1e6": value / 1e6,
        }
        if not args.no_cpu_baseline and world == 1:
            for key, fn in (("cpu_baseline", cpu_baseline), ("cpu_baseline_blas", cpu_baseline_blas),
                            ("cpu_baseline_blas_1core", lambda: cpu_baseline_blas(threads=1)),
                            ("cpu_baseline_all_cores", cpu_baseline_all_cores), ("cpu_baseline_julia", cpu_baseline_julia)):
                try:
                    res = fn()
                    if res is not None:     # cpu_baseline_julia: absent unless julia + a reference checkout are present
                        out[key] = res
                except Exception as e:      # the oracle is only a reported baseline
                    out[key] = {"value": None, "error": str(e)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        PPO.rccl_finalize()                        # no-op unless the in-library communicator was created
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
