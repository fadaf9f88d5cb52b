#!/usr/bin/env python3
"""Usage example: what the reference's driver scripts do (test/test_square_mesh.jl:9-33, test/random_quad.jl:40-65,
examples/triangle/distance_weighted/triangle_utilities.jl:352-387), on the MI355X engine.

    evaluator = SaveBestModel(...)             # callable (policy, env, optimizer), run at the top of every iteration
    PPO.ppo_iterate_(policy, env, optimizer, episodes_per_iteration, minibatch_size, num_ppo_iterations, evaluator,
                     epochs_per_iteration, discount, epsilon, entropy_weight)         # src/train.jl:210-222

The env is the built-in synthetic rand-poly-shaped batched env (QuadMeshGame is not in the reference tree); the
policy is SimplePolicy.Policy(72, 128, 2, 4) like the reference's trained fixtures.  The best policy is saved as a
BSON.jl document the reference can `BSON.@load`.
Run on the GPU box:  python examples/train_rand_poly.py --iterations 20 --out /tmp/best_policy.bson
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ppo_amd as PPO  # noqa: E402


class SaveBestModel:
    """Evaluator with the reference's 3-argument signature (src/train.jl:181,226): average return of the stochastic
    policy over `num_trajectories` episodes; keeps the best policy on disk (triangle_utilities.jl:352-387)."""

    def __init__(self, root_dir_file, num_trajectories=1000, eval_envs=1024, seed=99):
        self.file_path = root_dir_file
        self.num_trajectories = num_trajectories
        self.eval_env = PPO.HipVecEnv(num_envs=eval_envs, Q=8, max_actions=32, seed=seed)
        self.mean_returns, self.std_returns, self.best_return = [], [], -float("inf")
        self.loss = None

    def __call__(self, policy, env, optimizer):
        ret, dev = PPO.average_returns(policy, self.eval_env, self.num_trajectories)
        print("RET = %1.4f\tDEV = %1.4f" % (ret, dev))
        if ret > self.best_return:
            self.best_return = ret
            print("\tNEW BEST RETURN : %1.4f -> %s" % (ret, self.file_path))
            PPO.save_policy(self.file_path, policy)
        self.mean_returns.append(ret)
        self.std_returns.append(dev)


@PPO.save_loss.register(SaveBestModel)          # the reporting plugin (src/ProximalPolicyOptimization.jl:30)
def _(evaluator, loss):
    evaluator.loss = loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "best_policy.bson"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)

    # hyper-parameters in the style of test/random_quad.jl:40-50
    discount, epsilon, entropy_weight = 1.0, 0.05, 0.01
    minibatch_size, epochs_per_iteration = 1024, 4
    episodes_per_iteration = 4 * args.envs                   # four whole episodes per resident env

    env = PPO.HipVecEnv(num_envs=args.envs, Q=8, max_actions=32, seed=7)
    policy = PPO.HipPolicy(72, 128, 2, 4, seed=0, dtype=args.dtype)
    optimizer = PPO.Optimiser(PPO.Adam(3e-4))                # an iterable composite, like Flux.Optimiser(Adam(...))
    evaluator = SaveBestModel(args.out)
    PPO.ppo_iterate_(policy, env, optimizer, episodes_per_iteration, minibatch_size, args.iterations, evaluator,
                     epochs_per_iteration, discount, epsilon, entropy_weight, verbose=False)
    evaluator(policy, env, optimizer)
    best = PPO.load_policy(args.out)                          # BSON.@load path policy
    print("best average return %.3f (first %.3f); checkpoint holds Policy(%d, %d, %d, %d)"
          % (evaluator.best_return, evaluator.mean_returns[0], best.in_channels, best.hidden_channels,
             best.num_hidden_layers, best.num_output))
    return evaluator


if __name__ == "__main__":
    main()
