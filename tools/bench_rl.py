#!/usr/bin/env python3
"""RL-style loop on device buffers (BASELINE config 5 shape, without a learner): observation + action mask ->
masked random policy in PyTorch-ROCm -> step(actions). The env writes into / reads from torch CUDA tensors through
`io_device=1` (torch.Tensor.data_ptr()), nothing crosses PCIe inside the loop.

    python tools/bench_rl.py [--batch 16384] [--steps 200]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
import torch  # noqa: E402
import bench  # noqa: E402
from optical_networking_gym import _native as nat  # noqa: E402
from optical_networking_gym.envs.batched import BatchedQRMSAEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=300)
args = ap.parse_args()
B = args.batch
wl = bench.WORKLOADS["nsfnet320"]
env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(), batch_size=B,
                      num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000, auto_reset=True,
                      load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"], io_device=True)
c = env.holder.struct
obs_dim, nact = 3 + c.k_paths + c.k_paths * c.n_mods * 12, c.k_paths * c.n_mods * c.n_slots + 1
dev = torch.device("cuda", 0)
obs = torch.empty((B, obs_dim), dtype=torch.float32, device=dev)
mask = torch.empty((B, nact), dtype=torch.uint8, device=dev)
actions = torch.empty(B, dtype=torch.int32, device=dev)
recs = torch.empty((B, nat.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
env.seed(1)
env.reset()
env._check(env.lib.ongym_step_policy(env._h, 0, args.warmup, None), "warmup")   # fill the network with first fit
env.sync()


def rl_step():
    env._check(env.lib.ongym_observe(env._h, obs.data_ptr(), mask.data_ptr()), "observe")
    env.sync()
    a = torch.multinomial(mask.float(), 1).squeeze(1).to(torch.int32)           # masked random policy
    actions.copy_(a)
    torch.cuda.synchronize()
    env._check(env.lib.ongym_step_actions(env._h, actions.data_ptr(), recs.data_ptr()), "step")
    env.sync()


for _ in range(5):
    rl_step()
t0 = time.perf_counter()
for _ in range(args.steps):
    rl_step()
dt = time.perf_counter() - t0
acc = recs.cpu().numpy().view(nat.STEP_DTYPE)["accepted"].mean()
print(f"B={B}: {B * args.steps / dt:.3e} RL env-steps/s (observe + masked sampling in torch + step), "
      f"{dt / args.steps * 1e3:.2f} ms per batched step, accepted {acc:.3f}")
