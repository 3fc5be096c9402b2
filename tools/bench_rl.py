#!/usr/bin/env python3
"""RL-style loop on device buffers (BASELINE config 5 shape): observation + action mask -> masked policy in
PyTorch-ROCm -> step(actions).  Default: a masked random policy (env-side cost only).  `--learner`: a masked
actor-critic MLP (368 -> 512 -> 512 -> 9601 logits + value head, bf16 autocast) sampled every step and updated with
Adam on n-step returns every `--horizon` steps - the end-to-end steps/s of an on-device training loop (no PPO library
is installed in the image; this is the same data flow: rollout, masked log-probs, advantage, backward, optimizer).
The env writes into / reads from torch CUDA tensors through `io_device=1` (torch.Tensor.data_ptr()), nothing crosses PCIe inside the loop.

    python tools/bench_rl.py [--batch 16384] [--steps 200] [--learner [--horizon 16]]
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
ap.add_argument("--learner", action="store_true")
ap.add_argument("--horizon", type=int, default=16)
args = ap.parse_args()
B = args.batch
wl = bench.WORKLOADS["nsfnet320"]
env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(), batch_size=B,
                      num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000, auto_reset=True,
                      load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"], io_device=True)
c = env.holder.struct
obs_dim, nact = 3 + c.k_paths + c.k_paths * c.n_mods_consider * 12, c.k_paths * c.n_mods_consider * c.n_slots + 1
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


if args.learner:
    torch.manual_seed(0)
    body = torch.nn.Sequential(torch.nn.Linear(obs_dim, 512), torch.nn.Tanh(), torch.nn.Linear(512, 512), torch.nn.Tanh()).to(dev)
    pi_head, v_head = torch.nn.Linear(512, nact).to(dev), torch.nn.Linear(512, 1).to(dev)
    params = list(body.parameters()) + list(pi_head.parameters()) + list(v_head.parameters())
    opt = torch.optim.Adam(params, lr=3e-4)
    r_off = nat.STEP_DTYPE.fields["reward"][1]
    logps, values, rewards = [], [], []

    def rl_step():   # noqa: F811
        env._check(env.lib.ongym_observe(env._h, obs.data_ptr(), mask.data_ptr()), "observe")
        env.sync()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = body(obs)
            logits, v = pi_head(h).float(), v_head(h).float().squeeze(1)
        logits = logits.masked_fill(mask == 0, -1e9)
        dist_ = torch.distributions.Categorical(logits=logits)
        a = dist_.sample()
        actions.copy_(a.to(torch.int32))
        torch.cuda.synchronize()
        env._check(env.lib.ongym_step_actions(env._h, actions.data_ptr(), recs.data_ptr()), "step")
        env.sync()
        logps.append(dist_.log_prob(a)); values.append(v)
        rewards.append(recs[:, r_off:r_off + 8].contiguous().view(torch.float64).squeeze(1).float())
        if len(rewards) == args.horizon:
            ret, rets = torch.zeros(B, device=dev), []
            for r in reversed(rewards):
                ret = r + 0.99 * ret
                rets.append(ret)
            rets = torch.stack(rets[::-1]); vs = torch.stack(values); lp = torch.stack(logps)
            adv = (rets - vs).detach()
            loss = -(lp * adv).mean() + 0.5 * (rets - vs).pow(2).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            logps.clear(); values.clear(); rewards.clear()

for _ in range(5 if not args.learner else args.horizon):
    rl_step()
t0 = time.perf_counter()
for _ in range(args.steps):
    rl_step()
dt = time.perf_counter() - t0
acc = recs.cpu().numpy().view(nat.STEP_DTYPE)["accepted"].mean()
what = "observe + actor-critic MLP forward/sample + step + Adam update every %d steps" % args.horizon if args.learner \
    else "observe + masked sampling in torch + step"
print(f"B={B}: {B * args.steps / dt:.3e} RL env-steps/s ({what}), "
      f"{dt / args.steps * 1e3:.2f} ms per batched step, accepted {acc:.3f}")
